"""CPU-side checks of the device header (polycap_amd/csrc/hip/pc_device.h compiled as host code by
tests/emul): the same per-photon logic the kernels run, compared with the oracle and with the reference's
known answers.  The emulation is test tooling only; the product has no CPU trace path."""
import numpy as np
import pytest

from tests.common import synthetic_constants, PIN_E, make_pair, rel, PIN_AMU, PIN_SCATF


@pytest.fixture(scope="module")
def emul():
    from tests.emul import pyemul
    pyemul.lib()
    return pyemul


def test_known_answers_launch(emul, oracle, known):
    optic, src, prob, _ = make_pair(oracle, "ellip")
    l = known["launch"]
    starts = np.array([c["start"] for c in l["cases"]], dtype=np.float64)
    dirs = np.array([c["dir"] for c in l["cases"]], dtype=np.float64)
    ev = np.tile(np.array(l["start_elecv"], dtype=np.float64), (len(starts), 1))
    for literal in (False, True):
        r = emul.launch_batch(prob, starts, dirs, ev, literal=literal)
        assert r["rc"].tolist() == [c["rc"] for c in l["cases"]]
    # reference tests/capil.c:438-450 (trace with one reflection) seen through a launch
    r = emul.launch_batch(prob, [[0, 0, 0]], [[3e-5, 3e-5, 0.999]], [[0.5, 0.5, 0.]])
    o = oracle.launch_one(optic, [10.0], [PIN_AMU], [PIN_SCATF], (0, 0, 0), (3e-5, 3e-5, 0.999), (0.5, 0.5, 0.))
    assert r["rc"][0] == o["rc"] == 1 and r["i_refl"][0] == o["i_refl"]
    assert rel(r["weights"][0], o["weights"]).max() < 1e-9


def test_conical_17keV_known_answer(emul, oracle, known):
    c = known["launch"]["conical_17keV"]
    from polycap_amd import Problem
    rint_down = c["rad_int_upstream"] * (c["rad_ext_downstream"] / c["rad_ext_upstream"])
    optic = oracle.Optic.from_shape(c["type"], c["length"], c["rad_ext_upstream"], c["rad_ext_downstream"],
                                    c["rad_int_upstream"], rint_down, c["focal_dist_upstream"], c["focal_dist_downstream"],
                                    c["sig_rough"], c["n_cap"], 2.23)
    prob = Problem(optic.z, optic.cap, optic.ext, c["sig_rough"], c["n_cap"], 2.23, [c["energy"]], [8.1], [0.5005])
    r = emul.launch_batch(prob, [c["start"]], [c["dir"]], [c["elecv"]])
    assert r["rc"][0] == c["rc"]


def test_single_reflection_reflectivity(emul, oracle, known):
    """reference tests/capil.c:302-334: weights after one reflection at 2 / 3.1 / 20 mrad"""
    optic, src, prob, (E, A, S) = make_pair(oracle, "ellip")
    for alfa, expect in ((2e-3, 0.984522), (3.1e-3, 0.496310)):
        d = (np.sin(alfa), 0.0, np.cos(alfa))
        r = emul.launch_batch(prob, [[0, 0, 0]], [d], [[0, 1, 0]])
        o = oracle.launch_one(optic, E, A, S, (0, 0, 0), d, (0, 1, 0))
        assert rel(r["weights"][0], o["weights"]).max() < 1e-9
    # a direct check of the Fresnel value: photon that reflects exactly once in a wide straight capillary
    from polycap_amd import Problem
    z = np.linspace(0, 0.6, 201)
    prob1 = Problem(z, np.full(201, 1e-3), np.full(201, 0.05), 0.0, 7, 2.23, E, A, S)
    opt1 = oracle.Optic(z, np.full(201, 1e-3), np.full(201, 0.05), 0.0, 7, 2.23)
    for alfa, expect, rc in ((2e-3, 0.984522, 1), (3.1e-3, 0.496310, 1), (2e-2, 0.000035, 0)):
        d = (np.sin(alfa), 0.0, np.cos(alfa))
        o = oracle.launch_one(opt1, E, A, S, (0, 0, 0.0), d, (0, 1, 0))
        r = emul.launch_batch(prob1, [[0, 0, 0.0]], [d], [[0, 1, 0]])
        assert o["rc"] == rc and o["i_refl"] == rc and abs(o["weights"][0] - expect) < 1e-5
        assert r["rc"][0] == rc and r["i_refl"][0] == rc and abs(r["weights"][0, 0] - expect) < 1e-5
        assert rel(r["weights"][0], o["weights"]).max() < 1e-9


def test_form3_reflectivity_of_many_energy_runs(emul, oracle):
    """Runs whose weights live in memory (more than 8 energies) evaluate the Fresnel factor in FORM 3 (pc_fresnel3: g = sqrt(n^2 -
    sin^2) formed directly, one reciprocal).  One reflection (a few at the steep angles) in a wide straight capillary at angles below, at and above the
    critical angles of a 12-energy grid that contains the reference's pinned 10 keV point: the reference's published weights
    (tests/capil.c:302-334) at that energy and the oracle's at all of them, with and without roughness."""
    from polycap_amd import Problem
    E = np.concatenate(([PIN_E], np.linspace(1.0, 30.0, 11)))
    A, S = synthetic_constants(E)
    A[0], S[0] = PIN_AMU, PIN_SCATF
    z = np.linspace(0, 0.6, 201)
    for sig in (0.0, 5.0):
        prob = Problem(z, np.full(201, 1e-3), np.full(201, 0.05), sig, 7, 2.23, E, A, S)
        opt = oracle.Optic(z, np.full(201, 1e-3), np.full(201, 0.05), sig, 7, 2.23)
        for alfa, expect in ((2e-3, 0.984522), (3.1e-3, 0.496310), (2e-2, 0.000035), (1.7e-3, None), (4.5e-3, None), (0.3, None)):
            d = (np.sin(alfa), 0.0, np.cos(alfa))
            for ev in ((0, 1, 0), (0.6, 0.8, 0)):
                o = oracle.launch_one(opt, E, A, S, (0, 0, 0.0), d, ev)
                r = emul.launch_batch(prob, [[0, 0, 0.0]], [d], [ev])
                assert r["rc"][0] == o["rc"] and r["i_refl"][0] == o["i_refl"]
                # both sides carry ~1e-16 of absolute rounding in cos^2 - 2 delta (the reference's 1 - sin^2/n^2 cancels): relative
                # 1e-10 ... 1e-9 of the reflectivity where it is steep
                assert rel(r["weights"][0], o["weights"]).max() < 2e-9, (sig, alfa, rel(r["weights"][0], o["weights"]).max())
                if expect is not None and sig == 0.0 and ev == (0, 1, 0) and o["i_refl"] == 1:
                    assert abs(r["weights"][0, 0] - expect) < 1e-5


def test_sampler_matches_oracle(emul, oracle):
    for which, source in (("xos1", (2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0)),
                          ("ellip", (0.05, 0.1, 0.1, 0.2, 0.2, 0., 0., 0.5)),
                          ("ellip", (2000., 0.2065, 0.1, 0., 0., 0.01, -0.02, 0.9)),
                          ("ellip", (5., 0.01, 0.01, -1., 0., 0., 0., 0.0))):
        optic, src, prob, _ = make_pair(oracle, which, source=source)
        slots = np.arange(3000)
        ref = oracle.sample_photons(optic, src, 424242, slots, attempt=3)
        got = emul.sample(prob, 424242, slots, np.full(slots.shape, 3))
        assert np.abs(got - ref).max() < 1e-13, (which, source)


def test_certified_march_bit_identical_to_literal(emul, oracle):
    for which, energies in (("xos1", (10.0,)), ("ellip", (10.0,)), ("ellip", (6.0, 10.0, 17.0))):
        optic, src, prob, _ = make_pair(oracle, which, energies=energies)
        ph = oracle.sample_photons(optic, src, 11, np.arange(40000))
        fast = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], literal=False)
        lit = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], literal=True)
        for k in ("rc", "weights", "exit_coords", "exit_dir", "exit_elecv", "i_refl", "d_travel"):
            assert np.array_equal(fast[k], lit[k], equal_nan=True), (which, k)
        # the certificate removes >85 % of the full segment evaluations
        assert fast["events"] < 0.15 * lit["events"]
        assert lit["fast_nodes"] == 0


def test_register_and_memory_weight_paths_agree(emul, oracle):
    optic, src, prob, _ = make_pair(oracle, "xos1")
    ph = oracle.sample_photons(optic, src, 3, np.arange(5000))
    a = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], use_regs=True)
    b = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], use_regs=False)
    for k in ("rc", "weights", "exit_coords", "i_refl"):
        assert np.array_equal(a[k], b[k], equal_nan=True)


def test_device_logic_vs_oracle_statistics(emul, oracle):
    """Same photons through oracle and device logic: entrance decisions identical, short trajectories tight,
    transmitted weight within the chaos floor (see tests/test_chaos_floor.py for the floor itself)."""
    for which in ("xos1", "ellip"):
        optic, src, prob, (E, A, S) = make_pair(oracle, which)
        n = 120000
        ph = oracle.sample_photons(optic, src, 20000, np.arange(n))
        o = oracle.launch_batch(optic, E, A, S, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        g = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        ent_o = np.isin(o["rc"], (2, -2))
        assert np.array_equal(ent_o, np.isin(g["rc"], (2, -2))) and np.array_equal(o["rc"][ent_o], g["rc"][ent_o])
        flips = ((o["rc"] != g["rc"]) | (o["i_refl"] != g["i_refl"])).mean()
        assert flips < 0.10, flips
        short = (o["rc"] == g["rc"]) & (o["i_refl"] == g["i_refl"]) & (o["i_refl"] <= 3) & np.isin(o["rc"], (0, 1))
        assert short.sum() > 5
        assert np.abs(g["exit_coords"][short] - o["exit_coords"][short]).max() < 1e-6
        assert rel(g["weights"][short], o["weights"][short]).max() < 1e-6
        so, sg = o["weights"][o["rc"] == 1, 0].sum(), g["weights"][g["rc"] == 1, 0].sum()
        assert abs(sg - so) / so < 1.0 / np.sqrt(n), (which, so, sg)


def test_mono_capillary_and_boundary_capillaries(emul, oracle):
    """n_shells == 0 (reference mono-capillary branches src/polycap-photon.c:514-537, src/polycap-capil.c:1282-1286) and a
    one-shell optic where every capillary is a boundary capillary: the per-node hexagon tests run in the march."""
    from tests.common import make_custom, MONO_CASE, SEVEN_CASE
    for case in (MONO_CASE, SEVEN_CASE):
        optic, src, prob, (E, A, S) = make_custom(oracle, **case)
        n = 20000
        ph = oracle.sample_photons(optic, src, 5, np.arange(n))
        assert np.abs(emul.sample(prob, 5, np.arange(n), np.zeros(n)) - ph).max() < 1e-13
        o = oracle.launch_batch(optic, E, A, S, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        g = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        lit = emul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], literal=True)
        for k in ("rc", "weights", "exit_coords", "i_refl", "d_travel"):
            assert np.array_equal(g[k], lit[k], equal_nan=True)
        assert len(np.unique(o["rc"])) >= 3
        # few reflections per photon here: the discrete outcomes agree photon by photon
        assert np.array_equal(o["rc"], g["rc"]) and np.array_equal(o["i_refl"], g["i_refl"])
        m = np.isin(o["rc"], (0, 1))
        assert rel(g["weights"][m], o["weights"][m]).max() < 1e-8
        assert np.abs(g["exit_coords"][m] - o["exit_coords"][m]).max() < 1e-9
