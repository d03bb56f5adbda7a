"""GPU parity of the leak ("halo") path, leak_calc=true: pc_hip_launch_photons_leak / pc_hip_transmission_run_leak
against the CPU oracle (pinned to the reference's tests/leaks.c) and against the host compile of the same device
headers (bit-identical, including the order of the leak events)."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.test_oracle_leak_known_answers import constants

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import polycap_amd
    assert polycap_amd.device_count() >= 1, "no HIP device visible"
    return polycap_amd


@pytest.fixture(scope="module")
def leaks():
    with open(os.path.join(GOLDEN, "reference_leak_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def optic(oracle, known):
    t = known["test_optic"]
    return oracle.Optic.from_shape(t["type"], t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"],
                                   t["rad_int_upstream"], t["rad_int_downstream"], t["focal_dist_upstream"],
                                   t["focal_dist_downstream"], t["sig_rough"], t["n_cap"], known["glass"]["density"])


def problem(pa, optic, energies, amu, scatf, source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.5)):
    return pa.Problem(optic.z, optic.cap, optic.ext, optic.sig_rough, optic.n_cap, optic.density, energies, amu, scatf, *source)


def test_reference_known_answers_through_the_kernel(pa, oracle, optic, leaks):
    """tests/leaks.c:865-1260 (polycap_photon_launch with leak_calc=true): return codes, event counts, coordinates,
    directions and weights of the reference, reproduced by the HIP kernel."""
    t = leaks["photon_leak"]
    for c in t["cases"]:
        if c.get("must_not_crash"):
            E = [float(e) for e in c["energies"]]
            cs = [constants(leaks, e) for e in E]
            with pa.TraceContext(problem(pa, optic, E, [a for a, _ in cs], [s for _, s in cs])) as ctx:
                r = ctx.launch_photons([c["start"]], [c["dir"]], [c["elecv"]], leak_calc=True)
            assert r["rc"][0] in (1, 0, 2, -1, -2)
            continue
        amu, scatf = constants(leaks, c["energy"])
        with pa.TraceContext(problem(pa, optic, [float(c["energy"])], [amu], [scatf])) as ctx:
            r = ctx.launch_photons([c["start"]], [c["dir"]], [c["elecv"]], leak_calc=True)
            ext, intl = ctx.leaks()
        assert r["rc"][0] == c["rc"], c
        if "n_ext" in c:
            assert (len(ext), len(intl)) == (c["n_ext"], c["n_int"]), c
        for got, exp in ((ext, c.get("ext", [])), (intl, c.get("int", []))):
            for g, e in zip(got, exp):
                # long chaotic trajectories (41 reflections in the 6-event case) amplify rounding differences: 2e-5
                assert np.abs(g[2:5] - np.array(e["coords"])).max() < 2e-5
                assert np.abs(g[5:8] - np.array(e["dir"])).max() < 2e-5
                if "w" in e:
                    assert abs(g[12] - e["w"]) < (t["tol"] if c["energy"] != 10 else 5e-6)
        if "weight" in c:
            assert abs(r["weights"][0, 0] - c["weight"]) < t["tol"]
        if "i_refl" in c:
            assert r["i_refl"][0] == c["i_refl"] and abs(r["d_travel"][0] - c["d_travel"]) < c["d_travel_tol"]


def test_gpu_bit_identical_to_host_compile_with_leaks(pa, oracle, optic, leaks):
    """Same device headers compiled for the host (tests/emul): identical return codes, weights and leak events, event
    for event; geometry bit for bit, leak weights to the last bits (they contain exp(-d*amu), and the device's exp is
    not glibc's), for sampled photons of a divergent source at three energies."""
    from tests.emul import pyemul
    E = [10.0, 20.0, 40.0]
    cs = [constants(leaks, e) for e in E]
    src = (0.05, 0.1, 0.1, 0.01, 0.01, 0., 0., 0.5)     # 10 mrad divergence: reflections and wall crossings mix
    ph = oracle.sample_photons(optic, oracle.make_source(*src), 31337, np.arange(500))
    for sig_rough in (0.0, 5.0):                         # smooth walls, and 5 Angstrom roughness (exp() in every weight)
        prob = pa.Problem(optic.z, optic.cap, optic.ext, sig_rough, optic.n_cap, optic.density, E, [a for a, _ in cs],
                          [s for _, s in cs], *src)
        with pa.TraceContext(prob) as ctx:
            g = ctx.launch_photons(ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], leak_calc=True)
            gext, gint = ctx.leaks()
        e = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        eext, eint = pyemul.sort_leak_records(e["records"])
        for k in ("rc", "exit_coords", "exit_dir", "i_refl", "d_travel"):
            assert np.array_equal(g[k], e[k], equal_nan=True), k
        if sig_rough == 0.:
            assert np.array_equal(g["weights"], e["weights"], equal_nan=True)
        else:
            assert np.allclose(g["weights"], e["weights"], rtol=1e-12, atol=1e-300, equal_nan=True)
        assert len(gext) + len(gint) > 100
        # emulation records: slot, attempt, seq, kind, payload; kernel events: slot, attempt, payload
        for got, exp in ((gext, eext), (gint, eint)):
            assert got.shape[0] == exp.shape[0] and np.array_equal(got[:, 0], exp[:, 0])
            assert np.array_equal(got[:, 2:12], exp[:, 4:14])                              # coords, direction, elecv, n_refl
            assert np.allclose(got[:, 12:], exp[:, 14:], rtol=1e-12, atol=0.)


def test_explicit_photons_with_leaks_vs_oracle(pa, oracle, optic, leaks):
    """Identical photons through the oracle and the kernel at 40 keV (many wall crossings): the discrete outcome of the
    launched photon is unchanged by leak_calc except for the reference's own rejections (launch -1), and the leak
    events agree one by one wherever the chaotic trajectory has not yet amplified rounding differences."""
    amu, scatf = constants(leaks, 40)
    src = (0.05, 0.1, 0.1, 0.01, 0.01, 0., 0., 0.5)     # 10 mrad divergence: reflections and wall crossings mix
    prob = problem(pa, optic, [40.0], [amu], [scatf], source=src)
    n = 400
    ph = oracle.sample_photons(optic, oracle.make_source(*src), 99, np.arange(n))
    with pa.TraceContext(prob) as ctx:
        g = ctx.launch_photons(ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], leak_calc=True)
        gext, gint = ctx.leaks()
    same_events = total = 0
    for j in range(n):
        o = oracle.launch_one_leak(optic, [40.0], [amu], [scatf], ph[j, 0:3], ph[j, 3:6], ph[j, 6:9])
        assert o["rc"] == g["rc"][j] or o["i_refl"] > 3, j
        ge, gi = gext[gext[:, 0] == j], gint[gint[:, 0] == j]
        total += 1
        if len(ge) == len(o["ext"]) and len(gi) == len(o["int"]):
            same_events += 1
            # "short": neither the launched photon nor any leaked fraction has reflected more than a few times
            nrefl = np.concatenate([o["ext"][:, 9], o["int"][:, 9], [0.]])
            short = o["i_refl"] <= 3 and nrefl.max() <= 4
            if short and len(ge):
                assert np.abs(ge[:, 2:] - o["ext"]).max() < 1e-6
            if short and len(gi):
                assert np.abs(gi[:, 2:] - o["int"]).max() < 1e-6
    assert same_events / total > 0.9
    assert len(gext) > 50 and len(gint) > 5


def test_driver_with_leaks_vs_oracle(pa, oracle, optic, leaks):
    """polycap_source_get_transmission_efficiencies(leak_calc=true): every slot delivers an exit photon; efficiency and
    the per-slot leak statistics agree with the oracle on the same seeds within the reference's own chaotic noise;
    the totals are those of a run without leak_calc up to the photons the leak path rejects; results do not depend on
    how the slots are split into runs."""
    t = leaks["source_leak"]
    amu, scatf = constants(leaks, 10)
    prob = problem(pa, optic, [10.0], [amu], [scatf], source=tuple(t["source"]))
    n = 300
    with pa.TraceContext(prob) as ctx:
        g = ctx.transmission(20000, 0, n, keep_images=True, leak_calc=True)
        g0 = ctx.transmission(20000, 0, n, keep_images=True, leak_calc=False)
        parts = [ctx.transmission(20000, s0, c, leak_calc=True) for s0, c in ((0, 101), (101, n - 101))]
        ctx.set_option("leak_capacity", 64)           # far too small: the run repeats itself with what it needs
        small = ctx.transmission(20000, 0, n, leak_calc=True)
    o = oracle.transmission(optic, oracle.make_source(*t["source"]), [10.0], [amu], [scatf], 20000, 0, n, leak_calc=True)
    assert g["i_exit"] == n and g["failed_slots"] == 0
    assert abs(g["efficiencies"][0] - t["efficiencies"][2]) <= t["tol"]
    assert abs(g["efficiencies"][0] - o["efficiencies"][0]) <= 4. / np.sqrt(n) * o["efficiencies"][0] + 1e-12
    assert abs(g["efficiencies"][0] - g0["efficiencies"][0]) <= 0.05
    for kind in ("ext", "int"):
        assert len(g[kind]) > 0
        assert abs(len(g[kind]) - len(o[kind])) <= 0.15 * len(o[kind]) + 10
        assert np.all(np.diff(g[kind][:, 0]) >= 0)                                  # slot-major order
        assert abs(g[kind][:, 12].sum() - o[kind][:, 12].sum()) <= 0.2 * o[kind][:, 12].sum() + 0.05
        # partition invariance and buffer growth: identical events, bit for bit
        both = np.concatenate([p[kind] for p in parts])
        assert np.array_equal(both, g[kind]) and np.array_equal(small[kind], g[kind])
    assert np.array_equal(g["counters"][:4], parts[0]["counters"][:4] + parts[1]["counters"][:4])
    # the same through the multi-GPU entry point: two "ranks" on this device, events concatenated in rank order
    from polycap_amd import distributed as pcd
    ranks = [pcd.run_sharded(prob, 20000, n, rank=r, world_size=2, leak_calc=True) for r in (0, 1)]
    for kind in ("ext", "int"):
        assert np.array_equal(np.concatenate([r["local"][kind] for r in ranks]), g[kind])


def test_driver_with_leaks_is_event_for_event_the_host_compile(pa, oracle, optic, leaks):
    """The source driver with leak_calc=true on the workload of the leak bench (uniform illumination, 10 keV), 3000 slots = 7700
    launches: counters, the exit weight of every slot and every leak event (attempt by attempt, in list order inside an attempt)
    equal the host compile of the same headers running the slots one
    after the other -- geometry bit for bit, leak weights to 1e-12 (the device's exp is not glibc's).  Covers the certified
    wall search, probe and outer-hexagon scan of the kernel against the sequential loop at a size where every branch occurs."""
    from tests.emul import pyemul
    amu, scatf = constants(leaks, 10)
    src = (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5)
    prob = problem(pa, optic, [10.0], [amu], [scatf], source=src)
    n = 3000
    with pa.TraceContext(prob) as ctx:
        g = ctx.transmission(20000, 0, n, keep_images=True, leak_calc=True)
        gw = ctx.images()["exit_weights"]
    e = pyemul.transmission_leak(prob, 20000, 0, n)
    assert not e["stack_overflow"]
    assert np.array_equal(np.asarray(g["counters"][:4], dtype=np.int64), e["counters"])
    assert np.array_equal(np.asarray(gw).reshape(n, -1), e["exit_weights"])
    rec = e["records"]
    void = {(r[0], r[1]) for r in rec if r[3] < 0}
    rec = rec[np.array([(r[0], r[1]) not in void and r[3] >= 0 for r in rec], dtype=bool)]
    rec = rec[np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))]                          # slot, attempt, seq
    n_events = 0
    for kind, got in ((0, g["ext"]), (1, g["int"])):
        exp = rec[rec[:, 3] == kind]
        assert got.shape[0] == exp.shape[0]
        assert np.all(np.diff(got[:, 0]) >= 0)                                          # slot-major
        # the kernel's lists hold a slot's transmitted attempt first (which attempt that was is not part of the emulation's
        # output when it left no event): compared attempt by attempt, inside an attempt in list order
        got = got[np.lexsort((got[:, 1], got[:, 0]))]                                   # stable: keeps the order inside an attempt
        assert np.array_equal(got[:, 0:2], exp[:, 0:2])                                 # slot, attempt
        assert np.array_equal(got[:, 2:12], exp[:, 4:14])                               # coords, direction, elecv, n_refl
        assert np.allclose(got[:, 12:], exp[:, 14:], rtol=1e-12, atol=0.)
        n_events += got.shape[0]
    assert n_events > 20000


def test_order_of_the_slots_does_not_change_a_leak_run(pa, oracle, optic, leaks):
    """Leak runs hand out their slots heaviest first (predicted by a plain pre-pass of the same slots) and give the heaviest ones
    lanes of their own (pc_leak_kargs::order): totals, exit weights and every event of (a) a caller's arbitrary order with a heavy
    tier on a small run and (b) the automatic order on a run large enough for it equal those of the same run in slot order."""
    amu, scatf = constants(leaks, 10)
    src = (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5)
    prob = problem(pa, optic, [10.0], [amu], [scatf], source=src)

    def same(x, y):
        assert np.array_equal(x["counters"], y["counters"]) and np.array_equal(x["sum_weights"], y["sum_weights"])
        assert np.array_equal(x["ext"], y["ext"]) and np.array_equal(x["int"], y["int"])

    with pa.TraceContext(prob) as ctx:
        n = 6000
        ctx.set_option("leak_order", 0)
        ref = ctx.transmission(20000, 0, n, keep_images=True, leak_calc=True)
        wref = ctx.images()["exit_weights"].copy()
        rng = np.random.default_rng(5)
        for n_heavy, lanes_, every in ((0, 1, 1), (40, 1, 1), (300, 2, 3)):
            ctx.set_option("leak_heavy_lanes", lanes_)
            ctx.set_option("leak_heavy_every", every)
            ctx.leak_set_order(rng.permutation(n), n_heavy)
            got = ctx.transmission(20000, 0, n, keep_images=True, leak_calc=True)
            same(got, ref)
            assert np.array_equal(ctx.images()["exit_weights"], wref)
        with pytest.raises(pa.HipError):
            ctx.leak_set_order(np.zeros(n, dtype=np.uint32), 0)           # not a permutation
        ctx.leak_set_order(np.zeros(0, dtype=np.uint32), 0)               # back to the context's own choice
        ctx.set_option("leak_heavy_lanes", 1)
        ctx.set_option("leak_heavy_every", 1)
        n = 200_000
        plain = ctx.transmission(7, 0, n, leak_calc=True)
        ctx.set_option("leak_order", 1)
        ctx.set_option("leak_slot_units", 1)
        ordered = ctx.transmission(7, 0, n, leak_calc=True)
        same(ordered, plain)
        units = ctx.leak_slot_units(0, n)
        assert units.min() > 0 and units.max() > 20 * units.mean() / 4      # every slot traced; a heavy tail exists


# noise constants c = std(device - oracle, relative) x sqrt(N_started) of identical-seed leak runs, measured over the 16 seeds of
# tests/golden/oracle_leak_seeds.json (test_leak_driver_against_the_oracle_seed_by_seed prints them; profiles/r04/leak_parity_seeds.txt)
LEAK_NOISE = {"started": 0.47, "eff": 0.56, "n_ext": 1.76, "n_int": 0.52, "ext_w": 3.38, "int_w": 0.95}


def test_leak_driver_totals_against_the_oracle_fixture(pa, oracle, optic, leaks):
    """tests/golden/oracle_leak_totals.json (scripts/make_oracle_leak_totals.py): the CPU oracle's leak driver -- the reference's
    literal algorithm -- on 8000 exit-photon slots of the leak bench's workload, one seed.  Identical photon streams; the
    trajectories are chaotic, so the kernel agrees statistically, not event by event: every quantity within
    4 c / sqrt(N_started) of the oracle's, c = the noise constant of that quantity measured over 16 seeds (LEAK_NOISE)."""
    import json
    import os
    from tests.conftest import GOLDEN
    with open(os.path.join(GOLDEN, "oracle_leak_totals.json")) as f:
        fx = json.load(f)
    amu, scatf = constants(leaks, 10)
    prob = problem(pa, optic, [fx["energy_keV"]], [amu], [scatf], source=tuple(fx["source"]))
    n = sum(b["n"] for b in fx["blocks"])
    with pa.TraceContext(prob) as ctx:
        g = ctx.transmission(fx["seed"], 0, n, leak_calc=True)
    o_cnt = np.sum([b["counters"] for b in fx["blocks"]], axis=0)
    o_start = o_cnt[0] + o_cnt[1] + o_cnt[2]
    g_start = g["i_start"]
    o_eff = sum(b["sum_weight"] for b in fx["blocks"]) / o_start
    sigma = 1.0 / np.sqrt(o_start)
    assert g["i_exit"] == n == o_cnt[0]
    assert abs(g_start / o_start - 1.0) <= 4. * LEAK_NOISE["started"] * sigma
    assert abs(g["efficiencies"][0] / o_eff - 1.0) <= 4. * LEAK_NOISE["eff"] * sigma
    for kind, key in (("ext", "n_ext"), ("int", "n_int")):
        o_n = sum(b[key] for b in fx["blocks"])
        o_w = sum(b[kind + "_weight"] for b in fx["blocks"])
        assert abs(len(g[kind]) / o_n - 1.0) <= 4. * LEAK_NOISE[key] * sigma, kind
        assert abs(g[kind][:, 12].sum() / o_w - 1.0) <= 4. * LEAK_NOISE[kind + "_w"] * sigma, kind


def test_leak_driver_against_the_oracle_seed_by_seed(pa, optic):
    """tests/golden/oracle_leak_seeds.json (scripts/make_oracle_leak_seeds.py): the CPU oracle's leak driver -- the reference's
    literal algorithm -- for 16 seeds x 8000 exit-photon slots at 10 keV (the leak bench's workload) and 8 seeds x 4000 slots on
    the seven energies of the reference's source test, with the optical constants each group used.  Identical photon streams;
    the trajectories are chaotic, so the kernel agrees statistically: per quantity -- started photons,
    efficiency, numbers and summed weights of both kinds of leak event, per energy -- the per-seed deltas (device - oracle, in
    units of the oracle's mean over the seeds) must be consistent with zero: |mean| below the two-sided 0.1 % point of Student's
    t for the group's number of seeds (4.07 standard errors for 16 seeds, 5.41 for 8; the headline's fixture has 128 seeds and
    uses 3) -- and, whatever the noise, below 1 % + that.  The noise constants c = std(delta) sqrt(N_started per seed) are
    printed.  A bias of the certified wall search would show here."""
    import json
    import os
    from scipy import stats
    from tests.conftest import GOLDEN
    path = os.environ.get("POLYCAP_LEAK_SEEDS_FIXTURE", os.path.join(GOLDEN, "oracle_leak_seeds.json"))
    if not os.path.exists(path):
        pytest.skip("tests/golden/oracle_leak_seeds.json is missing: python scripts/make_oracle_leak_seeds.py (several CPU-hours)")
    with open(path) as f:
        fx = json.load(f)
    report = []
    for group in fx["groups"]:
        E = group["energies"]
        ne = len(E)
        prob = problem(pa, optic, E, group["amu"], group["scatf"], source=tuple(fx["source"]))
        dev, ora = {}, {}
        n_started = []
        with pa.TraceContext(prob) as ctx:
            for run in group["runs"]:
                g = ctx.transmission(run["seed"], 0, run["n"], leak_calc=True, leak_views=True)
                o_start = sum(run["counters"][:3])
                assert g["i_exit"] == run["n"] == run["counters"][0]
                n_started.append(o_start)
                q = {"started": (g["i_start"], o_start), "n_ext": (len(g["ext"]), run["n_ext"]), "n_int": (len(g["int"]), run["n_int"])}
                for e in range(ne):
                    q["eff_%g" % E[e]] = (g["sum_weights"][e] / g["i_start"], run["sum_weights"][e] / o_start)
                    q["ext_w_%g" % E[e]] = (g["ext"][:, 12 + e].sum(), run["ext_weights"][e])
                    q["int_w_%g" % E[e]] = (g["int"][:, 12 + e].sum(), run["int_weights"][e])
                for k, (a, b) in q.items():
                    dev.setdefault(k, []).append(float(a))
                    ora.setdefault(k, []).append(float(b))
        K = len(group["runs"])
        t_crit = float(stats.t.ppf(1.0 - 0.0005, K - 1))
        n_mean = float(np.mean(n_started))
        # a kind of event that hardly carries weight at an energy (extleak at 1 keV: the glass absorbs everything) has no
        # meaningful relative delta: such quantities are compared in units of the largest summed weight of their kind instead
        scale = {"ext_w": max(np.mean(ora["ext_w_%g" % x]) for x in E), "int_w": max(np.mean(ora["int_w_%g" % x]) for x in E)}
        for k in dev:
            a, b = np.array(dev[k]), np.array(ora[k])
            unit = float(b.mean())
            kind = k[:5] if k[:5] in scale else None
            if kind is not None and unit < 1e-3 * scale[kind]:
                unit = scale[kind]
            if unit <= 0.:
                assert np.all(a == 0.), k
                continue
            d = (a - b) / unit
            se = d.std(ddof=1) / np.sqrt(K)
            z = d.mean() / se if se > 0 else 0.0
            report.append((group["name"], k, d.mean(), se, z, d.std(ddof=1) * np.sqrt(n_mean), t_crit))
    for name, k, m, se, z, c, tc in report:
        print("%-15s %-12s mean %+.2e +- %.2e  t %+5.2f (limit %.2f)  c %.2f" % (name, k, m, se, z, tc, c))
    for r in report:
        assert abs(r[4]) < r[6], r
        assert abs(r[2]) < 0.01 + r[6] * r[3], r


@pytest.mark.parametrize("binding", ["ctypes", "cython"])
def test_public_api_with_leaks(pa, leaks, known, binding, monkeypatch):
    """The reference's Python test of the leak path (tests/python.py:147-201: one 40 keV photon, two extleak and three
    intleak events with published coordinates and weights, delta 1e-6) and its source-level test (tests/leaks.c:1264-1340,
    at the reference's own size and tolerance) through the reference-shaped API of both bindings, down to the leak groups of the HDF5 file."""
    import os
    import tempfile
    if binding == "ctypes":
        from polycap_amd import capi
    else:
        from polycap_amd.pyext import polycap as capi
    prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
    desc = capi.Description(prof, 0.0, 200000, {"O": 53.0, "Si": 47.0}, 2.23)
    c = [x for x in leaks["photon_leak"]["cases"] if x.get("energy") == 40][0]
    photon = capi.Photon(desc, c["start"], c["dir"], c["elecv"])
    weights = photon.launch(40.0, leak_calc=True)
    assert isinstance(weights, np.ndarray)
    for _ in range(2):                                   # twice: the lists are cached
        ext, intl = list(photon.extleak_data), list(photon.intleak_data)
        assert (len(ext), len(intl)) == (2, 3)
    for got, exp in ((ext, c["ext"]), (intl, c["int"])):
        for g, e in zip(got, exp):
            assert isinstance(g.coords, capi.VectorTuple)
            assert np.abs(np.array(g.coords) - e["coords"]).max() < 1e-6
            assert np.abs(np.array(g.direction) - e["dir"]).max() < 1e-6
            assert abs(g.weight[0] - e["w"]) < 1e-6
    assert photon.i_refl == 4 and abs(photon.d_travel - 2.744994) < 1e-6
    # a second launch without leak_calc drops the events (src/polycap-photon.c:434-450)
    photon.launch(40.0)
    with pytest.raises(ValueError):
        list(photon.extleak_data)

    t = leaks["source_leak"]
    monkeypatch.setenv("POLYCAP_SEED", "20000")          # both runs trace the same photon streams
    src = capi.Source(desc, *t["source"], np.array(t["energies"], dtype=np.float64))
    n = t["n_photons"]                                   # the reference's own size and tolerance: 2500 photons, 0.05 (tests/leaks.c:1291-1306)
    assert n == 2500 and t["tol"] == 0.05
    eff = src.get_transmission_efficiencies(-1, n, leak_calc=True)
    eff0 = src.get_transmission_efficiencies(-1, n, leak_calc=False)
    E, T = eff.data
    assert np.all(np.abs(T - np.array(t["efficiencies"])) <= t["tol"])
    assert np.all(np.abs(T - eff0.data[1]) <= t["tol"])
    ext, intl = list(eff.extleak_data), list(eff.intleak_data)
    assert len(ext) > 0 and len(intl) > 0 and len(list(eff.exit_coords)) == n
    for l in ext + intl:
        assert l.weight.shape == (7,) and l.weight.max() >= 1e-4 and np.all((l.weight >= 0) & (l.weight <= 1))
        assert 0. <= l.coords.z <= 9.001 and abs(np.linalg.norm(l.direction) - 1.) < 1e-9
    assert any(abs(l.coords.z - 9.) < 1e-3 for l in intl)
    with pytest.raises(ValueError):
        list(eff0.extleak_data)
    from tests import test_hdf5_writer as H
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "leaks.h5")
        eff.write_hdf5(path)
        if H.H5LS is not None:
            shapes = H._listing(path)
            assert shapes["/ExternalLeaks/Coordinates"] == (3, len(ext)) and shapes["/InternalLeaks/Electric_Vector"] == (2, len(intl))
            assert shapes["/ExternalLeaks/Weights"] == (len(ext), 7) and shapes["/InternalLeaks/Weight_Total"] == (7,)
            w = H._read(path, "/InternalLeaks/Weights", tmp).reshape(len(intl), 7)
            assert np.array_equal(w, np.array([l.weight for l in intl]))
            tot = H._read(path, "/ExternalLeaks/Weight_Total", tmp)        # sum of the event weights / started photons
            ratio = np.array([l.weight for l in ext]).sum(axis=0) / tot
            assert np.allclose(ratio, ratio[0], rtol=1e-12) and ratio[0] >= n
            assert H._units(path)["/ExternalLeaks/N_Reflections"] == "a.u."


def test_leak_runs_are_sharded_over_the_device_list(leaks, monkeypatch):
    """POLYCAP_HIP_DEVICES with leak_calc=true: the members of the device group trace contiguous slot ranges, every member
    orders its events on its own device and the lists are appended in member order (reference: the whole OpenMP team traces leak
    runs, src/polycap-source.c:744-884, and appends its lists, :925-1032).  Through the public call, devices 0,0 and 0,0,0 against
    one device: efficiencies, every image plane and every leak event equal, down to the leak groups of the HDF5 file."""
    import os
    import tempfile
    from polycap_amd import capi
    from tests import test_hdf5_writer as H
    prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
    desc = capi.Description(prof, 0.0, 200000, {"O": 53.0, "Si": 47.0}, 2.23)
    t = leaks["source_leak"]
    monkeypatch.setenv("POLYCAP_SEED", "31")
    monkeypatch.setenv("POLYCAP_RCCL", "0")              # one GPU listed several times: the host sum (RCCL wants distinct devices)
    n = 3001

    def run(devices):
        if devices is None:
            monkeypatch.delenv("POLYCAP_HIP_DEVICES", raising=False)
        else:
            monkeypatch.setenv("POLYCAP_HIP_DEVICES", devices)
        src = capi.Source(desc, *t["source"], np.array([10.0, 17.0]))
        eff = src.get_transmission_efficiencies(-1, n, leak_calc=True)
        ev = [np.array([list(l.coords) + list(l.direction) + list(l.elecv) + [l.n_refl] + list(l.weight) for l in lst])
              for lst in (eff.extleak_data, eff.intleak_data)]
        planes = [np.array(list(g)) for g in (eff.start_coords, eff.exit_coords, eff.exit_direction, eff.exit_weights)]
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "l.h5")
            eff.write_hdf5(path)
            h5 = None
            if H.H5LS is not None:
                h5 = [H._read(path, d, tmp) for d in ("/ExternalLeaks/Coordinates", "/ExternalLeaks/Weights", "/InternalLeaks/Coordinates",
                                                       "/InternalLeaks/Weights", "/InternalLeaks/Weight_Total", "/PC_Exit/Weights")]
        return eff.data[1].copy(), ev, planes, h5

    one = run(None)
    assert len(one[1][0]) > 100 and len(one[1][1]) > 1000
    for devices in ("0,0", "0,0,0"):
        got = run(devices)
        assert np.array_equal(one[0], got[0]), devices
        for a, b in zip(one[1], got[1]):                      # event for event, both lists (the reference's NaN tolerance kept)
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), devices
        for a, b in zip(one[2], got[2]):
            assert np.array_equal(a, b, equal_nan=True), devices
        if one[3] is not None:
            for a, b in zip(one[3], got[3]):
                assert np.array_equal(a, b), devices


def test_long_profile_with_leaks(pa, oracle, leaks, known):
    """1600-point profile: the leak kernel built for the wide LDS pitch (all seven tables, 144 KB) against the host compile"""
    from tests.emul import pyemul
    t = known["test_optic"]
    long_optic = oracle.Optic.from_shape(t["type"], t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"],
                                         t["rad_int_upstream"], t["rad_int_downstream"], t["focal_dist_upstream"],
                                         t["focal_dist_downstream"], t["sig_rough"], t["n_cap"], known["glass"]["density"], nmax=1599)
    amu, scatf = constants(leaks, 40)
    src = (0.05, 0.1, 0.1, 0.01, 0.01, 0., 0., 0.5)
    prob = problem(pa, long_optic, [40.0], [amu], [scatf], source=src)
    ph = oracle.sample_photons(long_optic, oracle.make_source(*src), 5, np.arange(300))
    with pa.TraceContext(prob) as ctx:
        g = ctx.launch_photons(ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], leak_calc=True)
        gext, gint = ctx.leaks()
    e = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], max_depth=2048)
    eext, eint = pyemul.sort_leak_records(e["records"])
    for k in ("rc", "weights", "exit_coords", "exit_dir", "i_refl", "d_travel"):
        assert np.array_equal(g[k], e[k], equal_nan=True), k
    assert len(gext) > 50
    for got, exp in ((gext, eext), (gint, eint)):
        assert got.shape[0] == exp.shape[0] and np.array_equal(got[:, 2:12], exp[:, 4:14])
        assert np.allclose(got[:, 12:], exp[:, 14:], rtol=1e-12, atol=0.)
