"""CPU checks of the device code of the leak ("halo") path (polycap_amd/csrc/hip/pc_leak.h) through its host compile
(tests/emul): the reference's known answers, certified skipping == literal stepping bit for bit, agreement with the
oracle, and the bookkeeping of the per-lane stack."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.test_oracle_leak_known_answers import constants


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


def problem(optic, energies, amu, scatf, source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.5)):
    from polycap_amd import Problem
    return Problem(optic.z, optic.cap, optic.ext, optic.sig_rough, optic.n_cap, optic.density, energies, amu, scatf, *source)


DIVERGENT = (0.05, 0.1, 0.1, 0.01, 0.01, 0., 0., 0.5)    # 10 mrad: reflections and wall crossings mix


def test_reference_known_answers(oracle, optic, leaks):
    """tests/leaks.c:865-1260 through the device code (host compile)"""
    from tests.emul import pyemul
    t = leaks["photon_leak"]
    for c in t["cases"]:
        if c.get("must_not_crash"):
            continue
        amu, scatf = constants(leaks, c["energy"])
        r = pyemul.launch_leak(problem(optic, [float(c["energy"])], [amu], [scatf]), [c["start"]], [c["dir"]], [c["elecv"]])
        ext, intl = pyemul.sort_leak_records(r["records"])
        assert r["rc"][0] == c["rc"], c
        if "n_ext" in c:
            assert (len(ext), len(intl)) == (c["n_ext"], c["n_int"]), c
        for got, exp in ((ext, c.get("ext", [])), (intl, c.get("int", []))):
            for g, e in zip(got, exp):
                assert np.abs(g[4:7] - np.array(e["coords"])).max() < 2e-5      # 41 chaotic reflections in the 6-event case
                assert np.abs(g[7:10] - np.array(e["dir"])).max() < 2e-5
                if "w" in e:
                    assert abs(g[14] - e["w"]) < (t["tol"] if c["energy"] != 10 else 5e-6)
        if "i_refl" in c:
            assert r["i_refl"][0] == c["i_refl"] and abs(r["d_travel"][0] - c["d_travel"]) < c["d_travel_tol"]


def test_certified_skipping_is_bit_identical_to_literal_stepping(oracle, optic, leaks):
    """The march certificates and the certified skipping of the leak path -- wall search by reach, blocks of the capillary probe,
    blocks of the outer-hexagon scan -- must not change one bit of any result or event record."""
    from tests.emul import pyemul
    # (scripts/analysis/leak_literal_check.py runs the same comparison on thousands of photons per source)
    for energies, src, n in (([10.0, 40.0], DIVERGENT, 300), ([10.0], (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5), 300),
                             ([40.0], (5., 0.15, 0.15, 0.04, 0.04, 0.02, 0.01, 0.5), 300)):
        cs = [constants(leaks, e) for e in energies]
        prob = problem(optic, energies, [a for a, _ in cs], [s for _, s in cs], source=src)
        ph = oracle.sample_photons(optic, oracle.make_source(*src), 4242, np.arange(n))
        fast = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        lit = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], literal=True)
        for k in fast:
            assert np.array_equal(fast[k], lit[k], equal_nan=True), k
        assert fast["records"].shape[0] > 100


def test_device_code_vs_oracle(oracle, optic, leaks):
    """Identical photons: same return codes, same events where the trajectory is short, events of every photon in the
    oracle's order (the reference's list order)."""
    from tests.emul import pyemul
    amu, scatf = constants(leaks, 40)
    prob = problem(optic, [40.0], [amu], [scatf], source=DIVERGENT)
    n = 150
    ph = oracle.sample_photons(optic, oracle.make_source(*DIVERGENT), 99, np.arange(n))
    g = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
    gext, gint = pyemul.sort_leak_records(g["records"])
    same = 0
    for j in range(n):
        o = oracle.launch_one_leak(optic, [40.0], [amu], [scatf], ph[j, 0:3], ph[j, 3:6], ph[j, 6:9])
        assert o["rc"] == g["rc"][j] or o["i_refl"] > 3, j
        ge, gi = gext[gext[:, 0] == j], gint[gint[:, 0] == j]
        if len(ge) == len(o["ext"]) and len(gi) == len(o["int"]):
            same += 1
            nrefl = np.concatenate([o["ext"][:, 9], o["int"][:, 9], [0.]])
            if o["i_refl"] <= 3 and nrefl.max() <= 4:
                if len(ge):
                    assert np.abs(ge[:, 4:] - o["ext"]).max() < 1e-6
                if len(gi):
                    assert np.abs(gi[:, 4:] - o["int"]).max() < 1e-6
    assert same / n > 0.9 and len(gext) > 20


def test_driver_and_stack_limits(oracle, optic, leaks):
    from tests.emul import pyemul
    t = leaks["source_leak"]
    amu, scatf = constants(leaks, 10)
    prob = problem(optic, [10.0], [amu], [scatf], source=tuple(t["source"]))
    n = 150
    g = pyemul.transmission_leak(prob, 20000, 0, n)
    o = oracle.transmission(optic, oracle.make_source(*t["source"]), [10.0], [amu], [scatf], 20000, 0, n, leak_calc=True)
    assert g["counters"][0] == n and not g["stack_overflow"]
    eg = g["sum_weights"][0] / (g["counters"][0] + g["counters"][1] + g["counters"][2])
    assert abs(eg - o["efficiencies"][0]) <= 4. / np.sqrt(n) * o["efficiencies"][0]
    ext, intl = pyemul.sort_leak_records(g["records"])
    assert abs(len(ext) - len(o["ext"])) <= 0.25 * len(o["ext"]) + 10 and abs(len(intl) - len(o["int"])) <= 0.25 * len(o["int"]) + 10
    # a steep 40 keV photon crosses hundreds of walls: one stack frame per wall; too few frames are reported, not ignored
    amu, scatf = constants(leaks, 40)
    prob = problem(optic, [40.0], [amu], [scatf])
    steep = dict(start=[[0.0005, 0., 0.]], direction=[[0.2, 0.05, 1.]], elecv=[[1., 0., 0.]])
    deep = pyemul.launch_leak(prob, steep["start"], steep["direction"], steep["elecv"], max_depth=1024)
    assert not deep["stack_overflow"] and deep["records"].shape[0] >= 1
    shallow = pyemul.launch_leak(prob, steep["start"], steep["direction"], steep["elecv"], max_depth=8)
    assert shallow["stack_overflow"]
