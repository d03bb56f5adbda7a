"""Development check (the tests hold a small version of it): certified skipping of the leak path == literal stepping, bit for
bit, on the host compile of the device headers, for many photons of several sources:
    python scripts/analysis/leak_literal_check.py [photons per source, default 600] [--shapes]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle
from tests.conftest import GOLDEN
from tests.emul import pyemul
from tests.test_device_leak_cpu import DIVERGENT, problem
from tests.test_oracle_leak_known_answers import constants

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 600
pyoracle.build()
known = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
leaks = json.load(open(os.path.join(GOLDEN, "reference_leak_known_answers.json")))
t = known["test_optic"]
optic = pyoracle.Optic.from_shape(t["type"], t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"], t["rad_int_upstream"],
                                  t["rad_int_downstream"], t["focal_dist_upstream"], t["focal_dist_downstream"], t["sig_rough"],
                                  t["n_cap"], known["glass"]["density"])
sources = (("divergent", DIVERGENT, [10.0, 40.0]), ("uniform", (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5), [10.0]),
           ("close, steep", (5., 0.15, 0.15, 0.04, 0.04, 0.02, 0.01, 0.5), [40.0]))
# other shapes of optic: a conical and a paraboloidal profile, a 7-capillary and a 91-capillary stack (rays leave the stack
# after a few cells: the branches outside the hexagon stacking and the outer-hexagon scan), a short profile (99 segments)
def shape(ptype, n_cap, nmax=999, rint=(t["rad_int_upstream"], t["rad_int_downstream"])):
    if nmax != 999:
        return pyoracle.Optic.from_shape(ptype, t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"], rint[0], rint[1],
                                         t["focal_dist_upstream"], t["focal_dist_downstream"], t["sig_rough"], n_cap, known["glass"]["density"], nmax=nmax)
    from polycap_amd import capi
    pr = capi.Profile(ptype, t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"], rint[0], rint[1], t["focal_dist_upstream"], t["focal_dist_downstream"])
    return pyoracle.Optic(pr.get_z(), pr.get_cap(), pr.get_ext(), t["sig_rough"], n_cap, known["glass"]["density"])
optics = [("ellipsoidal test optic", optic)]
if "--shapes" in sys.argv:
    optics += [("conical", shape(0, t["n_cap"])), ("paraboloidal", shape(1, t["n_cap"])),
               ("7 capillaries", shape(2, 7, rint=(0.05, 0.015))), ("91 capillaries", shape(2, 91, rint=(0.015, 0.004))),
               ("99 segments", shape(2, t["n_cap"], nmax=99))]
for oname, optic in optics:
  for name, src, energies in sources:
    name = oname + ", " + name
    cs = [constants(leaks, e) for e in energies]
    prob = problem(optic, energies, [a for a, _ in cs], [s for _, s in cs], source=src)
    ph = pyoracle.sample_photons(optic, pyoracle.make_source(*src), 777, np.arange(n))
    t0 = time.time()
    fast = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
    t1 = time.time()
    lit = pyemul.launch_leak(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9], literal=True)
    t2 = time.time()
    for k in fast:
        assert np.array_equal(fast[k], lit[k], equal_nan=True), (name, k)
    print("%-40s %d photons, %d event records: identical (certified %.1f s, literal %.1f s)" % (name, n, fast["records"].shape[0], t1 - t0, t2 - t1))
