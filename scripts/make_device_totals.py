"""Regression pin of the device arithmetic: counters and exact weight sums of the host compile of the device code
(tests/emul, bit-identical to the kernels) for a few (seed, slot range) runs of xos1 @ 10 keV.
    python scripts/make_device_totals.py  ->  tests/golden/device_totals_xos1_10keV.json
A kernel or march change that alters a single photon changes these integers; one that only reorganises the work does not
(v12 ... v15 of round 2 all reproduce them)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O            # problem construction shared with the tests
from tests.common import make_pair
from tests.emul import pyemul

_, _, prob, _ = make_pair(O, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
runs = []
for seed, slot0, n in ((11, 0, 200_000), (12, 1_000_003, 150_000), (20000, 0, 100_000)):
    e = pyemul.transmission(prob, seed, slot0, n)
    runs.append(dict(seed=seed, slot0=slot0, n_slots=n, counters=[int(c) for c in e["counters"]], sumw_exact=str(e["sumw_exact"])))
    print(runs[-1])
out = os.path.join(ROOT, "tests", "golden", "device_totals_xos1_10keV.json")
json.dump(dict(what="host compile of the device code (tests/emul): counters {exit, not entered, not transmitted, sum of reflections} "
               "and exact sum of floor(w 2^62), xos1 optic, 10 keV, parallel beam; scripts/make_device_totals.py", runs=runs),
          open(out, "w"), indent=1)
