"""Distribution of the certified march's steps over flights (host compile of the device code):
   python scripts/analysis/flight_stats.py [deck] [n_slots]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd
from tests.emul import pyemul
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
deck = sys.argv[1] if len(sys.argv) > 1 else "xos1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=[10.0])
hist, by = pyemul.flight_stats(prob, 20000, 0, n)
print(by)
steps = hist * np.arange(256)
tot = steps.sum()
print("flights %d, march steps %d (%.1f per flight)" % (hist.sum(), tot, tot / hist.sum()))
cf, cs = np.cumsum(hist) / hist.sum(), np.cumsum(steps) / tot
for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 254):
    print("flights of <= %3d steps: %5.1f %% of the flights, %5.1f %% of the steps" % (k, 100 * cf[k], 100 * cs[k]))
