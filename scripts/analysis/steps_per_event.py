"""March steps per event of the current device header on the host compile (xos1, 10 keV)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polycap_amd
from tests.emul import pyemul
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
ph = pyemul.sample(prob, 11, np.arange(n), np.zeros(n, dtype=np.uint32))
r = pyemul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
print("steps %d events %d steps/event %.2f  rc==1: %d" % (r["fast_nodes"], r["events"], r["fast_nodes"] / r["events"], (r["rc"] == 1).sum()))
