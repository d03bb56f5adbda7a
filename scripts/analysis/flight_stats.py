"""March steps per flight for different block-certificate stride sets (CPU analysis on the host compile of the device
header).  python scripts/analysis/flight_stats.py [deck] [photons]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import polycap_amd
from oracle import pyoracle as oracle

so = "/tmp/flight_stats.so"
subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(ROOT, "polycap_amd", "csrc", "hip"), os.path.join(ROOT, "scripts", "analysis", "flight_stats.cpp"), "-o", so])
lib = C.CDLL(so)
deck = sys.argv[1] if len(sys.argv) > 1 else "xos1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", deck + ".inp"), energies=[10.0])
from tests.emul import pyemul

ph = pyemul.sample(prob, 11, np.arange(n), np.zeros(n, dtype=np.uint32))
start, direc, elecv = (np.ascontiguousarray(ph[:, a:a + 3]) for a in (0, 3, 6))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
configs = [((8, 32), 2, 1, 0), ((8, 32), 2, 2, 0), ((4, 16), 2, 2, 0), ((6, 24), 2, 2, 0), ((6, 36), 2, 2, 0), ((5, 25), 2, 2, 0), ((8, 64), 2, 2, 0),
           ((12, 48), 2, 2, 0), ((4, 32), 2, 2, 0), ((4, 24), 2, 2, 0), ((3, 18), 2, 2, 0), ((5, 40), 2, 2, 0), ((6, 48), 2, 2, 0), ((4, 16, 64), 3, 3, 0), ((3, 12, 48), 3, 3, 0)]
if len(sys.argv) > 3:
    configs = [c for c in configs if c[3] == 0]
for strides, lf, ll, npr in configs:
    out = np.zeros(4, dtype=np.int64)
    hf, hl = np.zeros(64, dtype=np.int64), np.zeros(64, dtype=np.int64)
    sa = (C.c_int * len(strides))(*strides)
    kinds = np.zeros(8, dtype=np.int64)
    r = lib.flight_stats(C.byref(prob.s), C.c_int64(n), dp(start), dp(direc), dp(elecv), len(strides), sa, lf, ll, npr, ip(out), ip(hf), ip(hl), ip(kinds))
    assert r == 0
    tot = out[1] + out[3]
    print("strides %-16s probes %d lv first/later %d/%d: first flights %d avg %.1f steps | later flights %d avg %.1f steps | steps per started photon %.1f"
          % (strides, npr, lf, ll, out[0], out[1] / max(1, out[0]), out[2], out[3] / max(1, out[2]), tot / n))
    k = kinds / max(1, out[2])
    print('   per later flight: first-seg %.2f, singles before wide %.2f, wide ok %.2f, wide fail %.2f, singles after wide %.2f, segments %.1f, flights without wide %.2f, scan fail %.2f' % tuple(k[:8]))
    if strides == (8, 32) and ll == 1:
        c = np.cumsum(hl) / max(1, hl.sum())
        print("   later-flight step quantiles: p50 %d p90 %d p99 %d; first: p50 %d p90 %d" % (np.searchsorted(c, .5), np.searchsorted(c, .9), np.searchsorted(c, .99),
              np.searchsorted(np.cumsum(hf) / max(1, hf.sum()), .5), np.searchsorted(np.cumsum(hf) / max(1, hf.sum()), .9)))
