"""Units of work per slot of the leak bench workload (scripts/bench_leak.py: the reference's ellipsoidal test optic, uniform
illumination, 10 keV) on the host compile of the device headers:  python scripts/analysis/leak_units.py [slots]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import polycap_amd
from polycap_amd.decks import optical_constants
from polycap_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
so = "/tmp/libleak_units.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wno-unknown-pragmas",
                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "polycap_amd", "csrc", "hip"),
                       os.path.join(ROOT, "scripts", "analysis", "leak_units.cpp"), "-o", so])
L = C.CDLL(so)
prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
z, cap, ext = prof.get_z(), prof.get_cap(), prof.get_ext()
a, s, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, [10.0])
prob = polycap_amd.Problem(z, cap, ext, 0.0, 200000, 2.23, [10.0], a, s, 2000.0, 0.2065, 0.2065, -1.0, 0.0, 0.0, 0.0, 0.5)
out = np.zeros((n, 10), dtype=np.int64)
L.leak_units.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
rc = L.leak_units(C.byref(prob.s), 20000, 0, n, 532, out.ctypes.data)
assert rc == 0
tot = out[:, :4].sum(axis=1)
print("%d slots: units per slot mean %.0f median %.0f p99 %.0f max %d; attempts per slot %.2f" % (n, tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max(), out[:, 4].mean()))
print("share of all units: march %.2f wall step %.2f probe %.2f other %.2f" % tuple(out[:, k].sum() / tot.sum() for k in range(4)))
order = np.argsort(-tot)[:10]
print("the ten heaviest slots: slot, units (march, wall step, probe, other), attempts, deepest level")
for j in order:
    print("  %7d %9d (%8d %8d %8d %8d) %3d %4d" % (j, tot[j], out[j, 0], out[j, 1], out[j, 2], out[j, 3], out[j, 4], out[j, 5]))
print("share of the total in the heaviest 1 %% of the slots: %.2f; the heaviest slot alone is %.1f x the mean" % (np.sort(tot)[-max(1, n // 100):].sum() / tot.sum(), tot.max() / tot.mean()))
L.leak_stats.restype = C.POINTER(C.c_longlong * 16)
st = list(L.leak_stats().contents)
names = ["wall step: certified block", "wall step: literal step", "cell changes (probes begun)", "probe: 1-segment skip", "probe: %d-segment skip",
         "probe: %d-segment skip", "probe: segment visited, miss", "probe: segment visited, hit", "wall searches begun", "-", "outer hexagon searched", "outer hexagon: nodes tested", "outer hexagon: blocks skipped"]
print("units of the wall search by outcome, per wall search:")
for k, nm in enumerate(names):
    print("  %-36s %10d  %7.2f" % (nm % ((5, 25)[k - 4],) if "%d" in nm else nm, st[k], st[k] / max(1, st[8])))
L.leak_crit.restype = C.POINTER(C.c_longlong * 6)
cr = list(L.leak_crit().contents)
print("critical path of an attempt if every leaked fraction had a lane of its own from its spawn: all attempts %.3g of %.3g units (%.2f);"
      " attempts above 5000 units %.3g of %.3g (%.2f); the longest attempt %d of %d" % (cr[1], cr[0], cr[1] / max(1, cr[0]), cr[3], cr[2], cr[3] / max(1, cr[2]), cr[5], cr[4]))
# would handing out the slots heaviest first (by what a plain run of the same slots sees) shorten a launch?  List scheduling of
# the slots' units on P lanes (every lane takes the next slot when it is done), in slot order against by descending predictor
import heapq
pred = out[:, 7].astype(float)
print("plain-run predictor (reflections + 1 over a slot's attempts) against the slot's units: correlation %.3f; against its longest attempt %.3f"
      % (np.corrcoef(pred, tot)[0, 1], np.corrcoef(pred, out[:, 6])[0, 1]))
for P in (n // 4, n // 8):
    res = []
    for order in (np.arange(n), np.argsort(-pred, kind="stable"), np.argsort(-tot, kind="stable")):
        h = [0] * P
        for j in order:
            heapq.heappush(h, heapq.heappop(h) + int(tot[j]))
        res.append(max(h))
    print("%d lanes for %d slots: makespan in units, slot order %d, heaviest first by the predictor %d, by the true units %d (mean load %d)"
          % (P, n, res[0], res[1], res[2], tot.sum() // P))
# a better predictor?  least squares of the units on (reflections, attempts that hit the glass at the entrance, attempts)
X = np.stack([out[:, 7] - out[:, 9], out[:, 8], out[:, 9] - out[:, 8]], axis=1).astype(float)
coef, *_ = np.linalg.lstsq(X, tot.astype(float), rcond=None)
fit = X @ coef
k = max(1, n // 400)
top_true = set(np.argsort(-tot)[:k])
for name, p_ in (("reflections + 1 per attempt", pred), ("fit %.1f * reflections + %.0f * attempts into the glass + %.0f * other attempts" % tuple(coef), fit)):
    top_p = set(np.argsort(-p_)[:k])
    print("%s: correlation %.3f; of the %d heaviest slots (0.25 %%) it finds %d; the heaviest slot it misses has %d units"
          % (name, np.corrcoef(p_, tot)[0, 1], k, len(top_true & top_p), max([tot[j] for j in top_true - top_p] + [0])))
miss = sorted(top_true - set(np.argsort(-pred)[:k]), key=lambda j: -tot[j])[:12]
rank = np.empty(n, dtype=np.int64); rank[np.argsort(-pred)] = np.arange(n)
print("the heaviest slots the plain predictor misses: units, longest attempt, attempts, deepest level, predictor, its rank")
for j in miss:
    print("  %7d %7d %3d %3d %6d %6d" % (tot[j], out[j, 6], out[j, 4], out[j, 5], out[j, 7], rank[j]))
