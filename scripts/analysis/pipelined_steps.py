"""Consecutive headline launches (xos1 10 keV, 1e7 slots, compact planes) one after the other on one context against alternating
on two contexts (two streams, two image stores): how much of a launch's 2.8 ms tail the next launch fills.
    python scripts/analysis/pipelined_steps.py [steps, default 12]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polycap_amd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 10_000_000
ctxs = [polycap_amd.TraceContext(prob), polycap_amd.TraceContext(prob)]
for c in ctxs:
    c.set_option("plane_images", 1)
    c.set_option("compact_images", 1)
    c.run(1, 0, N, keep_images=True)
    c.wait()
    c.run(2, 0, N, keep_images=True)
    c.wait()


def started(t):
    return int(t["counters"][0] + t["counters"][1] + t["counters"][2])


for rep in range(2):
    c = ctxs[0]
    c.device_synchronize()
    t0 = time.perf_counter()
    n = 0
    for k in range(K):
        c.run(100 + k, 0, N, keep_images=True)
        c.wait()
        n += started(c.totals())
    c.device_synchronize()
    dt = time.perf_counter() - t0
    print("one context, one launch after the other: %.2f ms per step, %.3g started photons/s" % (dt / K * 1e3, n / dt), flush=True)
    t0 = time.perf_counter()
    n = 0
    for k in range(K):
        ctxs[k % 2].run(100 + k, 0, N, keep_images=True)
        if k >= 1:
            o = ctxs[(k - 1) % 2]
            o.wait()
            n += started(o.totals())
    o = ctxs[(K - 1) % 2]
    o.wait()
    n += started(o.totals())
    ctxs[0].device_synchronize()
    dt = time.perf_counter() - t0
    print("two contexts, launch k+1 enqueued before launch k is waited for: %.2f ms per step, %.3g started photons/s" % (dt / K * 1e3, n / dt), flush=True)
for c in ctxs:
    c.close()
