"""Fixed cost of a launch of the headline kernel: kernel time against the number of slots, and the latency of a lone photon
(1 slot per launch: time against its number of reflections).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
with polycap_amd.TraceContext(prob) as ctx:
    for k, v in [kv.split("=") for kv in sys.argv[1:]]:
        ctx.set_option(k, int(v))
    ctx.transmission(1, 0, 100000)
    xs, ys = [], []
    for n in (1000, 100000, 1000000, 2500000, 5000000, 10000000, 20000000):
        best = min(ctx.transmission(7, 0, n)["kernel_ms"] for _ in range(3))
        xs.append(n); ys.append(best)
        print("n_slots %9d  kernel %.3f ms" % (n, best))
    a, b = np.polyfit(xs[-4:], ys[-4:], 1)
    print("fit over the four largest: %.3f ms + %.3f ms per 1e6 slots" % (b, a*1e6))
    rows = []
    for seed in range(40):
        r = ctx.transmission(1000 + seed, 0, 1, keep_images=True)
        img = ctx.images()
        rows.append((int(r["counters"][3]), int(r["i_start"]), r["kernel_ms"]))
    rows.sort()
    for irefl, started, ms in rows:
        print("lone photon: %4d reflections (exit photon), %2d started, kernel %.3f ms" % (irefl, started, ms))
