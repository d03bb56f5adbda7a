"""Kernel time of the headline launch (xos1 10 keV, 1e7 slots, compact planes) for a list of context options, min of 5:
    python scripts/analysis/headline_ab.py name=value[,name=value...] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polycap_amd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
variants = sys.argv[1:] or ["march_stats=0", "march_stats=1"]
with polycap_amd.TraceContext(prob) as ctx:
    ctx.set_option("plane_images", 1)
    ctx.set_option("compact_images", 1)
    ctx.run(1, 0, 10_000_000, keep_images=True)
    ctx.wait()
    for rep in range(2):
        for v in variants:
            for kv in v.split(","):
                k, val = kv.split("=")
                ctx.set_option(k, int(val))
            ms = []
            for i in range(5):
                ctx.run(20000 + i, 0, 10_000_000, keep_images=True)
                ms.append(ctx.wait())
            t = ctx.totals()
            print("%-40s kernel ms min %.2f median %.2f  (%s)  sumw %s" % (v, min(ms), sorted(ms)[2], ctx.last_kernel(), int(t["sumw_fixed"][0, 0])), flush=True)
