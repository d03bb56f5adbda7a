"""Kernel time of the headline launch against one context option (run on the GPU box):
   python scripts/analysis/opt_scan.py event_march 0 2 4 6 8 [other=value ...]
Totals must not depend on the option (photon results depend on (seed, slot, attempt) only): checked against the first value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if "=" not in v]
fixed = [kv.split("=") for kv in sys.argv[2:] if "=" in kv]
n = int(os.environ.get("SLOTS", "10000000"))
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", os.environ.get("DECK", "xos1") + ".inp"), energies=[10.0])
ref = None
with polycap_amd.TraceContext(prob) as ctx:
    ctx.set_option("plane_images", 1)
    for k, v in fixed:
        ctx.set_option(k, int(v))
    ctx.transmission(1, 0, 100000)
    for v in vals:
        ctx.set_option(name, v)
        ms = []
        for _ in range(4):
            ctx.run(20000, 0, n, keep_images=True)
            ms.append(ctx.wait())
            r = ctx.totals()
        st = ctx.phase_stats()
        key = (r["counters"][:4].tolist(), r["sumw_fixed"].tolist())
        if ref is None: ref = key
        print("%s=%d: kernel %.2f ms (min of 4; %.2f median), march %.3g steps at %.1f lanes, event %.3g at %.1f, new %.3g at %.1f, %s"
              % (name, v, min(ms), sorted(ms)[2], st["march"]["phases"], st["march"]["avg_lanes"], st["event"]["phases"], st["event"]["avg_lanes"],
                 st["new"]["phases"], st["new"]["avg_lanes"], "same totals" if key == ref else "TOTALS DIFFER"), flush=True)
