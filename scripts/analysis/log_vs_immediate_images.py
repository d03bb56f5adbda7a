"""As log_vs_immediate.py, with the image planes kept (what polycap_source_get_transmission_efficiencies runs with)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for ne in (9, 12, 24, 32, 40):
    prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=np.linspace(1.0, 30.0, ne))
    with polycap_amd.TraceContext(prob) as ctx:
        ctx.set_option("plane_images", 1)
        ctx.set_option("compact_images", 1)
        out = []
        for mode in (0, 1):
            ctx.set_option("batch_reflections", mode)
            ctx.run(1, 0, 200000, keep_images=True); ctx.wait()
            best = None
            for rep in range(2):
                ctx.run(2 + rep, 0, 1000000, keep_images=True)
                ms = ctx.wait()
                t = ctx.totals()
                v = int(t["counters"][0] + t["counters"][1] + t["counters"][2]) / (ms * 1e-3)
                best = v if best is None or v > best else best
            out.append((ctx.last_kernel(), best))
        print("%d energies, images kept:" % ne, ", ".join("%s %.3g/s" % o for o in out), flush=True)
