"""From how many energies on does the logging kernel pay?  xos1, linspace(1, 30 keV, n), 1e6 slots, histogram only:
immediate sweep (batch_reflections 0) against logged reflections (1, option log_min_energies 9); counters and exact sums compared."""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for lo, hi in ((1.0, 30.0), (10.0, 30.0)):
  for ne in (9, 12, 16, 24, 32, 33, 40, 64):
    prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=np.linspace(lo, hi, ne))
    with polycap_amd.TraceContext(prob) as ctx:
        ctx.set_option("log_min_energies", 9)
        out, res = [], []
        for mode in (0, 1):
            ctx.set_option("batch_reflections", mode)
            ctx.transmission(1, 0, 200000)
            best = None
            for rep in range(2):
                r = ctx.transmission(2 + rep, 0, 1000000)
                v = r["i_start"] / (r["kernel_ms"] * 1e-3)
                best = v if best is None or v > best else best
            out.append((ctx.last_kernel(), best))
            res.append(r)
        same = np.array_equal(res[0]["counters"][:6], res[1]["counters"][:6]) and np.array_equal(res[0]["sumw_fixed"], res[1]["sumw_fixed"])
        print("%g-%g keV, %d energies:" % (lo, hi, ne), ", ".join("%s %.3g/s" % o for o in out), "identical" if same else "DIFFERENT", flush=True)
