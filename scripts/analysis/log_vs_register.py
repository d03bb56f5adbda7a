"""Experiment (needs a build whose log-kernel condition admits fewer than 9 energies: `kne == 0 &&` dropped from want_log and the
option's lower bound lowered in pc_kernels.hip): the logging kernel against the register-weight kernels at 2 ... 8 energies
(xos1, 1e6 slots, histogram only).  Result, round 4: 2 energies 3.55e8 against 1.17e8 started photons/s, 3: 3.66e8 / 2.11e8,
4: 3.58e8 / 2.11e8, 6: 2.78e8 / 2.10e8, 8: 2.69e8 / 2.15e8 -- the register kernels keep their range."""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for ne in (2, 3, 4, 6, 8):
    prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=np.linspace(5.0, 30.0, ne))
    with polycap_amd.TraceContext(prob) as ctx:
        out, res = [], []
        for lm in (9, 2):
            ctx.set_option("log_min_energies", lm)
            ctx.transmission(1, 0, 200000)
            best = None
            for rep in range(2):
                r = ctx.transmission(2 + rep, 0, 1000000)
                v = r["i_start"] / (r["kernel_ms"] * 1e-3)
                best = v if best is None or v > best else best
            out.append((ctx.last_kernel(), best))
            res.append(r)
        same = np.array_equal(res[0]["counters"][:6], res[1]["counters"][:6])
        print("%d energies:" % ne, ", ".join("%s %.3g/s" % o for o in out), "counters", "equal" if same else "DIFFERENT",
              "sums differ by %.1e" % np.abs(res[1]["sum_weights"]/res[0]["sum_weights"] - 1).max(), flush=True)
