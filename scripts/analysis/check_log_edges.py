"""Logging kernel at the edges of its range: 33 energies, 448 / 449 (where the immediate kernel's constants stop
fitting in LDS), 700 ... 1400 (the log capacity is halved until a log per wave fits beside the constants; beyond that the immediate
sweep runs), log capacities 1 ... 255: counters and exact sums equal the immediate sweep's."""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for ne in (33, 448, 449, 700, 1000, 1200, 1400):
    prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=np.linspace(2.0, 40.0, ne))
    with polycap_amd.TraceContext(prob) as ctx:
        ctx.set_option("batch_reflections", 0)
        a = ctx.transmission(5, 3, 30000)
        print(ne, 'immediate', ctx.last_kernel(), '%.2f ms' % a['kernel_ms'])
        ctx.set_option("batch_reflections", 1)
        for cap in (0, 1, 7, 255):
            ctx.set_option("log_cap", cap)
            b = ctx.transmission(5, 3, 30000)
            print(ne, "log_cap", cap, ctx.last_kernel(), "%.2f ms" % b["kernel_ms"], b["counters"][:4])
            assert np.array_equal(a["counters"][:6], b["counters"][:6]) and np.array_equal(a["sumw_fixed"], b["sumw_fixed"]), (ne, cap)
print("ok")
