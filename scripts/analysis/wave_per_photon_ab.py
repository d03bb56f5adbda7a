"""A/B of the photon-to-hardware mapping: one lane per photon with wave-wide phases (the product's
kernels) against one wave per photon (pc_wave_kernel.h, the north_star's wording), xos1 at 10 keV, histogram only.
    POLYCAP_EXPERIMENTS=1 python -m polycap_amd._build --force     # the experiment kernel is not part of the product build
    python scripts/analysis/wave_per_photon_ab.py [slots]      -> table on stdout (profiles/r03/wave_per_photon_ab.txt)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
rows = []
ref = None
with polycap_amd.TraceContext(prob) as ctx:
    for name, opts in (("lane per photon, launching wave (v17, the default)", dict(producer=1, wave_per_photon=0)),
                       ("lane per photon, lane kernel (v15)", dict(producer=0, wave_per_photon=0)),
                       ("wave per photon (pc_wave_kernel.h)", dict(producer=0, wave_per_photon=1))):
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.transmission(1, 0, min(n, 100_000))
        best = None
        for rep in range(3):
            r = ctx.transmission(20000, 0, n)
            best = r["kernel_ms"] if best is None else min(best, r["kernel_ms"])
        st = ctx.phase_stats()
        key = (tuple(int(c) for c in r["counters"][:4]), tuple(int(x) for x in r["sumw_fixed"].ravel()))
        if ref is None:
            ref = key
        rows.append((name, ctx.last_kernel(), best, r["i_start"] / (best * 1e-3), key == ref, st))
print("xos1.inp, 10 keV, %d exit-photon slots (%d started photons), histogram only, kernel time by HIP events, best of 3" % (n, r["i_start"]))
print("%-52s %-26s %10s %16s %s" % ("mapping", "kernel", "ms", "started/s", "totals == first row"))
for name, kern, ms, rate, same, st in rows:
    print("%-52s %-26s %10.2f %16.4g %s" % (name, kern, ms, rate, "bit-identical" if same else "DIFFERENT"))
name, kern, ms, rate, same, st = rows[-1]
print("wave per photon: %.3g 64-node scans and %.3g EVENT visits by whole waves (one photon each); the lane kernels run an EVENT phase for "
      "~55 photons at once" % (st["march"]["phases"], st["event"]["phases"]))
print("ratio wave-per-photon / default: %.1fx slower" % (rows[-1][2] / rows[0][2]))
