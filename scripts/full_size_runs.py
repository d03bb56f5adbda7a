#!/usr/bin/env python3
"""BASELINE configurations at (per-GPU) full size on one MI355X, histogram only: C3 (xos1.inp, 291 energies, 1e8 exit photons),
the per-GPU share of C4 (1e9 / 8) and of C5 (ellip_l9.inp, sig_rough 5 A, 291 energies, 1e9 / 8).  Prints started photons/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polycap_amd

import numpy as np
for name, deck, sig, n, E in (("C3  xos1 291 energies (the deck's grid), 1e8 exit photons", "xos1", None, 100_000_000, None),
                              ("C3  xos1 300 energies (linspace 1..30 keV, BASELINE's 300-bin), 1e8 exit photons", "xos1", None, 100_000_000, np.linspace(1.0, 30.0, 300)),
                              ("C4  xos1 291 energies, 1.25e8 exit photons (1e9 / 8 GPUs)", "xos1", None, 125_000_000, None),
                              ("C5  ellip_l9 291 energies sig 5 A, 1.25e8 exit photons (1e9 / 8 GPUs)", "ellip_l9", 5.0, 125_000_000, None)):
    prob = polycap_amd.problem_from_inp(os.path.join(ROOT, "tests", "golden", "example", deck + ".inp"), sig_rough=sig, energies=E)
    with polycap_amd.TraceContext(prob) as ctx:
        ctx.transmission(1, 0, 100_000)
        t0 = time.perf_counter()
        r = ctx.transmission(20000, 0, n)
        dt = time.perf_counter() - t0
        kernel = ctx.last_kernel()
    print("%s [%s]: kernel %.2f s (wall %.2f s), %d started photons, %.3g started photons/s; efficiency %.4f ... %.4f"
          % (name, kernel, r["kernel_ms"] * 1e-3, dt, r["i_start"], r["i_start"] / (r["kernel_ms"] * 1e-3), r["efficiencies"][0], r["efficiencies"][-1]), flush=True)
