#!/usr/bin/env python3
"""Throughput of polycap_source_get_transmission_efficiencies(leak_calc=true) on the GPU: kernel time against the number
of exit-photon slots (reference optic of tests/leaks.c, parallel beam, uniform illumination)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polycap_amd
from polycap_amd.decks import optical_constants
from polycap_amd import capi

prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
z, cap, ext = prof.get_z(), prof.get_cap(), prof.get_ext()
src = (2000.0, 0.2065, 0.2065, -1.0, 0.0, 0.0, 0.0, 0.5)
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 1024, 16384]
for E in ([10.0], [1., 5., 10., 15., 20., 25., 30.]):
    a, s, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, E)
    prob = polycap_amd.Problem(z, cap, ext, 0.0, 200000, 2.23, E, a, s, *src)
    with polycap_amd.TraceContext(prob) as ctx:
        for n in sizes:
            t0 = time.perf_counter()
            r = ctx.transmission(20000, 0, n, leak_calc=True)
            dt = time.perf_counter() - t0
            st = ctx.phase_stats()      # leak runs: the three pairs are wall steps, capillary probes, march steps
            util = "lanes/unit: wall %.1f probe %.1f march %.1f; units %.3g %.3g %.3g" % (
                st["march"]["avg_lanes"], st["event"]["avg_lanes"], st["new"]["avg_lanes"],
                st["march"]["phases"], st["event"]["phases"], st["new"]["phases"])
            print("nE=%d slots=%6d kernel %9.1f ms wall %7.2f s  started %7d  ext %7d int %7d  -> %.3g started photons/s" %
                  (len(E), n, r["kernel_ms"], dt, r["i_start"], len(r["ext"]), len(r["int"]), r["i_start"] / (r["kernel_ms"] * 1e-3)), flush=True)
            print("      " + util, flush=True)
