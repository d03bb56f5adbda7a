#!/usr/bin/env python3
"""One leak_calc=true run by itself (for rocprofv3): python3 scripts/leak_one.py [slots] [n_energies: 1 or 7]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polycap_amd
from polycap_amd.decks import optical_constants
from polycap_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
E = [10.0] if (len(sys.argv) <= 2 or sys.argv[2] == "1") else [1., 5., 10., 15., 20., 25., 30.]
prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
z, cap, ext = prof.get_z(), prof.get_cap(), prof.get_ext()
a, s, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, E)
prob = polycap_amd.Problem(z, cap, ext, 0.0, 200000, 2.23, E, a, s, 2000.0, 0.2065, 0.2065, -1.0, 0.0, 0.0, 0.0, 0.5)
with polycap_amd.TraceContext(prob) as ctx:
    seed = 20000
    for k, v in [kv.split("=") for kv in sys.argv[3:]]:
        if k == "seed":
            seed = int(v)
        else:
            ctx.set_option(k, int(v))
    r = ctx.transmission(seed, 0, n, leak_calc=True)
    st = ctx.phase_stats()
print("nE=%d slots=%d kernel %.1f ms started %d ext %d int %d -> %.3g started photons/s; lanes/unit wall %.1f probe %.1f march %.1f; units %.3g %.3g %.3g"
      % (len(E), n, r["kernel_ms"], r["i_start"], len(r["ext"]), len(r["int"]), r["i_start"] / (r["kernel_ms"] * 1e-3),
         st["march"]["avg_lanes"], st["event"]["avg_lanes"], st["new"]["avg_lanes"], st["march"]["phases"], st["event"]["phases"], st["new"]["phases"]))
