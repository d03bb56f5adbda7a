"""Does handing out the slots of a leak run heaviest first, the heaviest ones to a few lanes of every fourth wave, shorten it?
Run 1 counts the units of work per slot; run 2 of the same slots is ordered by them (the best any predictor could do).
    python scripts/analysis/leak_order_ab.py [slots] [heavy tier as a fraction of the slots, default 0.01] [heavy lanes] [every]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import polycap_amd
from polycap_amd.decks import optical_constants
from polycap_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
every = int(sys.argv[4]) if len(sys.argv) > 4 else 4
prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
a, s, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, [10.0])
prob = polycap_amd.Problem(prof.get_z(), prof.get_cap(), prof.get_ext(), 0.0, 200000, 2.23, [10.0], a, s, 2000.0, 0.2065, 0.2065, -1.0, 0.0, 0.0, 0.0, 0.5)
with polycap_amd.TraceContext(prob) as ctx:
    ctx.set_option("leak_slot_units", 1)
    ctx.transmission(1, 0, 4096, leak_calc=True)
    r0 = ctx.transmission(20000, 0, n, leak_calc=True)
    units = ctx.leak_slot_units(0, n).astype(np.int64)
    print("slot order: kernel %.1f ms; units per slot mean %.0f max %d" % (r0["kernel_ms"], units.mean(), units.max()), flush=True)
    order = np.argsort(-units, kind="stable")
    cfgs = [(0.0, 0, 0, 0), (frac, lanes, every, 0)] if len(sys.argv) > 2 else [(0.0, 0, 0, 0), (0.001, 1, 1, 0), (0.0025, 1, 1, 0), (0.004, 1, 1, 0), (0.0025, 1, 2, 0), (0.0025, 2, 2, 0)]
    for f, l, e, comp in cfgs:
        ctx.set_option("leak_heavy_lanes", l)
        ctx.set_option("leak_heavy_every", e)
        ctx.leak_set_order(order, int(f * n))
        r = ctx.transmission(20000, 0, n, leak_calc=True)
        same = (r["i_start"] == r0["i_start"] and len(r["ext"]) == len(r0["ext"]) and np.array_equal(r["int"], r0["int"]) and np.array_equal(r["ext"], r0["ext"]))
        print("heaviest first, heavy tier %.4f of the slots on %d lanes of every %d-th wave: kernel %.1f ms (%.3g started photons/s); results identical: %s"
              % (f, l, e, r["kernel_ms"], r["i_start"] / (r["kernel_ms"] * 1e-3), same), flush=True)
