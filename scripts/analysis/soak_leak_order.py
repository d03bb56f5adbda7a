"""Soak of the ordered leak launch: random sizes, random orders, random heavy tiers and heavy-lane layouts, two optics; totals and
every event must equal the run in slot order.    python scripts/analysis/soak_leak_order.py [runs, default 40]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import polycap_amd
from polycap_amd.decks import optical_constants
from polycap_amd import capi

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = 0
for optic in (0, 1):
    if optic == 0:
        prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
        n_cap, E = 200000, [10.0]
    else:
        prof = capi.Profile(capi.Profile.CONICAL, 6., 0.15, 0.05, 0.004, 0.0013, 1000., 0.5)
        n_cap, E = 91, [8.0, 17.0]
    a, s, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, E)
    prob = polycap_amd.Problem(prof.get_z(), prof.get_cap(), prof.get_ext(), 0.0, n_cap, 2.23, E, a, s, 2000.0, 0.15, 0.15, -1.0, 0.0, 0.0, 0.0, 0.5)
    with polycap_amd.TraceContext(prob) as ctx:
        for k in range(runs // 2):
            n = int(rng.integers(64, 40000))
            seed = int(rng.integers(1, 1 << 30))
            ctx.set_option("leak_order", 0)
            ctx.leak_set_order(np.zeros(0, dtype=np.uint32), 0)
            ref = ctx.transmission(seed, 0, n, leak_calc=True)
            lanes, every = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            n_heavy = int(rng.integers(0, max(1, n // 20)))
            ctx.set_option("leak_heavy_lanes", lanes)
            ctx.set_option("leak_heavy_every", every)
            order = rng.permutation(n) if k % 3 else np.arange(n)[::-1]
            ctx.leak_set_order(order, n_heavy)
            got = ctx.transmission(seed, 0, n, leak_calc=True)
            ok = (np.array_equal(got["counters"], ref["counters"]) and np.array_equal(got["sum_weights"], ref["sum_weights"])
                  and np.array_equal(got["ext"], ref["ext"]) and np.array_equal(got["int"], ref["int"]))
            bad += not ok
            print("optic %d n %6d heavy %5d on %d lanes of every %d-th wave: %s (%d + %d events)" % (optic, n, n_heavy, lanes, every, "same" if ok else "DIFFERENT", len(got["ext"]), len(got["int"])), flush=True)
print("%d runs, %d different" % (runs // 2 * 2, bad))
sys.exit(1 if bad else 0)
