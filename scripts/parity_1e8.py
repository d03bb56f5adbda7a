#!/usr/bin/env python3
"""Identical-seed parity of the GPU trace path against the CPU oracle at the north-star tolerance (run on the GPU box).

    python scripts/parity_1e8.py [--seeds K] [--live L] [--photons N] [--out profiles/r02/parity_1e8.json]

Three measurements on BASELINE config C2's workload (xos1, 10 keV):

1. Efficiency delta at N >= 1.2e8 started photons.  The oracle leg is tests/golden/oracle_totals_xos1_10keV.json
   (K seeds x 1e6 exit-photon slots traced by oracle/polycap_oracle.c, scripts/make_oracle_totals.py); the device leg
   traces the same (seed, slot) Philox streams through the C-ABI.  `--live L` re-runs the oracle for the first L seeds
   on this host and requires its counters and exact sums to equal the fixture's bit for bit (the fixture is what the
   oracle computes here, not a copy of the device's numbers).
2. Bias: the K per-seed deltas, mean +- standard error, and the noise constant c = std(delta) * sqrt(N_started).
3. Per-photon agreement against the number of reflections: identical explicit photons through the oracle, through the
   oracle with one start coordinate moved by 1 ulp, and through the device.  The trace amplifies rounding differences
   by a constant factor per reflection; the device-vs-oracle curve must grow like the oracle's own 1-ulp curve.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

FIX = os.path.join(ROOT, "tests", "golden", "oracle_totals_xos1_10keV.json")


def exact(fx):
    return int(fx[0, 0]) + (int(fx[0, 1]) << 64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=0, help="seeds of the fixture to use (0 = all)")
    ap.add_argument("--live", type=int, default=2, help="seeds whose oracle leg is recomputed on this host")
    ap.add_argument("--photons", type=int, default=400_000, help="explicit photons of the per-reflection study")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "parity_1e8.json"))
    args = ap.parse_args()
    import polycap_amd
    from oracle import pyoracle as O
    from tests.common import make_pair
    optic, src, prob, (E, A, S) = make_pair(O, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
    with open(FIX) as f:
        fix = json.load(f)
    runs = fix["runs"][:args.seeds] if args.seeds > 0 else fix["runs"]
    n = fix["n_slots"]
    out = {"workload": fix["workload"], "n_slots_per_seed": n, "seeds": len(runs)}

    # 1 + 2: device leg
    rows = []
    t0 = time.time()
    with polycap_amd.TraceContext(prob) as ctx:
        for r in runs:
            g = ctx.transmission(r["seed"], 0, n)
            rows.append((r["seed"], [int(c) for c in g["counters"][:4]], exact(g["sumw_fixed"]), r["counters"], int(r["sumw_exact"])))
    out["device_leg_s"] = time.time() - t0
    d = []
    Sg = So = Ng = No = 0
    for seed, cg, sg, co, so in rows:
        ng, no = cg[0] + cg[1] + cg[2], co[0] + co[1] + co[2]
        d.append((sg / ng) / (so / no) - 1.0)
        Sg += sg; So += so; Ng += ng; No += no
    d = np.array(d)
    nk = No / len(rows)
    out["n_started_oracle"] = No
    out["n_started_device"] = Ng
    out["efficiency_oracle"] = So / 2.0**62 / No
    out["efficiency_device"] = Sg / 2.0**62 / Ng
    out["eff_rel_delta_pooled"] = (Sg / Ng) / (So / No) - 1.0
    out["i_start_rel_delta_pooled"] = Ng / No - 1.0
    out["bias"] = {"mean": float(d.mean()), "standard_error": float(d.std(ddof=1) / np.sqrt(len(d))),
                   "z": float(d.mean() / (d.std(ddof=1) / np.sqrt(len(d)))),
                   "std_per_seed": float(d.std(ddof=1)), "c_noise": float(d.std(ddof=1) * np.sqrt(nk)),
                   "max_abs_per_seed": float(np.abs(d).max()), "positive": int((d > 0).sum()), "negative": int((d < 0).sum())}
    out["per_seed_delta"] = [float(x) for x in d]
    out["north_star_tolerance"] = 1e-4
    out["pass"] = bool(abs(out["eff_rel_delta_pooled"]) <= 1e-4)

    # live check of the fixture on this host
    live = []
    cores = len(os.sched_getaffinity(0))
    for r in runs[:args.live]:
        t0 = time.time()
        o = O.transmission(optic, src, E, A, S, r["seed"], 0, n, n_threads=cores)
        dt = time.time() - t0
        same = [int(c) for c in o["counters"]] == r["counters"] and exact(o["sumw_fixed"]) == int(r["sumw_exact"])
        live.append({"seed": r["seed"], "bit_identical_to_fixture": bool(same), "seconds": dt, "threads": cores,
                     "started_per_s": o["i_start"] / dt})
    out["oracle_live_check"] = live

    # 3: per-photon agreement vs reflection count
    m = args.photons
    with polycap_amd.TraceContext(prob) as ctx:
        ph = ctx.sample_photons(4242, np.arange(m))
        st, di, ev = ph[:, 0:3].copy(), ph[:, 3:6].copy(), ph[:, 6:9].copy()
        g = ctx.launch_photons(st, di, ev)
    a = O.launch_batch(optic, E, A, S, st, di, ev)
    st1 = st.copy()
    st1[:, 0] = np.nextafter(st1[:, 0], 1.0)
    b = O.launch_batch(optic, E, A, S, st1, di, ev)
    traced = np.isin(a["rc"], (0, 1))
    tab = []
    for k in list(range(1, 11)) + [15, 20, 30, 40, 60]:
        sel = traced & (a["i_refl"] == k) if k <= 10 else traced & (a["i_refl"] >= k) & (a["i_refl"] < k + 5)
        if sel.sum() < 50:
            continue
        dg = ((g["rc"][sel] != a["rc"][sel]) | (g["i_refl"][sel] != a["i_refl"][sel])).mean()
        db = ((b["rc"][sel] != a["rc"][sel]) | (b["i_refl"][sel] != a["i_refl"][sel])).mean()
        # among photons with the same outcome: how far apart are the exit points (cm); grows by a constant factor per reflection
        same_g = sel & (g["rc"] == a["rc"]) & (g["i_refl"] == a["i_refl"])
        same_b = sel & (b["rc"] == a["rc"]) & (b["i_refl"] == a["i_refl"])
        med = lambda x, mk: float(np.median(np.abs(x["exit_coords"][mk] - a["exit_coords"][mk]).max(axis=1))) if mk.sum() else None
        tab.append({"reflections": k if k <= 10 else "%d-%d" % (k, k + 4), "photons": int(sel.sum()),
                    "disagree_device_vs_oracle": float(dg), "disagree_oracle_1ulp_vs_oracle": float(db),
                    "median_exit_shift_device": med(g, same_g), "median_exit_shift_oracle_1ulp": med(b, same_b)})
    out["per_reflection"] = tab
    out["flip_rate_device_vs_oracle"] = float(((g["rc"] != a["rc"]) | (g["i_refl"] != a["i_refl"])).mean())
    out["flip_rate_oracle_1ulp"] = float(((b["rc"] != a["rc"]) | (b["i_refl"] != a["i_refl"])).mean())
    sa = a["weights"][a["rc"] == 1, 0].sum()
    out["explicit_weight_delta_device"] = float(g["weights"][g["rc"] == 1, 0].sum() / sa - 1.0)
    out["explicit_weight_delta_oracle_1ulp"] = float(b["weights"][b["rc"] == 1, 0].sum() / sa - 1.0)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    brief = {k: out[k] for k in ("seeds", "n_started_oracle", "efficiency_oracle", "efficiency_device", "eff_rel_delta_pooled",
                                 "bias", "pass", "oracle_live_check", "flip_rate_device_vs_oracle", "flip_rate_oracle_1ulp")}
    print(json.dumps(brief, indent=1))
    for t in tab:
        print(t)


if __name__ == "__main__":
    main()
