#!/usr/bin/env python3
"""Oracle totals for the identical-seed parity and bias tests.

For K seeds the CPU oracle (oracle/polycap_oracle.c: the reference's literal algorithm) traces exit-photon slots
[0, n) of the xos1 optic at 10 keV and the counters plus the exact fixed-point weight sum (sum of floor(w * 2^62), the
representation the GPU accumulates in) go to tests/golden/oracle_totals_xos1_10keV.json, one entry per seed.  The GPU
tests and scripts/parity_1e8.py trace the same (seed, slot) streams on the device and compare: the oracle leg costs
CPU-hours, the device leg seconds, so the oracle leg is computed once, here, by this script, and committed as data.

    python scripts/make_oracle_totals.py --seeds 128 --slots 1000000 [--threads 7] [--first-seed 1000]

Appends; seeds already present are skipped, so the run can be interrupted and resumed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "oracle_totals_xos1_10keV.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=128)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--slots", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    from oracle import pyoracle as O
    from tests.common import make_pair
    optic, src, prob, (E, A, S) = make_pair(O, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
    doc = {"workload": "xos1 profile tables, 10 keV, pinned constants (amu 42.544635, scatf 0.503696), parallel beam "
                       "(d_source 2000, src_x = src_y = 0.2065, sigma 0, hor_pol 0): BASELINE config C2",
           "generator": "scripts/make_oracle_totals.py (oracle/polycap_oracle.c, gcc -O2 -ffp-contract=off)",
           "slot0": 0, "n_slots": args.slots, "fix_scale_log2": 62, "runs": []}
    if os.path.exists(args.out):
        with open(args.out) as f:
            doc = json.load(f)
        assert doc["n_slots"] == args.slots
    have = {r["seed"] for r in doc["runs"]}
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        if seed in have:
            continue
        t0 = time.time()
        o = O.transmission(optic, src, E, A, S, seed, 0, args.slots, n_threads=args.threads)
        assert o["rc"] == 0
        exact = int(o["sumw_fixed"][0, 0]) + (int(o["sumw_fixed"][0, 1]) << 64)
        doc["runs"].append({"seed": seed, "counters": [int(c) for c in o["counters"]], "sumw_exact": str(exact)})
        tmp = args.out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(doc, f, indent=0)
        os.replace(tmp, args.out)
        print("seed %d: i_start %d eff %.6f (%.1f s)" % (seed, o["i_start"], o["efficiencies"][0], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
