"""Soak of the launching-wave kernel against the default kernel: random sizes, seeds, attempt limits, decks; totals must be
identical every time.  Run on the GPU box:  python scripts/analysis/soak_producer.py [n_cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rng = np.random.default_rng(12345)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
decks = [("xos1", {}), ("ellip_l9", dict(sig_rough=5.0)), ("cone", {}), ("ellip_l9", {})]
ctxs = []
for deck, kw in decks:
    prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=[10.0], **kw)
    ctxs.append((deck, polycap_amd.TraceContext(prob)))
bad = 0
t0 = time.time()
for k in range(ncases):
    deck, ctx = ctxs[k % len(ctxs)]
    n = int(10 ** rng.uniform(0, 6.5))
    seed = int(rng.integers(1, 1 << 40))
    slot0 = int(rng.integers(0, 1 << 30))
    att = int(rng.choice([1, 2, 3, 5, 1 << 20, 1 << 20, 1 << 20]))
    res = []
    for prod in (0, 1):
        ctx.set_option("producer", prod)
        ctx.run(seed, slot0, n, max_attempts=att)
        ctx.wait()
        res.append(ctx.totals(check=False))
    same = np.array_equal(res[0]["counters"], res[1]["counters"]) and np.array_equal(res[0]["sumw_fixed"], res[1]["sumw_fixed"])
    if not same or res[1]["counters"][4] > n:
        bad += 1
        print("MISMATCH", deck, n, seed, slot0, att, res[0]["counters"].tolist(), res[1]["counters"].tolist(), flush=True)
    if k % 10 == 9:
        print("%d cases, %d bad, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
for _, c in ctxs:
    c.close()
print("soak:", ncases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
