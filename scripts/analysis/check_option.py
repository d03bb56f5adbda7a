"""A kernel option against the default kernel: totals, counters and every image plane must be the same bits (run on the GPU
box):  python scripts/analysis/check_option.py producer"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
opt = sys.argv[1]
ok = True
for deck, n, att, kw in (("xos1", 300_001, 1 << 20, {}), ("ellip_l9", 200_000, 1 << 20, dict(sig_rough=5.0)), ("cone", 50_000, 1 << 20, {}),
                         ("xos1", 100_000, 2, {}), ("xos1", 37, 1 << 20, {})):
    prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=[10.0], **kw)
    res = []
    for q in (0, 1):
        with polycap_amd.TraceContext(prob) as ctx:
            ctx.set_option(opt, q)
            ctx.run(77, 5, n, max_attempts=att, keep_images=True)
            ms = ctx.wait()
            r = ctx.totals(check=False)       # with two attempts per slot some slots stay empty: part of the comparison
            r.update(ctx.images(0, n))
            res.append(r)
    a, b = res
    done = a["exit_weights"][:, 0] > 0        # a slot that ran out of attempts has weight 0 and no defined exit planes
    same = (np.array_equal(a["counters"], b["counters"]) and np.array_equal(a["sumw_fixed"], b["sumw_fixed"])
            and np.array_equal(a["images"][done], b["images"][done], equal_nan=True) and np.array_equal(a["exit_weights"], b["exit_weights"]))
    print(deck, n, "counters", a["counters"].tolist(), b["counters"].tolist(), "identical" if same else "DIFFERENT", flush=True)
    ok = ok and same
sys.exit(0 if ok else 1)
