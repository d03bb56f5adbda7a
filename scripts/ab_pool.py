"""A/B of the LDS photon-pool kernel against the one-photon-per-lane kernel (single energy): bit identity of totals and
image planes, then timing.  python scripts/ab_pool.py [deck] [slots] [key=value ...]   (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polycap_amd

deck = sys.argv[1] if len(sys.argv) > 1 else "xos1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000
opts = [kv.split("=") for kv in sys.argv[3:]]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=[10.0])


def stats(ctx):
    st = ctx.phase_stats()
    return "march %.1f x%d, event %.1f x%d, new %.1f x%d" % (st["march"]["avg_lanes"], st["march"]["phases"], st["event"]["avg_lanes"],
                                                              st["event"]["phases"], st["new"]["avg_lanes"], st["new"]["phases"])


with polycap_amd.TraceContext(prob) as ctx:
    # identity on a small run with images, then on the big one without
    m = min(n, 200000)
    ctx.set_option("pool", 0)
    ref = ctx.transmission(7, 123, m, keep_images=True)
    ctx.set_option("pool", 1)
    for k, v in opts:
        ctx.set_option(k, int(v))
    got = ctx.transmission(7, 123, m, keep_images=True)
    same = np.array_equal(ref["counters"], got["counters"]) and np.array_equal(ref["sumw_fixed"], got["sumw_fixed"])
    same_img = np.array_equal(ref["images"], got["images"], equal_nan=True) and np.array_equal(ref["exit_weights"], got["exit_weights"])
    print("identity (%d slots, images): totals %s, images %s" % (m, same, same_img), flush=True)
    if not (same and same_img):
        print(ref["counters"], got["counters"])
        bad = np.argwhere(ref["images"] != got["images"])
        print("first differing (slot, plane):", bad[:10].tolist())
        sys.exit(1)
    for keep in (False, True):
        ctx.set_option("pool", 0)
        a = ctx.transmission(2, 0, n, keep_images=False)
        ctx.run(2, 0, n, keep_images=keep); ms0 = ctx.wait(); s0 = stats(ctx)
        ctx.set_option("pool", 1)
        ctx.run(2, 0, n, keep_images=keep); ms1 = ctx.wait(); s1 = stats(ctx)
        b = ctx.totals()
        ok = np.array_equal(a["counters"], b["counters"]) and np.array_equal(a["sumw_fixed"], b["sumw_fixed"])
        print("%s %d slots images=%s %s: lane kernel %.2f ms [%s] | pool kernel %.2f ms [%s] | x%.2f | totals identical %s"
              % (deck, n, keep, opts, ms0, s0, ms1, s1, ms0 / ms1, ok), flush=True)
