"""Batched reflections of the many-energy kernel (option batch_reflections) against the immediate sweep: same bits."""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for deck, sig, n in (("xos1", None, 300000), ("ellip_l9", 5.0, 200000)):
    prob = polycap_amd.problem_from_inp('tests/golden/example/%s.inp' % deck, sig_rough=sig)
    out = []
    with polycap_amd.TraceContext(prob) as ctx:
        for b in (0, 1):
            ctx.set_option("batch_reflections", b)
            r = ctx.transmission(77, 0, n, keep_images=True)
            out.append(r)
            print(deck, "batch", b, "kernel %.2f ms, %.4g started/s" % (r["kernel_ms"], r["i_start"] / r["kernel_ms"] * 1e3), r["counters"][:6])
    a, b = out
    assert np.array_equal(a["counters"][:4], b["counters"][:4])
    assert np.array_equal(a["sumw_fixed"], b["sumw_fixed"])
    assert np.array_equal(a["exit_weights"], b["exit_weights"]) and np.array_equal(a["images"], b["images"], equal_nan=True)
    print(deck, "bit-identical")
