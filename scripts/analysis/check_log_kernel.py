"""Logged reflections (option batch_reflections 1, pc_trace_log_kernel) against the immediate sweep (0): same counters; without
roughness the same exact sums, weights and planes bit for bit; with roughness (one exponential per log instead of one per
reflection) weights to 1e-13 and sums to 1e-14.  Then timings of histogram-only runs.
python scripts/analysis/check_log_kernel.py [n_check] [n_time] [key=value ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd

n_check = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
n_time = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
opts = [kv.split("=") for kv in sys.argv[3:]]
for deck, sig, ne in (("xos1", None, 291), ("ellip_l9", 5.0, 291), ("xos1", None, 40)):
    E = None if ne == 291 else np.linspace(10.0, 30.0, ne)
    prob = polycap_amd.problem_from_inp('tests/golden/example/%s.inp' % deck, sig_rough=sig, energies=E)
    with polycap_amd.TraceContext(prob) as ctx:
        for k, v in opts:
            ctx.set_option(k, int(v))
        for keep in (True, False):
            out = {}
            for b in (0, 1):
                ctx.set_option("batch_reflections", b)
                out[b] = ctx.transmission(77, 0, n_check, keep_images=keep)
                print(deck, ne, "images" if keep else "histogram", "mode", b, ctx.last_kernel(), "kernel %.2f ms" % out[b]["kernel_ms"], out[b]["counters"][:6])
            for b in (1,):
                assert np.array_equal(out[0]["counters"][:6], out[b]["counters"][:6]), (deck, b)
                if sig is None:
                    assert np.array_equal(out[0]["sumw_fixed"], out[b]["sumw_fixed"]), (deck, b)
                else:
                    d = np.abs(out[0]["sum_weights"]/out[b]["sum_weights"] - 1.0).max()
                    print("   sums differ by at most %.2e relative" % d)
                    assert d < 1e-14
                if keep:
                    if sig is None:
                        assert np.array_equal(out[0]["exit_weights"], out[b]["exit_weights"]), (deck, b)
                    else:
                        d = np.abs(out[0]["exit_weights"] - out[b]["exit_weights"]) / out[0]["exit_weights"]
                        print("   weights differ by at most %.2e relative" % np.nanmax(d))
                        assert np.nanmax(d) < 1e-13
                    assert np.array_equal(out[0]["images"], out[b]["images"], equal_nan=True), (deck, b)
        # the take-back pass of the fused finalisation: fuse photons whatever their proxies say
        ctx.set_option("batch_reflections", 0)
        ref = ctx.transmission(78, 0, n_check)
        ctx.set_option("batch_reflections", 1)
        for fuse in (0, 2, 1):
            ctx.set_option("sweep_fuse", fuse)
            got = ctx.transmission(78, 0, n_check)
            assert np.array_equal(ref["counters"][:6], got["counters"][:6]), (deck, fuse)
            if sig is None:
                assert np.array_equal(ref["sumw_fixed"], got["sumw_fixed"]), (deck, fuse)
            else:
                assert np.abs(ref["sum_weights"]/got["sum_weights"] - 1.0).max() < 1e-14, (deck, fuse)
        print(deck, ne, "bit-identical", ctx.sweep_stats())
        for b in (1,):
            ctx.set_option("batch_reflections", b)
            ctx.transmission(1, 0, 20000)
            r = ctx.transmission(2, 0, n_time)
            st = ctx.sweep_stats()
            print("  %s nE=%d sig=%s mode %d: %d slots, %d started, kernel %.2f ms, %.4g started photons/s; sweep passes %.3g iterations %.3g"
                  % (deck, ne, sig, b, n_time, r["i_start"], r["kernel_ms"], r["i_start"]/(r["kernel_ms"]*1e-3), st["passes"], st["iterations"]), flush=True)
