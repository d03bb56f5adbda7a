"""Logging many-energy kernel: kernel time against launch size and log capacity (xos1, 291 energies, histogram only).
python scripts/analysis/log_kernel_sizes.py [deck] [sig] [key=value ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
deck = sys.argv[1] if len(sys.argv) > 1 else "xos1"
sig = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
opts = [kv.split("=") for kv in sys.argv[3:]]
prob = polycap_amd.problem_from_inp('tests/golden/example/%s.inp' % deck, sig_rough=sig)
with polycap_amd.TraceContext(prob) as ctx:
    for k, v in opts:
        ctx.set_option(k, int(v))
    ctx.transmission(1, 0, 20000)
    for cap in (64,):
        ctx.set_option("log_cap", cap)
        for n in (100000, 250000, 500000, 1000000, 2000000, 4000000):
            r = ctx.transmission(2, 0, n)
            st = ctx.sweep_stats()
            print("%s log_cap %3d: %8d slots, kernel %8.2f ms = %6.2f ms per 1e6 slots, %.4g started photons/s; passes %.3g iterations %.3g; waves finish at %.3f of the longest on average"
                  % (deck, cap, n, r["kernel_ms"], r["kernel_ms"]*1e6/n, r["i_start"]/(r["kernel_ms"]*1e-3), st["passes"], st["iterations"], st["wave_life_sum"]/2048.0/max(1, st["wave_life_max"])), flush=True)
