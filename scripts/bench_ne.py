"""Throughput of the trace kernel at other energy-grid sizes / decks (not the headline metric):
python scripts/bench_ne.py [deck] [n_energies] [exit_photons] [sig_rough] [option=value ...] [range=lo:hi]
(the energies are n_energies points from 1 to 30 keV, or from lo to hi)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polycap_amd

deck = sys.argv[1] if len(sys.argv) > 1 else "xos1"
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 291
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
sig = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
opts = [kv.split("=") for kv in sys.argv[5:]]
lo, hi = 1.0, 30.0
for kv in list(opts):
    if kv[0] == "range":
        lo, hi = (float(x) for x in kv[1].split(":"))
        opts.remove(kv)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = [10.0] if ne == 1 else np.linspace(lo, hi, ne)
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=E, sig_rough=sig)
with polycap_amd.TraceContext(prob) as ctx:
    for k, v in opts:
        ctx.set_option(k, int(v))
    ctx.transmission(1, 0, min(n, 20000))
    t0 = time.perf_counter()
    r = ctx.transmission(2, 0, n)
    dt = time.perf_counter() - t0
    st = ctx.phase_stats()
    ctx_kernel = ctx.last_kernel()
print("%s n_E=%d (%g-%g keV) sig=%s %s [%s]: %d exit slots, %d started, kernel %.2f ms, %.4g started photons/s (wall %.3f s), eff[0]=%.4f eff[-1]=%.4f"
      % (deck, ne, lo, hi, sig, opts, ctx_kernel, n, r["i_start"], r["kernel_ms"], r["i_start"] / (r["kernel_ms"] * 1e-3), dt, r["efficiencies"][0], r["efficiencies"][-1]))
print("  avg lanes: march %.1f event %.1f new %.1f; wave-level phases: march steps %.3g, event %.3g, new %.3g"
      % (st["march"]["avg_lanes"], st["event"]["avg_lanes"], st["new"]["avg_lanes"], st["march"]["phases"], st["event"]["phases"], st["new"]["phases"]))
