"""leak_calc=true with many energies (the frames of suspended parents hold a weight per energy): does the run fit, how long does it take?
python scripts/analysis/leak_many_energies.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import polycap_amd
for ne in (7, 40, 291):
    prob = polycap_amd.problem_from_inp('tests/golden/example/xos1.inp', energies=np.linspace(1.0, 30.0, ne))
    with polycap_amd.TraceContext(prob) as ctx:
        t0 = time.perf_counter()
        try:
            r = ctx.transmission(3, 0, 2000, leak_calc=True)
            ext, inn = r["leaks"] if "leaks" in r else ctx.leaks()
            print(ne, "energies: %.1f ms kernel, %.2f s wall, started %d, ext %d int %d events, efficiency %.4f ... %.4f" % (
                r["kernel_ms"], time.perf_counter() - t0, r["i_start"], len(ext), len(inn), r["efficiencies"][0], r["efficiencies"][-1]), flush=True)
        except Exception as e:
            print(ne, "energies:", type(e).__name__, str(e)[:300], flush=True)
