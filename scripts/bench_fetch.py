"""Where the time of a result fetch goes (run on the GPU box): kernel, images to fresh / touched host arrays, and the
public C API end to end.  python scripts/bench_fetch.py [slots]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polycap_amd
from polycap_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inp = os.path.join(root, "tests", "golden", "example", "xos1.inp")
prob = polycap_amd.problem_from_inp(inp, energies=[10.0])
with polycap_amd.TraceContext(prob) as ctx:
    ctx.transmission(1, 0, 100000, keep_images=True)
    for rep in range(2):
        t0 = time.perf_counter(); ctx.run(2, 0, n, keep_images=True); ms = ctx.wait(); t1 = time.perf_counter()
        r = ctx.images(0, n); t2 = time.perf_counter()
        print("rep %d: run+wait %.1f ms (kernel %.1f ms), images() %.1f ms = %.2f GB/s" % (rep, (t1 - t0)*1e3, ms, (t2 - t1)*1e3, 18*8*n/(t2 - t1)/1e9), flush=True)
    for parts in (1, 2, 4, 8, 16):
        ctx.set_option("run_parts", parts)
        ctx.run(2, 0, n, keep_images=True); ms = ctx.wait()
        print("run_parts=%d: kernel(s) %.2f ms" % (parts, ms), flush=True)
src0 = capi.Source.new_from_file(inp)
desc = capi.Description(None, 0, 0, None, 0, _handle=capi._lib().polycap_source_get_description(src0._h), _owner=src0)
src = capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0., np.array([10.0]))
src.get_transmission_efficiencies(-1, 10000)
for m in (1000000, n):
    t0 = time.perf_counter()
    eff = src.get_transmission_efficiencies(-1, m)
    t1 = time.perf_counter()
    print("C API polycap_source_get_transmission_efficiencies(%d): %.1f ms" % (m, (t1 - t0)*1e3), flush=True)
    del eff
