"""Why is the C API call slower inside bench.py's process?  Variants of the sequence (run on the GPU box)."""
import os, sys, time
sys.path.insert(0,'.')
import numpy as np
import polycap_amd
from polycap_amd import capi
inp='tests/golden/example/xos1.inp'
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
ctx = None
if mode in ("ctx", "ctxsteps", "ctxplanes"):
    prob = polycap_amd.problem_from_inp(inp, energies=[10.0])
    ctx = polycap_amd.TraceContext(prob, 0)
    if mode == "ctxplanes": ctx.set_option("plane_images", 1)
    if mode in ("ctxsteps", "ctxplanes"):
        for k in range(4):
            ctx.run(20000 + k, 0, 10_000_000, keep_images=True); ctx.wait(); ctx.totals()
src0 = capi.Source.new_from_file(inp)
desc = capi.Description(None, 0, 0, None, 0, _handle=capi._lib().polycap_source_get_description(src0._h), _owner=src0)
src = capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0., np.array([10.0]))
for rep in range(4):
    t0=time.perf_counter(); eff=src.get_transmission_efficiencies(-1, 10000000); dt=time.perf_counter()-t0
    print(mode, "rep", rep, "%.1f ms" % (dt*1e3), file=sys.stderr, flush=True); del eff
