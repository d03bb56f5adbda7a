import os, sys, time
sys.path.insert(0,'.')
import numpy as np
from polycap_amd import capi
inp='tests/golden/example/xos1.inp'
src0 = capi.Source.new_from_file(inp)
desc = capi.Description(None, 0, 0, None, 0, _handle=capi._lib().polycap_source_get_description(src0._h), _owner=src0)
src = capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0., np.array([10.0]))
src.get_transmission_efficiencies(-1, 100000)
for parts in os.environ.get("PARTS","4").split(","):
    os.environ["POLYCAP_RUN_PARTS"]=parts
    for rep in range(3):
        t0=time.perf_counter(); eff=src.get_transmission_efficiencies(-1, 10000000); dt=time.perf_counter()-t0
        print("parts",parts,"rep",rep,"%.1f ms"%(dt*1e3), file=sys.stderr, flush=True); del eff
