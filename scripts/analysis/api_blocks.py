"""Wall time of polycap_source_get_transmission_efficiencies(1e7) through the public C API for several block sizes of the
compact store (POLYCAP_BLOCK_SHIFT) and with / without the pinned host-plane pool; POLYCAP_TIMING lines on stderr."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from polycap_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
deck = os.path.join(ROOT, "tests", "golden", "example", "xos1.inp")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src0 = capi.Source.new_from_file(deck)
desc = capi.Description(None, 0, 0, None, 0, _handle=capi._lib().polycap_source_get_description(src0._h), _owner=src0)
src = capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0., np.array([10.0]))
os.environ["POLYCAP_TIMING"] = "1"
for shift, pool, compact, streams, depth in ((16, "1", "1", "2", "2"), (15, "1", "1", "2", "2"), (17, "1", "1", "2", "2"), (16, "1", "1", "1", "2"), (16, "0", "1", "2", "2")):
    os.environ["POLYCAP_FETCH_STREAMS"] = streams
    os.environ["POLYCAP_FETCH_DEPTH"] = depth
    os.environ["POLYCAP_BLOCK_SHIFT"] = str(shift)
    os.environ["POLYCAP_HOST_POOL"] = pool
    os.environ["POLYCAP_COMPACT"] = compact
    best = None
    for k in range(4):
        t0 = time.perf_counter()
        eff = src.get_transmission_efficiencies(-1, n)
        dt = time.perf_counter() - t0
        del eff
        if k > 0:
            best = dt if best is None else min(best, dt)
    sys.stderr.flush()
    print("block_shift %d pool %s compact %s streams %s depth %s: best of 3 after a warm-up %.2f ms" % (shift, pool, compact, streams, depth, best * 1e3), file=sys.stderr, flush=True)
