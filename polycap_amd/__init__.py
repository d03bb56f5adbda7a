"""polycap_amd -- MI355X-native implementation of polycap's per-photon Monte-Carlo trace path.

The package is a thin Python layer over libpolycap.so (host C + hand-written HIP kernels for gfx950):
  * polycap_amd.hip.TraceContext  -- the thin HIP C-ABI (include/polycap-hip.h): explicit photon batches,
    device-side source sampling and the slot-range transmission driver;
  * polycap_amd.capi              -- ctypes mirror of the reference's C API (polycap_profile/_description/
    _source/_photon/...), i.e. what the reference's Cython module binds;
  * polycap_amd.distributed       -- one-process-per-GPU sharding of the slot range + RCCL reduce of the
    per-energy histogram.
There is no CPU implementation of the trace path in this package.
"""
from ._cabi import Problem, lib  # noqa: F401
from .hip import TraceContext, TraceGroup, HipError, device_count, efficiencies, fixed_to_double, IMG_FIELDS  # noqa: F401
from .decks import problem_from_inp, optical_constants, optical_constants_provider  # noqa: F401

__version__ = "1.2"
