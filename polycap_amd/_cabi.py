"""ctypes declarations of the thin HIP C-ABI (include/polycap-hip.h) and of libpolycap's loader.

The library is built in-tree (polycap_amd/lib/libpolycap.so) by polycap_amd._build / __graft_entry__.build().
There is no Python or CPU fallback for the trace path: if the shared library is missing, import fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POLYCAP_AMD_LIB") or os.path.join(_HERE, "lib", "libpolycap.so")   # override: A/B builds only

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)

PC_HIP_OK = 0
PC_HIP_ERR_NO_DEVICE = -1
PC_HIP_ERR_INVALID = -2
PC_HIP_ERR_RUNTIME = -3
PC_HIP_ERR_MEMORY = -4
PC_HIP_ERR_ATTEMPTS = -5
PC_HIP_LEAK_HDR = 12   # doubles before the weights of one leak event (include/polycap-hip.h)


class ErrS(C.Structure):
    """polycap_error (include/polycap.h); the one ctypes type every binding module uses for it"""
    _fields_ = [("code", C.c_int), ("message", C.c_char_p)]


class ProblemS(C.Structure):
    """struct pc_hip_problem"""
    _fields_ = [("nmax", C.c_int32), ("z", c_double_p), ("cap", c_double_p), ("ext", c_double_p),
                ("sig_rough", C.c_double), ("n_cap", C.c_int64), ("density", C.c_double),
                ("n_energies", C.c_size_t), ("energies", c_double_p), ("amu", c_double_p), ("scatf", c_double_p),
                ("d_source", C.c_double), ("src_x", C.c_double), ("src_y", C.c_double),
                ("src_sigx", C.c_double), ("src_sigy", C.c_double),
                ("src_shiftx", C.c_double), ("src_shifty", C.c_double), ("hor_pol", C.c_double)]


class ImagesS(C.Structure):
    """struct pc_hip_images"""
    _fields_ = [("src_start_coords", c_double_p * 2), ("pc_start_coords", c_double_p * 2),
                ("pc_start_dir", c_double_p * 2), ("pc_start_elecv", c_double_p * 2),
                ("pc_exit_coords", c_double_p * 3), ("pc_exit_dir", c_double_p * 2),
                ("pc_exit_elecv", c_double_p * 2), ("pc_exit_nrefl", c_int64_p),
                ("pc_exit_dtravel", c_double_p), ("exit_coord_weights", c_double_p)]


def dptr(a):
    return a.ctypes.data_as(c_double_p)


# column order of the [n, 17] image array of TraceContext.images(): planes of struct pc_hip_images, nrefl as column 15
IMAGE_PLANES = [("src_start_coords", 2), ("pc_start_coords", 2), ("pc_start_dir", 2), ("pc_start_elecv", 2),
                ("pc_exit_coords", 3), ("pc_exit_dir", 2), ("pc_exit_elecv", 2)]


def images_struct(planes, nrefl, weights):
    """pc_hip_images over planes [17, n] (row 16 = d_travel; row 15 unused), nrefl [n] int64, weights [n, nE]."""
    s = ImagesS()
    k = 0
    for name, m in IMAGE_PLANES:
        arr = getattr(s, name)
        for j in range(m):
            arr[j] = dptr(planes[k])
            k += 1
    s.pc_exit_nrefl = nrefl.ctypes.data_as(c_int64_p)
    s.pc_exit_dtravel = dptr(planes[16])
    s.exit_coord_weights = dptr(weights)
    return s


class Problem:
    """Owns the numpy arrays behind a pc_hip_problem."""

    def __init__(self, z, cap, ext, sig_rough, n_cap, density, energies, amu, scatf,
                 d_source=2000.0, src_x=0.2065, src_y=0.2065, src_sigx=0.0, src_sigy=0.0,
                 src_shiftx=0.0, src_shifty=0.0, hor_pol=0.0):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.z, self.cap, self.ext = f(z), f(cap), f(ext)
        self.energies, self.amu, self.scatf = f(energies).ravel(), f(amu).ravel(), f(scatf).ravel()
        if not (self.z.shape == self.cap.shape == self.ext.shape and self.z.ndim == 1):
            raise ValueError("z, cap, ext must be 1-D arrays of equal length")
        if not (self.energies.shape == self.amu.shape == self.scatf.shape):
            raise ValueError("energies, amu, scatf must have equal length")
        self.nmax = self.z.shape[0] - 1
        self.n_energies = self.energies.shape[0]
        self.sig_rough, self.n_cap, self.density = float(sig_rough), int(n_cap), float(density)
        self.source = (float(d_source), float(src_x), float(src_y), float(src_sigx), float(src_sigy),
                       float(src_shiftx), float(src_shifty), float(hor_pol))
        self.s = ProblemS(self.nmax, dptr(self.z), dptr(self.cap), dptr(self.ext),
                          self.sig_rough, self.n_cap, self.density,
                          self.n_energies, dptr(self.energies), dptr(self.amu), dptr(self.scatf), *self.source)


_LIB = None


def lib():
    """Loads libpolycap.so (host C + HIP kernels). Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "polycap_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the trace path." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    L.pc_hip_device_count.restype = C.c_int
    L.pc_hip_last_error.restype = C.c_char_p
    L.pc_hip_ctx_create.argtypes = [P(ProblemS), C.c_int, P(C.c_void_p)]
    L.pc_hip_ctx_create.restype = C.c_int
    L.pc_hip_ctx_destroy.argtypes = [C.c_void_p]
    L.pc_hip_ctx_destroy.restype = None
    L.pc_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pc_hip_set_option.restype = C.c_int
    L.pc_hip_launch_photons.argtypes = [C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p,
                                        P(C.c_int32), c_double_p, c_double_p, c_double_p, c_double_p,
                                        c_int64_p, c_double_p]
    L.pc_hip_launch_photons.restype = C.c_int
    L.pc_hip_launch_photons_leak.argtypes = L.pc_hip_launch_photons.argtypes
    L.pc_hip_launch_photons_leak.restype = C.c_int
    L.pc_hip_transmission_run_leak.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_uint32, C.c_int]
    L.pc_hip_transmission_run_leak.restype = C.c_int
    L.pc_hip_leak_counts.argtypes = [C.c_void_p, c_int64_p, c_int64_p]
    L.pc_hip_leak_counts.restype = C.c_int
    L.pc_hip_leak_events.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, c_double_p]
    L.pc_hip_leak_events.restype = C.c_int
    L.pc_hip_leak_events_view.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_double)), c_int64_p]
    L.pc_hip_leak_events_view.restype = C.c_int
    L.pc_hip_sample_photons.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, c_int64_p, P(C.c_uint32), c_double_p]
    L.pc_hip_sample_photons.restype = C.c_int
    L.pc_hip_transmission_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_uint32, C.c_int]
    L.pc_hip_transmission_run.restype = C.c_int
    L.pc_hip_transmission_wait.argtypes = [C.c_void_p, P(C.c_float)]
    L.pc_hip_transmission_wait.restype = C.c_int
    L.pc_hip_transmission_totals.argtypes = [C.c_void_p, c_double_p, c_int64_p, P(C.c_uint64)]
    L.pc_hip_transmission_totals.restype = C.c_int
    L.pc_hip_transmission_images.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P(ImagesS)]
    L.pc_hip_transmission_images.restype = C.c_int
    L.pc_hip_transmission_records.argtypes = [C.c_void_p, C.c_int64, C.c_int64, P(C.c_double)]
    L.pc_hip_transmission_records.restype = C.c_int
    L.pc_hip_transmission_slot_ids.argtypes = [C.c_void_p, C.c_int64, C.c_int64, c_int64_p]
    L.pc_hip_transmission_slot_ids.restype = C.c_int
    L.pc_hip_leak_set_order.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int64, C.c_int64]
    L.pc_hip_leak_set_order.restype = C.c_int
    L.pc_hip_leak_slot_units.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint32)]
    L.pc_hip_leak_slot_units.restype = C.c_int
    L.pc_hip_phase_stats.argtypes = [C.c_void_p, c_int64_p]
    L.pc_hip_phase_stats.restype = C.c_int
    L.pc_hip_last_kernel.argtypes = [C.c_void_p]
    L.pc_hip_last_kernel.restype = C.c_int
    L.pc_hip_sweep_stats.argtypes = [C.c_void_p, c_int64_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pc_hip_sweep_stats.restype = C.c_int
    L.pc_hip_device_synchronize.argtypes = [C.c_void_p]
    L.pc_hip_device_synchronize.restype = C.c_int
    L.pc_hip_group_create.argtypes = [P(ProblemS), C.c_int, P(C.c_int), P(C.c_void_p)]
    L.pc_hip_group_create.restype = C.c_int
    L.pc_hip_group_destroy.argtypes = [C.c_void_p]
    L.pc_hip_group_destroy.restype = None
    L.pc_hip_group_size.argtypes = [C.c_void_p]
    L.pc_hip_group_size.restype = C.c_int
    L.pc_hip_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.pc_hip_group_set_option.restype = C.c_int
    L.pc_hip_group_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_int]
    L.pc_hip_group_run.restype = C.c_int
    L.pc_hip_group_last_kernel.argtypes = [C.c_void_p, C.c_int]
    L.pc_hip_group_last_kernel.restype = C.c_int
    L.pc_hip_group_images.argtypes = [C.c_void_p, P(ImagesS)]
    L.pc_hip_group_images.restype = C.c_int
    L.pc_hip_group_totals.argtypes = [C.c_void_p, C.c_int, c_double_p, c_int64_p, P(C.c_uint64), P(C.c_int), P(C.c_float)]
    L.pc_hip_group_totals.restype = C.c_int
    L.pc_hip_efficiencies.argtypes = [C.c_size_t, c_double_p, c_int64_p, c_double_p]
    L.pc_hip_efficiencies.restype = None
    L.pc_hip_fixed_to_double.argtypes = [C.c_uint64, C.c_uint64]
    L.pc_hip_fixed_to_double.restype = C.c_double
    L.pc_transmission_efficiencies_from_totals.argtypes = [C.c_void_p, C.c_int64, c_double_p, c_int64_p, P(ImagesS), C.c_void_p]
    L.pc_transmission_efficiencies_from_totals.restype = C.c_void_p
    _LIB = L
    return L
