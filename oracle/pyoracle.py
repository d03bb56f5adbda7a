"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under polycap_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tup(self):
        return (self.x, self.y, self.z)


class OpticS(C.Structure):
    _fields_ = [("nmax", C.c_int), ("z", c_double_p), ("cap", c_double_p), ("ext", c_double_p),
                ("sig_rough", C.c_double), ("n_cap", C.c_int64), ("density", C.c_double)]


class SourceS(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("d_source", "src_x", "src_y", "src_sigx", "src_sigy",
                                          "src_shiftx", "src_shifty", "hor_pol")]


class LeakS(C.Structure):
    _fields_ = [("coords", Vec3), ("direction", Vec3), ("elecv", Vec3), ("n_refl", C.c_int64),
                ("n_energies", C.c_size_t), ("weight", c_double_p)]


class PhotonS(C.Structure):
    _fields_ = [("start_coords", Vec3), ("start_direction", Vec3), ("start_electric_vector", Vec3),
                ("exit_coords", Vec3), ("exit_direction", Vec3), ("exit_electric_vector", Vec3),
                ("src_start_coords", Vec3),
                ("n_energies", C.c_size_t), ("energies", c_double_p), ("weight", c_double_p),
                ("amu", c_double_p), ("scatf", c_double_p),
                ("i_refl", C.c_int64), ("d_travel", C.c_double),
                ("leak_calc", C.c_int), ("extleak", C.POINTER(LeakS)), ("intleak", C.POINTER(LeakS)),
                ("n_extleak", C.c_int64), ("n_intleak", C.c_int64)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("polycap_oracle.c", "polycap_oracle_leak.c", "polycap_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_within_pc_boundary.argtypes = [C.c_double, Vec3]
        L.orc_within_pc_boundary.restype = C.c_int
        L.orc_n_shells.argtypes = [C.c_int64]
        L.orc_n_shells.restype = C.c_double
        L.orc_open_area.argtypes = [C.POINTER(OpticS)]
        L.orc_open_area.restype = C.c_double
        L.orc_profile_new.argtypes = [C.c_int] + [C.c_double] * 7 + [C.c_int, c_double_p, c_double_p, c_double_p]
        L.orc_profile_new.restype = C.c_int
        L.orc_segment.argtypes = [Vec3, Vec3, C.c_double, C.c_double, Vec3, Vec3, Vec3, C.POINTER(Vec3), C.POINTER(Vec3)]
        L.orc_segment.restype = C.c_int
        L.orc_refl_polar.argtypes = [C.c_double] * 4 + [Vec3, C.POINTER(PhotonS), C.POINTER(Vec3)]
        L.orc_refl_polar.restype = C.c_double
        L.orc_reflect.argtypes = [C.POINTER(OpticS), C.POINTER(PhotonS), Vec3]
        L.orc_reflect.restype = C.c_int
        L.orc_trace.argtypes = [C.POINTER(OpticS), C.POINTER(C.c_int), C.POINTER(PhotonS), c_double_p, c_double_p]
        L.orc_trace.restype = C.c_int
        L.orc_launch_one.argtypes = [C.POINTER(OpticS), C.c_size_t, c_double_p, c_double_p, c_double_p,
                                     c_double_p, c_double_p, c_double_p,
                                     c_double_p, c_double_p, c_double_p, c_double_p, c_int64_p, c_double_p]
        L.orc_launch_one.restype = C.c_int
        L.orc_launch_batch.argtypes = [C.POINTER(OpticS), C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_int64,
                                       c_double_p, c_double_p, c_double_p,
                                       C.POINTER(C.c_int), c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_int64_p, c_double_p, C.c_int]
        L.orc_launch_batch.restype = None
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32_10.restype = None
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_uniform.restype = C.c_double
        L.orc_sample_photon_flat.argtypes = [C.POINTER(OpticS), C.POINTER(SourceS), C.c_uint64, C.c_uint64, C.c_uint32, c_double_p]
        L.orc_sample_photon_flat.restype = None
        L.orc_transmission.argtypes = [C.POINTER(OpticS), C.POINTER(SourceS), C.c_size_t, c_double_p, c_double_p, c_double_p,
                                       C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_uint32,
                                       c_double_p, c_int64_p, c_double_p, c_double_p]
        L.orc_transmission.restype = C.c_int
        L.orc_transmission_fixed.argtypes = L.orc_transmission.argtypes + [C.POINTER(C.c_uint64)]
        L.orc_transmission_fixed.restype = C.c_int
        L.orc_efficiencies.argtypes = [C.c_size_t, c_double_p, c_int64_p, c_double_p]
        L.orc_efficiencies.restype = None
        L.orc_trace_wall.argtypes = [C.POINTER(OpticS), C.POINTER(PhotonS), c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_trace_wall.restype = C.c_int
        L.orc_pc_intersect.argtypes = [C.POINTER(OpticS), Vec3, Vec3, C.POINTER(Vec3)]
        L.orc_pc_intersect.restype = C.c_int
        L.orc_photon_clear_leaks.argtypes = [C.POINTER(PhotonS)]
        L.orc_photon_clear_leaks.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        pp, ip = C.POINTER(c_double_p), c_int64_p
        L.orc_launch_one_leak.argtypes = L.orc_launch_one.argtypes + [pp, ip, pp, ip]
        L.orc_launch_one_leak.restype = C.c_int
        L.orc_transmission_leak.argtypes = L.orc_transmission.argtypes + [pp, ip, pp, ip]
        L.orc_transmission_leak.restype = C.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def vec(t):
    return Vec3(float(t[0]), float(t[1]), float(t[2]))


class Optic:
    """Profile arrays + glass parameters; keeps the numpy arrays alive behind the C struct."""

    def __init__(self, z, cap, ext, sig_rough, n_cap, density):
        self.z = np.ascontiguousarray(z, dtype=np.float64)
        self.cap = np.ascontiguousarray(cap, dtype=np.float64)
        self.ext = np.ascontiguousarray(ext, dtype=np.float64)
        assert self.z.shape == self.cap.shape == self.ext.shape
        self.nmax = self.z.shape[0] - 1
        self.sig_rough, self.n_cap, self.density = float(sig_rough), int(n_cap), float(density)
        self.s = OpticS(self.nmax, _dp(self.z), _dp(self.cap), _dp(self.ext), self.sig_rough, self.n_cap, self.density)

    @classmethod
    def from_shape(cls, ptype, length, rext_up, rext_down, rint_up, rint_down, f_up, f_down,
                   sig_rough, n_cap, density, nmax=999):
        z = np.zeros(nmax + 1)
        cap = np.zeros(nmax + 1)
        ext = np.zeros(nmax + 1)
        rc = lib().orc_profile_new(ptype, length, rext_up, rext_down, rint_up, rint_down, f_up, f_down,
                                   nmax, _dp(z), _dp(cap), _dp(ext))
        if rc != 0:
            raise ValueError("unsupported profile type %r" % (ptype,))
        return cls(z, cap, ext, sig_rough, n_cap, density)

    def open_area(self):
        return lib().orc_open_area(C.byref(self.s))


def make_source(d_source, src_x, src_y, sigx, sigy, shiftx, shifty, hor_pol):
    return SourceS(d_source, src_x, src_y, sigx, sigy, shiftx, shifty, hor_pol)


class Photon:
    """Mutable photon for the per-function known-answer tests."""

    def __init__(self, start, direction, elecv, energies=(10.0,), amu=(0.0,), scatf=(0.0,), weights=None):
        self.energies = np.ascontiguousarray(energies, dtype=np.float64)
        self.amu = np.ascontiguousarray(amu, dtype=np.float64)
        self.scatf = np.ascontiguousarray(scatf, dtype=np.float64)
        self.weight = np.ones_like(self.energies) if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        s = PhotonS()
        s.start_coords = s.exit_coords = vec(start)
        s.start_direction = s.exit_direction = vec(direction)
        s.start_electric_vector = s.exit_electric_vector = vec(elecv)
        s.n_energies = self.energies.shape[0]
        s.energies, s.weight, s.amu, s.scatf = _dp(self.energies), _dp(self.weight), _dp(self.amu), _dp(self.scatf)
        s.i_refl = 0
        s.d_travel = 0.0
        self.s = s


def segment(cap0, cap1, rad0, rad1, phot0, phot1, pdir, last):
    pc = vec(last)
    sn = Vec3()
    rc = lib().orc_segment(vec(cap0), vec(cap1), rad0, rad1, vec(phot0), vec(phot1), vec(pdir), C.byref(pc), C.byref(sn))
    return rc, pc.tup(), sn.tup()


def refl_polar(e, density, scatf, amu, surface_norm, photon):
    ev = Vec3()
    r = lib().orc_refl_polar(e, density, scatf, amu, vec(surface_norm), C.byref(photon.s), C.byref(ev))
    return r, ev.tup()


def reflect(optic, photon, surface_norm):
    return lib().orc_reflect(C.byref(optic.s), C.byref(photon.s), vec(surface_norm))


def trace_wall(optic, photon):
    """polycap_capil_trace_wall on photon.s.exit_coords / exit_direction -> (rc, d_travel, r_cntr, q_cntr)"""
    d = C.c_double(0)
    r, q = C.c_int(0), C.c_int(0)
    rc = lib().orc_trace_wall(C.byref(optic.s), C.byref(photon.s), C.byref(d), C.byref(r), C.byref(q))
    return rc, d.value, r.value, q.value


def pc_intersect(optic, coord, direction):
    out = Vec3()
    ok = lib().orc_pc_intersect(C.byref(optic.s), vec(coord), vec(direction), C.byref(out))
    return out.tup() if ok else None


def _leaks_of(ptr, n):
    out = []
    for k in range(n):
        l = ptr[k]
        out.append(dict(coords=l.coords.tup(), direction=l.direction.tup(), elecv=l.elecv.tup(), n_refl=int(l.n_refl),
                        weights=np.ctypeslib.as_array(l.weight, shape=(l.n_energies,)).copy()))
    return out


def photon_leaks(photon):
    """(extleak, intleak) lists of a Photon whose s.leak_calc was set before reflect()/trace()"""
    return _leaks_of(photon.s.extleak, photon.s.n_extleak), _leaks_of(photon.s.intleak, photon.s.n_intleak)


def photon_clear_leaks(photon):
    lib().orc_photon_clear_leaks(C.byref(photon.s))


def _records(ptr, n, stride):
    a = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1) * stride,))[:int(n) * stride].reshape(int(n), stride).copy()
    lib().orc_free(C.cast(ptr, C.c_void_p))
    return a


LEAK_FIELDS = ("x", "y", "z", "dir_x", "dir_y", "dir_z", "elecv_x", "elecv_y", "elecv_z", "n_refl")


def launch_one_leak(optic, energies, amu, scatf, start, direction, elecv):
    """launch_one with leak_calc=true; ext / int = arrays [n, 10 + nE] (LEAK_FIELDS, then the weights)"""
    E = np.ascontiguousarray(energies, dtype=np.float64)
    A = np.ascontiguousarray(amu, dtype=np.float64)
    S = np.ascontiguousarray(scatf, dtype=np.float64)
    w = np.zeros_like(E)
    st, di, ev = (np.ascontiguousarray(v, dtype=np.float64) for v in (start, direction, elecv))
    ec, ed, ee = np.zeros(3), np.zeros(3), np.zeros(3)
    ir = C.c_int64(0)
    dt = C.c_double(0)
    pe, pi = c_double_p(), c_double_p()
    ne, ni = C.c_int64(0), C.c_int64(0)
    rc = lib().orc_launch_one_leak(C.byref(optic.s), E.shape[0], _dp(E), _dp(A), _dp(S), _dp(st), _dp(di), _dp(ev),
                                   _dp(w), _dp(ec), _dp(ed), _dp(ee), C.byref(ir), C.byref(dt),
                                   C.byref(pe), C.byref(ne), C.byref(pi), C.byref(ni))
    stride = 10 + E.shape[0]
    return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir.value, d_travel=dt.value,
                ext=_records(pe, ne.value, stride), int=_records(pi, ni.value, stride))


def trace(optic, ix, photon, cap_x, cap_y):
    cx = np.ascontiguousarray(cap_x, dtype=np.float64)
    cy = np.ascontiguousarray(cap_y, dtype=np.float64)
    ixc = C.c_int(ix)
    rc = lib().orc_trace(C.byref(optic.s), C.byref(ixc), C.byref(photon.s), _dp(cx), _dp(cy))
    return rc, ixc.value


def launch_one(optic, energies, amu, scatf, start, direction, elecv):
    E = np.ascontiguousarray(energies, dtype=np.float64)
    A = np.ascontiguousarray(amu, dtype=np.float64)
    S = np.ascontiguousarray(scatf, dtype=np.float64)
    w = np.zeros_like(E)
    st, di, ev = (np.ascontiguousarray(v, dtype=np.float64) for v in (start, direction, elecv))
    ec, ed, ee = np.zeros(3), np.zeros(3), np.zeros(3)
    ir = C.c_int64(0)
    dt = C.c_double(0)
    rc = lib().orc_launch_one(C.byref(optic.s), E.shape[0], _dp(E), _dp(A), _dp(S), _dp(st), _dp(di), _dp(ev),
                              _dp(w), _dp(ec), _dp(ed), _dp(ee), C.byref(ir), C.byref(dt))
    return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir.value, d_travel=dt.value)


def launch_batch(optic, energies, amu, scatf, start, direction, elecv, n_threads=0):
    E = np.ascontiguousarray(energies, dtype=np.float64)
    A = np.ascontiguousarray(amu, dtype=np.float64)
    S = np.ascontiguousarray(scatf, dtype=np.float64)
    st = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, 3)
    di = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    ev = np.ascontiguousarray(elecv, dtype=np.float64).reshape(-1, 3)
    n = st.shape[0]
    rc = np.zeros(n, dtype=np.int32)
    w = np.zeros((n, E.shape[0]))
    ec, ed, ee = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
    ir = np.zeros(n, dtype=np.int64)
    dt = np.zeros(n)
    lib().orc_launch_batch(C.byref(optic.s), E.shape[0], _dp(E), _dp(A), _dp(S), n, _dp(st), _dp(di), _dp(ev),
                           rc.ctypes.data_as(C.POINTER(C.c_int)), _dp(w), _dp(ec), _dp(ed), _dp(ee),
                           ir.ctypes.data_as(c_int64_p), _dp(dt), n_threads)
    return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir, d_travel=dt)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def uniform(seed, slot, attempt, d):
    return lib().orc_uniform(seed, slot, attempt, d)


def sample_photons(optic, source, seed, slots, attempt=0):
    """(n,12) array: start(3), dir(3), elecv(3), src_start(3) of attempt `attempt` of each slot."""
    slots = np.asarray(slots, dtype=np.int64)
    out = np.zeros((slots.shape[0], 12))
    buf = np.zeros(12)
    for i, s in enumerate(slots):
        lib().orc_sample_photon_flat(C.byref(optic.s), C.byref(source), seed, int(s), attempt, _dp(buf))
        out[i] = buf
    return out


IMG_FIELDS = ("src_start_x", "src_start_y", "pc_start_x", "pc_start_y", "pc_start_dir_x", "pc_start_dir_y",
              "pc_start_elecv_x", "pc_start_elecv_y", "pc_exit_x", "pc_exit_y", "pc_exit_z",
              "pc_exit_dir_x", "pc_exit_dir_y", "pc_exit_elecv_x", "pc_exit_elecv_y", "nrefl", "dtravel")


def transmission(optic, source, energies, amu, scatf, seed, slot0, n_slots, n_threads=0,
                 max_attempts=1 << 20, images=False, leak_calc=False):
    """leak_calc=True adds ext / int: arrays [n, 12 + nE] = slot, attempt, LEAK_FIELDS, weights"""
    E = np.ascontiguousarray(energies, dtype=np.float64)
    A = np.ascontiguousarray(amu, dtype=np.float64)
    S = np.ascontiguousarray(scatf, dtype=np.float64)
    sw = np.zeros_like(E)
    cnt = np.zeros(4, dtype=np.int64)
    img = np.zeros((n_slots, 17)) if images else None
    ew = np.zeros((n_slots, E.shape[0])) if images else None
    args = (C.byref(optic.s), C.byref(source), E.shape[0], _dp(E), _dp(A), _dp(S),
            seed, slot0, n_slots, n_threads, max_attempts, _dp(sw), cnt.ctypes.data_as(c_int64_p),
            _dp(img) if images else None, _dp(ew) if images else None)
    leaks = {}
    fixed = np.zeros((E.shape[0], 2), dtype=np.uint64)     # exact sum of floor(w * 2^62) per energy, (lo, hi); plain runs only
    if leak_calc:
        pe, pi = c_double_p(), c_double_p()
        ne, ni = C.c_int64(0), C.c_int64(0)
        rc = lib().orc_transmission_leak(*args, C.byref(pe), C.byref(ne), C.byref(pi), C.byref(ni))
        leaks = dict(ext=_records(pe, ne.value, 12 + E.shape[0]), int=_records(pi, ni.value, 12 + E.shape[0]))
    else:
        rc = lib().orc_transmission_fixed(*args, fixed.ctypes.data_as(C.POINTER(C.c_uint64)))
    eff = np.zeros_like(E)
    if cnt[0] + cnt[1] + cnt[2] > 0:
        lib().orc_efficiencies(E.shape[0], _dp(sw), cnt.ctypes.data_as(c_int64_p), _dp(eff))
    return dict(rc=rc, sum_weights=sw, sumw_fixed=fixed, counters=cnt, efficiencies=eff, images=img, exit_weights=ew,
                i_exit=int(cnt[0]), not_entered=int(cnt[1]), not_transmitted=int(cnt[2]), sum_irefl=int(cnt[3]),
                i_start=int(cnt[0] + cnt[1] + cnt[2]), **leaks)
