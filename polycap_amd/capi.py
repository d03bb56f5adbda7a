"""Python classes over libpolycap's reference-shaped C API (include/polycap.h).

Same class and method surface as the reference's Cython module (python/polycap.pyx): Profile, Rng, Description,
Photon, Source, TransmissionEfficiencies, VectorTuple; polycap_error codes map to the same Python exceptions
(python/polycap.pyx:91-107).  All tracing happens inside libpolycap.so on the GPU.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _cabi

VectorTuple = namedtuple("VectorTuple", ["x", "y", "z"])

__version__ = "1.2"


class _Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class _Leak(C.Structure):
    """polycap_leak (include/polycap.h)"""
    _fields_ = [("coords", _Vec3), ("direction", _Vec3), ("elecv", _Vec3), ("n_energies", C.c_size_t),
                ("weight", C.POINTER(C.c_double)), ("n_refl", C.c_int64)]


_Err = _cabi.ErrS


_ErrP = C.POINTER(_Err)
_dp = C.POINTER(C.c_double)

_SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
            "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
            "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu").split()
_Z = {s: i + 1 for i, s in enumerate(_SYMBOLS)}
# standard atomic weights (g/mol), Z = 1..94
_AW = [1.008, 4.0026, 6.94, 9.0122, 10.81, 12.011, 14.007, 15.9994, 18.998, 20.180, 22.990, 24.305, 26.982, 28.0855,
       30.974, 32.06, 35.45, 39.948, 39.098, 40.078, 44.956, 47.867, 50.942, 51.996, 54.938, 55.845, 58.933, 58.693,
       63.546, 65.38, 69.723, 72.630, 74.922, 78.971, 79.904, 83.798, 85.468, 87.62, 88.906, 91.224, 92.906, 95.95,
       98.0, 101.07, 102.91, 106.42, 107.87, 112.41, 114.82, 118.71, 121.76, 127.60, 126.90, 131.29, 132.91, 137.33,
       138.91, 140.12, 140.91, 144.24, 145.0, 150.36, 151.96, 157.25, 158.93, 162.50, 164.93, 167.26, 168.93, 173.05,
       174.97, 178.49, 180.95, 183.84, 186.21, 190.23, 192.22, 195.08, 196.97, 200.59, 204.38, 207.2, 208.98, 209.0,
       210.0, 222.0, 223.0, 226.0, 227.0, 232.04, 231.04, 238.03, 237.0, 244.0]


def _parse_formula(formula):
    """'SiO2' -> {Z: mass fraction in %}; flat formulas with optional integer/decimal counts and parentheses."""
    import re
    if not formula or not re.match(r"^[A-Z]", formula):
        raise ValueError("Invalid chemical formula: Found a lowercase character or digit where not allowed")

    def parse(s, pos):
        counts = {}
        while pos < len(s):
            ch = s[pos]
            if ch == "(":
                inner, pos = parse(s, pos + 1)
                m = re.match(r"\d+(\.\d+)?", s[pos:])
                mult = float(m.group(0)) if m else 1.0
                pos += len(m.group(0)) if m else 0
                for k, v in inner.items():
                    counts[k] = counts.get(k, 0.0) + v * mult
            elif ch == ")":
                return counts, pos + 1
            else:
                m = re.match(r"([A-Z][a-z]?)(\d+(\.\d+)?)?", s[pos:])
                if not m:
                    raise ValueError("Invalid chemical formula: Found a lowercase character or digit where not allowed")
                sym = m.group(1)
                if sym not in _Z:
                    raise ValueError("Invalid chemical symbol")
                counts[sym] = counts.get(sym, 0.0) + (float(m.group(2)) if m.group(2) else 1.0)
                pos += len(m.group(0))
        return counts, pos

    counts, _ = parse(formula, 0)
    masses = {_Z[s]: n * _AW[_Z[s] - 1] for s, n in counts.items()}
    tot = sum(masses.values())
    return {z: 100.0 * m / tot for z, m in masses.items()}


def _lib():
    L = _cabi.lib()
    if getattr(L, "_capi_ready", False):
        return L
    P = C.POINTER
    vp = C.c_void_p
    epp = P(_ErrP)
    sig = {
        "polycap_error_free": (None, [_ErrP]),
        "polycap_free": (None, [vp]),
        "polycap_profile_new": (vp, [C.c_int] + [C.c_double] * 7 + [epp]),
        "polycap_profile_new_from_file": (vp, [C.c_char_p] * 3 + [epp]),
        "polycap_profile_new_from_arrays": (vp, [C.c_int, _dp, _dp, _dp, epp]),
        "polycap_profile_validate": (C.c_int, [vp, C.c_int64, epp]),
        "polycap_profile_get_ext": (C.c_bool, [vp, P(C.c_size_t), P(_dp), epp]),
        "polycap_profile_get_cap": (C.c_bool, [vp, P(C.c_size_t), P(_dp), epp]),
        "polycap_profile_get_z": (C.c_bool, [vp, P(C.c_size_t), P(_dp), epp]),
        "polycap_profile_free": (None, [vp]),
        "polycap_description_new": (vp, [vp, C.c_double, C.c_int64, C.c_uint, P(C.c_int), _dp, C.c_double, epp]),
        "polycap_description_get_profile": (vp, [vp]),
        "polycap_description_free": (None, [vp]),
        "polycap_rng_new": (vp, []),
        "polycap_rng_new_with_seed": (vp, [C.c_ulong]),
        "polycap_rng_free": (None, [vp]),
        "polycap_photon_new": (vp, [vp, _Vec3, _Vec3, _Vec3, epp]),
        "polycap_photon_launch": (C.c_int, [vp, C.c_size_t, _dp, P(_dp), C.c_bool, epp]),
        "polycap_photon_free": (None, [vp]),
        "polycap_photon_get_extleak_data": (C.c_bool, [vp, P(P(P(_Leak))), P(C.c_int64), epp]),
        "polycap_photon_get_intleak_data": (C.c_bool, [vp, P(P(P(_Leak))), P(C.c_int64), epp]),
        "polycap_transmission_efficiencies_get_extleak_data": (C.c_bool, [vp, P(P(P(_Leak))), P(C.c_int64), epp]),
        "polycap_transmission_efficiencies_get_intleak_data": (C.c_bool, [vp, P(P(P(_Leak))), P(C.c_int64), epp]),
        "polycap_leak_free": (None, [P(_Leak)]),
        "polycap_photon_get_dtravel": (C.c_double, [vp]),
        "polycap_photon_get_irefl": (C.c_int64, [vp]),
        "polycap_source_new": (vp, [vp] + [C.c_double] * 8 + [C.c_size_t, _dp, epp]),
        "polycap_source_new_from_file": (vp, [C.c_char_p, epp]),
        "polycap_source_free": (None, [vp]),
        "polycap_source_get_photon": (vp, [vp, vp, epp]),
        "polycap_source_get_description": (vp, [vp]),
        "polycap_source_get_transmission_efficiencies": (vp, [vp, C.c_int, C.c_int, C.c_bool, vp, epp]),
        "polycap_transmission_efficiencies_free": (None, [vp]),
        "polycap_transmission_efficiencies_get_data": (C.c_bool, [vp, P(C.c_size_t), P(_dp), P(_dp), epp]),
        "polycap_transmission_efficiencies_get_start_data": (C.c_bool, [vp, P(C.c_int64), P(C.c_int64)] + [P(P(_Vec3))] * 4 + [epp]),
        "polycap_transmission_efficiencies_get_exit_data": (C.c_bool, [vp, P(C.c_int64)] + [P(P(_Vec3))] * 3 +
                                                             [P(P(C.c_int64)), P(_dp), P(C.c_size_t), P(P(_dp)), epp]),
        "polycap_transmission_efficiencies_write_hdf5": (C.c_bool, [vp, C.c_char_p, epp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    for g in ("start_coords", "start_direction", "start_electric_vector", "exit_coords", "exit_direction", "exit_electric_vector"):
        f = getattr(L, "polycap_photon_get_" + g)
        f.restype = _Vec3
        f.argtypes = [vp]
    L._capi_ready = True
    return L


_EXC = {0: MemoryError, 1: ValueError, 2: IOError, 3: IOError, 4: TypeError, 5: NotImplementedError, 6: RuntimeError}


def _check(err):
    """Raise the Python exception the reference's binding raises for this polycap_error (python/polycap.pyx:91-107)."""
    if err:
        e = err.contents
        msg = e.message.decode() if e.message else ""
        exc = _EXC.get(e.code, RuntimeError)
        _lib().polycap_error_free(err)
        raise exc(msg)


def _v3(t):
    if t is None or len(t) != 3:
        raise ValueError("vectors must have three components")
    return _Vec3(float(t[0]), float(t[1]), float(t[2]))


def _take(ptr, n, dtype=np.float64):
    arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    _lib().polycap_free(C.cast(ptr, C.c_void_p))
    return arr


class Profile:
    CONICAL, PARABOLOIDAL, ELLIPSOIDAL = 0, 1, 2

    def __init__(self, type, length, rad_ext_upstream, rad_ext_downstream, rad_int_upstream, rad_int_downstream,
                 focal_dist_upstream, focal_dist_downstream, _handle=None):
        L = _lib()
        if _handle is not None:
            self._h = _handle
            return
        err = _ErrP()
        self._h = L.polycap_profile_new(int(type), float(length), float(rad_ext_upstream), float(rad_ext_downstream),
                                        float(rad_int_upstream), float(rad_int_downstream), float(focal_dist_upstream),
                                        float(focal_dist_downstream), C.byref(err))
        _check(err)

    @classmethod
    def new_from_arrays(cls, ext, cap, z):
        L = _lib()
        ext, cap, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (ext, cap, z))
        if not (ext.ndim == cap.ndim == z.ndim == 1 and ext.shape == cap.shape == z.shape):
            raise ValueError("ext, cap and z must be 1-D arrays of identical length")
        err = _ErrP()
        h = L.polycap_profile_new_from_arrays(ext.shape[0] - 1, ext.ctypes.data_as(_dp), cap.ctypes.data_as(_dp),
                                              z.ctypes.data_as(_dp), C.byref(err))
        _check(err)
        return cls(None, *([None] * 7), _handle=h)

    @classmethod
    def new_from_file(cls, single_cap_profile_file, central_axis_file, external_shape_file):
        err = _ErrP()
        h = _lib().polycap_profile_new_from_file(str(single_cap_profile_file).encode(), str(central_axis_file).encode(),
                                                 str(external_shape_file).encode(), C.byref(err))
        _check(err)
        return cls(None, *([None] * 7), _handle=h)

    def _get(self, fn):
        n = C.c_size_t(0)
        p = _dp()
        err = _ErrP()
        ok = fn(self._h, C.byref(n), C.byref(p), C.byref(err))
        _check(err)
        if not ok:
            raise RuntimeError("profile getter failed")
        return _take(p, n.value + 1)

    def get_ext(self):
        return self._get(_lib().polycap_profile_get_ext)

    def get_cap(self):
        return self._get(_lib().polycap_profile_get_cap)

    def get_z(self):
        return self._get(_lib().polycap_profile_get_z)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().polycap_profile_free(self._h)
            self._h = None


class Rng:
    def __init__(self, seed=None):
        L = _lib()
        if seed is None:
            self._h = L.polycap_rng_new()
        else:
            if not isinstance(seed, (int, np.integer)):
                raise TypeError("seed must be an integer")
            if seed < 0:
                raise OverflowError("can't convert negative value to unsigned long")
            self._h = L.polycap_rng_new_with_seed(int(seed))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().polycap_rng_free(self._h)
            self._h = None


class Description:
    def __init__(self, profile, sig_rough, n_cap, composition, density, _handle=None, _owner=None):
        L = _lib()
        self._owner = _owner
        if _handle is not None:
            self._h = _handle
            return
        if profile is None or not isinstance(profile, Profile):
            raise ValueError("profile must be a Profile")
        if isinstance(composition, str):
            comp = _parse_formula(composition)
        elif isinstance(composition, dict):
            if len(composition) == 0:
                raise ValueError("composition cannot be empty")
            comp = {}
            for k, v in composition.items():
                if k not in _Z:
                    raise ValueError("Invalid chemical symbol")
                comp[_Z[k]] = float(v)
        else:
            raise TypeError("composition must be a dictionary or a string")
        iz = (C.c_int * len(comp))(*comp.keys())
        wi = (C.c_double * len(comp))(*comp.values())
        err = _ErrP()
        self._h = L.polycap_description_new(profile._h, float(sig_rough), int(n_cap), len(comp), iz, wi, float(density), C.byref(err))
        _check(err)

    def __del__(self):
        if getattr(self, "_h", None) and self._owner is None:
            _lib().polycap_description_free(self._h)
        self._h = None


class Leak:
    """One leak event: where the transmitted fraction of a reflection left the optic (extleak) or reached the exit
    plane inside the glass (intleak), with the per-energy weights it carried."""

    def __init__(self, coords, direction, elecv, weight, n_refl):
        self._coords, self._direction, self._elecv, self._weight, self._n_refl = coords, direction, elecv, weight, n_refl

    coords = property(lambda self: self._coords)
    direction = property(lambda self: self._direction)
    elecv = property(lambda self: self._elecv)
    weight = property(lambda self: self._weight)
    n_refl = property(lambda self: self._n_refl)


def _leak_list(getter, handle):
    """Calls one of the *_get_extleak_data / *_get_intleak_data functions; the C arrays are converted and freed.
    No events -> the C function's error (ValueError), as with the reference's binding."""
    L = _lib()
    arr = C.POINTER(C.POINTER(_Leak))()
    n = C.c_int64(0)
    err = _ErrP()
    getattr(L, getter)(handle, C.byref(arr), C.byref(n), C.byref(err))
    out = []
    for k in range(n.value if arr else 0):
        l = arr[k].contents
        out.append(Leak(VectorTuple(l.coords.x, l.coords.y, l.coords.z), VectorTuple(l.direction.x, l.direction.y, l.direction.z),
                        VectorTuple(l.elecv.x, l.elecv.y, l.elecv.z),
                        np.ctypeslib.as_array(l.weight, shape=(l.n_energies,)).copy(), int(l.n_refl)))
        L.polycap_leak_free(arr[k])
    if arr:
        L.polycap_free(C.cast(arr, C.c_void_p))
    _check(err)
    return out


class _LeakData:
    """extleak_data / intleak_data generator properties shared by Photon and TransmissionEfficiencies (cached lists)."""
    _leak_getters = (None, None)

    def _leaks(self, kind):
        cache = self.__dict__.setdefault("_leak_cache", {})
        if kind not in cache:
            cache[kind] = _leak_list(self._leak_getters[kind], self._h)
        return cache[kind]

    @property
    def extleak_data(self):
        return (l for l in self._leaks(0))

    @property
    def intleak_data(self):
        return (l for l in self._leaks(1))


class Photon(_LeakData):
    _leak_getters = ("polycap_photon_get_extleak_data", "polycap_photon_get_intleak_data")

    def __init__(self, description, start_coords, start_direction, start_electric_vector, _handle=None):
        L = _lib()
        self._description = description
        if _handle is not None:
            self._h = _handle
            return
        if description is None or not isinstance(description, Description):
            raise ValueError("description must be a Description")
        err = _ErrP()
        self._h = L.polycap_photon_new(description._h, _v3(start_coords), _v3(start_direction), _v3(start_electric_vector), C.byref(err))
        _check(err)

    def launch(self, energies, leak_calc=False):
        """ndarray of weights, or None when the photon hit the glass at the entrance (return code 2)."""
        L = _lib()
        E = np.atleast_1d(np.ascontiguousarray(energies, dtype=np.float64))
        w = _dp()
        err = _ErrP()
        self.__dict__.pop("_leak_cache", None)
        rc = L.polycap_photon_launch(self._h, E.shape[0], E.ctypes.data_as(_dp), C.byref(w), bool(leak_calc), C.byref(err))
        weights = _take(w, E.shape[0]) if w else None
        _check(err)
        self.return_code = rc
        if rc == 2 or rc == -1:
            return None
        return weights

    def _vec(self, name):
        v = getattr(_lib(), "polycap_photon_get_" + name)(self._h)
        return VectorTuple(v.x, v.y, v.z)

    start_coords = property(lambda self: self._vec("start_coords"))
    start_direction = property(lambda self: self._vec("start_direction"))
    start_electric_vector = property(lambda self: self._vec("start_electric_vector"))
    exit_coords = property(lambda self: self._vec("exit_coords"))
    exit_direction = property(lambda self: self._vec("exit_direction"))
    exit_electric_vector = property(lambda self: self._vec("exit_electric_vector"))
    d_travel = property(lambda self: _lib().polycap_photon_get_dtravel(self._h))
    i_refl = property(lambda self: _lib().polycap_photon_get_irefl(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().polycap_photon_free(self._h)
            self._h = None


class TransmissionEfficiencies(_LeakData):
    _leak_getters = ("polycap_transmission_efficiencies_get_extleak_data", "polycap_transmission_efficiencies_get_intleak_data")

    def __init__(self, handle, source):
        self._h = handle
        self._source = source   # the C object borrows the source (reference src/polycap-source.c:682)
        self._data = None

    @property
    def data(self):
        """(energies, efficiencies) as read-only arrays; the same tuple object on every access."""
        if self._data is None:
            n = C.c_size_t(0)
            e, f = _dp(), _dp()
            err = _ErrP()
            _lib().polycap_transmission_efficiencies_get_data(self._h, C.byref(n), C.byref(e), C.byref(f), C.byref(err))
            _check(err)
            E, F = _take(e, n.value), _take(f, n.value)
            E.flags.writeable = False
            F.flags.writeable = False
            self._data = (E, F)
        return self._data

    def _start(self):
        L = _lib()
        ns, nx = C.c_int64(0), C.c_int64(0)
        ptrs = [C.POINTER(_Vec3)() for _ in range(4)]
        err = _ErrP()
        L.polycap_transmission_efficiencies_get_start_data(self._h, C.byref(ns), C.byref(nx), *[C.byref(p) for p in ptrs], C.byref(err))
        _check(err)
        out = []
        for p in ptrs:
            a = np.ctypeslib.as_array(C.cast(p, _dp), shape=(nx.value * 3,)).reshape(-1, 3).copy()
            L.polycap_free(C.cast(p, C.c_void_p))
            out.append(a)
        return ns.value, nx.value, out

    def _exit(self):
        L = _lib()
        nx = C.c_int64(0)
        vp = [C.POINTER(_Vec3)() for _ in range(3)]
        nrefl = C.POINTER(C.c_int64)()
        dtr = _dp()
        ne = C.c_size_t(0)
        ww = C.POINTER(_dp)()
        err = _ErrP()
        L.polycap_transmission_efficiencies_get_exit_data(self._h, C.byref(nx), *[C.byref(p) for p in vp], C.byref(nrefl),
                                                         C.byref(dtr), C.byref(ne), C.byref(ww), C.byref(err))
        _check(err)
        n = nx.value
        vecs = []
        for p in vp:
            vecs.append(np.ctypeslib.as_array(C.cast(p, _dp), shape=(n * 3,)).reshape(-1, 3).copy())
            L.polycap_free(C.cast(p, C.c_void_p))
        nr = np.ctypeslib.as_array(nrefl, shape=(n,)).copy()
        L.polycap_free(C.cast(nrefl, C.c_void_p))
        dt = _take(dtr, n)
        W = np.zeros((n, ne.value))
        for i in range(n):
            W[i] = np.ctypeslib.as_array(ww[i], shape=(ne.value,))
            L.polycap_free(C.cast(ww[i], C.c_void_p))
        L.polycap_free(C.cast(ww, C.c_void_p))
        return n, vecs, nr, dt, W

    @property
    def start_coords(self):
        return (VectorTuple(*r) for r in self._start()[2][0])

    @property
    def start_direction(self):
        return (VectorTuple(*r) for r in self._start()[2][1])

    @property
    def start_electric_vector(self):
        return (VectorTuple(*r) for r in self._start()[2][2])

    @property
    def src_start_coords(self):
        return (VectorTuple(*r) for r in self._start()[2][3])

    @property
    def exit_coords(self):
        return (VectorTuple(*r) for r in self._exit()[1][0])

    @property
    def exit_direction(self):
        return (VectorTuple(*r) for r in self._exit()[1][1])

    @property
    def exit_electric_vector(self):
        return (VectorTuple(*r) for r in self._exit()[1][2])

    @property
    def n_refl(self):
        return self._exit()[2]

    @property
    def d_travel(self):
        return self._exit()[3]

    @property
    def exit_weights(self):
        return self._exit()[4]

    @classmethod
    def from_totals(cls, source, sum_weights, counters, images=None, exit_weights=None):
        """Result object from totals (and optionally the [n_exit, 17] image array + [n_exit, nE] weights of
        TraceContext.images()) produced elsewhere, e.g. by polycap_amd.distributed.run_sharded over several GPUs:
        the getters and write_hdf5 then serve the sharded run as they do a single-device one (no reference counterpart)."""
        from . import _cabi
        sw = np.ascontiguousarray(sum_weights, dtype=np.float64)
        cnt = np.zeros(6, dtype=np.int64)
        cc = np.asarray(counters, dtype=np.int64).ravel()
        cnt[:min(6, cc.size)] = cc[:6]
        n = int(cnt[0])
        sp = None
        if images is not None:
            img = np.asarray(images, dtype=np.float64)
            if img.shape != (n, 17):
                raise ValueError("images must have shape (counters[0], 17)")
            planes = np.ascontiguousarray(img.T)
            nrefl = np.ascontiguousarray(np.rint(planes[15]).astype(np.int64))
            w = np.zeros((n, sw.size)) if exit_weights is None else np.ascontiguousarray(exit_weights, dtype=np.float64)
            if w.shape != (n, sw.size):
                raise ValueError("exit_weights must have shape (counters[0], n_energies)")
            sp = C.byref(_cabi.images_struct(planes, nrefl, w))
        err = _ErrP()
        L = _lib()
        h = L.pc_transmission_efficiencies_from_totals(source._h, n, sw.ctypes.data_as(_dp), cnt.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       sp, C.byref(err))
        _check(err)
        return cls(h, source)

    def write_hdf5(self, filename):
        err = _ErrP()
        _lib().polycap_transmission_efficiencies_write_hdf5(self._h, None if filename is None else str(filename).encode(), C.byref(err))
        _check(err)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().polycap_transmission_efficiencies_free(self._h)
            self._h = None


class Source:
    def __init__(self, description, d_source, src_x, src_y, src_sigx, src_sigy, src_shiftx, src_shifty, hor_pol, energies,
                 _handle=None):
        L = _lib()
        if _handle is not None:
            self._h = _handle
            return
        if description is None or not isinstance(description, Description):
            raise ValueError("description must be a Description")
        E = np.atleast_1d(np.ascontiguousarray(energies, dtype=np.float64))
        err = _ErrP()
        self._h = L.polycap_source_new(description._h, float(d_source), float(src_x), float(src_y), float(src_sigx), float(src_sigy),
                                       float(src_shiftx), float(src_shifty), float(hor_pol), E.shape[0], E.ctypes.data_as(_dp), C.byref(err))
        _check(err)

    @classmethod
    def new_from_file(cls, filename):
        err = _ErrP()
        h = _lib().polycap_source_new_from_file(None if filename is None else str(filename).encode(), C.byref(err))
        _check(err)
        return cls(None, *([None] * 9), _handle=h)

    def get_photon(self, rng):
        if rng is None or not isinstance(rng, Rng):
            raise ValueError("rng must be an Rng")
        err = _ErrP()
        h = _lib().polycap_source_get_photon(self._h, rng._h, C.byref(err))
        _check(err)
        d = Description(None, 0, 0, None, 0, _handle=_lib().polycap_source_get_description(self._h), _owner=self)
        return Photon(d, None, None, None, _handle=h)

    def get_transmission_efficiencies(self, max_threads, n_photons, leak_calc=False):
        err = _ErrP()
        h = _lib().polycap_source_get_transmission_efficiencies(self._h, int(max_threads), int(n_photons), bool(leak_calc), None, C.byref(err))
        _check(err)
        return TransmissionEfficiencies(h, self)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().polycap_source_free(self._h)
            self._h = None
