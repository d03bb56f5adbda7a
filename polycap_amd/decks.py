"""Build pc_hip problems from the reference's positional .inp decks, using libpolycap's own parser
(polycap_source_new_from_file) and optical-constants provider."""
import ctypes as C

import numpy as np

from . import _cabi
from ._cabi import Problem, ProblemS, c_double_p
from ._cabi import ErrS as _ErrS


# polycap_error codes -> Python exceptions as in the reference's binding (python/polycap.pyx:91-107)
_EXC = {0: MemoryError, 1: ValueError, 2: IOError, 3: IOError, 4: TypeError, 5: NotImplementedError, 6: RuntimeError}


def _raise(L, err, where):
    msg = "%s failed" % where
    exc = ValueError
    if err:
        e = err.contents
        msg = "%s: [%d] %s" % (where, e.code, e.message.decode() if e.message else "")
        exc = _EXC.get(e.code, RuntimeError)
        L.polycap_error_free(err)
    raise exc(msg)


def _protos(L):
    if getattr(L, "_decks_ready", False):
        return
    L.polycap_source_new_from_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_ErrS))]
    L.polycap_source_new_from_file.restype = C.c_void_p
    L.polycap_source_free.argtypes = [C.c_void_p]
    L.polycap_source_free.restype = None
    L.polycap_error_free.argtypes = [C.POINTER(_ErrS)]
    L.polycap_error_free.restype = None
    L.pc_source_problem.argtypes = [C.c_void_p, C.POINTER(ProblemS), c_double_p, c_double_p, C.POINTER(C.c_int),
                                    C.POINTER(C.POINTER(_ErrS))]
    L.pc_source_problem.restype = C.c_int
    L.pc_optconst_scatf.argtypes = [C.c_uint, C.POINTER(C.c_int), c_double_p, C.c_double, C.c_size_t, c_double_p,
                                    c_double_p, c_double_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(_ErrS))]
    L.pc_optconst_scatf.restype = C.c_int
    L.pc_optconst_provider.restype = C.c_char_p
    L.pc_optconst_library.restype = C.c_char_p
    L._decks_ready = True


def optical_constants_provider():
    """"xraylib" when a libxrl could be bound (and POLYCAP_OPTCONST is not "builtin"), else a description of the built-in tables."""
    L = _cabi.lib()
    _protos(L)
    p = L.pc_optconst_provider().decode()
    return "xraylib" if p == "xraylib" else p


def optical_constants(iz, wi, density, energies):
    """(amu, scatf, synthetic) for a composition: what polycap_photon_scatf computes in the reference."""
    L = _cabi.lib()
    _protos(L)
    iz = np.ascontiguousarray(iz, dtype=np.int32)
    wi = np.ascontiguousarray(wi, dtype=np.float64)
    if wi.sum() > 1.0:
        wi = wi / 100.0
    E = np.ascontiguousarray(energies, dtype=np.float64).ravel()
    amu, scatf = np.zeros_like(E), np.zeros_like(E)
    syn = C.c_int(0)
    err = C.POINTER(_ErrS)()
    rc = L.pc_optconst_scatf(iz.shape[0], iz.ctypes.data_as(C.POINTER(C.c_int)), _cabi.dptr(wi), float(density),
                             E.shape[0], _cabi.dptr(E), _cabi.dptr(amu), _cabi.dptr(scatf), C.byref(syn), C.byref(err))
    if rc != 0:
        _raise(L, err, "pc_optconst_scatf")
    return amu, scatf, bool(syn.value)


def problem_from_inp(path, energies=None, sig_rough=None):
    """Problem for a .inp deck; `energies` overrides the deck's grid (e.g. [10.0]), `sig_rough` its roughness."""
    L = _cabi.lib()
    _protos(L)
    err = C.POINTER(_ErrS)()
    src = L.polycap_source_new_from_file(str(path).encode(), C.byref(err))
    if not src:
        _raise(L, err, "polycap_source_new_from_file(%s)" % path)
    try:
        ps = ProblemS()
        rc = L.pc_source_problem(src, C.byref(ps), None, None, None, C.byref(err))
        if rc != 0:
            _raise(L, err, "pc_source_problem")
        n = ps.nmax + 1
        z = np.ctypeslib.as_array(ps.z, shape=(n,)).copy()
        cap = np.ctypeslib.as_array(ps.cap, shape=(n,)).copy()
        ext = np.ctypeslib.as_array(ps.ext, shape=(n,)).copy()
        E = np.ctypeslib.as_array(ps.energies, shape=(ps.n_energies,)).copy() if energies is None \
            else np.ascontiguousarray(energies, dtype=np.float64).ravel()
        # composition is not exposed by the plain problem view: the decks of the reference all use the O/Si glass,
        # read it back from the deck itself (line 7.. : nelem, then Z w% pairs)
        iz, wi, density = _deck_composition(path)
        amu, scatf, syn = optical_constants(iz, wi, density, E)
        prob = Problem(z, cap, ext, ps.sig_rough if sig_rough is None else sig_rough, ps.n_cap, ps.density, E, amu, scatf,
                       ps.d_source, ps.src_x, ps.src_y, ps.src_sigx, ps.src_sigy, ps.src_shiftx, ps.src_shifty, ps.hor_pol)
        prob.synthetic_constants = syn
        prob.composition = (iz, wi, density)
        return prob
    finally:
        L.polycap_source_free(src)


def _deck_composition(path):
    toks = open(path).read().split()
    # sig_rough, d_source, src_x, src_y, sigx, sigy, shiftx, shifty, hor_pol, nelem, (Z w)*nelem, density
    nelem = int(float(toks[9]))
    iz = [int(float(toks[10 + 2 * k])) for k in range(nelem)]
    wi = [float(toks[11 + 2 * k]) for k in range(nelem)]
    density = float(toks[10 + 2 * nelem])
    return iz, wi, density
