# cython: language_level=3, boundscheck=False, wraparound=False
"""polycap -- Cython binding of libpolycap (MI355X build), same surface as the reference's python/polycap.pyx:
Profile, Rng, Description, Photon, Source, TransmissionEfficiencies, VectorTuple, __version__; polycap_error codes map
to the same Python exceptions.  Every call goes straight to the C API of include/polycap.h; tracing happens on the GPU.
Composition strings / element symbols are resolved by polycap_amd.capi's small parser (the reference calls xraylib's
CompoundParser for this; it is not on the trace path)."""
from collections import namedtuple
from libc.stdint cimport int64_t
from libc.stddef cimport size_t
from libcpp cimport bool as cbool
import numpy as np

from polycap_amd.capi import _Z as _SYMBOL_TO_Z, _parse_formula

cdef extern from "polycap.h" nogil:
    int POLYCAP_VERSION_MAJOR
    int POLYCAP_VERSION_MINOR
    cdef enum polycap_error_code:
        POLYCAP_ERROR_MEMORY
        POLYCAP_ERROR_INVALID_ARGUMENT
        POLYCAP_ERROR_IO
        POLYCAP_ERROR_OPENMP
        POLYCAP_ERROR_TYPE
        POLYCAP_ERROR_UNSUPPORTED
        POLYCAP_ERROR_RUNTIME
    ctypedef struct polycap_error:
        polycap_error_code code
        char *message
    void polycap_error_free(polycap_error *error)
    void polycap_free(void *data)

    ctypedef enum polycap_profile_type:
        POLYCAP_PROFILE_CONICAL
        POLYCAP_PROFILE_PARABOLOIDAL
        POLYCAP_PROFILE_ELLIPSOIDAL
    ctypedef struct polycap_profile
    polycap_profile *polycap_profile_new(polycap_profile_type type, double length, double rad_ext_upstream, double rad_ext_downstream,
        double rad_int_upstream, double rad_int_downstream, double focal_dist_upstream, double focal_dist_downstream, polycap_error **error)
    polycap_profile *polycap_profile_new_from_file(const char *a, const char *b, const char *c, polycap_error **error)
    polycap_profile *polycap_profile_new_from_arrays(int nid, double *ext, double *cap, double *z, polycap_error **error)
    cbool polycap_profile_get_ext(polycap_profile *profile, size_t *nid, double **ext, polycap_error **error)
    cbool polycap_profile_get_cap(polycap_profile *profile, size_t *nid, double **cap, polycap_error **error)
    cbool polycap_profile_get_z(polycap_profile *profile, size_t *nid, double **z, polycap_error **error)
    void polycap_profile_free(polycap_profile *profile)

    ctypedef struct polycap_description
    polycap_description *polycap_description_new(polycap_profile *profile, double sig_rough, int64_t n_cap, unsigned int nelem,
        int *iz, double *wi, double density, polycap_error **error)
    void polycap_description_free(polycap_description *description)

    ctypedef struct polycap_vector3:
        double x
        double y
        double z
    ctypedef struct polycap_leak:
        polycap_vector3 coords
        polycap_vector3 direction
        polycap_vector3 elecv
        size_t n_energies
        double *weight
        int64_t n_refl
    void polycap_leak_free(polycap_leak *leak)
    ctypedef struct polycap_photon
    cbool polycap_photon_get_extleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
    cbool polycap_photon_get_intleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
    polycap_photon *polycap_photon_new(polycap_description *description, polycap_vector3 start_coords, polycap_vector3 start_direction,
        polycap_vector3 start_electric_vector, polycap_error **error)
    int polycap_photon_launch(polycap_photon *photon, size_t n_energies, double *energies, double **weights, cbool leak_calc, polycap_error **error)
    polycap_vector3 polycap_photon_get_start_coords(polycap_photon *photon)
    polycap_vector3 polycap_photon_get_start_direction(polycap_photon *photon)
    polycap_vector3 polycap_photon_get_start_electric_vector(polycap_photon *photon)
    polycap_vector3 polycap_photon_get_exit_coords(polycap_photon *photon)
    polycap_vector3 polycap_photon_get_exit_direction(polycap_photon *photon)
    polycap_vector3 polycap_photon_get_exit_electric_vector(polycap_photon *photon)
    double polycap_photon_get_dtravel(polycap_photon *photon)
    int64_t polycap_photon_get_irefl(polycap_photon *photon)
    void polycap_photon_free(polycap_photon *photon)

    ctypedef struct polycap_rng
    polycap_rng *polycap_rng_new()
    polycap_rng *polycap_rng_new_with_seed(unsigned long seed)
    void polycap_rng_free(polycap_rng *rng)

    ctypedef struct polycap_transmission_efficiencies
    void polycap_transmission_efficiencies_free(polycap_transmission_efficiencies *efficiencies)
    cbool polycap_transmission_efficiencies_get_extleak_data(polycap_transmission_efficiencies *efficiencies, polycap_leak ***leaks,
        int64_t *n_leaks, polycap_error **error)
    cbool polycap_transmission_efficiencies_get_intleak_data(polycap_transmission_efficiencies *efficiencies, polycap_leak ***leaks,
        int64_t *n_leaks, polycap_error **error)
    cbool polycap_transmission_efficiencies_write_hdf5(polycap_transmission_efficiencies *efficiencies, const char *filename, polycap_error **error)
    cbool polycap_transmission_efficiencies_get_data(polycap_transmission_efficiencies *efficiencies, size_t *n_energies,
        double **energies_arr, double **efficiencies_arr, polycap_error **error)
    cbool polycap_transmission_efficiencies_get_start_data(polycap_transmission_efficiencies *efficiencies, int64_t *n_start, int64_t *n_exit,
        polycap_vector3 **start_coords, polycap_vector3 **start_direction, polycap_vector3 **start_elecv, polycap_vector3 **src_start_coords,
        polycap_error **error)
    cbool polycap_transmission_efficiencies_get_exit_data(polycap_transmission_efficiencies *efficiencies, int64_t *n_exit,
        polycap_vector3 **exit_coords, polycap_vector3 **exit_direction, polycap_vector3 **exit_elecv, int64_t **n_refl, double **d_travel,
        size_t *n_energies, double ***exit_weights, polycap_error **error)

    ctypedef struct pc_hip_images:
        double *src_start_coords[2]
        double *pc_start_coords[2]
        double *pc_start_dir[2]
        double *pc_start_elecv[2]
        double *pc_exit_coords[3]
        double *pc_exit_dir[2]
        double *pc_exit_elecv[2]
        int64_t *pc_exit_nrefl
        double *pc_exit_dtravel
        double *exit_coord_weights
    void *pc_transmission_efficiencies_from_totals(void *source, int64_t n_exit, const double *sum_weights, const int64_t *counters,
        const pc_hip_images *planes, void *error)

    ctypedef struct polycap_progress_monitor
    ctypedef struct polycap_source
    polycap_source *polycap_source_new(polycap_description *description, double d_source, double src_x, double src_y, double src_sigx,
        double src_sigy, double src_shiftx, double src_shifty, double hor_pol, size_t n_energies, double *energies, polycap_error **error)
    polycap_source *polycap_source_new_from_file(const char *filename, polycap_error **error)
    void polycap_source_free(polycap_source *source)
    polycap_photon *polycap_source_get_photon(polycap_source *source, polycap_rng *rng, polycap_error **error)
    polycap_transmission_efficiencies *polycap_source_get_transmission_efficiencies(polycap_source *source, int max_threads, int n_photons,
        cbool leak_calc, polycap_progress_monitor *progress_monitor, polycap_error **error)
    const polycap_description *polycap_source_get_description(polycap_source *source)

__version__ = "%d.%d" % (POLYCAP_VERSION_MAJOR, POLYCAP_VERSION_MINOR)

VectorTuple = namedtuple("VectorTuple", ["x", "y", "z"])

_error_map = {
    POLYCAP_ERROR_MEMORY: MemoryError,
    POLYCAP_ERROR_INVALID_ARGUMENT: ValueError,
    POLYCAP_ERROR_IO: IOError,
    POLYCAP_ERROR_OPENMP: IOError,
    POLYCAP_ERROR_TYPE: TypeError,
    POLYCAP_ERROR_UNSUPPORTED: NotImplementedError,
    POLYCAP_ERROR_RUNTIME: RuntimeError,
}


cdef int _raise_if(polycap_error *error) except -1:
    if error == NULL:
        return 0
    exc = _error_map.get(error.code, RuntimeError)
    msg = error.message.decode("utf-8", "replace") if error.message != NULL else ""
    polycap_error_free(error)
    raise exc(msg)


cdef polycap_vector3 _vec(object t) except *:
    cdef polycap_vector3 v
    if t is None or len(t) != 3:
        raise ValueError("vectors must have three components")
    v.x = t[0]; v.y = t[1]; v.z = t[2]
    return v


cdef object _tuple(polycap_vector3 v):
    return VectorTuple(v.x, v.y, v.z)


cdef object _take_doubles(double *p, size_t n):
    arr = np.empty(n, dtype=np.float64)
    cdef double[::1] view = arr
    cdef size_t i
    for i in range(n):
        view[i] = p[i]
    polycap_free(p)
    return arr


cdef object _take_vectors(polycap_vector3 *p, size_t n):
    arr = np.empty((n, 3), dtype=np.float64)
    cdef double[:, ::1] view = arr
    cdef size_t i
    for i in range(n):
        view[i, 0] = p[i].x; view[i, 1] = p[i].y; view[i, 2] = p[i].z
    polycap_free(p)
    return arr


cdef class Leak:
    """One leak event: where the transmitted fraction of a reflection left the optic (extleak) or reached the exit plane
    inside the glass (intleak), with the per-energy weights it carried."""
    cdef object _coords, _direction, _elecv, _weight
    cdef int64_t _n_refl

    @property
    def coords(self):
        return self._coords

    @property
    def direction(self):
        return self._direction

    @property
    def elecv(self):
        return self._elecv

    @property
    def weight(self):
        return self._weight

    @property
    def n_refl(self):
        return self._n_refl


cdef object _take_leaks(polycap_leak **leaks, int64_t n):
    """polycap_leak array of a getter -> list of Leak; the C structs are freed"""
    out = []
    cdef int64_t i
    cdef size_t e
    cdef Leak l
    cdef double[::1] view
    if leaks == NULL:
        return out
    for i in range(n):
        l = Leak.__new__(Leak)
        l._coords = _tuple(leaks[i].coords)
        l._direction = _tuple(leaks[i].direction)
        l._elecv = _tuple(leaks[i].elecv)
        w = np.empty(leaks[i].n_energies, dtype=np.float64)
        view = w
        for e in range(leaks[i].n_energies):
            view[e] = leaks[i].weight[e]
        l._weight = w
        l._n_refl = leaks[i].n_refl
        out.append(l)
        polycap_leak_free(leaks[i])
    polycap_free(leaks)
    return out


cdef class Profile:
    """Shape of the optic: exterior radius, capillary radius and z along the optic (1000 points when generated)."""
    CONICAL = POLYCAP_PROFILE_CONICAL
    PARABOLOIDAL = POLYCAP_PROFILE_PARABOLOIDAL
    ELLIPSOIDAL = POLYCAP_PROFILE_ELLIPSOIDAL

    cdef polycap_profile *_profile

    def __cinit__(self, type=None, length=0., rad_ext_upstream=0., rad_ext_downstream=0., rad_int_upstream=0.,
                  rad_int_downstream=0., focal_dist_upstream=0., focal_dist_downstream=0.):
        cdef polycap_error *error = NULL
        self._profile = NULL
        if type is None:
            return
        cdef int itype = int(type)
        self._profile = polycap_profile_new(<polycap_profile_type> itype, length, rad_ext_upstream, rad_ext_downstream,
                                            rad_int_upstream, rad_int_downstream, focal_dist_upstream, focal_dist_downstream, &error)
        _raise_if(error)

    def __dealloc__(self):
        if self._profile != NULL:
            polycap_profile_free(self._profile)

    @staticmethod
    def new_from_arrays(ext, cap, z):
        cdef polycap_error *error = NULL
        e = np.ascontiguousarray(ext, dtype=np.float64)
        c = np.ascontiguousarray(cap, dtype=np.float64)
        zz = np.ascontiguousarray(z, dtype=np.float64)
        if e.ndim != 1 or c.ndim != 1 or zz.ndim != 1 or e.shape[0] != c.shape[0] or e.shape[0] != zz.shape[0]:
            raise ValueError("ext, cap and z must be 1-D arrays of identical length")
        cdef double[::1] ev = e
        cdef double[::1] cv = c
        cdef double[::1] zv = zz
        cdef Profile p = Profile.__new__(Profile)
        p._profile = polycap_profile_new_from_arrays(<int> e.shape[0] - 1, &ev[0], &cv[0], &zv[0], &error)
        _raise_if(error)
        return p

    @staticmethod
    def new_from_file(single_cap_profile_file, central_axis_file, external_shape_file):
        cdef polycap_error *error = NULL
        a = str(single_cap_profile_file).encode(); b = str(central_axis_file).encode(); c = str(external_shape_file).encode()
        cdef Profile p = Profile.__new__(Profile)
        p._profile = polycap_profile_new_from_file(a, b, c, &error)
        _raise_if(error)
        return p

    def get_ext(self):
        cdef size_t n = 0
        cdef double *p = NULL
        cdef polycap_error *error = NULL
        polycap_profile_get_ext(self._profile, &n, &p, &error)
        _raise_if(error)
        return _take_doubles(p, n + 1)

    def get_cap(self):
        cdef size_t n = 0
        cdef double *p = NULL
        cdef polycap_error *error = NULL
        polycap_profile_get_cap(self._profile, &n, &p, &error)
        _raise_if(error)
        return _take_doubles(p, n + 1)

    def get_z(self):
        cdef size_t n = 0
        cdef double *p = NULL
        cdef polycap_error *error = NULL
        polycap_profile_get_z(self._profile, &n, &p, &error)
        _raise_if(error)
        return _take_doubles(p, n + 1)


cdef class Rng:
    """Random number stream (Philox4x32-10, keyed by the seed)."""
    cdef polycap_rng *_rng

    def __cinit__(self, seed=None):
        if seed is None:
            self._rng = polycap_rng_new()
        else:
            if not isinstance(seed, (int, np.integer)):
                raise TypeError("seed must be an integer")
            if seed < 0:
                raise OverflowError("can't convert negative value to unsigned long")
            self._rng = polycap_rng_new_with_seed(<unsigned long> seed)

    def __dealloc__(self):
        if self._rng != NULL:
            polycap_rng_free(self._rng)


cdef class Description:
    """Optic description: profile + surface roughness + number of capillaries + glass composition and density."""
    cdef polycap_description *_description
    cdef object _owner          # set when the C object belongs to a Source

    def __cinit__(self, Profile profile=None, double sig_rough=0., int64_t n_cap=0, object composition=None, double density=0.):
        cdef polycap_error *error = NULL
        self._description = NULL
        self._owner = None
        if profile is None and composition is None:
            return
        if profile is None:
            raise ValueError("profile cannot be None")
        if isinstance(composition, str):
            comp = _parse_formula(composition)
        elif isinstance(composition, dict):
            if len(composition) == 0:
                raise ValueError("composition cannot be empty")
            comp = {}
            for k, v in composition.items():
                if k not in _SYMBOL_TO_Z:
                    raise ValueError("Invalid chemical symbol")
                comp[_SYMBOL_TO_Z[k]] = float(v)
        else:
            raise TypeError("composition must be a dictionary or a string")
        iz = np.ascontiguousarray(list(comp.keys()), dtype=np.intc)
        wi = np.ascontiguousarray(list(comp.values()), dtype=np.float64)
        cdef int[::1] izv = iz
        cdef double[::1] wiv = wi
        self._description = polycap_description_new(profile._profile, sig_rough, n_cap, <unsigned int> iz.shape[0], &izv[0], &wiv[0], density, &error)
        _raise_if(error)

    def __dealloc__(self):
        if self._description != NULL and self._owner is None:
            polycap_description_free(self._description)


cdef class Photon:
    """One photon: start coordinates, direction and electric vector; launch() traces it through the optic on the GPU."""
    cdef polycap_photon *_photon
    cdef object _description
    cdef public int return_code
    cdef object _leak_cache

    def _leaks(self, int kind):
        cdef polycap_error *error = NULL
        cdef polycap_leak **arr = NULL
        cdef int64_t n = 0
        if self._leak_cache is None:
            self._leak_cache = {}
        if kind not in self._leak_cache:
            if kind == 0:
                polycap_photon_get_extleak_data(self._photon, &arr, &n, &error)
            else:
                polycap_photon_get_intleak_data(self._photon, &arr, &n, &error)
            lst = _take_leaks(arr, n)
            _raise_if(error)
            self._leak_cache[kind] = lst
        return self._leak_cache[kind]

    @property
    def extleak_data(self):
        return (l for l in self._leaks(0))

    @property
    def intleak_data(self):
        return (l for l in self._leaks(1))

    def __cinit__(self, Description description=None, object start_coords=None, object start_direction=None, object start_electric_vector=None):
        cdef polycap_error *error = NULL
        self._photon = NULL
        self._description = description
        self.return_code = 0
        if description is None and start_coords is None:
            return
        if description is None:
            raise ValueError("description cannot be None")
        self._photon = polycap_photon_new(description._description, _vec(start_coords), _vec(start_direction), _vec(start_electric_vector), &error)
        _raise_if(error)

    def __dealloc__(self):
        if self._photon != NULL:
            polycap_photon_free(self._photon)

    def launch(self, object energies, cbool leak_calc=False):
        """weights per energy (ndarray), or None when the photon hit the glass at the entrance"""
        cdef polycap_error *error = NULL
        cdef double *weights = NULL
        e = np.atleast_1d(np.ascontiguousarray(energies, dtype=np.float64))
        cdef double[::1] ev = e
        self._leak_cache = None
        cdef int rc = polycap_photon_launch(self._photon, <size_t> e.shape[0], &ev[0], &weights, leak_calc, &error)
        out = _take_doubles(weights, e.shape[0]) if weights != NULL else None
        _raise_if(error)
        self.return_code = rc
        if rc == 2 or rc == -1:
            return None
        return out

    @property
    def start_coords(self):
        return _tuple(polycap_photon_get_start_coords(self._photon))

    @property
    def start_direction(self):
        return _tuple(polycap_photon_get_start_direction(self._photon))

    @property
    def start_electric_vector(self):
        return _tuple(polycap_photon_get_start_electric_vector(self._photon))

    @property
    def exit_coords(self):
        return _tuple(polycap_photon_get_exit_coords(self._photon))

    @property
    def exit_direction(self):
        return _tuple(polycap_photon_get_exit_direction(self._photon))

    @property
    def exit_electric_vector(self):
        return _tuple(polycap_photon_get_exit_electric_vector(self._photon))

    @property
    def d_travel(self):
        return polycap_photon_get_dtravel(self._photon)

    @property
    def i_refl(self):
        return polycap_photon_get_irefl(self._photon)


cdef class TransmissionEfficiencies:
    """Result of Source.get_transmission_efficiencies(): efficiency curve + per-exit-photon data."""
    cdef polycap_transmission_efficiencies *_eff
    cdef object _source
    cdef object _data
    cdef object _leak_cache

    def _leaks(self, int kind):
        cdef polycap_error *error = NULL
        cdef polycap_leak **arr = NULL
        cdef int64_t n = 0
        if self._leak_cache is None:
            self._leak_cache = {}
        if kind not in self._leak_cache:
            if kind == 0:
                polycap_transmission_efficiencies_get_extleak_data(self._eff, &arr, &n, &error)
            else:
                polycap_transmission_efficiencies_get_intleak_data(self._eff, &arr, &n, &error)
            lst = _take_leaks(arr, n)
            _raise_if(error)
            self._leak_cache[kind] = lst
        return self._leak_cache[kind]

    @property
    def extleak_data(self):
        return (l for l in self._leaks(0))

    @property
    def intleak_data(self):
        return (l for l in self._leaks(1))

    def __cinit__(self):
        self._eff = NULL
        self._data = None

    def __dealloc__(self):
        if self._eff != NULL:
            polycap_transmission_efficiencies_free(self._eff)

    @property
    def data(self):
        cdef size_t n = 0
        cdef double *e = NULL
        cdef double *f = NULL
        cdef polycap_error *error = NULL
        if self._data is None:
            polycap_transmission_efficiencies_get_data(self._eff, &n, &e, &f, &error)
            _raise_if(error)
            E = _take_doubles(e, n); F = _take_doubles(f, n)
            E.flags.writeable = False; F.flags.writeable = False
            self._data = (E, F)
        return self._data

    def _start(self):
        cdef int64_t n_start = 0, n_exit = 0
        cdef polycap_vector3 *a = NULL
        cdef polycap_vector3 *b = NULL
        cdef polycap_vector3 *c = NULL
        cdef polycap_vector3 *d = NULL
        cdef polycap_error *error = NULL
        polycap_transmission_efficiencies_get_start_data(self._eff, &n_start, &n_exit, &a, &b, &c, &d, &error)
        _raise_if(error)
        return n_start, n_exit, _take_vectors(a, n_exit), _take_vectors(b, n_exit), _take_vectors(c, n_exit), _take_vectors(d, n_exit)

    def _exit(self):
        cdef int64_t n_exit = 0
        cdef polycap_vector3 *a = NULL
        cdef polycap_vector3 *b = NULL
        cdef polycap_vector3 *c = NULL
        cdef int64_t *nrefl = NULL
        cdef double *dtr = NULL
        cdef size_t ne = 0
        cdef double **ww = NULL
        cdef polycap_error *error = NULL
        cdef int64_t i
        cdef size_t k
        polycap_transmission_efficiencies_get_exit_data(self._eff, &n_exit, &a, &b, &c, &nrefl, &dtr, &ne, &ww, &error)
        _raise_if(error)
        nr = np.empty(n_exit, dtype=np.int64)
        W = np.empty((n_exit, ne), dtype=np.float64)
        cdef int64_t[::1] nrv = nr
        cdef double[:, ::1] Wv = W
        for i in range(n_exit):
            nrv[i] = nrefl[i]
            for k in range(ne):
                Wv[i, k] = ww[i][k]
            polycap_free(ww[i])
        polycap_free(ww)
        polycap_free(nrefl)
        return n_exit, _take_vectors(a, n_exit), _take_vectors(b, n_exit), _take_vectors(c, n_exit), nr, _take_doubles(dtr, n_exit), W

    @property
    def start_coords(self):
        return (VectorTuple(*r) for r in self._start()[2])

    @property
    def start_direction(self):
        return (VectorTuple(*r) for r in self._start()[3])

    @property
    def start_electric_vector(self):
        return (VectorTuple(*r) for r in self._start()[4])

    @property
    def src_start_coords(self):
        return (VectorTuple(*r) for r in self._start()[5])

    @property
    def exit_coords(self):
        return (VectorTuple(*r) for r in self._exit()[1])

    @property
    def exit_direction(self):
        return (VectorTuple(*r) for r in self._exit()[2])

    @property
    def exit_electric_vector(self):
        return (VectorTuple(*r) for r in self._exit()[3])

    @property
    def n_refl(self):
        return self._exit()[4]

    @property
    def d_travel(self):
        return self._exit()[5]

    @property
    def exit_weights(self):
        return self._exit()[6]

    @staticmethod
    def from_totals(source, sum_weights, counters, images=None, exit_weights=None):
        """Extension of this build (no reference counterpart): result object from totals, and optionally the
        [n_exit, 17] image array + [n_exit, nE] weights of polycap_amd.TraceContext.images(), produced elsewhere
        (several GPUs / ranks), so that the getters and write_hdf5 serve a sharded run too."""
        return source._efficiencies_from_totals(sum_weights, counters, images, exit_weights)

    def write_hdf5(self, filename):
        cdef polycap_error *error = NULL
        cdef const char *fn = NULL
        if filename is not None:
            enc = str(filename).encode()
            fn = enc
        polycap_transmission_efficiencies_write_hdf5(self._eff, fn, &error)
        _raise_if(error)


cdef class Source:
    """X-ray source + optic + energy grid."""
    cdef polycap_source *_source

    def __cinit__(self, Description description=None, double d_source=0., double src_x=0., double src_y=0., double src_sigx=0.,
                  double src_sigy=0., double src_shiftx=0., double src_shifty=0., double hor_pol=0., object energies=None):
        cdef polycap_error *error = NULL
        self._source = NULL
        if description is None and energies is None:
            return
        if description is None:
            raise ValueError("description cannot be None")
        e = np.atleast_1d(np.ascontiguousarray(energies, dtype=np.float64))
        cdef double[::1] ev = e
        self._source = polycap_source_new(description._description, d_source, src_x, src_y, src_sigx, src_sigy, src_shiftx, src_shifty,
                                          hor_pol, <size_t> e.shape[0], &ev[0], &error)
        _raise_if(error)

    def __dealloc__(self):
        if self._source != NULL:
            polycap_source_free(self._source)

    @staticmethod
    def new_from_file(filename):
        cdef polycap_error *error = NULL
        cdef const char *fn = NULL
        if filename is not None:
            enc = str(filename).encode()
            fn = enc
        cdef Source s = Source.__new__(Source)
        s._source = polycap_source_new_from_file(fn, &error)
        _raise_if(error)
        return s

    def get_photon(self, Rng rng):
        cdef polycap_error *error = NULL
        if rng is None:
            raise ValueError("rng cannot be None")
        cdef polycap_photon *p = polycap_source_get_photon(self._source, rng._rng, &error)
        _raise_if(error)
        cdef Description d = Description.__new__(Description)
        d._description = <polycap_description *> polycap_source_get_description(self._source)
        d._owner = self
        cdef Photon ph = Photon.__new__(Photon)
        ph._photon = p
        ph._description = d
        return ph

    def get_transmission_efficiencies(self, int max_threads, int n_photons, cbool leak_calc=False):
        cdef polycap_error *error = NULL
        cdef polycap_transmission_efficiencies *e = polycap_source_get_transmission_efficiencies(self._source, max_threads, n_photons,
                                                                                                 leak_calc, NULL, &error)
        _raise_if(error)
        cdef TransmissionEfficiencies t = TransmissionEfficiencies.__new__(TransmissionEfficiencies)
        t._eff = e
        t._source = self
        return t

    def _efficiencies_from_totals(self, sum_weights, counters, images, exit_weights):
        cdef polycap_error *error = NULL
        cdef pc_hip_images im
        cdef pc_hip_images *imp = NULL
        cdef double[::1] sw = np.ascontiguousarray(sum_weights, dtype=np.float64).ravel()
        cdef double[:, ::1] planes
        cdef int64_t[::1] nrefl
        cdef double[:, ::1] w
        cnt_np = np.zeros(6, dtype=np.int64)
        cc = np.asarray(counters, dtype=np.int64).ravel()
        cnt_np[:min(6, cc.size)] = cc[:6]
        cdef int64_t[::1] cnt = cnt_np
        cdef int64_t n = cnt[0]
        cdef int k
        if images is not None:
            img = np.asarray(images, dtype=np.float64)
            if img.ndim != 2 or img.shape[0] != n or img.shape[1] != 17:
                raise ValueError("images must have shape (counters[0], 17)")
            wnp = np.zeros((n, sw.shape[0])) if exit_weights is None else np.ascontiguousarray(exit_weights, dtype=np.float64)
            if wnp.ndim != 2 or wnp.shape[0] != n or wnp.shape[1] != sw.shape[0]:
                raise ValueError("exit_weights must have shape (counters[0], n_energies)")
            if n > 0:
                planes = np.ascontiguousarray(img.T)
                nrefl = np.ascontiguousarray(np.rint(img[:, 15]).astype(np.int64))
                w = wnp
                for k in range(2):
                    im.src_start_coords[k] = &planes[k, 0]
                    im.pc_start_coords[k] = &planes[2 + k, 0]
                    im.pc_start_dir[k] = &planes[4 + k, 0]
                    im.pc_start_elecv[k] = &planes[6 + k, 0]
                    im.pc_exit_dir[k] = &planes[11 + k, 0]
                    im.pc_exit_elecv[k] = &planes[13 + k, 0]
                for k in range(3):
                    im.pc_exit_coords[k] = &planes[8 + k, 0]
                im.pc_exit_nrefl = &nrefl[0]
                im.pc_exit_dtravel = &planes[16, 0]
                im.exit_coord_weights = &w[0, 0]
                imp = &im
        cdef void *e = pc_transmission_efficiencies_from_totals(<void *> self._source, n, &sw[0], &cnt[0], imp, <void *> &error)
        _raise_if(error)
        cdef TransmissionEfficiencies t = TransmissionEfficiencies.__new__(TransmissionEfficiencies)
        t._eff = <polycap_transmission_efficiencies *> e
        t._source = self
        return t
