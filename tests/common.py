"""Shared builders for the parity tests: the same optic as an oracle Optic and as a product Problem."""
import os

import numpy as np

from tests.conftest import EXAMPLE

GLASS = dict(iz=[8, 14], wi=[53.0, 47.0], density=2.23)
PIN_E, PIN_AMU, PIN_SCATF = 10.0, 42.544635, 0.503696   # reference tests/photon.c:75-76
TEST_SHAPE = (2, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)   # ellipsoidal optic of the reference's tests


def load_xos1_tables():
    prf = np.loadtxt(os.path.join(EXAMPLE, "xos1.prf"), skiprows=1)
    ext = np.loadtxt(os.path.join(EXAMPLE, "xos1.ext"), skiprows=1)
    return prf[:, 0].copy(), prf[:, 1].copy(), ext[:, 1].copy()


def synthetic_constants(energies):
    """Smooth stand-in optical constants anchored at the pinned 10 keV pair (tests only)."""
    E = np.asarray(energies, dtype=np.float64)
    return PIN_AMU * (E / PIN_E) ** -2.8, np.full_like(E, PIN_SCATF)


def make_pair(oracle, which="ellip", energies=(PIN_E,), amu=None, scatf=None, sig_rough=0.0,
              source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.5), n_cap=200000):
    """Returns (oracle Optic, oracle source struct, product Problem) describing the same run."""
    from polycap_amd import Problem
    E = np.asarray(energies, dtype=np.float64)
    if amu is None:
        if len(E) == 1 and E[0] == PIN_E:
            amu, scatf = np.array([PIN_AMU]), np.array([PIN_SCATF])
        else:
            amu, scatf = synthetic_constants(E)
    if which == "ellip":
        optic = oracle.Optic.from_shape(*TEST_SHAPE, sig_rough, n_cap, GLASS["density"])
    elif which == "xos1":
        z, cap, ext = load_xos1_tables()
        optic = oracle.Optic(z, cap, ext, sig_rough, n_cap, GLASS["density"])
    else:
        raise ValueError(which)
    src = oracle.make_source(*source)
    prob = Problem(optic.z, optic.cap, optic.ext, sig_rough, n_cap, GLASS["density"], E, amu, scatf, *source)
    return optic, src, prob, (E, np.asarray(amu, dtype=np.float64), np.asarray(scatf, dtype=np.float64))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def make_custom(oracle, shape, n_cap, source, energies=(PIN_E,), sig_rough=0.0):
    """(optic, source, problem, (E, amu, scatf)) for an analytic profile `shape` = (type, length, rext_up, rext_down,
    rint_up, rint_down, f_up, f_down)."""
    from polycap_amd import Problem
    E = np.asarray(energies, dtype=np.float64)
    if len(E) == 1 and E[0] == PIN_E:
        amu, scatf = np.array([PIN_AMU]), np.array([PIN_SCATF])
    else:
        amu, scatf = synthetic_constants(E)
    optic = oracle.Optic.from_shape(*shape, sig_rough, n_cap, GLASS["density"])
    src = oracle.make_source(*source)
    prob = Problem(optic.z, optic.cap, optic.ext, sig_rough, n_cap, GLASS["density"], E, amu, scatf, *source)
    return optic, src, prob, (E, amu, scatf)


# monocap.inp-like single conical capillary under uniform illumination; 7-capillary optic with a divergent source
MONO_CASE = dict(shape=(0, 15., 0.012, 0.012, 0.01, 0.0005, 1000., 0.5), n_cap=2, source=(5., 0.0001, 0.0001, -1., 0., 0., 0., 0.))
SEVEN_CASE = dict(shape=(0, 5., 0.05, 0.04, 0.012, 0.009, 1000., 0.5), n_cap=7, source=(50., 0.05, 0.05, 0.002, 0.002, 0., 0., 0.3))
