"""Wall time of small calls through the public API (the sizes of the reference's own tests: tests/source.c, tests/python.py):
polycap_source_get_transmission_efficiencies with 1000 ... 100000 photons, first call and later calls of the same source."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from polycap_amd import capi as polycap

prof = polycap.Profile(polycap.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-05, 1000., 0.5)
desc = polycap.Description(prof, 0., 200000, {"O": 53.0, "Si": 47.0}, 2.23)
for energies in (np.array([10.0]), np.array([1., 5., 10., 15., 20., 25., 30.]), np.linspace(1, 30, 291)):
    src = polycap.Source(desc, 2000., 0.2065, 0.2065, 0.0, 0.0, 0., 0., 0.5, energies)
    for n in (1000, 1000, 30000, 30000, 100000, 100000):
        t0 = time.perf_counter()
        eff = src.get_transmission_efficiencies(-1, n, False)
        dt = time.perf_counter() - t0
        d = eff.data
        print("%3d energies, %6d photons: %.1f ms" % (len(energies), n, dt*1e3), flush=True)
        del eff
