"""Host-side logic of libpolycap through the reference-shaped API (no GPU needed): profiles, descriptions,
sources, the .inp parser, optical constants and the error convention.  Expected values are the reference's
own test expectations (tests/profile.c, tests/description.c, tests/source.c, tests/python.py)."""
import os

import numpy as np
import pytest

from tests.conftest import EXAMPLE
from tests.common import TEST_SHAPE


@pytest.fixture(scope="module", params=["ctypes", "cython"])
def capi(request):
    """The reference-shaped Python API, once through ctypes (polycap_amd.capi) and once through the compiled Cython
    module `polycap` (polycap_amd/pyext, built by build())."""
    if request.param == "ctypes":
        from polycap_amd import capi
        return capi
    import os
    import sys
    from tests.conftest import ROOT
    from polycap_amd import _build
    _build.build_cython()
    ext = os.path.join(ROOT, "polycap_amd", "pyext")
    if ext not in sys.path:
        sys.path.insert(0, ext)
    import polycap
    assert polycap.__version__ == "1.2"
    return polycap


def test_profile_shapes_match_oracle(capi, oracle):
    # conical and ellipsoidal generators: same formulas as the oracle restatement, bit for bit
    for ptype in (0, 2):
        args = (9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
        p = capi.Profile(ptype, *args)
        o = oracle.Optic.from_shape(ptype, *args, 0., 200000, 2.23)
        assert np.array_equal(p.get_z(), o.z) and np.array_equal(p.get_cap(), o.cap) and np.array_equal(p.get_ext(), o.ext)
    # confocal (collimating) ellipsoid branch
    p = capi.Profile(2, 6., 2e-5, 2e-4, 1e-5, 1e-4, 1.0, 1.0)
    o = oracle.Optic.from_shape(2, 6., 2e-5, 2e-4, 1e-5, 1e-4, 1.0, 1.0, 0., 1000, 2.23)
    assert np.array_equal(p.get_ext(), o.ext, equal_nan=True)


def test_profile_paraboloidal_fit(capi):
    # reference tests/profile.c:55-57: a paraboloidal profile can be built; the exterior is the least-squares parabola
    # through the entrance, exit and the two focal-line points (src/polycap-profile.c:149-169)
    L, ru, rd, fu, fd = 6., 2e-5, 2e-4, 1.0, 1.0
    p = capi.Profile(capi.Profile.PARABOLOIDAL, L, ru, rd, 1e-5, 1e-4, fu, fd)
    x = np.array([0., fu / 10., L - fd / 10., L])
    y = np.array([ru, ru / fu * x[1] + ru, (rd / (L - (L + fd))) * (x[2] - L) + rd, rd])
    coef = np.linalg.lstsq(np.vander(x, 3, increasing=True), y, rcond=None)[0]
    z = p.get_z()
    assert np.allclose(p.get_ext(), coef[0] + coef[1] * z + coef[2] * z * z, rtol=1e-9, atol=1e-15)
    assert z[0] == 0 and z[-1] == L and len(z) == 1000


def test_profile_errors_and_arrays(capi):
    with pytest.raises(ValueError, match="polycap_profile_new: length must be greater than 0.0"):
        capi.Profile(capi.Profile.CONICAL, -1, 2e-5, 2e-4, 1e-5, 1e-4, 1.0, 1.0)
    with pytest.raises(IOError):
        capi.Profile.new_from_file("this-file-does-not-exist", "this-file-also-does-not-exist", "neither-does-this-one")
    assert (capi.Profile.CONICAL, capi.Profile.PARABOLOIDAL, capi.Profile.ELLIPSOIDAL) == (0, 1, 2)
    ext, cap, z = np.linspace(2e-5, 2e-4, 1000), np.linspace(1e-5, 1e-4, 1000), np.linspace(0., 6., 1000)
    p = capi.Profile.new_from_arrays(ext, cap, z)
    assert np.array_equal(p.get_ext(), ext) and np.array_equal(p.get_cap(), cap) and np.array_equal(p.get_z(), z)
    p = capi.Profile.new_from_file(*(os.path.join(EXAMPLE, "xos1." + e) for e in ("prf", "axs", "ext")))
    assert len(p.get_z()) == 1000 and p.get_cap()[0] == 0.00035 and p.get_ext()[0] == 0.2065


def test_description_errors(capi):
    prof = capi.Profile(*TEST_SHAPE)
    comp = {"O": 53.0, "Si": 47.0}
    with pytest.raises(ValueError, match="Invalid chemical symbol"):
        capi.Description(prof, 0.0, 1000, {"Bad": 53.0, "Ugly": 47.0}, 2.23)
    with pytest.raises(ValueError, match="polycap_description_new: n_cap must be greater than 1"):
        capi.Description(prof, 0.0, 0, comp, 2.23)
    with pytest.raises(ValueError, match="polycap_description_new: density must be greater than 0.0"):
        capi.Description(prof, 0.0, 1000, comp, 0.0)
    with pytest.raises(ValueError, match="polycap_description_new: sig_rough must be greater than or equal to zero"):
        capi.Description(prof, -1.0, 1000, comp, 2.23)
    with pytest.raises(ValueError, match="composition cannot be empty"):
        capi.Description(prof, 0.0, 1000, {}, 2.23)
    with pytest.raises(ValueError, match="Invalid chemical formula"):
        capi.Description(prof, 0.0, 1000, "sjalalala", 2.23)
    with pytest.raises(TypeError, match="composition must be a dictionary or a string"):
        capi.Description(prof, 0.0, 1000, 25, 2.23)
    capi.Description(prof, 0.0, 200000, comp, 2.23)
    capi.Description(prof, 0.0, 200000, "SiO2", 2.23)
    # a profile whose capillaries poke out of the exterior is rejected (src/polycap-description.c:218-229)
    bad = capi.Profile.new_from_arrays(np.full(200, 0.01), np.full(200, 0.009), np.linspace(0, 1, 200))
    with pytest.raises(ValueError, match="description->profile is faulty"):
        capi.Description(bad, 0.0, 200000, comp, 2.23)


def test_rng_and_photon_argument_checks(capi):
    assert isinstance(capi.Rng(), capi.Rng) and isinstance(capi.Rng(12345678), capi.Rng)
    with pytest.raises(TypeError):
        capi.Rng("this-is-not-a-seed")
    with pytest.raises(OverflowError):
        capi.Rng(-523)
    prof = capi.Profile(*TEST_SHAPE)
    desc = capi.Description(prof, 0.0, 200000, {"O": 53.0, "Si": 47.0}, 2.23)
    with pytest.raises(ValueError):
        capi.Photon(None, (0, 0, 0), (0.005, -0.005, 0.1), (0.5, 0.5, 0))
    with pytest.raises(ValueError):
        capi.Photon(desc, (0, 0, -1.), (0.005, -0.005, 0.1), (0.5, 0.5, 0))
    ph = capi.Photon(desc, (0, 0, 0), (0.005, -0.005, 0.1), (0.5, 0.5, 0))
    assert ph.start_coords == (0, 0, 0) and ph.d_travel == 0 and ph.i_refl == 0
    with pytest.raises(ValueError, match="energies"):
        ph.launch([0.5])              # energy below 1 keV is rejected before any device work
    import polycap_amd
    if polycap_amd.device_count() == 0:
        # no CPU trace path, with or without the leak calculation: the launch fails loudly
        for leak in (False, True):
            with pytest.raises(RuntimeError, match="HIP"):
                ph.launch([10.0], leak_calc=leak)


def test_source_new_and_from_file(capi, known):
    prof = capi.Profile(*TEST_SHAPE)
    desc = capi.Description(prof, 0.0, 200000, {"O": 53.0, "Si": 47.0}, 2.23)
    E = np.array([1, 5, 10, 15, 20, 25, 30.])
    for bad in (dict(d_source=-1), dict(src_x=-1), dict(src_y=0), dict(hor_pol=1.5)):
        kw = dict(d_source=2000., src_x=0.2065, src_y=0.2065, hor_pol=0.5)
        kw.update(bad)
        with pytest.raises(ValueError, match="polycap_source_new"):
            capi.Source(desc, kw["d_source"], kw["src_x"], kw["src_y"], 0., 0., 0., 0., kw["hor_pol"], E)
    with pytest.raises(ValueError, match="energies must be greater than 1 and smaller than 100"):
        capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.5, np.array([0.5]))
    capi.Source(desc, 2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.5, E)
    with pytest.raises(ValueError):
        capi.Source.new_from_file(None)
    with pytest.raises(IOError):
        capi.Source.new_from_file("this-file-does-not-exist")
    src = capi.Source.new_from_file(os.path.join(EXAMPLE, "ellip_l9.inp"))
    with pytest.raises(ValueError, match="n_photons must be greater than 1"):
        src.get_transmission_efficiencies(-1, -1)
    import polycap_amd
    if polycap_amd.device_count() == 0:
        with pytest.raises(RuntimeError, match="HIP"):
            src.get_transmission_efficiencies(1, 10, leak_calc=True)


def test_inp_decks_and_open_area(known):
    import polycap_amd
    p = polycap_amd.problem_from_inp(os.path.join(EXAMPLE, "ellip_l9.inp"))
    # reference tests/source.c:116: open area of ellip_l9.inp
    n_shells = round(np.sqrt(12. * p.n_cap - 3.) / 6. - 0.5)
    ncap = ((n_shells + 0.5) * 6.) ** 2
    ncap = (ncap + 3) / 12
    open_area = (p.cap[0] ** 2 * np.pi) * ncap / (3. * np.sin(np.pi / 3) * p.ext[0] ** 2)
    assert abs(open_area - known["test_optic"]["open_area"]) < 1e-5
    assert p.n_energies == 291 and p.energies[0] == 1.0 and abs(p.energies[-1] - 30.0) < 1e-9   # (30-1)/0.1+1 truncated
    assert p.n_cap == 200000 and p.nmax == 999 and p.source == (2000.0, 0.2065, 0.2065, 0.0, 0.0, 0.0, 0.0, 0.0)
    x = polycap_amd.problem_from_inp(os.path.join(EXAMPLE, "xos1.inp"), energies=[10.0])
    assert x.nmax == 999 and x.cap[0] == 0.00035 and x.ext[-1] == 0.0585 and x.sig_rough == 0.0
    c = polycap_amd.problem_from_inp(os.path.join(EXAMPLE, "cone.inp"), energies=[10.0])
    assert c.source[2] == 0.0   # the degenerate src_y = 0 deck


def test_optical_constants_pin(known, monkeypatch):
    import polycap_amd
    monkeypatch.setenv("POLYCAP_OPTCONST", "builtin")     # the tables of B, Na, ... are used only on request (tests/test_xraylib_binding.py)
    g = known["glass"]
    amu, scatf, synthetic = polycap_amd.optical_constants(g["iz"], g["wi_percent"], g["density"], [g["energy_keV"]])
    assert abs(scatf[0] - g["scatf"]) < g["scatf_tol"] and abs(amu[0] - g["amu"]) < g["amu_tol"]
    assert not synthetic
    amu, scatf, synthetic = polycap_amd.optical_constants(g["iz"], g["wi_percent"], g["density"], [1., 1.8, 1.9, 5., 30., 100.])
    assert synthetic and np.all(amu > 0) and np.all(scatf > 0.3) and amu[2] > amu[1]    # Si K edge between 1.8 and 1.9 keV
    # elements of common capillary glasses come from the built-in tables, always flagged synthetic (unverified offline);
    # anything else needs xraylib
    with pytest.raises(NotImplementedError, match="no optical constants for Z=26"):
        polycap_amd.optical_constants([26], [1.0], 7.9, [10.0])
    boro = ([5, 8, 11, 13, 14, 19], [4.0, 53.9, 2.8, 1.1, 37.7, 0.5], 2.23)      # borosilicate glass
    amu, scatf, synthetic = polycap_amd.optical_constants(*boro, [3.0, 3.7, 10.0, 30.0])
    assert synthetic and np.all(np.diff(amu[[0, 2, 3]]) < 0) and np.all((scatf > 0.45) & (scatf < 0.52))
    lead = ([8, 14, 19, 82], [30., 25., 5., 40.], 4.0)                          # lead glass: L edges at 13.0, 15.2 and 15.9 keV
    amu, scatf, synthetic = polycap_amd.optical_constants(*lead, [12.9, 13.2, 15.1, 16.0, 87.0, 89.0])
    assert synthetic and amu[1] > 1.5 * amu[0] and amu[3] > amu[2] and amu[5] > 2 * amu[4]
    # Z / A of the compound minus the anomalous part: scatf stays a little below sum(w Z / A)
    zoa = sum(w / 100. * z / a for z, w, a in zip(lead[0], lead[1], (15.9994, 28.0855, 39.0983, 207.2)))
    assert np.all(scatf < zoa) and np.all(scatf > 0.9 * zoa)
    # the pinned 10 keV pair of the reference glass is untouched by the other tables
    amu, scatf, synthetic = polycap_amd.optical_constants(g["iz"], g["wi_percent"], g["density"], [10.0])
    assert abs(amu[0] - g["amu"]) < g["amu_tol"] and not synthetic
    with pytest.raises(ValueError, match="energies"):
        polycap_amd.optical_constants(g["iz"], g["wi_percent"], g["density"], [0.5])


def test_command_line_program_without_gpu(tmp_path):
    """reference src/main.c: usage line without arguments, message + status 1 for an unreadable deck; with a deck but
    no GPU the run fails loudly (no CPU fallback)."""
    import subprocess
    import polycap_amd
    from tests.conftest import ROOT
    exe = os.path.join(ROOT, "polycap_amd", "bin", "polycap")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "input-file should be supplied" in r.stdout
    r = subprocess.run([exe, str(tmp_path / "missing.inp")], capture_output=True, text=True)
    assert r.returncode == 1 and "could not open" in r.stderr
    if polycap_amd.device_count() == 0:
        r = subprocess.run([exe, os.path.join(EXAMPLE, "cone.inp"), str(tmp_path / "out.h5")], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "out.h5").exists()
