"""The dlopen binding of xraylib (polycap_amd/csrc/host/pc_optconst.c) against a test double of libxrl.

The reference takes CS_Total / Fi / AtomicWeight from xraylib (src/polycap-photon.c:87-88); this image has no libxrl,
so the binding is exercised with tests/fake_xrl/fake_xrl.c built as libxrl.so.11 on LD_LIBRARY_PATH: the provider must be
xraylib, the values must pass through, the reference's 10 keV pin (tests/photon.c:75-76) must hold, and an error reported
through xrl_error** must surface as POLYCAP_ERROR_RUNTIME.  CPU only."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

CHILD = r"""
import ctypes, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import polycap_amd
from polycap_amd import _cabi, decks
L = _cabi.lib(); decks._protos(L)
L.pc_optconst_library.restype = ctypes.c_char_p
out = {"provider": L.pc_optconst_provider().decode(), "library": L.pc_optconst_library().decode()}
try:
    amu, scatf, syn = polycap_amd.optical_constants([8, 14], [53.0, 47.0], 2.23, [10.0])
    out.update(amu=float(amu[0]), scatf=float(scatf[0]), synthetic=bool(syn))
    # any composition, any energy: the stand-in formulas, i.e. the numbers really come from the library
    amu2, scatf2, syn2 = polycap_amd.optical_constants([26, 82], [40.0, 60.0], 7.0, [3.0, 55.0])
    out.update(amu2=[float(x) for x in amu2], scatf2=[float(x) for x in scatf2], synthetic2=bool(syn2))
    fake = ctypes.CDLL(out["library"]) if os.path.sep in out["library"] else ctypes.CDLL("libxrl.so.11")
    out["calls"] = [fake.fake_xrl_calls(k) for k in range(3)]
except Exception as e:
    out["error"] = "%%s: %%s" %% (type(e).__name__, e)
print(json.dumps(out))
"""


@pytest.fixture(scope="module")
def fake_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_xrl")
    so = d / "libxrl.so.11"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-fvisibility=hidden",
                           os.path.join(ROOT, "tests", "fake_xrl", "fake_xrl.c"), "-o", str(so)])
    return str(d)


def _child(fake_dir, how="ld_path", **extra):
    import json
    env = dict(os.environ)
    env.pop("POLYCAP_OPTCONST", None)
    if how == "ld_path":
        env["LD_LIBRARY_PATH"] = fake_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    elif how == "env_path":
        env["POLYCAP_XRL_LIBRARY"] = os.path.join(fake_dir, "libxrl.so.11")
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


@pytest.mark.parametrize("how", ["ld_path", "env_path"])
def test_values_pass_through_the_xraylib_binding(fake_dir, how):
    out, _ = _child(fake_dir, how)
    assert out["provider"] == "xraylib" and "libxrl.so.11" in out["library"]
    assert "error" not in out, out
    # the reference's pin: scatf 0.503696 +- 1e-5, amu 42.544635 +- 1e-3 (tests/photon.c:75-76)
    assert abs(out["scatf"] - 0.503696) < 1e-5 and abs(out["amu"] - 42.544635) < 1e-3
    assert out["synthetic"] is False and out["synthetic2"] is False
    # the stand-in formulas for an arbitrary compound: amu = density * 42.544677/2.23 * sum(w), scatf = 0.503696 * sum(w)
    for a in out["amu2"]:
        assert abs(a - 7.0 * 42.544677 / 2.23) < 1e-9
    for s in out["scatf2"]:
        assert abs(s - 0.503696) < 1e-12
    # 1 energy x 2 elements + 2 energies x 2 elements = 6 lookups of each kind went through the library
    assert out["calls"] == [6, 6, 6]


@pytest.mark.parametrize("which", ["CS_Total", "Fi", "AtomicWeight"])
def test_an_xraylib_error_is_a_runtime_error(fake_dir, which):
    out, _ = _child(fake_dir, FAKE_XRL_FAIL=which)
    assert out["provider"] == "xraylib"
    # POLYCAP_ERROR_RUNTIME -> RuntimeError in the Python layer (reference python/polycap.pyx:91-107)
    assert out.get("error", "").startswith("RuntimeError") and "xraylib %s" % which in out["error"] and "requested failure" in out["error"]


def test_builtin_choice_wins_over_an_installed_xraylib(fake_dir):
    out, _ = _child(fake_dir, POLYCAP_OPTCONST="builtin")
    assert out["provider"].startswith("built-in") and out["library"] == ""
    assert abs(out["scatf"] - 0.503696) < 1e-5 and abs(out["amu"] - 42.544635) < 1e-3    # the pinned pair comes from the table too


def test_unverified_tables_need_the_opt_in(monkeypatch):
    """built-in tables for elements other than O/Si are used only on request (POLYCAP_OPTCONST=builtin)."""
    import numpy as np
    import polycap_amd
    if polycap_amd.optical_constants_provider() == "xraylib":
        pytest.skip("a real xraylib is installed")
    boro = ([5, 8, 11, 13, 14, 19], [4.0, 53.9, 2.8, 1.1, 37.7, 0.5], 2.23)
    monkeypatch.delenv("POLYCAP_OPTCONST", raising=False)
    with pytest.raises(NotImplementedError, match="POLYCAP_OPTCONST=builtin"):
        polycap_amd.optical_constants(*boro, [10.0])
    # the O/Si glass of the reference's decks needs no opt-in and is flagged synthetic away from 10 keV
    amu, scatf, syn = polycap_amd.optical_constants([8, 14], [53., 47.], 2.23, [5.0])
    assert syn and amu[0] > 0
    monkeypatch.setenv("POLYCAP_OPTCONST", "builtin")
    amu, scatf, syn = polycap_amd.optical_constants(*boro, [10.0])
    assert syn and np.all(amu > 0)
