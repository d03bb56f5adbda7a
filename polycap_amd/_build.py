"""In-tree build of libpolycap.so: host C (gcc, C11) + HIP kernels (hipcc, gfx950), one shared library.

    python -m polycap_amd._build        # or __graft_entry__.build()

hipcc cross-compiles for gfx950 without a GPU.  The built .so stays in-tree (polycap_amd/lib/) so it
travels with the repository snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
HOST = os.path.join(HERE, "csrc", "host")
HIPD = os.path.join(HERE, "csrc", "hip")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libpolycap.so")

HOST_SRCS = ["pc_error.c", "pc_rng.c", "pc_profile.c", "pc_description.c", "pc_optconst.c",
             "pc_photon.c", "pc_source.c", "pc_transeff.c", "pc_hdf5.c"]
HIP_SRCS = ["pc_kernels.hip"]
HIP_DEPS = ["pc_device.h", "pc_problem.h", "pc_leak.h", "pc_leak_kernels.h", "pc_pool_kernel.h", "pc_producer_kernel.h", "pc_wave_kernel.h", "pc_sweep_kernel.h", "pc_group.h"]

CFLAGS = ["-std=c11", "-O2", "-fPIC", "-Wall", "-Wextra", "-fvisibility=hidden", "-I" + INC, "-I" + HOST]
# -ffp-contract=off: fused multiply-adds appear only where pc_device.h writes fma() explicitly, so the device
# arithmetic is the same IEEE operation sequence on gfx950 and in the host-compiled test emulation
HIPFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
            "-I" + INC, "-I" + HIPD]
if os.environ.get("POLYCAP_EXPERIMENTS"):      # A/B scripts of kernels that are not part of the product (pc_wave_kernel.h)
    HIPFLAGS.append("-DPC_EXPERIMENTS")
# what the library exports: the reference's C API, the thin HIP C-ABI and the few host helpers the Python layer binds; the
# kernels' host stubs and the C++ runtime's weak instantiations stay local (reference: meson.build:85-100, default-hidden
# visibility with POLYCAP_EXTERN only)
VERSION_SCRIPT = os.path.join(HERE, "csrc", "libpolycap.map")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(INC, "polycap.h"), os.path.join(INC, "polycap-hip.h"), os.path.join(HOST, "pc_private.h")]
    objs = []
    for s in HOST_SRCS:
        src = os.path.join(HOST, s)
        obj = os.path.join(OBJDIR, s + ".o")
        if force or _newer(obj, [src] + headers):
            _run(["gcc"] + CFLAGS + ["-c", src, "-o", obj], verbose)
        objs.append(obj)
    for s in HIP_SRCS:
        src = os.path.join(HIPD, s)
        obj = os.path.join(OBJDIR, s + ".o")
        deps = [src] + [os.path.join(HIPD, d) for d in HIP_DEPS] + headers[:2]
        if force or _newer(obj, deps):
            _run(["hipcc"] + HIPFLAGS + ["-c", src, "-o", obj], verbose)
        objs.append(obj)
    if force or _newer(LIB, objs + [VERSION_SCRIPT]):
        _run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,--version-script=" + VERSION_SCRIPT, "-o", LIB] + objs +
             ["-ldl", "-lm", "-lpthread"], verbose)
    build_cli(force=force, verbose=verbose)
    build_cython(force=force, verbose=verbose)
    return LIB


def build_cli(force=False, verbose=False):
    """The `polycap` command-line program (reference src/main.c): polycap_amd/bin/polycap, rpath to ../lib."""
    src = os.path.join(HOST, "pc_main.c")
    exe = os.path.join(HERE, "bin", "polycap")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if force or _newer(exe, [src, LIB]):
        _run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-I" + INC, src, "-L" + LIBDIR, "-lpolycap",
              "-Wl,-rpath,$ORIGIN/../lib", "-o", exe], verbose)
    return exe


def build_cython(force=False, verbose=False):
    """The `polycap` Python module (Cython, same surface as the reference's python/polycap.pyx), linked against
    libpolycap.so with an $ORIGIN-relative rpath; lands in polycap_amd/pyext/ (add that directory to sys.path)."""
    import sysconfig
    pyx = os.path.join(HERE, "pyext", "polycap.pyx")
    cpp = os.path.join(HERE, "pyext", "polycap.cpp")
    so = os.path.join(HERE, "pyext", "polycap" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if force or _newer(cpp, [pyx]):
        _run([sys.executable, "-m", "cython", "-3", "--cplus", pyx, "-o", cpp], verbose)
    if force or _newer(so, [cpp, LIB]):
        _run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w", "-I" + sysconfig.get_paths()["include"], "-I" + INC, cpp,
              "-L" + LIBDIR, "-lpolycap", "-Wl,-rpath,$ORIGIN/../lib", "-o", so], verbose)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
