"""TEST-ONLY: host compile of the device header (see pc_emul.cpp); never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from polycap_amd._cabi import ProblemS, c_double_p, c_int64_p, dptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        extra = os.environ.get("PC_EMUL_FLAGS", "").split()      # analysis builds (other strides ...): a library of their own
        so = os.path.join(_HERE, "libpc_emul.so" if not extra else "libpc_emul_%08x.so" % (hash(tuple(extra)) & 0xffffffff))
        srcs = [os.path.join(_HERE, "pc_emul.cpp"),
                os.path.join(_ROOT, "polycap_amd", "csrc", "hip", "pc_device.h"),
                os.path.join(_ROOT, "polycap_amd", "csrc", "hip", "pc_problem.h"),
                os.path.join(_ROOT, "polycap_amd", "csrc", "hip", "pc_leak.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                                   "-ffp-contract=off", "-mfma", "-fopenmp",   # same IEEE operation sequence as the gfx950 build (fma only where written)
                                   "-I" + os.path.join(_ROOT, "include"),
                                   "-I" + os.path.join(_ROOT, "polycap_amd", "csrc", "hip"),
                                   "-o", so, srcs[0]] + extra)
        L = C.CDLL(so)
        L.emul_launch_batch.argtypes = [C.POINTER(ProblemS), C.c_int, C.c_int, C.c_int64, c_double_p, c_double_p, c_double_p,
                                        C.POINTER(C.c_int32), c_double_p, c_double_p, c_double_p, c_double_p,
                                        c_int64_p, c_double_p, c_int64_p]
        L.emul_launch_batch.restype = C.c_int
        L.emul_launch_leak.argtypes = [C.POINTER(ProblemS), C.c_int, C.c_int64, c_double_p, c_double_p, c_double_p, C.c_int, C.c_int64,
                                       C.POINTER(C.c_int32), c_double_p, c_double_p, c_double_p, c_double_p,
                                       c_int64_p, c_double_p, c_double_p, c_int64_p, C.POINTER(C.c_int32)]
        L.emul_launch_leak.restype = C.c_int
        L.emul_transmission_leak.argtypes = [C.POINTER(ProblemS), C.c_uint64, C.c_int64, C.c_int64, C.c_uint32, C.c_int, C.c_int64,
                                             c_double_p, c_int64_p, c_double_p, c_double_p, c_int64_p, C.POINTER(C.c_int32)]
        L.emul_transmission_leak.restype = C.c_int
        L.emul_transmission.argtypes = [C.POINTER(ProblemS), C.c_uint64, C.c_int64, C.c_int64, C.c_uint32, C.c_int,
                                        c_int64_p, C.POINTER(C.c_uint64), c_double_p]
        L.emul_transmission.restype = C.c_int
        L.emul_sample.argtypes = [C.POINTER(ProblemS), C.c_uint64, C.c_int64, c_int64_p, C.POINTER(C.c_uint32), c_double_p]
        L.emul_sample.restype = C.c_int
        _LIB = L
    return _LIB


def launch_batch(problem, start, direction, elecv, literal=False, use_regs=True):
    st = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, 3)
    di = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    ev = np.ascontiguousarray(elecv, dtype=np.float64).reshape(-1, 3)
    n = st.shape[0]
    rc = np.zeros(n, dtype=np.int32)
    w = np.zeros((n, problem.n_energies))
    ec, ed, ee = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
    ir = np.zeros(n, dtype=np.int64)
    dt = np.zeros(n)
    stats = np.zeros(2, dtype=np.int64)
    r = lib().emul_launch_batch(C.byref(problem.s), int(literal), int(use_regs), n, dptr(st), dptr(di), dptr(ev),
                                rc.ctypes.data_as(C.POINTER(C.c_int32)), dptr(w), dptr(ec), dptr(ed), dptr(ee),
                                ir.ctypes.data_as(c_int64_p), dptr(dt), stats.ctypes.data_as(c_int64_p))
    if r:
        raise RuntimeError("emul_launch_batch failed: %d" % r)
    return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir, d_travel=dt,
                fast_nodes=int(stats[0]), events=int(stats[1]))


def transmission(problem, seed, slot0, n_slots, max_attempts=1 << 20, n_threads=0, per_slot=False):
    """Driver loop on the host compile of the device code (single energy): counters, exact fixed-point weight sum, efficiency."""
    cnt = np.zeros(4, dtype=np.int64)
    fixed = np.zeros(2, dtype=np.uint64)
    ps = np.zeros((n_slots, 3)) if per_slot else None
    r = lib().emul_transmission(C.byref(problem.s), seed, slot0, n_slots, max_attempts, n_threads, cnt.ctypes.data_as(c_int64_p),
                                fixed.ctypes.data_as(C.POINTER(C.c_uint64)), dptr(ps) if per_slot else None)
    if r:
        raise RuntimeError("emul_transmission failed: %d" % r)
    exact = int(fixed[0]) + (int(fixed[1]) << 64)
    i_start = int(cnt[0] + cnt[1] + cnt[2])
    return dict(counters=cnt, sumw_exact=exact, i_start=i_start, i_exit=int(cnt[0]), sum_irefl=int(cnt[3]),
                efficiency=(exact / 2.0**62) / i_start if i_start else 0.0, per_slot=ps)


def flight_stats(problem, seed, slot0, n_slots):
    """Histogram of march steps per flight and the kinds of steps (analysis aid, see emul_flight_stats)."""
    hist = np.zeros(256, dtype=np.int64)
    by = np.zeros(10, dtype=np.int64)
    r = lib().emul_flight_stats(C.byref(problem.s), seed, slot0, n_slots, hist.ctypes.data_as(c_int64_p), by.ctypes.data_as(c_int64_p))
    if r:
        raise RuntimeError("emul_flight_stats failed: %d" % r)
    return hist, dict(zip(("adv1", "advL1", "advL2", "failL2", "failL1", "to_event", "first", "flights", "events", "event_misses"), by.tolist()))


def sample(problem, seed, slots, attempts):
    slots = np.ascontiguousarray(slots, dtype=np.int64)
    attempts = np.ascontiguousarray(attempts, dtype=np.uint32)
    out = np.zeros((slots.shape[0], 12))
    r = lib().emul_sample(C.byref(problem.s), seed, slots.shape[0], slots.ctypes.data_as(c_int64_p),
                          attempts.ctypes.data_as(C.POINTER(C.c_uint32)), dptr(out))
    if r:
        raise RuntimeError("emul_sample failed: %d" % r)
    return out


def sort_leak_records(rec):
    """Records [n, 14 + nE] (slot, attempt, seq, kind, ...) -> (ext, int) in the reference's list order: per slot the
    events of the transmitted attempt (the last one) first, then those of earlier attempts in attempt order; attempts
    that appended a VOID record (kind -1) are dropped, as are the events of an attempt later than the slot's last."""
    if rec.shape[0] == 0:
        return rec[:0], rec[:0]
    void = {(r[0], r[1]) for r in rec if r[3] < 0}
    keep = np.array([(r[0], r[1]) not in void and r[3] >= 0 for r in rec], dtype=bool)
    rec = rec[keep]
    if rec.shape[0] == 0:
        return rec, rec
    order = np.lexsort((rec[:, 2], rec[:, 1], rec[:, 0]))
    rec = rec[order]
    return rec[rec[:, 3] == 0], rec[rec[:, 3] == 1]


def launch_leak(problem, start, direction, elecv, literal=False, max_depth=1024, capacity=None):
    st = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, 3)
    di = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    ev = np.ascontiguousarray(elecv, dtype=np.float64).reshape(-1, 3)
    n = st.shape[0]
    ne = problem.n_energies
    capacity = capacity or max(4096, 64 * n)
    rc = np.zeros(n, dtype=np.int32)
    w = np.zeros((n, ne))
    ec, ed, ee = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
    ir = np.zeros(n, dtype=np.int64)
    dt = np.zeros(n)
    rec = np.zeros((capacity, 14 + ne))
    nrec = C.c_int64(0)
    ovf = C.c_int32(0)
    r = lib().emul_launch_leak(C.byref(problem.s), int(literal), n, dptr(st), dptr(di), dptr(ev), max_depth, capacity,
                               rc.ctypes.data_as(C.POINTER(C.c_int32)), dptr(w), dptr(ec), dptr(ed), dptr(ee),
                               ir.ctypes.data_as(c_int64_p), dptr(dt), dptr(rec), C.byref(nrec), C.byref(ovf))
    if r:
        raise RuntimeError("emul_launch_leak failed: %d" % r)
    if nrec.value > capacity:
        raise RuntimeError("leak record buffer too small: %d > %d" % (nrec.value, capacity))
    return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir, d_travel=dt,
                records=rec[:nrec.value].copy(), stack_overflow=bool(ovf.value))


def transmission_leak(problem, seed, slot0, n_slots, max_attempts=1 << 20, max_depth=1024, capacity=None):
    ne = problem.n_energies
    capacity = capacity or max(4096, 64 * n_slots)
    sw = np.zeros(ne)
    cnt = np.zeros(4, dtype=np.int64)
    ew = np.zeros((n_slots, ne))
    rec = np.zeros((capacity, 14 + ne))
    nrec = C.c_int64(0)
    ovf = C.c_int32(0)
    r = lib().emul_transmission_leak(C.byref(problem.s), seed, slot0, n_slots, max_attempts, max_depth, capacity, dptr(sw),
                                     cnt.ctypes.data_as(c_int64_p), dptr(ew), dptr(rec), C.byref(nrec), C.byref(ovf))
    if r:
        raise RuntimeError("emul_transmission_leak failed: %d" % r)
    if nrec.value > capacity:
        raise RuntimeError("leak record buffer too small: %d > %d" % (nrec.value, capacity))
    return dict(sum_weights=sw, counters=cnt, exit_weights=ew, records=rec[:nrec.value].copy(), stack_overflow=bool(ovf.value))
