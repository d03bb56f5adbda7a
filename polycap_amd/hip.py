"""Python front-end of the thin HIP C-ABI (include/polycap-hip.h): numpy in, numpy out.

TraceContext wraps one pc_hip_ctx (problem resident on one GPU).  Everything here calls into
libpolycap.so; nothing is computed in Python and nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _cabi
from ._cabi import Problem, dptr, c_int64_p


class HipError(RuntimeError):
    def __init__(self, where, status):
        msg = _cabi.lib().pc_hip_last_error()
        super().__init__("%s failed (%d): %s" % (where, status, msg.decode() if msg else ""))
        self.status = status


def device_count():
    return int(_cabi.lib().pc_hip_device_count())


IMG_FIELDS = ("src_start_x", "src_start_y", "pc_start_x", "pc_start_y", "pc_start_dir_x", "pc_start_dir_y",
              "pc_start_elecv_x", "pc_start_elecv_y", "pc_exit_x", "pc_exit_y", "pc_exit_z",
              "pc_exit_dir_x", "pc_exit_dir_y", "pc_exit_elecv_x", "pc_exit_elecv_y", "nrefl", "dtravel")


class TraceContext:
    """One problem (optic + glass + energies + source) uploaded to one MI355X."""

    def __init__(self, problem, device=0):
        if not isinstance(problem, Problem):
            raise TypeError("problem must be a polycap_amd.Problem")
        self.problem = problem
        self._L = _cabi.lib()
        h = C.c_void_p()
        st = self._L.pc_hip_ctx_create(C.byref(problem.s), int(device), C.byref(h))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_ctx_create", st)
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pc_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name, value):
        st = self._L.pc_hip_set_option(self._h, name.encode(), int(value))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_set_option", st)

    def device_synchronize(self):
        """hipDeviceSynchronize on the context's device (every stream)."""
        st = self._L.pc_hip_device_synchronize(self._h)
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_device_synchronize", st)

    # -- polycap_photon_launch for a batch of explicit photons
    def launch_photons(self, start, direction, elecv, leak_calc=False):
        """leak_calc=True: polycap_photon_launch(..., leak_calc=true); the events are then available from leaks()"""
        st_ = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, 3)
        di = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        ev = np.ascontiguousarray(elecv, dtype=np.float64).reshape(-1, 3)
        n = st_.shape[0]
        ne = self.problem.n_energies
        rc = np.zeros(n, dtype=np.int32)
        w = np.zeros((n, ne))
        ec, ed, ee = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        ir = np.zeros(n, dtype=np.int64)
        dt = np.zeros(n)
        fn = self._L.pc_hip_launch_photons_leak if leak_calc else self._L.pc_hip_launch_photons
        st = fn(self._h, n, dptr(st_), dptr(di), dptr(ev),
                rc.ctypes.data_as(C.POINTER(C.c_int32)), dptr(w), dptr(ec), dptr(ed), dptr(ee),
                ir.ctypes.data_as(c_int64_p), dptr(dt))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_launch_photons_leak" if leak_calc else "pc_hip_launch_photons", st)
        return dict(rc=rc, weights=w, exit_coords=ec, exit_dir=ed, exit_elecv=ee, i_refl=ir, d_travel=dt)

    # -- polycap_source_get_photon on the device
    def sample_photons(self, seed, slots, attempts=None):
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        n = slots.shape[0]
        attempts = np.zeros(n, dtype=np.uint32) if attempts is None else np.ascontiguousarray(attempts, dtype=np.uint32)
        out = np.zeros((n, 12))
        st = self._L.pc_hip_sample_photons(self._h, int(seed), n, slots.ctypes.data_as(c_int64_p),
                                           attempts.ctypes.data_as(C.POINTER(C.c_uint32)), dptr(out))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_sample_photons", st)
        return out

    # -- polycap_source_get_transmission_efficiencies for a slot range
    def run(self, seed, slot0, n_slots, max_attempts=1 << 20, keep_images=False, leak_calc=False):
        fn = self._L.pc_hip_transmission_run_leak if leak_calc else self._L.pc_hip_transmission_run
        st = fn(self._h, int(seed), int(slot0), int(n_slots), int(max_attempts), int(bool(keep_images)))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_transmission_run_leak" if leak_calc else "pc_hip_transmission_run", st)
        self._last_n = int(n_slots)

    def leaks(self, copy=True):
        """(ext, int): the leak events of the last leak_calc run, arrays [n, 12 + nE] with columns slot, attempt,
        x, y, z, dir x y z, elecv x y z, n_refl, weights; in the reference's list order (the device puts them into it).
        copy=False: read-only views of the context's own pinned lists, valid until its next leak run."""
        out = []
        stride = _cabi.PC_HIP_LEAK_HDR + self.problem.n_energies
        for kind in (0, 1):
            ptr = C.POINTER(C.c_double)()
            n = C.c_int64(0)
            st = self._L.pc_hip_leak_events_view(self._h, kind, C.byref(ptr), C.byref(n))
            if st != _cabi.PC_HIP_OK:
                raise HipError("pc_hip_leak_events_view", st)
            if n.value == 0:
                out.append(np.zeros((0, stride)))
                continue
            a = np.ctypeslib.as_array(ptr, shape=(n.value, stride))
            if copy:
                a = a.copy()
            else:
                a.flags.writeable = False
            out.append(a)
        return out[0], out[1]

    def wait(self):
        ms = C.c_float(0)
        st = self._L.pc_hip_transmission_wait(self._h, C.byref(ms))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_transmission_wait", st)
        return float(ms.value)

    def totals(self, check=True):
        ne = self.problem.n_energies
        sw = np.zeros(ne)
        cnt = np.zeros(6, dtype=np.int64)
        fx = np.zeros(2 * ne, dtype=np.uint64)
        st = self._L.pc_hip_transmission_totals(self._h, dptr(sw), cnt.ctypes.data_as(c_int64_p),
                                                fx.ctypes.data_as(C.POINTER(C.c_uint64)))
        if st != _cabi.PC_HIP_OK and (check or st != _cabi.PC_HIP_ERR_ATTEMPTS):
            raise HipError("pc_hip_transmission_totals", st)
        return dict(sum_weights=sw, counters=cnt, sumw_fixed=fx.reshape(ne, 2),
                    i_exit=int(cnt[0]), not_entered=int(cnt[1]), not_transmitted=int(cnt[2]), sum_irefl=int(cnt[3]),
                    failed_slots=int(cnt[4]), launches=int(cnt[5]), i_start=int(cnt[0] + cnt[1] + cnt[2]))

    KERNELS = {0: "pc_trace_kernel", 1: "pc_trace_pool_kernel", 2: "pc_trace_producer_kernel", 3: "pc_trace_wave_kernel",
               4: "pc_trace_log_kernel", 5: "pc_leak_kernel"}

    def last_kernel(self):
        """Name of the kernel that traced the last source run (None before the first)."""
        return self.KERNELS.get(int(self._L.pc_hip_last_kernel(self._h)))

    def phase_stats(self):
        """Average active lanes per scheduler phase of the last run (diagnostics)."""
        st = np.zeros(6, dtype=np.int64)
        rc = self._L.pc_hip_phase_stats(self._h, st.ctypes.data_as(c_int64_p))
        if rc != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_phase_stats", rc)
        names = ("march", "event", "new")
        return {n: dict(phases=int(st[2 * i]), lanes=int(st[2 * i + 1]),
                        avg_lanes=float(st[2 * i + 1]) / max(1, int(st[2 * i]))) for i, n in enumerate(names)}

    def sweep_stats(self):
        """Weight sweeps of the last run of the logging many-energy kernel: wave-level passes and (pass, reflection) iterations,
        the host's tameness threshold and the proxy energies."""
        st = np.zeros(4, dtype=np.int64)
        ct = C.c_double(0.)
        pr = (C.c_int * 2)(-1, -1)
        rc = self._L.pc_hip_sweep_stats(self._h, st.ctypes.data_as(c_int64_p), C.byref(ct), pr)
        if rc != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_sweep_stats", rc)
        return dict(passes=int(st[0]), iterations=int(st[1]), ct_tame=float(ct.value), proxies=[int(pr[0]), int(pr[1])],
                    wave_life_sum=int(st[2]), wave_life_max=int(st[3]))

    def images(self, first=0, count=None):
        """Image data of slots [first, first+count) of the last run: images [count, 17] (the planes of pc_hip_images in
        their order, one row per slot, the reflection count as a float in column 15), exit_weights [count, nE], nrefl.
        One row per slot is also how the device keeps them, so the records are fetched as they are."""
        count = self._last_n - first if count is None else count
        ne = self.problem.n_energies
        rec = np.empty((count, 17 + ne))
        st = self._L.pc_hip_transmission_records(self._h, int(first), int(count), dptr(rec))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_transmission_records", st)
        nrefl = rec[:, 15].view(np.int64).copy() if count else np.zeros(0, dtype=np.int64)
        rec[:, 15] = nrefl
        return dict(images=rec[:, :17], exit_weights=rec[:, 17:], nrefl=nrefl)

    def image_planes(self, first=0, count=None):
        """The same through pc_hip_transmission_images: SoA planes [17, count] as the reference's struct _polycap_images
        holds them (what the C host layer uses), exit_weights [count, nE], nrefl."""
        count = self._last_n - first if count is None else count
        ne = self.problem.n_energies
        planes = np.zeros((17, count))
        nrefl = np.zeros(count, dtype=np.int64)
        w = np.zeros((count, ne))
        s = _cabi.images_struct(planes, nrefl, w)
        st = self._L.pc_hip_transmission_images(self._h, int(first), int(count), C.byref(s))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_transmission_images", st)
        planes[15] = nrefl
        return dict(planes=planes, exit_weights=w, nrefl=nrefl)

    def slot_ids(self, first=0, count=None):
        """Slot of the photon at positions [first, first+count) of the image planes of the last run: the identity, except
        after a compact run (options compact_images + slot_ids), whose planes are in the order of completion."""
        count = self._last_n - first if count is None else count
        ids = np.zeros(count, dtype=np.int64)
        st = self._L.pc_hip_transmission_slot_ids(self._h, int(first), int(count), ids.ctypes.data_as(c_int64_p))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_transmission_slot_ids", st)
        return ids

    def leak_set_order(self, order, n_heavy=0):
        """Order in which the next leak_calc source runs of len(order) slots hand out their slots (heaviest first); the first
        n_heavy go to the heavy lanes.  An empty order restores slot order."""
        o = np.ascontiguousarray(order, dtype=np.uint32)
        st = self._L.pc_hip_leak_set_order(self._h, o.ctypes.data_as(C.POINTER(C.c_uint32)), int(o.size), int(n_heavy))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_leak_set_order", st)

    def leak_slot_units(self, first=0, count=None):
        """Units of work per slot of the last leak_calc source run (option leak_slot_units = 1)."""
        count = self._last_n - first if count is None else count
        u = np.zeros(count, dtype=np.uint32)
        st = self._L.pc_hip_leak_slot_units(self._h, int(first), int(count), u.ctypes.data_as(C.POINTER(C.c_uint32)))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_leak_slot_units", st)
        return u

    def transmission(self, seed, slot0, n_slots, max_attempts=1 << 20, keep_images=False, leak_calc=False, leak_views=False):
        """run + wait + totals (+ images, + leak events: copies, or with leak_views views of the context's lists) in one call."""
        self.run(seed, slot0, n_slots, max_attempts, keep_images, leak_calc)
        ms = self.wait()
        r = self.totals()
        r["kernel_ms"] = ms
        r["efficiencies"] = efficiencies(r["sum_weights"], r["counters"])
        if keep_images:
            r.update(self.images(0, n_slots))
        if leak_calc:
            r["ext"], r["int"] = self.leaks(copy=not leak_views)
        return r


class TraceGroup:
    """One problem on several devices driven from this process (pc_hip_group_*): contiguous slot ranges per member, one
    RCCL all-reduce (or the identical host sum) of the totals.  `devices` may repeat an index."""

    def __init__(self, problem, devices):
        self.problem = problem
        self._L = _cabi.lib()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        st = self._L.pc_hip_group_create(C.byref(problem.s), len(devices), devs, C.byref(h))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_group_create", st)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.pc_hip_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name, value):
        st = self._L.pc_hip_group_set_option(self._h, name.encode(), int(value))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_group_set_option", st)

    def last_kernels(self):
        """Names of the kernels that traced the members' shares of the last run."""
        n = int(self._L.pc_hip_group_size(self._h))
        return [TraceContext.KERNELS.get(int(self._L.pc_hip_group_last_kernel(self._h, k))) for k in range(n)]

    def transmission(self, seed, n_slots, max_attempts=1 << 20, keep_images=False, reduce=-1):
        """reduce: -1 automatic (RCCL when the devices are distinct and librccl loads), 0 host sum, 1 RCCL or fail"""
        import time
        t0 = time.perf_counter()
        st = self._L.pc_hip_group_run(self._h, int(seed), int(n_slots), int(max_attempts), int(bool(keep_images)))
        self.enqueue_s = time.perf_counter() - t0          # pc_hip_group_run only enqueues: the members trace asynchronously
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_group_run", st)
        ne = self.problem.n_energies
        r = {}
        if keep_images:
            planes = np.zeros((17, n_slots))
            nrefl = np.zeros(n_slots, dtype=np.int64)
            w = np.zeros((n_slots, ne))
            s = _cabi.images_struct(planes, nrefl, w)
            st = self._L.pc_hip_group_images(self._h, C.byref(s))
            if st != _cabi.PC_HIP_OK:
                raise HipError("pc_hip_group_images", st)
            planes[15] = nrefl
            r.update(images=planes.T.copy(), exit_weights=w, nrefl=nrefl)
        sw = np.zeros(ne)
        cnt = np.zeros(6, dtype=np.int64)
        fx = np.zeros(2 * ne, dtype=np.uint64)
        by, ms = C.c_int(0), C.c_float(0)
        st = self._L.pc_hip_group_totals(self._h, int(reduce), dptr(sw), cnt.ctypes.data_as(c_int64_p),
                                         fx.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(by), C.byref(ms))
        if st != _cabi.PC_HIP_OK:
            raise HipError("pc_hip_group_totals", st)
        r.update(sum_weights=sw, counters=cnt, sumw_fixed=fx.reshape(ne, 2), reduced_by_rccl=bool(by.value), kernel_ms=float(ms.value),
                 i_exit=int(cnt[0]), i_start=int(cnt[0] + cnt[1] + cnt[2]), efficiencies=efficiencies(sw, cnt))
        return r


def efficiencies(sum_weights, counters):
    sw = np.ascontiguousarray(sum_weights, dtype=np.float64)
    cnt = np.ascontiguousarray(counters, dtype=np.int64)
    if cnt.shape[0] < 6:
        cnt = np.concatenate([cnt, np.zeros(6 - cnt.shape[0], dtype=np.int64)])
    eff = np.zeros_like(sw)
    _cabi.lib().pc_hip_efficiencies(sw.shape[0], dptr(sw), cnt.ctypes.data_as(c_int64_p), dptr(eff))
    return eff


def fixed_to_double(lo, hi):
    return float(_cabi.lib().pc_hip_fixed_to_double(int(lo), int(hi)))
