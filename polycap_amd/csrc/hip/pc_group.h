/* Several devices driven from one process (included by pc_kernels.hip): pc_hip_group_* of include/polycap-hip.h.
 *
 * The reference parallelises the photon loop with an OpenMP team and adds the threads' partial sums in an omp critical
 * (src/polycap-source.c:697-745, 973-980).  Here the team is a set of device contexts.  Slots are independent and keyed by
 * their global index, so the members get contiguous ranges and nothing is exchanged while they trace; at the end their
 * totals -- 6 counters + the exact fixed-point weight sums as four 32-bit limbs per energy -- are summed by one
 * ncclAllReduce (RCCL over xGMI; librccl is bound with dlopen like libhdf5, so the library loads without it) or, when RCCL
 * is absent or a device appears twice in the group, limb by limb on the host.  Integer sums: identical bits both ways. */
#ifndef PC_GROUP_H
#define PC_GROUP_H

#include <dlfcn.h>

typedef void *pc_nccl_comm;
struct pc_rccl_api {
	int probed = 0;
	void *handle = nullptr;
	int (*CommInitAll)(pc_nccl_comm *, int, const int *) = nullptr;
	int (*CommDestroy)(pc_nccl_comm) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, pc_nccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)(void) = nullptr;
	int (*GroupEnd)(void) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};
static pc_rccl_api g_rccl;
static std::once_flag g_rccl_once;
static const int PC_NCCL_INT64 = 4, PC_NCCL_SUM = 0;   /* ncclInt64, ncclSum (rccl.h) */

static bool pc_rccl_available()
{
	std::call_once(g_rccl_once, []() {
		g_rccl.probed = 1;
		const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr };
		for (int i = 0; names[i] && !g_rccl.handle; i++)
			g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if (g_rccl.handle) {
			g_rccl.CommInitAll = (int (*)(pc_nccl_comm *, int, const int *))dlsym(g_rccl.handle, "ncclCommInitAll");
			g_rccl.CommDestroy = (int (*)(pc_nccl_comm))dlsym(g_rccl.handle, "ncclCommDestroy");
			g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, pc_nccl_comm, hipStream_t))dlsym(g_rccl.handle, "ncclAllReduce");
			g_rccl.GroupStart = (int (*)(void))dlsym(g_rccl.handle, "ncclGroupStart");
			g_rccl.GroupEnd = (int (*)(void))dlsym(g_rccl.handle, "ncclGroupEnd");
			g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.handle, "ncclGetErrorString");
			if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
				dlclose(g_rccl.handle);
				g_rccl.handle = nullptr;
			}
		}
	});
	return g_rccl.handle != nullptr;
}

struct pc_hip_group {
	std::vector<pc_hip_ctx *> ctx;
	std::vector<int> devices;
	std::vector<long long> first, count;       /* slot range of every member in the last run */
	bool distinct = true;                       /* no device twice: the members can form one RCCL communicator */
	std::vector<pc_nccl_comm> comms;
	std::vector<long long *> d_vec;             /* per member: packed totals on its device, 6 + 4 n_energies int64 */
	size_t vec_len = 0;
	long long run_slots = 0;
	int keep_images = 0;
	int leak_run = 0;                           /* the last run was a leak run (pc_hip_group_run_leak) */
};

/* totals of one device -> the vector that is all-reduced: counters, then the (lo, hi) sums as 32-bit limbs */
__global__ void pc_pack_totals_kernel(const pc_totals *t, int n_energies, long long *vec)
{
	const int k = blockIdx.x*blockDim.x + threadIdx.x;
	if (k < 6) vec[k] = (long long)t->counters[k];
	if (k < n_energies) {
		const unsigned long long *sw = (const unsigned long long *)(t + 1);
		const unsigned long long lo = sw[2*k], hi = sw[2*k + 1];
		vec[6 + 4*k] = (long long)(lo & 0xffffffffull);
		vec[6 + 4*k + 1] = (long long)(lo >> 32);
		vec[6 + 4*k + 2] = (long long)(hi & 0xffffffffull);
		vec[6 + 4*k + 3] = (long long)(hi >> 32);
	}
}

static void pc_unpack_limbs(const long long *vec, size_t ne, int64_t counters[6], uint64_t *fixed)
{
	for (int k = 0; k < 6; k++) counters[k] = vec[k];
	for (size_t e = 0; e < ne; e++) {
		unsigned __int128 v = 0;
		for (int l = 3; l >= 0; l--) v = (v << 32) + (unsigned __int128)(unsigned long long)vec[6 + 4*e + l];
		fixed[2*e] = (uint64_t)v;
		fixed[2*e + 1] = (uint64_t)(v >> 64);
	}
}

extern "C" {

void pc_hip_group_destroy(pc_hip_group *g)
{
	if (!g) return;
	for (size_t k = 0; k < g->comms.size(); k++)
		if (g->comms[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comms[k]);
	for (size_t k = 0; k < g->d_vec.size(); k++)
		if (g->d_vec[k]) { (void)hipSetDevice(g->devices[k]); (void)hipFree(g->d_vec[k]); }
	for (pc_hip_ctx *c : g->ctx) pc_hip_ctx_destroy(c);
	delete g;
}

int pc_hip_group_create(const pc_hip_problem *problem, int n_devices, const int *devices, pc_hip_group **out)
{
	if (!out) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_create: group must not be NULL");
	*out = nullptr;
	if (n_devices < 1 || n_devices > 64 || !devices) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_create: between 1 and 64 devices");
	pc_hip_group *g = new pc_hip_group();
	for (int k = 0; k < n_devices; k++) {
		pc_hip_ctx *c = nullptr;
		int st = pc_hip_ctx_create(problem, devices[k], &c);
		if (st) { pc_hip_group_destroy(g); return st; }
		g->ctx.push_back(c);
		g->devices.push_back(devices[k]);
		for (int j = 0; j < k; j++) if (devices[j] == devices[k]) g->distinct = false;
	}
	g->first.assign(n_devices, 0);
	g->count.assign(n_devices, 0);
	g->vec_len = 6 + 4*(size_t)problem->n_energies;
	*out = g;
	return PC_HIP_OK;
}

int pc_hip_group_size(const pc_hip_group *g)
{
	return g ? (int)g->ctx.size() : 0;
}

int pc_hip_group_set_option(pc_hip_group *g, const char *name, int64_t value)
{
	if (!g) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_set_option: group must not be NULL");
	for (pc_hip_ctx *c : g->ctx) {
		int st = pc_hip_set_option(c, name, value);
		if (st) return st;
	}
	return PC_HIP_OK;
}

static int pc_group_run_impl(pc_hip_group *g, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images, bool leak)
{
	if (!g) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_run: group must not be NULL");
	if (n_slots < 1) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_run: n_slots must be >= 1");
	g->leak_run = 0;                            /* set once every member's leak run has succeeded */
	const size_t N = g->ctx.size();
	g->run_slots = n_slots;
	g->keep_images = keep_images ? 1 : 0;
	/* One measure of the photons' lifetime decides the kernel for every member: the first member's previous run, or a probe on
	 * it (3 ms, once per group) when the WHOLE run is big -- a member's share of it may well be below the size from which a
	 * context probes by itself (1e7 slots over 8 devices), and eight probes one after the other would serialise the enqueue. */
	{
		pc_hip_ctx *c0 = g->ctx[0];
		if (!leak && c0->producer < 0 && c0->refl_per_launch < 0. && c0->host.pm.n_energies == 1 && n_slots >= 2000000) {
			if (max_attempts < 1) max_attempts = 1;
			int st = pc_probe_lifetime(c0, seed, 0, max_attempts);
			if (st) return st;
		}
		for (size_t k = 1; k < N; k++)
			if (g->ctx[k]->refl_per_launch < 0.) g->ctx[k]->refl_per_launch = c0->refl_per_launch;
	}
	for (size_t k = 0; k < N; k++) {
		/* contiguous ranges that differ by at most one slot */
		const long long lo = (long long)((__int128)n_slots*(long long)k/(long long)N), hi = (long long)((__int128)n_slots*(long long)(k + 1)/(long long)N);
		g->first[k] = lo; g->count[k] = hi - lo;
	}
	/* every member is enqueued from a host thread of its own (hipSetDevice is per thread) */
	std::vector<int> status(N, PC_HIP_OK);
	std::vector<std::string> msg(N);
	auto enqueue = [&](size_t k) {
		if (g->count[k] == 0) return;
		status[k] = leak ? pc_hip_transmission_run_leak(g->ctx[k], seed, g->first[k], g->count[k], max_attempts, keep_images)
		                 : pc_hip_transmission_run(g->ctx[k], seed, g->first[k], g->count[k], max_attempts, keep_images);
		if (status[k]) msg[k] = g_last_error;       /* the error text is per thread */
	};
	{
		std::vector<std::thread> th;
		for (size_t k = 1; k < N; k++) th.emplace_back(enqueue, k);
		enqueue(0);
		for (auto &t : th) t.join();
	}
	for (size_t k = 0; k < N; k++) {
		if (status[k] == PC_HIP_OK) continue;
		/* a member failed: the others are waited for, so that nothing of this run is pending when the error is reported */
		for (size_t j = 0; j < N; j++)
			if (g->ctx[j]->run_pending) (void)pc_hip_transmission_wait(g->ctx[j], nullptr);
		return pc_fail(status[k], msg[k]);
	}
	g->leak_run = leak ? 1 : 0;
	return PC_HIP_OK;
}

int pc_hip_group_run(pc_hip_group *g, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images)
{
	return pc_group_run_impl(g, seed, n_slots, max_attempts, keep_images, false);
}

/* leak_calc = true over the group (reference: the whole OpenMP team traces leak runs too, src/polycap-source.c:744-884, and the
 * threads' event lists are appended one after the other, :925-1032): every member traces its contiguous slot range with its
 * events ordered on its own device; the group's lists are the members' lists in member order, i.e. in slot order. */
int pc_hip_group_run_leak(pc_hip_group *g, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images)
{
	return pc_group_run_impl(g, seed, n_slots, max_attempts, keep_images, true);
}

int pc_hip_group_leak_counts(pc_hip_group *g, int64_t *n_ext, int64_t *n_int)
{
	if (!g) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_counts: group must not be NULL");
	if (!g->leak_run) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_counts: the last run of the group was not a leak run");
	int64_t e = 0, i = 0;
	for (size_t k = 0; k < g->ctx.size(); k++) {
		if (g->count[k] == 0) continue;
		int64_t a = 0, b = 0;
		int st = pc_hip_leak_counts(g->ctx[k], &a, &b);
		if (st) return st;
		e += a; i += b;
	}
	if (n_ext) *n_ext = e;
	if (n_int) *n_int = i;
	return PC_HIP_OK;
}

int pc_hip_group_leak_events(pc_hip_group *g, int kind, int64_t first, int64_t count, double *records)
{
	if (!g || (count > 0 && !records)) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_events: NULL argument");
	if (!g->leak_run) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_events: the last run of the group was not a leak run");
	if (kind != 0 && kind != 1) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_events: kind must be 0 (extleak) or 1 (intleak)");
	if (first < 0 || count < 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_events: range out of bounds");
	const size_t stride = PC_HIP_LEAK_HDR + (size_t)g->ctx[0]->host.pm.n_energies;
	int64_t base = 0, left = count, at = first;
	for (size_t k = 0; k < g->ctx.size() && left > 0; k++) {
		if (g->count[k] == 0) continue;
		int64_t n[2] = {0, 0};
		int st = pc_hip_leak_counts(g->ctx[k], &n[0], &n[1]);
		if (st) return st;
		const int64_t have = n[kind];
		if (at < base + have) {
			const int64_t lo = at - base, take = (have - lo < left) ? have - lo : left;
			st = pc_hip_leak_events(g->ctx[k], kind, lo, take, records);
			if (st) return st;
			records += (size_t)take*stride; at += take; left -= take;
		}
		base += have;
	}
	if (left > 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_leak_events: range out of bounds");
	return PC_HIP_OK;
}

/* kernel that traced member k's share of the last run (pc_hip_last_kernel) */
int pc_hip_group_last_kernel(pc_hip_group *g, int k)
{
	return (g && k >= 0 && (size_t)k < g->ctx.size()) ? g->ctx[(size_t)k]->last_kernel : -1;
}

int pc_hip_group_images(pc_hip_group *g, const pc_hip_images *dst)
{
	if (!g || !dst) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_images: NULL argument");
	if (!g->keep_images) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_images: the last run kept no images");
	const size_t N = g->ctx.size();
	std::vector<int> status(N, PC_HIP_OK);
	std::vector<std::string> msg(N);
	const size_t ne = (size_t)g->ctx[0]->host.pm.n_energies;
	auto fetch = [&](size_t k) {
		if (g->count[k] == 0) return;
		const size_t o = (size_t)g->first[k];
		pc_hip_images d = *dst;
		for (int j = 0; j < 2; j++) {
			if (d.src_start_coords[j]) d.src_start_coords[j] += o;
			if (d.pc_start_coords[j]) d.pc_start_coords[j] += o;
			if (d.pc_start_dir[j]) d.pc_start_dir[j] += o;
			if (d.pc_start_elecv[j]) d.pc_start_elecv[j] += o;
			if (d.pc_exit_dir[j]) d.pc_exit_dir[j] += o;
			if (d.pc_exit_elecv[j]) d.pc_exit_elecv[j] += o;
		}
		for (int j = 0; j < 3; j++) if (d.pc_exit_coords[j]) d.pc_exit_coords[j] += o;
		if (d.pc_exit_nrefl) d.pc_exit_nrefl += o;
		if (d.pc_exit_dtravel) d.pc_exit_dtravel += o;
		if (d.exit_coord_weights) d.exit_coord_weights += o*ne;
		status[k] = pc_hip_transmission_images(g->ctx[k], 0, g->count[k], &d);
		if (status[k]) msg[k] = g_last_error;       /* the error text is per thread */
	};
	/* The destination planes are pinned here, once and whole: the members' sub-ranges share pages at their boundaries, and
	 * a member that pinned and unpinned its own range would unpin a neighbour's first page under its running copy (the members
	 * find their ranges pinned already and leave them alone). */
	std::vector<void *> pinned;
	if (N > 1) {
		void *planes[PC_N_FIELDS + 1] = {
			dst->src_start_coords[0], dst->src_start_coords[1], dst->pc_start_coords[0], dst->pc_start_coords[1],
			dst->pc_start_dir[0], dst->pc_start_dir[1], dst->pc_start_elecv[0], dst->pc_start_elecv[1],
			dst->pc_exit_coords[0], dst->pc_exit_coords[1], dst->pc_exit_coords[2],
			dst->pc_exit_dir[0], dst->pc_exit_dir[1], dst->pc_exit_elecv[0], dst->pc_exit_elecv[1],
			dst->pc_exit_nrefl, dst->pc_exit_dtravel, dst->exit_coord_weights };
		(void)hipSetDevice(g->devices[0]);
		std::vector<std::pair<char *, size_t>> ranges;
		for (int f = 0; f <= PC_N_FIELDS; f++)
			if (planes[f]) ranges.emplace_back((char *)planes[f], (size_t)g->run_slots*sizeof(double)*(f < PC_N_FIELDS ? 1 : ne));
		/* pinned or not, the members pin nothing themselves: they would pin neighbouring pieces of the same pages */
		(void)pc_pin_ranges(ranges, hipHostRegisterPortable, pinned);
		for (pc_hip_ctx *c : g->ctx) c->dst_prepinned = 1;
	}
	std::vector<std::thread> th;
	for (size_t k = 1; k < N; k++) th.emplace_back(fetch, k);
	fetch(0);
	for (auto &t : th) t.join();
	for (pc_hip_ctx *c : g->ctx) c->dst_prepinned = 0;
	for (void *p : pinned) (void)hipHostUnregister(p);
	for (size_t k = 0; k < N; k++)
		if (status[k]) return pc_fail(status[k], msg[k]);
	return PC_HIP_OK;
}

int pc_hip_group_totals(pc_hip_group *g, int reduce, double *sum_weights, int64_t counters[6], uint64_t *sumw_fixed,
                        int *reduced_by, float *kernel_ms)
{
	if (!g || !counters) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_totals: NULL argument");
	const size_t N = g->ctx.size(), ne = (size_t)g->ctx[0]->host.pm.n_energies, len = g->vec_len;
	float ms_max = 0.f;
	for (size_t k = 0; k < N; k++) {
		if (g->count[k] == 0) continue;
		float ms = 0.f;
		int st = pc_hip_transmission_wait(g->ctx[k], &ms);
		if (st) return st;
		if (ms > ms_max) ms_max = ms;
	}
	if (kernel_ms) *kernel_ms = ms_max;
	bool use_rccl = (reduce != 0) && g->distinct && pc_rccl_available();
	if (reduce == 1 && !use_rccl)
		return pc_fail(PC_HIP_ERR_RUNTIME, g->distinct ? "pc_hip_group_totals: librccl could not be loaded"
		                                               : "pc_hip_group_totals: a device appears twice in the group, RCCL needs distinct devices");
	std::vector<long long> total(len, 0);
	if (use_rccl) {
		if (g->comms.empty()) {
			g->comms.assign(N, nullptr);
			int rc = g_rccl.CommInitAll(g->comms.data(), (int)N, g->devices.data());
			if (rc != 0) {
				g->comms.clear();
				if (reduce == 1) return pc_fail(PC_HIP_ERR_RUNTIME, std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed"));
				use_rccl = false;
			}
		}
	}
	if (use_rccl) {
		if (g->d_vec.empty()) {
			g->d_vec.assign(N, nullptr);
			for (size_t k = 0; k < N; k++) {
				PC_HIP_CHECK(hipSetDevice(g->devices[k]));
				PC_HIP_CHECK(hipMalloc(&g->d_vec[k], len*sizeof(long long)));
			}
		}
		const int threads = 64, blocks = (int)((std::max<size_t>(ne, 6) + threads - 1)/threads);
		for (size_t k = 0; k < N; k++) {
			PC_HIP_CHECK(hipSetDevice(g->devices[k]));
			if (g->count[k] == 0) {
				PC_HIP_CHECK(hipMemsetAsync(g->d_vec[k], 0, len*sizeof(long long), g->ctx[k]->stream));
			} else {
				hipLaunchKernelGGL(pc_pack_totals_kernel, dim3(blocks), dim3(threads), 0, g->ctx[k]->stream, g->ctx[k]->d_totals, (int)ne, g->d_vec[k]);
				PC_HIP_CHECK(hipGetLastError());
			}
		}
		int rc = g_rccl.GroupStart();
		for (size_t k = 0; k < N && rc == 0; k++)
			rc = g_rccl.AllReduce(g->d_vec[k], g->d_vec[k], len, PC_NCCL_INT64, PC_NCCL_SUM, g->comms[k], g->ctx[k]->stream);
		const int rc_end = g_rccl.GroupEnd();
		if (rc == 0) rc = rc_end;
		if (rc != 0) return pc_fail(PC_HIP_ERR_RUNTIME, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed"));
		for (size_t k = 0; k < N; k++) {
			PC_HIP_CHECK(hipSetDevice(g->devices[k]));
			PC_HIP_CHECK(hipStreamSynchronize(g->ctx[k]->stream));
		}
		PC_HIP_CHECK(hipSetDevice(g->devices[0]));
		PC_HIP_CHECK(hipMemcpy(total.data(), g->d_vec[0], len*sizeof(long long), hipMemcpyDeviceToHost));
	} else {
		/* the same sum on the host: limb by limb, so that it cannot differ from the all-reduce */
		std::vector<int64_t> c(6);
		std::vector<uint64_t> fx(2*ne);
		for (size_t k = 0; k < N; k++) {
			if (g->count[k] == 0) continue;
			int st = pc_hip_transmission_totals(g->ctx[k], nullptr, c.data(), fx.data());
			if (st != PC_HIP_OK && st != PC_HIP_ERR_ATTEMPTS) return st;
			for (int j = 0; j < 6; j++) total[j] += c[j];
			for (size_t e = 0; e < ne; e++) {
				total[6 + 4*e] += (long long)(fx[2*e] & 0xffffffffull);
				total[6 + 4*e + 1] += (long long)(fx[2*e] >> 32);
				total[6 + 4*e + 2] += (long long)(fx[2*e + 1] & 0xffffffffull);
				total[6 + 4*e + 3] += (long long)(fx[2*e + 1] >> 32);
			}
		}
	}
	if (reduced_by) *reduced_by = use_rccl ? 1 : 0;
	std::vector<uint64_t> fixed(2*ne);
	pc_unpack_limbs(total.data(), ne, counters, fixed.data());
	for (size_t e = 0; e < ne; e++) {
		if (sum_weights) sum_weights[e] = pc_hip_fixed_to_double(fixed[2*e], fixed[2*e + 1]);
		if (sumw_fixed) { sumw_fixed[2*e] = fixed[2*e]; sumw_fixed[2*e + 1] = fixed[2*e + 1]; }
	}
	if (counters[4] != 0)
		return pc_fail(PC_HIP_ERR_ATTEMPTS, "pc_hip_group_totals: some slots exhausted max_attempts without a transmitted photon");
	return PC_HIP_OK;
}

} /* extern "C" */

#endif /* PC_GROUP_H */
