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
static const int PC_NCCL_INT64 = 4, PC_NCCL_SUM = 0;   /* ncclInt64, ncclSum (rccl.h) */

static bool pc_rccl_available()
{
	if (!g_rccl.probed) {
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
	}
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

int pc_hip_group_run(pc_hip_group *g, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images)
{
	if (!g) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_run: group must not be NULL");
	if (n_slots < 1) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_group_run: n_slots must be >= 1");
	const long long N = (long long)g->ctx.size();
	g->run_slots = n_slots;
	g->keep_images = keep_images ? 1 : 0;
	for (long long k = 0; k < N; k++) {
		/* contiguous ranges that differ by at most one slot */
		const long long lo = (long long)((__int128)n_slots*k/N), hi = (long long)((__int128)n_slots*(k + 1)/N);
		g->first[k] = lo; g->count[k] = hi - lo;
		if (hi == lo) continue;
		int st = pc_hip_transmission_run(g->ctx[k], seed, lo, hi - lo, max_attempts, keep_images);
		if (st) return st;
	}
	return PC_HIP_OK;
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
	std::vector<std::thread> th;
	for (size_t k = 1; k < N; k++) th.emplace_back(fetch, k);
	fetch(0);
	for (auto &t : th) t.join();
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
