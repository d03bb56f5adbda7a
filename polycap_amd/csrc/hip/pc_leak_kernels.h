/*
 * pc_leak_kernels.h -- leak_calc=true on the device: kernel + host code, included by pc_kernels.hip after the
 * context definition (same translation unit, so the two modes share tables, totals and image records).
 *
 * Shape.  The leak run of one launched photon (pc_leak.h) is a long, irregular, strictly sequential job: thousands
 * of cap/10 steps through the glass per reflection, a depth-first tree of leaked fractions, every decision discrete.
 * So this kernel does not schedule phases across a wave like pc_trace_kernel; every lane owns one exit-photon slot at
 * a time (taken from the same global counter), runs it to the end and appends its leak events to the record buffer.
 * What keeps the machine busy is the number of independent lanes, bounded only by the HBM given to the per-lane stacks
 * (max_depth x (24 + n_energies) doubles each).  All seven profile tables sit in LDS, including ext, which the wall
 * search reads at every step.
 */
#ifndef PC_LEAK_KERNELS_H
#define PC_LEAK_KERNELS_H

#include <algorithm>
#include <unordered_set>

#define PC_LEAK_BLOCK 256

struct pc_leak_kargs {
	const double *amu;             /* [n_energies] */
	double *frames;                /* total_threads x max_depth x (PC_LF_HDR + n_energies) */
	int max_depth;
	double *records;               /* capacity x (PC_LR_HDR + n_energies) */
	unsigned long long *cursor;    /* [0] records appended, [1] lanes whose stack overflowed */
	long long capacity;
	unsigned int *final_attempt;   /* [n_slots]: attempt index of the transmitted photon of each slot (driver mode) */
};

template <int MODE, int PITCH>
__global__ void __launch_bounds__(PC_LEAK_BLOCK)
pc_leak_kernel(pc_kargs a, pc_leak_kargs lk)
{
	constexpr bool EXPLICIT = (MODE == PC_MODE_EXPLICIT);
	__shared__ double lds[7*PITCH];
	__shared__ float ldsf[4*PITCH];
	const int npts = a.pm.nmax + 1;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		lds[k] = a.g_z[k];
		lds[PITCH + k] = a.g_cap[k];
		lds[2*PITCH + k] = a.g_zh[k];
		lds[3*PITCH + k] = a.g_cap2[k];
		lds[4*PITCH + k] = a.g_hexd[k];
		lds[5*PITCH + k] = a.g_idz[k];
		lds[6*PITCH + k] = a.g_ext[k];
		ldsf[k] = a.g_mb1[k]; ldsf[PITCH + k] = a.g_md1[k]; ldsf[2*PITCH + k] = a.g_mb2[k]; ldsf[3*PITCH + k] = a.g_md2[k];
	}
	__syncthreads();
	pc_tables T;
	T.z = lds; T.cap = lds + PITCH; T.zh = lds + 2*PITCH; T.cap2 = lds + 3*PITCH; T.hexd = lds + 4*PITCH; T.idz = lds + 5*PITCH;
	T.ext = lds + 6*PITCH;
	T.mb1 = ldsf; T.md1 = ldsf + PITCH; T.mb2 = ldsf + 2*PITCH; T.md2 = ldsf + 3*PITCH;
	const pc_params &Pm = a.pm;
	const int ne = Pm.n_energies;
	const long long rec = PC_N_FIELDS + (long long)ne;
	const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;

	pc_leak_ctx cx;
	cx.ec = a.ec; cx.amu = lk.amu; cx.ne = ne;
	cx.frames = lk.frames + gtid * (long long)lk.max_depth * (PC_LF_HDR + ne);
	cx.max_depth = lk.max_depth;
	cx.sink.records = lk.records; cx.sink.cursor = lk.cursor; cx.sink.capacity = lk.capacity;
	cx.stack_overflow = 0;
	const double *w0 = cx.frames + PC_LF_HDR;      /* weights of the launched photon */

	unsigned int n_exit = 0, n_not_entered = 0, n_not_trans = 0, n_failed = 0, n_launch = 0;
	unsigned long long s_irefl = 0;

	for (;;) {
		const long long slot = (long long)atomicAdd(&a.totals->next_slot, 1ull);
		if (slot >= a.n_slots) break;
		pc_photon<0> ph;
		ph.wmem = nullptr; ph.wstride = 1; ph.wset = 0; ph.rc = 0;
		if (EXPLICIT) {
			const long long j = slot;
			n_launch++;
			const double z0 = a.in_start[3*j+2];
			const int st = pc_launch_init(T, Pm, ph, a.in_start[3*j], a.in_start[3*j+1], z0, a.in_dir[3*j], a.in_dir[3*j+1], a.in_dir[3*j+2],
			                              a.in_elecv[3*j], a.in_elecv[3*j+1], a.in_elecv[3*j+2]);
			cx.slot = (double)j; cx.attempt = 0.;
			const int rc = pc_leak_launch(T, Pm, cx, ph, st, z0);
			a.out_rc[j] = rc;
			for (int e = 0; e < ne; e++) a.out_weights[j*ne + e] = w0[e];
			a.out_exit_coords[3*j] = ph.Px; a.out_exit_coords[3*j+1] = ph.Py; a.out_exit_coords[3*j+2] = ph.Pz;
			a.out_exit_dir[3*j] = ph.dx; a.out_exit_dir[3*j+1] = ph.dy; a.out_exit_dir[3*j+2] = ph.dz;
			a.out_exit_elecv[3*j] = ph.ex; a.out_exit_elecv[3*j+1] = ph.ey; a.out_exit_elecv[3*j+2] = ph.ez;
			a.out_irefl[j] = ph.irefl;
			a.out_dtravel[j] = ph.dtravel;
			continue;
		}
		/* src/polycap-source.c:744-884, leak_calc=true */
		unsigned int attempt = 0;
		int ok = 0;
		for (; attempt < a.max_attempts; attempt++) {
			n_launch++;
			pc_start s;
			pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), attempt, s);
			const int st = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
			const double cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
			cx.slot = (double)(a.slot0 + slot); cx.attempt = (double)attempt;
			const int rc = pc_leak_launch(T, Pm, cx, ph, st, s.z);
			if (rc == 0) n_not_trans++;
			else if (rc == 2) n_not_entered++;
			else if (rc == 1) ok = pc_in_exit_window(Pm, ph);
			if (!ok) continue;
			n_exit++;
			s_irefl += (unsigned long long)ph.irefl;
			for (int e = 0; e < ne; e++) {
				const double w = w0[e];
				pc_atomic_add128(a.sumw + 2*e, (unsigned long long)(w * PC_FIX_SCALE), 0ull);
				if (a.keep_images) a.img[slot*rec + PC_F_WEIGHTS + e] = w;
			}
			if (a.keep_images) {
				/* src/polycap-source.c:779-798, 900-923 */
				double *r = a.img + slot*rec;
				const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
				r[PC_F_SRCX] = s.srcx; r[PC_F_SRCY] = s.srcy;
				r[PC_F_STARTX] = s.x; r[PC_F_STARTY] = s.y;
				r[PC_F_SDIRX] = s.dx; r[PC_F_SDIRY] = s.dy;
				double tx = s.ex*c_ae + s.dx*c_be, ty = s.ey*c_ae + s.dy*c_be, tz = s.ez*c_ae + s.dz*c_be;
				pc_norm3(tx, ty, tz);
				r[PC_F_SEVX] = round(tx); r[PC_F_SEVY] = round(ty);
				const double t = (Pm.z_end - ph.Pz) / ph.dz;
				const double ex = ph.Px + ph.dx*t, ey = ph.Py + ph.dy*t, ez = ph.Pz + ph.dz*t;
				r[PC_F_EXITX] = ex; r[PC_F_EXITY] = ey; r[PC_F_EXITZ] = ez;
				r[PC_F_EDIRX] = ph.dx; r[PC_F_EDIRY] = ph.dy;
				tx = ph.ex*c_ae + ph.dx*c_be; ty = ph.ey*c_ae + ph.dy*c_be; tz = ph.ez*c_ae + ph.dz*c_be;
				pc_norm3(tx, ty, tz);
				r[PC_F_EEVX] = round(tx); r[PC_F_EEVY] = round(ty);
				((long long *)r)[PC_F_NREFL] = ph.irefl;
				const double lx = ex - ph.Px, ly = ey - ph.Py, lz = Pm.z_end - ph.Pz;
				r[PC_F_DTRAVEL] = ph.dtravel + sqrt(lx*lx + ly*ly + lz*lz);
			}
			break;
		}
		lk.final_attempt[slot] = attempt;
		if (!ok) {
			n_failed++;
			if (a.keep_images)
				for (int e = 0; e < ne; e++) a.img[slot*rec + PC_F_WEIGHTS + e] = 0.;
		}
	}

	if (cx.stack_overflow) atomicAdd(&lk.cursor[1], 1ull);
	if (!EXPLICIT) {
		const int lane = threadIdx.x & (PC_WAVE - 1);
		unsigned long long v0 = pc_wave_sum_u64(n_exit), v1 = pc_wave_sum_u64(n_not_entered), v2 = pc_wave_sum_u64(n_not_trans);
		unsigned long long v3 = pc_wave_sum_u64(s_irefl), v4 = pc_wave_sum_u64(n_failed), v5 = pc_wave_sum_u64(n_launch);
		if (lane == 0) {
			atomicAdd(&a.totals->counters[0], v0);
			atomicAdd(&a.totals->counters[1], v1);
			atomicAdd(&a.totals->counters[2], v2);
			atomicAdd(&a.totals->counters[3], v3);
			if (v4) atomicAdd(&a.totals->counters[4], v4);
			atomicAdd(&a.totals->counters[5], v5);
		}
	}
}

/* =========================================================================== host side */

/* grid of the leak kernel: as many lanes as the stack budget allows, at most two resident blocks per CU */
static int pc_leak_grid(pc_hip_ctx *ctx, long long n_items, long long &lanes)
{
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	const size_t per_lane = (size_t)ctx->leak_max_depth * (PC_LF_HDR + ne) * sizeof(double);
	long long by_mem = (long long)(ctx->leak_stack_bytes / per_lane);
	long long want = (n_items + PC_LEAK_BLOCK - 1) / PC_LEAK_BLOCK;
	long long max_blocks = (long long)ctx->n_cu * 2;
	long long blocks = std::min(want, max_blocks);
	blocks = std::min(blocks, std::max(1ll, by_mem / PC_LEAK_BLOCK));
	if (blocks < 1) blocks = 1;
	lanes = blocks * PC_LEAK_BLOCK;
	return (int)blocks;
}

static int pc_leak_buffers(pc_hip_ctx *ctx, long long lanes, long long capacity, long long n_slots)
{
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	const size_t frames = (size_t)lanes * (size_t)ctx->leak_max_depth * (PC_LF_HDR + ne);
	if (frames > ctx->leak_frames_elems) {
		if (ctx->d_leak_frames) PC_HIP_CHECK(hipFree(ctx->d_leak_frames));
		ctx->d_leak_frames = nullptr; ctx->leak_frames_elems = 0;
		if (hipMalloc(&ctx->d_leak_frames, frames*sizeof(double)) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the per-lane stacks (lower leak_stack_mb or leak_max_depth)");
		ctx->leak_frames_elems = frames;
	}
	const size_t recs = (size_t)capacity * (PC_LR_HDR + ne);
	if (recs > ctx->leak_records_elems) {
		if (ctx->d_leak_records) PC_HIP_CHECK(hipFree(ctx->d_leak_records));
		ctx->d_leak_records = nullptr; ctx->leak_records_elems = 0;
		if (hipMalloc(&ctx->d_leak_records, recs*sizeof(double)) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the leak record buffer");
		ctx->leak_records_elems = recs;
	}
	if (!ctx->d_leak_cursor) {
		if (hipMalloc(&ctx->d_leak_cursor, 2*sizeof(unsigned long long)) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the record cursor");
	}
	if (!ctx->d_amu) {
		if (hipMalloc(&ctx->d_amu, ne*sizeof(double)) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the attenuation table");
		PC_HIP_CHECK(hipMemcpy(ctx->d_amu, ctx->host.amu.data(), ne*sizeof(double), hipMemcpyHostToDevice));
	}
	if (n_slots > ctx->leak_attempt_slots) {
		if (ctx->d_leak_attempts) PC_HIP_CHECK(hipFree(ctx->d_leak_attempts));
		ctx->d_leak_attempts = nullptr; ctx->leak_attempt_slots = 0;
		if (hipMalloc(&ctx->d_leak_attempts, (size_t)n_slots*sizeof(unsigned int)) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the per-slot attempt table");
		ctx->leak_attempt_slots = n_slots;
	}
	return PC_HIP_OK;
}

template <int MODE>
static int pc_leak_enqueue(pc_hip_ctx *ctx, pc_kargs &a, long long n_items, long long capacity)
{
	long long lanes = 0;
	const int grid = pc_leak_grid(ctx, n_items, lanes);
	int st = pc_leak_buffers(ctx, lanes, capacity, n_items);
	if (st) return st;
	pc_leak_kargs lk;
	lk.amu = ctx->d_amu; lk.frames = ctx->d_leak_frames; lk.max_depth = ctx->leak_max_depth;
	lk.records = ctx->d_leak_records; lk.cursor = ctx->d_leak_cursor; lk.capacity = capacity;
	lk.final_attempt = ctx->d_leak_attempts;
	a.total_threads = lanes;
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_leak_cursor, 0, 2*sizeof(unsigned long long), ctx->stream));
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_totals, 0, ctx->totals_bytes, ctx->stream));
	PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
	if (ctx->host.pm.nmax + 1 <= 1024)
		hipLaunchKernelGGL((pc_leak_kernel<MODE, 1024>), dim3(grid), dim3(PC_LEAK_BLOCK), 0, ctx->stream, a, lk);
	else
		hipLaunchKernelGGL((pc_leak_kernel<MODE, PC_MAX_PITCH>), dim3(grid), dim3(PC_LEAK_BLOCK), 0, ctx->stream, a, lk);
	PC_HIP_CHECK(hipGetLastError());
	PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
	return PC_HIP_OK;
}

/* Brings the event records of the finished run to the host and turns them into the two lists of the reference
 * (src/polycap-source.c:799-879): attempts that appended a VOID record are dropped, the rest is ordered by slot, inside
 * a slot the transmitted attempt first and then the earlier attempts in attempt order, inside an attempt by seq.
 * Returns PC_HIP_OK, or 1 when the record buffer was too small (*needed = records the run produced). */
static int pc_leak_collect(pc_hip_ctx *ctx, long long n_slots, bool explicit_mode, long long *needed)
{
	unsigned long long cur[2] = {0, 0};
	PC_HIP_CHECK(hipMemcpy(cur, ctx->d_leak_cursor, sizeof(cur), hipMemcpyDeviceToHost));
	ctx->leak_ext.clear(); ctx->leak_int.clear();
	ctx->leak_n_ext = ctx->leak_n_int = 0;
	if (cur[1] != 0)
		return pc_fail(PC_HIP_ERR_RUNTIME, "leak run: the chain of wall crossings was deeper than leak_max_depth for " + std::to_string(cur[1]) + " lane(s); raise the option leak_max_depth");
	if ((long long)cur[0] > ctx->leak_capacity_used) { *needed = (long long)cur[0]; return 1; }
	const size_t ne = (size_t)ctx->host.pm.n_energies, stride = PC_LR_HDR + ne, n = (size_t)cur[0];
	std::vector<double> recs(n*stride);
	if (n) PC_HIP_CHECK(hipMemcpy(recs.data(), ctx->d_leak_records, n*stride*sizeof(double), hipMemcpyDeviceToHost));
	std::vector<unsigned int> final_attempt;
	if (!explicit_mode) {
		final_attempt.resize((size_t)n_slots);
		PC_HIP_CHECK(hipMemcpy(final_attempt.data(), ctx->d_leak_attempts, (size_t)n_slots*sizeof(unsigned int), hipMemcpyDeviceToHost));
	}
	struct key_hash { size_t operator()(const std::pair<long long, long long> &k) const { return std::hash<long long>()(k.first*1000003ll + k.second); } };
	std::unordered_set<std::pair<long long, long long>, key_hash> voided;
	for (size_t k = 0; k < n; k++) {
		const double *r = recs.data() + k*stride;
		if (r[PC_LR_KIND] < 0.) voided.insert({(long long)r[PC_LR_SLOT], (long long)r[PC_LR_ATTEMPT]});
	}
	struct item { long long slot, order, seq; size_t idx; };
	std::vector<item> items;
	items.reserve(n);
	const long long slot0 = ctx->leak_slot0;
	for (size_t k = 0; k < n; k++) {
		const double *r = recs.data() + k*stride;
		if (r[PC_LR_KIND] < 0.) continue;
		const long long slot = (long long)r[PC_LR_SLOT], att = (long long)r[PC_LR_ATTEMPT];
		if (!voided.empty() && voided.count({slot, att})) continue;
		long long order = att + 1;
		if (!explicit_mode && (long long)final_attempt[(size_t)(slot - slot0)] == att) order = 0;   /* the transmitted photon's own events come first */
		items.push_back({slot, order, (long long)r[PC_LR_SEQ], k});
	}
	std::sort(items.begin(), items.end(), [](const item &x, const item &y) {
		if (x.slot != y.slot) return x.slot < y.slot;
		if (x.order != y.order) return x.order < y.order;
		return x.seq < y.seq; });
	const size_t ostride = PC_HIP_LEAK_HDR + ne;
	for (const item &it : items) {
		const double *r = recs.data() + it.idx*stride;
		std::vector<double> &dst = (r[PC_LR_KIND] == (double)PC_LEAK_EXT) ? ctx->leak_ext : ctx->leak_int;
		const size_t at = dst.size();
		dst.resize(at + ostride);
		double *o = dst.data() + at;
		o[0] = r[PC_LR_SLOT]; o[1] = r[PC_LR_ATTEMPT];
		for (int c = 0; c < 10; c++) o[2 + c] = r[PC_LR_X + c];      /* coords, direction, elecv, n_refl */
		for (size_t e = 0; e < ne; e++) o[PC_HIP_LEAK_HDR + e] = r[PC_LR_WEIGHTS + e];
	}
	ctx->leak_n_ext = (long long)(ctx->leak_ext.size() / ostride);
	ctx->leak_n_int = (long long)(ctx->leak_int.size() / ostride);
	return PC_HIP_OK;
}

#endif /* PC_LEAK_KERNELS_H */
