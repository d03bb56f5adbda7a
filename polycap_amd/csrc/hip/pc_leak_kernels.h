/*
 * pc_leak_kernels.h -- leak_calc=true on the device: kernel + host code, included by pc_kernels.hip after the
 * context definition (same translation unit, so the two modes share tables, totals and image records).
 *
 * Shape.  The leak run of one launched photon (pc_leak.h) is a long, irregular, strictly sequential job: blocks and
 * steps through the glass for every reflection, a depth-first tree of leaked fractions, every decision discrete.  A lane
 * owns one exit-photon slot at a time (taken from the global counter) and carries its whole state (pc_leak_lane) in
 * registers; the wave is scheduled by KIND OF WORK like pc_trace_kernel: ballots count the lanes waiting for a wall
 * step, a capillary probe, a march step or one of the short bookkeeping states, the most populated class runs a burst,
 * so the 64 photons of a wave do not serialise each other.  The number of independent lanes is bounded by the HBM given
 * to the per-lane stacks (max_depth x (24 + n_energies) doubles each).  All seven profile tables sit in LDS, including
 * ext, which the wall search reads at every step.
 */
#ifndef PC_LEAK_KERNELS_H
#define PC_LEAK_KERNELS_H

#include <algorithm>

#ifndef PC_LEAK_BLOCK
#define PC_LEAK_BLOCK 256
#endif
#ifndef PC_LEAK_OTHER_REPS
#define PC_LEAK_OTHER_REPS 2   /* short states a lane may pass in one unit of that class (1 / 2 / 3: mean wave life 125.1 / 122.4 / 125.7 ms) */
#endif
#ifndef PC_LEAK_MIN_WAVES
#define PC_LEAK_MIN_WAVES 1    /* waves per SIMD the register allocator leaves room for */
#endif

struct pc_leak_kargs {
	const double *amu;             /* [n_energies] */
	double *frames;                /* total_threads x max_depth x (PC_LF_HDR + n_energies) */
	int max_depth;
	double *records;               /* capacity x (PC_LR_HDR + n_energies) */
	unsigned long long *cursor;    /* [0] records appended, [1] lanes whose stack overflowed */
	long long capacity;
	unsigned int *final_attempt;   /* [n_slots]: attempt index of the transmitted photon of each slot (driver mode) */
	unsigned long long *timing;    /* diagnostics (POLYCAP_LEAK_TIMING): 8 clock sums per wave, or null */
	/* order in which the slots are handed out (source runs): order[k] = relative slot index, heaviest first, or null = 0, 1, 2, ...
	 * The first n_heavy of them go to lanes 0 .. heavy_lanes-1 of every heavy_every-th wave, whose other lanes stay out of work
	 * while that tier lasts: a slot of 20 000 units of work on a lane that shares its wave with three others advances several
	 * times faster than among 63, and it is the heaviest slots that decide when the launch ends.  cursor[2], cursor[3]: positions
	 * handed out in the heavy tier and behind it. */
	const unsigned int *order;
	long long n_heavy;
	int heavy_lanes, heavy_every;
	unsigned int *slot_units;      /* [n_slots] units of work spent on each slot (what the next run of the same slots is ordered by), or null */
};

/* lane modes of the scheduler on top of pc_leak_lane::st */
enum { PC_LM_NEED = 0, PC_LM_RUN = 1, PC_LM_IDLE = 2, PC_LM_PARKED = 3 };

template <int MODE, int PITCH>
__global__ void __launch_bounds__(PC_LEAK_BLOCK, PC_LEAK_MIN_WAVES)
pc_leak_kernel(pc_kargs a, pc_leak_kargs lk)
{
	constexpr bool EXPLICIT = (MODE == PC_MODE_EXPLICIT);
	constexpr int NT = (PITCH <= 1024) ? 9 : 7;      /* the step tables of the wall search fit next to the others up to 1024 points */
	__shared__ double lds[NT*PITCH];
	__shared__ pc_marg4 ldsg[PITCH];
	__shared__ pc_drdev ldsd[(PITCH <= 1024) ? PITCH : 1];
	const int npts = a.pm.nmax + 1;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		lds[k] = a.g_z[k];
		lds[PITCH + k] = a.g_cap[k];
		lds[2*PITCH + k] = a.g_zh[k];
		lds[3*PITCH + k] = a.g_cap2[k];
		lds[4*PITCH + k] = a.g_hexd[k];
		lds[5*PITCH + k] = a.g_idz[k];
		lds[6*PITCH + k] = a.g_ext[k];
		if (NT == 9) { lds[7*PITCH + k] = a.g_stp[k]; lds[8*PITCH + k] = a.g_istp[k]; }
		ldsg[k] = a.g_mg[k];
		if (NT == 9) ldsd[k] = a.g_dr[k];
	}
	__syncthreads();
	pc_tables T;
	T.z = lds; T.cap = lds + PITCH; T.zh = lds + 2*PITCH; T.cap2 = lds + 3*PITCH; T.hexd = lds + 4*PITCH; T.idz = lds + 5*PITCH;
	T.ext = lds + 6*PITCH;
	T.stp = (NT == 9) ? lds + 7*PITCH : a.g_stp; T.istp = (NT == 9) ? lds + 8*PITCH : a.g_istp;
	T.mg = ldsg;
	T.dr = (NT == 9) ? ldsd : a.g_dr;
	const pc_params &Pm = a.pm;
	const int ne = Pm.n_energies;
	const long long rec = PC_N_FIELDS + (long long)ne;
	const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & (PC_WAVE - 1);

	pc_leak_lane L;
	L.cx.ec = a.ec; L.cx.amu = lk.amu; L.cx.ne = ne;
	L.cx.frames = lk.frames + gtid * (long long)lk.max_depth * (PC_LF_HDR + ne);
	L.cx.max_depth = lk.max_depth;
	L.cx.sink.records = lk.records; L.cx.sink.cursor = lk.cursor; L.cx.sink.capacity = lk.capacity;
	L.cx.stack_overflow = 0;
	L.cx.seq = 0; L.cx.slot = 0.; L.cx.attempt = 0.;
	L.st = PC_LS_DONE;
	L.ph.wmem = nullptr; L.ph.wstride = 1; L.ph.wset = 0; L.ph.rc = 0;
	const double *w0 = L.cx.frames + PC_LF_HDR;      /* weights of the launched photon */

	int mode = PC_LM_NEED;
	int launched = 0;                 /* a launch of this lane has finished and waits for the driver's verdict */
	long long slot = -1;
	unsigned int attempt = 0;
	unsigned int my_units = 0;           /* units of work of this lane since it took its slot */
	int my_heavy = 0;                    /* the slot of this lane comes from the heavy tier */
	const bool heavy_wave = lk.order && lk.heavy_every > 0 && ((gtid / PC_WAVE) % lk.heavy_every) == 0;
	double cosalpha0 = 0.;
	unsigned int n_exit = 0, n_not_entered = 0, n_not_trans = 0, n_failed = 0, n_launch = 0;
	unsigned long long s_irefl = 0;
	unsigned long long st_units[4] = {0, 0, 0, 0}, st_lanes[4] = {0, 0, 0, 0};   /* scheduler statistics per class */

	unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_first_idle = 0;
	const unsigned long long t_begin = lk.timing ? wall_clock64() : 0ull;
	unsigned long long t_last = t_begin;
	int cls = 5;
	for (;;) {
		if (lk.timing) {
			const unsigned long long now = wall_clock64();
			tk[cls] += now - t_last; t_last = now;
			if (t_first_idle == 0 && __ballot(mode == PC_LM_IDLE) != 0ull) t_first_idle = now - t_begin;
		}
		if (mode == PC_LM_RUN && L.st == PC_LS_DONE) { mode = PC_LM_NEED; launched = 1; }
		const unsigned long long mM = __ballot(mode == PC_LM_RUN && L.st == PC_LS_MARCH);
		const unsigned long long mS = __ballot(mode == PC_LM_RUN && L.st == PC_LS_WALL_STEP);
		const unsigned long long mP = __ballot(mode == PC_LM_RUN && L.st == PC_LS_WALL_PROBE);
		const unsigned long long mO = __ballot(mode == PC_LM_RUN && L.st != PC_LS_MARCH && L.st != PC_LS_WALL_STEP && L.st != PC_LS_WALL_PROBE);
		const unsigned long long mN = __ballot(mode == PC_LM_NEED);
		if ((mM | mS | mP | mO | mN) == 0ull) break;
		const int nM = __popcll(mM), nS = __popcll(mS), nP = __popcll(mP), nO = __popcll(mO), nN = __popcll(mN);
		/* the short states unblock lanes for the long ones: they count double */
		const int best = max(max(nM, nS), max(nP, 2*max(nO, nN)));
		if (nS > 0 && nS == best) {
			/* ---- wall search: blocks / steps through the glass */
			cls = 0;
			const int stop = (nS + 1) / 2;
			for (int b = 0; b < 32; b++) {
				if (mode == PC_LM_RUN && L.st == PC_LS_WALL_STEP) {
					L.st = pc_wall_step(T, Pm, L, L.after_wall);
					my_units++;
				}
				const int c = __popcll(__ballot(mode == PC_LM_RUN && L.st == PC_LS_WALL_STEP));
				st_units[0]++; st_lanes[0] += (unsigned)c;
				if (c < stop) break;
			}
		} else if (nP > 0 && nP == best) {
			/* ---- wall search: segments of the neighbouring capillary */
			cls = 1;
			const int stop = (nP + 1) / 2;
			for (int b = 0; b < 32; b++) {
				if (mode == PC_LM_RUN && L.st == PC_LS_WALL_PROBE) {
					L.st = pc_wall_probe(T, Pm, L, L.after_wall);
					my_units++;
				}
				const int c = __popcll(__ballot(mode == PC_LM_RUN && L.st == PC_LS_WALL_PROBE));
				st_units[1]++; st_lanes[1] += (unsigned)c;
				if (c < stop) break;
			}
		} else if (nM > 0 && nM == best) {
			/* ---- certified march between interactions */
			cls = 2;
			const int stop = (nM + 1) / 2;
			for (int b = 0; b < 32; b++) {
				if (mode == PC_LM_RUN && L.st == PC_LS_MARCH) {
					pc_leak_unit_march(T, Pm, L);
					my_units++;
				}
				const int c = __popcll(__ballot(mode == PC_LM_RUN && L.st == PC_LS_MARCH));
				st_units[2]++; st_lanes[2] += (unsigned)c;
				if (c < stop) break;
			}
		} else if (nO > 0 && nO >= nN) {
			/* ---- segment visits, reflections with their leak bookkeeping, end of a photon */
			cls = 3;
			st_units[3]++; st_lanes[3] += (unsigned)nO;
			/* the short states come in chains (segment visit -> reflection -> ... -> wall search; end of a leaked fraction ->
			 * its parent's reflection resumed): lanes that walk a chain together take its next link in the same pass */
#pragma unroll 1
			for (int rep = 0; rep < PC_LEAK_OTHER_REPS; rep++) {
				if (mode == PC_LM_RUN && L.st != PC_LS_MARCH && L.st != PC_LS_WALL_STEP && L.st != PC_LS_WALL_PROBE && L.st != PC_LS_DONE) {
					pc_leak_unit_other(T, Pm, L);
					my_units++;
				}
			}
		} else {
			/* ---- driver: verdict on finished launches, next attempt or next slot */
			cls = 4;
			if (mode == PC_LM_PARKED) mode = PC_LM_NEED;            /* parked beside the heavy lanes of its wave: look again */
			/* lanes of this wave that are at work on a slot of the heavy tier and will still be after this pass */
			const unsigned long long heavy_busy = __ballot(my_heavy && (mode == PC_LM_RUN || (mode == PC_LM_NEED && launched)));
			if (mode == PC_LM_NEED) {
				int need_slot = 1;
				if (launched) {
					launched = 0;
					const int rc = L.rc;
					if (EXPLICIT) {
						const long long j = slot;
						a.out_rc[j] = rc;
						for (int e = 0; e < ne; e++) a.out_weights[j*ne + e] = w0[e];
						a.out_exit_coords[3*j] = L.ph.Px; a.out_exit_coords[3*j+1] = L.ph.Py; a.out_exit_coords[3*j+2] = L.ph.Pz;
						a.out_exit_dir[3*j] = L.ph.dx; a.out_exit_dir[3*j+1] = L.ph.dy; a.out_exit_dir[3*j+2] = L.ph.dz;
						a.out_exit_elecv[3*j] = L.ph.ex; a.out_exit_elecv[3*j+1] = L.ph.ey; a.out_exit_elecv[3*j+2] = L.ph.ez;
						a.out_irefl[j] = L.ph.irefl;
						a.out_dtravel[j] = L.ph.dtravel;
					} else {
						/* src/polycap-source.c:758-777 */
						int ok = 0;
						if (rc == 0) n_not_trans++;
						else if (rc == 2) n_not_entered++;
						else if (rc == 1) ok = pc_in_exit_window(Pm, L.ph);
						if (ok) {
							n_exit++;
							s_irefl += (unsigned long long)L.ph.irefl;
							for (int e = 0; e < ne; e++) {
								const double w = w0[e];
								pc_atomic_add128(a.sumw + 2*e, (unsigned long long)(w * PC_FIX_SCALE), 0ull);
								if (a.keep_images) a.img[slot*rec + PC_F_WEIGHTS + e] = w;
							}
							if (a.keep_images) {
								/* src/polycap-source.c:900-923 */
								double *r = a.img + slot*rec;
								const pc_photon<0> &ph = L.ph;
								const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
								const double t = (Pm.z_end - ph.Pz) / ph.dz;
								const double ex = ph.Px + ph.dx*t, ey = ph.Py + ph.dy*t, ez = ph.Pz + ph.dz*t;
								r[PC_F_EXITX] = ex; r[PC_F_EXITY] = ey; r[PC_F_EXITZ] = ez;
								r[PC_F_EDIRX] = ph.dx; r[PC_F_EDIRY] = ph.dy;
								double tx = ph.ex*c_ae + ph.dx*c_be, ty = ph.ey*c_ae + ph.dy*c_be, tz = ph.ez*c_ae + ph.dz*c_be;
								pc_norm3(tx, ty, tz);
								r[PC_F_EEVX] = round(tx); r[PC_F_EEVY] = round(ty);
								((long long *)r)[PC_F_NREFL] = ph.irefl;
								const double lx = ex - ph.Px, ly = ey - ph.Py, lz = Pm.z_end - ph.Pz;
								r[PC_F_DTRAVEL] = ph.dtravel + sqrt(lx*lx + ly*ly + lz*lz);
							}
							lk.final_attempt[slot] = attempt;
						} else {
							attempt++;
							if (attempt >= a.max_attempts) {
								n_failed++;
								lk.final_attempt[slot] = attempt;
								if (a.keep_images)
									for (int e = 0; e < ne; e++) a.img[slot*rec + PC_F_WEIGHTS + e] = 0.;
							} else {
								need_slot = 0;
							}
						}
					}
				}
				if (need_slot) {
					if (!EXPLICIT && lk.slot_units && slot >= 0 && slot < a.n_slots) lk.slot_units[slot] = my_units;
					my_units = 0;
					attempt = 0;
					if (EXPLICIT || !lk.order) {
						slot = (long long)atomicAdd(&a.totals->next_slot, 1ull);
						if (slot >= a.n_slots) mode = PC_LM_IDLE;
					} else {
						const bool heavy_lane = heavy_wave && lane < lk.heavy_lanes;
						long long pos = -1;
						int park = 0;
						my_heavy = 0;
						if (heavy_lane) {
							pos = (long long)atomicAdd(&lk.cursor[2], 1ull);
							if (pos >= lk.n_heavy) pos = -1; else my_heavy = 1;
						} else if (heavy_wave && ((heavy_busy & ~(1ull << lane)) != 0ull
						           || (long long)__hip_atomic_load(&lk.cursor[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lk.n_heavy)) {
							park = 1;                                    /* the heavy lanes of this wave have the wave to themselves */
						}
						if (pos < 0 && !park) {
							pos = lk.n_heavy + (long long)atomicAdd(&lk.cursor[3], 1ull);
							if (pos >= a.n_slots) {
								/* nothing light left: whatever is left of the heavy tier */
								pos = (long long)atomicAdd(&lk.cursor[2], 1ull);
								if (pos >= lk.n_heavy) pos = -1;
							}
						}
						if (pos >= 0) slot = (long long)lk.order[pos];
						else { slot = -1; mode = park ? PC_LM_PARKED : PC_LM_IDLE; }
					}
				}
				if (mode != PC_LM_IDLE && mode != PC_LM_PARKED) {
					/* start an attempt */
					n_launch++;
					double z0;
					int st;
					if (EXPLICIT) {
						const long long j = slot;
						z0 = a.in_start[3*j+2];
						st = pc_launch_init(T, Pm, L.ph, a.in_start[3*j], a.in_start[3*j+1], z0, a.in_dir[3*j], a.in_dir[3*j+1], a.in_dir[3*j+2],
						                    a.in_elecv[3*j], a.in_elecv[3*j+1], a.in_elecv[3*j+2]);
						L.cx.slot = (double)j; L.cx.attempt = 0.;
					} else {
						pc_start s;
						pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), attempt, s);
						z0 = s.z;
						st = pc_launch_init(T, Pm, L.ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
						L.cx.slot = (double)(a.slot0 + slot); L.cx.attempt = (double)attempt;
						if (st == PC_ST_MARCH) {
							/* src/polycap-source.c:779-798: start images of the attempt that is now inside a capillary; the slot
							 * belongs to this lane, so a later (transmitted) attempt simply overwrites them */
							cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
							if (a.keep_images) {
								const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
								double *r = a.img + slot*rec;
								r[PC_F_SRCX] = s.srcx; r[PC_F_SRCY] = s.srcy;
								r[PC_F_STARTX] = s.x; r[PC_F_STARTY] = s.y;
								r[PC_F_SDIRX] = s.dx; r[PC_F_SDIRY] = s.dy;
								double tx = s.ex*c_ae + s.dx*c_be, ty = s.ey*c_ae + s.dy*c_be, tz = s.ez*c_ae + s.dz*c_be;
								pc_norm3(tx, ty, tz);
								r[PC_F_SEVX] = round(tx); r[PC_F_SEVY] = round(ty);
							}
						}
					}
					pc_leak_begin(T, Pm, L, st, z0);
					mode = PC_LM_RUN;
				}
			}
		}
	}

	if (lk.timing && lane == 0) {
		unsigned long long *o = lk.timing + 8*(gtid / PC_WAVE);
		for (int k = 0; k < 5; k++) o[k] = tk[k];
		o[5] = wall_clock64() - t_begin; o[6] = t_first_idle; o[7] = t_begin;
	}
	if (L.cx.stack_overflow) atomicAdd(&lk.cursor[1], 1ull);
	if (!EXPLICIT) {
		unsigned long long v0 = pc_wave_sum_u64(n_exit), v1 = pc_wave_sum_u64(n_not_entered), v2 = pc_wave_sum_u64(n_not_trans);
		unsigned long long v3 = pc_wave_sum_u64(s_irefl), v4 = pc_wave_sum_u64(n_failed), v5 = pc_wave_sum_u64(n_launch);
		if (lane == 0) {
			atomicAdd(&a.totals->counters[0], v0);
			atomicAdd(&a.totals->counters[1], v1);
			atomicAdd(&a.totals->counters[2], v2);
			atomicAdd(&a.totals->counters[3], v3);
			if (v4) atomicAdd(&a.totals->counters[4], v4);
			atomicAdd(&a.totals->counters[5], v5);
			/* scheduler statistics (pc_hip_phase_stats): units and lane-units of wall steps, wall probes, march */
			atomicAdd(&a.totals->phase[0], st_units[0]); atomicAdd(&a.totals->phase[1], st_lanes[0]);
			atomicAdd(&a.totals->phase[2], st_units[1]); atomicAdd(&a.totals->phase[3], st_lanes[1]);
			atomicAdd(&a.totals->phase[4], st_units[2]); atomicAdd(&a.totals->phase[5], st_lanes[2]);
			atomicAdd(&a.totals->phase[6], st_units[3]); atomicAdd(&a.totals->phase[7], st_lanes[3]);
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
		if (hipMalloc(&ctx->d_leak_cursor, 4*sizeof(unsigned long long)) != hipSuccess)
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
	lk.timing = nullptr;
	lk.order = nullptr; lk.n_heavy = 0; lk.heavy_lanes = 0; lk.heavy_every = 0; lk.slot_units = nullptr;
	if (MODE != PC_MODE_EXPLICIT) {
		if (ctx->d_leak_order && ctx->leak_order_n == n_items) {
			lk.order = ctx->d_leak_order;
			lk.heavy_lanes = ctx->leak_heavy_lanes; lk.heavy_every = ctx->leak_heavy_every;
			lk.n_heavy = (lk.heavy_lanes > 0 && lk.heavy_every > 0) ? std::min<long long>(ctx->leak_n_heavy, n_items) : 0;
		}
		if (ctx->leak_slot_units) {
			if (ctx->d_leak_slot_units && ctx->leak_slot_units_n < n_items) { (void)hipFree(ctx->d_leak_slot_units); ctx->d_leak_slot_units = nullptr; }
			if (!ctx->d_leak_slot_units) { PC_HIP_CHECK(hipMalloc(&ctx->d_leak_slot_units, (size_t)n_items*sizeof(unsigned int))); ctx->leak_slot_units_n = n_items; }
			PC_HIP_CHECK(hipMemsetAsync(ctx->d_leak_slot_units, 0, (size_t)n_items*sizeof(unsigned int), ctx->stream));
			lk.slot_units = ctx->d_leak_slot_units;
		}
	}
	if (getenv("POLYCAP_LEAK_TIMING")) {
		/* diagnostics: where the waves of the leak kernel spend their time (printed by pc_leak_collect) */
		const size_t nb = (size_t)(lanes / PC_WAVE) * 8 * sizeof(unsigned long long);
		if (ctx->d_leak_timing && ctx->leak_timing_bytes < nb) { (void)hipFree(ctx->d_leak_timing); ctx->d_leak_timing = nullptr; }
		if (!ctx->d_leak_timing) { PC_HIP_CHECK(hipMalloc(&ctx->d_leak_timing, nb)); ctx->leak_timing_bytes = nb; }
		PC_HIP_CHECK(hipMemsetAsync(ctx->d_leak_timing, 0, nb, ctx->stream));
		lk.timing = ctx->d_leak_timing;
		ctx->leak_timing_waves = lanes / PC_WAVE;
	}
	a.total_threads = lanes;
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_leak_cursor, 0, 4*sizeof(unsigned long long), ctx->stream));
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_totals, 0, ctx->totals_bytes, ctx->stream));
	if (!ctx->leak_ev0_done) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
	if (ctx->host.pm.nmax + 1 <= 1024)
		hipLaunchKernelGGL((pc_leak_kernel<MODE, 1024>), dim3(grid), dim3(PC_LEAK_BLOCK), 0, ctx->stream, a, lk);
	else
		hipLaunchKernelGGL((pc_leak_kernel<MODE, PC_MAX_PITCH>), dim3(grid), dim3(PC_LEAK_BLOCK), 0, ctx->stream, a, lk);
	PC_HIP_CHECK(hipGetLastError());
	ctx->last_kernel = 5;
	PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
	return PC_HIP_OK;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * The events of a finished run, put into the two lists of the reference (src/polycap-source.c:799-879) ON THE DEVICE: attempts
 * that appended a VOID record are dropped, the rest is ordered by slot, inside a slot the transmitted attempt first and then the
 * earlier attempts in attempt order, inside an attempt by seq; extleak and intleak events go to lists of their own in that
 * order.  Round 3 fetched the raw records (120 B each) and did this with host threads: 243 ms for the 2.46e6 events of the leak
 * bench, more than the kernel.  Now: keys -> two stable radix sorts (seq, then (slot, order)) -> flags and positions (prefix sums)
 * -> one gather into the output rows (PC_HIP_LEAK_HDR + n_energies doubles), which are copied once into pinned host memory. */
#include <hipcub/hipcub.hpp>

#define PC_LEAK_ORDER_BITS 22          /* order = 0 (the transmitted attempt) or attempt + 1 <= 2^20 + 1 */

__global__ void __launch_bounds__(256) pc_leak_keys_kernel(const double *recs, long long stride, long long n, long long slot0, long long n_slots,
	const unsigned int *final_attempt, unsigned long long *key, unsigned int *seq, unsigned int *idx,
	unsigned long long *void_keys, unsigned int *void_n, unsigned int *bad)
{
	const long long k = (long long)blockIdx.x*blockDim.x + threadIdx.x;
	if (k >= n) return;
	const double *r = recs + k*stride;
	const long long sl = (long long)r[PC_LR_SLOT] - slot0;
	const long long att = (long long)r[PC_LR_ATTEMPT];
	if (sl < 0 || sl >= n_slots || att < 0 || att + 1 >= (1ll << PC_LEAK_ORDER_BITS)) { atomicAdd(bad, 1u); key[k] = ~0ull; seq[k] = 0u; idx[k] = (unsigned int)k; return; }
	/* the transmitted photon's own events come first (source runs; explicit photons have one attempt each) */
	const long long order = (final_attempt != nullptr && (long long)final_attempt[sl] == att) ? 0 : att + 1;
	const unsigned long long kk = ((unsigned long long)sl << PC_LEAK_ORDER_BITS) | (unsigned long long)order;
	key[k] = kk;
	seq[k] = (unsigned int)r[PC_LR_SEQ];
	idx[k] = (unsigned int)k;
	if (r[PC_LR_KIND] < 0.) void_keys[atomicAdd(void_n, 1u)] = kk;
}

__global__ void __launch_bounds__(256) pc_leak_gather_keys_kernel(const unsigned long long *key, const unsigned int *idx, long long n, unsigned long long *out)
{
	const long long k = (long long)blockIdx.x*blockDim.x + threadIdx.x;
	if (k < n) out[k] = key[idx[k]];
}

/* in sorted order: does the event go to the extleak / intleak list?  (not a VOID record, not of a voided attempt) */
__global__ void __launch_bounds__(256) pc_leak_flags_kernel(const double *recs, long long stride, const unsigned long long *skey, const unsigned int *sidx,
	long long n, const unsigned long long *void_sorted, const unsigned int *void_n, unsigned int *f_ext, unsigned int *f_int)
{
	const long long k = (long long)blockIdx.x*blockDim.x + threadIdx.x;
	if (k > n) return;
	if (k == n) { f_ext[k] = 0u; f_int[k] = 0u; return; }     /* the prefix sums run over n + 1 flags: the last sums are the counts */
	const double kind = recs[(long long)sidx[k]*stride + PC_LR_KIND];
	const unsigned long long kk = skey[k];
	bool keep = kind >= 0.;
	if (keep) {
		unsigned int lo = 0, hi = *void_n;
		while (lo < hi) { const unsigned int mid = (lo + hi) >> 1; if (void_sorted[mid] < kk) lo = mid + 1; else hi = mid; }
		if (lo < *void_n && void_sorted[lo] == kk) keep = false;
	}
	f_ext[k] = (keep && kind == (double)PC_LEAK_EXT) ? 1u : 0u;
	f_int[k] = (keep && kind == (double)PC_LEAK_INT) ? 1u : 0u;
}

/* output rows: slot, attempt, coords, direction, electric vector, n_refl, weights; the intleak list behind the extleak list */
__global__ void __launch_bounds__(256) pc_leak_rows_kernel(const double *recs, long long stride, const unsigned int *sidx, long long n, int ne,
	const unsigned int *f_ext, const unsigned int *f_int, const unsigned int *p_ext, const unsigned int *p_int, double *out)
{
	const long long ostride = PC_HIP_LEAK_HDR + ne;
	const long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x;
	const long long k = t / ostride;
	const int c = (int)(t - k*ostride);
	if (k >= n) return;
	long long pos;
	if (f_ext[k]) pos = (long long)p_ext[k];
	else if (f_int[k]) pos = (long long)p_ext[n] + (long long)p_int[k];
	else return;
	const double *r = recs + (long long)sidx[k]*stride;
	const int from = (c == 0) ? PC_LR_SLOT : ((c == 1) ? PC_LR_ATTEMPT : ((c < PC_HIP_LEAK_HDR) ? PC_LR_X + (c - 2) : PC_LR_WEIGHTS + (c - PC_HIP_LEAK_HDR)));
	out[pos*ostride + c] = r[from];
}

static int pc_leak_order_buffers(pc_hip_ctx *ctx, size_t n, size_t ostride, size_t temp_bytes)
{
	/* keys n x 8 x 3, seq n x 4 x 2, idx n x 4 x 3, flags + positions (n + 1) x 4 x 4, void keys n x 8 x 2, counters */
	const size_t need = n*8*3 + n*4*2 + n*4*3 + (n + 1)*4*4 + n*8*2 + 256 + temp_bytes + 4096;
	if (need > ctx->leak_order_bytes) {
		if (ctx->d_leak_order_tmp) (void)hipFree(ctx->d_leak_order_tmp);
		ctx->d_leak_order_tmp = nullptr; ctx->leak_order_bytes = 0;
		if (hipMalloc(&ctx->d_leak_order_tmp, need) != hipSuccess) { (void)hipGetLastError(); return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the ordering buffers"); }
		ctx->leak_order_bytes = need;
	}
	const size_t out_elems = (n ? n : 1)*ostride;
	if (out_elems > ctx->leak_out_elems) {
		if (ctx->d_leak_out) (void)hipFree(ctx->d_leak_out);
		if (ctx->h_leak_out) (void)hipHostFree(ctx->h_leak_out);
		ctx->d_leak_out = nullptr; ctx->h_leak_out = nullptr; ctx->leak_out_elems = 0;
		const size_t grow = out_elems + out_elems/8;
		if (hipMalloc(&ctx->d_leak_out, grow*sizeof(double)) != hipSuccess || hipHostMalloc(&ctx->h_leak_out, grow*sizeof(double)) != hipSuccess) {
			(void)hipGetLastError();
			return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the event lists");
		}
		ctx->leak_out_elems = grow;
	}
	return PC_HIP_OK;
}

/* Orders the event records of the finished run on the device and brings the two lists to the host (pinned memory, kept by the
 * context until its next leak run).  Returns PC_HIP_OK, or 1 when the record buffer was too small (*needed = records the run
 * produced). */
static int pc_leak_collect(pc_hip_ctx *ctx, long long n_slots, bool explicit_mode, long long *needed)
{
	unsigned long long cur[2] = {0, 0};
	PC_HIP_CHECK(hipMemcpy(cur, ctx->d_leak_cursor, sizeof(cur), hipMemcpyDeviceToHost));
	if (ctx->d_leak_timing && getenv("POLYCAP_LEAK_TIMING")) {
		const size_t nw = (size_t)ctx->leak_timing_waves;
		std::vector<unsigned long long> t(nw*8);
		PC_HIP_CHECK(hipMemcpy(t.data(), ctx->d_leak_timing, nw*8*sizeof(unsigned long long), hipMemcpyDeviceToHost));
		double sum[8] = {0}, mx[8] = {0}; size_t ran = 0;
		unsigned long long t0 = ~0ull, t1 = 0;
		for (size_t w = 0; w < nw; w++) {
			if (t[w*8+5] == 0) continue;
			ran++;
			for (int k = 0; k < 7; k++) { sum[k] += (double)t[w*8+k]; if ((double)t[w*8+k] > mx[k]) mx[k] = (double)t[w*8+k]; }
			if (t[w*8+7] < t0) t0 = t[w*8+7];
			if (t[w*8+7] + t[w*8+5] > t1) t1 = t[w*8+7] + t[w*8+5];
		}
		std::vector<unsigned long long> life;
		for (size_t w = 0; w < nw; w++) if (t[w*8+5]) life.push_back(t[w*8+5]);
		std::sort(life.begin(), life.end());
		const double tick = 1e-5;     /* wall_clock64: 100 MHz -> ms */
		fprintf(stderr, "leak timing: %zu waves ran; span %.1f ms; per wave mean (max) in ms: wall %.1f (%.1f) probe %.1f (%.1f) march %.1f (%.1f) other %.1f (%.1f) driver %.1f (%.1f) | life %.1f (%.1f) median %.1f p90 %.1f | first idle lane after %.1f (%.1f)\n",
		        ran, (double)(t1 - t0)*tick, sum[0]/ran*tick, mx[0]*tick, sum[1]/ran*tick, mx[1]*tick, sum[2]/ran*tick, mx[2]*tick, sum[3]/ran*tick, mx[3]*tick,
		        sum[4]/ran*tick, mx[4]*tick, sum[5]/ran*tick, mx[5]*tick, life.empty() ? 0. : (double)life[life.size()/2]*tick, life.empty() ? 0. : (double)life[life.size()*9/10]*tick,
		        sum[6]/ran*tick, mx[6]*tick);
	}
	ctx->leak_n_ext = ctx->leak_n_int = 0;
	if (cur[1] != 0)
		return pc_fail(PC_HIP_ERR_RUNTIME, "leak run: the chain of wall crossings was deeper than leak_max_depth for " + std::to_string(cur[1]) + " lane(s); raise the option leak_max_depth");
	if ((long long)cur[0] > ctx->leak_capacity_used) { *needed = (long long)cur[0]; return 1; }
	const size_t ne = (size_t)ctx->host.pm.n_energies, stride = PC_LR_HDR + ne, ostride = PC_HIP_LEAK_HDR + ne, n = (size_t)cur[0];
	if (n == 0) return PC_HIP_OK;
	if (n >= (1ull << 31) - 2) return pc_fail(PC_HIP_ERR_INVALID, "leak run: more than 2^31 event records in one run (the sorts count in int); trace the slots in several runs");
	hipStream_t st = ctx->stream;
	const int end_bit = 64;
	/* temporary storage of the library calls: the largest of the three sorts and the prefix sum */
	size_t tb = 0, t1 = 0;
	PC_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const unsigned int *)nullptr, (unsigned int *)nullptr, (const unsigned int *)nullptr, (unsigned int *)nullptr, (int)n, 0, 32, st)); tb = std::max(tb, t1);
	PC_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const unsigned int *)nullptr, (unsigned int *)nullptr, (int)n, 0, end_bit, st)); tb = std::max(tb, t1);
	PC_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, t1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n, 0, end_bit, st)); tb = std::max(tb, t1);
	PC_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (const unsigned int *)nullptr, (unsigned int *)nullptr, (int)n + 1, st)); tb = std::max(tb, t1);
	int rc = pc_leak_order_buffers(ctx, n, ostride, tb);
	if (rc) return rc;
	char *base = (char *)ctx->d_leak_order_tmp;
	auto take = [&](size_t bytes) { char *p = base; base += (bytes + 255) & ~(size_t)255; return (void *)p; };
	unsigned long long *key = (unsigned long long *)take(n*8), *key2 = (unsigned long long *)take(n*8), *key3 = (unsigned long long *)take(n*8);
	unsigned int *seq = (unsigned int *)take(n*4), *seq2 = (unsigned int *)take(n*4);
	unsigned int *idx = (unsigned int *)take(n*4), *idx2 = (unsigned int *)take(n*4), *idx3 = (unsigned int *)take(n*4);
	unsigned int *f_ext = (unsigned int *)take((n + 1)*4), *f_int = (unsigned int *)take((n + 1)*4), *p_ext = (unsigned int *)take((n + 1)*4), *p_int = (unsigned int *)take((n + 1)*4);
	unsigned long long *vk = (unsigned long long *)take(n*8), *vk2 = (unsigned long long *)take(n*8);
	unsigned int *cnt = (unsigned int *)take(64);       /* [0] void records, [1] records with a slot or attempt outside the run */
	void *tmp = take(tb);
	const unsigned blocks = (unsigned)((n + 255)/256);
	PC_HIP_CHECK(hipMemsetAsync(cnt, 0, 64, st));
	hipLaunchKernelGGL(pc_leak_keys_kernel, dim3(blocks), dim3(256), 0, st, ctx->d_leak_records, (long long)stride, (long long)n, (long long)ctx->leak_slot0, n_slots,
	                   explicit_mode ? (const unsigned int *)nullptr : ctx->d_leak_attempts, key, seq, idx, vk, cnt, cnt + 1);
	PC_HIP_CHECK(hipGetLastError());
	unsigned int hc[2] = {0, 0};
	PC_HIP_CHECK(hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
	PC_HIP_CHECK(hipStreamSynchronize(st));
	if (hc[1]) return pc_fail(PC_HIP_ERR_RUNTIME, "leak run: event record with a slot or attempt outside the run");
	if (hc[0]) PC_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, vk, vk2, (int)hc[0], 0, end_bit, st));
	/* stable sorts: by seq, then by (slot, order) */
	PC_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, seq, seq2, idx, idx2, (int)n, 0, 32, st));
	hipLaunchKernelGGL(pc_leak_gather_keys_kernel, dim3(blocks), dim3(256), 0, st, key, idx2, (long long)n, key2);
	PC_HIP_CHECK(hipGetLastError());
	PC_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key2, key3, idx2, idx3, (int)n, 0, end_bit, st));
	hipLaunchKernelGGL(pc_leak_flags_kernel, dim3((unsigned)((n + 1 + 255)/256)), dim3(256), 0, st, ctx->d_leak_records, (long long)stride, key3, idx3, (long long)n,
	                   hc[0] ? vk2 : vk, cnt, f_ext, f_int);
	PC_HIP_CHECK(hipGetLastError());
	PC_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, f_ext, p_ext, (int)n + 1, st));
	PC_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb, f_int, p_int, (int)n + 1, st));
	const unsigned long long cells = (unsigned long long)n*ostride;
	hipLaunchKernelGGL(pc_leak_rows_kernel, dim3((unsigned)((cells + 255)/256)), dim3(256), 0, st, ctx->d_leak_records, (long long)stride, idx3, (long long)n, (int)ne,
	                   f_ext, f_int, p_ext, p_int, ctx->d_leak_out);
	PC_HIP_CHECK(hipGetLastError());
	unsigned int tot[2] = {0, 0};
	PC_HIP_CHECK(hipMemcpyAsync(&tot[0], p_ext + n, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
	PC_HIP_CHECK(hipMemcpyAsync(&tot[1], p_int + n, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
	PC_HIP_CHECK(hipStreamSynchronize(st));
	const size_t rows = (size_t)tot[0] + (size_t)tot[1];
	if (rows) PC_HIP_CHECK(hipMemcpyAsync(ctx->h_leak_out, ctx->d_leak_out, rows*ostride*sizeof(double), hipMemcpyDeviceToHost, st));
	PC_HIP_CHECK(hipStreamSynchronize(st));
	ctx->leak_n_ext = (long long)tot[0];
	ctx->leak_n_int = (long long)tot[1];
	return PC_HIP_OK;
}

#endif /* PC_LEAK_KERNELS_H */
