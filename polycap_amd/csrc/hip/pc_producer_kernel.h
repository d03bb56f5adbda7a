/* Single-energy trace kernel with a launching wave per workgroup (included by pc_kernels.hip).  Option "producer".
 *
 * In pc_trace_kernel<1, MODE> a lane that has finished a photon finalises it, takes a slot, samples the source and runs the
 * entrance tests in one NEW phase: ~1280 VALU instructions that run with 16 of 64 lanes on xos1 (13 % of the kernel), and
 * the finished lanes idle until it runs (another ~13 %).  Doing the launches 64 at a time inside the tracing wave does not
 * work: the sampling code next to a live photon does not fit into 128 registers (profiles/r02/queue_kernel_experiment.txt).
 * Here wave 0 of the 1024-thread workgroup (one per CU) traces nothing.  It takes (slot, attempt) requests -- fresh slots from
 * the work counter, or the next attempt of a slot whose photon failed -- samples the source and runs the entrance tests for
 * 64 of them at a time (what polycap_source_get_photon and the head of polycap_photon_launch do) and hands the entered
 * photons to the 15 tracing waves through single-producer / single-consumer rings in LDS; finished photons come back through
 * rings of the same kind and are finalised by it, again 64 at a time (exit window, exit record, exact sums:
 * src/polycap-source.c:758-777, 900-923); an absorbed photon or one that missed the exit window becomes the lane's request for
 * the slot's next attempt.  A tracing wave's NEW phase is a push and a pop.  A photon's result depends on (seed, slot,
 * attempt) only and the totals are exact integers, so the output is bit-identical to pc_trace_kernel's (tests/test_gpu_parity.py,
 * scripts/analysis/check_option.py, scripts/analysis/soak_producer.py).
 *
 * The launching wave ends the run (flag `done`: work counter exhausted, no slot in flight); its own waits are bounded
 * (PC3_MAX_POLLS consecutive polls without progress: it then marks the run failed), and a tracing wave with nothing to trace
 * leaves on either flag.  The slots in flight are capped (PC3_MAX_OUTSTANDING) below the number at which every ring could be
 * full while every lane waits to push: there is always a wave that can move. */
#ifndef PC_PRODUCER_KERNEL_H
#define PC_PRODUCER_KERNEL_H

#ifndef PC3_ROUTE_BND
#define PC3_ROUTE_BND 1       /* photons of boundary capillaries are handed to the last tracing wave (see producer_step) */
#endif
#ifndef PC3_BLOCK
#define PC3_BLOCK 1024         /* one workgroup per CU: 1 launching + 15 tracing waves, the tables once in LDS */
#endif
#define PC3_WAVES (PC3_BLOCK / PC_WAVE)
#define PC3_CONSUMERS (PC3_WAVES - 1)
#define PC3_MAXCONS PC3_WAVES
#define PC3_PITCH 1024
#ifndef PC3_CAP
#define PC3_CAP 28            /* launched photons waiting per tracing wave */
#endif
#define PC3_FIELDS 11         /* x, y, dx, dy, dz, ex, ey, ez, kn, (slot, attempt), (qr, bnd) */
#define PC3_DCAP 16           /* finished photons waiting per tracing wave */
#define PC3_DFIELDS 13        /* P, d, e (9), dtravel, weight, (slot, attempt), (reflections, return code) */
#define PC3_MAX_POLLS 4000000
#define PC3_MAX_OUTSTANDING (PC3_CONSUMERS*(PC_WAVE + PC3_CAP) + 20)
#define PC3_MIN_REFL 4.0      /* option "producer" = -1: reflections per launch from which this kernel is used (xos1 at 10-30 keV: 26-12,
                               * always 10-16 % faster; cone.inp: 0.3, 2x slower; scripts/analysis/producer_crossover.py) */
#ifndef PC3_SLEEP
#define PC3_SLEEP 127          /* the launching wave waits for room in the rings 90 % of the time: long naps (8128 clocks) */
#endif

struct pc3_ctrl {
	unsigned int q_head[16], q_tail[16];      /* rings of launched photons: tail written by the producer, head by the consumer */
	unsigned int r_head[16], r_tail[16];      /* rings of finished photons: tail written by the consumer, head by the producer */
	unsigned int outstanding;               /* slots taken from the work counter and not finished yet */
	unsigned int done;                      /* set by the producer when nothing is left to launch and nothing is in flight */
	unsigned int failed;                    /* a wave gave up waiting */
	unsigned int pad;
};

__device__ __forceinline__ unsigned int pc3_load(const unsigned int *p)
{
	return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pc3_store(unsigned int *p, unsigned int v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* STATS: the march burst counts its steps and their active lanes (pc_hip_phase_stats; option "march_stats").  The production
 * instantiation leaves those two counters at zero: their ballot + popcount sat inside the unrolled hot loop. */
template <int MODE, bool STATS = false>
__global__ void __launch_bounds__(PC3_BLOCK, 4)
pc_trace_producer_kernel(pc_kargs a)
{
	__shared__ double lds[6*PC3_PITCH];
	__shared__ pc_marg4 ldsg[PC3_PITCH];
	__shared__ double l_ring[PC3_MAXCONS*PC3_FIELDS*PC3_CAP];
	__shared__ double l_done[PC3_MAXCONS*PC3_DFIELDS*PC3_DCAP];
	__shared__ unsigned long long l_req[PC_WAVE];      /* the launching wave's requests, one per lane: (slot << 24 | attempt) + 1, 0 = none */
	__shared__ pc3_ctrl ctl;
	const int npts = a.pm.nmax + 1;
	double *l_z = lds, *l_cap = lds + PC3_PITCH, *l_zh = lds + 2*PC3_PITCH, *l_cap2 = lds + 3*PC3_PITCH, *l_hexd = lds + 4*PC3_PITCH, *l_idz = lds + 5*PC3_PITCH;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		l_z[k] = a.g_z[k];
		l_cap[k] = a.g_cap[k];
		l_zh[k] = a.g_zh[k];
		l_cap2[k] = a.g_cap2[k];
		l_hexd[k] = a.g_hexd[k];
		l_idz[k] = a.g_idz[k];
		ldsg[k] = a.g_mg[k];
	}
	if (threadIdx.x < (int)(sizeof(pc3_ctrl)/sizeof(unsigned int))) ((unsigned int *)&ctl)[threadIdx.x] = 0u;
	if (threadIdx.x < PC_WAVE) l_req[threadIdx.x] = 0ull;
	__syncthreads();
	const int lane = threadIdx.x & (PC_WAVE - 1);
	const int wave = threadIdx.x / PC_WAVE;
	const unsigned long long below = (1ull << lane) - 1ull;
	pc_tables T;
	T.z = l_z; T.cap = l_cap; T.zh = l_zh; T.cap2 = l_cap2; T.hexd = l_hexd; T.idz = l_idz; T.ext = a.g_ext;
	T.mg = ldsg;
	const long long fs = a.img_fs, ss = a.img_ss, ws = a.img_ws;
	const pc_params &Pm = a.pm;
	unsigned long long u_exit = 0, u_not_entered = 0, u_not_trans = 0, u_irefl = 0, u_failed = 0, u_launch = 0;
	unsigned long long u_acc_lo = 0, u_acc_hi = 0;
	unsigned long long st_march = 0, st_march_l = 0, st_event = 0, st_event_l = 0, st_new = 0, st_new_l = 0, st_batches = 0;
	long long polls = 0;

	/* wave-uniform state of the launching wave */
	long long chunk_next = 0, chunk_end = 0;
	int fresh_left = 1;
	/* One step of the launching wave: gather requests, launch a batch when there is room for it.  Returns 0 after a batch,
	 * 1 when there was nothing to do or no room, 2 when the run is over, 3 when it gave up. */
	auto producer_step = [&]() -> int {
		unsigned long long rq = l_req[lane];
		int have = rq != 0ull;
		long long f_slot = have ? (long long)((rq - 1ull) >> 24) : 0;
		unsigned int f_att = have ? (unsigned int)((rq - 1ull) & 0xffffffull) : 0u;
		{
			if (polls > PC3_MAX_POLLS) { if (lane == 0) atomicAdd(&ctl.failed, 1u); return 3; }
			/* finished photons of the tracing waves, ring by ring, into the lanes that hold no request: finalised here, 64 at a
			 * time (src/polycap-source.c:758-777, 900-923); a failed one becomes this lane's request for the slot's next attempt */
			{
				int fin = 0;
				double gPx = 0., gPy = 0., gPz = 0., gdx = 0., gdy = 0., gdz = 1., gex = 0., gey = 0., gez = 0., gdt = 0., gw = 0.;
				long long g_slot = 0;
				unsigned int g_att = 0;
				int g_irefl = 0, g_rc = 0;
#pragma unroll 1
				for (int c = 0; c < PC3_CONSUMERS; c++) {
					const unsigned long long mFree = __ballot(!have && !fin);
					if (mFree == 0ull) break;
					const unsigned int rh = ctl.r_head[c], rt = pc3_load(&ctl.r_tail[c]);
					int n = (int)(rt - rh);
					const int nfree = __popcll(mFree);
					if (n > nfree) n = nfree;
					if (n > 0) {
						const int rk = __popcll(mFree & below);
						if (!have && !fin && rk < n) {
							const double *q = l_done + (size_t)c*(PC3_DFIELDS*PC3_DCAP) + ((rh + (unsigned)rk) % PC3_DCAP);
							gPx = q[0*PC3_DCAP]; gPy = q[1*PC3_DCAP]; gPz = q[2*PC3_DCAP];
							gdx = q[3*PC3_DCAP]; gdy = q[4*PC3_DCAP]; gdz = q[5*PC3_DCAP];
							gex = q[6*PC3_DCAP]; gey = q[7*PC3_DCAP]; gez = q[8*PC3_DCAP];
							gdt = q[9*PC3_DCAP]; gw = q[10*PC3_DCAP];
							const unsigned long long w0 = (unsigned long long)__double_as_longlong(q[11*PC3_DCAP]);
							const unsigned long long w1 = (unsigned long long)__double_as_longlong(q[12*PC3_DCAP]);
							g_slot = (long long)(w0 >> 24); g_att = (unsigned int)(w0 & 0xffffffull);
							g_irefl = (int)(w1 >> 8); g_rc = (int)(w1 & 0xffull) - 2;
							fin = 1;
						}
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0) pc3_store(&ctl.r_head[c], rh + (unsigned)n);
					}
				}
				const unsigned long long mFin = __ballot(fin);
				if (mFin) {
					polls = 0;
					int f_exit = 0, f_nt = 0, f_fail = 0;
					unsigned long long f_w = 0;
					if (fin) {
						int ok = 0;
						if (g_rc == 0) f_nt = 1;
						else if (g_rc == 1) {
							pc_photon<1> fp;
							fp.Px = gPx; fp.Py = gPy; fp.Pz = gPz; fp.dx = gdx; fp.dy = gdy; fp.dz = gdz;
							ok = pc_in_exit_window(Pm, fp);
						}
						if (ok) {
							f_exit = 1;
							f_w = (unsigned long long)(gw * PC_FIX_SCALE);
							if (a.keep_images && !a.img_cursor) {
								double *r = a.img + g_slot*ss;
								const double cosalpha0 = __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)(r + PC_F_EEVX*fs),
								                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
								a.img_w[g_slot*ws] = gw;
								double t = (Pm.z_end - gPz) / gdz;
								double ex = gPx + gdx*t, ey = gPy + gdy*t, ez = gPz + gdz*t;
								r[PC_F_EXITX*fs] = ex; r[PC_F_EXITY*fs] = ey; r[PC_F_EXITZ*fs] = ez;
								r[PC_F_EDIRX*fs] = gdx; r[PC_F_EDIRY*fs] = gdy;
								const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
								double tx = gex*c_ae + gdx*c_be, ty = gey*c_ae + gdy*c_be, tz = gez*c_ae + gdz*c_be;
								pc_norm3(tx, ty, tz);
								r[PC_F_EEVX*fs] = round(tx); r[PC_F_EEVY*fs] = round(ty);
								((long long *)r)[PC_F_NREFL*fs] = g_irefl;
								double lx = ex - gPx, ly = ey - gPy, lz = Pm.z_end - gPz;
								r[PC_F_DTRAVEL*fs] = gdt + sqrt(lx*lx + ly*ly + lz*lz);
							}
						} else if (g_att + 1 >= a.max_attempts) {
							f_fail = 1;
							if (a.keep_images && !a.img_cursor) { a.img_w[g_slot*ws] = 0.; a.img[g_slot*ss + PC_F_EEVX*fs] = 0.; }
						} else {
							f_slot = g_slot; f_att = g_att + 1; have = 1;
						}
					}
					u_not_trans += (unsigned long long)__popcll(__ballot(f_nt));
					u_failed += (unsigned long long)__popcll(__ballot(f_fail));
					const unsigned long long mX = __ballot(f_exit);
					if (a.keep_images && a.img_cursor && mX) {
						/* compact store: the exit photons of this batch take the next positions of the planes -- one coalesced run per
						 * plane -- and everything about them is written here, once: the start images are sampled again from
						 * (seed, slot, attempt), which costs this wave ~400 instructions per batch and no memory traffic */
						const int kx = __popcll(mX);
						unsigned long long base = 0ull;
						if (lane == 0) base = atomicAdd(a.img_cursor, (unsigned long long)kx);
						base = __shfl(base, 0, PC_WAVE);
						if (f_exit) {
							const long long pos = (long long)(base + (unsigned long long)__popcll(mX & below));
							pc_start s;
							pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + g_slot), g_att, s);
							const double cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
							double evx, evy;
							pc_start_elecv_image(s, cosalpha0, evx, evy);
							pc_write_start_fields<true>(a, pos, s.srcx, s.srcy, s.x, s.y, s.dx, s.dy, evx, evy);
							pc_write_exit_fields<true>(a, Pm, pos, gPx, gPy, gPz, gdx, gdy, gdz, gex, gey, gez, cosalpha0, (long long)g_irefl, gdt);
							pc_store_wt(a.img_w + pos*ws, gw);
							if (a.img_ids) pc_store_wt(a.img_ids + pos, g_slot);
						}
						if (a.blk_done) {
							asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the stores above have reached memory */
							if (lane == 0) pc_blocks_written(a, base, kx);
						}
					}
					const int nfin = __popcll(mX) + __popcll(__ballot(f_fail));
					if (nfin > 0 && lane == 0) atomicSub(&ctl.outstanding, (unsigned int)nfin);
					if (mX) {
						u_exit += (unsigned long long)__popcll(mX);
						u_irefl += pc_wave_sum_u64((unsigned long long)(f_exit ? g_irefl : 0));
						const unsigned long long s_low = pc_wave_sum_u64(f_w & 0xffffffffull), s_high = pc_wave_sum_u64(f_w >> 32);
						const unsigned long long lo = s_low + (s_high << 32);
						const unsigned long long hi = (s_high >> 32) + ((lo < s_low) ? 1ull : 0ull);
						const unsigned long long old = u_acc_lo;
						u_acc_lo = old + lo;
						u_acc_hi += hi + ((u_acc_lo < old) ? 1ull : 0ull);
					}
				}
			}
			/* fresh slots for the lanes that are still without a request -- while the slots in flight stay below what the
			 * lanes and the rings of launched photons hold (PC3_MAX_OUTSTANDING): with more, retry requests could fill every
			 * retry ring while every tracing lane waits to file one and every launching lane holds one -- nobody could move */
			if (fresh_left) {
				const unsigned long long mFree = __ballot(!have);
				int nf = __popcll(mFree);
				{
					const int room = PC3_MAX_OUTSTANDING - (int)pc3_load(&ctl.outstanding);
					if (nf > room) nf = (room > 0) ? room : 0;
				}
				if (nf > 0) {
					const int rank = __popcll(mFree & below);
					long long got = -1;
					if (chunk_end - chunk_next < nf) {
						const long long left = chunk_end - chunk_next;
						long long base_new = 0;
						if (lane == 0) base_new = (long long)atomicAdd(a.work, (unsigned long long)PC_CHUNK);
						base_new = __shfl(base_new, 0, PC_WAVE);
						if (!have && rank < nf) got = (rank < left) ? (chunk_next + rank) : (base_new + (rank - left));
						chunk_next = base_new + (nf - left);
						chunk_end = base_new + PC_CHUNK;
						if (base_new >= a.n_slots) fresh_left = 0;
					} else {
						if (!have && rank < nf) got = chunk_next + rank;
						chunk_next += nf;
					}
					const int take = (!have && got >= 0 && got < a.n_slots) ? 1 : 0;
					if (take) { f_slot = got; f_att = 0; have = 1; }
					const int ntake = __popcll(__ballot(take));
					if (ntake > 0 && lane == 0) atomicAdd(&ctl.outstanding, (unsigned int)ntake);
				}
			}
			const unsigned long long mReq = __ballot(have);
			if (mReq == 0ull) {
				if (!fresh_left && pc3_load(&ctl.outstanding) == 0u) { if (lane == 0) pc3_store(&ctl.done, 1u); return 2; }
				polls++;
				l_req[lane] = 0ull;
				return 1;
			}
			/* room for every photon that may enter */
			int myfree = 0;
			if (lane < PC3_CONSUMERS) myfree = PC3_CAP - (int)(ctl.q_tail[lane] - pc3_load(&ctl.q_head[lane]));
			int cum = myfree;                 /* inclusive prefix sum over the first PC3_CONSUMERS lanes */
#pragma unroll
			for (int off = 1; off < 16; off <<= 1) {
				const int v = __shfl_up(cum, off, PC_WAVE);
				if (lane >= off) cum += v;
			}
			const int total_free = __shfl(cum, PC3_CONSUMERS - 1, PC_WAVE);
			const int nreq = __popcll(mReq);
			{
				/* a batch is worth its ~1000 instructions when a ring's worth of photons fits (the lanes with the lowest ranks go) */
				const int need = (polls > 64) ? 1 : ((nreq < PC3_CAP) ? nreq : PC3_CAP);     /* waited long: whatever fits */
				if (total_free < need) { polls++; l_req[lane] = have ? ((((unsigned long long)f_slot << 24) | (unsigned long long)(f_att & 0xffffffu)) + 1ull) : 0ull; return 1; }
			}
			const int go = have && (__popcll(mReq & below) < total_free);
			st_batches++;
			polls = 0;                         /* the limit is on consecutive polls without progress */
			/* ---------------- launch */
			int entered = 0, f_ne = 0, f_fail = 0;
			pc_photon<1> np;
			np.wmem = nullptr; np.wstride = 1;
			if (go) {
				pc_start s;
				pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + f_slot), f_att, s);
				const int st = pc_launch_init(T, Pm, np, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
				if (st == PC_ST_MARCH) {
					entered = 1;
					if (a.keep_images && !a.img_cursor) {
						/* src/polycap-source.c:779-798 */
						const double cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
						const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
						double *r = a.img + f_slot*ss;
						r[PC_F_SRCX*fs] = s.srcx; r[PC_F_SRCY*fs] = s.srcy;
						r[PC_F_STARTX*fs] = s.x; r[PC_F_STARTY*fs] = s.y;
						r[PC_F_SDIRX*fs] = s.dx; r[PC_F_SDIRY*fs] = s.dy;
						double tx = s.ex*c_ae + s.dx*c_be, ty = s.ey*c_ae + s.dy*c_be, tz = s.ez*c_ae + s.dz*c_be;
						pc_norm3(tx, ty, tz);
						r[PC_F_SEVX*fs] = round(tx); r[PC_F_SEVY*fs] = round(ty);
						r[PC_F_EEVX*fs] = cosalpha0;      /* parked here until the photon leaves the optic */
					}
				} else {
					if (np.rc == 2) f_ne = 1;
					f_att++;
					if (f_att >= a.max_attempts) {
						f_fail = 1;
						have = 0;
						if (a.keep_images && !a.img_cursor) { a.img_w[f_slot*ws] = 0.; a.img[f_slot*ss + PC_F_EEVX*fs] = 0.; }
					}
				}
			}
			u_launch += (unsigned long long)__popcll(__ballot(go));
			u_not_entered += (unsigned long long)__popcll(__ballot(f_ne));
			u_failed += (unsigned long long)__popcll(__ballot(f_fail));
			{
				const int nfail = __popcll(__ballot(f_fail));
				if (nfail > 0 && lane == 0) atomicSub(&ctl.outstanding, (unsigned int)nfail);
			}
			/* entered photons into the rings.  Photons of boundary capillaries (about one in a hundred on a 200 000-capillary
			 * optic) go to the LAST ring while it has room: the hexagon tests they need in every march step and segment visit
			 * are then executed by one tracing wave instead of by every wave that happens to hold one.  The r-th of the others
			 * goes to the ring whose share of the remaining free places holds r. */
			const int f_last = __shfl(myfree, PC3_CONSUMERS - 1, PC_WAVE);
			const unsigned long long mB = __ballot(entered && PC3_ROUTE_BND && np.bnd);
			const int nb_last = min(__popcll(mB), f_last);
			const int rb = __popcll(mB & below);
			const int to_last = entered && PC3_ROUTE_BND && np.bnd && rb < nb_last;
			const unsigned long long mOth = __ballot(entered && !to_last);
			const int r = __popcll(mOth & below);
			int my_c = -1, my_pos = 0;
#pragma unroll
			for (int c = 0; c < PC3_CONSUMERS; c++) {
				const int hi = __shfl(cum, c, PC_WAVE) - ((c == PC3_CONSUMERS - 1) ? nb_last : 0);
				const int lo = (c == 0) ? 0 : __shfl(cum, c - 1, PC_WAVE);
				if (entered && !to_last && my_c < 0 && r >= lo && r < hi) { my_c = c; my_pos = r - lo + ((c == PC3_CONSUMERS - 1) ? nb_last : 0); }
			}
			if (to_last) { my_c = PC3_CONSUMERS - 1; my_pos = rb; }
			if (entered && my_c >= 0) {
				const unsigned int e = (ctl.q_tail[my_c] + (unsigned)my_pos) % PC3_CAP;
				double *q = l_ring + (size_t)my_c*(PC3_FIELDS*PC3_CAP) + e;
				q[0*PC3_CAP] = np.Px; q[1*PC3_CAP] = np.Py;
				q[2*PC3_CAP] = np.dx; q[3*PC3_CAP] = np.dy; q[4*PC3_CAP] = np.dz;
				q[5*PC3_CAP] = np.ex; q[6*PC3_CAP] = np.ey; q[7*PC3_CAP] = np.ez;
				q[8*PC3_CAP] = np.kn;
				q[9*PC3_CAP] = __longlong_as_double((long long)(((unsigned long long)f_slot << 24) | (unsigned long long)(f_att & 0xffffffu)));
				q[10*PC3_CAP] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)np.qr << 1) | (unsigned long long)(np.bnd & 1)));
				have = 0;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			{
				/* publish: ring c received min(n_entered, cum[c]) - min(n_entered, cum[c-1]) photons */
				const int ne_tot = __popcll(mOth);
				const int prev = __shfl_up(cum, 1, PC_WAVE);
				if (lane < PC3_CONSUMERS) {
					const int last = (lane == PC3_CONSUMERS - 1);
					const int lo = (lane == 0) ? 0 : prev;
					const int hi = cum - (last ? nb_last : 0);
					const int n_c = ((ne_tot < hi) ? ne_tot : hi) - ((ne_tot < lo) ? ne_tot : lo) + (last ? nb_last : 0);
					if (n_c > 0) pc3_store(&ctl.q_tail[lane], ctl.q_tail[lane] + (unsigned)n_c);
				}
			}
		}
		l_req[lane] = have ? ((((unsigned long long)f_slot << 24) | (unsigned long long)(f_att & 0xffffffu)) + 1ull) : 0ull;
		return 0;
	};

	if (wave == 0) {
		/* ================================================================ the launching wave */
		for (;;) {
			const int r = producer_step();
			if (r >= 2) break;
			if (r == 1) __builtin_amdgcn_s_sleep(PC3_SLEEP);
		}
	} else {
		/* ================================================================ a tracing wave */
		const int c = wave - 1;
		double *ring = l_ring + (size_t)c*(PC3_FIELDS*PC3_CAP);
		double *dring = l_done + (size_t)c*(PC3_DFIELDS*PC3_DCAP);
		pc_photon<1> ph;
		ph.wmem = nullptr; ph.wstride = 1; ph.wset = 0; ph.rc = 0; ph.qr = 0; ph.first = 0; ph.lv = 0; ph.bnd = 0; ph.i = 0; ph.irefl = 0;
		ph.Px = ph.Py = ph.Pz = ph.dx = ph.dy = ph.dz = ph.ex = ph.ey = ph.ez = ph.dtravel = ph.C0 = 0.; ph.w[0] = 0.;
		ph.kx = ph.ky = ph.kn = ph.sx = ph.sy = ph.ox = ph.oy = ph.idzd = 0.;
		int state = LS_NEED_SLOT;
		long long slot = 0;
		unsigned int attempt = 0;
		unsigned int q_head = 0, r_tail = 0;      /* this wave's ends of its two rings */
		auto march_burst = [&](int nM, bool do_new, int nE) {
			if (state == LS_MARCH && ph.first)
				state = pc_march_step(T, Pm, ph);
			for (int b = 0; b < a.march_burst; b++) {
				unsigned int lanes_in_burst = 0;
#pragma unroll
				for (int u = 0; u < PC_MARCH_UNROLL; u++) {
					if (STATS) lanes_in_burst += (unsigned)__popcll(__ballot(state == LS_MARCH));
					if (state == LS_MARCH)
						state = pc_march_step_hot(T, Pm, ph);
				}
				const int cM = __popcll(__ballot(state == LS_MARCH));
				if (STATS) { st_march += PC_MARCH_UNROLL; st_march_l += lanes_in_burst; }
				if (cM == 0) break;
				if (cM < a.march_stop && (cM != nM || do_new || nE > 0)) break;
			}
		};
		for (;;) {
			const int nM = __popcll(__ballot(state == LS_MARCH)), nE = __popcll(__ballot(state == LS_EVENT));
			const int nD = __popcll(__ballot(state == LS_DONE)), nQ = __popcll(__ballot(state == LS_NEED_SLOT));
			const int avail = (int)(pc3_load(&ctl.q_tail[c]) - q_head);
			if (nM + nE + nD == 0) {
				if (avail == 0) {
					/* nothing to trace: the launching wave says when to leave (it ends the run, or gives up after PC3_MAX_POLLS
					 * polls without progress and marks the run failed) */
					if (pc3_load(&ctl.done) || pc3_load(&ctl.failed)) break;
					__builtin_amdgcn_s_sleep(32);
					continue;
				}
			}
			const int nN = nD + ((nQ < avail) ? nQ : avail);
			const bool do_new = (nN >= a.new_threshold) || (nM == 0 && nE == 0);
			if (nM > 0 && (nM >= a.event_threshold || (nE == 0 && !(do_new && nN > 0))) && !(nN >= a.pool_event_min)) {
				/* ---------------- MARCH */
				march_burst(nM, do_new, nE);
			} else if (nE > 0 && !(do_new && nN > nE) && !(nN >= a.pool_event_min)) {
				/* ---------------- EVENT */
				st_event += 1; st_event_l += (unsigned)nE;
				if (state == LS_EVENT)
					state = pc_event<1, true>(T, Pm, a.ec, ph);
			} else if (nN > 0) {
				/* ---------------- NEW: finalise finished photons, pop launched ones */
				st_new += 1; st_new_l += (unsigned)nN;
				int moved = 0;                /* photons pushed or popped in this phase */
				{
					/* finished photons go to the launching wave (as many as its ring takes; the others wait for the next NEW phase) */
					const unsigned long long mD = __ballot(state == LS_DONE);
					if (mD) {
						const int room = PC3_DCAP - (int)(r_tail - pc3_load(&ctl.r_head[c]));
						const int rk = __popcll(mD & below);
						if (state == LS_DONE && rk < room) {
							double *q = dring + ((r_tail + (unsigned)rk) % PC3_DCAP);
							q[0*PC3_DCAP] = ph.Px; q[1*PC3_DCAP] = ph.Py; q[2*PC3_DCAP] = ph.Pz;
							q[3*PC3_DCAP] = ph.dx; q[4*PC3_DCAP] = ph.dy; q[5*PC3_DCAP] = ph.dz;
							q[6*PC3_DCAP] = ph.ex; q[7*PC3_DCAP] = ph.ey; q[8*PC3_DCAP] = ph.ez;
							q[9*PC3_DCAP] = ph.dtravel; q[10*PC3_DCAP] = ph.w[0];
							q[11*PC3_DCAP] = __longlong_as_double((long long)(((unsigned long long)slot << 24) | (unsigned long long)(attempt & 0xffffffu)));
							q[12*PC3_DCAP] = __longlong_as_double((long long)(((unsigned long long)(unsigned int)ph.irefl << 8) | (unsigned long long)((ph.rc + 2) & 0xff)));
							state = LS_NEED_SLOT;
						}
						const int k = __popcll(mD);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						r_tail += (unsigned)(k < room ? k : room);
						moved += (k < room ? k : room);
						if (lane == 0) pc3_store(&ctl.r_tail[c], r_tail);
					}
				}
				{
					/* ---------------- pop launched photons */
					const unsigned long long mQ = __ballot(state == LS_NEED_SLOT);
					const int av = (int)(pc3_load(&ctl.q_tail[c]) - q_head);
					if (mQ && av > 0) {
						const int rk = __popcll(mQ & below);
						if (state == LS_NEED_SLOT && rk < av) {
							const unsigned int e = (q_head + (unsigned)rk) % PC3_CAP;
							const double *q = ring + e;
							ph.Px = q[0*PC3_CAP]; ph.Py = q[1*PC3_CAP]; ph.Pz = 0.;
							ph.dx = q[2*PC3_CAP]; ph.dy = q[3*PC3_CAP]; ph.dz = q[4*PC3_CAP];
							ph.ex = q[5*PC3_CAP]; ph.ey = q[6*PC3_CAP]; ph.ez = q[7*PC3_CAP];
							ph.kn = q[8*PC3_CAP];
							const unsigned long long w0 = (unsigned long long)__double_as_longlong(q[9*PC3_CAP]);
							const unsigned long long w1 = (unsigned long long)__double_as_longlong(q[10*PC3_CAP]);
							slot = (long long)(w0 >> 24);
							attempt = (unsigned int)(w0 & 0xffffffull);
							ph.qr = (int)(unsigned int)(w1 >> 1);
							ph.bnd = (int)(w1 & 1ull);
							/* what pc_launch_init left besides: pc_axis_setup by its expressions, a fresh photon at node 0 */
							const double q_i = (double)((int)((unsigned int)ph.qr >> 16) - 32768), r_i = (double)((int)((unsigned int)ph.qr & 0xffffu) - 32768);
							ph.ky = r_i * (3./2);
							ph.kx = (2.*q_i + r_i) * PC_COSPI_6;
							ph.irefl = 0; ph.dtravel = 0.; ph.rc = 0; ph.C0 = 0.; ph.wset = 0; ph.w[0] = 1.; ph.i = 0;
							pc_trace_begin(ph);
							state = LS_MARCH;
						}
						const int k = __popcll(mQ);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     /* the entries are read before the places are given back */
						q_head += (unsigned)(k < av ? k : av);
						moved += (k < av ? k : av);
						if (lane == 0) pc3_store(&ctl.q_head[c], q_head);
					}
				}
				if (moved == 0 && nM == 0 && nE == 0) {
					/* nothing could move (the ring of finished photons is full and nothing is there to pop) and nothing else can
					 * run: wait for the launching wave -- or leave when it has given up (this state used to spin, and
					 * to spin forever after a give-up) */
					if (pc3_load(&ctl.failed)) break;
					__builtin_amdgcn_s_sleep(8);
				}
			} else {
				/* lanes wait for launched photons and nothing else can run */
				if (pc3_load(&ctl.failed)) break;
				__builtin_amdgcn_s_sleep(8);
			}
		}
	}

	if (lane == 0) {
		if (u_exit) atomicAdd(&a.totals->counters[0], u_exit);
		if (u_not_entered) atomicAdd(&a.totals->counters[1], u_not_entered);
		if (u_not_trans) atomicAdd(&a.totals->counters[2], u_not_trans);
		if (u_irefl) atomicAdd(&a.totals->counters[3], u_irefl);
		if (u_failed) atomicAdd(&a.totals->counters[4], u_failed);
		if (u_launch) atomicAdd(&a.totals->counters[5], u_launch);
		if (wave != 0) {
			atomicAdd(&a.totals->phase[0], st_march); atomicAdd(&a.totals->phase[1], st_march_l);
			atomicAdd(&a.totals->phase[2], st_event); atomicAdd(&a.totals->phase[3], st_event_l);
			atomicAdd(&a.totals->phase[4], st_new); atomicAdd(&a.totals->phase[5], st_new_l);
		}
		if (u_acc_lo | u_acc_hi) pc_atomic_add128(a.sumw, u_acc_lo, u_acc_hi);
		if (wave == 0) atomicAdd(&a.totals->phase[6], st_batches);
	}
	__syncthreads();
	/* a wave that gave up waiting: the run is reported as failed (more failed slots than the run has slots) */
	if (threadIdx.x == 0 && ctl.failed) atomicAdd(&a.totals->counters[4], (unsigned long long)a.n_slots + 1ull);
}

#endif /* PC_PRODUCER_KERNEL_H */
