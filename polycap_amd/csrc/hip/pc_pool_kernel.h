/* Single-energy trace kernel with a per-wave photon pool in LDS (included by pc_kernels.hip).  Option "pool", on by
 * default; the one-photon-per-lane kernel pc_trace_kernel<1, MODE> serves what this one does not (pc_pool_applies).
 *
 * pc_trace_kernel keeps one photon per lane, so a MARCH phase runs with the lanes that happen to be in flight (about 20
 * of 64 on xos1: flights are a few steps long and every lane then waits for an EVENT phase).  Here every wave owns
 * PQ_P more photons parked in LDS.  Before a phase of class X (MARCH / EVENT / NEW) the lanes whose photon is not in X
 * exchange it with a parked photon that is, so phases run with fuller waves: on xos1 the wave-level MARCH steps drop by
 * a third and the EVENT and NEW phases by almost half.  A photon's result depends on (seed, slot, attempt) only and
 * the totals are exact integers, so the output is bit-identical to pc_trace_kernel's (tests/test_gpu_parity.py).
 *
 * A parked photon is PQ_R doubles (position, direction, electric vector, d_travel, weight, 1/dz, |k|, two packed
 * words) + 16 bits of state; the ray constants, the axis factors and the certificate value are recomputed on load with
 * the expressions that made them.  Source modes only (the explicit-photon launch has no retry loop to feed a pool).
 *
 * Measured on MI355X (scripts/ab_pool.sh, xos1 10 keV, 4e6 slots; one-photon-per-lane kernel: 15.2 ms):
 *   1024 threads/CU (128 VGPRs, 195 spilled), 48 parked: 55 ms;  512 threads/CU (212 VGPRs, no spills), 64 parked: 15.7 ms;
 *   768 threads/CU (168 VGPRs, 40 spilled), 64 parked: 14.3 ms  <- the build default.
 * The fuller phases are paid for with the registers of the exchange code and fewer resident waves (the lane kernel
 * itself is 1.6x slower at 12 waves per CU than at 16), so the net gain was 5 % with the march of that time and is 6 %
 * on the full benchmark (1e7 slots: 28.8 ms against 31.3 ms, scripts/ab_pool_default.sh) with the current one. */
#ifndef PC_POOL_KERNEL_H
#define PC_POOL_KERNEL_H

#ifndef PQ_BLOCK
#define PQ_BLOCK 768       /* one workgroup per CU */
#endif
#ifndef PQ_MIN_WAVES
#define PQ_MIN_WAVES 3     /* waves per SIMD the register allocator leaves room for */
#endif
#define PQ_WAVES (PQ_BLOCK / PC_WAVE)
#define PQ_PITCH 1024
#ifndef PQ_P
#define PQ_P 64            /* parked photons per wave (at most 64: one mask bit each) */
#endif
#ifndef PQ_UNROLL
#define PQ_UNROLL 16        /* march steps between two top-ups of the wave (2: 32.0 ms, 4: 30.3, 8: 29.4, 12-24: 29.0, 32: 29.6:
                             * topping up in mid-burst does not pay, the pool earns its keep in the EVENT and NEW phases) */
#endif
#define PQ_R 15            /* doubles per parked photon */

enum { PQ_PX = 0, PQ_PY, PQ_PZ, PQ_DX, PQ_DY, PQ_DZ, PQ_EX, PQ_EY, PQ_EZ, PQ_DTRAVEL, PQ_W, PQ_IDZ, PQ_KN, PQ_I0, PQ_I1 };

__device__ __forceinline__ int pq_class(int s)
{
	return (s == LS_MARCH) ? 0 : ((s == LS_EVENT) ? 1 : ((s == LS_IDLE) ? 3 : 2));
}

struct pq_lane {               /* what a lane holds besides the photon */
	int state;
	long long slot;
	unsigned int attempt;
};

/* pool class masks from the per-entry state words (lanes < PQ_P look at one entry each) */
__device__ __forceinline__ void pq_masks(const unsigned short *est, int lane, unsigned long long &pM, unsigned long long &pE, unsigned long long &pN)
{
	const int s = (lane < PQ_P) ? (int)(est[lane] & 15) : (int)LS_IDLE;
	pM = __ballot(s == LS_MARCH);
	pE = __ballot(s == LS_EVENT);
	pN = __ballot(s == LS_DONE || s == LS_NEED_SLOT || s == LS_START);
}

/* lanes whose photon is not of class X take the parked photons of class X (mask poolX) and park their own */
__device__ __forceinline__ void pq_swap_in(const pc_tables &T, int X, unsigned long long poolX, pc_photon<1> &ph, pq_lane &L,
                                           double *pool, unsigned short *est, unsigned char *sel, int lane)
{
	const unsigned long long takers = __ballot(pq_class(L.state) != X);
	const int nt = __popcll(takers), np = __popcll(poolX);
	const int n = nt < np ? nt : np;
	if (n == 0) return;
	const unsigned long long below = (1ull << lane) - 1ull;
	if ((poolX >> lane) & 1ull) {
		const int rk = __popcll(poolX & below);
		if (rk < n) sel[rk] = (unsigned char)lane;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int rk = __popcll(takers & below);
	if (((takers >> lane) & 1ull) && rk < n) {
		const int e = sel[rk];
		/* exchange field by field: one temporary at a time */
#define PQ_XCHG(F, V) { const double t_ = pool[(F)*PQ_P + e]; pool[(F)*PQ_P + e] = (V); (V) = t_; }
		PQ_XCHG(PQ_PX, ph.Px) PQ_XCHG(PQ_PY, ph.Py) PQ_XCHG(PQ_PZ, ph.Pz)
		PQ_XCHG(PQ_DX, ph.dx) PQ_XCHG(PQ_DY, ph.dy) PQ_XCHG(PQ_DZ, ph.dz)
		PQ_XCHG(PQ_EX, ph.ex) PQ_XCHG(PQ_EY, ph.ey) PQ_XCHG(PQ_EZ, ph.ez)
		PQ_XCHG(PQ_DTRAVEL, ph.dtravel) PQ_XCHG(PQ_W, ph.w[0]) PQ_XCHG(PQ_IDZ, ph.idzd) PQ_XCHG(PQ_KN, ph.kn)
#undef PQ_XCHG
		const unsigned int es = est[e];
		est[e] = (unsigned short)((L.state & 15) | ((ph.first & 1) << 4) | ((ph.lv & 3) << 5) | ((ph.bnd & 1) << 7) | (((ph.rc + 2) & 7) << 8));
		double w0 = __longlong_as_double((long long)(((unsigned long long)L.slot << 24) | (unsigned long long)(L.attempt & 0xffffffu)));
		double w1 = __longlong_as_double((long long)(((unsigned long long)(unsigned int)ph.qr << 32)
		            | ((unsigned long long)(ph.irefl & 0xffff) << 16) | (unsigned long long)(ph.i & 0xffff)));
		{ const double t_ = pool[PQ_I0*PQ_P + e]; pool[PQ_I0*PQ_P + e] = w0; w0 = t_; }
		{ const double t_ = pool[PQ_I1*PQ_P + e]; pool[PQ_I1*PQ_P + e] = w1; w1 = t_; }
		const unsigned long long i0 = (unsigned long long)__double_as_longlong(w0);
		const unsigned long long i1 = (unsigned long long)__double_as_longlong(w1);
		L.slot = (long long)(i0 >> 24);
		L.attempt = (unsigned int)(i0 & 0xffffffull);
		ph.i = (int)(i1 & 0xffffull);
		ph.irefl = (int)((i1 >> 16) & 0xffffull);
		ph.qr = (int)(unsigned int)(i1 >> 32);
		L.state = (int)(es & 15u);
		ph.first = (int)((es >> 4) & 1u); ph.lv = (int)((es >> 5) & 3u); ph.bnd = (int)((es >> 7) & 1u);
		ph.rc = (int)((es >> 8) & 7u) - 2;
		/* derived values, by the expressions that made them (pc_axis_setup, pc_ray_setup, pc_node_C); 1/dz and |k| travel */
		{
			const double q_i = (double)((int)((unsigned int)ph.qr >> 16) - 32768), r_i = (double)((int)((unsigned int)ph.qr & 0xffffu) - 32768);
			ph.ky = r_i * (3./2);
			ph.kx = (2.*q_i + r_i) * PC_COSPI_6;
			ph.sx = ph.dx * ph.idzd;
			ph.sy = ph.dy * ph.idzd;
			ph.ox = ph.Px - ph.sx * ph.Pz;
			ph.oy = ph.Py - ph.sy * ph.Pz;
			ph.C0 = pc_node_C(T, ph, ph.i);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__global__ void __launch_bounds__(PQ_BLOCK, PQ_MIN_WAVES)
pc_trace_pool_kernel(pc_kargs a)
{
	__shared__ double lds[6*PQ_PITCH];
	__shared__ pc_marg4 ldsg[PQ_PITCH];
	__shared__ double l_pool[PQ_WAVES*PQ_R*PQ_P];
	__shared__ unsigned short l_est[PQ_WAVES*PC_WAVE];
	__shared__ unsigned char l_sel[PQ_WAVES*PC_WAVE];
	const int npts = a.pm.nmax + 1;
	double *l_z = lds, *l_cap = lds + PQ_PITCH, *l_zh = lds + 2*PQ_PITCH, *l_cap2 = lds + 3*PQ_PITCH, *l_hexd = lds + 4*PQ_PITCH, *l_idz = lds + 5*PQ_PITCH;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		l_z[k] = a.g_z[k];
		l_cap[k] = a.g_cap[k];
		l_zh[k] = a.g_zh[k];
		l_cap2[k] = a.g_cap2[k];
		l_hexd[k] = a.g_hexd[k];
		l_idz[k] = a.g_idz[k];
		ldsg[k] = a.g_mg[k];
	}
	const int lane = threadIdx.x & (PC_WAVE - 1);
	const int wave = threadIdx.x / PC_WAVE;
	double *pool = l_pool + wave*PQ_R*PQ_P;
	unsigned short *est = l_est + wave*PC_WAVE;
	unsigned char *sel = l_sel + wave*PC_WAVE;
	est[lane] = (unsigned short)((lane < PQ_P) ? LS_NEED_SLOT : LS_IDLE);    /* every parked place starts by asking for a slot */
	__syncthreads();
	pc_tables T;
	T.z = l_z; T.cap = l_cap; T.zh = l_zh; T.cap2 = l_cap2; T.hexd = l_hexd; T.idz = l_idz; T.ext = a.g_ext;
	T.mg = ldsg;
	const long long fs = a.img_fs, ss = a.img_ss, ws = a.img_ws;      /* strides of the image store: records or planes (pc_kargs) */
	const pc_params &Pm = a.pm;

	pc_photon<1> ph;
	ph.wmem = nullptr; ph.wstride = 1; ph.wset = 0; ph.rc = 0; ph.qr = 0; ph.first = 0; ph.lv = 0; ph.bnd = 0; ph.i = 0; ph.irefl = 0;
	ph.Px = ph.Py = ph.Pz = ph.dx = ph.dy = ph.dz = ph.ex = ph.ey = ph.ez = ph.dtravel = ph.C0 = 0.; ph.w[0] = 0.;
	pq_lane L;
	L.state = LS_NEED_SLOT; L.slot = 0; L.attempt = 0;
	long long chunk_next = 0, chunk_end = 0;
	unsigned long long u_exit = 0, u_not_entered = 0, u_not_trans = 0, u_irefl = 0, u_failed = 0, u_launch = 0;
	unsigned long long u_acc_lo = 0, u_acc_hi = 0;
	unsigned long long st_march = 0, st_march_l = 0, st_event = 0, st_event_l = 0, st_new = 0, st_new_l = 0, st_swap = 0;
	unsigned long long pM, pE, pN;
	pq_masks(est, lane, pM, pE, pN);

	for (;;) {
		const int nM = __popcll(__ballot(L.state == LS_MARCH)), nE = __popcll(__ballot(L.state == LS_EVENT));
		const int nN = __popcll(__ballot(L.state == LS_DONE || L.state == LS_NEED_SLOT || L.state == LS_START));
		const int tM = nM + __popcll(pM), tE = nE + __popcll(pE), tN = nN + __popcll(pN);
		if (tM + tE + tN == 0) break;
		/* Which class runs next.  MARCH while the wave is (nearly) full of marchers or can be topped up from the pool; else a
		 * full EVENT or NEW phase if one is ready; else MARCH with what is left down to march_min lanes; else whatever waits. */
		int X;
		if (nM >= PC_WAVE - a.pool_refill || (tM > 0 && pM != 0ull)) X = 0;
		else if (tE >= a.pool_event_min) X = 1;
		else if (tN >= a.new_threshold) X = 2;
		else if (nM >= a.event_threshold || (tM > 0 && tE + tN == 0)) X = 0;
		else X = (tE > 0 && tE >= tN) ? 1 : ((tN > 0) ? 2 : ((tE > 0) ? 1 : 0));
		/* lanes whose photon is of another class exchange it with a parked photon of this class */
		{
			const unsigned long long pX = (X == 0) ? pM : ((X == 1) ? pE : pN);
			const int nX = (X == 0) ? nM : ((X == 1) ? nE : nN);
			if (pX != 0ull && nX <= PC_WAVE - ((X == 0) ? a.pool_refill : 1)) {
				pq_swap_in(T, X, pX, ph, L, pool, est, sel, lane);
				pq_masks(est, lane, pM, pE, pN);
				st_swap++;
			}
		}
		if (X == 0) {
			/* ---------------- MARCH: PQ_UNROLL certified steps */
			if (L.state == LS_MARCH && ph.first)
				L.state = pc_march_step(T, Pm, ph);
			unsigned int lanes_in_burst = 0;     /* lanes that take each of the burst's steps (scheduler statistics) */
#pragma unroll
			for (int u = 0; u < PQ_UNROLL; u++) {
				if ((u & 3) == 0) lanes_in_burst += 4u*(unsigned)__popcll(__ballot(L.state == LS_MARCH));   /* sampled every 4th step */
				if (L.state == LS_MARCH)
					L.state = pc_march_step_hot(T, Pm, ph);
			}
			st_march += PQ_UNROLL; st_march_l += lanes_in_burst;
		} else if (X == 1) {
			/* ---------------- EVENT */
			st_event += 1; st_event_l += (unsigned)__popcll(__ballot(L.state == LS_EVENT));
			if (L.state == LS_EVENT)
				L.state = pc_event<1, true>(T, Pm, a.ec, ph);
			/* The wave is now full of flights that have just begun, most of them a few steps long: their first steps are taken
			 * here, at the EVENT phase's lane count, instead of in a MARCH burst after an exchange with the pool. */
			if (a.event_march > 0) {
				if (L.state == LS_MARCH && ph.first)
					L.state = pc_march_step(T, Pm, ph);
				unsigned int lanes = 0;
				int u = 0;
				for (; u < a.event_march; u++) {
					const unsigned int nm = (unsigned)__popcll(__ballot(L.state == LS_MARCH));
					if (nm == 0) break;
					lanes += nm;
					if (L.state == LS_MARCH)
						L.state = pc_march_step_hot(T, Pm, ph);
				}
				st_march += (unsigned)u; st_march_l += lanes;
			}
		} else {
			/* ---------------- NEW: finalise finished photons, hand out slots, sample + entrance tests */
			st_new += 1; st_new_l += (unsigned)__popcll(__ballot(L.state == LS_DONE || L.state == LS_NEED_SLOT || L.state == LS_START));
			int f_exit = 0, f_not_entered = 0, f_not_trans = 0, f_failed = 0, f_launch = 0;
			unsigned int f_irefl = 0;
			unsigned long long f_w = 0;
			if (L.state == LS_DONE) {
				/* src/polycap-source.c:758-777 */
				const int rc = ph.rc;
				const long long slot = L.slot;
				int ok = 0;
				if (rc == 0) f_not_trans = 1;
				else if (rc == 2) f_not_entered = 1;
				else if (rc == 1) ok = pc_in_exit_window(Pm, ph);
				if (ok) {
					f_exit = 1;
					f_irefl = (unsigned int)ph.irefl;
					const double w = ph.w[0];
					f_w = (unsigned long long)(w * PC_FIX_SCALE);
					if (a.keep_images) {
						/* src/polycap-source.c:900-923; cos(alpha) of the start vectors was left in the record by the launch */
						double *r = a.img + slot*ss;
						const double cosalpha0 = __longlong_as_double((long long)__hip_atomic_load((unsigned long long *)(r + PC_F_EEVX*fs),
						                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
						a.img_w[slot*ws] = w;
						double t = (Pm.z_end - ph.Pz) / ph.dz;
						double ex = ph.Px + ph.dx*t, ey = ph.Py + ph.dy*t, ez = ph.Pz + ph.dz*t;
						r[PC_F_EXITX*fs] = ex; r[PC_F_EXITY*fs] = ey; r[PC_F_EXITZ*fs] = ez;
						r[PC_F_EDIRX*fs] = ph.dx; r[PC_F_EDIRY*fs] = ph.dy;
						const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
						double tx = ph.ex*c_ae + ph.dx*c_be, ty = ph.ey*c_ae + ph.dy*c_be, tz = ph.ez*c_ae + ph.dz*c_be;
						pc_norm3(tx, ty, tz);
						r[PC_F_EEVX*fs] = round(tx); r[PC_F_EEVY*fs] = round(ty);
						((long long *)r)[PC_F_NREFL*fs] = ph.irefl;
						double lx = ex - ph.Px, ly = ey - ph.Py, lz = Pm.z_end - ph.Pz;
						r[PC_F_DTRAVEL*fs] = ph.dtravel + sqrt(lx*lx + ly*ly + lz*lz);
					}
					L.state = LS_NEED_SLOT;
				} else {
					L.attempt++;
					if (L.attempt >= a.max_attempts) {
						f_failed = 1;
						if (a.keep_images) { a.img_w[slot*ws] = 0.; a.img[slot*ss + PC_F_EEVX*fs] = 0.; }
						L.state = LS_NEED_SLOT;
					} else {
						L.state = LS_START;
					}
				}
			}
			{
				const unsigned long long need = __ballot(L.state == LS_NEED_SLOT);
				if (need) {
					const int k = __popcll(need);
					const int rank = __popcll(need & ((1ull << lane) - 1ull));
					if (chunk_end - chunk_next < k) {
						long long have = chunk_end - chunk_next;
						long long base_new = 0;
						if (lane == 0) base_new = (long long)atomicAdd(a.work, (unsigned long long)PC_CHUNK);
						base_new = __shfl(base_new, 0, PC_WAVE);
						if (L.state == LS_NEED_SLOT)
							L.slot = (rank < have) ? (chunk_next + rank) : (base_new + (rank - have));
						chunk_next = base_new + (k - have);
						chunk_end = base_new + PC_CHUNK;
					} else {
						if (L.state == LS_NEED_SLOT) L.slot = chunk_next + rank;
						chunk_next += k;
					}
					if (L.state == LS_NEED_SLOT) {
						if (L.slot >= a.n_slots) { L.state = LS_IDLE; }
						else { L.attempt = 0; L.state = LS_START; }
					}
				}
			}
			if (L.state == LS_START) {
				f_launch = 1;
				const long long slot = L.slot;
				pc_start s;
				pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), L.attempt, s);
				L.state = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
				if (L.state == LS_MARCH && a.keep_images) {
					/* src/polycap-source.c:779-798 */
					const double cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
					const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
					double *r = a.img + slot*ss;
					r[PC_F_SRCX*fs] = s.srcx; r[PC_F_SRCY*fs] = s.srcy;
					r[PC_F_STARTX*fs] = s.x; r[PC_F_STARTY*fs] = s.y;
					r[PC_F_SDIRX*fs] = s.dx; r[PC_F_SDIRY*fs] = s.dy;
					double tx = s.ex*c_ae + s.dx*c_be, ty = s.ey*c_ae + s.dy*c_be, tz = s.ez*c_ae + s.dz*c_be;
					pc_norm3(tx, ty, tz);
					r[PC_F_SEVX*fs] = round(tx); r[PC_F_SEVY*fs] = round(ty);
					r[PC_F_EEVX*fs] = cosalpha0;      /* parked here until the photon leaves the optic (read back above) */
				}
			}
			u_not_trans += (unsigned long long)__popcll(__ballot(f_not_trans));
			u_not_entered += (unsigned long long)__popcll(__ballot(f_not_entered));
			u_failed += (unsigned long long)__popcll(__ballot(f_failed));
			u_launch += (unsigned long long)__popcll(__ballot(f_launch));
			const unsigned long long mX = __ballot(f_exit);
			if (mX) {
				u_exit += (unsigned long long)__popcll(mX);
				u_irefl += pc_wave_sum_u64((unsigned long long)f_irefl);
				const unsigned long long s_low = pc_wave_sum_u64(f_w & 0xffffffffull), s_high = pc_wave_sum_u64(f_w >> 32);
				const unsigned long long lo = s_low + (s_high << 32);
				const unsigned long long hi = (s_high >> 32) + ((lo < s_low) ? 1ull : 0ull);
				const unsigned long long old = u_acc_lo;
				u_acc_lo = old + lo;
				u_acc_hi += hi + ((u_acc_lo < old) ? 1ull : 0ull);
			}
		}
	}

	if (lane == 0) {
		atomicAdd(&a.totals->counters[0], u_exit);
		atomicAdd(&a.totals->counters[1], u_not_entered);
		atomicAdd(&a.totals->counters[2], u_not_trans);
		atomicAdd(&a.totals->counters[3], u_irefl);
		if (u_failed) atomicAdd(&a.totals->counters[4], u_failed);
		atomicAdd(&a.totals->counters[5], u_launch);
		atomicAdd(&a.totals->phase[0], st_march); atomicAdd(&a.totals->phase[1], st_march_l);
		atomicAdd(&a.totals->phase[2], st_event); atomicAdd(&a.totals->phase[3], st_event_l);
		atomicAdd(&a.totals->phase[4], st_new); atomicAdd(&a.totals->phase[5], st_new_l);
		atomicAdd(&a.totals->phase[6], st_swap);
		pc_atomic_add128(a.sumw, u_acc_lo, u_acc_hi);
	}
}

#endif /* PC_POOL_KERNEL_H */
