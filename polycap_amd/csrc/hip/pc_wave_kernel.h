/* EXPERIMENT (option "wave_per_photon", never chosen by itself): the mapping BASELINE.json's north_star words -- one
 * wavefront owns one photon -- next to the one the product uses (one lane per photon, phases scheduled wave-wide).
 *
 * A wave takes one slot at a time.  The photon's state is wave-uniform (every lane holds the same values and executes the
 * same launch / EVENT code: that arithmetic has nothing to distribute over lanes); the MARCH is where 64 lanes help: lane l
 * evaluates the single-segment certificate value at node i + 1 + l (pc_node_C, the same 6 FMAs as pc_march_ok at stride 1), a
 * ballot + ffs finds the first node that is not certified and the photon goes to EVENT at the segment in front of it.  The
 * literal visits, hence every photon and the exact totals, are those of the lane kernels (certified skipping at any stride
 * visits the same segments: tests/test_gpu_parity.py), which the A/B script checks.  Histogram only (no image planes),
 * single energy, source modes.  Measured: profiles/r03/wave_per_photon_ab.txt. */
#ifndef PC_WAVE_KERNEL_H
#define PC_WAVE_KERNEL_H

#define PCW_BLOCK 256

template <int MODE>
__global__ void __launch_bounds__(PCW_BLOCK, 4)
pc_trace_wave_kernel(pc_kargs a)
{
	__shared__ double lds[6*1024];
	__shared__ pc_marg4 ldsg[1024];
	const int npts = a.pm.nmax + 1;
	double *l_z = lds, *l_cap = lds + 1024, *l_zh = lds + 2*1024, *l_cap2 = lds + 3*1024, *l_hexd = lds + 4*1024, *l_idz = lds + 5*1024;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		l_z[k] = a.g_z[k]; l_cap[k] = a.g_cap[k]; l_zh[k] = a.g_zh[k]; l_cap2[k] = a.g_cap2[k];
		l_hexd[k] = a.g_hexd[k]; l_idz[k] = a.g_idz[k]; ldsg[k] = a.g_mg[k];
	}
	__syncthreads();
	pc_tables T;
	T.z = l_z; T.cap = l_cap; T.zh = l_zh; T.cap2 = l_cap2; T.hexd = l_hexd; T.idz = l_idz; T.ext = a.g_ext; T.mg = ldsg;
	const pc_params &Pm = a.pm;
	const int lane = threadIdx.x & (PC_WAVE - 1);
	const int nmax = Pm.nmax;
	const float adjf = Pm.adjf;
	unsigned long long u_exit = 0, u_not_entered = 0, u_not_trans = 0, u_irefl = 0, u_failed = 0, u_launch = 0;
	unsigned long long u_acc_lo = 0, u_acc_hi = 0;
	unsigned long long st_scan = 0, st_event = 0;
	for (;;) {
		long long slot = 0;
		if (lane == 0) slot = (long long)atomicAdd(a.work, 1ull);
		slot = __shfl(slot, 0, PC_WAVE);
		if (slot >= a.n_slots) break;
		for (unsigned int attempt = 0;; attempt++) {
			if (attempt >= a.max_attempts) { u_failed++; break; }
			pc_photon<1> ph;
			ph.wmem = nullptr; ph.wstride = 1;
			pc_start s;
			pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), attempt, s);
			int st = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
			u_launch++;
			while (st != PC_ST_DONE) {
				if (st == PC_ST_EVENT) {
					st_event++;
					st = pc_event<1, true>(T, Pm, a.ec, ph);
					continue;
				}
				if (ph.i >= nmax) { ph.rc = 1; st = PC_ST_DONE; break; }
				if (Pm.literal) { st = PC_ST_EVENT; continue; }
				if (ph.first || ph.bnd) {
					/* the segment that holds the last interaction point, and boundary capillaries (hexagon tests at every node): as a lane does them */
					ph.lv = 0;
					st = ph.first ? (pc_march_first_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT) : (pc_march_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT);
					continue;
				}
				/* 64 nodes at a time */
				st_scan++;
				const int i0 = ph.i;
				if (!((float)ph.C0 < -adjf)) { st = PC_ST_EVENT; continue; }
				const int node = i0 + 1 + lane;
				const int valid = node <= nmax;
				const double Cn = valid ? pc_node_C(T, ph, node) : 0.;
				const unsigned long long mValid = __ballot(valid);
				const unsigned long long mBad = __ballot(valid && !((float)Cn < -adjf));
				if (mBad == 0ull) {
					const int nv = __popcll(mValid);
					ph.i = i0 + nv;
					ph.C0 = __shfl(Cn, nv - 1, PC_WAVE);
				} else {
					const int f = __ffsll((long long)mBad) - 1;     /* node i0 + 1 + f is the first that is not certified */
					ph.i = i0 + f;
					if (f > 0) ph.C0 = __shfl(Cn, f - 1, PC_WAVE);
					st = PC_ST_EVENT;
				}
			}
			/* src/polycap-source.c:758-777 */
			int ok = 0;
			if (ph.rc == 0) u_not_trans++;
			else if (ph.rc == 2) u_not_entered++;
			else if (ph.rc == 1) ok = pc_in_exit_window(Pm, ph);
			if (ok) {
				u_exit++;
				u_irefl += (unsigned long long)ph.irefl;
				const unsigned long long f = (unsigned long long)(ph.w[0] * PC_FIX_SCALE);
				const unsigned long long old = u_acc_lo;
				u_acc_lo += f;
				if (u_acc_lo < old) u_acc_hi++;
				break;
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
		atomicAdd(&a.totals->phase[0], st_scan); atomicAdd(&a.totals->phase[1], st_scan*64ull);
		atomicAdd(&a.totals->phase[2], st_event); atomicAdd(&a.totals->phase[3], st_event);
		if (u_acc_lo | u_acc_hi) pc_atomic_add128(a.sumw, u_acc_lo, u_acc_hi);
	}
}

#endif /* PC_WAVE_KERNEL_H */
