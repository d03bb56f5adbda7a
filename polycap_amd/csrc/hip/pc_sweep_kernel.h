/*
 * pc_sweep_kernel.h -- the trace kernel of source runs with many energies (more than 8: whenever the weights do not fit in registers): reflections are LOGGED and the
 * weights of a photon are swept once per log, with the weight of a (photon, energy) pair in a register across all the
 * logged reflections.  Included by pc_kernels.hip.
 *
 * Reference: the per-energy loop of polycap_capil_reflect, src/polycap-capil.c:625-645 (w *= R(E, theta) * r_rough for every
 * energy; the photon lives on while some energy keeps w >= 1e-4), reached from polycap_capil_trace (:1338-1346) once per
 * reflection of polycap_photon_launch's loop (src/polycap-photon.c:910-927).
 *
 * The trajectory of a photon depends on its weights only through "no energy holds >= 1e-4 any more".  So a reflection does
 * not touch the weights: it appends its three numbers (cos theta and the polarisation fractions fs, fp of pc_refl_geom3 -- 24
 * bytes) to the log of its lane in global memory and the photon flies on.  The sweep runs when a log is full (a.log_cap
 * entries), when the photon has left the optic, or when the lane's PROXY says the photon may be dead (below); it is flat over
 * (photon, energy) pairs: the photons of the wave that are due form one list of nP x n_energies items, lane l takes items l,
 * l + 64, ...  Round 3's kernel kept four waiting reflections per lane in LDS and made a read-modify-write of the 2.3 KB weight
 * row per four reflections (33 KB of HBM traffic per started photon); here a reflection costs 48 B and the row of a photon is
 * written once -- when the log was full and the photon lives on, or when it has left the optic.
 *
 *   proxy.  Every lane carries the weights of one or two energies of its photon itself (the energies the host found most
 *     reflective at 3 and 30 mrad), multiplied at every reflection.  While a proxy weight is >= 1e-4 the photon is alive and
 *     nothing it does is speculative.  Once both are below, the log is swept after every reflection (after every fourth once a
 *     sweep has found the photon alive all the same).  The proxy schedules, the sweep decides: a photon ends exactly where the
 *     reference ends it.
 *   tame reflections.  The reference rejects a reflectivity outside [0, 1] at any energy (:633-637, photon ends with rc -1).
 *     The host certifies a grazing cosine ct_tame above which no energy of the run can come within 1e-11 of 1
 *     (pc_sweep_certificate: scan of 1 - R_s, 1 - R_p in extended precision over 13 decades of cos theta, per energy); a
 *     reflection with cos theta >= ct_tame and sane fractions is "tame": every factor is in [0, 1).  A log of tame reflections
 *     is swept by the FAST loop -- no range test, no counting: weights only fall, so the photon is alive iff some energy ends
 *     >= 1e-4, and dead means absorbed (rc 0), since no factor was rejected.  A log with a reflection that is not tame (3e-8
 *     of them on xos1) is swept by the EXACT loop, which tests every factor and counts the leading keeps like the reference.
 *   roughness.  r_rough = exp(-(k_E cos theta)^2) (:626-627) multiplies every factor.  The FAST loop applies the product of a
 *     log's roughness factors at once, exp(-k_E^2 sum cos^2 theta): one exponential per energy and sweep instead of one per
 *     energy and reflection (25 of 71 instructions), and one rounding of the device's 6e-15 exponential instead of one per
 *     reflection.  With roughness the weights therefore equal those of the immediate sweep to ~1e-14, not bit for bit; where a
 *     log is cut depends on the photon alone, so a photon's weights do not depend on scheduling, partition or device count.
 *   fused finalisation, dead weights (histogram-only runs).  A photon that has reached the exit window with a live proxy is alive
 *     at the end of its log, so its sweep adds its weights to the exact LDS sums itself: no weight row is written for it and
 *     none read back by the NEW phase.  Should the sweep find it dead all the same, a second pass over its items takes the sums
 *     back exactly (a.sweep_fuse = 2 fuses whatever the proxies say: the tests' way to that pass).  Totals are exact sums of
 *     floor(w 2^62): a weight below 2^-64 contributes 0 and, every later factor of a tame log being < 1, stays below: a pass
 *     stops once all its 64 weights are there (tested every eighth reflection).  Runs that keep images multiply every weight to
 *     the end and hand the rows to the NEW phase.
 *   gathered sweeps.  A pass takes 64 (photon, energy) pairs and the last pass of a round is seldom full (291 energies: 9 % of a
 *     lone photon's lanes idle, 2.6 % for three photons): finished photons and lanes with a full log wait until a.flush_min of
 *     them can be swept together (or nothing else can run); up to a.stage_ps logs are staged in LDS per round.
 *   slots.  Waves take chunks of 128 slots from one counter; once fewer than 128 per wave are left a request is served with
 *     its share of the rest (lanes that get none retire), so that the launch does not end with some waves still holding a
 *     hundred photons and the others none: the sweeps, 92 % of the work, are as efficient for three photons as for sixty-four.
 *   registers.  Two waves per SIMD (512-thread workgroups), 239 registers, no scratch; the FAST loop evaluates two reflections
 *     per step as two interleaved dependent chains (pc_fresnel3xN).
 */
#ifndef PCS_BLOCK
#define PCS_BLOCK 512          /* 8 waves per CU, 2 per SIMD: 256 registers per lane (at 3 per SIMD and 168 registers a hundred of them lived in
                                * scratch, on the path of every EVENT phase: 28.9 against 28.2 ms at 291 energies, 18.1 against 15.4 ms at 100:
                                * profiles/r04/kernel_history.md); the sweeps keep the SIMD busy with two interleaved chains per wave */
#endif
#ifndef PCS_WAVES
#define PCS_WAVES 2
#endif
#define PCS_PITCH 1024
#define PCS_MAXPS 16           /* photons of a wave swept in one round (their logs are staged in LDS) */
#define PCS_ENT 4              /* doubles of a staged log entry: cos sqrt(2), cos^2, fs, fp */
#define PCS_DEAD 5.421010862427522e-20    /* 2^-64 */
#ifndef PCS_CHAINS
#define PCS_CHAINS 2           /* reflections the FAST loop takes per step (pc_fresnel3xN: that many interleaved chains) */
#endif
#ifndef PCS_LEASH
#define PCS_LEASH 4            /* reflections between sweeps of a photon whose proxies are dead but which a sweep found alive */
#endif

/* dynamic LDS of pc_trace_log_kernel: exact sums, per-energy constants (5 fields), per wave the sweep tables (4 x 16 words +
 * 16 doubles) and `stage` doubles of staged logs */
static size_t pcs_dyn_lds(size_t ne, int block, size_t stage_doubles_per_wave)
{
	return 2*ne*sizeof(unsigned long long) + 5*ne*sizeof(double)
	     + (size_t)(block/PC_WAVE)*(4*PCS_MAXPS*sizeof(unsigned int) + PCS_MAXPS*sizeof(double) + stage_doubles_per_wave*sizeof(double));
}

template <int MODE>
__global__ void __launch_bounds__(PCS_BLOCK, PCS_WAVES)
pc_trace_log_kernel(pc_kargs a)
{
	static_assert(MODE != PC_MODE_EXPLICIT, "the log kernel serves source runs");
	__shared__ double lds[6*PCS_PITCH];
	__shared__ pc_marg4 ldsg[PCS_PITCH];
	extern __shared__ unsigned long long l_acc[];
	const pc_params &Pm = a.pm;
	const int npts = Pm.nmax + 1, ne = Pm.n_energies;
	double *l_z = lds, *l_cap = lds + PCS_PITCH, *l_zh = lds + 2*PCS_PITCH, *l_cap2 = lds + 3*PCS_PITCH;
	double *l_hexd = lds + 4*PCS_PITCH, *l_idz = lds + 5*PCS_PITCH;
	/* per-energy constants of FORM 3 in LDS: d2, Re n^2, Im n^2, zi2 (fields 0-3 of ec_soa) and rough_c^2 (field 6) */
	double *const ecs = (double *)(l_acc + 2*ne);
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		l_z[k] = a.g_z[k];
		l_cap[k] = a.g_cap[k];
		l_zh[k] = a.g_zh[k];
		l_cap2[k] = a.g_cap2[k];
		l_hexd[k] = a.g_hexd[k];
		l_idz[k] = a.g_idz[k];
		ldsg[k] = a.g_mg[k];
	}
	for (int k = threadIdx.x; k < 2*ne; k += blockDim.x) l_acc[k] = 0ull;
	for (int k = threadIdx.x; k < 4*ne; k += blockDim.x) ecs[k] = a.ec_soa[k];
	for (int k = threadIdx.x; k < ne; k += blockDim.x) ecs[4*ne + k] = a.ec_soa[6*ne + k];
	__syncthreads();
	pc_tables T;
	T.z = l_z; T.cap = l_cap; T.zh = l_zh; T.cap2 = l_cap2; T.ext = a.g_ext;
	T.hexd = l_hexd; T.idz = l_idz;              /* read once per flight and per segment visit: on the path of a lone photon at the end of a launch */
	T.mg = ldsg;
	const long long ws = a.img_ws;
	const int lane = threadIdx.x & (PC_WAVE - 1), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const long long wave_gtid0 = gtid - lane;
	const long long grid_waves = (long long)gridDim.x * nwaves;
	const int K = a.log_cap, PS = a.stage_ps;
	double *const l_csum = ecs + 5*ne + wave*PCS_MAXPS;     /* per staged photon: sum of cos^2 over its log */
	unsigned int *const l_tab = (unsigned int *)(ecs + 5*ne + nwaves*PCS_MAXPS);
	unsigned int *const map = l_tab + wave*(4*PCS_MAXPS), *const vflag = map + PCS_MAXPS, *const vcnt = map + 2*PCS_MAXPS, *const vbad = map + 3*PCS_MAXPS;
	double *const stage = (double *)(l_tab + nwaves*(4*PCS_MAXPS)) + (size_t)wave*(size_t)(PS*K*PCS_ENT);
	double *const my_log = a.rlog + gtid*(long long)(3*K);
	const int rough = a.sweep_rough;
	const bool skip = a.sweep_skip != 0;

	const unsigned long long t_begin = __builtin_readcyclecounter();   /* s_memtime: how evenly the waves finish (pc_hip_sweep_stats) */

	pc_photon<0> ph;
	ph.wmem = nullptr; ph.wstride = 1; ph.wset = 0; ph.rc = 0;

	int state = LS_NEED_SLOT;
	int npend = 0;                /* reflections in this lane's log */
	int lim = K;                  /* the log is swept when it holds this many: K while a proxy is alive */
	int untame = 0;               /* the log holds a reflection that is not tame: EXACT sweep */
	double wprox0 = 1.0, wprox1 = 1.0;   /* the proxy energies' weights of this lane's photon */
	int okpre = -1;               /* exit-window test of a finished photon made ahead of its sweep (-1: not made) */
	int summed = 0;               /* the sweep has added the finished photon's weights to the sums itself (fused finalisation) */
	int fresh = 0;                /* the last EVENT phase left a reflection's raw numbers (cos theta, (E.s)^2, |n x d|^2) in fr_*: fractions, log entry,
	                               * tameness and proxies are due (the bookkeeping step at the top of the loop) */
	double fr_c = 0., fr_es2 = 0., fr_sd2 = 0.;
	long long slot = -1;
	unsigned int attempt = 0;
	double cosalpha0 = 0.;
	long long chunk_next = 0, chunk_end = 0;
	unsigned long long u_exit = 0, u_not_entered = 0, u_not_trans = 0, u_irefl = 0, u_failed = 0, u_launch = 0;
	unsigned long long st_march = 0, st_march_l = 0, st_event = 0, st_event_l = 0, st_new = 0, st_new_l = 0, st_pass = 0, st_iter = 0;

	/* ---------------- the sweep of the photons in mR (at most PS of this wave): stage their logs, multiply, verdicts */
	auto sweep_round = [&](unsigned long long mR) __attribute__((always_inline)) {
		const int mine = (int)((mR >> lane) & 1ull);
		const int rank = __popcll(mR & ((1ull << lane) - 1ull));
		const int nP = __popcll(mR);
		/* Fused finalisation (histogram-only runs): a photon that has reached the end of the optic inside the exit window, whose log
		 * is tame and one of whose proxies is alive, is alive at the end of its log, so the sweep adds its weights to the exact sums
		 * itself -- no weight row is written for it and none read back by the NEW phase.  Should the sweep find it dead all the
		 * same (a proxy that differs from the sweep's weight in the last bit), a second pass takes the sums back exactly. */
		int fin = 0;
		if (mine && state == LS_DONE && ph.rc == 1) {
			okpre = pc_in_exit_window(Pm, ph);
			fin = (okpre && a.sweep_fuse && !untame && (lim == K || a.sweep_fuse > 1)) ? 1 : 0;
		}
		if (mine) {
			map[rank] = (unsigned)lane | (ph.wset ? 0x40u : 0u) | (untame ? 0x80u : 0u) | ((unsigned)npend << 8) | (fin ? 0x10000u : 0u);
			vflag[rank] = 0u; vcnt[rank] = 0u; vbad[rank] = 255u;
		}
		/* the lanes' log entries (global stores of the EVENT phases) have arrived; the tables above are visible to the wave */
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		__builtin_amdgcn_wave_barrier();
		for (int qq = 0; qq < nP; qq++) {
			const unsigned info = map[qq];
			const int n = (int)((info >> 8) & 255u);
			const double *src = a.rlog + (wave_gtid0 + (long long)(info & 63u))*(long long)(3*K);
			double *dst = stage + qq*(PCS_ENT*K);
			double part = 0.;
			for (int r = lane; r < n; r += PC_WAVE) {
				const double c = src[3*r], c2 = c*c;
				dst[PCS_ENT*r] = pc_refl_cr2(c); dst[PCS_ENT*r + 1] = c2; dst[PCS_ENT*r + 2] = src[3*r + 1]; dst[PCS_ENT*r + 3] = src[3*r + 2];
				part += c2;
			}
			if (rough) {
				/* sum of cos^2 over the log, in a fixed order (lane partial sums, then a butterfly): the same bits wherever the photon runs */
#pragma unroll
				for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, PC_WAVE);
				if (lane == 0) l_csum[qq] = part;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const int total = nP*ne;
		for (int rep = 0; rep < 2; rep++) {
		const bool undo = rep != 0;
		int q = 0, e = lane;
		while (e >= ne) { e -= ne; q++; }
		for (int base = 0; base < total; base += PC_WAVE) {
			bool act = base + lane < total;
			if (undo) {
				/* only the photons whose sums have to be taken back */
				act = act && vflag[q] == 2u;
				if (__builtin_amdgcn_ballot_w64(act) == 0ull) { e += PC_WAVE; while (e >= ne) { e -= ne; q++; } continue; }
			}
			const int qc = act ? q : 0, ee = act ? e : 0;
			const unsigned info = map[qc];
			const int p = (int)(info & 63u);
			const int n = act ? (int)((info >> 8) & 255u) : 0;
			const bool exact = act && (info & 0x80u);
			double *const wrow = a.wscratch + (wave_gtid0 + p)*(long long)ne + ee;
			double w = 1.0;
			if (act && (info & 0x40u)) w = *wrow;
			const double d2 = ecs[ee], n2r = ecs[ne + ee], n2i = ecs[2*ne + ee], zi2 = ecs[3*ne + ee];
			const double *gq = stage + qc*(PCS_ENT*K);
			int n_pass = n;               /* the longest log of the pass (it holds the items of up to three photons) */
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(n_pass, off, PC_WAVE); n_pass = (o > n_pass) ? o : n_pass; }
			n_pass = __builtin_amdgcn_readfirstlane(n_pass);
			bool alive;
			st_pass++;
			if (__builtin_amdgcn_ballot_w64(exact) == 0ull) {
				/* FAST: every reflection of these logs is tame.  A pass whose items all belong to logs of one length (one photon,
				 * nearly always) runs two reflections per step: two independent Fresnel chains for the fp64 pipe */
				int r = 0;
				if (PCS_CHAINS > 1 && __builtin_amdgcn_ballot_w64(act && n != n_pass) == 0ull) {
					/* (reading the next step's entries from LDS ahead of the current step's arithmetic was measured: 28.5-28.9 against
					 * 28.1-28.4 ms, with two waves per SIMD the other wave covers the LDS latency already: profiles/r04/kernel_history.md) */
					for (; r + PCS_CHAINS <= n_pass; r += PCS_CHAINS) {
						if (skip && (r & 7) + PCS_CHAINS > 7 && __builtin_amdgcn_ballot_w64(act && w >= PCS_DEAD) == 0ull) break;
						double in[PCS_CHAINS][4], f[PCS_CHAINS];
#pragma unroll
						for (int k = 0; k < PCS_CHAINS; k++) {
							in[k][0] = gq[PCS_ENT*(r + k)]; in[k][1] = gq[PCS_ENT*(r + k) + 1]; in[k][2] = gq[PCS_ENT*(r + k) + 2]; in[k][3] = gq[PCS_ENT*(r + k) + 3];
						}
						pc_fresnel3xN<PCS_CHAINS>(d2, n2r, n2i, zi2, in, f);
#pragma unroll
						for (int k = 0; k < PCS_CHAINS; k++) w = w*f[k];
					}
					for (; r < n_pass; r++)
						w = w*pc_fresnel3(d2, n2r, n2i, zi2, gq[PCS_ENT*r], gq[PCS_ENT*r + 1], gq[PCS_ENT*r + 2], gq[PCS_ENT*r + 3]);
				} else {
					for (; r < n_pass; r++) {
						if (skip && (r & 7) == 7 && __builtin_amdgcn_ballot_w64(r < n && w >= PCS_DEAD) == 0ull) break;
						if (r < n) {
							const double f = pc_fresnel3(d2, n2r, n2i, zi2, gq[PCS_ENT*r], gq[PCS_ENT*r + 1], gq[PCS_ENT*r + 2], gq[PCS_ENT*r + 3]);
							w = w*f;
						}
					}
				}
				st_iter += (unsigned long long)r;
				if (rough) w = w*pc_exp_neg_fast(-(ecs[4*ne + ee]*l_csum[qc]));
				alive = act && (w >= 1.e-4);
			} else {
				/* EXACT: the reference's tests at every reflection.  cnt = leading reflections after which this energy still holds
				 * >= 1e-4; bad = first reflection whose rtot the reference rejects at this energy */
				unsigned cnt = 0u, bad = 255u;
				bool lead = true;
				const double k2 = rough ? ecs[4*ne + ee] : 0.;
				for (int r = 0; r < n_pass; r++) {
					if (r < n) {
						const double rt = pc_fresnel3(d2, n2r, n2i, zi2, gq[PCS_ENT*r], gq[PCS_ENT*r + 1], gq[PCS_ENT*r + 2], gq[PCS_ENT*r + 3]);
						if ((rt < 0. || rt > 1.) && bad == 255u) bad = (unsigned)r;          /* src/polycap-capil.c:633-637 */
						double f = rt;
						if (rough) f = rt*pc_exp_neg_fast(-(k2*gq[PCS_ENT*r + 1]));
						w = w*f;
						lead = lead && (w >= 1.e-4);
						cnt += lead ? 1u : 0u;
					}
				}
				st_iter += (unsigned long long)n_pass;
				if (act) {
					atomicMax(&vcnt[qc], cnt);
					if (bad != 255u) atomicMin(&vbad[qc], bad);
				}
				alive = act && (cnt == (unsigned)n);
			}
			if (act) {
				if (info & 0x10000u) {
					const unsigned long long f = (unsigned long long)(w * PC_FIX_SCALE);
					if (f) {
						if (!undo) {
							const unsigned long long old = atomicAdd(&l_acc[2*ee], f);
							if (old + f < old) atomicAdd(&l_acc[2*ee + 1], 1ull);
						} else {
							const unsigned long long old = atomicSub(&l_acc[2*ee], f);
							if (old < f) atomicSub(&l_acc[2*ee + 1], 1ull);
						}
					}
				} else {
					*wrow = w;
				}
			}
			/* one lane per photon of the pass reports "some energy of these is alive at the end of the log" */
			if (!undo) {
				const unsigned long long mK = __builtin_amdgcn_ballot_w64(alive);
				const int q_left = __shfl_up(qc, 1, PC_WAVE);
				if (act && (lane == 0 || q_left != qc)) {
					const int len = (ne - ee < PC_WAVE - lane) ? ne - ee : PC_WAVE - lane;
					const unsigned long long run = ((len >= 64) ? ~0ull : ((1ull << len) - 1ull)) << lane;
					if (mK & run) vflag[qc] = 1u;
				}
			}
			e += PC_WAVE;
			while (e >= ne) { e -= ne; q++; }
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (undo) break;
		/* a fused photon the sweep found dead: its sums are taken back by a second pass over its items */
		const bool back = mine && fin && vflag[rank] == 0u;
		if (back) vflag[rank] = 2u;
		if (__builtin_amdgcn_ballot_w64(back) == 0ull) break;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		}
		if (mine) {
			if (fin) summed = (vflag[rank] == 1u) ? 1 : 0;
			if (vflag[rank] == 2u) vflag[rank] = 0u;
			if (untame) {
				/* first logged reflection that ends the photon: an error at any energy (rc -1), or no energy left above 1e-4 (rc 0) */
				const unsigned c = vcnt[rank], b = vbad[rank];
				const unsigned fail = (c < b) ? c : b;
				if (fail < (unsigned)npend) { state = LS_DONE; ph.rc = (b <= c) ? -1 : 0; }
			} else if (vflag[rank] == 0u) {
				state = LS_DONE; ph.rc = 0;
			}
			npend = 0; untame = 0;
			if (!fin) ph.wset = 1;
			if (lim < K) lim = (PCS_LEASH < K) ? PCS_LEASH : K;       /* proxies dead, photon swept: a longer leash from here on */
		}
	};
	/* Rounds of exactly a.flush_min photons (the count whose last pass is nearly full); what is left over waits for the next
	 * flush unless nothing else can run (`all`) */
	auto flush = [&](unsigned long long mF, bool all) __attribute__((always_inline)) {
		const int per = all ? PS : a.flush_min;
		while (mF && (all || __popcll(mF) >= per)) {
			unsigned long long mR = 0ull, t = mF;
			for (int k = 0; k < per && t; k++) { mR |= t & (0ull - t); t &= t - 1ull; }
			mF &= ~mR;
			sweep_round(mR);
		}
	};

	for (;;) {
		if (__builtin_amdgcn_ballot_w64(fresh != 0) != 0ull) {
			if (fresh) {
				fresh = 0;
				double *gq = my_log + 3*npend;
				pc_refl_geom g;
				g.alfa = fr_c; g.es2 = fr_es2; g.sd2 = fr_sd2; g.ep2 = fr_sd2 - fr_es2; g.st2 = 0.;
				double c2, fs, fp;
				pc_refl_geom3(g, c2, fs, fp);
				gq[0] = fr_c; gq[1] = fs; gq[2] = fp;
				npend++;
				/* tame: cos theta above the host's certificate, fractions as the geometry makes them (fs in [0, 1], fp = 1 - fs to
				 * rounding).  NaNs fail every comparison. */
				if (!(fr_c >= a.ct_tame && fr_c <= 1.0 && fs >= 0. && fs <= 1.0000001 && fp >= -1.e-7 && fp <= 1.0000001)) untame = 1;
				if (lim == K) {
					{
						const int pe = a.proxy_e[0];
						double f = pc_fresnel3(ecs[pe], ecs[ne + pe], ecs[2*ne + pe], ecs[3*ne + pe], pc_refl_cr2(fr_c), c2, fs, fp);
						if (rough) f = f*pc_exp_neg_fast(-(ecs[4*ne + pe]*c2));
						wprox0 = wprox0*f;
					}
					if (a.n_proxy > 1) {
						const int pe = a.proxy_e[1];
						double f = pc_fresnel3(ecs[pe], ecs[ne + pe], ecs[2*ne + pe], ecs[3*ne + pe], pc_refl_cr2(fr_c), c2, fs, fp);
						if (rough) f = f*pc_exp_neg_fast(-(ecs[4*ne + pe]*c2));
						wprox1 = wprox1*f;
					}
					if (!(wprox0 >= 1.e-4 || (a.n_proxy > 1 && wprox1 >= 1.e-4))) lim = 1;
				}
			}
		}
		/* sweeps: a lane whose log has reached its limit cannot reflect again (it waits at its next wall); a finished photon is
		 * swept before the NEW phase finalises it.  Those who wait are swept together once there are a.flush_min of them (a pass
		 * takes 64 (photon, energy) pairs and the last pass of a round is seldom full: 291 energies leave 9 % of the lanes idle
		 * when one photon is swept alone, 2.6 % with three), or when nothing else can run */
		const unsigned long long mBlk = __ballot(state == LS_EVENT && npend > 0 && npend >= lim);
		const unsigned long long mDn = __ballot(state == LS_DONE && npend > 0);
		const unsigned long long mM = __ballot(state == LS_MARCH);
		const unsigned long long mE = __ballot(state == LS_EVENT) & ~mBlk;
		const unsigned long long mN = __ballot((state == LS_DONE && npend == 0) || state == LS_NEED_SLOT || state == LS_START);
		if ((mM | mE | mN | mBlk | mDn) == 0ull) break;
		const int nM = __popcll(mM), nE = __popcll(mE), nN = __popcll(mN);
		const bool do_new = (nN >= a.new_threshold) || (nM == 0 && nE == 0);
		const int phase = (nM > 0 && (nM >= a.event_threshold || (nE == 0 && !do_new))) ? 0 : ((nE > 0 && !(do_new && nN > nE)) ? 1 : ((nN > 0 && do_new) ? 2 : 3));
		if (mBlk | mDn) {
			if (__popcll(mBlk | mDn) >= a.flush_min || phase == 3) { flush(mBlk | mDn, phase == 3); continue; }
		}
		if (phase == 0) {
			/* ---------------- MARCH burst */
			if (Pm.literal) {
				if (state == LS_MARCH) state = pc_march_step(T, Pm, ph);
			} else {
				if (state == LS_MARCH && ph.first) state = pc_march_step(T, Pm, ph);
				for (int b = 0; b < a.march_burst; b++) {
					unsigned int lanes_in_burst = 0;
#pragma unroll
					for (int u = 0; u < PC_MARCH_UNROLL; u++) {
						lanes_in_burst += (unsigned)__popcll(__ballot(state == LS_MARCH));
						if (state == LS_MARCH) state = pc_march_step_hot(T, Pm, ph);
					}
					const int cM = __popcll(__ballot(state == LS_MARCH));
					st_march += PC_MARCH_UNROLL; st_march_l += lanes_in_burst;
					if (cM == 0) break;
					if (cM < a.march_stop && (cM != nM || do_new || nE > 0)) break;
				}
			}
		} else if (phase == 1) {
			/* ---------------- EVENT: literal segment visit; a reflection is logged, the proxies multiplied */
			st_event += 1; st_event_l += (unsigned)nE;
			pc_hit h;
			pc_refl_geom g;
			int pend = 0;
			h.nx = h.ny = h.nz = h.cosalfa = 0.; h.ix = 0;
			g.alfa = g.st2 = g.es2 = g.ep2 = g.sd2 = 0.;
			if (state == LS_EVENT && ((mE >> lane) & 1ull)) {
				int st = pc_event_pre(T, Pm, ph, h);
				if (st == PC_ST_REFLECT) {
					pend = (pc_reflect_geom(ph, h.nx, h.ny, h.nz, g) < 0) ? 2 : 1;
				} else {
					state = st;
				}
			}
			if (pend) {
				/* pend == 2: a geometry the reference rejects ends the photon with rc -1 unless a logged reflection ends it first:
				 * settled by the sweep that precedes the finalisation */
				if (pend == 1) { ph.ex = fabs(ph.ex); ph.ey = fabs(ph.ey); ph.ez = fabs(ph.ez); }
				state = pc_event_post(Pm, ph, h, (pend == 1) ? 1 : -1);
			}
			if (pend == 1) {
				/* the reflection's numbers wait for the bookkeeping step at the top of the loop (fractions, log entry, tameness,
				 * proxies): the EVENT phase is where registers are scarcest */
				fr_c = g.alfa; fr_es2 = g.es2; fr_sd2 = g.sd2;
				fresh = 1;
			}
		} else if (phase == 2) {
			st_new += 1; st_new_l += (unsigned)nN;
			/* ---------------- NEW: finalise finished photons (their logs are empty), hand out slots, sample + entrance tests */
			int coop = 0;
			int f_exit = 0, f_not_entered = 0, f_not_trans = 0, f_failed = 0, f_launch = 0;
			unsigned int f_irefl = 0;
			long long done_slot = slot;
			int ok = 0;                   /* the photon left through the exit window: src/polycap-source.c:758-777 */
			const bool fin_now = state == LS_DONE && npend == 0;       /* finished photons whose logs are still to be swept wait */
			if (fin_now) {
				const int rc = ph.rc;
				if (rc == 0) f_not_trans = 1;
				else if (rc == 2) f_not_entered = 1;
				else if (rc == 1) ok = (okpre >= 0) ? okpre : pc_in_exit_window(Pm, ph);
			}
			const bool compact = a.keep_images && a.img_cursor != nullptr;
			unsigned long long c_base = 0ull;
			int c_k = 0;
			if (compact) {
				const unsigned long long mOK = __ballot(ok);
				if (mOK) {
					c_k = __popcll(mOK);
					if (lane == 0) c_base = atomicAdd(a.img_cursor, (unsigned long long)c_k);
					c_base = __shfl(c_base, 0, PC_WAVE);
					if (ok) done_slot = (long long)(c_base + (unsigned long long)__popcll(mOK & ((1ull << lane) - 1ull)));
				}
			}
			if (fin_now) {
				if (ok) {
					f_exit = 1;
					f_irefl = (unsigned int)ph.irefl;
					coop = summed ? 0 : 1;    /* sums and image weights: the cooperative sweep below, unless the photon's sweep has added them */
					if (a.keep_images) {
						/* src/polycap-source.c:900-923 */
						if (compact) {
							const double *ls = a.lane_start + gtid*8;
							pc_write_start_fields<true>(a, done_slot, ls[0], ls[1], ls[2], ls[3], ls[4], ls[5], ls[6], ls[7]);
							pc_write_exit_fields<true>(a, Pm, done_slot, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, cosalpha0, (long long)ph.irefl, ph.dtravel);
							if (a.img_ids) pc_store_wt(a.img_ids + done_slot, slot);
						} else {
							pc_write_exit_fields<false>(a, Pm, done_slot, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, cosalpha0, (long long)ph.irefl, ph.dtravel);
						}
					}
					state = LS_NEED_SLOT;
				} else {
					attempt++;
					if (attempt >= a.max_attempts) {
						f_failed = 1;
						if (a.keep_images && !compact) coop = 2;   /* zero weights */
						state = LS_NEED_SLOT;
					} else {
						state = LS_START;
					}
				}
			}
			{
				/* cooperative sweep over the weights of the photons finalised above: 64 lanes over energies */
				unsigned long long mC = __ballot(coop != 0);
				while (mC) {
					const int p = __ffsll((long long)mC) - 1;
					mC &= mC - 1ull;
					const int what = __shfl(coop, p, PC_WAVE);
					const int wset_p = __shfl(ph.wset, p, PC_WAVE);
					const long long slot_p = __shfl(done_slot, p, PC_WAVE);
					const double *wp = a.wscratch + (wave_gtid0 + p)*(long long)ne;
					for (int e = lane; e < ne; e += PC_WAVE) {
						const double w = (what == 2) ? 0. : (wset_p ? wp[e] : 1.0);
						if (what == 1) {
							const unsigned long long f = (unsigned long long)(w * PC_FIX_SCALE);
							if (f) {
								const unsigned long long old = atomicAdd(&l_acc[2*e], f);
								if (old + f < old) atomicAdd(&l_acc[2*e + 1], 1ull);
							}
						}
						if (a.keep_images) { if (compact) pc_store_wt(a.img_w + slot_p*ws + e, w); else a.img_w[slot_p*ws + e] = w; }
					}
				}
			}
			/* compact store: the positions taken above are complete: count them into their blocks (pc_trace_kernel does the same) */
			if (compact && c_k > 0 && a.blk_done) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0) pc_blocks_written(a, c_base, c_k);
			}
			/* hand out slots: wave-uniform chunk, refilled from the global counter by one lane.  Near the end of the launch a
			 * request is served with its share of what is left (header comment, "slots") */
			{
				const unsigned long long need = __ballot(state == LS_NEED_SLOT);
				if (need) {
					const int k = __popcll(need);
					const int rank = __popcll(need & ((1ull << lane) - 1ull));
					const long long have = chunk_end - chunk_next;
					if (have < k) {
						long long base_new = 0, take = PC_CHUNK;
						if (lane == 0) {
							const long long seen = (long long)__hip_atomic_load(a.work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							const long long rem = a.n_slots - seen;
							if (rem < (long long)PC_CHUNK*grid_waves) {
								take = (rem + grid_waves - 1)/grid_waves;
								if (take < 1) take = 1;
							}
							base_new = (long long)atomicAdd(a.work, (unsigned long long)take);
						}
						base_new = __shfl(base_new, 0, PC_WAVE);
						take = __shfl(take, 0, PC_WAVE);
						if (state == LS_NEED_SLOT)
							slot = (rank < have) ? (chunk_next + rank) : ((rank - have < take) ? (base_new + (rank - have)) : a.n_slots);
						const long long used = (k - have < take) ? k - have : take;
						chunk_next = base_new + used;
						chunk_end = base_new + take;
					} else {
						if (state == LS_NEED_SLOT) slot = chunk_next + rank;
						chunk_next += k;
					}
					if (state == LS_NEED_SLOT) {
						if (slot >= a.n_slots) { state = LS_IDLE; }
						else { attempt = 0; state = LS_START; }
					}
				}
			}
			/* start an attempt */
			if (state == LS_START) {
				f_launch = 1;
				pc_start s;
				pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), attempt, s);
				state = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
				npend = 0; untame = 0; lim = K; okpre = -1; summed = 0;
				wprox0 = 1.0; wprox1 = 1.0;
				if (state == LS_MARCH) {
					cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
					if (a.keep_images) {
						double evx, evy;
						pc_start_elecv_image(s, cosalpha0, evx, evy);
						if (a.img_cursor) {
							double *ls = a.lane_start + gtid*8;
							ls[0] = s.srcx; ls[1] = s.srcy; ls[2] = s.x; ls[3] = s.y; ls[4] = s.dx; ls[5] = s.dy; ls[6] = evx; ls[7] = evy;
						} else {
							pc_write_start_fields<false>(a, slot, s.srcx, s.srcy, s.x, s.y, s.dx, s.dy, evx, evy);
						}
					}
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
			}
		}
	}

	__syncthreads();          /* every wave of the workgroup has finished its photons */
	for (int e = threadIdx.x; e < ne; e += blockDim.x)
		if (l_acc[2*e] | l_acc[2*e + 1]) pc_atomic_add128(a.sumw + 2*e, l_acc[2*e], l_acc[2*e + 1]);
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
		atomicAdd(&a.totals->phase[6], st_pass); atomicAdd(&a.totals->phase[7], st_iter);
		const unsigned long long life = __builtin_readcyclecounter() - t_begin;
		atomicAdd(&a.totals->counters[6], life);
		atomicMax(&a.totals->counters[7], life);
	}
}
