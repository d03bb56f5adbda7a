/*
 * pc_kernels.hip -- gfx950 kernels of the photon trace path + the thin C-ABI of include/polycap-hip.h.
 *
 * Kernel shape (CDNA4, wave64):
 *   - persistent waves: grid = CUs x resident blocks; every wave pulls chunks of exit-photon slots from one
 *     global counter, every lane owns one slot at a time and retries it until a photon is transmitted
 *     (reference driver loop src/polycap-source.c:744-884);
 *   - profile tables (z, cap, zh, cap^2, hexd: 5 x (nmax+1) fp64 = 40 KB for nmax=999) are staged once per
 *     workgroup into LDS; per-energy constants are wave-uniform scalar loads;
 *   - the photon life cycle is scheduled wave-wide by phase (MARCH / EVENT / NEW) with ballots so that the
 *     expensive, rare code (segment quadratic + Fresnel, source sampling) runs with many active lanes;
 *   - totals are accumulated in exact 128-bit fixed point, so they do not depend on scheduling or on how
 *     slots are split over devices.
 * No MFMA: there is no dense contraction in this path.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "polycap-hip.h"
#include "pc_device.h"
#include "pc_leak.h"
#include "pc_problem.h"

#ifndef PC_BLOCK
#define PC_BLOCK 512            /* maximum workgroup size the trace kernel is compiled for */
#endif
#define PC_WAVE 64
#define PC_MAX_PITCH 2048      /* largest profile kept in static LDS: (6 x 8 + 4 x 4) B x 2048 = 128 KB */
#ifndef PC_MARCH_UNROLL
#define PC_MARCH_UNROLL 4      /* march steps between two ballots of the burst loop.  With march_stop = 8 (a burst goes on while 8 lanes march):
                                * 3: 23.6 ms, 4: 23.4, 5: 23.9, 8: 24.1 (scripts/ab_build.sh, xos1 10 keV, 1e7 slots) */
#endif
#define PC_KE 5                /* energies per lane whose weights are in flight together in a cooperative sweep */
#ifndef PC_CHUNK
#define PC_CHUNK 128           /* slots a wave takes from the global counter at a time */
#endif
#define PC_FIX_SCALE 4611686018427387904.0 /* 2^62 */
#ifndef PC_MIN_WAVES_NE0
#define PC_MIN_WAVES_NE0 2     /* the any-n_energies kernel: 256 VGPRs (it spills 470 B per lane at 128), 8 waves per CU */
#endif
#ifndef PC_MIN_WAVES
#define PC_MIN_WAVES 4         /* __launch_bounds__ waves per SIMD the register allocator must leave room for */
#endif

/* --------------------------------------------------------------------------- kernel arguments */

/* Per-exit-photon image record in HBM: one contiguous record per slot (17 + n_energies doubles) so that a lane
 * writes whole 64/128-byte segments instead of 18 scattered 8-byte words; the host side of the fetch (pc_hip_transmission_images) turns a
 * range of records into the reference's SoA planes (struct _polycap_images) when the host asks for them.
 * Field order = pc_hip_images / the reference's plane order. */
enum { PC_F_SRCX = 0, PC_F_SRCY, PC_F_STARTX, PC_F_STARTY, PC_F_SDIRX, PC_F_SDIRY, PC_F_SEVX, PC_F_SEVY,
       PC_F_EXITX, PC_F_EXITY, PC_F_EXITZ, PC_F_EDIRX, PC_F_EDIRY, PC_F_EEVX, PC_F_EEVY, PC_F_NREFL, PC_F_DTRAVEL,
       PC_F_WEIGHTS, PC_N_FIELDS = 17 };

struct pc_totals {             /* device-resident totals of one run */
	unsigned long long counters[8];   /* iexit, not_entered, not_transmitted, sum_irefl, failed_slots, launches */
	unsigned long long phase[8];      /* scheduler statistics: march steps, march lane-steps, event phases, event lanes, new phases, new lanes */
	unsigned long long next_slot;     /* work counter (relative slot index) */
	unsigned long long pad;
	/* followed by 2*n_energies u64: (lo, hi) fixed-point weight sums */
};

struct pc_kargs {
	const double *g_z, *g_cap, *g_zh, *g_cap2, *g_hexd, *g_idz, *g_ext, *g_stp, *g_istp;
	const pc_marg4 *g_mg;         /* block-certificate record per start node (pc_problem.h) */
	const pc_drdev *g_dr;         /* leak path: chord deviations of cap per start node */
	unsigned int *work_est;       /* [n_slots] or null: reflections + 1 of every attempt, summed per slot (lane kernels, source runs) */
	const pc_energy_const *ec;
	const double *ec_soa;         /* the sweeps' constants field-major [7][n_energies] (FORM 3: d2, n2_re, n2_im, zi2, rough_c, valid, rough_k2) */
	pc_params pm;
	unsigned long long seed;
	long long slot0, n_slots;
	unsigned int max_attempts;
	int keep_images;
	int event_threshold;
	int march_burst;
	int march_stop;               /* a burst goes on while at least this many lanes march (<= event_threshold, which starts it) */
	pc_totals *totals;
	unsigned long long *work;     /* the launch's work counter (relative slot index handed out next) */
	unsigned long long *sumw;     /* 2*n_energies */
	double *img;                  /* image store of the launch's slots, or NULL: field f of slot j at img[j*img_ss + f*img_fs], weight e at
	                               * img_w[j*img_ws + e].  Records (one per slot): img_ss = 17 + n_energies, img_fs = 1, img_w = img + 17,
	                               * img_ws = img_ss.  Planes (struct _polycap_images): img_ss = 1, img_fs = slots of the run, img_w = weights
	                               * [slot][n_energies], img_ws = n_energies */
	double *img_w;
	long long img_ss, img_fs, img_ws;
	/* Compact image store (option "compact_images", planes only): an exit photon takes the next free position of the run's
	 * planes instead of the position of its slot -- the photons a wave finalises together are written as one coalesced run per
	 * plane -- and the blocks of 2^blk_shift positions are published as they fill (the copy engine fetches behind the kernel).
	 * The order of the photons in the planes is then the order of completion; img_ids, when asked for, says which slot sits
	 * where. */
	unsigned long long *img_cursor;   /* next free position, or NULL: a photon is stored at its slot */
	long long *img_ids;               /* slot of position p (option "slot_ids"), or NULL */
	unsigned int *blk_done;           /* per block: positions written so far */
	unsigned int *blk_flag;           /* per block, host-visible: complete */
	int blk_shift;
	long long img_n;                  /* positions of the launch: its slots */
	double *lane_start;               /* compact store in the kernels that launch in the tracing lane: the 8 start fields of the
	                                   * lane's photon wait here, one 64-byte line per lane, until the photon has left the optic */
	int new_threshold;
	int lds_acc;                  /* NE == 0: accumulate weight sums in LDS (2*n_energies u64 of dynamic LDS) */
	int lds_ec;                   /* NE == 0: per-energy constants staged in LDS behind the sums (6*n_energies doubles) */
	int sweep_rough;              /* NE == 0: some energy has a roughness factor (sig_rough != 0): the sweeps evaluate exp(-(c alfa)^2) */
	int pool_event_min;           /* pool kernel: photons waiting for an EVENT phase that make it run before anything else */
	int event_march;              /* pool kernel: march steps taken right after an EVENT phase, while the wave is still full of fresh flights */
	int pool_refill;              /* pool kernel: lanes that must be free before a march burst tops itself up from the pool */
	double *wscratch;             /* NE==0: n_energies * total_threads */
	long long total_threads;
	/* pc_trace_log_kernel (pc_sweep_kernel.h): many-energy source runs whose reflections are logged */
	double *rlog;                 /* [total_threads][log_cap][3]: cos theta, fs, fp (pc_refl_geom3) of the lane's logged reflections */
	int log_cap;                  /* reflections per log */
	int stage_ps;                 /* photons of a wave swept per round: their logs are staged in LDS (stage_ps*log_cap*PCS_ENT doubles per wave) */
	int n_proxy, proxy_e[2];      /* energies whose weights every lane carries itself (pc_sweep_certificate) */
	int flush_min;                /* photons of a wave that wait for a sweep before one is run for them alone */
	int sweep_skip;               /* histogram-only runs: a weight below 2^-64 is not multiplied any further */
	int sweep_fuse;               /* histogram-only runs: the sweep of a finished photon adds its weights to the sums itself (2: whatever its proxies say -- tests) */
	double ct_tame;               /* a reflection with cos theta >= ct_tame has 0 <= rtot < 1 - 1e-11 at every energy of the run */
	/* explicit-photon mode */
	const double *in_start, *in_dir, *in_elecv;
	int *out_rc;
	double *out_weights, *out_exit_coords, *out_exit_dir, *out_exit_elecv, *out_dtravel;
	long long *out_irefl;
};

/* Stores of the compact image store: written through to memory (system-coherent), so that a block can be handed to the copy
 * engine while the kernel runs without a write-back of the whole L2 (an agent-scope release on gfx950) per batch of photons */
__device__ __forceinline__ void pc_store_wt(double *p, double v)
{
	__hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pc_store_wt(long long *p, long long v)
{
	__hip_atomic_store((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* positions [base, base + k) have been written by this wave (k <= 64 < block size): count them into their blocks and
 * publish a block that is complete.  Called by one lane after the wave's stores have been acknowledged. */
__device__ __forceinline__ void pc_blocks_written(const pc_kargs &a, unsigned long long base, int k)
{
	if (!a.blk_done || k <= 0) return;
	const unsigned long long B = 1ull << a.blk_shift;
	unsigned long long b = base >> a.blk_shift;
	unsigned long long left = (unsigned long long)k, at = base;
	while (left) {
		const unsigned long long end = (b + 1ull) << a.blk_shift;
		const unsigned long long c = (end - at < left) ? end - at : left;
		const unsigned long long size = ((unsigned long long)a.img_n - (b << a.blk_shift) < B) ? (unsigned long long)a.img_n - (b << a.blk_shift) : B;
		const unsigned int old = __hip_atomic_fetch_add(&a.blk_done[b], (unsigned int)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((unsigned long long)old + c == size)
			__hip_atomic_store(&a.blk_flag[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		left -= c; at += c; b++;
	}
}

/* the 18 fields of one exit photon at position `pos` of the image store: src/polycap-source.c:779-798 (start images, from
 * the sampled photon `s`) and :900-923 (exit images).  WT: stores written through (compact store). */
template <bool WT>
__device__ __forceinline__ void pc_store_field(double *p, double v)
{
	if (WT) pc_store_wt(p, v); else *p = v;
}

template <bool WT>
__device__ __forceinline__ void pc_write_start_fields(const pc_kargs &a, long long pos, double srcx, double srcy, double x, double y,
                                                      double dx, double dy, double evx, double evy)
{
	const long long fs = a.img_fs;
	double *r = a.img + pos*a.img_ss;
	pc_store_field<WT>(r + PC_F_SRCX*fs, srcx); pc_store_field<WT>(r + PC_F_SRCY*fs, srcy);
	pc_store_field<WT>(r + PC_F_STARTX*fs, x); pc_store_field<WT>(r + PC_F_STARTY*fs, y);
	pc_store_field<WT>(r + PC_F_SDIRX*fs, dx); pc_store_field<WT>(r + PC_F_SDIRY*fs, dy);
	pc_store_field<WT>(r + PC_F_SEVX*fs, evx); pc_store_field<WT>(r + PC_F_SEVY*fs, evy);
}

/* start_electric_vector projected on the plane perpendicular to the direction, components rounded (:789-796) */
__device__ __forceinline__ void pc_start_elecv_image(const pc_start &s, double cosalpha0, double &evx, double &evy)
{
	const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
	double tx = s.ex*c_ae + s.dx*c_be, ty = s.ey*c_ae + s.dy*c_be, tz = s.ez*c_ae + s.dz*c_be;
	pc_norm3(tx, ty, tz);
	evx = round(tx); evy = round(ty);
}

template <bool WT>
__device__ __forceinline__ void pc_write_exit_fields(const pc_kargs &a, const pc_params &Pm, long long pos, double Px, double Py, double Pz,
                                                     double dx, double dy, double dz, double ex, double ey, double ez,
                                                     double cosalpha0, long long irefl, double dtravel)
{
	const long long fs = a.img_fs;
	double *r = a.img + pos*a.img_ss;
	const double t = (Pm.z_end - Pz) / dz;
	const double xx = Px + dx*t, xy = Py + dy*t, xz = Pz + dz*t;
	pc_store_field<WT>(r + PC_F_EXITX*fs, xx); pc_store_field<WT>(r + PC_F_EXITY*fs, xy); pc_store_field<WT>(r + PC_F_EXITZ*fs, xz);
	pc_store_field<WT>(r + PC_F_EDIRX*fs, dx); pc_store_field<WT>(r + PC_F_EDIRY*fs, dy);
	const double c_ae = 1.0 / sqrt(1.0 - cosalpha0*cosalpha0), c_be = -1.*c_ae*cosalpha0;
	double tx = ex*c_ae + dx*c_be, ty = ey*c_ae + dy*c_be, tz = ez*c_ae + dz*c_be;
	pc_norm3(tx, ty, tz);
	pc_store_field<WT>(r + PC_F_EEVX*fs, round(tx)); pc_store_field<WT>(r + PC_F_EEVY*fs, round(ty));
	if (WT) pc_store_wt((long long *)r + PC_F_NREFL*fs, irefl); else ((long long *)r)[PC_F_NREFL*fs] = irefl;
	const double lx = xx - Px, ly = xy - Py, lz = Pm.z_end - Pz;
	pc_store_field<WT>(r + PC_F_DTRAVEL*fs, dtravel + sqrt(lx*lx + ly*ly + lz*lz));
}

/* lane states on top of pc_device.h's: what the NEW phase has to do for the lane */
enum { LS_IDLE = 0, LS_NEED_SLOT = 1, LS_START = 5, LS_MARCH = PC_ST_MARCH, LS_EVENT = PC_ST_EVENT, LS_DONE = PC_ST_DONE };

__device__ __forceinline__ unsigned long long pc_wave_sum_u64(unsigned long long v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_xor(v, off, PC_WAVE);
	return v;
}

/* exact add of a 128-bit (hi:lo) value to a global (lo,hi) pair; carries are derived from each add's old value */
__device__ __forceinline__ void pc_atomic_add128(unsigned long long *lohi, unsigned long long lo, unsigned long long hi)
{
	unsigned long long old = atomicAdd(&lohi[0], lo);
	unsigned long long carry = (old + lo < old) ? 1ull : 0ull;
	if (hi + carry) atomicAdd(&lohi[1], hi + carry);
}

/* the six per-energy constants of the immediate sweeps, field-major in ec_soa / its LDS copy: d2, n2_re, n2_im, zi2, rough_c,
 * valid (FORM 3, pc_device.h); the seventh field of ec_soa, rough_k2, is what pc_trace_log_kernel reads instead of rough_c */
__device__ __forceinline__ pc_energy_const pc_ec_from_soa(const double *ecs, int ne, int e)
{
	pc_energy_const ec;
	ec.n_re = ec.n_im = ec.ninv2_re = ec.ninv2_im = ec.rough_k2 = 0.;
	ec.d2 = ecs[e]; ec.n2_re = ecs[ne + e]; ec.n2_im = ecs[2*ne + e]; ec.zi2 = ecs[3*ne + e];
	ec.rough_c = ecs[4*ne + e]; ec.valid = ecs[5*ne + e];
	return ec;
}

/* one energy of one reflection in the immediate sweeps of the any-n_energies kernel: FORM 3 of pc_device.h.  Same return
 * values as pc_reflect_energy_f. */
__device__ __forceinline__ int pc_reflect_energy_sweep(const pc_energy_const &ec, double c, double c2, double fs, double fp, double &w)
{
	if (ec.valid == 0.) return -1;
	return pc_reflect_energy3(ec, c, c2, fs, fp, w);
}

/* --------------------------------------------------------------------------- the trace kernel
 * NE > 0: up to NE energies, weights in registers (NE = 1, 4, 8 are instantiated; a run with fewer energies than NE
 * pads with copies of the last one whose weights are pinned to 0).  NE == 0: any n_energies, weights in wscratch.
 * MODE: PC_MODE_EXPLICIT: photons come from in_start/in_dir/in_elecv (polycap_photon_launch), no retry, no
 * source; PC_MODE_SRC_CIRCULAR / _GENERIC: photons are sampled from the source (circular / elliptical). */
enum { PC_MODE_SRC_CIRCULAR = 0, PC_MODE_SRC_GENERIC = 1, PC_MODE_EXPLICIT = 2 };

/* Source runs with more than 8 energies have a kernel of their own: pc_trace_log_kernel (pc_sweep_kernel.h). */
template <int NE, int MODE, int PITCH>
__global__ void __launch_bounds__(PC_BLOCK, NE == 0 ? PC_MIN_WAVES_NE0 : PC_MIN_WAVES)
pc_trace_kernel(pc_kargs a)
{
	constexpr bool EXPLICIT = (MODE == PC_MODE_EXPLICIT);
	/* static LDS with a compile-time pitch: table reads become ds_read with immediate offsets */
	__shared__ double lds[6*PITCH];
	__shared__ pc_marg4 ldsg[PITCH];
	/* NE == 0: per-workgroup exact weight sums, (lo, hi) per energy, when they fit (a.lds_acc); else global atomics */
	extern __shared__ unsigned long long l_acc[];
	const int npts = a.pm.nmax + 1;
	double *l_z = lds, *l_cap = lds + PITCH, *l_zh = lds + 2*PITCH, *l_cap2 = lds + 3*PITCH;
	double *l_hexd = lds + 4*PITCH, *l_idz = lds + 5*PITCH;
	for (int k = threadIdx.x; k < npts; k += blockDim.x) {
		l_z[k] = a.g_z[k];
		l_cap[k] = a.g_cap[k];
		l_zh[k] = a.g_zh[k];
		l_cap2[k] = a.g_cap2[k];
		l_hexd[k] = a.g_hexd[k];
		l_idz[k] = a.g_idz[k];
		ldsg[k] = a.g_mg[k];
	}
	if (NE != 1 && a.lds_acc)
		for (int k = threadIdx.x; k < 2*a.pm.n_energies; k += blockDim.x) l_acc[k] = 0ull;
	/* NE == 0: the per-energy constants of the cooperative sweeps, staged behind the sums when they fit (a.lds_ec):
	 * every reflection of every photon reads all 6*n_energies of them */
	if (NE == 0 && a.lds_ec) {
		double *l_ec = (double *)(l_acc + 2*a.pm.n_energies);
		for (int k = threadIdx.x; k < 6*a.pm.n_energies; k += blockDim.x) l_ec[k] = a.ec_soa[k];
	}
	__syncthreads();
	pc_tables T;
	T.z = l_z; T.cap = l_cap; T.zh = l_zh; T.cap2 = l_cap2; T.ext = a.g_ext;
	T.hexd = l_hexd; T.idz = l_idz;
	T.mg = ldsg;
	const long long fs = a.img_fs, ss = a.img_ss, ws = a.img_ws;      /* strides of the image store: records or planes (pc_kargs) */
	const pc_params &Pm = a.pm;
	const int ne = (NE > 0) ? NE : Pm.n_energies;
	const int ner = Pm.n_energies;   /* NE > 1 serves any n_energies <= NE: the surplus weights start at 0 and stay there */
	const int lane = threadIdx.x & (PC_WAVE - 1);
	const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;

	pc_photon<NE> ph;
	/* NE == 0: the lane's n_energies weights are contiguous, so the wave can sweep one photon's weights with 64
	 * lanes over energies (coalesced) in the cooperative loops below */
	ph.wmem = (NE > 0) ? nullptr : (a.wscratch + gtid*(long long)a.pm.n_energies);
	ph.wstride = 1;
	ph.wset = 0;
	ph.rc = 0;

	int state = LS_NEED_SLOT;
	long long slot = -1;          /* relative slot index in [0, n_slots) */
	unsigned int attempt = 0;
	double cosalpha0 = 0.;         /* start_electric_vector . start_direction: projection constants of src/polycap-source.c:789-796 */
	/* wave-uniform chunk of slots */
	long long chunk_next = 0, chunk_end = 0;

	/* per-lane totals */
	/* 32 bits per lane are plenty (a lane handles n_slots / total_threads slots); the wave sums are 64-bit */
	/* wave-uniform totals (scalar registers): the lanes' contributions are gathered with ballots / wave sums at the end
	 * of every NEW phase, so no per-lane counter stays live across the march and event loops */
	unsigned long long u_exit = 0, u_not_entered = 0, u_not_trans = 0, u_irefl = 0, u_failed = 0, u_launch = 0;
	unsigned long long u_acc_lo = 0, u_acc_hi = 0;   /* NE == 1: exact 128-bit weight sum; NE > 1 and NE == 0 sum in LDS */

	/* wave-uniform scheduler statistics (diagnostics: lane utilisation per phase type) */
	unsigned long long st_march = 0, st_march_l = 0, st_event = 0, st_event_l = 0, st_new = 0, st_new_l = 0;

	for (;;) {
		const unsigned long long mM = __ballot(state == LS_MARCH);
		const unsigned long long mE = __ballot(state == LS_EVENT);
		const unsigned long long mN = __ballot(state == LS_DONE || state == LS_NEED_SLOT || state == LS_START);
		if ((mM | mE | mN) == 0ull) break;
		const int nM = __popcll(mM), nE = __popcll(mE), nN = __popcll(mN);

		/* NEW is worth a phase once enough lanes wait for it, or when nothing else can run */
		const bool do_new = (nN >= a.new_threshold) || (nM == 0 && nE == 0);
		if (nM > 0 && (nM >= a.event_threshold || (nE == 0 && !do_new))) {
			/* ---------------- MARCH burst: certified node skipping, 6 FMA + 3 LDS reads per node */
			if (Pm.literal) {
				if (state == LS_MARCH)
					state = pc_march_step(T, Pm, ph);
			} else {
				/* lanes fresh from an event or a launch first clear the segment that holds their last interaction point */
				if (state == LS_MARCH && ph.first)
					state = pc_march_step(T, Pm, ph);
				for (int b = 0; b < a.march_burst; b++) {
					unsigned int lanes_in_burst = 0;     /* lanes that take each of these steps (scheduler statistics) */
#pragma unroll
					for (int u = 0; u < PC_MARCH_UNROLL; u++) {
						lanes_in_burst += (unsigned)__popcll(__ballot(state == LS_MARCH));
						if (state == LS_MARCH)
							state = pc_march_step_hot(T, Pm, ph);
					}
					const int cM = __popcll(__ballot(state == LS_MARCH));
					st_march += PC_MARCH_UNROLL; st_march_l += lanes_in_burst;
					if (cM == 0) break;
					if (cM < a.march_stop && (cM != nM || do_new || nE > 0)) break;
				}
			}
		} else if (nE > 0 && !(do_new && nN > nE)) {
			/* ---------------- EVENT: full quadratic of one segment (+ wall hit, Fresnel reflection) */
			st_event += 1; st_event_l += (unsigned)nE;
			if (NE > 0) {
				if (state == LS_EVENT)
					state = pc_event<NE, !EXPLICIT>(T, Pm, a.ec, ph);      /* source runs: pc_fresnel3s (pc_device.h) */
			} else {
				/* many energies: geometry per lane, then the wave sweeps each pending photon's weights with all 64
				 * lanes over energies (coalesced, full lane utilisation whatever the number of pending photons) */
				pc_hit h;
				pc_refl_geom g;
				int pend = 0, res = 0;
				h.nx = h.ny = h.nz = h.cosalfa = 0.; h.ix = 0;
				g.alfa = g.st2 = g.es2 = g.ep2 = g.sd2 = 0.;
				double g_c2 = 0., g_fs = 0., g_fp = 0.;      /* what FORM 3 takes from the geometry (pc_refl_geom3) */
				if (state == LS_EVENT) {
					int st = pc_event_pre(T, Pm, ph, h);
					if (st == PC_ST_REFLECT) {
						if (pc_reflect_geom(ph, h.nx, h.ny, h.nz, g) < 0) { pend = 2; res = -1; }
						else { pend = 1; pc_refl_geom3(g, g_c2, g_fs, g_fp); }
					} else {
						state = st;
					}
				}
				/* the sweep is instantiated once per address space of the per-energy constants (LDS copy or global table): with one
				 * merged pointer the compiler has to use flat loads, which are several times slower than ds_read for LDS data */
				auto sweep = [&](const double *ecs) {
					unsigned long long mP = __ballot(pend == 1);
					const long long wave_gtid0 = gtid - lane;
					if (ne <= 32) {
						/* up to 32 energies: the wave is split into 64/G groups of G = 16 or 32 lanes and sweeps that many
						 * pending photons per pass (lane = photon group x energy); four passes are in flight together so that
						 * the latency of their weight loads (the weights live in HBM/L2) is paid once per batch, not per pass */
						const int G = (ne <= 16) ? 16 : 32, PP = PC_WAVE / G;
						const int sub = lane / G, e = lane - sub*G;
						const unsigned long long gm = (G == 32) ? 0xffffffffull : 0xffffull;
						const pc_energy_const ec = pc_ec_from_soa(ecs, ne, (e < ne) ? e : 0);
						while (mP) {
							int srcv[4], myslot = -1;
	#pragma unroll
							for (int j = 0; j < 4; j++) {
								srcv[j] = -1;
								for (int k = 0; k < PP && mP; k++) {
									const int p = __ffsll((long long)mP) - 1;
									mP &= mP - 1ull;
									if (sub == k) srcv[j] = p;
									if (lane == p) myslot = 4*j + k;
								}
							}
							double wv[4];
	#pragma unroll
							for (int j = 0; j < 4; j++) {
								const int from = (srcv[j] < 0) ? 0 : srcv[j];
								const int wset_p = __shfl(ph.wset, from, PC_WAVE);
								wv[j] = (srcv[j] >= 0 && e < ne && wset_p) ? a.wscratch[(wave_gtid0 + srcv[j])*(long long)ne + e] : 1.0;
							}
							unsigned long long mBv[4], mKv[4];
	#pragma unroll
							for (int j = 0; j < 4; j++) {
								const int from = (srcv[j] < 0) ? 0 : srcv[j];
								const double p_c = __shfl(g.alfa, from, PC_WAVE), p_c2 = __shfl(g_c2, from, PC_WAVE);
								const double p_fs = __shfl(g_fs, from, PC_WAVE), p_fp = __shfl(g_fp, from, PC_WAVE);
								int bad = 0, keep = 0;
								if (srcv[j] >= 0 && e < ne) {
									int r = pc_reflect_energy_sweep(ec, p_c, p_c2, p_fs, p_fp, wv[j]);
									a.wscratch[(wave_gtid0 + srcv[j])*(long long)ne + e] = wv[j];
									bad = (r < 0);
									keep = (r > 0);
								}
								mBv[j] = __ballot(bad);
								mKv[j] = __ballot(keep);
							}
							if (myslot >= 0) {
								const int j = myslot >> 2, k = myslot & 3;
								const unsigned long long m = gm << (k*G);
								const unsigned long long B = (j == 0) ? mBv[0] : ((j == 1) ? mBv[1] : ((j == 2) ? mBv[2] : mBv[3]));
								const unsigned long long K = (j == 0) ? mKv[0] : ((j == 1) ? mKv[1] : ((j == 2) ? mKv[2] : mKv[3]));
								res = (B & m) ? -1 : ((K & m) ? 1 : 0);
							}
						}
					} else
					while (mP) {
						const int p = __ffsll((long long)mP) - 1;
						mP &= mP - 1ull;
						const double p_c = __shfl(g.alfa, p, PC_WAVE), p_c2 = __shfl(g_c2, p, PC_WAVE);
						const double p_fs = __shfl(g_fs, p, PC_WAVE), p_fp = __shfl(g_fp, p, PC_WAVE);
						const int wset_p = __shfl(ph.wset, p, PC_WAVE);
						double *wp = a.wscratch + (wave_gtid0 + p)*(long long)ne;
						int bad = 0, keep = 0;
						for (int e0 = 0; e0 < ne; e0 += PC_WAVE*PC_KE) {
							/* all loads of this sweep are issued before the first Fresnel evaluation */
							double wv[PC_KE];
	#pragma unroll
							for (int k = 0; k < PC_KE; k++) {
								const int e = e0 + k*PC_WAVE + lane;
								wv[k] = (wset_p && e < ne) ? wp[e] : 1.0;
							}
	#pragma unroll
							for (int k = 0; k < PC_KE; k++) {
								const int e = e0 + k*PC_WAVE + lane;
								if (e < ne) {
									const pc_energy_const ec = pc_ec_from_soa(ecs, ne, e);
									int r = pc_reflect_energy_sweep(ec, p_c, p_c2, p_fs, p_fp, wv[k]);
									wp[e] = wv[k];
									bad |= (r < 0);
									keep |= (r > 0);
								}
							}
						}
						const int anybad = __any(bad), anykeep = __any(keep);
						if (lane == p) res = anybad ? -1 : (anykeep ? 1 : 0);
					}
				};
				if (NE == 0 && a.lds_ec) sweep((const double *)(l_acc + 2*a.pm.n_energies));
				else sweep(a.ec_soa);
				if (pend) {
					if (pend == 1) { ph.wset = 1; ph.ex = fabs(ph.ex); ph.ey = fabs(ph.ey); ph.ez = fabs(ph.ez); }
					state = pc_event_post(Pm, ph, h, res);
				}
			}
		} else if (nN > 0 && do_new) {
			st_new += 1; st_new_l += (unsigned)nN;
			/* ---------------- NEW: finalise finished photons, hand out slots, sample + entrance tests */
			int coop = 0;                 /* NE == 0: what the cooperative weight sweep has to do for this lane's photon */
			int f_exit = 0, f_not_entered = 0, f_not_trans = 0, f_failed = 0, f_launch = 0;   /* this lane's contributions */
			unsigned int f_irefl = 0;
			unsigned long long f_w = 0;
			long long done_slot = slot;   /* where the finished photon's images go: its slot, or (compact store) the next free position */
			int ok = 0;                   /* the photon left through the exit window: src/polycap-source.c:758-777 */
			if (state == LS_DONE) {
				const int rc = ph.rc;
				if (EXPLICIT) {
					const long long j = slot;
					a.out_rc[j] = rc;
					if (NE > 0) {
#pragma unroll
						for (int e = 0; e < (NE > 0 ? NE : 1); e++)
							if (e < ner) a.out_weights[j*ner + e] = ph.w[NE > 0 ? e : 0];
					} else
						coop = 1;    /* weights are copied by the cooperative sweep below */
					a.out_exit_coords[3*j] = ph.Px; a.out_exit_coords[3*j+1] = ph.Py; a.out_exit_coords[3*j+2] = ph.Pz;
					a.out_exit_dir[3*j] = ph.dx; a.out_exit_dir[3*j+1] = ph.dy; a.out_exit_dir[3*j+2] = ph.dz;
					a.out_exit_elecv[3*j] = ph.ex; a.out_exit_elecv[3*j+1] = ph.ey; a.out_exit_elecv[3*j+2] = ph.ez;
					a.out_irefl[j] = ph.irefl;
					a.out_dtravel[j] = ph.dtravel;
					state = LS_NEED_SLOT;
				} else {
					if (rc == 0) f_not_trans = 1;
					else if (rc == 2) f_not_entered = 1;
					else if (rc == 1) ok = pc_in_exit_window(Pm, ph);
					/* what a leak run of the same slots is ordered by (pc_leak_auto_order) */
					if (a.work_est) atomicAdd(&a.work_est[slot], (unsigned int)ph.irefl + 1u);
				}
			}
			/* compact store: the exit photons of this phase take the next positions of the planes, one coalesced run per plane */
			const bool compact = !EXPLICIT && a.keep_images && a.img_cursor != nullptr;
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
			if (!EXPLICIT && state == LS_DONE) {
				if (ok) {
					f_exit = 1;
					f_irefl = (unsigned int)ph.irefl;
					if (NE == 1) {
						double w = ph.w[0];
						f_w = (unsigned long long)(w * PC_FIX_SCALE);
						if (a.keep_images) { if (compact) pc_store_wt(a.img_w + done_slot*ws, w); else a.img_w[done_slot*ws] = w; }
					} else if (NE > 1) {
						/* a few energies: exact sums in LDS (2 x u64 per energy), flushed once per workgroup */
#pragma unroll
						for (int e = 0; e < (NE > 0 ? NE : 1); e++) {
							if (e < ner) {
								double w = ph.w[NE > 0 ? e : 0];
								unsigned long long f = (unsigned long long)(w * PC_FIX_SCALE);
								unsigned long long old = atomicAdd(&l_acc[2*e], f);
								if (old + f < old) atomicAdd(&l_acc[2*e + 1], 1ull);
								if (a.keep_images) { if (compact) pc_store_wt(a.img_w + done_slot*ws + e, w); else a.img_w[done_slot*ws + e] = w; }
							}
						}
					} else {
						coop = 1;    /* sums and image weights are handled by the cooperative sweep below */
					}
					if (a.keep_images) {
						/* src/polycap-source.c:900-923 */
						if (compact) {
							/* the start images waited in the lane's own line (written at the launch, below) */
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
						if (a.keep_images && !compact) {
							if (NE > 0) for (int e = 0; e < ner; e++) a.img_w[slot*ws + e] = 0.;
							else coop = 2;   /* zero weights */
						}
						state = LS_NEED_SLOT;
					} else {
						state = LS_START;
					}
				}
			}
			if (NE == 0) {
				/* cooperative sweep over the weights of the photons finalised above: 64 lanes over energies */
				unsigned long long mC = __ballot(coop != 0);
				const long long wave_gtid0 = gtid - lane;
				while (mC) {
					const int p = __ffsll((long long)mC) - 1;
					mC &= mC - 1ull;
					const int what = __shfl(coop, p, PC_WAVE);
					const int wset_p = __shfl(ph.wset, p, PC_WAVE);
					const long long slot_p = __shfl(done_slot, p, PC_WAVE);
					const double *wp = a.wscratch + (wave_gtid0 + p)*(long long)ne;
					for (int e = lane; e < ne; e += PC_WAVE) {
						double w = (what == 2) ? 0. : (wset_p ? wp[e] : 1.0);
						if (EXPLICIT) {
							a.out_weights[slot_p*ne + e] = w;
						} else {
							if (what == 1) {
								unsigned long long f = (unsigned long long)(w * PC_FIX_SCALE);
								if (a.lds_acc) {
									unsigned long long old = atomicAdd(&l_acc[2*e], f);
									if (old + f < old) atomicAdd(&l_acc[2*e + 1], 1ull);
								} else {
									pc_atomic_add128(a.sumw + 2*e, f, 0ull);
								}
							}
							if (a.keep_images) { if (compact) pc_store_wt(a.img_w + slot_p*ws + e, w); else a.img_w[slot_p*ws + e] = w; }
						}
					}
				}
			}
			/* compact store: the positions [c_base, c_base + c_k) are complete (fields above, weights by the lanes or the cooperative
			 * sweep): count them into their blocks once the stores have reached memory, so that the fetch can copy a finished block
			 * while the kernel runs (the launching-wave kernel does the same, pc_producer_kernel.h) */
			if (compact && c_k > 0 && a.blk_done) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0) pc_blocks_written(a, c_base, c_k);
			}
			/* hand out slots: wave-uniform chunk, refilled from the global counter by one lane */
			{
				const unsigned long long need = __ballot(state == LS_NEED_SLOT);
				if (need) {
					const int k = __popcll(need);
					if (chunk_end - chunk_next < k) {
						/* top up: take what is left of the old chunk first, then a fresh chunk */
						long long have = chunk_end - chunk_next;
						long long base_new = 0;
						if (lane == 0) base_new = (long long)atomicAdd(a.work, (unsigned long long)PC_CHUNK);
						base_new = __shfl(base_new, 0, PC_WAVE);
						const int rank = __popcll(need & ((1ull << lane) - 1ull));
						if (state == LS_NEED_SLOT) {
							slot = (rank < have) ? (chunk_next + rank) : (base_new + (rank - have));
						}
						chunk_next = base_new + (k - have);
						chunk_end = base_new + PC_CHUNK;
					} else {
						const int rank = __popcll(need & ((1ull << lane) - 1ull));
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
				if (EXPLICIT) {
					const long long j = slot;
					state = pc_launch_init(T, Pm, ph, a.in_start[3*j], a.in_start[3*j+1], a.in_start[3*j+2],
					                       a.in_dir[3*j], a.in_dir[3*j+1], a.in_dir[3*j+2],
					                       a.in_elecv[3*j], a.in_elecv[3*j+1], a.in_elecv[3*j+2]);
					if (NE > 1) {
#pragma unroll
						for (int e = 0; e < (NE > 0 ? NE : 1); e++) if (e >= ner) ph.w[NE > 0 ? e : 0] = 0.;
					}
				} else {
					pc_start s;
					pc_sample_photon<MODE == PC_MODE_SRC_GENERIC>(Pm, a.seed, (unsigned long long)(a.slot0 + slot), attempt, s);
					state = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
					if (NE > 1) {
#pragma unroll
						for (int e = 0; e < (NE > 0 ? NE : 1); e++) if (e >= ner) ph.w[NE > 0 ? e : 0] = 0.;
					}
					if (state == LS_MARCH) {
						/* src/polycap-source.c:779-798: start images of the attempt that is now inside a capillary;
						 * the slot belongs to this lane, so a later (transmitted) attempt simply overwrites them */
						cosalpha0 = s.ex*s.dx + s.ey*s.dy + s.ez*s.dz;
						if (a.keep_images) {
							double evx, evy;
							pc_start_elecv_image(s, cosalpha0, evx, evy);
							if (a.img_cursor) {
								/* compact store: the position is known when the photon leaves; until then its line */
								double *ls = a.lane_start + gtid*8;
								ls[0] = s.srcx; ls[1] = s.srcy; ls[2] = s.x; ls[3] = s.y; ls[4] = s.dx; ls[5] = s.dy; ls[6] = evx; ls[7] = evy;
							} else {
								pc_write_start_fields<false>(a, slot, s.srcx, s.srcy, s.x, s.y, s.dx, s.dy, evx, evy);
							}
						}
					}
				}
			}
			/* gather this phase's contributions into the wave-uniform totals */
			u_not_trans += (unsigned long long)__popcll(__ballot(f_not_trans));
			u_not_entered += (unsigned long long)__popcll(__ballot(f_not_entered));
			u_failed += (unsigned long long)__popcll(__ballot(f_failed));
			u_launch += (unsigned long long)__popcll(__ballot(f_launch));
			const unsigned long long mX = __ballot(f_exit);
			if (mX) {
				u_exit += (unsigned long long)__popcll(mX);
				u_irefl += pc_wave_sum_u64((unsigned long long)f_irefl);
				if (NE == 1) {
					/* exact 128-bit accumulation of the wave's 64-bit fixed-point weights through two 32-bit partial sums */
					const unsigned long long s_low = pc_wave_sum_u64(f_w & 0xffffffffull), s_high = pc_wave_sum_u64(f_w >> 32);
					const unsigned long long lo = s_low + (s_high << 32);
					const unsigned long long hi = (s_high >> 32) + ((lo < s_low) ? 1ull : 0ull);
					const unsigned long long old = u_acc_lo;
					u_acc_lo = old + lo;
					u_acc_hi += hi + ((u_acc_lo < old) ? 1ull : 0ull);
				}
			}
		}
	}

	if (NE != 1 && !EXPLICIT && a.lds_acc) {
		__syncthreads();          /* every wave of the workgroup has finished its photons */
		for (int e = threadIdx.x; e < a.pm.n_energies; e += blockDim.x)
			if (l_acc[2*e] | l_acc[2*e + 1]) pc_atomic_add128(a.sumw + 2*e, l_acc[2*e], l_acc[2*e + 1]);
	}
	if (!EXPLICIT) {
		/* one set of atomics per wave */
		const unsigned long long v0 = u_exit, v1 = u_not_entered, v2 = u_not_trans, v3 = u_irefl, v4 = u_failed, v5 = u_launch;
		if (lane == 0) {
			atomicAdd(&a.totals->counters[0], v0);
			atomicAdd(&a.totals->counters[1], v1);
			atomicAdd(&a.totals->counters[2], v2);
			atomicAdd(&a.totals->counters[3], v3);
			if (v4) atomicAdd(&a.totals->counters[4], v4);
			atomicAdd(&a.totals->counters[5], v5);
			atomicAdd(&a.totals->phase[0], st_march); atomicAdd(&a.totals->phase[1], st_march_l);
			atomicAdd(&a.totals->phase[2], st_event); atomicAdd(&a.totals->phase[3], st_event_l);
			atomicAdd(&a.totals->phase[4], st_new); atomicAdd(&a.totals->phase[5], st_new_l);
		}
		if (NE == 1 && lane == 0)
			pc_atomic_add128(a.sumw, u_acc_lo, u_acc_hi);
	}
}

#include "pc_pool_kernel.h"
#include "pc_producer_kernel.h"
#ifdef PC_EXPERIMENTS
#include "pc_wave_kernel.h"      /* the one-wave-per-photon experiment (profiles/r03/wave_per_photon_ab.txt): not part of the product build */
#endif
#include "pc_sweep_kernel.h"

/* Compact store: slots that used up their attempts without a transmitted photon wrote nothing (the slot-ordered store writes
 * zero weights for them), so the positions behind the run's cursor hold whatever the buffer held: zero them once the trace
 * kernel has ended -- every plane and the weights -- so that a result with failed slots never carries stale data. */
__global__ void __launch_bounds__(256) pc_compact_tail_kernel(double *soa, long long n_total, int ne, const unsigned long long *cursor)
{
	const long long c = (long long)*cursor;
	if (c >= n_total) return;
	const long long cells = (n_total - c)*(long long)(PC_N_FIELDS + ne);
	for (long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x; t < cells; t += (long long)gridDim.x*blockDim.x) {
		const long long p = c + t / (PC_N_FIELDS + ne);
		const int f = (int)(t % (PC_N_FIELDS + ne));
		if (f < PC_N_FIELDS) soa[(long long)f*n_total + p] = 0.;
		else soa[(long long)PC_N_FIELDS*n_total + p*ne + (f - PC_N_FIELDS)] = 0.;
	}
}

/* Image records (one contiguous record of 17 + n_energies doubles per slot) -> the planes of struct _polycap_images: 17
 * planes of n_total doubles each, then the weights as [slot][n_energies].  A workgroup stages PC_SOA_TILE records in LDS
 * with coalesced reads and writes every plane with coalesced stores: 2 x 144 B of HBM traffic per photon, about a
 * millisecond per 1e7 photons, instead of a strided gather by host threads behind PCIe. */
#define PC_SOA_TILE 128
__global__ void __launch_bounds__(256) pc_soa_kernel(const double *rec, double *soa, long long first, long long count, long long n_total, int ne)
{
	extern __shared__ double tile[];
	const int recd = PC_N_FIELDS + ne;
	const long long j0 = first + (long long)blockIdx.x*PC_SOA_TILE;
	const int m = (int)((first + count - j0 < PC_SOA_TILE) ? first + count - j0 : PC_SOA_TILE);
	if (m <= 0) return;
	const double *src = rec + j0*recd;
	for (int k = threadIdx.x; k < m*recd; k += blockDim.x) tile[k] = src[k];
	__syncthreads();
	for (int k = threadIdx.x; k < m*PC_N_FIELDS; k += blockDim.x) {
		const int plane = k / m, t = k - plane*m;
		soa[(long long)plane*n_total + j0 + t] = tile[t*recd + plane];
	}
	double *w = soa + (long long)PC_N_FIELDS*n_total + j0*ne;
	for (int k = threadIdx.x; k < m*ne; k += blockDim.x) {
		const int t = k / ne, e = k - t*ne;
		w[k] = tile[t*recd + PC_N_FIELDS + e];
	}
}

/* source sampling only (parity of polycap_source_get_photon) */
__global__ void pc_sample_kernel(pc_params pm, unsigned long long seed, long long n,
                                 const long long *slots, const unsigned int *attempts, double *out)
{
	long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	pc_start s;
	if (pm.generic_src) pc_sample_photon<true>(pm, seed, (unsigned long long)slots[j], attempts[j], s);
	else pc_sample_photon<false>(pm, seed, (unsigned long long)slots[j], attempts[j], s);
	double *o = out + 12*j;
	o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.dx; o[4] = s.dy; o[5] = s.dz;
	o[6] = s.ex; o[7] = s.ey; o[8] = s.ez; o[9] = s.srcx; o[10] = s.srcy; o[11] = 0.;
}

/* host threads that turn fetched image records (AoS, `rec` doubles per slot) into the caller's SoA planes */
struct pc_copy_piece { const double *from; size_t slot; size_t n; };      /* n records at `from` belong to slots [slot, slot + n) */

class pc_copy_workers {
public:
	explicit pc_copy_workers(int n)
	{
		for (int t = 1; t < n; t++) threads_.emplace_back([this]() { loop(); });
	}
	~pc_copy_workers()
	{
		{ std::lock_guard<std::mutex> g(m_); stop_ = true; }
		cv_work_.notify_all();
		for (auto &t : threads_) t.join();
	}
	/* scatters every piece; returns when all are done (the calling thread works too) */
	void run(const std::vector<pc_copy_piece> &pieces, void *const *planes, double *weights, size_t rec, size_t ne, double *raw = nullptr)
	{
		{
			std::lock_guard<std::mutex> g(m_);
			pieces_ = &pieces; planes_ = planes; weights_ = weights; rec_ = rec; ne_ = ne; raw_ = raw;
			next_.store(0); busy_ = (int)threads_.size(); gen_++;
		}
		cv_work_.notify_all();
		drain();
		std::unique_lock<std::mutex> g(m_);
		cv_done_.wait(g, [this]() { return busy_ == 0; });
		pieces_ = nullptr;
	}
private:
	void drain()
	{
		const std::vector<pc_copy_piece> &pieces = *pieces_;
		const size_t rec = rec_, ne = ne_;
		const size_t nplanes = rec - ne;
		for (size_t j = next_.fetch_add(1); j < pieces.size(); j = next_.fetch_add(1)) {
			const pc_copy_piece &p = pieces[j];
			if (raw_) { memcpy(raw_ + p.slot*rec, p.from, p.n*rec*sizeof(double)); continue; }     /* records as they are */
			/* plane by plane: strided reads of a piece that fits the cache, contiguous writes (8-byte words: the
			 * reflection count is an int64 plane) */
			for (size_t k = 0; k < nplanes; k++) {
				if (!planes_[k]) continue;
				double *to = (double *)planes_[k] + p.slot;
				const double *from = p.from + k;
				for (size_t i = 0; i < p.n; i++) to[i] = from[i*rec];
			}
			if (weights_) {
				if (ne == 1) {
					double *to = weights_ + p.slot;
					const double *from = p.from + nplanes;
					for (size_t i = 0; i < p.n; i++) to[i] = from[i*rec];
				} else {
					for (size_t i = 0; i < p.n; i++)
						memcpy(weights_ + (p.slot + i)*ne, p.from + i*rec + nplanes, ne*sizeof(double));
				}
			}
		}
	}
	void loop()
	{
		unsigned long seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> g(m_);
				cv_work_.wait(g, [&]() { return stop_ || gen_ != seen; });
				if (stop_) return;
				seen = gen_;
			}
			drain();
			{
				std::lock_guard<std::mutex> g(m_);
				if (--busy_ == 0) cv_done_.notify_all();
			}
		}
	}
	std::vector<std::thread> threads_;
	std::mutex m_;
	std::condition_variable cv_work_, cv_done_;
	const std::vector<pc_copy_piece> *pieces_ = nullptr;
	void *const *planes_ = nullptr;
	double *weights_ = nullptr, *raw_ = nullptr;
	size_t rec_ = 0, ne_ = 0;
	std::atomic<size_t> next_{0};
	unsigned long gen_ = 0;
	int busy_ = 0;
	bool stop_ = false;
};

static thread_local std::string g_last_error;

static int pc_fail(int code, const std::string &msg)
{
	g_last_error = msg;
	return code;
}

#define PC_HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
	return pc_fail(PC_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

#define PC_MAX_PARTS 16

struct pc_hip_ctx {
	int device = 0;
	int n_cu = 256;
	int cu_share = 1;              /* option "cu_share": the context's launches fill n_cu / cu_share compute units.  Tried for device groups that list
	                                * a device m times (m kernels side by side on a quarter of the CUs each): the kernels of one process's streams
	                                * did not overlap (21.7 ms against 15.1 ms one after the other, xos1 5e6 slots, 4 members), so groups leave it at 1 */
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	pc_host_tables host;
	double *d_tables = nullptr;            /* z, cap, zh, cap2, hexd, idz, ext: 7 x npts */
	pc_energy_const *d_ec = nullptr;
	double *d_ec_soa = nullptr;
	pc_marg4 *d_mg = nullptr;              /* block-certificate records, npts */
	pc_drdev *d_dr = nullptr;              /* leak path: chord deviations of cap, npts */
	/* options */
	int literal = 0;
	int event_threshold = 48;      /* lanes that must be marching for a MARCH burst to run before the waiting EVENTs.  With the short flights of
	                                * the current march (5.5 steps) the wave works almost in lockstep: 20 -> 44..48 is 26.2 -> 23.3 ms on xos1
	                                * (profiles/r02/kernel_history.md); optics with long flights (cone.inp) prefer ~24, ellip_l9 with roughness ~32 */
	int new_threshold = 2;
	int march_burst = 16;
	int march_stop = 8;            /* a burst that has started goes on while this many lanes march (0: event_threshold): most flights end within it */
	int blocks_per_cu = 2;
	int block_size = 512;
	int producer = -1;             /* single-energy source runs with a launching wave per workgroup (pc_producer_kernel.h): 1 always, 0 never,
	                                * -1 when photons live long enough for one launching wave per CU to keep up (refl_per_launch, below):
	                                * -5 % on xos1 and ellip_l9, but 2.3x slower on cone.inp, whose photons hardly reflect */
	double refl_per_launch = -1.;  /* EVENT visits (reflections, mostly) per launch in the last source run of this context; < 0: not known.
	                                * A big first run is preceded by a probe of 32768 slots (results unused) */
	int in_probe = 0;
	int last_kernel = -1;          /* pc_hip_last_kernel */
	int last_run_plain = 0;        /* the last run was pc_hip_transmission_run (its counters tell refl_per_launch) */
	int producer_new_min = 2, producer_new_first = 6;
	int march_stats = 0;           /* option "march_stats": the launching-wave kernel counts march steps and their lanes (pc_hip_phase_stats); off in
	                                * production runs, bench.py switches it on for one extra launch outside the timed steps */
	int wave_per_photon = 0;       /* EXPERIMENT (pc_wave_kernel.h): 1 = single-energy histogram-only source runs with one wave per photon */
	int pool = 0;                  /* 1: single-energy source runs on profiles of up to 1024 points use the per-wave photon pool in LDS (pc_pool_kernel.h).
	                                * Was the default up to v14 (+6 %); since flights take 5.5 steps instead of 8.8 the exchanges with the pool cost more
	                                * than its fuller phases save (26.3 ms against 23.3 ms for the one-photon-per-lane kernel) */
	int pool_refill = 20;
	int pool_march_min = 16;
	int pool_event_min = 48;
	int event_march = 0;
	int pool_new_min = 48;
	int lds_ec = 1;                /* many-energy runs: per-energy constants in LDS, one 1024-thread workgroup per CU */
	int batch_reflections = 1;     /* more than 8 energies, source runs: 1 = reflections are logged and a photon's weights swept once per log
	                                * (pc_sweep_kernel.h), 0 = every reflection sweeps the weights at once */
	int log_cap = 0;               /* option "log_cap": reflections per log of pc_trace_log_kernel; 0 = 64 from 64 energies on, 32 below (shorter logs
	                                * leave room in LDS for the logs of more photons per sweep, which few energies need to fill their passes) */
	int sweep_skip = 1;            /* option "sweep_skip": histogram-only log runs stop multiplying a weight below 2^-64 */
	int flush_max = 8;             /* option "flush_max": at most this many finished photons of a wave wait for a common sweep */
	int log_min_energies = 9;      /* option "log_min_energies": source runs with at least this many energies log their reflections (9: every run whose
	                                * weights are not in registers; measured 9 ... 100 energies: +2 ... +130 % against the immediate sweep) */
	int sweep_fuse = 1;            /* option "sweep_fuse": histogram-only log runs add a finished photon's weights to the sums in its sweep; 2 = also when
	                                * its proxies are dead, so that photons the sweep finds dead exercise the take-back pass (tests) */
	double *d_rlog = nullptr;
	size_t rlog_elems = 0;
	int sweep_cert = 0;            /* pc_sweep_certificate has run */
	double sweep_ct_tame = 1.;
	int sweep_n_proxy = 0, sweep_proxy_e[2] = {0, 0};
	/* last run */
	pc_totals *d_totals = nullptr;         /* pc_totals + 2*nE u64 */
	size_t totals_bytes = 0;
	double *d_img = nullptr;               /* image records: n_slots x (17 + n_energies) doubles */
	double *h_stage = nullptr;             /* image fetches: two pinned chunks of records on the host */
	size_t h_stage_elems = 0;
	bool h_stage_pinned = false;           /* false: pinning was refused (locked-memory limit), plain memory is used instead */
	hipEvent_t ev_fetch[2] = {nullptr, nullptr};
	hipStream_t fetch_stream = nullptr;    /* copies of finished parts run beside the kernel of the next part */
	hipStream_t fetch_stream_b = nullptr;  /* compact runs: the planes of a group of blocks alternate between two copy streams */
	hipEvent_t ev_group[2][4] = {{nullptr}};   /* compact runs: end of a group of copies, per stream, ring of 4 */
	hipStream_t stream2 = nullptr;         /* odd parts: a part's first workgroups start as the previous part's last ones leave */
	unsigned long long *d_work = nullptr;  /* one work counter per part */
	hipEvent_t ev_sync = nullptr;
	/* a transmission run can be cut into parts (kernel launches over consecutive slot ranges, same totals): the images of
	 * a finished part are fetched while the next part is traced */
	int run_parts = 1;
	int n_parts = 1;
	long long part_end[PC_MAX_PARTS] = {0};
	hipEvent_t ev_part[PC_MAX_PARTS] = {nullptr};
	bool rec_ev0 = true, rec_ev1 = true;
	int fetch_threads = 0;                 /* host threads that scatter a fetched chunk into the caller's planes; 0 = min(16, cores) */
	long long img_slots = 0;
	int img_valid = 0;
	/* plane (SoA) copy of the image records on the device: 17 planes of soa_slots doubles, then the weights [slot][n_energies].
	 * pc_hip_transmission_images copies from here straight into the caller's (registered) planes -- no host transposition */
	double *d_soa = nullptr;
	long long soa_slots = 0;
	int plane_images = 0;                  /* option "plane_images": runs that keep images write the planes themselves (no records) */
	int run_planes = 0;                    /* the last run did so */
	/* option "compact_images" (with plane_images): exit photons are stored in the order of completion, one coalesced run per
	 * plane and batch, and the planes are published block by block while the kernel runs (pc_kargs::img_cursor) */
	int compact_images = 0;
	int compact_parts = 1;                 /* option "compact_parts": launches a compact run of 4e6 slots or more is traced in (alternating between two
	                                        * streams).  Measured, not adopted: 2 launches 20.95 ms against 19.6 ms for one (1e7 slots; profiles/r04/kernel_history.md) */
	int run_compact = 0;                   /* the last run did so */
	int dst_prepinned = 0;                 /* the caller (a device group) has pinned the destination planes itself: the fetch pins nothing */
	int keep_pinned = 0;                   /* option "keep_pinned": pc_hip_transmission_images leaves the destination planes pinned */
	int slot_ids = 0;                      /* option "slot_ids": compact runs also store which slot sits at which position */
	int blk_shift = 16;                    /* option "block_shift": published blocks of 2^blk_shift positions (65536: 512 KB per plane; the fetch
	                                        * copies all the blocks that are complete at a time in one go) */
	int run_blk_shift = 16;
	long long run_blocks = 0;
	unsigned long long *d_cursor = nullptr;
	unsigned int *d_blk_done = nullptr;
	size_t blk_capacity = 0;
	unsigned int *h_blk_flag = nullptr;    /* host memory mapped into the device: 1 when a block is complete */
	unsigned int *d_blk_flag = nullptr;    /* its device address */
	long long *d_ids = nullptr;
	long long ids_slots = 0;
	double *d_lane_start = nullptr;
	size_t lane_start_elems = 0;
	double *d_wscratch = nullptr;
	size_t wscratch_elems = 0;
	/* explicit-photon calls (polycap_photon_launch, polycap_source_get_photon): one device buffer and one pinned host
	 * buffer, kept between calls, so that a single photon costs two copies and a launch instead of ten copies and
	 * as many allocations */
	double *d_batch = nullptr, *h_batch = nullptr;
	size_t batch_elems = 0;
	bool h_batch_pinned = false;
	long long run_slots = 0;
	int run_pending = 0;
	float last_ms = 0.f;
	/* leak_calc=true runs (pc_leak_kernels.h) */
	int leak_max_depth = 0;                /* frames per lane; default 2*n_shells + 16 (one frame per wall crossed) */
	size_t leak_stack_bytes = (size_t)8 << 30;
	long long leak_capacity = 0;           /* record buffer size of the next run; 0 = 8 per slot, grown on demand */
	long long leak_capacity_used = 0;
	double *d_leak_frames = nullptr;
	size_t leak_frames_elems = 0;
	double *d_leak_records = nullptr;
	size_t leak_records_elems = 0;
	unsigned long long *d_leak_cursor = nullptr;
	double *d_amu = nullptr;
	unsigned int *d_leak_attempts = nullptr;
	unsigned long long *d_leak_timing = nullptr;   /* POLYCAP_LEAK_TIMING diagnostics */
	size_t leak_timing_bytes = 0;
	long long leak_timing_waves = 0;
	unsigned int *d_leak_order = nullptr;  /* order in which the next leak run hands out its slots (pc_hip_leak_set_order) */
	int leak_order = 1;                    /* option: 1 = source runs of >= 196608 slots order their slots by a plain pre-pass, 0 = slot order */
	int leak_order_user = 0;               /* the order was set by the caller */
	unsigned long long leak_order_seed = 0; long long leak_order_slot0 = -1; unsigned int leak_order_attempts = 0;   /* what the automatic order was made for */
	unsigned int *d_work_est = nullptr; long long work_est_n = 0, leak_order_cap = 0;
	int leak_ev0_done = 0;                 /* ev0 of the run in flight was recorded before its pre-pass */
	long long leak_order_n = 0, leak_n_heavy = 0;
	int leak_heavy_lanes = 1, leak_heavy_every = 1;
	int leak_slot_units = 0;               /* option: keep the units of work per slot of leak runs (pc_hip_leak_slot_units) */
	unsigned int *d_leak_slot_units = nullptr;
	long long leak_slot_units_n = 0;
	long long leak_attempt_slots = 0;
	int leak_pending = 0;                  /* a leak transmission run is in flight: wait() collects its events */
	unsigned long long leak_seed = 0;
	long long leak_slot0 = 0, leak_n_slots = 0;
	unsigned int leak_max_attempts = 0;
	int leak_keep_images = 0;
	/* events of the last leak run in the reference's list order, PC_HIP_LEAK_HDR + n_energies doubles each: the extleak list, then
	 * the intleak list, ordered on the device (pc_leak_collect) and kept in pinned host memory */
	double *d_leak_out = nullptr, *h_leak_out = nullptr;
	size_t leak_out_elems = 0;
	void *d_leak_order_tmp = nullptr;
	size_t leak_order_bytes = 0;
	long long leak_n_ext = 0, leak_n_int = 0;
};

static int pc_cus(const pc_hip_ctx *ctx)
{
	const int n = ctx->n_cu / (ctx->cu_share > 0 ? ctx->cu_share : 1);
	return n > 0 ? n : 1;
}

static void pc_fill_common(pc_hip_ctx *ctx, pc_kargs &a)
{
	const size_t npts = (size_t)ctx->host.pm.nmax + 1;
	memset(&a, 0, sizeof(a));
	a.g_z = ctx->d_tables; a.g_cap = ctx->d_tables + npts; a.g_zh = ctx->d_tables + 2*npts;
	a.g_cap2 = ctx->d_tables + 3*npts; a.g_hexd = ctx->d_tables + 4*npts; a.g_idz = ctx->d_tables + 5*npts;
	a.g_ext = ctx->d_tables + 6*npts; a.g_stp = ctx->d_tables + 7*npts; a.g_istp = ctx->d_tables + 8*npts;
	a.g_mg = ctx->d_mg;
	a.g_dr = ctx->d_dr;
	a.ec = ctx->d_ec;
	a.ec_soa = ctx->d_ec_soa;
	a.pm = ctx->host.pm;
	a.pm.literal = ctx->literal;
	a.event_threshold = ctx->event_threshold;
	a.new_threshold = ctx->new_threshold;
	a.march_burst = ctx->march_burst;
	a.march_stop = (ctx->march_stop > 0 && ctx->march_stop < ctx->event_threshold) ? ctx->march_stop : ctx->event_threshold;
	a.pool_refill = ctx->pool_refill;
	a.totals = ctx->d_totals;
	a.work = &ctx->d_totals->next_slot;
	a.sumw = (unsigned long long *)(ctx->d_totals + 1);
}

/* dynamic LDS of the any-n_energies kernel: exact sums and per-energy constants */
static size_t pc_ne0_dyn_lds(size_t ne, int lds_acc, int lds_ec)
{
	return (lds_acc ? 2*ne*sizeof(unsigned long long) : 0) + (lds_ec ? 6*ne*sizeof(double) : 0);
}


/* What pc_trace_log_kernel needs to know about the run's energies (once per context):
 *   ct_tame -- a cosine of the angle to the surface normal above which every energy's reflectivity stays at least 1e-11 below 1
 *     (and, being a ratio of sums of squares weighted by fs, fp >= -1e-16, above 0): no factor of such a reflection can be
 *     rejected by the reference's range test (src/polycap-capil.c:633-637) and every weight only falls.  Found by evaluating
 *     1 - R_s = 4 c Re(g) / |c + g|^2 and 1 - R_p = 4 c Re(conj(g) n^2) / |g + n^2 c|^2, g = sqrt(n^2 - sin^2) from the device's own
 *     constants (pc_fresnel3), in extended precision on 64 points per decade of c from 1 down to 1e-13, per energy: ct_tame =
 *     4 x the largest grid point at which some energy comes closer than 1e-11 (the device's factors are good to ~2e-14).
 *     Below the critical angle 1 - R ~ 4 c beta / (2 delta)^1.5, so for glass ct_tame ~ 1e-11.
 *   proxies -- the energies that reflect best at 3 and at 30 mrad (roughness included): the last ones to fall below 1e-4. */
static void pc_sweep_certificate(pc_hip_ctx *ctx)
{
	if (ctx->sweep_cert) return;
	const std::vector<pc_energy_const> &ec = ctx->host.ec;
	const int ne = (int)ec.size();
	auto refl = [](const pc_energy_const &k, long double c, long double &one_minus_rs, long double &one_minus_rp) {
		const long double zr = c*c - (long double)k.d2, zi = k.n2_im;
		const long double mag = sqrtl(zr*zr + zi*zi);
		long double gr = sqrtl(0.5L*(mag + fabsl(zr))), gi = (gr > 0.0L) ? 0.5L*fabsl(zi)/gr : 0.0L;
		if (zr < 0.0L) { const long double x = gr; gr = gi; gi = x; }
		if (zi < 0.0L) gi = -gi;
		const long double ar = (long double)k.n2_re*c, ai = (long double)k.n2_im*c;
		one_minus_rs = 4.0L*c*gr/((c + gr)*(c + gr) + gi*gi);
		one_minus_rp = 4.0L*(gr*ar + gi*ai)/((gr + ar)*(gr + ar) + (gi + ai)*(gi + ai));
	};
	long double worst = 0.0L;               /* largest grid point at which some energy is not safely below 1 */
	const long double step = powl(10.0L, -1.0L/64.0L);
	for (int e = 0; e < ne; e++) {
		long double c = 1.0L;
		for (int k = 0; k <= 13*64; k++, c *= step) {
			long double a, b;
			refl(ec[e], c, a, b);
			if (!(a >= 1.e-11L && b >= 1.e-11L) || !(a <= 1.0L && b <= 1.0L)) { if (c > worst) worst = c; break; }   /* scanning downwards: the first failure is the largest */
		}
	}
	long double tame = 4.0L*worst;
	if (tame < 4.e-13L) tame = 4.e-13L;     /* below the scanned range nothing is certified */
	ctx->sweep_ct_tame = (tame > 2.0L) ? 2.0 : (double)tame;      /* 2: no reflection is tame (cos theta <= 1) */
	int best[2] = {0, 0};
	const long double at[2] = {3.e-3L, 3.e-2L};
	for (int j = 0; j < 2; j++) {
		long double top = -1.0L;
		for (int e = 0; e < ne; e++) {
			long double a, b;
			refl(ec[e], at[j], a, b);
			const long double x = (long double)ec[e].rough_c*at[j];
			const long double r = (1.0L - 0.5L*(a + b))*expl(-x*x);
			if (r > top) { top = r; best[j] = e; }
		}
	}
	ctx->sweep_proxy_e[0] = best[0]; ctx->sweep_proxy_e[1] = best[1];
	ctx->sweep_n_proxy = (best[0] == best[1]) ? 1 : 2;
	ctx->sweep_cert = 1;
}

/* pc_trace_log_kernel applies to source runs with more than 8 valid energies on a profile of up to 1024 points whose sums and
 * constants fit in LDS beside a stage of at least one log per wave; returns the stage size (doubles per wave), 0 if not */
static size_t pc_log_stage_doubles(const pc_hip_ctx *ctx, int ne, int log_cap)
{
	const size_t fixed = 6*PCS_PITCH*sizeof(double) + PCS_PITCH*sizeof(pc_marg4) + pcs_dyn_lds((size_t)ne, PCS_BLOCK, 0);
	if (fixed >= 163840) return 0;
	size_t per_wave = ((163840 - fixed)/(PCS_BLOCK/PC_WAVE))/sizeof(double);
	const size_t one = PCS_ENT*(size_t)log_cap;
	if (per_wave < one) return 0;
	size_t ps = per_wave/one;
	if (ps > PCS_MAXPS) ps = PCS_MAXPS;
	(void)ctx;
	return ps*one;
}

template <int NE, int MODE>
static int pc_launch_one(pc_hip_ctx *ctx, const pc_kargs &a, int grid)
{
	/* table pitch: 1024 entries (48 KB of LDS) covers the reference's generated profiles (nmax = 999) and its example decks */
	const int block = (int)(a.total_threads / grid);
	const size_t dyn = (NE == 0) ? pc_ne0_dyn_lds((size_t)ctx->host.pm.n_energies, a.lds_acc, a.lds_ec)
	                             : ((NE != 1 && a.lds_acc) ? 2*(size_t)ctx->host.pm.n_energies*sizeof(unsigned long long) : 0);
	if (ctx->host.pm.nmax + 1 <= 1024)
		hipLaunchKernelGGL((pc_trace_kernel<NE, MODE, 1024>), dim3(grid), dim3(block), dyn, ctx->stream, a);
	else if (NE <= 1)   /* long profiles: only the NE = 1 and the any-n_energies kernels are built for the 2048 pitch */
		hipLaunchKernelGGL((pc_trace_kernel<(NE <= 1 ? NE : 0), MODE, PC_MAX_PITCH>), dim3(grid), dim3(block), dyn, ctx->stream, a);
	else
		return pc_fail(PC_HIP_ERR_INVALID, "internal: register-weight kernels are built for profiles of up to 1024 points");
	PC_HIP_CHECK(hipGetLastError());
	return PC_HIP_OK;
}

/* the pool kernel serves single-energy source runs on profiles of up to 1024 points (what its packed records hold) */
template <int MODE>
static bool pc_pool_applies(const pc_hip_ctx *ctx, const pc_kargs &a)
{
	const pc_params &pm = ctx->host.pm;
	return MODE != PC_MODE_EXPLICIT && ctx->pool && pm.n_energies == 1 && !ctx->literal && pm.nmax + 1 <= PQ_PITCH
	    && a.max_attempts <= (1u << 24) && pm.n_shells < 16000. && a.n_slots < (1ll << 39);
}

template <int MODE>
static void pc_launch_pool(pc_hip_ctx *ctx, const pc_kargs &a, int grid)
{
	if constexpr (MODE != PC_MODE_EXPLICIT)
		hipLaunchKernelGGL((pc_trace_pool_kernel<MODE>), dim3(grid), dim3(PQ_BLOCK), 0, ctx->stream, a);
}

template <int MODE>
static int pc_launch_kernel(pc_hip_ctx *ctx, pc_kargs &a, long long n_items)
{
	const int ne = ctx->host.pm.n_energies;
	/* weights in registers for up to 8 energies (kernels NE = 1, 4, 8), in the per-lane scratch beyond */
	const int kne = (ne == 1) ? 1 : ((ne <= 4 && ctx->host.pm.nmax + 1 <= 1024) ? 4 : ((ne <= 8 && ctx->host.pm.nmax + 1 <= 1024) ? 8 : 0));
	a.lds_acc = (ne != 1 && 2*(size_t)ne*sizeof(unsigned long long) <= 16384) ? 1 : 0;
	/* many energies on a profile of up to 1024 points: one workgroup of 1024 threads per CU (the same 16 waves as two of
	 * 512) leaves room in LDS for the per-energy constants next to the tables and the sums */
	a.lds_ec = (kne == 0 && a.lds_acc && ctx->lds_ec && ctx->host.pm.nmax + 1 <= 1024 && 64*(size_t)ne <= 28672) ? 1 : 0;
	bool all_valid = true;
	a.sweep_rough = 0;
	for (const pc_energy_const &c : ctx->host.ec) {
		if (c.valid == 0.) all_valid = false;
		if (c.rough_c != 0.) a.sweep_rough = 1;
	}
	/* source runs with more than 8 (valid) energies log their reflections (pc_trace_log_kernel); an explicit photon reports its
	 * state at the absorbing reflection, which the logging kernel's speculation overwrites */
	const bool want_log = kne == 0 && ctx->lds_ec && ctx->host.pm.nmax + 1 <= 1024 && ne >= ctx->log_min_energies && ctx->batch_reflections && all_valid
	                      && MODE != PC_MODE_EXPLICIT;
	if constexpr (MODE != PC_MODE_EXPLICIT) {
#ifdef PC_EXPERIMENTS
		if (ctx->wave_per_photon && ne == 1 && !a.keep_images && ctx->host.pm.nmax + 1 <= 1024) {
			/* the experiment of pc_wave_kernel.h: one wave per photon, 16 waves per CU */
			long long want = (n_items + 3) / 4;
			int grid = (int)(want < 4ll*ctx->n_cu ? want : 4ll*ctx->n_cu);
			if (grid < 1) grid = 1;
			a.total_threads = (long long)grid * PCW_BLOCK;
			if (ctx->rec_ev0) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
			hipLaunchKernelGGL((pc_trace_wave_kernel<MODE>), dim3(grid), dim3(PCW_BLOCK), 0, ctx->stream, a);
			ctx->last_kernel = 3;
			PC_HIP_CHECK(hipGetLastError());
			if (ctx->rec_ev1) PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
			return PC_HIP_OK;
		}
#endif
		const pc_params &pm = ctx->host.pm;
		const bool want_producer = ctx->producer == 1 || (ctx->producer < 0 && !ctx->in_probe && ctx->refl_per_launch >= PC3_MIN_REFL);
		if (want_producer && pm.n_energies == 1 && !ctx->literal && pm.nmax + 1 <= PC3_PITCH && a.max_attempts <= (1u << 24)
		    && pm.n_shells < 16000. && a.n_slots < (1ll << 39)) {
			const long long per_block = (long long)PC3_CONSUMERS*PC_WAVE;
			long long want = (n_items + per_block - 1) / per_block;
			const long long per_cu = (PC3_BLOCK > 512) ? 1 : 2;
			int grid = (int)(want < per_cu*pc_cus(ctx) ? want : per_cu*pc_cus(ctx));
			if (grid < 1) grid = 1;
			a.total_threads = (long long)grid * PC3_BLOCK;
			a.event_threshold = ctx->event_threshold;
			a.new_threshold = ctx->producer_new_min;
			a.pool_event_min = ctx->producer_new_first;
			if (ctx->rec_ev0) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
			if (ctx->march_stats) hipLaunchKernelGGL((pc_trace_producer_kernel<MODE, true>), dim3(grid), dim3(PC3_BLOCK), 0, ctx->stream, a);
			else hipLaunchKernelGGL((pc_trace_producer_kernel<MODE, false>), dim3(grid), dim3(PC3_BLOCK), 0, ctx->stream, a);
			ctx->last_kernel = 2;
			PC_HIP_CHECK(hipGetLastError());
			if (ctx->rec_ev1) PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
			return PC_HIP_OK;
		}
	}
	if (pc_pool_applies<MODE>(ctx, a)) {
		/* one 1024-thread workgroup per CU; a wave holds 64 + PQ_P photons */
		const long long per_block = (long long)PQ_WAVES*(PC_WAVE + PQ_P);
		long long want = (n_items + per_block - 1) / per_block;
		int grid = (int)(want < pc_cus(ctx) ? want : pc_cus(ctx));
		if (grid < 1) grid = 1;
		a.total_threads = (long long)grid * PQ_BLOCK;
		a.event_threshold = ctx->pool_march_min;
		a.pool_event_min = ctx->pool_event_min;
		a.event_march = ctx->event_march;
		a.new_threshold = ctx->pool_new_min;
		if (ctx->rec_ev0) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
		pc_launch_pool<MODE>(ctx, a, grid);
		ctx->last_kernel = 1;
		PC_HIP_CHECK(hipGetLastError());
		if (ctx->rec_ev1) PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
		return PC_HIP_OK;
	}
	if constexpr (MODE != PC_MODE_EXPLICIT) {
		/* log capacity: 64 reflections (32 below 64 energies), halved while not even one log per wave fits in the stage beside
		 * the constants of very many energies (beyond ~450) */
		int log_cap = ctx->log_cap > 0 ? ctx->log_cap : (ne >= 64 ? 64 : 32);
		size_t stage = want_log ? pc_log_stage_doubles(ctx, ne, log_cap) : 0;
		while (want_log && !stage && ctx->log_cap <= 0 && log_cap > 8) {
			log_cap /= 2;
			stage = pc_log_stage_doubles(ctx, ne, log_cap);
		}
		if (stage) {
			/* reflections are logged, a photon's weights swept once per log (pc_sweep_kernel.h): one workgroup of 12 waves per CU */
			pc_sweep_certificate(ctx);
			long long want = (n_items + PCS_BLOCK - 1) / PCS_BLOCK;
			int grid = (int)(want < pc_cus(ctx) ? want : pc_cus(ctx));
			if (grid < 1) grid = 1;
			a.total_threads = (long long)grid * PCS_BLOCK;
			const size_t need_w = (size_t)ne * (size_t)a.total_threads, need_l = 3*(size_t)log_cap * (size_t)a.total_threads;
			if (need_w > ctx->wscratch_elems) {
				if (ctx->d_wscratch) PC_HIP_CHECK(hipFree(ctx->d_wscratch));
				ctx->d_wscratch = nullptr; ctx->wscratch_elems = 0;
				if (hipMalloc(&ctx->d_wscratch, need_w*sizeof(double)) != hipSuccess) return pc_fail(PC_HIP_ERR_MEMORY, "could not allocate the per-lane weight scratch");
				ctx->wscratch_elems = need_w;
			}
			if (need_l > ctx->rlog_elems) {
				if (ctx->d_rlog) PC_HIP_CHECK(hipFree(ctx->d_rlog));
				ctx->d_rlog = nullptr; ctx->rlog_elems = 0;
				if (hipMalloc(&ctx->d_rlog, need_l*sizeof(double)) != hipSuccess) return pc_fail(PC_HIP_ERR_MEMORY, "could not allocate the reflection logs");
				ctx->rlog_elems = need_l;
			}
			a.wscratch = ctx->d_wscratch;
			a.rlog = ctx->d_rlog;
			a.log_cap = log_cap;
			a.stage_ps = (int)(stage/(PCS_ENT*(size_t)log_cap));
			{
				/* photons that wait for a sweep before one is run: the fewest (up to the stage's capacity) whose last pass leaves at
				 * most 3 % of the round's lanes idle, else the count that leaves the fewest */
				int best = 1;
				double best_w = 2.;
				for (int n = 1; n <= a.stage_ps && n <= ctx->flush_max; n++) {
					const double w = (double)((64 - (n*ne) % 64) % 64) / (double)(n*ne);
					if (w < best_w - 1e-12) { best_w = w; best = n; }
					if (w <= 0.03) { best = n; break; }
				}
				a.flush_min = best;
			}
			a.n_proxy = ctx->sweep_n_proxy; a.proxy_e[0] = ctx->sweep_proxy_e[0]; a.proxy_e[1] = ctx->sweep_proxy_e[1];
			a.ct_tame = ctx->sweep_ct_tame;
			a.sweep_skip = (ctx->sweep_skip && !a.keep_images) ? 1 : 0;
			a.sweep_fuse = a.keep_images ? 0 : ctx->sweep_fuse;
			if (ctx->rec_ev0) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
			hipLaunchKernelGGL((pc_trace_log_kernel<MODE>), dim3(grid), dim3(PCS_BLOCK), pcs_dyn_lds((size_t)ne, PCS_BLOCK, stage), ctx->stream, a);
			ctx->last_kernel = 4;
			PC_HIP_CHECK(hipGetLastError());
			if (ctx->rec_ev1) PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
			return PC_HIP_OK;
		}
	}
	long long max_blocks = (long long)pc_cus(ctx) * ((kne == 0) ? 1 : ctx->blocks_per_cu);
	const int block = ctx->block_size;
	long long want_blocks = (n_items + block - 1) / block;
	int grid = (int)(want_blocks < max_blocks ? want_blocks : max_blocks);
	if (grid < 1) grid = 1;
	a.total_threads = (long long)grid * block;
	if (kne == 0) {
		size_t need = (size_t)ne * (size_t)a.total_threads;
		if (need > ctx->wscratch_elems) {
			if (ctx->d_wscratch) PC_HIP_CHECK(hipFree(ctx->d_wscratch));
			ctx->d_wscratch = nullptr; ctx->wscratch_elems = 0;
			hipError_t e = hipMalloc(&ctx->d_wscratch, need*sizeof(double));
			if (e != hipSuccess) return pc_fail(PC_HIP_ERR_MEMORY, "could not allocate the per-lane weight scratch");
			ctx->wscratch_elems = need;
		}
		a.wscratch = ctx->d_wscratch;
	}
	if (ctx->rec_ev0) PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
	ctx->last_kernel = 0;
	int st = (kne == 1) ? pc_launch_one<1, MODE>(ctx, a, grid) : (kne == 4) ? pc_launch_one<4, MODE>(ctx, a, grid)
	       : (kne == 8) ? pc_launch_one<8, MODE>(ctx, a, grid) : pc_launch_one<0, MODE>(ctx, a, grid);
	if (st) return st;
	if (ctx->rec_ev1) PC_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
	return PC_HIP_OK;
}

#include "pc_leak_kernels.h"

static int pc_transmission_enqueue_leak(pc_hip_ctx *ctx);
static int pc_leak_auto_order(pc_hip_ctx *ctx);

extern "C" {

int pc_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char *pc_hip_last_error(void)
{
	return g_last_error.c_str();
}

void pc_hip_ctx_destroy(pc_hip_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	if (ctx->d_tables) (void)hipFree(ctx->d_tables);
	if (ctx->d_ec) (void)hipFree(ctx->d_ec);
	if (ctx->d_ec_soa) (void)hipFree(ctx->d_ec_soa);
	if (ctx->d_mg) (void)hipFree(ctx->d_mg);
	if (ctx->d_dr) (void)hipFree(ctx->d_dr);
	if (ctx->d_totals) (void)hipFree(ctx->d_totals);
	if (ctx->d_img) (void)hipFree(ctx->d_img);
	if (ctx->d_soa) (void)hipFree(ctx->d_soa);
	if (ctx->h_stage) { if (ctx->h_stage_pinned) (void)hipHostFree(ctx->h_stage); else free(ctx->h_stage); }
	for (int k = 0; k < 2; k++) if (ctx->ev_fetch[k]) (void)hipEventDestroy(ctx->ev_fetch[k]);
	for (int k = 0; k < PC_MAX_PARTS; k++) if (ctx->ev_part[k]) (void)hipEventDestroy(ctx->ev_part[k]);
	if (ctx->fetch_stream) (void)hipStreamDestroy(ctx->fetch_stream);
	if (ctx->fetch_stream_b) (void)hipStreamDestroy(ctx->fetch_stream_b);
	for (int k = 0; k < 2; k++) for (int j = 0; j < 4; j++) if (ctx->ev_group[k][j]) (void)hipEventDestroy(ctx->ev_group[k][j]);
	if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
	if (ctx->d_work) (void)hipFree(ctx->d_work);
	if (ctx->ev_sync) (void)hipEventDestroy(ctx->ev_sync);
	if (ctx->d_wscratch) (void)hipFree(ctx->d_wscratch);
	if (ctx->d_rlog) (void)hipFree(ctx->d_rlog);
	if (ctx->d_cursor) (void)hipFree(ctx->d_cursor);
	if (ctx->d_blk_done) (void)hipFree(ctx->d_blk_done);
	if (ctx->h_blk_flag) (void)hipHostFree(ctx->h_blk_flag);
	if (ctx->d_ids) (void)hipFree(ctx->d_ids);
	if (ctx->d_lane_start) (void)hipFree(ctx->d_lane_start);
	if (ctx->d_batch) (void)hipFree(ctx->d_batch);
	if (ctx->h_batch) { if (ctx->h_batch_pinned) (void)hipHostFree(ctx->h_batch); else free(ctx->h_batch); }
	if (ctx->d_leak_frames) (void)hipFree(ctx->d_leak_frames);
	if (ctx->d_leak_records) (void)hipFree(ctx->d_leak_records);
	if (ctx->d_leak_out) (void)hipFree(ctx->d_leak_out);
	if (ctx->h_leak_out) (void)hipHostFree(ctx->h_leak_out);
	if (ctx->d_leak_order_tmp) (void)hipFree(ctx->d_leak_order_tmp);
	if (ctx->d_leak_cursor) (void)hipFree(ctx->d_leak_cursor);
	if (ctx->d_amu) (void)hipFree(ctx->d_amu);
	if (ctx->d_leak_attempts) (void)hipFree(ctx->d_leak_attempts);
	if (ctx->d_leak_timing) (void)hipFree(ctx->d_leak_timing);
	if (ctx->d_leak_order) (void)hipFree(ctx->d_leak_order);
	if (ctx->d_work_est) (void)hipFree(ctx->d_work_est);
	if (ctx->d_leak_slot_units) (void)hipFree(ctx->d_leak_slot_units);
	if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
	if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int pc_hip_ctx_create(const pc_hip_problem *problem, int device, pc_hip_ctx **out)
{
	if (!out) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_ctx_create: ctx must not be NULL");
	*out = nullptr;
	int ndev = pc_hip_device_count();
	if (ndev <= 0) return pc_fail(PC_HIP_ERR_NO_DEVICE, "pc_hip_ctx_create: no HIP device available (the trace path has no CPU fallback)");
	if (device < 0 || device >= ndev) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_ctx_create: device index out of range");
	pc_hip_ctx *ctx = new pc_hip_ctx();
	ctx->device = device;
	std::string err;
	int rc = pc_build_tables(problem, ctx->host, err);
	if (rc) { delete ctx; return pc_fail(rc, "pc_hip_ctx_create: " + err); }
	const size_t npts = (size_t)ctx->host.pm.nmax + 1;
	if (npts > PC_MAX_PITCH) { delete ctx; return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_ctx_create: profile too long for the LDS tables (nmax <= 2047)"); }
#define PC_CTX_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::string m = std::string(#expr) + ": " + hipGetErrorString(_e); pc_hip_ctx_destroy(ctx); return pc_fail(PC_HIP_ERR_RUNTIME, m); } } while (0)
	PC_CTX_CHECK(hipSetDevice(device));
	hipDeviceProp_t prop;
	PC_CTX_CHECK(hipGetDeviceProperties(&prop, device));
	ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	PC_CTX_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
	if (const char *e = getenv("POLYCAP_PRODUCER"))          /* tests: force (1) or forbid (0) the launching-wave kernel */
		if (*e == '0' || *e == '1') ctx->producer = *e - '0';
	PC_CTX_CHECK(hipEventCreate(&ctx->ev0));
	PC_CTX_CHECK(hipEventCreate(&ctx->ev1));
	PC_CTX_CHECK(hipMalloc(&ctx->d_tables, 9*npts*sizeof(double)));
	const std::vector<double> *src[9] = { &ctx->host.z, &ctx->host.cap, &ctx->host.zh, &ctx->host.cap2, &ctx->host.hexd, &ctx->host.idz, &ctx->host.ext,
	                                      &ctx->host.stp, &ctx->host.istp };
	for (int k = 0; k < 9; k++)
		PC_CTX_CHECK(hipMemcpy(ctx->d_tables + k*npts, src[k]->data(), npts*sizeof(double), hipMemcpyHostToDevice));
	PC_CTX_CHECK(hipMalloc(&ctx->d_mg, npts*sizeof(pc_marg4)));
	PC_CTX_CHECK(hipMemcpy(ctx->d_mg, ctx->host.mg.data(), npts*sizeof(pc_marg4), hipMemcpyHostToDevice));
	PC_CTX_CHECK(hipMalloc(&ctx->d_dr, npts*sizeof(pc_drdev)));
	PC_CTX_CHECK(hipMemcpy(ctx->d_dr, ctx->host.dr.data(), npts*sizeof(pc_drdev), hipMemcpyHostToDevice));
	{
		/* at least 8 entries: the register-weight kernels read NE constants whatever n_energies is (surplus = copies of the last) */
		std::vector<pc_energy_const> ecp(ctx->host.ec);
		while (ecp.size() < 8) ecp.push_back(ecp.back());
		PC_CTX_CHECK(hipMalloc(&ctx->d_ec, ecp.size()*sizeof(pc_energy_const)));
		PC_CTX_CHECK(hipMemcpy(ctx->d_ec, ecp.data(), ecp.size()*sizeof(pc_energy_const), hipMemcpyHostToDevice));
	}
	{
		const size_t ne = ctx->host.ec.size();
		/* field-major constants of the weight sweeps (FORM 3): d2, Re n^2, Im n^2, max((Im n^2)^2, 2^-200), rough_c, valid, rough_c^2 */
		std::vector<double> soa(7*ne);
		for (size_t e = 0; e < ne; e++) {
			const pc_energy_const &c = ctx->host.ec[e];
			soa[e] = c.d2; soa[ne + e] = c.n2_re; soa[2*ne + e] = c.n2_im; soa[3*ne + e] = c.zi2;
			soa[4*ne + e] = c.rough_c; soa[5*ne + e] = c.valid; soa[6*ne + e] = c.rough_k2;
		}
		PC_CTX_CHECK(hipMalloc(&ctx->d_ec_soa, soa.size()*sizeof(double)));
		PC_CTX_CHECK(hipMemcpy(ctx->d_ec_soa, soa.data(), soa.size()*sizeof(double), hipMemcpyHostToDevice));
	}
	ctx->leak_max_depth = (int)std::min(65536.0, 2.0*ctx->host.pm.n_shells + 16.0);
	ctx->totals_bytes = sizeof(pc_totals) + 2*ctx->host.ec.size()*sizeof(unsigned long long);
	PC_CTX_CHECK(hipMalloc(&ctx->d_totals, ctx->totals_bytes));
	PC_CTX_CHECK(hipMemset(ctx->d_totals, 0, ctx->totals_bytes));
#undef PC_CTX_CHECK
	*out = ctx;
	return PC_HIP_OK;
}

int pc_hip_set_option(pc_hip_ctx *ctx, const char *name, int64_t value)
{
	if (!ctx || !name) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_set_option: NULL argument");
	std::string n(name);
	if (n == "literal_march") ctx->literal = value ? 1 : 0;
	else if (n == "event_threshold") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "event_threshold must be in [1,64]"); ctx->event_threshold = (int)value; }
	else if (n == "new_threshold") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "new_threshold must be in [1,64]"); ctx->new_threshold = (int)value; }
	else if (n == "march_stop") { if (value < 0 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "march_stop must be in [0,64]"); ctx->march_stop = (int)value; }
	else if (n == "march_burst") { if (value < 1) return pc_fail(PC_HIP_ERR_INVALID, "march_burst must be >= 1"); ctx->march_burst = (int)value; }
	else if (n == "block_size") { if (value < 64 || value > PC_BLOCK || (value % 64) != 0) return pc_fail(PC_HIP_ERR_INVALID, "block_size must be a multiple of 64 up to the compiled maximum"); ctx->block_size = (int)value; }
	else if (n == "blocks_per_cu") { if (value < 1 || value > 8) return pc_fail(PC_HIP_ERR_INVALID, "blocks_per_cu must be in [1,8]"); ctx->blocks_per_cu = (int)value; }
	else if (n == "lds_ec") ctx->lds_ec = value ? 1 : 0;
	else if (n == "batch_reflections") ctx->batch_reflections = value ? 1 : 0;
	else if (n == "log_cap") { if (value < 0 || value > 255) return pc_fail(PC_HIP_ERR_INVALID, "log_cap must be in [0,255] (0 = automatic)"); ctx->log_cap = (int)value; }
	else if (n == "sweep_skip") ctx->sweep_skip = value ? 1 : 0;
	else if (n == "log_min_energies") { if (value < 9) return pc_fail(PC_HIP_ERR_INVALID, "log_min_energies must be >= 9 (up to 8 energies have their weights in registers)"); ctx->log_min_energies = (int)value; }
	else if (n == "flush_max") { if (value < 1 || value > 16) return pc_fail(PC_HIP_ERR_INVALID, "flush_max must be in [1,16]"); ctx->flush_max = (int)value; }
	else if (n == "sweep_fuse") { if (value < 0 || value > 2) return pc_fail(PC_HIP_ERR_INVALID, "sweep_fuse must be 0, 1 or 2"); ctx->sweep_fuse = (int)value; }
	else if (n == "plane_images") ctx->plane_images = value ? 1 : 0;
	else if (n == "compact_images") ctx->compact_images = value ? 1 : 0;
	else if (n == "compact_parts") { if (value < 1 || value > PC_MAX_PARTS) return pc_fail(PC_HIP_ERR_INVALID, "compact_parts must be in [1,16]"); ctx->compact_parts = (int)value; }
	else if (n == "slot_ids") ctx->slot_ids = value ? 1 : 0;
	else if (n == "keep_pinned") ctx->keep_pinned = value ? 1 : 0;
	else if (n == "block_shift") { if (value < 7 || value > 30) return pc_fail(PC_HIP_ERR_INVALID, "block_shift must be in [7,30]"); ctx->blk_shift = (int)value; }
	else if (n == "run_parts") { if (value < 1 || value > PC_MAX_PARTS) return pc_fail(PC_HIP_ERR_INVALID, "run_parts must be in [1,16]"); ctx->run_parts = (int)value; }
	else if (n == "fetch_threads") { if (value < 0 || value > 256) return pc_fail(PC_HIP_ERR_INVALID, "fetch_threads must be in [0,256]"); ctx->fetch_threads = (int)value; }
	else if (n == "pool") ctx->pool = value ? 1 : 0;
	else if (n == "wave_per_photon") {
#ifdef PC_EXPERIMENTS
		ctx->wave_per_photon = value ? 1 : 0;
#else
		if (value) return pc_fail(PC_HIP_ERR_INVALID, "wave_per_photon: the experiment kernel is compiled only with -DPC_EXPERIMENTS (scripts/analysis/wave_per_photon_ab.py)");
#endif
	}
	else if (n == "march_stats") ctx->march_stats = value ? 1 : 0;
	else if (n == "cu_share") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "cu_share must be in [1,64]"); ctx->cu_share = (int)value; }
	else if (n == "producer") { if (value < -1 || value > 1) return pc_fail(PC_HIP_ERR_INVALID, "producer must be -1 (automatic), 0 or 1"); ctx->producer = (int)value; }
	else if (n == "producer_new_min") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "producer_new_min must be in [1,64]"); ctx->producer_new_min = (int)value; }
	else if (n == "producer_new_first") { if (value < 1 || value > 65) return pc_fail(PC_HIP_ERR_INVALID, "producer_new_first must be in [1,65]"); ctx->producer_new_first = (int)value; }
	else if (n == "pool_refill") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "pool_refill must be in [1,64]"); ctx->pool_refill = (int)value; }
	else if (n == "event_march") { if (value < 0 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "event_march must be in [0,64]"); ctx->event_march = (int)value; }
	else if (n == "pool_event_min") { if (value < 1 || value > 128) return pc_fail(PC_HIP_ERR_INVALID, "pool_event_min must be in [1,128]"); ctx->pool_event_min = (int)value; }
	else if (n == "pool_new_min") { if (value < 1 || value > 128) return pc_fail(PC_HIP_ERR_INVALID, "pool_new_min must be in [1,128]"); ctx->pool_new_min = (int)value; }
	else if (n == "pool_march_min") { if (value < 1 || value > 64) return pc_fail(PC_HIP_ERR_INVALID, "pool_march_min must be in [1,64]"); ctx->pool_march_min = (int)value; }
	else if (n == "leak_max_depth") { if (value < 2 || value > (1 << 20)) return pc_fail(PC_HIP_ERR_INVALID, "leak_max_depth must be in [2, 2^20]"); ctx->leak_max_depth = (int)value; }
	else if (n == "leak_stack_mb") { if (value < 1) return pc_fail(PC_HIP_ERR_INVALID, "leak_stack_mb must be >= 1"); ctx->leak_stack_bytes = (size_t)value << 20; }
	else if (n == "leak_order") { ctx->leak_order = value != 0; }
	else if (n == "leak_slot_units") { ctx->leak_slot_units = value != 0; }
	else if (n == "leak_heavy_lanes") { if (value < 0 || value > PC_WAVE) return pc_fail(PC_HIP_ERR_INVALID, "leak_heavy_lanes must be in 0..64"); ctx->leak_heavy_lanes = (int)value; }
	else if (n == "leak_heavy_every") { if (value < 0) return pc_fail(PC_HIP_ERR_INVALID, "leak_heavy_every must be >= 0"); ctx->leak_heavy_every = (int)value; }
	else if (n == "leak_capacity") { if (value < 0) return pc_fail(PC_HIP_ERR_INVALID, "leak_capacity must be >= 0"); ctx->leak_capacity = (long long)value; }
	else return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_set_option: unknown option " + n);
	return PC_HIP_OK;
}

/* device + pinned host buffer of at least `elems` doubles for the explicit-photon calls */
static int pc_batch_buffers(pc_hip_ctx *ctx, size_t elems)
{
	if (ctx->batch_elems >= elems) return PC_HIP_OK;
	if (ctx->d_batch) (void)hipFree(ctx->d_batch);
	if (ctx->h_batch) { if (ctx->h_batch_pinned) (void)hipHostFree(ctx->h_batch); else free(ctx->h_batch); }
	ctx->d_batch = ctx->h_batch = nullptr; ctx->batch_elems = 0;
	const size_t want = elems < 4096 ? 4096 : elems + elems/4;
	if (hipMalloc(&ctx->d_batch, want*sizeof(double)) != hipSuccess) { ctx->d_batch = nullptr; return pc_fail(PC_HIP_ERR_MEMORY, "explicit-photon batch: device allocation failed"); }
	ctx->h_batch_pinned = true;
	if (hipHostMalloc(&ctx->h_batch, want*sizeof(double), hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		ctx->h_batch_pinned = false;
		ctx->h_batch = (double *)malloc(want*sizeof(double));
		if (!ctx->h_batch) { (void)hipFree(ctx->d_batch); ctx->d_batch = nullptr; return pc_fail(PC_HIP_ERR_MEMORY, "explicit-photon batch: host allocation failed"); }
	}
	ctx->batch_elems = want;
	return PC_HIP_OK;
}

static int pc_launch_photons_impl(pc_hip_ctx *ctx, int64_t n, const double *start_coords, const double *start_dir, const double *start_elecv,
                                  int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                                  int64_t *i_refl, double *d_travel, int leak)
{
	if (!ctx || n < 0 || !start_coords || !start_dir || !start_elecv || !rc || !weights || !exit_coords || !exit_dir || !exit_elecv || !i_refl || !d_travel)
		return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_launch_photons: NULL argument");
	if (n == 0) return PC_HIP_OK;
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	const size_t N = (size_t)n;
	/* one device buffer: 3 inputs [3N], rc [N ints padded], weights [N*ne], 3 outputs [3N], irefl [N], dtravel [N]; the pinned
	 * host buffer mirrors it: one copy in (the inputs), one copy out (everything behind them) */
	const size_t doubles = 9*N + N + N*ne + 9*N + N + N;
	{
		int st = pc_batch_buffers(ctx, doubles);
		if (st) return st;
	}
	double *d = ctx->d_batch, *h = ctx->h_batch;
	double *d_start = d, *d_dir = d + 3*N, *d_ev = d + 6*N;
	int *d_rc = (int *)(d + 9*N);
	double *d_w = d + 10*N, *d_ec = d_w + N*ne, *d_ed = d_ec + 3*N, *d_ee = d_ed + 3*N;
	long long *d_ir = (long long *)(d_ee + 3*N);
	double *d_dt = (double *)(d_ir + N);
	int status = PC_HIP_OK;
#define PC_LP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { status = pc_fail(PC_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); goto done; } } while (0)
	{
		memcpy(h, start_coords, 3*N*sizeof(double));
		memcpy(h + 3*N, start_dir, 3*N*sizeof(double));
		memcpy(h + 6*N, start_elecv, 3*N*sizeof(double));
		PC_LP_CHECK(hipMemcpyAsync(d, h, 9*N*sizeof(double), hipMemcpyHostToDevice, ctx->stream));
		PC_LP_CHECK(hipMemsetAsync(ctx->d_totals, 0, ctx->totals_bytes, ctx->stream));
		pc_kargs a;
		pc_fill_common(ctx, a);
		a.n_slots = n; a.slot0 = 0; a.max_attempts = 1; a.keep_images = 0;
		a.in_start = d_start; a.in_dir = d_dir; a.in_elecv = d_ev;
		a.out_rc = d_rc; a.out_weights = d_w; a.out_exit_coords = d_ec; a.out_exit_dir = d_ed; a.out_exit_elecv = d_ee;
		a.out_irefl = d_ir; a.out_dtravel = d_dt;
		if (leak) {
			/* polycap_photon_launch(..., leak_calc=true): rerun with a larger record buffer until every event fits */
			long long capacity = ctx->leak_capacity > 0 ? ctx->leak_capacity : std::max<long long>(4096, (16 + 8*(long long)ne)*n);
			ctx->leak_slot0 = 0;
			for (;;) {
				ctx->leak_capacity_used = capacity;
				ctx->leak_ev0_done = 0;
				status = pc_leak_enqueue<PC_MODE_EXPLICIT>(ctx, a, n, capacity);
				if (status) goto done;
				PC_LP_CHECK(hipStreamSynchronize(ctx->stream));
				long long needed = 0;
				status = pc_leak_collect(ctx, n, true, &needed);
				if (status == 1) { capacity = needed + needed/4 + 1024; status = PC_HIP_OK; continue; }
				if (status) goto done;
				break;
			}
		} else {
			status = pc_launch_kernel<PC_MODE_EXPLICIT>(ctx, a, n);
			if (status) goto done;
		}
		PC_LP_CHECK(hipMemcpyAsync(h + 9*N, d + 9*N, (doubles - 9*N)*sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		PC_LP_CHECK(hipStreamSynchronize(ctx->stream));
		{
			const double *o = h + 9*N;        /* rc (ints, padded to N doubles), weights, exit coords / dir / elecv, irefl, dtravel */
			memcpy(rc, o, N*sizeof(int)); o += N;
			memcpy(weights, o, N*ne*sizeof(double)); o += N*ne;
			memcpy(exit_coords, o, 3*N*sizeof(double)); o += 3*N;
			memcpy(exit_dir, o, 3*N*sizeof(double)); o += 3*N;
			memcpy(exit_elecv, o, 3*N*sizeof(double)); o += 3*N;
			memcpy(i_refl, o, N*sizeof(long long)); o += N;
			memcpy(d_travel, o, N*sizeof(double));
		}
		/* The kernels work with the normalised electric vector (polycap_refl_polar normalises it in place at the first
		 * reflection, src/polycap-capil.c:492-494); a photon that never reached a reflection keeps the caller's vector */
		for (size_t j = 0; j < N; j++) {
			const bool untouched = (rc[j] == -2) || (!leak && (rc[j] == 2 || (rc[j] == 1 && i_refl[j] == 0)));
			if (untouched)
				for (int c = 0; c < 3; c++) exit_elecv[3*j + c] = start_elecv[3*j + c];
		}
	}
done:
#undef PC_LP_CHECK
	ctx->img_valid = 0;
	return status;
}

int pc_hip_launch_photons(pc_hip_ctx *ctx, int64_t n, const double *start_coords, const double *start_dir, const double *start_elecv,
                          int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                          int64_t *i_refl, double *d_travel)
{
	return pc_launch_photons_impl(ctx, n, start_coords, start_dir, start_elecv, rc, weights, exit_coords, exit_dir, exit_elecv, i_refl, d_travel, 0);
}

int pc_hip_launch_photons_leak(pc_hip_ctx *ctx, int64_t n, const double *start_coords, const double *start_dir, const double *start_elecv,
                               int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                               int64_t *i_refl, double *d_travel)
{
	return pc_launch_photons_impl(ctx, n, start_coords, start_dir, start_elecv, rc, weights, exit_coords, exit_dir, exit_elecv, i_refl, d_travel, 1);
}

int pc_hip_sample_photons(pc_hip_ctx *ctx, uint64_t seed, int64_t n, const int64_t *slots, const uint32_t *attempts, double *out)
{
	if (!ctx || n < 0 || !slots || !attempts || !out) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_sample_photons: NULL argument");
	if (n == 0) return PC_HIP_OK;
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	const size_t N = (size_t)n;
	/* layout in the batch buffers: slots [N int64], attempts [N uint32, padded to N/2 + 1 doubles], out [12 N] */
	const size_t off_att = N, off_out = N + N/2 + 1, total = off_out + 12*N;
	{
		int st = pc_batch_buffers(ctx, total);
		if (st) return st;
	}
	double *d = ctx->d_batch, *h = ctx->h_batch;
	memcpy(h, slots, N*sizeof(long long));
	memcpy(h + off_att, attempts, N*sizeof(unsigned int));
	int status = PC_HIP_OK;
	hipError_t e = hipMemcpyAsync(d, h, off_out*sizeof(double), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(pc_sample_kernel, dim3((unsigned)((N + 255)/256)), dim3(256), 0, ctx->stream,
		                   ctx->host.pm, (unsigned long long)seed, (long long)n, (const long long *)d, (const unsigned int *)(d + off_att), d + off_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(h + off_out, d + off_out, 12*N*sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_sample_photons: ") + hipGetErrorString(e));
	else memcpy(out, h + off_out, 12*N*sizeof(double));
	return status;
}

/* where the launch for slots [lo, ...) of a run of n_total slots stores its images: see pc_kargs::img */
static void pc_set_img(const pc_hip_ctx *ctx, pc_kargs &a, long long lo, long long n_total, bool keep, bool planes)
{
	const long long ne = ctx->host.pm.n_energies, rec = PC_N_FIELDS + ne;
	if (!keep) { a.img = a.img_w = nullptr; a.img_ss = a.img_fs = a.img_ws = 0; return; }
	if (planes) {
		a.img = ctx->d_soa + lo; a.img_ss = 1; a.img_fs = n_total;
		a.img_w = ctx->d_soa + (long long)PC_N_FIELDS*n_total + lo*ne; a.img_ws = ne;
	} else {
		a.img = ctx->d_img + lo*rec; a.img_ss = rec; a.img_fs = 1;
		a.img_w = a.img + PC_N_FIELDS; a.img_ws = rec;
	}
}

/* device plane buffer for a run of n_slots (see pc_soa_kernel) */
static int pc_soa_ensure(pc_hip_ctx *ctx, long long n_slots)
{
	if (ctx->d_soa && ctx->soa_slots >= n_slots) return PC_HIP_OK;
	if (ctx->d_soa) (void)hipFree(ctx->d_soa);
	ctx->d_soa = nullptr; ctx->soa_slots = 0;
	const size_t bytes = ((size_t)PC_N_FIELDS + (size_t)ctx->host.pm.n_energies) * (size_t)n_slots * sizeof(double);
	if (hipMalloc(&ctx->d_soa, bytes) != hipSuccess) { (void)hipGetLastError(); ctx->d_soa = nullptr; return PC_HIP_ERR_MEMORY; }
	ctx->soa_slots = n_slots;
	return PC_HIP_OK;
}

/* records of slots [lo, lo + count) of the current run -> planes (pitch = the run's n_slots), on `stream` */
static int pc_soa_launch(pc_hip_ctx *ctx, hipStream_t stream, long long lo, long long count, long long n_total)
{
	const int ne = ctx->host.pm.n_energies;
	const size_t lds = (size_t)PC_SOA_TILE*(PC_N_FIELDS + ne)*sizeof(double);
	if (lds > 65536) return PC_HIP_ERR_INVALID;
	const unsigned blocks = (unsigned)((count + PC_SOA_TILE - 1)/PC_SOA_TILE);
	hipLaunchKernelGGL(pc_soa_kernel, dim3(blocks), dim3(256), lds, stream, ctx->d_img, ctx->d_soa, lo, count, n_total, ne);
	PC_HIP_CHECK(hipGetLastError());
	return PC_HIP_OK;
}

/* First slot of part k of `parts`.  The fetch of the images can start when the first part is done and has the last part
 * left when the kernel ends, so with three or more parts the first and the last are half the size of the others. */
static long long pc_part_begin(long long n_slots, int parts, int k)
{
	if (k <= 0) return 0;
	if (k >= parts) return n_slots;
	if (parts < 3) return n_slots*k/parts;
	const double unit = 1.0/(double)(parts - 1);           /* 1/2 + (parts - 2) + 1/2 units */
	return (long long)((double)n_slots*unit*((double)k - 0.5));
}

/* image planes: 17 double-sized planes of n_slots entries followed by the weights plane */
static const int PC_N_PLANES = 17;

/* How long do photons live on this optic?  32768 slots with the default kernel (3 ms, results unused) set refl_per_launch, by
 * which the context -- or, for a device group, every member -- picks the kernel of its source runs. */
static int pc_probe_lifetime(pc_hip_ctx *ctx, uint64_t seed, int64_t slot0, uint32_t max_attempts)
{
	ctx->in_probe = 1;
	int64_t c[6];
	int st = pc_hip_transmission_run(ctx, seed, slot0, 32768, max_attempts, 0);
	if (st == PC_HIP_OK) st = pc_hip_transmission_totals(ctx, nullptr, c, nullptr);
	ctx->in_probe = 0;
	return (st == PC_HIP_ERR_ATTEMPTS) ? PC_HIP_OK : st;
}

/* buffers of a compact run of n_slots (pc_kargs::img_cursor): position counter, per-block counters, the host-visible block
 * flags, the lanes' start-image lines and, on request, the plane of slot indices; counters and flags are cleared */
static int pc_compact_prepare(pc_hip_ctx *ctx, long long n_slots)
{
	/* the block flags are cleared from the host below: a run of this context that is still in flight would set flags of its own
	 * after that (and the fetch of the new run would copy blocks the new kernel has not written) */
	if (ctx->run_pending) {
		int st = pc_hip_transmission_wait(ctx, nullptr);
		if (st) return st;
	}
	const int shift = ctx->blk_shift;
	const size_t blocks = (size_t)((n_slots + (1ll << shift) - 1) >> shift);
	if (!ctx->d_cursor) PC_HIP_CHECK(hipMalloc(&ctx->d_cursor, sizeof(unsigned long long)));
	if (ctx->blk_capacity < blocks) {
		if (ctx->d_blk_done) (void)hipFree(ctx->d_blk_done);
		if (ctx->h_blk_flag) (void)hipHostFree(ctx->h_blk_flag);
		ctx->d_blk_done = nullptr; ctx->h_blk_flag = nullptr; ctx->d_blk_flag = nullptr; ctx->blk_capacity = 0;
		const size_t cap = blocks + blocks/2 + 16;
		PC_HIP_CHECK(hipMalloc(&ctx->d_blk_done, cap*sizeof(unsigned int)));
		if (hipHostMalloc(&ctx->h_blk_flag, cap*sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
			(void)hipGetLastError();
			PC_HIP_CHECK(hipHostMalloc(&ctx->h_blk_flag, cap*sizeof(unsigned int), hipHostMallocMapped));
		}
		PC_HIP_CHECK(hipHostGetDevicePointer((void **)&ctx->d_blk_flag, ctx->h_blk_flag, 0));
		ctx->blk_capacity = cap;
	}
	if (ctx->slot_ids && ctx->ids_slots < n_slots) {
		if (ctx->d_ids) (void)hipFree(ctx->d_ids);
		ctx->d_ids = nullptr; ctx->ids_slots = 0;
		if (hipMalloc(&ctx->d_ids, (size_t)n_slots*sizeof(long long)) != hipSuccess) { (void)hipGetLastError(); return pc_fail(PC_HIP_ERR_MEMORY, "pc_hip_transmission_run: could not allocate the slot-index plane"); }
		ctx->ids_slots = n_slots;
	}
	{
		/* one 64-byte line per lane of the largest launch the context makes */
		const size_t lanes = (size_t)ctx->n_cu * (size_t)std::max(ctx->blocks_per_cu*ctx->block_size, 1024);
		if (ctx->lane_start_elems < 8*lanes) {
			if (ctx->d_lane_start) (void)hipFree(ctx->d_lane_start);
			ctx->d_lane_start = nullptr; ctx->lane_start_elems = 0;
			PC_HIP_CHECK(hipMalloc(&ctx->d_lane_start, 8*lanes*sizeof(double)));
			ctx->lane_start_elems = 8*lanes;
		}
	}
	ctx->run_blk_shift = shift;
	ctx->run_blocks = (long long)blocks;
	memset(ctx->h_blk_flag, 0, blocks*sizeof(unsigned int));      /* the previous run has been waited for (above): nobody looks at them now */
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_cursor, 0, sizeof(unsigned long long), ctx->stream));
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_blk_done, 0, blocks*sizeof(unsigned int), ctx->stream));
	return PC_HIP_OK;
}

int pc_hip_transmission_run(pc_hip_ctx *ctx, uint64_t seed, int64_t slot0, int64_t n_slots, uint32_t max_attempts, int keep_images)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_run: ctx must not be NULL");
	if (n_slots < 1 || slot0 < 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_run: n_slots must be >= 1 and slot0 >= 0");
	if (max_attempts < 1) max_attempts = 1;
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	if (ctx->producer < 0 && ctx->refl_per_launch < 0. && !ctx->in_probe && ne == 1 && n_slots >= 2000000) {
		/* first big run of the context: 32768 slots with the default kernel tell how long photons live here (3 ms, results unused) */
		int st = pc_probe_lifetime(ctx, seed, slot0, max_attempts);
		if (st != PC_HIP_OK) return st;
	}
	ctx->last_run_plain = 1;
	pc_kargs a;
	pc_fill_common(ctx, a);
	ctx->img_valid = 0;
	/* option "plane_images" (set by polycap_source_get_transmission_efficiencies): the kernels store the planes of struct
	 * _polycap_images themselves -- 18 scattered 8-byte stores per exit photon instead of two contiguous pieces of a record,
	 * 5 % of the HBM bandwidth at most -- and the fetch is a plain copy of planes into the caller's pinned memory */
	const bool planes = keep_images && ctx->plane_images && pc_soa_ensure(ctx, n_slots) == PC_HIP_OK;
	ctx->run_planes = planes ? 1 : 0;
	const bool compact = planes && ctx->compact_images;
	ctx->run_compact = compact ? 1 : 0;
	if (compact) {
		int st = pc_compact_prepare(ctx, n_slots);
		if (st) return st;
		a.img_cursor = ctx->d_cursor;
		a.img_ids = ctx->slot_ids ? ctx->d_ids : nullptr;
		a.blk_done = ctx->d_blk_done;
		a.blk_flag = ctx->d_blk_flag;
		a.blk_shift = ctx->run_blk_shift;
		a.img_n = n_slots;
		a.lane_start = ctx->d_lane_start;
	}
	if (keep_images && !planes) {
		if (ctx->img_slots < n_slots) {
			if (ctx->d_img) PC_HIP_CHECK(hipFree(ctx->d_img));
			ctx->d_img = nullptr; ctx->img_slots = 0;
			size_t bytes = ((size_t)PC_N_PLANES + ne) * (size_t)n_slots * sizeof(double);
			if (hipMalloc(&ctx->d_img, bytes) != hipSuccess)
				return pc_fail(PC_HIP_ERR_MEMORY, "pc_hip_transmission_run: could not allocate the image planes; use keep_images=0");
			ctx->img_slots = n_slots;
		}
	}
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_totals, 0, ctx->totals_bytes, ctx->stream));
	a.seed = seed; a.max_attempts = max_attempts; a.keep_images = keep_images ? 1 : 0;
	/* parts: consecutive slot ranges traced by consecutive launches into the same totals and image records (a photon
	 * depends on its global slot number only, so the result does not depend on the cut) */
	/* A compact run publishes its blocks itself: it needs no parts for the copy-back.  Option "compact_parts" > 1 traces a big one
	 * as that many launches on two streams all the same (the positions, block counters and totals are the run's, so a launch simply
	 * goes on where the one before leaves off): meant to cover the tail of one launch with the head of the next, it costs more
	 * than it saves (default 1). */
	int parts = (keep_images && ctx->run_parts > 1 && !compact) ? ctx->run_parts : 1;
	if (compact && ctx->compact_parts > 1 && n_slots >= 4000000) parts = ctx->compact_parts;
	if (parts > PC_MAX_PARTS) parts = PC_MAX_PARTS;
	if ((long long)parts > n_slots / 65536) parts = (int)(n_slots / 65536);
	if (parts < 1) parts = 1;
	ctx->n_parts = compact ? 1 : parts;     /* what the fetch goes by: a compact run is fetched block by block whatever its launches */
	const size_t rec = (size_t)PC_N_PLANES + ne;
	int status = PC_HIP_OK;
	hipStream_t main_stream = ctx->stream;
	struct restore_ctx {          /* the launch helpers read the stream and the event flags from the context */
		pc_hip_ctx *c; hipStream_t s;
		~restore_ctx() { c->stream = s; c->rec_ev0 = c->rec_ev1 = true; }
	} restore{ctx, main_stream};
	if (parts > 1) {
		/* Parts alternate between two streams.  Every launch fills the device with persistent workgroups, so the
		 * workgroups of part k+1 start exactly as those of part k run out of slots and leave: the tail of one part (its
		 * longest photons) is covered by the head of the next, and the parts still finish in order. */
		if (!ctx->stream2) PC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
		if (!ctx->ev_sync) PC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming));
		if (!ctx->d_work) PC_HIP_CHECK(hipMalloc(&ctx->d_work, PC_MAX_PARTS*sizeof(unsigned long long)));
		PC_HIP_CHECK(hipMemsetAsync(ctx->d_work, 0, PC_MAX_PARTS*sizeof(unsigned long long), main_stream));
		PC_HIP_CHECK(hipEventRecord(ctx->ev0, main_stream));
		PC_HIP_CHECK(hipEventRecord(ctx->ev_sync, main_stream));
		PC_HIP_CHECK(hipStreamWaitEvent(ctx->stream2, ctx->ev_sync, 0));     /* totals and counters are zero */
		ctx->rec_ev0 = ctx->rec_ev1 = false;
	}
	for (int k = 0; k < parts && status == PC_HIP_OK; k++) {
		const long long lo = pc_part_begin(n_slots, parts, k), hi = pc_part_begin(n_slots, parts, k + 1);
		a.slot0 = slot0 + lo; a.n_slots = hi - lo;
		pc_set_img(ctx, a, compact ? 0 : lo, n_slots, keep_images != 0, planes);      /* compact: positions are the run's, not the part's */
		if (parts > 1) {
			a.work = ctx->d_work + k;
			ctx->stream = (k & 1) ? ctx->stream2 : main_stream;
		}
		if (compact) ctx->rec_ev1 = false;       /* the kernel time ends behind the tail kernel below */
		status = ctx->host.pm.generic_src ? pc_launch_kernel<PC_MODE_SRC_GENERIC>(ctx, a, hi - lo)
		                                  : pc_launch_kernel<PC_MODE_SRC_CIRCULAR>(ctx, a, hi - lo);
		ctx->part_end[k] = hi;
		if (status == PC_HIP_OK && parts > 1) {
			if (!ctx->ev_part[k]) PC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_part[k], hipEventDisableTiming));
			PC_HIP_CHECK(hipEventRecord(ctx->ev_part[k], ctx->stream));
		}
	}
	ctx->stream = main_stream;
	if (parts > 1 && status == PC_HIP_OK) {
		/* the main stream ends after every part: wait() synchronises it, and the kernel time runs to here */
		for (int k = 0; k < parts; k++)
			if (k & 1) PC_HIP_CHECK(hipStreamWaitEvent(main_stream, ctx->ev_part[k], 0));
	}
	if (compact && status == PC_HIP_OK) {
		hipLaunchKernelGGL(pc_compact_tail_kernel, dim3(64), dim3(256), 0, main_stream, ctx->d_soa, (long long)n_slots, (int)ne, ctx->d_cursor);
		PC_HIP_CHECK(hipGetLastError());
	}
	if ((parts > 1 || compact) && status == PC_HIP_OK)
		PC_HIP_CHECK(hipEventRecord(ctx->ev1, main_stream));
	if (status) return status;
	ctx->run_slots = n_slots;
	ctx->run_pending = 1;
	ctx->img_valid = keep_images ? 1 : 0;
	return PC_HIP_OK;
}

int pc_hip_transmission_run_leak(pc_hip_ctx *ctx, uint64_t seed, int64_t slot0, int64_t n_slots, uint32_t max_attempts, int keep_images)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_run_leak: ctx must not be NULL");
	if (n_slots < 1 || slot0 < 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_run_leak: n_slots must be >= 1 and slot0 >= 0");
	if (max_attempts < 1) max_attempts = 1;
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	ctx->img_valid = 0;
	if (keep_images && ctx->img_slots < n_slots) {
		if (ctx->d_img) PC_HIP_CHECK(hipFree(ctx->d_img));
		ctx->d_img = nullptr; ctx->img_slots = 0;
		size_t bytes = ((size_t)PC_N_PLANES + ne) * (size_t)n_slots * sizeof(double);
		if (hipMalloc(&ctx->d_img, bytes) != hipSuccess)
			return pc_fail(PC_HIP_ERR_MEMORY, "pc_hip_transmission_run_leak: could not allocate the image planes; use keep_images=0");
		ctx->img_slots = n_slots;
	}
	ctx->n_parts = 1;
	ctx->run_planes = 0;
	ctx->last_run_plain = 0;
	ctx->leak_seed = seed; ctx->leak_slot0 = slot0; ctx->leak_n_slots = n_slots;
	ctx->leak_max_attempts = max_attempts; ctx->leak_keep_images = keep_images ? 1 : 0;
	/* record buffer: events per slot grow with the number of energies (a leak is kept while ANY energy holds >= 1e-4):
	 * 9 per slot at one energy, 29 at seven on the reference's test optic; a run that outgrows the buffer is repeated, so
	 * the first guess is generous (16 + 8 n_energies records per slot) */
	ctx->leak_capacity_used = ctx->leak_capacity > 0 ? ctx->leak_capacity : std::max<long long>(65536, (16 + 8*(long long)ne)*n_slots);
	ctx->leak_ev0_done = 0;
	int status = pc_leak_auto_order(ctx);
	if (status) return status;
	status = pc_transmission_enqueue_leak(ctx);
	if (status) return status;
	ctx->run_slots = n_slots;
	ctx->run_pending = 1;
	ctx->leak_pending = 1;
	ctx->img_valid = keep_images ? 1 : 0;
	return PC_HIP_OK;
}

int pc_hip_transmission_wait(pc_hip_ctx *ctx, float *kernel_ms)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_wait: ctx must not be NULL");
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	PC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	while (ctx->leak_pending) {
		/* leak run: fetch and order its events; a run that outgrew the record buffer is repeated with a larger one
		 * (the photon streams are counter-based, so the repetition is the same run) */
		long long needed = 0;
		int st = pc_leak_collect(ctx, ctx->leak_n_slots, false, &needed);
		if (st == 1) {
			ctx->leak_capacity_used = needed + needed/4 + 1024;
			st = pc_transmission_enqueue_leak(ctx);
			if (st) { ctx->leak_pending = 0; ctx->run_pending = 0; return st; }
			PC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
			continue;
		}
		ctx->leak_pending = 0;
		if (st) { ctx->run_pending = 0; return st; }
	}
	if (ctx->run_pending) {
		float ms = 0.f;
		PC_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
		ctx->last_ms = ms;
		ctx->run_pending = 0;
	}
	if (kernel_ms) *kernel_ms = ctx->last_ms;
	return PC_HIP_OK;
}

/* is this host address registered with the HIP runtime (pinned)? */
int pc_hip_host_is_pinned(const void *p)
{
	if (p == nullptr) return 0;
	unsigned int flags = 0;
	if (hipHostGetFlags(&flags, const_cast<void *>(p)) == hipSuccess) return 1;
	(void)hipGetLastError();
	return 0;
}

int pc_hip_leak_counts(pc_hip_ctx *ctx, int64_t *n_ext, int64_t *n_int)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_counts: ctx must not be NULL");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	if (n_ext) *n_ext = ctx->leak_n_ext;
	if (n_int) *n_int = ctx->leak_n_int;
	return PC_HIP_OK;
}

int pc_hip_leak_events(pc_hip_ctx *ctx, int kind, int64_t first, int64_t count, double *records)
{
	if (!ctx || (count > 0 && !records)) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_events: NULL argument");
	if (kind != 0 && kind != 1) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_events: kind must be 0 (extleak) or 1 (intleak)");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	const long long have = kind == 0 ? ctx->leak_n_ext : ctx->leak_n_int;
	if (first < 0 || count < 0 || first + count > have) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_events: range out of bounds");
	const size_t stride = PC_HIP_LEAK_HDR + (size_t)ctx->host.pm.n_energies;
	const double *src = ctx->h_leak_out + (kind == 0 ? 0 : (size_t)ctx->leak_n_ext*stride);
	if (count) memcpy(records, src + (size_t)first*stride, (size_t)count*stride*sizeof(double));
	return PC_HIP_OK;
}

int pc_hip_leak_events_view(pc_hip_ctx *ctx, int kind, const double **records, int64_t *count)
{
	if (!ctx || !records || !count) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_events_view: NULL argument");
	if (kind != 0 && kind != 1) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_events_view: kind must be 0 (extleak) or 1 (intleak)");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	const size_t stride = PC_HIP_LEAK_HDR + (size_t)ctx->host.pm.n_energies;
	*count = kind == 0 ? ctx->leak_n_ext : ctx->leak_n_int;
	*records = (*count > 0) ? ctx->h_leak_out + (kind == 0 ? 0 : (size_t)ctx->leak_n_ext*stride) : nullptr;
	return PC_HIP_OK;
}

double pc_hip_fixed_to_double(uint64_t lo, uint64_t hi)
{
	long double v = (long double)hi * 18446744073709551616.0L + (long double)lo;
	return (double)(v / 4611686018427387904.0L);
}

int pc_hip_transmission_totals(pc_hip_ctx *ctx, double *sum_weights, int64_t counters[6], uint64_t *sumw_fixed)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_totals: ctx must not be NULL");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	std::vector<unsigned char> buf(ctx->totals_bytes);
	PC_HIP_CHECK(hipMemcpy(buf.data(), ctx->d_totals, ctx->totals_bytes, hipMemcpyDeviceToHost));
	const pc_totals *t = (const pc_totals *)buf.data();
	const unsigned long long *sw = (const unsigned long long *)(t + 1);
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	if (counters)
		for (int k = 0; k < 6; k++) counters[k] = (int64_t)t->counters[k];
	if (t->counters[5] > 0 && ctx->last_run_plain)
		ctx->refl_per_launch = (double)t->phase[3] / (double)t->counters[5];        /* segment visits (reflections, mostly) per launch,
		                                                                              * absorbed photons included: what the next run chooses its kernel by */
	for (size_t e = 0; e < ne; e++) {
		if (sum_weights) sum_weights[e] = pc_hip_fixed_to_double(sw[2*e], sw[2*e+1]);
		if (sumw_fixed) { sumw_fixed[2*e] = sw[2*e]; sumw_fixed[2*e+1] = sw[2*e+1]; }
	}
	if (t->counters[4] != 0)
		return pc_fail(PC_HIP_ERR_ATTEMPTS, "pc_hip_transmission_totals: some slots exhausted max_attempts without a transmitted photon");
	return PC_HIP_OK;
}

int pc_hip_last_kernel(pc_hip_ctx *ctx)
{
	return ctx ? ctx->last_kernel : -1;
}

int pc_hip_phase_stats(pc_hip_ctx *ctx, int64_t stats[6])
{
	if (!ctx || !stats) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_phase_stats: NULL argument");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	pc_totals t;
	PC_HIP_CHECK(hipMemcpy(&t, ctx->d_totals, sizeof(t), hipMemcpyDeviceToHost));
	for (int k = 0; k < 6; k++) stats[k] = (int64_t)t.phase[k];
	return PC_HIP_OK;
}

int pc_hip_sweep_stats(pc_hip_ctx *ctx, int64_t stats[4], double *ct_tame, int proxies[2])
{
	if (!ctx || !stats) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_sweep_stats: NULL argument");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	pc_totals t;
	PC_HIP_CHECK(hipMemcpy(&t, ctx->d_totals, sizeof(t), hipMemcpyDeviceToHost));
	const bool log_run = ctx->last_kernel == 4;
	stats[0] = log_run ? (int64_t)t.phase[6] : 0;
	stats[1] = log_run ? (int64_t)t.phase[7] : 0;
	stats[2] = log_run ? (int64_t)t.counters[6] : 0;
	stats[3] = log_run ? (int64_t)t.counters[7] : 0;
	if (ct_tame) *ct_tame = ctx->sweep_cert ? ctx->sweep_ct_tame : -1.;
	if (proxies) { proxies[0] = ctx->sweep_cert ? ctx->sweep_proxy_e[0] : -1; proxies[1] = (ctx->sweep_cert && ctx->sweep_n_proxy > 1) ? ctx->sweep_proxy_e[1] : -1; }
	return PC_HIP_OK;
}

/* The stream of the image copies.  HIP multiplexes a process's streams over a few hardware queues (4 by default): with one
 * more context alive in the process the copies of a finished part landed in the queue of the next part's kernel and waited
 * for it (40 -> 54 ms per 1e7 photons through the C API, scripts/analysis/api_time2.py).  A stream of the highest priority
 * gets a queue of its own class, apart from the kernels' queues. */
/* is this host address pinned already (an earlier fetch with "keep_pinned", a slab from the host pool, the caller's own
 * hipHostRegister / hipHostMalloc)?  Pinning a range inside an existing registration a second time is not something to try. */
static bool pc_host_is_pinned(void *p)
{
	unsigned int flags = 0;
	if (hipHostGetFlags(&flags, p) == hipSuccess) return true;
	(void)hipGetLastError();
	return false;
}

/* Pins the host ranges (address, bytes) for the copy engine: rounded out to pages, overlapping or touching ranges merged (small
 * planes from malloc share pages with their neighbours, and a page cannot be registered twice), ranges that are pinned already
 * left alone.  All or nothing: when one range cannot be pinned the ones pinned here are released again and false is returned
 * -- a destination that is pinned only in part is not something to hand to hipMemcpyAsync.  `pinned` receives what to
 * hipHostUnregister afterwards. */
static bool pc_pin_ranges(std::vector<std::pair<char *, size_t>> ranges, unsigned int flags, std::vector<void *> &pinned)
{
	const uintptr_t page = 4096;
	std::vector<std::pair<uintptr_t, uintptr_t>> r;
	for (auto &x : ranges) {
		if (!x.first || !x.second) continue;
		const uintptr_t lo = (uintptr_t)x.first & ~(page - 1), hi = ((uintptr_t)x.first + x.second + page - 1) & ~(page - 1);
		r.emplace_back(lo, hi);
	}
	std::sort(r.begin(), r.end());
	std::vector<std::pair<uintptr_t, uintptr_t>> m;
	for (auto &x : r) {
		if (!m.empty() && x.first <= m.back().second) m.back().second = std::max(m.back().second, x.second);
		else m.push_back(x);
	}
	const size_t before = pinned.size();
	for (auto &x : m) {
		if (pc_host_is_pinned((void *)x.first) && pc_host_is_pinned((void *)(x.second - 1))) continue;
		if (hipHostRegister((void *)x.first, (size_t)(x.second - x.first), flags) != hipSuccess) {
			(void)hipGetLastError();
			while (pinned.size() > before) { (void)hipHostUnregister(pinned.back()); pinned.pop_back(); }
			return false;
		}
		pinned.push_back((void *)x.first);
	}
	return true;
}

static hipError_t pc_fetch_stream_ensure(pc_hip_ctx *ctx)
{
	if (ctx->fetch_stream) return hipSuccess;
	int least = 0, greatest = 0;
	if (!getenv("POLYCAP_FETCH_PRIORITY_OFF") && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least
	    && hipStreamCreateWithPriority(&ctx->fetch_stream, hipStreamNonBlocking, greatest) == hipSuccess)
		return hipSuccess;
	(void)hipGetLastError();
	return hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking);
}

/* see pc_fetch_images.  Returns PC_HIP_OK, an error, or 1 when the direct path cannot be used */
static int pc_fetch_planes_direct(pc_hip_ctx *ctx, int64_t first, int64_t count, void *const *planes, double *weights)
{
	const size_t ne = (size_t)ctx->host.pm.n_energies;
	const long long n_total = ctx->run_slots;
	if (!ctx->run_planes && ((size_t)PC_SOA_TILE*(PC_N_FIELDS + ne)*sizeof(double) > 65536 || pc_soa_ensure(ctx, n_total) != PC_HIP_OK)) return 1;
	PC_HIP_CHECK(pc_fetch_stream_ensure(ctx));
	if (!ctx->ev_sync) PC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming));
	const bool timing = getenv("POLYCAP_TIMING") != nullptr;
	auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_begin = now_ms();
	/* pin the destinations */
	std::vector<void *> pinned;
	auto unpin = [&]() { for (void *p : pinned) (void)hipHostUnregister(p); pinned.clear(); };
	/* Planes at one common stride (polycap_source_get_transmission_efficiencies allocates its result as one slab): pinned in one
	 * piece, and a group of blocks is ONE pitched copy of 17 (one energy: 18, the weights are the 18th plane) rows instead of as
	 * many linear ones -- a copy costs the engine 4-5 us whatever its size (scripts/analysis/copy2d_probe.hip: 56.5 against
	 * 49 GB/s at 0.5 M positions per group). */
	long long slab_stride = 0;      /* bytes between two planes, 0: no common stride */
	int slab_rows = 0;
	const bool pin_here = !ctx->dst_prepinned;
	if (ctx->run_compact && planes[0] && planes[1]) {
		const long long st = (long long)((char *)planes[1] - (char *)planes[0]);
		bool uniform = st >= (long long)((size_t)count*sizeof(double));
		for (int f = 2; f < PC_N_FIELDS && uniform; f++)
			uniform = planes[f] && (char *)planes[f] - (char *)planes[0] == (long long)f*st;
		if (uniform) {
			slab_stride = st;
			slab_rows = PC_N_FIELDS;
			if (ne == 1 && weights && (char *)weights - (char *)planes[0] == (long long)PC_N_FIELDS*st) slab_rows = PC_N_FIELDS + 1;
		}
	}
	if (slab_stride && pin_here) {
		const bool w_in = weights && (char *)weights - (char *)planes[0] == (long long)PC_N_FIELDS*slab_stride;
		const size_t bytes = w_in ? (size_t)PC_N_FIELDS*(size_t)slab_stride + (size_t)count*ne*sizeof(double)
		                          : (size_t)(PC_N_FIELDS - 1)*(size_t)slab_stride + (size_t)count*sizeof(double);
		const hipError_t re = pc_host_is_pinned(planes[0]) ? hipErrorHostMemoryAlreadyRegistered : hipHostRegister(planes[0], bytes, hipHostRegisterDefault);
		if (re == hipSuccess) pinned.push_back(planes[0]);
		else {
			(void)hipGetLastError();
			if (re != hipErrorHostMemoryAlreadyRegistered) { slab_stride = 0; slab_rows = 0; }      /* plane by plane below */
		}
		if (slab_stride && weights && !w_in) {
			const hipError_t rw = pc_host_is_pinned(weights) ? hipErrorHostMemoryAlreadyRegistered : hipHostRegister(weights, (size_t)count*ne*sizeof(double), hipHostRegisterDefault);
			if (rw == hipSuccess) pinned.push_back(weights); else (void)hipGetLastError();
		}
	}
	if (!slab_stride && pin_here) {
		std::vector<std::pair<char *, size_t>> ranges;
		for (int k = 0; k <= PC_N_FIELDS; k++) {
			void *p = (k < PC_N_FIELDS) ? planes[k] : (void *)weights;
			if (p) ranges.emplace_back((char *)p, (size_t)count*sizeof(double)*(k < PC_N_FIELDS ? 1 : ne));
		}
		if (!pc_pin_ranges(ranges, hipHostRegisterDefault, pinned) && !ctx->run_planes) {
			unpin();
			return 1;         /* a record run: the staging pipeline copies without pinning the destination */
		}
		/* (a plane run whose destination cannot be pinned is copied unpinned: slower, still right) */
	}
	const double t_pinned = now_ms();
	int status = PC_HIP_OK;
	const int parts = ctx->run_compact ? 0 : (ctx->n_parts > 1 ? ctx->n_parts : 1);
	if (ctx->run_compact) {
		/* the kernel publishes its planes block by block (pc_blocks_written): every block is copied as soon as its flag is up,
		 * while the kernel goes on.  The kernel's end also ends the wait (every block is complete then). */
		const long long B = 1ll << ctx->run_blk_shift;
		const long long b_end = std::min<long long>(ctx->run_blocks, (first + count + B - 1) >> ctx->run_blk_shift);
		bool kernel_done = false;
		long long b = first >> ctx->run_blk_shift;
		int group = 0;
		/* POLYCAP_FETCH_STREAMS (1 or 2, default 2): copy streams the planes of a group alternate between; POLYCAP_FETCH_DEPTH
		 * (1..3, default 2): groups of copies in flight */
		int n_streams = 2, depth = 2;
		if (const char *ev = getenv("POLYCAP_FETCH_STREAMS")) n_streams = (*ev == '1') ? 1 : 2;
		if (const char *ev = getenv("POLYCAP_FETCH_DEPTH")) depth = (*ev >= '1' && *ev <= '3') ? *ev - '0' : 2;
		if (n_streams == 2 && !ctx->fetch_stream_b) {
			int least = 0, greatest = 0;
			if (!(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least
			      && hipStreamCreateWithPriority(&ctx->fetch_stream_b, hipStreamNonBlocking, greatest) == hipSuccess)) {
				(void)hipGetLastError();
				PC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->fetch_stream_b, hipStreamNonBlocking));
			}
		}
		for (int k = 0; k < 2; k++)
			for (int j = 0; j < 4; j++)
				if (!ctx->ev_group[k][j]) PC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_group[k][j], hipEventDisableTiming));
		hipStream_t streams[2] = { ctx->fetch_stream, n_streams == 2 ? ctx->fetch_stream_b : ctx->fetch_stream };
		while (b < b_end && status == PC_HIP_OK) {
			volatile unsigned int *flag = ctx->h_blk_flag;
			unsigned long spins = 0;
			while (!kernel_done && flag[b] == 0u) {
				if ((++spins & 63ul) == 0ul) {
					const hipError_t q = hipEventQuery(ctx->ev1);
					if (q == hipSuccess) kernel_done = true;
					else if (q != hipErrorNotReady) { status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_transmission_images (kernel event query): ") + hipGetErrorString(q)); break; }
					(void)hipGetLastError();
				}
				std::this_thread::yield();
			}
			if (status != PC_HIP_OK) break;
			/* `depth` groups of copies are kept in flight: the next one is put together when the oldest has finished, from every
			 * block that is complete by then.  The first block is ready a fraction of a millisecond into the run; as the kernel
			 * produces faster than PCIe carries, every group is larger than the one before (up to 64 blocks) and the copy engines
			 * never wait -- nor are they fed thousands of small copies (4-5 us each, whatever their size). */
			if (group >= depth) {
				for (int k = 0; k < n_streams && status == PC_HIP_OK; k++) {
					const hipError_t we = hipEventSynchronize(ctx->ev_group[k][(group - depth) & 3]);
					if (we != hipSuccess) status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_transmission_images (wait for a group of copies): ") + hipGetErrorString(we));
				}
				if (status != PC_HIP_OK) break;
			}
			long long e = b + 1;
			while (e < b_end && e - b < 64 && (kernel_done || flag[e] != 0u)) e++;
			std::atomic_thread_fence(std::memory_order_acquire);
			const long long lo = std::max<long long>(b*B, first), hi = std::min<long long>(std::min<long long>(e*B, n_total), first + count);
			hipError_t err = hipSuccess;
			if (slab_stride) {
				err = hipMemcpy2DAsync((double *)planes[0] + (lo - first), (size_t)slab_stride, ctx->d_soa + lo, (size_t)n_total*sizeof(double),
				                       (size_t)(hi - lo)*sizeof(double), (size_t)slab_rows, hipMemcpyDeviceToHost, streams[0]);
				if (err == hipSuccess && weights && slab_rows == PC_N_FIELDS)
					err = hipMemcpyAsync(weights + (size_t)(lo - first)*ne, ctx->d_soa + (size_t)PC_N_FIELDS*n_total + (size_t)lo*ne,
					                     (size_t)(hi - lo)*ne*sizeof(double), hipMemcpyDeviceToHost, streams[1]);
			}
			for (int f = 0; f <= PC_N_FIELDS && err == hipSuccess && !slab_stride; f++) {
				if (f < PC_N_FIELDS) {
					if (!planes[f]) continue;
					err = hipMemcpyAsync((double *)planes[f] + (lo - first), ctx->d_soa + (size_t)f*n_total + lo, (size_t)(hi - lo)*sizeof(double),
					                     hipMemcpyDeviceToHost, streams[f & 1]);
				} else if (weights) {
					err = hipMemcpyAsync(weights + (size_t)(lo - first)*ne, ctx->d_soa + (size_t)PC_N_FIELDS*n_total + (size_t)lo*ne,
					                     (size_t)(hi - lo)*ne*sizeof(double), hipMemcpyDeviceToHost, streams[f & 1]);
				}
			}
			if (err != hipSuccess) status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_transmission_images (copy of a group of blocks): ") + hipGetErrorString(err));
			for (int k = 0; k < n_streams && err == hipSuccess; k++) {
				err = hipEventRecord(ctx->ev_group[k][group & 3], streams[k]);
				if (err != hipSuccess) status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_transmission_images (event after a group of copies): ") + hipGetErrorString(err));
			}
			b = e;
			group++;
		}
		if (n_streams == 2 && hipStreamSynchronize(ctx->fetch_stream_b) != hipSuccess && status == PC_HIP_OK)
			status = pc_fail(PC_HIP_ERR_RUNTIME, "pc_hip_transmission_images: the plane copies failed");
	}
	for (int k = 0; k < parts && status == PC_HIP_OK; k++) {
		const long long plo = (parts > 1 && k > 0) ? ctx->part_end[k - 1] : 0, phi = (parts > 1) ? ctx->part_end[k] : n_total;
		const long long lo = std::max<long long>(plo, first), hi = std::min<long long>(phi, first + count);
		if (hi <= lo) continue;
		hipError_t e = hipSuccess;
		if (parts > 1) {
			e = hipStreamWaitEvent(ctx->fetch_stream, ctx->ev_part[k], 0);         /* traced, and turned into planes if eager */
		} else {
			int st = pc_hip_transmission_wait(ctx, nullptr);
			if (st) { status = st; break; }
		}
		if (e == hipSuccess && !ctx->run_planes) {
			/* the run kept records: turn the part into planes now, behind its trace */
			int st = pc_soa_launch(ctx, ctx->fetch_stream, plo, phi - plo, n_total);
			if (st) { status = st; break; }
		}
		for (int f = 0; f <= PC_N_FIELDS && e == hipSuccess; f++) {
			if (f < PC_N_FIELDS) {
				if (!planes[f]) continue;
				e = hipMemcpyAsync((double *)planes[f] + (lo - first), ctx->d_soa + (size_t)f*n_total + lo, (size_t)(hi - lo)*sizeof(double),
				                   hipMemcpyDeviceToHost, ctx->fetch_stream);
			} else if (weights) {
				e = hipMemcpyAsync(weights + (size_t)(lo - first)*ne, ctx->d_soa + (size_t)PC_N_FIELDS*n_total + (size_t)lo*ne,
				                   (size_t)(hi - lo)*ne*sizeof(double), hipMemcpyDeviceToHost, ctx->fetch_stream);
			}
		}
		if (e != hipSuccess) status = pc_fail(PC_HIP_ERR_RUNTIME, std::string("pc_hip_transmission_images: ") + hipGetErrorString(e));
	}
	const double t_queued = now_ms();
	if (hipStreamSynchronize(ctx->fetch_stream) != hipSuccess && status == PC_HIP_OK)
		status = pc_fail(PC_HIP_ERR_RUNTIME, "pc_hip_transmission_images: the plane copies failed");
	const double t_copied = now_ms();
	if (ctx->keep_pinned && slab_stride) pinned.clear();      /* the caller keeps its slab pinned (and unpins it itself: pc_hip_host_unregister) */
	unpin();
	if (timing)
		fprintf(stderr, "polycap timing [ms]: plane fetch: pin %.1f, enqueue %.1f, wait for trace + copies %.1f, unpin %.1f\n",
		        t_pinned - t_begin, t_queued - t_pinned, t_copied - t_queued, now_ms() - t_copied);
	return status;
}

static int pc_fetch_images(pc_hip_ctx *ctx, int64_t first, int64_t count, const pc_hip_images *dst, double *raw)
{
	if (!ctx->img_valid) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_images: the last run kept no images");
	if (first < 0 || count < 0 || first + count > ctx->run_slots) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_images: slot range out of bounds");
	/* a leak run is complete (and possibly repeated) only after wait(); a plain run is fetched part by part below */
	if (ctx->leak_pending || (ctx->n_parts <= 1 && !(ctx->run_compact && dst && !raw))) {
		int st = pc_hip_transmission_wait(ctx, nullptr);
		if (st) return st;
	}
	if (count == 0) return PC_HIP_OK;
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	const size_t ne = (size_t)ctx->host.pm.n_energies, rec = (size_t)PC_N_PLANES + ne;
	void *planes[PC_N_PLANES] = {nullptr};
	if (dst) {
		void *p[PC_N_PLANES] = {
			dst->src_start_coords[0], dst->src_start_coords[1], dst->pc_start_coords[0], dst->pc_start_coords[1],
			dst->pc_start_dir[0], dst->pc_start_dir[1], dst->pc_start_elecv[0], dst->pc_start_elecv[1],
			dst->pc_exit_coords[0], dst->pc_exit_coords[1], dst->pc_exit_coords[2],
			dst->pc_exit_dir[0], dst->pc_exit_dir[1], dst->pc_exit_elecv[0], dst->pc_exit_elecv[1],
			dst->pc_exit_nrefl, dst->pc_exit_dtravel };
		memcpy(planes, p, sizeof(p));
	}
	/* Fast path for plane destinations: the caller's planes are pinned for the duration of the call (hipHostRegister: 3 ms
	 * for 1.4 GB of faulted-in memory) and the copy engine writes them straight from the device's plane copy of the records
	 * (pc_soa_kernel), part by part behind the trace.  No host thread touches the data.  Anything that does not fit (a plane
	 * that cannot be pinned, a record too long for the LDS tile) takes the staging pipeline below. */
	if (ctx->run_planes && (!dst || raw))
		return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_records: the last run stored planes (option plane_images); fetch them with pc_hip_transmission_images");
	if (dst && !raw && (ctx->run_planes || (size_t)count*sizeof(double) >= ((size_t)1 << 20))) {
		int st = pc_fetch_planes_direct(ctx, first, count, planes, dst->exit_coord_weights);
		if (st != 1) return st;          /* 1 = not applicable (record runs only) */
		if (ctx->run_planes) return pc_fail(PC_HIP_ERR_RUNTIME, "pc_hip_transmission_images: internal: plane run without a plane fetch");
	}
	/* Pipeline over chunks of <= 16 MB of records: one asynchronous copy (DMA engine, no compute units) of the chunk into
	 * pinned host memory, then host threads turn the records of the previous chunk into the caller's SoA planes while the
	 * next one is in flight.  The copies run on their own stream and wait only for the part of the run that holds the
	 * chunk, so they overlap the kernel of the following parts. */
	size_t chunk = ((size_t)16 << 20) / (rec*sizeof(double));
	if (chunk < 256) chunk = 256;
	if (chunk > (size_t)count) chunk = (size_t)count;
	if (ctx->h_stage_elems < 2*chunk*rec) {
		if (ctx->h_stage) { if (ctx->h_stage_pinned) PC_HIP_CHECK(hipHostFree(ctx->h_stage)); else free(ctx->h_stage); }
		ctx->h_stage = nullptr; ctx->h_stage_elems = 0;
		ctx->h_stage_pinned = true;
		if (hipHostMalloc(&ctx->h_stage, 2*chunk*rec*sizeof(double), hipHostMallocDefault) != hipSuccess) {
			/* no pinned memory to be had: the copies then go through the runtime's own staging, slower but correct */
			(void)hipGetLastError();
			ctx->h_stage_pinned = false;
			ctx->h_stage = (double *)malloc(2*chunk*rec*sizeof(double));
			if (!ctx->h_stage)
				return pc_fail(PC_HIP_ERR_MEMORY, "pc_hip_transmission_images: could not allocate the staging buffer");
		}
		ctx->h_stage_elems = 2*chunk*rec;
	}
	for (int k = 0; k < 2; k++)
		if (!ctx->ev_fetch[k]) PC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fetch[k], hipEventDisableTiming));
	PC_HIP_CHECK(pc_fetch_stream_ensure(ctx));
	int nthreads = ctx->fetch_threads;
	if (nthreads <= 0) {
		const unsigned hw = std::thread::hardware_concurrency();
		nthreads = (int)(hw == 0 ? 4 : (hw > 16 ? 16 : hw));
	}
	if ((size_t)count*rec*sizeof(double) < ((size_t)4 << 20)) nthreads = 1;
	pc_copy_workers workers(nthreads);
	std::vector<pc_copy_piece> pieces;
	/* records [done, done + n) in the pinned buffer `src` -> planes: pieces of 4096 records for the worker threads */
	auto scatter = [&](const double *src, size_t done, size_t n) {
		pieces.clear();
		for (size_t o = 0; o < n; o += 4096)
			pieces.push_back({src + o*rec, done + o, n - o < 4096 ? n - o : 4096});
		workers.run(pieces, planes, dst ? dst->exit_coord_weights : nullptr, rec, ne, raw);
	};
	size_t prev_done = 0, prev_n = 0;
	int c = 0, part = 0, waited = -1;
	for (size_t done = 0; done < (size_t)count; done += chunk, c++) {
		const size_t n = ((size_t)count - done < chunk) ? (size_t)count - done : chunk;
		const int b = c & 1;
		double *h_buf = ctx->h_stage + (size_t)b*chunk*rec;
		if (ctx->n_parts > 1) {
			/* the last slot of the chunk decides which part has to be finished */
			const long long last = first + (long long)(done + n) - 1;
			while (part < ctx->n_parts - 1 && ctx->part_end[part] <= last) part++;
			/* every part up to that one: consecutive parts run on two streams, so the event of part p says nothing
			 * about part p-1, whose tail a chunk that straddles the boundary also reads */
			for (int q = waited + 1; q <= part; q++)
				PC_HIP_CHECK(hipStreamWaitEvent(ctx->fetch_stream, ctx->ev_part[q], 0));
			if (part > waited) waited = part;
		}
		PC_HIP_CHECK(hipMemcpyAsync(h_buf, ctx->d_img + ((size_t)first + done)*rec, n*rec*sizeof(double), hipMemcpyDeviceToHost,
		                            ctx->n_parts > 1 ? ctx->fetch_stream : ctx->stream));
		PC_HIP_CHECK(hipEventRecord(ctx->ev_fetch[b], ctx->n_parts > 1 ? ctx->fetch_stream : ctx->stream));
		if (c > 0) {
			PC_HIP_CHECK(hipEventSynchronize(ctx->ev_fetch[b ^ 1]));
			scatter(ctx->h_stage + (size_t)(b ^ 1)*chunk*rec, prev_done, prev_n);
		}
		prev_done = done; prev_n = n;
	}
	PC_HIP_CHECK(hipEventSynchronize(ctx->ev_fetch[(c - 1) & 1]));
	scatter(ctx->h_stage + (size_t)((c - 1) & 1)*chunk*rec, prev_done, prev_n);
	return PC_HIP_OK;
}

int pc_hip_transmission_images(pc_hip_ctx *ctx, int64_t first, int64_t count, const pc_hip_images *dst)
{
	if (!ctx || !dst) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_images: NULL argument");
	return pc_fetch_images(ctx, first, count, dst, nullptr);
}

int pc_hip_transmission_slot_ids(pc_hip_ctx *ctx, int64_t first, int64_t count, int64_t *slots)
{
	if (!ctx || (count > 0 && !slots)) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_slot_ids: NULL argument");
	if (first < 0 || count < 0 || first + count > ctx->run_slots) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_slot_ids: range out of bounds");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	if (!ctx->img_valid || !ctx->run_compact || !ctx->d_ids || !ctx->slot_ids) {
		/* a run that stores every photon at its slot: the identity */
		for (int64_t k = 0; k < count; k++) slots[k] = first + k;
		return PC_HIP_OK;
	}
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	if (count) PC_HIP_CHECK(hipMemcpy(slots, ctx->d_ids + first, (size_t)count*sizeof(long long), hipMemcpyDeviceToHost));
	return PC_HIP_OK;
}

int pc_hip_leak_set_order(pc_hip_ctx *ctx, const uint32_t *order, int64_t n, int64_t n_heavy)
{
	if (!ctx || n < 0 || (n > 0 && !order) || n_heavy < 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_set_order: invalid argument");
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	if (ctx->d_leak_order) { PC_HIP_CHECK(hipFree(ctx->d_leak_order)); ctx->d_leak_order = nullptr; }
	ctx->leak_order_n = 0; ctx->leak_n_heavy = 0; ctx->leak_order_user = 0; ctx->leak_order_slot0 = -1;
	if (n == 0) return PC_HIP_OK;
	{
		/* a permutation of 0 .. n-1, or slots would be traced twice or not at all */
		std::vector<unsigned char> seen((size_t)n, 0);
		for (int64_t k = 0; k < n; k++) {
			if ((int64_t)order[k] >= n || seen[order[k]]) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_set_order: order is not a permutation of the slots");
			seen[order[k]] = 1;
		}
	}
	PC_HIP_CHECK(hipMalloc(&ctx->d_leak_order, (size_t)n*sizeof(unsigned int)));
	ctx->leak_order_cap = n;
	PC_HIP_CHECK(hipMemcpy(ctx->d_leak_order, order, (size_t)n*sizeof(unsigned int), hipMemcpyHostToDevice));
	ctx->leak_order_n = n; ctx->leak_n_heavy = n_heavy; ctx->leak_order_user = 1;
	return PC_HIP_OK;
}

int pc_hip_leak_slot_units(pc_hip_ctx *ctx, int64_t first, int64_t count, uint32_t *units)
{
	if (!ctx || (count > 0 && !units) || first < 0 || count < 0) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_slot_units: invalid argument");
	int st = pc_hip_transmission_wait(ctx, nullptr);
	if (st) return st;
	if (!ctx->d_leak_slot_units || first + count > ctx->leak_slot_units_n) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_leak_slot_units: no leak run with the option leak_slot_units covers this range");
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	if (count) PC_HIP_CHECK(hipMemcpy(units, ctx->d_leak_slot_units + first, (size_t)count*sizeof(unsigned int), hipMemcpyDeviceToHost));
	return PC_HIP_OK;
}

int pc_hip_transmission_records(pc_hip_ctx *ctx, int64_t first, int64_t count, double *records)
{
	if (!ctx || !records) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_transmission_records: NULL argument");
	return pc_fetch_images(ctx, first, count, nullptr, records);
}

/* src/polycap-source.c:1066-1076 */
void pc_hip_efficiencies(size_t n_energies, const double *sum_weights, const int64_t counters[6], double *efficiencies)
{
	int64_t sum_iexit = counters[0], sum_not_entered = counters[1], sum_not_transmitted = counters[2];
	double open_area = (double)(sum_iexit+sum_not_transmitted)/(sum_iexit+sum_not_entered+sum_not_transmitted);
	for (size_t i = 0; i < n_energies; i++)
		efficiencies[i] = (sum_weights[i] / ((double)sum_iexit+(double)sum_not_transmitted)) * open_area;
}

void pc_hip_host_unregister(void *ptr)
{
	if (ptr && hipHostUnregister(ptr) != hipSuccess) (void)hipGetLastError();
}

int pc_hip_device_synchronize(pc_hip_ctx *ctx)
{
	if (!ctx) return pc_fail(PC_HIP_ERR_INVALID, "pc_hip_device_synchronize: ctx must not be NULL");
	PC_HIP_CHECK(hipSetDevice(ctx->device));
	PC_HIP_CHECK(hipDeviceSynchronize());
	return PC_HIP_OK;
}

} /* extern "C" */

#include "pc_group.h"

/* Heaviest slots first.  A leak launch ends with its longest slot: 20 000 units of work on one lane, which advances several
 * times faster alone in its wave than among 63 others (a wave runs one class of work at a time).  Which slots are long is known
 * beforehand to a good approximation (correlation 0.95 with the units of the leak run, scripts/analysis/leak_units.py): a plain
 * run of the same slots -- the same photons without their leaks, a hundredth of the leak run's time -- counts the reflections
 * of every attempt per slot.  The slots are then handed out in descending order of that count, the first n/400 of them to lane
 * 0 of the waves, whose other lanes wait while such a slot is at work (pc_leak_kargs::order).  The results do not depend on the
 * order (photon streams are keyed by slot and attempt, events are ordered by slot on the host). */
static int pc_leak_auto_order(pc_hip_ctx *ctx)
{
	const long long n = ctx->leak_n_slots;
	if (ctx->leak_order_user) return PC_HIP_OK;                       /* the caller's order (used if it is for n slots) */
	long long lanes = 0;
	(void)pc_leak_grid(ctx, n, lanes);
	/* considered when every lane gets two to five slots: with fewer there is nothing to order, with more the launch is not bound
	 * by its longest slot (the check below, made beforehand with the reference optic's ratio of longest to mean slot, 11.6) */
	if (!ctx->leak_order || n < 2*lanes || n >= 5*lanes || n >= (1ll << 32)) {
		if (ctx->d_leak_order) { PC_HIP_CHECK(hipFree(ctx->d_leak_order)); ctx->d_leak_order = nullptr; }
		ctx->leak_order_n = 0; ctx->leak_order_slot0 = -1; ctx->leak_order_cap = 0;
		return PC_HIP_OK;
	}
	if (ctx->d_leak_order && ctx->leak_order_n == n && ctx->leak_order_seed == ctx->leak_seed && ctx->leak_order_slot0 == ctx->leak_slot0
	    && ctx->leak_order_attempts == ctx->leak_max_attempts)
		return PC_HIP_OK;                                               /* the same slots as last time */
	if (ctx->work_est_n < n) {
		if (ctx->d_work_est) PC_HIP_CHECK(hipFree(ctx->d_work_est));
		ctx->d_work_est = nullptr; ctx->work_est_n = 0;
		if (hipMalloc(&ctx->d_work_est, (size_t)n*sizeof(unsigned int)) != hipSuccess) return pc_fail(PC_HIP_ERR_MEMORY, "leak run: could not allocate the work estimate");
		ctx->work_est_n = n;
	}
	const bool tim = getenv("POLYCAP_LEAK_TIMING") != nullptr;
	const auto t_0 = std::chrono::steady_clock::now();
	/* the time of the launch starts here */
	PC_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
	ctx->leak_ev0_done = 1;
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_work_est, 0, (size_t)n*sizeof(unsigned int), ctx->stream));
	PC_HIP_CHECK(hipMemsetAsync(ctx->d_totals, 0, ctx->totals_bytes, ctx->stream));
	pc_kargs a;
	pc_fill_common(ctx, a);
	a.seed = ctx->leak_seed; a.max_attempts = ctx->leak_max_attempts; a.keep_images = 0;
	a.slot0 = ctx->leak_slot0; a.n_slots = n;
	pc_set_img(ctx, a, 0, n, false, false);
	a.work_est = ctx->d_work_est;
	struct restore_ctx {          /* the lane kernel, no events of its own */
		pc_hip_ctx *c; int producer, pool;
		~restore_ctx() { c->producer = producer; c->pool = pool; c->rec_ev0 = c->rec_ev1 = true; }
	} restore{ctx, ctx->producer, ctx->pool};
	ctx->producer = 0; ctx->pool = 0; ctx->rec_ev0 = ctx->rec_ev1 = false;
	int st = ctx->host.pm.generic_src ? pc_launch_kernel<PC_MODE_SRC_GENERIC>(ctx, a, n) : pc_launch_kernel<PC_MODE_SRC_CIRCULAR>(ctx, a, n);
	if (st) return st;
	std::vector<unsigned int> est((size_t)n);
	PC_HIP_CHECK(hipMemcpyAsync(est.data(), ctx->d_work_est, (size_t)n*sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
	PC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	const auto t_1 = std::chrono::steady_clock::now();
	/* Is the launch bound by its longest slot?  A lone lane works through a unit in about 4 us, a lane among the 64 of a busy wave
	 * in about 13 us: with the slots in slot order the launch lasts about (all work / lanes) x 13 us, the longest slot alone
	 * (its work) x 4 us.  When the second is not most of the first (launches of many slots per lane), handing the heaviest slots
	 * to lanes of their own only takes lanes away from the rest: slot order stays (measured: 524288 and 1048576 slots lose 4-7 %,
	 * 262144 gain 10-15 %). */
	unsigned int top = 0;
	unsigned long long total = 0;
	for (long long k = 0; k < n; k++) { if (est[(size_t)k] > top) top = est[(size_t)k]; total += est[(size_t)k]; }
	if (!((double)top * 4.0 * (double)lanes > 0.8 * 13.0 * (double)total)) {
		ctx->leak_order_n = 0; ctx->leak_order_slot0 = -1;      /* the buffer stays for the next run; it is not used */
		if (tim) fprintf(stderr, "leak order: slot order kept (longest slot %u of %llu predicted units, %lld lanes)\n", top, total, lanes);
		return PC_HIP_OK;
	}
	/* descending counting sort (stable: equal counts keep slot order) */
	std::vector<unsigned int> order((size_t)n);
	if (top < (1u << 22)) {
		std::vector<unsigned int> first((size_t)top + 2, 0);
		for (long long k = 0; k < n; k++) first[(size_t)(top - est[(size_t)k]) + 1]++;
		for (size_t v = 0; v + 1 < first.size(); v++) first[v + 1] += first[v];
		for (long long k = 0; k < n; k++) order[first[(size_t)(top - est[(size_t)k])]++] = (unsigned int)k;
	} else {
		for (long long k = 0; k < n; k++) order[(size_t)k] = (unsigned int)k;
		std::stable_sort(order.begin(), order.end(), [&](unsigned int x, unsigned int y) { return est[x] > est[y]; });
	}
	const auto t_2 = std::chrono::steady_clock::now();
	if (ctx->d_leak_order && ctx->leak_order_cap < n) { PC_HIP_CHECK(hipFree(ctx->d_leak_order)); ctx->d_leak_order = nullptr; }
	if (!ctx->d_leak_order) { PC_HIP_CHECK(hipMalloc(&ctx->d_leak_order, (size_t)n*sizeof(unsigned int))); ctx->leak_order_cap = n; }
	PC_HIP_CHECK(hipMemcpyAsync(ctx->d_leak_order, order.data(), (size_t)n*sizeof(unsigned int), hipMemcpyHostToDevice, ctx->stream));
	PC_HIP_CHECK(hipStreamSynchronize(ctx->stream));      /* `order` leaves scope */
	if (tim) {
		const auto t_3 = std::chrono::steady_clock::now();
		auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
		fprintf(stderr, "leak order: plain pre-pass + copy %.2f ms, sort %.2f ms, upload %.2f ms\n", ms(t_0, t_1), ms(t_1, t_2), ms(t_2, t_3));
	}
	ctx->leak_order_n = n;
	/* one heavy slot per wave at most: the wave is the heavy lane's alone while it lasts */
	ctx->leak_n_heavy = std::min<long long>(n / 400, (long long)ctx->n_cu * 2);
	ctx->leak_order_seed = ctx->leak_seed; ctx->leak_order_slot0 = ctx->leak_slot0; ctx->leak_order_attempts = ctx->leak_max_attempts;
	return PC_HIP_OK;
}

static int pc_transmission_enqueue_leak(pc_hip_ctx *ctx)
{
	pc_kargs a;
	pc_fill_common(ctx, a);
	pc_set_img(ctx, a, 0, ctx->leak_n_slots, ctx->leak_keep_images != 0, false);
	a.seed = ctx->leak_seed; a.slot0 = ctx->leak_slot0; a.n_slots = ctx->leak_n_slots;
	a.max_attempts = ctx->leak_max_attempts; a.keep_images = ctx->leak_keep_images;
	return ctx->host.pm.generic_src ? pc_leak_enqueue<PC_MODE_SRC_GENERIC>(ctx, a, ctx->leak_n_slots, ctx->leak_capacity_used)
	                                : pc_leak_enqueue<PC_MODE_SRC_CIRCULAR>(ctx, a, ctx->leak_n_slots, ctx->leak_capacity_used);
}
