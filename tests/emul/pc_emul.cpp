/*
 * pc_emul.cpp -- TEST-ONLY host compile of the device header (polycap_amd/csrc/hip/pc_device.h).
 *
 * Drives the per-photon state machine (NEW -> MARCH <-> EVENT -> DONE) one photon at a time on the CPU so
 * that the `-m "not gpu"` suite can compare the device logic (certified skipping, hoisted Fresnel terms)
 * with the oracle on identical photons.  It is built into tests/emul/libpc_emul.so by the tests and is
 * never linked into, loaded by or shipped with libpolycap: the product has no CPU trace path.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "pc_problem.h"
#include "pc_leak.h"

namespace {

struct Emul {
	pc_host_tables t;
	pc_tables T;
};

int setup(const pc_hip_problem *p, int literal, Emul &E)
{
	std::string err;
	int rc = pc_build_tables(p, E.t, err);
	if (rc) return rc;
	E.t.pm.literal = literal;
	E.T.z = E.t.z.data(); E.T.cap = E.t.cap.data(); E.T.zh = E.t.zh.data();
	E.T.cap2 = E.t.cap2.data(); E.T.hexd = E.t.hexd.data(); E.T.idz = E.t.idz.data(); E.T.ext = E.t.ext.data();
	E.T.mg = E.t.mg.data();
	E.T.stp = E.t.stp.data(); E.T.istp = E.t.istp.data(); E.T.dr = E.t.dr.data();
	return 0;
}

/* FASTF: the weights as the kernels of source runs form them (pc_fresnel3s; explicit launches keep FORMs 0/1) */
template <int NE, bool FASTF = false>
int run_photon(const Emul &E, pc_photon<NE> &ph, double x, double y, double z, double dx, double dy, double dz,
               double ex, double ey, double ez, int64_t *stats)
{
	int st = pc_launch_init(E.T, E.t.pm, ph, x, y, z, dx, dy, dz, ex, ey, ez);
	while (st != PC_ST_DONE) {
		if (st == PC_ST_MARCH) {
			st = pc_march_step(E.T, E.t.pm, ph);
			if (stats && st == PC_ST_MARCH) stats[0]++;
		} else {
			st = pc_event<NE, FASTF>(E.T, E.t.pm, E.t.ec.data(), ph);
			if (stats) stats[1]++;
		}
	}
	return ph.rc;
}

} // namespace

extern "C" {

int emul_launch_batch(const pc_hip_problem *p, int literal, int use_regs, int64_t n,
                      const double *start, const double *dir, const double *elecv,
                      int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                      int64_t *i_refl, double *d_travel, int64_t *stats)
{
	Emul E;
	int r = setup(p, literal, E);
	if (r) return r;
	const size_t ne = p->n_energies;
	if (stats) stats[0] = stats[1] = 0;
	for (int64_t j = 0; j < n; j++) {
		const double *s = start + 3*j, *d = dir + 3*j, *e = elecv + 3*j;
		double P[3], D[3], Ev[3], dt; int ir, code;
		if (use_regs && ne == 1) {
			pc_photon<1> ph; ph.wmem = nullptr; ph.wstride = 0;
			code = run_photon(E, ph, s[0], s[1], s[2], d[0], d[1], d[2], e[0], e[1], e[2], stats);
			weights[j] = ph.w[0];
			P[0]=ph.Px; P[1]=ph.Py; P[2]=ph.Pz; D[0]=ph.dx; D[1]=ph.dy; D[2]=ph.dz; Ev[0]=ph.ex; Ev[1]=ph.ey; Ev[2]=ph.ez;
			ir = ph.irefl; dt = ph.dtravel;
		} else {
			pc_photon<0> ph; ph.wmem = weights + (size_t)j*ne; ph.wstride = 1;
			code = run_photon(E, ph, s[0], s[1], s[2], d[0], d[1], d[2], e[0], e[1], e[2], stats);
			if (!ph.wset)
				for (size_t k = 0; k < ne; k++) ph.wmem[k] = 1.0;   /* never reflected (or rejected at the entrance) */
			P[0]=ph.Px; P[1]=ph.Py; P[2]=ph.Pz; D[0]=ph.dx; D[1]=ph.dy; D[2]=ph.dz; Ev[0]=ph.ex; Ev[1]=ph.ey; Ev[2]=ph.ez;
			ir = ph.irefl; dt = ph.dtravel;
		}
		rc[j] = code;
		memcpy(exit_coords + 3*j, P, sizeof P);
		memcpy(exit_dir + 3*j, D, sizeof D);
		memcpy(exit_elecv + 3*j, Ev, sizeof Ev);
		i_refl[j] = ir;
		d_travel[j] = dt;
	}
	return 0;
}

/* polycap_photon_launch with leak_calc=true for explicit photons; records: capacity x (14 + n_energies) doubles */
int emul_launch_leak(const pc_hip_problem *p, int literal, int64_t n,
                     const double *start, const double *dir, const double *elecv, int max_depth, int64_t capacity,
                     int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                     int64_t *i_refl, double *d_travel, double *records, int64_t *n_records, int32_t *stack_overflow)
{
	Emul E;
	int r = setup(p, literal, E);
	if (r) return r;
	const int ne = (int)p->n_energies;
	std::vector<double> frames((size_t)max_depth * (PC_LF_HDR + ne));
	unsigned long long cursor = 0;
	pc_leak_lane L;
	pc_leak_ctx &cx = L.cx;
	pc_photon<0> &ph = L.ph;
	cx.ec = E.t.ec.data(); cx.amu = E.t.amu.data(); cx.ne = ne;
	cx.frames = frames.data(); cx.max_depth = max_depth;
	cx.sink.records = records; cx.sink.cursor = &cursor; cx.sink.capacity = capacity;
	cx.stack_overflow = 0;
	for (int64_t j = 0; j < n; j++) {
		const double *s = start + 3*j, *d = dir + 3*j, *e = elecv + 3*j;
		ph.wmem = nullptr; ph.wstride = 1;
		int st = pc_launch_init(E.T, E.t.pm, ph, s[0], s[1], s[2], d[0], d[1], d[2], e[0], e[1], e[2]);
		cx.slot = (double)j; cx.attempt = 0.;
		rc[j] = pc_leak_launch(E.T, E.t.pm, L, st, s[2]);
		for (int k = 0; k < ne; k++) weights[(size_t)j*ne + k] = frames[PC_LF_HDR + k];
		exit_coords[3*j] = ph.Px; exit_coords[3*j+1] = ph.Py; exit_coords[3*j+2] = ph.Pz;
		exit_dir[3*j] = ph.dx; exit_dir[3*j+1] = ph.dy; exit_dir[3*j+2] = ph.dz;
		exit_elecv[3*j] = ph.ex; exit_elecv[3*j+1] = ph.ey; exit_elecv[3*j+2] = ph.ez;
		i_refl[j] = ph.irefl;
		d_travel[j] = ph.dtravel;
	}
	*n_records = (int64_t)cursor;
	*stack_overflow = cx.stack_overflow;
	return 0;
}

/* the driver loop of src/polycap-source.c:744-884 with leak_calc=true for slots [slot0, slot0 + n_slots):
 * counters = {iexit, not_entered, not_transmitted, sum_irefl}; exit_weights [n_slots x n_energies] */
int emul_transmission_leak(const pc_hip_problem *p, uint64_t seed, int64_t slot0, int64_t n_slots, uint32_t max_attempts,
                           int max_depth, int64_t capacity, double *sum_weights, int64_t *counters, double *exit_weights,
                           double *records, int64_t *n_records, int32_t *stack_overflow)
{
	Emul E;
	int r = setup(p, 0, E);
	if (r) return r;
	const int ne = (int)p->n_energies;
	std::vector<double> frames((size_t)max_depth * (PC_LF_HDR + ne));
	unsigned long long cursor = 0;
	pc_leak_lane L;
	pc_leak_ctx &cx = L.cx;
	pc_photon<0> &ph = L.ph;
	cx.ec = E.t.ec.data(); cx.amu = E.t.amu.data(); cx.ne = ne;
	cx.frames = frames.data(); cx.max_depth = max_depth;
	cx.sink.records = records; cx.sink.cursor = &cursor; cx.sink.capacity = capacity;
	cx.stack_overflow = 0;
	for (int k = 0; k < ne; k++) sum_weights[k] = 0.;
	for (int k = 0; k < 4; k++) counters[k] = 0;
	for (int64_t j = 0; j < n_slots; j++) {
		for (uint32_t attempt = 0; attempt < max_attempts; attempt++) {
			pc_start s;
			if (E.t.pm.generic_src) pc_sample_photon<true>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
			else pc_sample_photon<false>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
			ph.wmem = nullptr; ph.wstride = 1;
			int st = pc_launch_init(E.T, E.t.pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
			cx.slot = (double)(slot0 + j); cx.attempt = (double)attempt;
			int rc = pc_leak_launch(E.T, E.t.pm, L, st, s.z);
			int ok = 0;
			if (rc == 0) counters[2]++;
			else if (rc == 2) counters[1]++;
			else if (rc == 1) {
				ok = pc_in_exit_window(E.t.pm, ph);
				if (!ok) {
					/* reclassified as not transmitted by the driver (:762-777): its events wait like those of rc 0 */
				}
			}
			if (ok) {
				counters[0]++;
				counters[3] += ph.irefl;
				for (int k = 0; k < ne; k++) {
					sum_weights[k] += frames[PC_LF_HDR + k];
					if (exit_weights) exit_weights[(size_t)j*ne + k] = frames[PC_LF_HDR + k];
				}
				break;
			}
		}
	}
	*n_records = (int64_t)cursor;
	*stack_overflow = cx.stack_overflow;
	return 0;
}

/* the driver loop of src/polycap-source.c:744-884 (leak_calc=false) for slots [slot0, slot0 + n_slots) on the host compile
 * of the device code -- photon for photon what pc_trace_kernel computes (tests/test_gpu_parity.py checks that bit for
 * bit), with the kernel's exact fixed-point sums.  counters = {iexit, not_entered, not_transmitted, sum_irefl};
 * sumw_fixed = (lo, hi) of sum floor(w * 2^62) (single energy); per_slot (optional) = weight, i_refl, attempts used per slot.
 * OpenMP over slots: used by the bias study (scripts/parity_bias.py) next to the oracle. */
int emul_transmission(const pc_hip_problem *p, uint64_t seed, int64_t slot0, int64_t n_slots, uint32_t max_attempts,
                      int n_threads, int64_t *counters, uint64_t *sumw_fixed, double *per_slot)
{
	Emul E;
	int r = setup(p, 0, E);
	if (r) return r;
	if (p->n_energies != 1) return -1;
	int64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
	unsigned __int128 tot = 0;
#ifdef _OPENMP
	if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads) reduction(+: c0, c1, c2, c3)
#endif
	{
		unsigned __int128 mine = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
		for (int64_t j = 0; j < n_slots; j++) {
			for (uint32_t attempt = 0; attempt < max_attempts; attempt++) {
				pc_start s;
				if (E.t.pm.generic_src) pc_sample_photon<true>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
				else pc_sample_photon<false>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
				pc_photon<1> ph; ph.wmem = nullptr; ph.wstride = 0;
				int rc = run_photon<1, true>(E, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez, nullptr);
				int ok = 0;
				if (rc == 0) c2++;
				else if (rc == 2) c1++;
				else if (rc == 1) ok = pc_in_exit_window(E.t.pm, ph);
				if (ok) {
					c0++; c3 += ph.irefl;
					mine += (unsigned __int128)(uint64_t)(ph.w[0] * 4611686018427387904.0);
					if (per_slot) { per_slot[3*j] = ph.w[0]; per_slot[3*j + 1] = (double)ph.irefl; per_slot[3*j + 2] = (double)(attempt + 1); }
					break;
				}
			}
		}
#ifdef _OPENMP
#pragma omp critical
#endif
		tot += mine;
	}
	counters[0] = c0; counters[1] = c1; counters[2] = c2; counters[3] = c3;
	sumw_fixed[0] = (uint64_t)tot; sumw_fixed[1] = (uint64_t)(tot >> 64);
	return 0;
}

/* Analysis aid (scripts/analysis/flight_stats.py): how the march steps of a run distribute over flights.  hist[256]: flights
 * by number of march steps (last bin: >= 255); steps_by[10]: steps that advanced by 1 / PC_L1 / PC_L2 segments, probes that
 * failed at stride PC_L2 / PC_L1 (the stride is lowered), steps that ended in an EVENT without advancing, first-segment steps,
 * flights, EVENT phases (literal visits), literal visits without a hit. */
static int64_t dbg_adv1_lvcap, dbg_adv1_end, dbg_first_flight;
int emul_flight_stats(const pc_hip_problem *p, uint64_t seed, int64_t slot0, int64_t n_slots, int64_t *hist, int64_t *steps_by)
{
	Emul E;
	int r = setup(p, 0, E);
	if (r) return r;
	for (int k = 0; k < 256; k++) hist[k] = 0;
	for (int k = 0; k < 10; k++) steps_by[k] = 0;
	for (int64_t j = 0; j < n_slots; j++) {
		for (uint32_t attempt = 0; attempt < (1u << 20); attempt++) {
			pc_start s;
			if (E.t.pm.generic_src) pc_sample_photon<true>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
			else pc_sample_photon<false>(E.t.pm, seed, (uint64_t)(slot0 + j), attempt, s);
			pc_photon<1> ph; ph.wmem = nullptr; ph.wstride = 0;
			int st = pc_launch_init(E.T, E.t.pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
			int64_t in_flight = 0;
			while (st != PC_ST_DONE) {
				if (st == PC_ST_MARCH) {
					const int i0 = ph.i, first = ph.first, lv0 = ph.lv;
					const double C0b = ph.C0;
					st = pc_march_step(E.T, E.t.pm, ph);
										if (getenv("PC_FS_TRACE") && j < 3) {
						const pc_marg4 g = E.T.mg[i0];
						const float knf = (float)ph.kn;
						fprintf(stderr, "slot %lld refl %d: i %d first %d cap %d C0 %.3e  m1 %.3e m2 %.3e (md1 %.2e r2 %.2e kn %.1f) -> i %d cap %d st %d\n", (long long)j, ph.irefl, i0, first, lv0, C0b,
						        (double)(knf*g.md1*(g.r2 + knf*g.md1) + pc_bits_as_float(g.mb12 & 0xffff0000u)), (double)(knf*g.md2*(g.r2 + knf*g.md2) + pc_bits_as_float(g.mb12 << 16)), (double)g.md1, (double)g.r2, (double)knf, ph.i, ph.lv, st);
					}
					in_flight++;
					if (ph.irefl == 0) dbg_first_flight++;
					if (first) steps_by[6]++;
					else if (st == PC_ST_MARCH || ph.i != i0) {      
						const int adv = ph.i - i0;
						if (adv == 1) { steps_by[0]++; if (getenv("PC_FS_DEBUG") && ph.lv > 0) dbg_adv1_lvcap++; if (getenv("PC_FS_DEBUG") && i0 + PC_L1 > E.t.pm.nmax) dbg_adv1_end++; }
						else if (adv == PC_L1) steps_by[1]++;
						else if (adv == PC_L2) steps_by[2]++;
						else if (adv == 0) steps_by[(lv0 == 2 && ph.lv == 1) ? 3 : 4]++;
					} else if (st == PC_ST_EVENT) steps_by[5]++;
				} else {
					st = pc_event(E.T, E.t.pm, E.t.ec.data(), ph);
					steps_by[8]++;
					if (st == PC_ST_MARCH && !ph.first) steps_by[9]++;      /* literal visit without a hit */
					if (ph.first || st == PC_ST_DONE) {      /* a reflection (a new flight begins) or the end */
						hist[in_flight < 255 ? in_flight : 255]++;
						steps_by[7]++;
						in_flight = 0;
					}
				}
			}
			if (in_flight) { hist[in_flight < 255 ? in_flight : 255]++; steps_by[7]++; }
			if (ph.rc == 1 && pc_in_exit_window(E.t.pm, ph)) break;
		}
	}
	if (getenv("PC_FS_DEBUG")) fprintf(stderr, "adv1 with stride cap > 0: %lld, adv1 near the end of the profile: %lld, steps before the first reflection: %lld\n", (long long)dbg_adv1_lvcap, (long long)dbg_adv1_end, (long long)dbg_first_flight);
	return 0;
}

int emul_sample(const pc_hip_problem *p, uint64_t seed, int64_t n, const int64_t *slots, const uint32_t *attempts, double *out)
{
	Emul E;
	int r = setup(p, 0, E);
	if (r) return r;
	for (int64_t j = 0; j < n; j++) {
		pc_start s;
		if (E.t.pm.generic_src) pc_sample_photon<true>(E.t.pm, seed, (uint64_t)slots[j], attempts[j], s);
		else pc_sample_photon<false>(E.t.pm, seed, (uint64_t)slots[j], attempts[j], s);
		double *o = out + 12*j;
		o[0]=s.x; o[1]=s.y; o[2]=s.z; o[3]=s.dx; o[4]=s.dy; o[5]=s.dz; o[6]=s.ex; o[7]=s.ey; o[8]=s.ez;
		o[9]=s.srcx; o[10]=s.srcy; o[11]=0.;
	}
	return 0;
}

} // extern "C"
