/* Analysis tool (not part of the product or the tests): units of work per exit-photon slot of a leak_calc=true run, by class
 * (march steps, wall steps, capillary probes, bookkeeping), on the host compile of the device headers -- which slots make the
 * ~250 ms floor of the leak kernel, and out of what.  Built and driven by scripts/analysis/leak_units.py. */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#define PC_LEAK_STATS 1
#include "pc_problem.h"
#include "pc_leak.h"

long long pc_leak_stats[16];
long long crit_sum[4], crit_best[2];
extern "C" long long *leak_crit(void) { static long long r[6]; for (int k = 0; k < 4; k++) r[k] = crit_sum[k]; r[4] = crit_best[0]; r[5] = crit_best[1]; return r; }
extern "C" long long *leak_stats(void) { return pc_leak_stats; }

extern "C" int leak_units(const pc_hip_problem *p, uint64_t seed, int64_t slot0, int64_t n_slots, int max_depth,
                          int64_t *per_slot /* [n_slots][8]: march, wall step, probe, other units, attempts, deepest level, units of the longest attempt,
                                                  reflections + 1 summed over the attempts of the PLAIN trace of the same slot */)
{
	pc_host_tables t; std::string err;
	if (pc_build_tables(p, t, err)) return 1;
	pc_tables T;
	T.z = t.z.data(); T.cap = t.cap.data(); T.zh = t.zh.data(); T.cap2 = t.cap2.data(); T.hexd = t.hexd.data(); T.idz = t.idz.data(); T.ext = t.ext.data();
	T.mg = t.mg.data(); T.stp = t.stp.data(); T.istp = t.istp.data(); T.dr = t.dr.data();
	const pc_params &Pm = t.pm;
	const int ne = (int)p->n_energies;
#pragma omp parallel
	{
		std::vector<double> frames((size_t)max_depth * (PC_LF_HDR + ne));
		std::vector<double> records((size_t)4096 * (PC_LR_HDR + ne));
		unsigned long long cursor = 0;
		pc_leak_lane L;
		pc_leak_ctx &cx = L.cx;
		pc_photon<0> &ph = L.ph;
		cx.ec = t.ec.data(); cx.amu = t.amu.data(); cx.ne = ne;
		cx.frames = frames.data(); cx.max_depth = max_depth;
		cx.sink.records = records.data(); cx.sink.cursor = &cursor; cx.sink.capacity = 0;      /* events are counted, not kept */
		cx.stack_overflow = 0;
#pragma omp for schedule(dynamic, 16)
		for (int64_t j = 0; j < n_slots; j++) {
			int64_t *o = per_slot + 10*j;
			memset(o, 0, 10*sizeof(int64_t));
			for (uint32_t attempt = 0; attempt < (1u << 20); attempt++) {
				pc_start s;
				if (Pm.generic_src) pc_sample_photon<true>(Pm, seed, (uint64_t)(slot0 + j), attempt, s);
				else pc_sample_photon<false>(Pm, seed, (uint64_t)(slot0 + j), attempt, s);
				ph.wmem = nullptr; ph.wstride = 1;
				int st = pc_launch_init(T, Pm, ph, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
				cx.slot = (double)(slot0 + j); cx.attempt = (double)attempt;
				pc_leak_begin(T, Pm, L, st, s.z);
				/* critical path of the attempt if every leaked fraction were traced by a lane of its own from the moment it is
				 * spawned: own[l] = units of the photon at level l so far, kid[l] = latest end among its finished children */
				std::vector<long long> own(1, 0), kid(1, 0), at(1, 0);
				long long att_units = 0;
				int prev_lvl = 0;
				while (L.st != PC_LS_DONE) {
					if (L.lvl > prev_lvl) { own.push_back(0); kid.push_back(0); at.push_back(own[prev_lvl]); }
					else if (L.lvl < prev_lvl) {
						while ((int)own.size() - 1 > L.lvl) {
							const long long c = std::max(own.back(), kid.back()), a0 = at.back();
							own.pop_back(); kid.pop_back(); at.pop_back();
							kid.back() = std::max(kid.back(), a0 + c);
						}
					}
					prev_lvl = L.lvl;
					own[L.lvl]++; att_units++;
					if (L.lvl > o[5]) o[5] = L.lvl;
					if (L.st == PC_LS_MARCH) { o[0]++; pc_leak_unit_march(T, Pm, L); }
					else if (L.st == PC_LS_WALL_STEP) { o[1]++; L.st = pc_wall_step(T, Pm, L, L.after_wall); }
					else if (L.st == PC_LS_WALL_PROBE) { o[2]++; L.st = pc_wall_probe(T, Pm, L, L.after_wall); }
					else { o[3]++; pc_leak_unit_other(T, Pm, L); }
				}
				while (own.size() > 1) {
					const long long c = std::max(own.back(), kid.back()), a0 = at.back();
					own.pop_back(); kid.pop_back(); at.pop_back();
					kid.back() = std::max(kid.back(), a0 + c);
				}
				{
					const long long crit = std::max(own[0], kid[0]);
					if (att_units > crit_best[0]) { crit_best[0] = att_units; crit_best[1] = crit; }
					if (att_units > o[6]) o[6] = att_units;
					__atomic_fetch_add(&crit_sum[0], att_units, __ATOMIC_RELAXED);
					__atomic_fetch_add(&crit_sum[1], crit, __ATOMIC_RELAXED);
					if (att_units > 5000) { __atomic_fetch_add(&crit_sum[2], att_units, __ATOMIC_RELAXED); __atomic_fetch_add(&crit_sum[3], crit, __ATOMIC_RELAXED); }
				}
				o[4]++;
				if (L.rc == 1 && pc_in_exit_window(Pm, ph)) break;
			}
			/* what a plain run (leak_calc=false) of the same slot sees: reflections of every attempt up to the transmitted one */
			for (uint32_t attempt = 0; attempt < (1u << 20); attempt++) {
				pc_start s;
				if (Pm.generic_src) pc_sample_photon<true>(Pm, seed, (uint64_t)(slot0 + j), attempt, s);
				else pc_sample_photon<false>(Pm, seed, (uint64_t)(slot0 + j), attempt, s);
				pc_photon<1> q;
				q.wmem = nullptr; q.wstride = 1;
				int st = pc_launch_init(T, Pm, q, s.x, s.y, s.z, s.dx, s.dy, s.dz, s.ex, s.ey, s.ez);
				while (st != PC_ST_DONE) st = (st == PC_ST_MARCH) ? pc_march_step(T, Pm, q) : pc_event(T, Pm, t.ec.data(), q);
				o[7] += q.irefl + 1;
				if (q.rc == 2) o[8]++;
				o[9]++;
				if (q.rc == 1 && pc_in_exit_window(Pm, q)) break;
			}
		}
	}
	return 0;
}
