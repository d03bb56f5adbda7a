/* Analysis tool (not part of the product or the tests): march steps per flight for a configurable set of block-certificate
 * strides, on the host compile of the device header.  Built and driven by scripts/analysis/flight_stats.py. */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "pc_problem.h"

namespace {
struct Lev { int L; std::vector<double> mb, md; };

void build_level(const pc_hip_problem *p, const pc_host_tables &t, int L, Lev &lv)
{
	const int n = p->nmax + 1;
	double capmin = HUGE_VAL, capmax = 0., extmax = 0.;
	for (int i = 0; i < n; i++) { capmin = std::fmin(capmin, p->cap[i]); capmax = std::fmax(capmax, p->cap[i]); extmax = std::fmax(extmax, std::fabs(p->ext[i])); }
	const double m = std::fmax(1e-6*capmin*capmin, 1e-10*capmax*extmax);
	lv.L = L; lv.mb.assign(n, HUGE_VAL); lv.md.assign(n, HUGE_VAL);
	for (int i = 0; i + L < n; i++) {
		const double za = p->z[i], span = p->z[i+L] - za;
		double dzh = 0., dr = 0.;
		for (int j = i + 1; j < i + L; j++) {
			double u = (p->z[j] - za)/span;
			dzh = std::fmax(dzh, std::fabs(t.zh[j] - (t.zh[i] + (t.zh[i+L] - t.zh[i])*u)));
			dr = std::fmax(dr, std::fabs(p->cap[j] - (p->cap[i] + (p->cap[i+L] - p->cap[i])*u)));
		}
		dzh = dzh*(1. + 1e-9) + 1e-12*std::fabs(t.zh[i]);
		dr = dr*(1. + 1e-9) + 1e-12*capmax;
		const double dR = p->cap[i+L] - p->cap[i];
		lv.mb[i] = 0.25*dR*dR + 2.*capmax*dr + m;
		lv.md[i] = dzh;
	}
}
}

extern "C" int flight_stats(const pc_hip_problem *p, int64_t n, const double *start, const double *dir, const double *elecv,
                            int nlev, const int *strides, int lv_first, int lv_later, int nprobe, int64_t *out /* [4]: first flights, their steps, later flights, their steps */,
                            int64_t *hist_first, int64_t *hist_later /* [64] */, int64_t *kinds /* [8] */)
{
	pc_host_tables t; std::string err;
	if (pc_build_tables(p, t, err)) return 1;
	pc_tables T;
	T.z = t.z.data(); T.cap = t.cap.data(); T.zh = t.zh.data(); T.cap2 = t.cap2.data(); T.hexd = t.hexd.data(); T.idz = t.idz.data(); T.ext = t.ext.data();
	T.mg = t.mg.data();
	std::vector<Lev> lev(nlev);
	for (int l = 0; l < nlev; l++) build_level(p, t, strides[l], lev[l]);
	const pc_params &Pm = t.pm;
	memset(out, 0, 4*sizeof(int64_t)); memset(kinds, 0, 8*sizeof(int64_t));
	memset(hist_first, 0, 64*sizeof(int64_t)); memset(hist_later, 0, 64*sizeof(int64_t));
	for (int64_t j = 0; j < n; j++) {
		pc_photon<1> ph; ph.wmem = nullptr; ph.wstride = 0;
		const double *s = start + 3*j, *d = dir + 3*j, *e = elecv + 3*j;
		int st = pc_launch_init(T, Pm, ph, s[0], s[1], s[2], d[0], d[1], d[2], e[0], e[1], e[2]);
		int steps = 0, lvcap = ph.bnd ? 0 : lv_first; int seenwide = 0, i_begin = ph.i;
		bool firstflight = true;
		while (st != PC_ST_DONE) {
			if (st == PC_ST_MARCH) {
				steps++;
				if (ph.i >= Pm.nmax) { ph.rc = 1; st = PC_ST_DONE; }
				else if (ph.first) { st = pc_march_first_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT; if (!firstflight) kinds[0]++; }
				else {
					/* generic-level version of pc_march_ok */
					const int i0 = ph.i;
					if (nprobe > 0 && !ph.bnd) {
						/* multi-probe step: level l > 0 probes nodes i0 + k*L/nprobe (k = 1..nprobe) against the block margin of stride L
						 * (far probes against twice the margin: they are compared with the chord), level 0 scans nprobe
						 * consecutive nodes with the single-segment margin */
						int lv = 0; double marg = Pm.adj;
						for (int l = 0; l < lvcap && l < nlev; l++) {
							if (i0 + lev[l].L > Pm.nmax) break;
							double kd = ph.kn*lev[l].md[i0];
							double ml = kd*(Pm.two_rmax + kd) + lev[l].mb[i0];
							if (ph.C0 < -ml) { lv = l + 1; marg = ml; }
						}
						int adv = 0; double Cadv = ph.C0;
						if (lv == 0) {
							for (int k = 1; k <= nprobe && i0 + k <= Pm.nmax; k++) {
								double Ck = pc_node_C(T, ph, i0 + k);
								if (!(Ck < -Pm.adj) || !(Cadv < -Pm.adj)) break;
								adv = k; Cadv = Ck;
							}
							if (!firstflight) kinds[adv ? (seenwide ? 4 : 1) : 7]++;
							if (adv) { ph.C0 = Cadv; ph.i = i0 + adv; }
							else st = PC_ST_EVENT;
						} else {
							const int stp = lev[lv-1].L / nprobe;
							for (int k = nprobe; k >= 1; k--) {
								double Ck = pc_node_C(T, ph, i0 + k*stp);
								if (Ck < -(k == nprobe ? marg : 2.*marg)) { adv = k*stp; Cadv = Ck; break; }
							}
							if (!firstflight) { kinds[adv ? 2 : 3]++; if (adv) seenwide = 1; }
							if (adv) { ph.C0 = Cadv; ph.i = i0 + adv; if (adv < lev[lv-1].L) lvcap = lv - 1; }
							else lvcap = lv - 1;
						}
						continue;
					}
					int lv = 0; double marg = Pm.adj;
					for (int l = 0; l < lvcap && l < nlev; l++) {
						if (i0 + lev[l].L > Pm.nmax) break;
						double kd = ph.kn*lev[l].md[i0];
						double ml = kd*(Pm.two_rmax + kd) + lev[l].mb[i0];
						if (ph.C0 < -ml) { lv = l + 1; marg = ml; }
					}
					const int i1 = i0 + (lv ? lev[lv-1].L : 1);
					double C1 = pc_node_C(T, ph, i1);
					int ok = (ph.C0 < -marg) & (C1 < -marg);
					if (ph.bnd) {
						double px = fma(ph.sx, T.z[i0], ph.ox), py = fma(ph.sy, T.z[i0], ph.oy);
						ok &= !pc_outside_hexd(T.hexd[i0], ph.kx*T.zh[i0], ph.ky*T.zh[i0]);
						ok &= !pc_outside_hexd(T.hexd[i1], ph.kx*T.zh[i1], ph.ky*T.zh[i1]);
						ok &= !pc_outside_hexd(T.hexd[i0], px, py);
					}
					if (!firstflight) {
						if (lv > 0) { kinds[ok ? 2 : 3]++; if (ok) seenwide = 1; }
						else kinds[seenwide ? 4 : 1]++;      /* single-segment steps before / after the first wide stride */
					}
					if (ok) { ph.C0 = C1; ph.i = i1; }
					else if (lv > 0) lvcap = lv - 1;
					else st = PC_ST_EVENT;
				}
			}
			if (st == PC_ST_EVENT || st == PC_ST_DONE) {
				if (st == PC_ST_EVENT) {
					const int before = ph.irefl;
					st = pc_event(T, Pm, t.ec.data(), ph);
					if (st == PC_ST_MARCH && ph.irefl == before) continue;   /* no hit: the flight goes on */
				}
				/* flight over */
				out[firstflight ? 0 : 2]++; out[firstflight ? 1 : 3] += steps;
				(firstflight ? hist_first : hist_later)[steps < 63 ? steps : 63]++;
				if (!firstflight) { kinds[5] += ph.i - i_begin; kinds[6] += (seenwide == 0); }
				steps = 0; firstflight = false; seenwide = 0; i_begin = ph.i;
				lvcap = ph.bnd ? 0 : lv_later;
			}
		}
	}
	return 0;
}
