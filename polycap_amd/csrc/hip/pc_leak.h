/*
 * pc_leak.h -- the leak ("halo") half of the photon trace, leak_calc=true, as device functions.
 *
 * What the reference does (file:line of the reference checkout):
 *   polycap_capil_trace_wall             src/polycap-capil.c:893-1194   path of the refracted ray through the glass
 *   leak branch of polycap_capil_reflect src/polycap-capil.c:610-619, 657-887
 *   polycap_photon_pc_intersect          src/polycap-photon.c:171-362   where a ray leaves the outer hexagon
 *   in-wall branch of photon_launch      src/polycap-photon.c:645-907
 *
 * How it is laid out here.  The reference recurses: every reflection may send the transmitted fraction into the
 * neighbouring capillary as a photon of its own, which reflects and leaks again (depth = number of walls crossed,
 * several hundred for a steep hard photon).  A lane runs that tree depth-first with an explicit stack of suspended
 * parents in HBM (frames of PC_LF_HDR + n_energies doubles), so there is no device recursion and no per-photon
 * allocation.  Leak events are appended to one record buffer through an atomic cursor, tagged (slot, attempt, seq);
 * seq numbers the events of one attempt in the order the reference would append them to photon->extleak/intleak, so a
 * sort by (slot, attempt, seq) reproduces its lists.  An attempt that ends in an error appends one VOID record
 * instead of retracting its events (the reference frees the photon together with its lists).
 *
 * The certified march of pc_device.h carries the photons between reflections unchanged.  The wall search keeps the
 * reference's cap/10 stepping (its answers -- d_travel, the cell indices -- are quantised by those steps).
 */
#ifndef PC_LEAK_H
#define PC_LEAK_H

#include "pc_device.h"

/* leak record, in doubles */
enum { PC_LR_SLOT = 0, PC_LR_ATTEMPT, PC_LR_SEQ, PC_LR_KIND, PC_LR_X, PC_LR_Y, PC_LR_Z, PC_LR_DX, PC_LR_DY, PC_LR_DZ,
       PC_LR_EX, PC_LR_EY, PC_LR_EZ, PC_LR_NREFL, PC_LR_WEIGHTS, PC_LR_HDR = 14 };
enum { PC_LEAK_EXT = 0, PC_LEAK_INT = 1, PC_LEAK_VOID = -1 };

/* suspended parent frame, in doubles; the parent's weights follow */
enum { PC_LF_PX = 0, PC_LF_PY, PC_LF_PZ, PC_LF_DX, PC_LF_DY, PC_LF_DZ, PC_LF_EX, PC_LF_EY, PC_LF_EZ, PC_LF_KX, PC_LF_KY,
       PC_LF_DTRAVEL, PC_LF_NX, PC_LF_NY, PC_LF_NZ, PC_LF_COSALFA, PC_LF_IX, PC_LF_IREFL, PC_LF_CALLS, PC_LF_KEEP,
       PC_LF_BND, PC_LF_ENTRANCE, PC_LF_HDR = 24 };

struct pc_leak_sink {
	double *records;               /* capacity x (PC_LR_HDR + n_energies) */
	unsigned long long *cursor;    /* records appended so far (keeps counting past capacity: the caller sees the need) */
	long long capacity;
};

struct pc_leak_ctx {
	const pc_energy_const *ec;
	const double *amu;             /* linear attenuation coefficient per energy (src/polycap-photon.c:87) */
	int ne;
	double *frames;                /* this lane's stack: max_depth x (PC_LF_HDR + ne); frame 0 holds the launched photon's weights */
	int max_depth;
	pc_leak_sink sink;
	double slot, attempt;
	int seq;
	int stack_overflow;            /* the tree was deeper than max_depth: the run is reported as failed */
};

PC_HD unsigned long long pc_cursor_next(unsigned long long *c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return atomicAdd(c, 1ull);
#else
	return (*c)++;
#endif
}

PC_HD void pc_leak_emit(pc_leak_ctx &cx, int kind, double x, double y, double z, double dx, double dy, double dz,
                        double ex, double ey, double ez, int nrefl, const double *w)
{
	const unsigned long long k = pc_cursor_next(cx.sink.cursor);
	const int seq = cx.seq++;
	if ((long long)k >= cx.sink.capacity) return;
	double *r = cx.sink.records + (long long)k * (PC_LR_HDR + cx.ne);
	r[PC_LR_SLOT] = cx.slot; r[PC_LR_ATTEMPT] = cx.attempt; r[PC_LR_SEQ] = (double)seq; r[PC_LR_KIND] = (double)kind;
	r[PC_LR_X] = x; r[PC_LR_Y] = y; r[PC_LR_Z] = z;
	r[PC_LR_DX] = dx; r[PC_LR_DY] = dy; r[PC_LR_DZ] = dz;
	r[PC_LR_EX] = ex; r[PC_LR_EY] = ey; r[PC_LR_EZ] = ez;
	r[PC_LR_NREFL] = (double)nrefl;
	for (int e = 0; e < cx.ne; e++) r[PC_LR_WEIGHTS + e] = (w != nullptr) ? w[e] : 0.;
}

/* axial hexagon coordinates of the cell containing (x, y) with cube rounding: src/polycap-capil.c:959-971 */
PC_HD void pc_hex_index(double x, double y, double zz, double &q_i, double &r_i)
{
	r_i = y * (2./3) / zz;
	q_i = (x/(2.*PC_COSPI_6) - y/3) / zz;
	double rq = round(q_i), rr = round(r_i), rs = round(-1.*q_i - r_i);
	double dq = fabs(q_i - rq), dr = fabs(r_i - rr), ds = fabs(-1.*q_i - r_i - rs);
	if (dq > dr && dq > ds) {
		q_i = -1.*rr - rs;
		r_i = rr;
	} else if (dr > ds) {
		r_i = -1.*rq - rs;
		q_i = rq;
	} else {
		q_i = rq;
		r_i = rr;
	}
}

/* last node index in [0, upto) whose z does not exceed zval (0 when there is none): what the reference's linear
 * rescans compute (e.g. src/polycap-capil.c:921-924), by bisection since z is strictly increasing */
PC_HD int pc_node_find(const pc_tables &T, int upto, double zval)
{
	int lo = 0, hi = upto - 1;
	if (hi < 0 || !(T.z[0] <= zval)) return 0;
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (T.z[mid] <= zval) lo = mid; else hi = mid - 1;
	}
	return lo;
}

/* last node index below `upto` whose z does not exceed zval, moving from a previous answer (the reference rescans
 * all nodes, src/polycap-capil.c:1023-1026; z is strictly increasing, so the answers agree) */
PC_HD int pc_node_follow(const pc_tables &T, int upto, int z_id, double zval)
{
	while (z_id + 1 < upto && T.z[z_id+1] <= zval) z_id++;
	while (z_id > 0 && T.z[z_id] > zval) z_id--;
	return z_id;
}

/* analysis builds (scripts/analysis/leak_units.cpp) count what the units of the wall search end as */
#ifdef PC_LEAK_STATS
extern long long pc_leak_stats[16];
#define PC_LSTAT(k) (__atomic_fetch_add(&pc_leak_stats[k], 1, __ATOMIC_RELAXED))
#else
#define PC_LSTAT(k) ((void)0)
#endif

/* ------------------------------------------------------------------ src/polycap-photon.c:171-362
 * Where the ray that ended at (cx, cy, cz) outside the optic crossed its outer hexagon, searched backwards.
 * Returns 0 where the reference returns NULL. */
PC_HD int pc_outer_intersect(const pc_tables &T, const pc_params &Pm, double cx, double cy, double cz,
                             double dx, double dy, double dz, double &ox, double &oy, double &oz)
{
	const int nmax = Pm.nmax;
	if (dz == 0.) return 0;
	double bx = -1.*dx, by = -1.*dy, bz = -1.*dz;
	pc_norm3(bx, by, bz);
	int z_id = (nmax >= 1 && cz >= T.z[nmax-1]) ? nmax-1 : pc_node_find(T, nmax, cz);       /* called with the exit plane's z */
	double cur_ext = (T.ext[z_id+1]-T.ext[z_id])/(T.z[z_id+1]-T.z[z_id]) * (cz - T.z[z_id]) + T.ext[z_id];
	/* polycap_photon_within_pc_boundary: 1 inside, 0 outside, -1 for a non-positive radius */
	const int here = (cur_ext <= 0.) ? -1 : (pc_outside_hex(cur_ext, cx, cy) ? 0 : 1);
	if (here == 1) return 0;
	int dir;
	if (bz < 0.) { z_id = z_id + 1; dir = -1; } else { dir = 1; }
	int broke = 0;
	PC_LSTAT(10);
	/* ---- certified skipping.  The scan looks for the first node where the ray is not outside the hexagon any more; from the
	 * side of the optic back to that node it is often a hundred nodes.  The ray is linear in z and ext (= zh * hexscale) stays
	 * within hexscale*md_L of its chord over L segments (pc_marg4), so when at both end nodes of a block the same edge of the
	 * hexagon is exceeded by more than cos30*hexscale*md_L (+ margin), every node of the block tests "outside" (and has a
	 * positive ext): the block is passed without visiting its nodes. */
	const double ibz = 1./bz;
	int cool = 0;
	for (;;) {
		if (!Pm.literal && here == 0 && cool == 0) {
			int did = 0;
			for (int lv = 2; lv >= 1 && !did; lv--) {
				const int Lb = (lv == 2) ? PC_L2 : PC_L1;
				const int ia = (dir < 0) ? z_id - Lb : z_id, ib = ia + Lb;      /* nodes ia .. ib; z_id itself lies behind */
				if (ia < 0 || ib > nmax) continue;
				const double dev = Pm.hexscale * (double)((lv == 2) ? T.mg[ia].md2 : T.mg[ia].md1) * (1. + 1.e-6);
				const double ea = T.ext[ia], eb = T.ext[ib];
				if (!(ea > dev && eb > dev)) continue;
				const double ta = (T.z[ia] - cz)*ibz, tb = (T.z[ib] - cz)*ibz;
				const double xa = cx + bx*ta, ya = cy + by*ta, xb = cx + bx*tb, yb = cy + by*tb;
				const double m = PC_COSPI_6*dev + 1.e-9*((ea > eb) ? ea : eb);
				const double da = T.hexd[ia] + m, db = T.hexd[ib] + m;
				const double s2a = PC_COSPI_6*xa + 0.5*ya, s2b = PC_COSPI_6*xb + 0.5*yb;
				const double s3a = PC_COSPI_6*xa - 0.5*ya, s3b = PC_COSPI_6*xb - 0.5*yb;
				if ((ya > da && yb > db) || (-ya > da && -yb > db) || (s2a > da && s2b > db) || (-s2a > da && -s2b > db)
				    || (s3a > da && s3b > db) || (-s3a > da && -s3b > db)) {
					z_id = (dir < 0) ? ia : ib;
					did = 1;
				}
			}
			if (did) { PC_LSTAT(12); continue; }
			cool = PC_L1;
		}
		if (cool > 0) cool--;
		z_id += dir;
		PC_LSTAT(11);
		if (z_id < 0 || z_id > nmax) break;     /* the reference reads past the profile here; nothing can be found there */
		double t = (T.z[z_id] - cz);
		double tx = cx + bx * t / bz;
		double ty = cy + by * t / bz;
		const int there = (T.ext[z_id] <= 0.) ? -1 : (pc_outside_hex(T.ext[z_id], tx, ty) ? 0 : 1);
		if (here != there) { broke = 1; break; }
	}
	if (!broke) return 0;
	const int zo = z_id - dir;
	if (zo < 0 || zo > nmax) return 0;
	double tb = T.z[z_id] - cz, te = T.z[zo] - cz;
	double begx = cx + bx * tb / bz, begy = cy + by * tb / bz;
	double endx = cx + bx * te / bz, endy = cy + by * te / bz, endz = cz + bz * te / bz;
	double eb = T.ext[z_id], ee = T.ext[zo];
	double hb = sqrt((eb * eb) - ((eb/2.) * (eb/2.)));
	double he = sqrt((ee * ee) - ((ee/2.) * (ee/2.)));
	double dp1b = fabs(0*begx + 1*begy), dp2b = fabs(PC_COSPI_6*begx + 0.5*begy), dp3b = fabs(PC_COSPI_6*begx + -0.5*begy);
	double dp1e = fabs(0*endx + 1*endy), dp2e = fabs(PC_COSPI_6*endx + 0.5*endy), dp3e = fabs(PC_COSPI_6*endx + -0.5*endy);
	/* :262-264 as written: an interpolation between the two ext values that is then compared with z */
	double z1 = (dp1b - hb) / (hb-he - dp1b+dp1e) * (eb-ee) + eb;
	double z2 = (dp2b - hb) / (hb-he - dp2b+dp2e) * (eb-ee) + eb;
	double z3 = (dp3b - hb) / (hb-he - dp3b+dp3e) * (eb-ee) + eb;
	const double lo = (dir < 0) ? T.z[z_id] : T.z[zo];
	const double hi = (dir < 0) ? T.z[zo] : T.z[z_id];
	const int v1 = (z1 >= lo && z1 <= hi), v2 = (z2 >= lo && z2 <= hi), v3 = (z3 >= lo && z3 <= hi);
	double z_fin;
	if (dir < 0) {
		if (v1 && v2 && v3) {
			if (z1 >= z2 && z1 >= z3) z_fin = z1;
			else if (z2 >= z1 && z2 >= z3) z_fin = z2;
			else if (z3 >= z1 && z3 >= z2) z_fin = z3;
			else return 0;
		} else if (v2 && v3) z_fin = (z3 > z2) ? z3 : z2;
		else if (v1 && v3) z_fin = (z1 > z3) ? z1 : z3;
		else if (v1 && v2) z_fin = (z1 > z2) ? z1 : z2;
		else if (v1) z_fin = z1;
		else if (v2) z_fin = z2;
		else if (v3) z_fin = z3;
		else { ox = endx; oy = endy; oz = endz; return 1; }
	} else {
		if (v1 && v2 && v3) {
			if (z1 <= z2 && z1 <= z3) z_fin = z1;
			else if (z2 <= z1 && z2 <= z3) z_fin = z2;
			else if (z3 <= z1 && z3 <= z2) z_fin = z3;
			else return 0;
		} else if (v2 && v3) z_fin = (z3 < z2) ? z3 : z2;
		else if (v1 && v3) z_fin = (z1 < z3) ? z1 : z3;
		else if (v1 && v2) z_fin = (z1 < z2) ? z1 : z2;
		else if (v1) z_fin = z1;
		else if (v2) z_fin = z2;
		else if (v3) z_fin = z3;
		else { ox = endx; oy = endy; oz = endz; return 1; }
	}
	double tf = z_fin - cz;
	ox = cx + bx * tf / bz;
	oy = cy + by * tf / bz;
	oz = cz + bz * tf / bz;
	return 1;
}

/* ==================================================================== the lane state machine
 *
 * A launch is cut into units of work so that the lanes of a wave can be scheduled by kind of work (pc_leak_kernels.h):
 * every unit advances one lane by one march step, one segment visit, one block/step of the wall search, one segment
 * probe of the neighbouring capillary, or one of the short bookkeeping steps.  Run one after the other on a single lane
 * (pc_leak_launch below, which the host compile of the tests uses) the units are exactly the sequential algorithm.
 */
enum {
	PC_LS_MARCH = 0,       /* certified march between interactions (pc_march_step) */
	PC_LS_EVENT,           /* literal visit of one segment (pc_event_pre) */
	PC_LS_REFLECT,         /* a wall hit is due: geometry of the reflection, start of the wall search */
	PC_LS_WALL_STEP,       /* wall search: cap/10 stepping through the glass (src/polycap-capil.c:1016-1064) */
	PC_LS_WALL_PROBE,      /* wall search: segments of the neighbouring capillary (:1105-1128) */
	PC_LS_REFLECT_END,     /* weights, leak events, child photon or mirror reflection (:625-887, :1345-1355) */
	PC_LS_INWALL_END,      /* launch inside the glass: what the wall search found (src/polycap-photon.c:698-870) */
	PC_LS_ENDED,           /* the photon being traced has ended: pop a suspended parent or finish the launch */
	PC_LS_DONE             /* launch finished, rc holds polycap_photon_launch's return code */
};

struct pc_wall {
	int z_id, seg_step, seg_slope, cool, iesc, wt;
	int units;                         /* units of work spent on this search (guard against a search that cannot end) */
	long long nst;
	double q_i, r_i, q_new, r_new;
	double px, py, pz;                 /* point reached by the stepping */
	double dist, base, step;           /* dist = base + nst*step */
	double ext_slope, cap_slope, seg_z, seg_ext, seg_cap;
	double hx, hy, hz;                 /* intersection with the neighbouring capillary */
	double d_travel, q_out, r_out;     /* results */
	double kn_new;                     /* |(kx, ky)| of the capillary (q_new, r_new) being probed */
};

struct pc_leak_lane {
	pc_photon<0> ph;
	pc_hit h;
	pc_refl_geom g;
	pc_wall w;
	pc_leak_ctx cx;
	double *wts;                       /* weights of the photon being traced (in its stack frame) */
	double z0, sdx, sdy, sdz;
	int st, rc;
	int lvl, calls, entrance, have_final, how, geom_ok;
	int after_wall;                    /* state that takes over when the wall search is finished */
};

/* how the photon being traced came to an end, in polycap_capil_trace's return codes */
enum { PC_END_ABSORBED = 0, PC_END_CALLS = 1, PC_END_ERROR = -1, PC_END_EXIT = -2 };

template <int NE>
PC_HD void pc_frame_save(double *f, const pc_photon<NE> &ph, const pc_hit &h, int calls, int keep, int entrance)
{
	f[PC_LF_PX] = ph.Px; f[PC_LF_PY] = ph.Py; f[PC_LF_PZ] = ph.Pz;
	f[PC_LF_DX] = ph.dx; f[PC_LF_DY] = ph.dy; f[PC_LF_DZ] = ph.dz;
	f[PC_LF_EX] = ph.ex; f[PC_LF_EY] = ph.ey; f[PC_LF_EZ] = ph.ez;
	f[PC_LF_KX] = ph.kx; f[PC_LF_KY] = ph.ky;
	f[PC_LF_DTRAVEL] = ph.dtravel;
	f[PC_LF_NX] = h.nx; f[PC_LF_NY] = h.ny; f[PC_LF_NZ] = h.nz; f[PC_LF_COSALFA] = h.cosalfa;
	f[PC_LF_IX] = (double)h.ix; f[PC_LF_IREFL] = (double)ph.irefl; f[PC_LF_CALLS] = (double)calls;
	f[PC_LF_KEEP] = (double)keep; f[PC_LF_BND] = (double)ph.bnd; f[PC_LF_ENTRANCE] = (double)entrance;
}

template <int NE>
PC_HD void pc_frame_load(const double *f, pc_photon<NE> &ph, pc_hit &h, int &calls, int &keep, int &entrance)
{
	ph.Px = f[PC_LF_PX]; ph.Py = f[PC_LF_PY]; ph.Pz = f[PC_LF_PZ];
	ph.dx = f[PC_LF_DX]; ph.dy = f[PC_LF_DY]; ph.dz = f[PC_LF_DZ];
	ph.ex = f[PC_LF_EX]; ph.ey = f[PC_LF_EY]; ph.ez = f[PC_LF_EZ];
	ph.kx = f[PC_LF_KX]; ph.ky = f[PC_LF_KY];
	ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
	ph.dtravel = f[PC_LF_DTRAVEL];
	h.nx = f[PC_LF_NX]; h.ny = f[PC_LF_NY]; h.nz = f[PC_LF_NZ]; h.cosalfa = f[PC_LF_COSALFA];
	h.ix = (int)f[PC_LF_IX]; ph.irefl = (int)f[PC_LF_IREFL]; calls = (int)f[PC_LF_CALLS];
	keep = (int)f[PC_LF_KEEP]; ph.bnd = (int)f[PC_LF_BND]; entrance = (int)f[PC_LF_ENTRANCE];
}

/* boundary-capillary flag of pc_launch_init for the axis (kx, ky) */
template <int NE>
PC_HD void pc_set_boundary_flag(const pc_params &Pm, pc_photon<NE> &ph)
{
	if (!Pm.mono) {
		double m1 = fabs(ph.ky);
		double m2 = fabs(PC_COSPI_6*ph.kx + 0.5*ph.ky);
		double m3 = fabs(PC_COSPI_6*ph.kx - 0.5*ph.ky);
		double M = fmax(m1, fmax(m2, m3));
		ph.bnd = (PC_COSPI_6 - M/Pm.hexscale > Pm.bnd_thresh) ? 0 : 1;
	} else {
		ph.bnd = 1;
	}
}


/* ---- wall search, src/polycap-capil.c:893-1194: from ph.P along ph.d through the glass.
 * Result in L.w.wt: 1 enters capillary (q_out, r_out) after d_travel; 2 reaches the exit plane inside the glass; 3 leaves
 * the optic through its side; <= 0 nothing to trace. */

/* end of the search without an intersection (:1068-1100 with the stepped point, :1147-1172 with the exit-plane point) */
PC_HD void pc_wall_to_exit(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int stepped)
{
	pc_wall &W = L.w;
	const pc_photon<0> &ph = L.ph;
	const int nmax = Pm.nmax;
	const double zend = T.z[nmax], ext_end = T.ext[nmax];
	const double tx = ph.Px + ph.dx * (zend-ph.Pz)/ph.dz;
	const double ty = ph.Py + ph.dy * (zend-ph.Pz)/ph.dz;
	W.r_out = W.r_new; W.q_out = W.q_new;
	double rx, ry, rz;
	if (stepped) { rx = W.px - ph.Px; ry = W.py - ph.Py; rz = W.pz - ph.Pz; }
	else { rx = tx - ph.Px; ry = ty - ph.Py; rz = zend - ph.Pz; }
	if (pc_outside_hex(ext_end, tx, ty)) {
		double ix_, iy_, iz_;
		if (pc_outer_intersect(T, Pm, tx, ty, zend, ph.dx, ph.dy, ph.dz, ix_, iy_, iz_)) {
			rx = ix_ - ph.Px; ry = iy_ - ph.Py; rz = iz_ - ph.Pz;
		}
		W.wt = 3;
	} else {
		W.wt = 2;
	}
	W.d_travel = sqrt(rx*rx + ry*ry + rz*rz);
}

/* :918-971.  Returns the next state: WALL_STEP / WALL_PROBE, or `after` with W.wt <= 0 when there is nothing to trace */
/* `hint`: a node index near ph.Pz (the segment of the reflection) or -1 */
PC_HD int pc_wall_begin(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int after, int hint = -1)
{
	pc_wall &W = L.w;
	const pc_photon<0> &ph = L.ph;
	const int nmax = Pm.nmax;
	L.after_wall = after;
	W.d_travel = 0.; W.q_out = 0.; W.r_out = 0.; W.q_new = 0.; W.r_new = 0.;
	W.wt = -2;
	if (ph.Pz >= T.z[nmax]) return after;
	int z_id = (hint >= 0 && hint < nmax) ? pc_node_follow(T, nmax, hint, ph.Pz) : pc_node_find(T, nmax, ph.Pz);     /* the same index either way */
	double cur_ext;
	if (T.z[z_id] != ph.Pz)
		cur_ext = ((T.ext[z_id+1] - T.ext[z_id])/(T.z[z_id+1] - T.z[z_id])) * (ph.Pz - T.z[z_id]) + T.ext[z_id];
	else
		cur_ext = T.ext[z_id];
	if (Pm.mono) {
		if (sqrt(ph.Px*ph.Px + ph.Py*ph.Py) > cur_ext) return after;
	} else {
		if (pc_outside_hex(cur_ext, ph.Px, ph.Py)) return after;
	}
	pc_hex_index(ph.Px, ph.Py, cur_ext/Pm.hexscale, W.q_i, W.r_i);
	W.z_id = z_id;
	W.px = ph.Px; W.py = ph.Py; W.pz = ph.Pz;
	W.hx = ph.Px; W.hy = ph.Py; W.hz = ph.Pz;
	W.dist = 0.; W.base = 0.; W.nst = 0; W.step = 0.;
	W.seg_step = -1; W.seg_slope = -1; W.cool = 0;
	W.iesc = 0; W.units = 0;
	W.kn_new = 0.;                     /* mono-capillary: the probed axis is the optic's */
	PC_LSTAT(8);
	return Pm.mono ? PC_LS_WALL_PROBE : PC_LS_WALL_STEP;
}

#ifndef PC_PROBE_BLOCKS
#define PC_PROBE_BLOCKS 16 /* certified blocks of segments the capillary probe may skip between two segment visits; measured on MI355X
                            * (scripts/ab_leak.sh, mean life of a wave for 262144 slots): 1 -> 155 ms, 2 -> 147, 4 -> 138, 8 -> 130, 16 -> 127, 64 -> 127 */
#endif
#ifndef PC_PROBE_VISITS
#define PC_PROBE_VISITS 1  /* segment visits (each preceded by its skipped blocks) one unit of the capillary probe may make: 1, 4, 16 measure the same */
#endif
#ifndef PC_WALL_SEGS
#define PC_WALL_SEGS 300      /* profile segments one unit of the wall search may step through (the rest of a certified stretch waits for the next unit) */
#endif
#ifndef PC_WALL_LEVELS
#define PC_WALL_LEVELS 0   /* widest piece of the wall search: 0 = one profile segment, 1 = PC_L1, 2 = PC_L2 segments.  The wider pieces
                            * halve the units of a search (host count: 2.0 pieces + 2.0 literal steps per search instead of 4.6 + 2.0) and
                            * lose on the GPU: the lanes of a unit then step through 1 ... 25 segments each and wait for the longest
                            * (mean wave life 149 ms against 125 ms for 262144 slots, scripts/ab_leak.sh) */
#endif
#ifndef PC_WALL_PIECES
#define PC_WALL_PIECES 3   /* straight pieces (blocks of 1, PC_L1 or PC_L2 segments) one unit of the wall search may certify */
#endif

/* Largest certified fraction of one straight piece of the wall search.  In the plane of the cell (q_i, r_i) and relative to
 * its centre the piece runs from ua to ub (the ray minus K times the chord of zh over the block); ha, hb are the inradius of
 * the cell hexagon at both ends less every allowance (chord deviation, margin), rr the square of the radius the piece must
 * stay outside of (negative: no capillary to avoid).  Returns t in [0, 1]: on [0, t] of the piece every hexagon test holds
 * (|n.u| is convex along the piece and the bound linear, so both ends suffice) and the closest approach to the centre stays
 * outside rr; 0 when already the start point cannot be certified.  A t < 1 is estimated in single precision and then
 * VERIFIED in double precision, so the estimate decides nothing. */
PC_HD double pc_wall_reach(double uax, double uay, double ubx, double uby, double ha, double hb, double rr)
{
	const double a1 = uax, a2 = 0.5*uax + PC_COSPI_6*uay, a3 = 0.5*uax - PC_COSPI_6*uay;
	if (!(fabs(a1) <= ha && fabs(a2) <= ha && fabs(a3) <= ha)) return 0.;
	const double ex = ubx - uax, ey = uby - uay;
	const double ee = ex*ex + ey*ey, ae = uax*ex + uay*ey, aa = uax*uax + uay*uay;
	if (rr >= 0. && !(aa > rr)) return 0.;
	const double b1 = ubx, b2 = 0.5*ubx + PC_COSPI_6*uby, b3 = 0.5*ubx - PC_COSPI_6*uby;
	float t = 1.f;
	if (fabs(b1) > hb) { const double num = ha - ((b1 < 0.) ? -a1 : a1); const float tk = (float)num / (float)(num + (fabs(b1) - hb)); if (tk < t) t = tk; }
	if (fabs(b2) > hb) { const double num = ha - ((b2 < 0.) ? -a2 : a2); const float tk = (float)num / (float)(num + (fabs(b2) - hb)); if (tk < t) t = tk; }
	if (fabs(b3) > hb) { const double num = ha - ((b3 < 0.) ? -a3 : a3); const float tk = (float)num / (float)(num + (fabs(b3) - hb)); if (tk < t) t = tk; }
	if (rr >= 0. && ae < 0.) {
		const double disc = ae*ae - ee*(aa - rr);
		if (disc > 0.) { const float tc = (float)(-ae - sqrt(disc)) / (float)ee; if (tc < t) t = tc; }
	}
	double td = (t >= 1.f) ? 1. : (double)t * (1. - 1./32768.);
	for (int tries = 0; tries < 3; tries++) {
		if (!(td > 0.)) break;
		const double ux = uax + td*ex, uy = uay + td*ey, ht = ha + td*(hb - ha);
		int ok = (fabs(ux) <= ht && fabs(0.5*ux + PC_COSPI_6*uy) <= ht && fabs(0.5*ux - PC_COSPI_6*uy) <= ht);
		if (ok && rr >= 0. && ae < 0.) {
			/* squared distance of u_a + e s, s in [0, td], to the centre: at s = td when the foot of the perpendicular lies
			 * beyond, else |u_a|^2 - (u_a.e)^2/|e|^2 (compared after multiplying by |e|^2) */
			if (-ae >= ee*td) ok = (ux*ux + uy*uy) > rr;
			else ok = (aa*ee - ae*ae) > rr*ee;
		}
		if (ok) return td;
		td = (td == 1.) ? 0.99 : 0.5*td;
	}
	return 0.;
}

/* one unit of the stepping loop (:1016-1064): either one certified block of steps, or one literal step with its tests */
PC_HD int pc_wall_step(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int after)
{
	pc_wall &W = L.w;
	const pc_photon<0> &ph = L.ph;
	const int nmax = Pm.nmax;
	const double zend = T.z[nmax], ns = Pm.n_shells;
	const double Px = ph.Px, Py = ph.Py, Pz = ph.Pz, dx = ph.dx, dy = ph.dy, dz = ph.dz;
	int z_id = W.z_id;

	/* A search that has not ended after 2^28 units cannot end (e.g. a direction without z component inside a cell that
	 * never changes; the reference would loop forever): the reflection is reported as failed (launch returns -1). */
	if (++W.units > (1 << 28)) { W.wt = -3; return after; }

	/* ---- certified skipping.  Relative to the centre of cell (q_i, r_i) the ray is u(z) = p(z) - K*zh(z).  Inside one
	 * profile segment zh is linear, and over a block of PC_L1 / PC_L2 segments it stays within md_L (tabulated, pc_marg4) of
	 * the chord between the block's end nodes: u stays within |K| md_L of a straight piece, the cell's inradius within
	 * cos30 md_L of the chord's, the capillary never wider than the block's largest radius.  pc_wall_reach gives the
	 * fraction of such a piece on which no literal step can leave the loop (neither into another cell nor into the
	 * capillary), with the allowances taken off and a margin of 1e-6 of the cell size, far above rounding; a piece that is
	 * safe to its end is followed by the next one.  Up to the height zT reached this way only the count of steps per
	 * segment -- exactly the literal arithmetic's dist = base + nst*step -- is carried out. */
	if (!Pm.literal && W.cool == 0 && dz > 0. && T.z[z_id] <= W.pz && W.pz < T.z[z_id+1]) {
		const int inside_stack = (fabs(W.q_i) <= ns && fabs(W.r_i) <= ns && fabs(-1.*W.q_i-W.r_i) <= ns);
		const double Kx = (2.*W.q_i + W.r_i) * PC_COSPI_6, Ky = W.r_i * 1.5;
		const double kab = fabs(Kx) + fabs(Ky);          /* >= |K| */
		double zs = W.pz, xs = W.px, ys = W.py;          /* start of the next piece, in segment `is` (or at its first node) */
		int is = z_id;
		double zT = zs;
		for (int piece = 0; piece < PC_WALL_PIECES && is < nmax; piece++) {
			const pc_marg4 g = T.mg[is];
			const double zA = T.z[is], zhA = T.zh[is], zh1 = T.zh[is+1];
			const double zh_s = zhA + (zh1 - zhA)*((zs - zA)*T.idz[is]);
			const double capA = T.cap[is];
			const double r_s = capA + (T.cap[is+1] - capA)*((zs - zA)*T.idz[is]);      /* radius at the start of the piece */
			/* the widest block whose allowances are small against the room the start point has to the cell edge and to the
			 * capillary (a choice, not a certificate) */
			const double usx = xs - Kx*zh_s, usy = ys - Ky*zh_s;
			const double s1 = fabs(usx), s2 = fabs(0.5*usx + PC_COSPI_6*usy), s3 = fabs(0.5*usx - PC_COSPI_6*usy);
			const double room_h = PC_COSPI_6*zh_s - ((s1 > s2) ? ((s1 > s3) ? s1 : s3) : ((s2 > s3) ? s2 : s3));
			const double uu_s = usx*usx + usy*usy;
			int lv = 0;
			for (int l = 1; l <= PC_WALL_LEVELS; l++) {
				if (is + ((l == 2) ? PC_L2 : PC_L1) > nmax) break;
				const double al = 4.*((PC_COSPI_6 + kab)*(double)((l == 2) ? g.md2 : g.md1) + (double)((l == 2) ? T.dr[is].d2 : T.dr[is].d1));
				if (!(al < room_h) || (inside_stack && !(uu_s > (r_s + al)*(r_s + al)))) break;
				lv = l;
			}
			const int ib = is + ((lv == 2) ? PC_L2 : ((lv == 1) ? PC_L1 : 1));
			const double md = (lv == 2) ? (double)g.md2 : ((lv == 1) ? (double)g.md1 : 0.);
			const double zB = T.z[ib], zhB = T.zh[ib];
			const double fs = (lv == 0) ? 0. : pc_div_fast(zs - zA, zB - zA);
			const double zha = (lv == 0) ? zh_s : zhA + (zhB - zhA)*fs;    /* the chord at the start */
			const double tB = (zB - zs)*ph.idzd;
			const double xb = xs + tB*dx, yb = ys + tB*dy;
			const double margin = 1.e-6 * zha;
			const double dev = (PC_COSPI_6 + kab)*md*(1. + 1.e-6) + margin;
			double rr = -1.;
			if (inside_stack) {
				/* the radius along the piece: never above the larger end value of its chord plus the chord deviation of cap */
				const double capB = T.cap[ib];
				const double rc = (lv == 0) ? r_s : capA + (capB - capA)*fs;
				const double drd = (lv == 2) ? (double)T.dr[is].d2 : ((lv == 1) ? (double)T.dr[is].d1 : 0.);
				const double reach = ((rc > capB) ? rc : capB) + drd + kab*md*(1. + 1.e-6) + margin*(1. + 1.e-3);
				rr = reach*reach*(1. + 1.e-12);
			}
			const double t = pc_wall_reach(xs - Kx*zha, ys - Ky*zha, xb - Kx*zhB, yb - Ky*zhB,
			                               PC_COSPI_6*zha - dev, PC_COSPI_6*zhB - dev, rr);
			if (!(t > 0.)) break;
			zT = zs + t*(zB - zs);
			if (t < 1.) break;
			zs = zB; xs = xb; ys = yb; is = ib;
		}
		int advanced = 0;
		if (zT > W.pz) {
			for (int guard = 0; guard < PC_WALL_SEGS; guard++) {
				const double stp = T.stp[z_id];
				const double z1 = T.z[z_id+1];
				const double zlim = (z1 < zT) ? z1 : zT;
				/* candidate count of steps that land before zlim, checked against the positions the literal arithmetic produces */
				const double room = (zlim - W.pz)*ph.idzd*T.istp[z_id];
				int m = (room > 1.e6) ? 1000000 : (int)room;
				if (m < 0) m = 0;
				const long long n0 = (z_id != W.seg_step) ? 0 : W.nst;
				const double b0 = (z_id != W.seg_step) ? W.dist : W.base;
				if (m >= 1 && !(Pz + (b0 + (double)(n0 + m)*stp)*dz < zlim)) m--;
				if (m >= 1 && !(Pz + (b0 + (double)(n0 + m)*stp)*dz < zlim)) m--;
				if (m >= 1 && !(Pz + (b0 + (double)(n0 + m)*stp)*dz < zlim)) m = 0;
				/* the step over the node, when the certified stretch goes on behind it */
				int steps = m, on = 0;
				if (z1 < zT && Pz + (b0 + (double)(n0 + m + 1)*stp)*dz < zT) { steps = m + 1; on = 1; }
				if (steps > 0) {
					if (z_id != W.seg_step) { W.seg_step = z_id; W.step = stp; W.base = W.dist; W.nst = 0; }
					W.nst += steps;
					W.dist = W.base + (double)W.nst*W.step;
					W.px = Px + W.dist*dx;
					W.py = Py + W.dist*dy;
					W.pz = Pz + W.dist*dz;
					z_id = pc_node_follow(T, nmax, z_id, W.pz);
					advanced = 1;
				}
				if (!on) break;
			}
		}
		if (advanced) { W.z_id = z_id; PC_LSTAT(0); return PC_LS_WALL_STEP; }
		W.cool = 1;      /* a cell edge or the capillary is within a step or two: one literal step, then try again */
	}
	if (W.cool > 0) W.cool--;
	PC_LSTAT(1);

	/* ---- one literal step.  dist = base + nst*step: the path length after nst steps of the current size (the reference
	 * adds the steps one by one; the product differs from that sum by rounding only and lets a block be skipped in O(1)) */
	if (z_id != W.seg_step) { W.seg_step = z_id; W.step = T.cap[z_id]/10.; W.base = W.dist; W.nst = 0; }
	W.nst++;
	W.dist = W.base + (double)W.nst*W.step;
	const double px = Px + W.dist*dx, py = Py + W.dist*dy, pz = Pz + W.dist*dz;
	W.px = px; W.py = py; W.pz = pz;
	z_id = pc_node_follow(T, nmax, z_id, pz);
	W.z_id = z_id;
	if (z_id != W.seg_slope) {
		W.seg_slope = z_id;
		const double dzs = (T.z[z_id+1] - T.z[z_id]);
		W.seg_z = T.z[z_id]; W.seg_ext = T.ext[z_id]; W.seg_cap = T.cap[z_id];
		W.ext_slope = (T.ext[z_id+1] - W.seg_ext)/dzs;
		W.cap_slope = (T.cap[z_id+1] - W.seg_cap)/dzs;
	}
	const double cur_ext = W.ext_slope * (pz - W.seg_z) + W.seg_ext;
	const double rad0 = W.cap_slope * (pz - W.seg_z) + W.seg_cap;
	const double zz = cur_ext/Pm.hexscale;
	pc_hex_index(px, py, zz, W.q_new, W.r_new);
	/* :1043-1063 the ray found the capillary (q_i, r_i) it started next to */
	const double ccy = W.r_i * (3./2) * zz;
	const double ccx = (2.* W.q_i+W.r_i) * PC_COSPI_6 * zz;
	const double d_phot0 = sqrt((px-ccx)*(px-ccx)+(py-ccy)*(py-ccy));
	if (d_phot0 < rad0 && fabs(W.q_i) <= ns && fabs(W.r_i) <= ns && fabs(-1.*W.q_i-W.r_i) <= ns) {
		const double rx = px - Px, ry = py - Py, rz = pz - Pz;
		const double dt = sqrt(rx*rx + ry*ry + rz*rz);
		if (dt > 1.e-5) {
			W.d_travel = dt; W.r_out = W.r_i; W.q_out = W.q_i; W.wt = 1;
			return after;
		}
	}
	if (W.q_new == W.q_i && W.r_new == W.r_i && pz <= zend)
		return PC_LS_WALL_STEP;

	/* :1068-1100 outside the hexagon stacking, or beyond the exit plane */
	if (fabs(W.q_new) > ns || fabs(W.r_new) > ns || fabs(-1.*W.q_new-W.r_new) > ns || pz > zend) {
		pc_wall_to_exit(T, Pm, L, 1);
		return after;
	}
	/* :1105 on to the wall of capillary (q_new, r_new) */
	W.iesc = 0;
	W.hx = Px; W.hy = Py; W.hz = Pz;
	PC_LSTAT(2);
	{
		const double kx = (2.* W.q_new+W.r_new) * PC_COSPI_6, ky = W.r_new * (3./2);
		W.kn_new = sqrt(kx*kx + ky*ky);
	}
	return PC_LS_WALL_PROBE;
}

/* one unit of the segment search in the neighbouring capillary (:1107-1128; mono-capillary :991-1011) */
PC_HD int pc_wall_probe(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int after)
{
	pc_wall &W = L.w;
	const int nmax = Pm.nmax;
	pc_photon<0> probe = L.ph;          /* same ray, other capillary axis */
	if (Pm.mono) { probe.kx = 0.; probe.ky = 0.; }
	else { probe.ky = W.r_new * (3./2); probe.kx = (2.* W.q_new+W.r_new) * PC_COSPI_6; }
	/* ---- certain misses.  Relative to the axis of the probed capillary the ray is q(z) = p(z) - K*zh(z).  Over L segments
	 * q stays within kn*md_L of the chord between its two end values (md_L: tabulated chord deviation of zh, 0 for L = 1)
	 * and the radius never exceeds R_blk (the largest of the block, pc_marg4::r2 / 2), so when the chord's closest approach to
	 * the axis is farther than R_blk + kn*md_L (+ margin) the ray is outside the capillary on all L segments: the reference's
	 * quadratic (src/polycap-capil.c:119-171) has no root inside any of them and every one of the L visits is a miss.  A unit
	 * skips up to PC_PROBE_BLOCKS such blocks, visits the segment that could not be skipped, and goes on PC_PROBE_VISITS times. */
	for (int visit = 0; visit < PC_PROBE_VISITS; visit++) {
	int skipped = 0;
	for (int blk = 0; blk < (Pm.literal ? 0 : PC_PROBE_BLOCKS); blk++) {
		skipped = 0;
		const int i0 = W.z_id;
		const double kn = W.kn_new;        /* set where the probe of this capillary begins: one root per capillary, not per unit */
		const pc_marg4 g = T.mg[i0];
		const double z0 = T.z[i0], zh0 = T.zh[i0];
		const double ax = fma(-probe.kx, zh0, fma(probe.sx, z0, probe.ox));
		const double ay = fma(-probe.ky, zh0, fma(probe.sy, z0, probe.oy));
		for (int lv = 2; lv >= 0 && !skipped; lv--) {
			const int Ls = (lv == 2) ? PC_L2 : ((lv == 1) ? PC_L1 : 1);
			if (Ls > 1 && i0 + Ls > nmax-1) continue;      /* the literal loop stops before segment nmax-1 */
			const int i1 = i0 + Ls;
			const double z1 = T.z[i1], zh1 = T.zh[i1];
			const double bx = fma(-probe.kx, zh1, fma(probe.sx, z1, probe.ox));
			const double by = fma(-probe.ky, zh1, fma(probe.sy, z1, probe.oy));
			const double ex = bx - ax, ey = by - ay;
			const double ee = ex*ex + ey*ey, ae = ax*ex + ay*ey, aa = ax*ax + ay*ay;
			double reach;
			if (Ls == 1) {
				const double r0 = T.cap[i0], r1 = T.cap[i1];
				reach = (r0 > r1) ? r0 : r1;
			} else {
				reach = 0.5*(double)g.r2 + kn * (double)((lv == 2) ? g.md2 : g.md1);
			}
			reach += 1.e-7 * (0.5*Pm.two_rmax);
			/* squared distance of the chord a + e t, t in [0, 1], to the axis, without the quotient of the foot point: at t = 0
			 * when the chord moves away, at t = 1 when the foot lies beyond, else |a|^2 - (a.e)^2/|e|^2 (compared after
			 * multiplying by |e|^2; a certificate: 1e-7 of margin against 1e-16 of rounding) */
			const double rr = reach*reach;
			int far;
			if (!(ae < 0.)) far = aa > rr;
			else if (-ae >= ee) far = (bx*bx + by*by) > rr;
			else far = (aa*ee - ae*ae) > rr*ee*(1. + 1.e-12);
			if (far) {
				W.z_id = i1;
				W.iesc = -3;
				skipped = 1;
				PC_LSTAT(3 + lv);
			}
		}
		if (!skipped || W.z_id >= nmax-1) break;
	}
	if (!skipped) {
		double p0x, p0y, nx, ny, nz;
		W.iesc = pc_segment(T, probe, W.z_id, p0x, p0y, W.hx, W.hy, W.hz, nx, ny, nz);
		PC_LSTAT(W.iesc == 1 ? 7 : 6);
		W.z_id++;
	}
	if (!(W.iesc != 1 && W.z_id < nmax-1)) break;          /* found the wall, or the end of the capillary */
	}
	if (W.iesc != 1 && W.z_id < nmax-1)
		return PC_LS_WALL_PROBE;
	if (!Pm.mono && W.z_id >= nmax && W.iesc != 0) {
		/* :1129-1135 nothing in this capillary: on to the next hexagon cell */
		W.q_i = W.q_new;
		W.r_i = W.r_new;
		W.z_id = nmax-1;
		return PC_LS_WALL_STEP;
	}
	/* :1142-1190 */
	W.r_out = W.r_new; W.q_out = W.q_new;
	if (W.iesc != 1) {
		pc_wall_to_exit(T, Pm, L, 0);
		return after;
	}
	const double rx = W.hx - L.ph.Px, ry = W.hy - L.ph.Py, rz = W.hz - L.ph.Pz;
	W.d_travel = sqrt(rx*rx + ry*ry + rz*rz);
	W.wt = (W.z_id >= nmax) ? 2 : 1;
	return after;
}

/* ---- launch ------------------------------------------------------------------------------------------------------ */

/* `L.ph` comes from pc_launch_init (st0 = its return value), z0 = start_coords.z; L.cx is set up by the caller. */
PC_HD void pc_leak_begin(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int st0, double z0)
{
	pc_photon<0> &ph = L.ph;
	const int ne = L.cx.ne;
	L.lvl = 0;
	L.wts = L.cx.frames + PC_LF_HDR;
	for (int e = 0; e < ne; e++) L.wts[e] = 1.;
	ph.wmem = L.wts; ph.wstride = 1; ph.wset = 1;
	L.cx.seq = 0;
	L.calls = Pm.nmax + 1;             /* polycap_capil_trace calls left in the loop that drives this photon */
	L.entrance = 0; L.have_final = 0; L.how = 0; L.rc = 0;
	L.z0 = z0;
	L.sdx = ph.dx; L.sdy = ph.dy; L.sdz = ph.dz;
	L.h.nx = L.h.ny = L.h.nz = L.h.cosalfa = 0.; L.h.ix = 0;
	if (st0 != PC_ST_DONE) { L.st = PC_LS_MARCH; return; }
	if (ph.rc != 2) { L.rc = ph.rc; L.st = PC_LS_DONE; return; }       /* -2: missed the optic */
	pc_trace_begin(ph);                                                  /* ray constants for the wall search */
	if (z0 == 0.) {
		/* src/polycap-photon.c:647-672: reflection off the entrance face, normal = optic axis */
		L.h.nx = 0.; L.h.ny = 0.; L.h.nz = 1.; L.h.cosalfa = ph.dz; L.h.ix = 0;
		L.entrance = 1;
		L.have_final = 1;
		L.st = PC_LS_REFLECT;
	} else {
		/* :674-870 launched inside the glass */
		L.have_final = 2;
		L.st = pc_wall_begin(T, Pm, L, PC_LS_INWALL_END);
	}
}

/* the short states; MARCH, WALL_STEP and WALL_PROBE units are called directly by the schedulers */
PC_HD void pc_leak_unit_other(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L)
{
	pc_photon<0> &ph = L.ph;
	pc_leak_ctx &cx = L.cx;
	const int ne = cx.ne, nmax = Pm.nmax;
	const long fstride = PC_LF_HDR + ne;
	switch (L.st) {
	case PC_LS_EVENT: {
		const int st = pc_event_pre(T, Pm, ph, L.h);
		if (st == PC_ST_MARCH) { L.st = PC_LS_MARCH; }
		else if (st == PC_ST_REFLECT) { L.calls--; L.st = PC_LS_REFLECT; }
		else { L.calls--; L.how = (ph.rc == 1) ? PC_END_EXIT : PC_END_ERROR; L.st = PC_LS_ENDED; }
		break;
	}
	case PC_LS_REFLECT: {
		/* src/polycap-capil.c:596-619 */
		L.geom_ok = pc_reflect_geom(ph, L.h.nx, L.h.ny, L.h.nz, L.g);     /* -1: alfa < 0 */
		L.w.wt = 0;
		L.st = (L.geom_ok >= 0) ? pc_wall_begin(T, Pm, L, PC_LS_REFLECT_END, L.h.ix) : PC_LS_REFLECT_END;
		break;
	}
	case PC_LS_REFLECT_END: {
		/* :625-887 */
		int r = L.geom_ok;
		int wt = L.w.wt;
		const double dtr = L.w.d_travel, qn = L.w.q_out, rn = L.w.r_out;
		if (r >= 0 && wt <= 0) r = -1;
		int leak_flag = 0, keep = 0;
		double *w = L.wts, *wl = nullptr;
		/* the transmitted fractions go straight into the next frame's weight slots: they are the child's weights if one
		 * is spawned, and scratch otherwise */
		if (r >= 0 && L.lvl + 1 >= cx.max_depth) { cx.stack_overflow = 1; r = -1; }
		if (r >= 0) {
			wl = cx.frames + (long)(L.lvl + 1)*fstride + PC_LF_HDR;
			for (int e = 0; e < ne; e++) {
				double rtot, r_rough;
				if (pc_fresnel(cx.ec[e], L.g, rtot, r_rough, cx.ne == 1) < 0) { r = -1; break; }
				wl[e] = (1.-rtot * r_rough) * w[e] * exp(-1.*dtr*cx.amu[e]);
				if (wl[e] >= 1.e-4) leak_flag = 1;
				w[e] = w[e] * rtot * r_rough;
				if (w[e] >= 1.e-4) keep = 1;
			}
		}
		if (r >= 0) {
			ph.ex = fabs(ph.ex); ph.ey = fabs(ph.ey); ph.ez = fabs(ph.ez);
			r = keep;
			if (leak_flag) {
				const double f = dtr / sqrt(ph.dx*ph.dx + ph.dy*ph.dy + ph.dz*ph.dz);
				const double lx = ph.Px + f*ph.dx, ly = ph.Py + f*ph.dy, lz = ph.Pz + f*ph.dz;
				if (wt == 1) {
					const int zi = pc_node_find(T, nmax, lz);
					const double ce = ((T.ext[zi+1] - T.ext[zi])/(T.z[zi+1] - T.z[zi])) * (lz - T.z[zi]) + T.ext[zi];
					if (Pm.mono) { if (sqrt(lx*lx + ly*ly) >= ce) wt = 3; }
					else if (ce > 0. && pc_outside_hex(ce, lx, ly)) wt = 3;
				}
				if (wt == 3) pc_leak_emit(cx, PC_LEAK_EXT, lx, ly, lz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wl);
				if (wt == 2) pc_leak_emit(cx, PC_LEAK_INT, lx, ly, lz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wl);
				if (wt == 1 && lz < T.z[nmax]) {
					/* :711-803 suspend this photon, go on with the leaked fraction in capillary (qn, rn) */
					pc_frame_save(cx.frames + (long)L.lvl*fstride, ph, L.h, L.calls, keep, L.entrance);
					L.lvl++;
					L.wts = wl;
					ph.wmem = wl;
					ph.Px = lx; ph.Py = ly; ph.Pz = lz;
					ph.dtravel = ph.dtravel + dtr;
					if (Pm.mono) { ph.kx = 0.; ph.ky = 0.; }
					else { ph.ky = (3./2) * rn; ph.kx = (2.*qn + rn) * PC_COSPI_6; }
					ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
					pc_set_boundary_flag(Pm, ph);
					ph.i = pc_node_find(T, nmax + 1, lz);
					L.calls = nmax + 1 - ph.i;
					ph.rc = 0;
					L.entrance = 0;
					pc_trace_begin(ph);
					L.st = (L.calls <= 0) ? PC_LS_ENDED : PC_LS_MARCH;
					L.how = PC_END_CALLS;
					break;
				}
			}
		}
		/* no child: finish the reflection (:1345-1355) */
		if (L.entrance) { L.entrance = 0; L.how = (r < 0) ? PC_END_ERROR : PC_END_ABSORBED; L.st = PC_LS_ENDED; }
		else if (r < 0) { L.how = PC_END_ERROR; L.st = PC_LS_ENDED; }
		else if (r == 0) { L.how = PC_END_ABSORBED; L.st = PC_LS_ENDED; }
		else {
			ph.dx = fma(-2.0*L.h.cosalfa, L.h.nx, ph.dx);
			ph.dy = fma(-2.0*L.h.cosalfa, L.h.ny, ph.dy);
			ph.dz = fma(-2.0*L.h.cosalfa, L.h.nz, ph.dz);
			ph.irefl++;
			ph.i = L.h.ix;
			pc_trace_begin(ph);
			if (L.calls <= 0) { L.how = PC_END_CALLS; L.st = PC_LS_ENDED; }
			else L.st = PC_LS_MARCH;
		}
		break;
	}
	case PC_LS_INWALL_END: {
		/* src/polycap-photon.c:676-806 */
		const int wt = L.w.wt;
		const double dtr = L.w.d_travel, qn = L.w.q_out, rn = L.w.r_out;
		double *w = L.wts;
		if (wt <= 0) { ph.rc = -1; L.rc = -1; L.st = PC_LS_DONE; break; }
		for (int e = 0; e < ne; e++) w[e] = w[e] * exp(-1.*dtr*cx.amu[e]);
		const double f = dtr / sqrt(ph.dx*ph.dx + ph.dy*ph.dy + ph.dz*ph.dz);
		ph.Px = ph.Px + f*ph.dx; ph.Py = ph.Py + f*ph.dy; ph.Pz = ph.Pz + f*ph.dz;
		if (wt == 3) pc_leak_emit(cx, PC_LEAK_EXT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
		if (wt == 2) pc_leak_emit(cx, PC_LEAK_INT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
		if (wt == 1) {
			ph.dtravel = ph.dtravel + dtr;
			ph.ky = rn * (3./2);
			ph.kx = (2.*qn + rn) * PC_COSPI_6;
			ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
			pc_set_boundary_flag(Pm, ph);
			ph.i = pc_node_find(T, nmax + 1, ph.Pz);
			ph.rc = 0;
			pc_trace_begin(ph);
			L.st = PC_LS_MARCH;
		} else {
			L.have_final = 3;        /* nothing to trace: straight to the common tail of :860-870 */
			L.how = PC_END_CALLS;
			L.st = PC_LS_ENDED;
		}
		break;
	}
	case PC_LS_ENDED: {
		/* the photon being traced has ended */
		for (;;) {
			double *w = L.wts;
			if (L.lvl == 0) {
				if (L.have_final == 1) {                  /* entrance reflection: launch returns 2 whatever happened inside */
					if (L.how == PC_END_ERROR && cx.seq > 0) pc_leak_emit(cx, PC_LEAK_VOID, 0,0,0, 0,0,0, 0,0,0, 0, nullptr);
					ph.rc = 2; L.rc = 2; L.st = PC_LS_DONE; break;
				}
				if (L.how == PC_END_ERROR) {
					if (cx.seq > 0) pc_leak_emit(cx, PC_LEAK_VOID, 0,0,0, 0,0,0, 0,0,0, 0, nullptr);
					ph.rc = -1; L.rc = -1; L.st = PC_LS_DONE; break;
				}
				if (L.have_final >= 2) {
					/* src/polycap-photon.c:808-870: a photon launched inside the glass ends as a leak event itself */
					if (L.have_final == 2 && (L.how == PC_END_CALLS || L.how == PC_END_EXIT)) {
						const double t = (T.z[nmax]-ph.Pz)/ph.dz;
						ph.Px = ph.Px + ph.dx * t; ph.Py = ph.Py + ph.dy * t; ph.Pz = ph.Pz + ph.dz * t;
						const int inside = (T.ext[nmax] <= 0.) ? -1 : (pc_outside_hex(T.ext[nmax], ph.Px, ph.Py) ? 0 : 1);
						if (inside == 0) pc_leak_emit(cx, PC_LEAK_EXT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
						else if (inside == 1) pc_leak_emit(cx, PC_LEAK_INT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
					}
					for (int e = 0; e < ne; e++) w[e] = 0.;
					ph.Px = T.ext[nmax]+1.; ph.Py = T.ext[nmax]+1.; ph.Pz = T.z[nmax];
					ph.dx = L.sdx; ph.dy = L.sdy; ph.dz = L.sdz;
					ph.rc = 1; L.rc = 1; L.st = PC_LS_DONE; break;
				}
				ph.rc = (L.how == PC_END_ABSORBED) ? 0 : 1;
				L.rc = ph.rc; L.st = PC_LS_DONE; break;
			}
			/* a leaked fraction has ended: src/polycap-capil.c:810-880 */
			if (L.how == PC_END_ERROR) {
				/* reflect returns -2 -> the parent's trace returns -1 -> ... : the whole launch fails */
				L.lvl = 0;
				L.wts = cx.frames + PC_LF_HDR;
				continue;
			}
			double cxl = 0., cyl = 0., czl = 0.;
			int final_kind = -2;
			if (L.how == PC_END_CALLS || L.how == PC_END_EXIT) {
				const double t = (T.z[nmax]-ph.Pz)/ph.dz;
				cxl = ph.Px + ph.dx * t; cyl = ph.Py + ph.dy * t; czl = ph.Pz + ph.dz * t;
				const int inside = (T.ext[nmax] <= 0.) ? -1 : (pc_outside_hex(T.ext[nmax], cxl, cyl) ? 0 : 1);
				final_kind = (inside == 0) ? PC_LEAK_EXT : ((inside == 1) ? PC_LEAK_INT : -2);
			}
			const double *wchild = w;
			L.lvl--;
			int keep;
			pc_frame_load(cx.frames + (long)L.lvl*fstride, ph, L.h, L.calls, keep, L.entrance);
			L.wts = cx.frames + (long)L.lvl*fstride + PC_LF_HDR;
			ph.wmem = L.wts;
			/* its last state is one more event, with THIS photon's direction, electric vector and reflection count */
			if (final_kind != -2)
				pc_leak_emit(cx, final_kind, cxl, cyl, czl, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wchild);
			/* resume the suspended reflection where it stopped */
			if (L.entrance) { L.how = PC_END_ABSORBED; L.entrance = 0; continue; }
			if (keep == 0) { L.how = PC_END_ABSORBED; continue; }
			ph.dx = fma(-2.0*L.h.cosalfa, L.h.nx, ph.dx);
			ph.dy = fma(-2.0*L.h.cosalfa, L.h.ny, ph.dy);
			ph.dz = fma(-2.0*L.h.cosalfa, L.h.nz, ph.dz);
			ph.irefl++;
			ph.i = L.h.ix;
			ph.rc = 0;
			pc_trace_begin(ph);
			if (L.calls <= 0) { L.how = PC_END_CALLS; continue; }
			L.st = PC_LS_MARCH;
			break;
		}
		break;
	}
	default:
		break;
	}
}

/* one march step of a lane in PC_LS_MARCH */
PC_HD void pc_leak_unit_march(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L)
{
	const int st = pc_march_step(T, Pm, L.ph);
	if (st == PC_ST_EVENT) L.st = PC_LS_EVENT;
	else if (st == PC_ST_DONE) { L.calls--; L.how = (L.ph.rc == 1) ? PC_END_EXIT : PC_END_ERROR; L.st = PC_LS_ENDED; }
}

/* the whole launch on one lane, unit after unit: the sequential algorithm (used by the host compile of the tests) */
PC_HD int pc_leak_launch(const pc_tables &T, const pc_params &Pm, pc_leak_lane &L, int st0, double z0)
{
	pc_leak_begin(T, Pm, L, st0, z0);
	while (L.st != PC_LS_DONE) {
		if (L.st == PC_LS_MARCH) pc_leak_unit_march(T, Pm, L);
		else if (L.st == PC_LS_WALL_STEP) L.st = pc_wall_step(T, Pm, L, L.after_wall);
		else if (L.st == PC_LS_WALL_PROBE) L.st = pc_wall_probe(T, Pm, L, L.after_wall);
		else pc_leak_unit_other(T, Pm, L);
	}
	return L.rc;
}

#endif /* PC_LEAK_H */
