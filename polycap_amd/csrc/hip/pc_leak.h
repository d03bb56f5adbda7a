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

/* last node index below `upto` whose z does not exceed zval, moving from a previous answer (the reference rescans
 * all nodes, src/polycap-capil.c:1023-1026; z is strictly increasing, so the answers agree) */
PC_HD int pc_node_follow(const pc_tables &T, int upto, int z_id, double zval)
{
	while (z_id + 1 < upto && T.z[z_id+1] <= zval) z_id++;
	while (z_id > 0 && T.z[z_id] > zval) z_id--;
	return z_id;
}

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
	int z_id = pc_last_node_le(T, nmax, cz);
	double cur_ext = (T.ext[z_id+1]-T.ext[z_id])/(T.z[z_id+1]-T.z[z_id]) * (cz - T.z[z_id]) + T.ext[z_id];
	/* polycap_photon_within_pc_boundary: 1 inside, 0 outside, -1 for a non-positive radius */
	const int here = (cur_ext <= 0.) ? -1 : (pc_outside_hex(cur_ext, cx, cy) ? 0 : 1);
	if (here == 1) return 0;
	int dir;
	if (bz < 0.) { z_id = z_id + 1; dir = -1; } else { dir = 1; }
	int broke = 0;
	for (;;) {
		z_id += dir;
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

/* ------------------------------------------------------------------ src/polycap-capil.c:893-1194
 * From the last interaction point ph.P along ph.d through the glass.  1: enters capillary (q, r) after d_travel;
 * 2: reaches the exit plane inside the glass; 3: leaves the optic through its side; <= 0: nothing to trace. */
template <int NE>
PC_HD int pc_trace_wall(const pc_tables &T, const pc_params &Pm, const pc_photon<NE> &ph,
                        double &d_travel, double &q_out, double &r_out)
{
	const int nmax = Pm.nmax;
	const double zend = T.z[nmax], ext_end = T.ext[nmax], ns = Pm.n_shells;
	const double Px = ph.Px, Py = ph.Py, Pz = ph.Pz, dx = ph.dx, dy = ph.dy, dz = ph.dz;
	d_travel = 0.; q_out = 0.; r_out = 0.;
	if (Pz >= zend) return -2;
	int z_id = pc_last_node_le(T, nmax, Pz);
	double cur_ext;
	if (T.z[z_id] != Pz)
		cur_ext = ((T.ext[z_id+1] - T.ext[z_id])/(T.z[z_id+1] - T.z[z_id])) * (Pz - T.z[z_id]) + T.ext[z_id];
	else
		cur_ext = T.ext[z_id];
	if (Pm.mono) {
		if (sqrt(Px*Px + Py*Py) > cur_ext) return -2;
	} else {
		if (pc_outside_hex(cur_ext, Px, Py)) return -2;
	}
	double q_i, r_i, q_new = 0., r_new = 0.;
	pc_hex_index(Px, Py, cur_ext/Pm.hexscale, q_i, r_i);

	pc_photon<NE> probe = ph;      /* same ray, other capillary axes */
	int iesc = 0;
	double p0x, p0y, hx = Px, hy = Py, hz = Pz, nx, ny, nz;
	double px = Px, py = Py, pz = Pz;

	if (Pm.mono) {
		/* :991-1011 */
		probe.kx = 0.; probe.ky = 0.;
		do {
			iesc = pc_segment(T, probe, z_id, p0x, p0y, hx, hy, hz, nx, ny, nz);
			z_id++;
		} while (iesc != 1 && z_id < nmax-1);
	} else {
		double dist = 0.;
		for (;;) {
			/* :1016-1064 cap/10 steps until the hexagon cell changes */
			do {
				dist += T.cap[z_id]/10.;
				px = Px + dist*dx;
				py = Py + dist*dy;
				pz = Pz + dist*dz;
				z_id = pc_node_follow(T, nmax, z_id, pz);
				const double idzs = (T.z[z_id+1] - T.z[z_id]);
				cur_ext = ((T.ext[z_id+1] - T.ext[z_id])/idzs) * (pz - T.z[z_id]) + T.ext[z_id];
				const double rad0 = ((T.cap[z_id+1] - T.cap[z_id])/idzs) * (pz - T.z[z_id]) + T.cap[z_id];
				const double zz = cur_ext/Pm.hexscale;
				pc_hex_index(px, py, zz, q_new, r_new);
				/* :1043-1063 the ray found the capillary (q_i, r_i) it started next to */
				const double ccy = r_i * (3./2) * zz;
				const double ccx = (2.* q_i+r_i) * PC_COSPI_6 * zz;
				const double d_phot0 = sqrt((px-ccx)*(px-ccx)+(py-ccy)*(py-ccy));
				if (d_phot0 < rad0 && fabs(q_i) <= ns && fabs(r_i) <= ns && fabs(-1.*q_i-r_i) <= ns) {
					const double rx = px - Px, ry = py - Py, rz = pz - Pz;
					const double dt = sqrt(rx*rx + ry*ry + rz*rz);
					if (dt > 1.e-5) {
						d_travel = dt; r_out = r_i; q_out = q_i;
						return 1;
					}
				}
			} while (q_new == q_i && r_new == r_i && pz <= zend);

			/* :1068-1100 outside the hexagon stacking, or beyond the exit plane */
			if (fabs(q_new) > ns || fabs(r_new) > ns || fabs(-1.*q_new-r_new) > ns || pz > zend) {
				const double tx = Px + dx * (zend-Pz)/dz;
				const double ty = Py + dy * (zend-Pz)/dz;
				r_out = r_new; q_out = q_new;
				double rx = px - Px, ry = py - Py, rz = pz - Pz;
				if (pc_outside_hex(ext_end, tx, ty)) {
					double ix_, iy_, iz_;
					if (pc_outer_intersect(T, Pm, tx, ty, zend, dx, dy, dz, ix_, iy_, iz_)) {
						rx = ix_ - Px; ry = iy_ - Py; rz = iz_ - Pz;
					}
					d_travel = sqrt(rx*rx + ry*ry + rz*rz);
					return 3;
				}
				d_travel = sqrt(rx*rx + ry*ry + rz*rz);
				return 2;
			}

			/* :1105-1128 wall of capillary (q_new, r_new), segment by segment */
			probe.ky = r_new * (3./2);
			probe.kx = (2.* q_new+r_new) * PC_COSPI_6;
			iesc = 0;
			hx = Px; hy = Py; hz = Pz;
			do {
				iesc = pc_segment(T, probe, z_id, p0x, p0y, hx, hy, hz, nx, ny, nz);
				z_id++;
			} while (iesc != 1 && z_id < nmax-1);
			if (z_id >= nmax && iesc != 0) {
				/* :1129-1135 */
				q_i = q_new;
				r_i = r_new;
				z_id = nmax-1;
				continue;
			}
			break;
		}
	}

	/* :1142-1190 */
	r_out = r_new; q_out = q_new;
	if (iesc != 1) {
		const double tx = Px + dx * (zend-Pz)/dz;
		const double ty = Py + dy * (zend-Pz)/dz;
		double rx = tx - Px, ry = ty - Py, rz = zend - Pz;
		if (pc_outside_hex(ext_end, tx, ty)) {
			double ix_, iy_, iz_;
			if (pc_outer_intersect(T, Pm, tx, ty, zend, dx, dy, dz, ix_, iy_, iz_)) {
				rx = ix_ - Px; ry = iy_ - Py; rz = iz_ - Pz;
			}
			d_travel = sqrt(rx*rx + ry*ry + rz*rz);
			return 3;
		}
		d_travel = sqrt(rx*rx + ry*ry + rz*rz);
		return 2;
	}
	{
		const double rx = hx - Px, ry = hy - Py, rz = hz - Pz;
		d_travel = sqrt(rx*rx + ry*ry + rz*rz);
	}
	return (z_id >= nmax) ? 2 : 1;
}

/* ------------------------------------------------------------------ the depth-first run of one launched photon */

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

/* Runs the launched photon and everything that leaks out of it.  `ph` comes from pc_launch_init (st0 = its return
 * value); weights live in frame 0 of cx.frames.  z0 = start_coords.z.  Returns polycap_photon_launch's return code. */
PC_HD int pc_leak_launch(const pc_tables &T, const pc_params &Pm, pc_leak_ctx &cx, pc_photon<0> &ph, int st0, double z0)
{
	const int ne = cx.ne, nmax = Pm.nmax;
	const long fstride = PC_LF_HDR + ne;
	int lvl = 0;                       /* frame of the photon being traced; frames below it hold its suspended ancestors */
	double *w = cx.frames + PC_LF_HDR;  /* its weights */
	for (int e = 0; e < ne; e++) w[e] = 1.;
	ph.wmem = w; ph.wstride = 1; ph.wset = 1;
	cx.seq = 0;
	int calls = nmax + 1;              /* polycap_capil_trace calls left in the loop that drives this photon */
	int st = st0;
	int entrance = 0;                  /* the reflection in progress is the one off the entrance face (return code ignored) */
	int pending = 0;                   /* 1: a reflection at hit h is due for the current photon */
	pc_hit h;
	h.nx = h.ny = h.nz = h.cosalfa = 0.; h.ix = 0;
	int have_final = 0;               /* 1: entrance reflection (launch returns 2); 2, 3: launched inside the glass */

	const double sdx = ph.dx, sdy = ph.dy, sdz = ph.dz;       /* normalised start direction */
	if (st0 == PC_ST_DONE) {
		if (ph.rc != 2) return ph.rc;                        /* -2: missed the optic */
		pc_trace_begin(ph);                                  /* ray constants for the wall search */
		if (z0 == 0.) {
			/* src/polycap-photon.c:647-672: reflection off the entrance face, normal = optic axis */
			h.nx = 0.; h.ny = 0.; h.nz = 1.; h.cosalfa = ph.dz; h.ix = 0;
			entrance = 1; pending = 1;
			have_final = 1;
		} else {
			/* :674-870 launched inside the glass */
			double dtr, qn, rn;
			const int wt = pc_trace_wall(T, Pm, ph, dtr, qn, rn);
			if (wt <= 0) { ph.rc = -1; return -1; }
			for (int e = 0; e < ne; e++) w[e] = w[e] * exp(-1.*dtr*cx.amu[e]);
			const double f = dtr / sqrt(ph.dx*ph.dx + ph.dy*ph.dy + ph.dz*ph.dz);
			ph.Px = ph.Px + f*ph.dx; ph.Py = ph.Py + f*ph.dy; ph.Pz = ph.Pz + f*ph.dz;
			if (wt == 3) pc_leak_emit(cx, PC_LEAK_EXT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
			if (wt == 2) pc_leak_emit(cx, PC_LEAK_INT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
			have_final = 2;      /* the launched photon itself ends as a leak event and is reported absorbed (:808-870) */
			if (wt == 1) {
				ph.dtravel = ph.dtravel + dtr;
				ph.ky = rn * (3./2);
				ph.kx = (2.*qn + rn) * PC_COSPI_6;
				ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
				pc_set_boundary_flag(Pm, ph);
				ph.i = pc_last_node_le(T, nmax + 1, ph.Pz);
				ph.rc = 0;
				pc_trace_begin(ph);
				st = PC_ST_MARCH;
			} else {
				calls = 0;       /* nothing to trace: straight to the common tail below */
				have_final = 3;
			}
		}
	}

	for (;;) {
		int ended = 0, how = 0;
		if (!pending) {
			/* ---- fly to the next wall hit: certified march + literal segment visits of pc_device.h */
			if (calls <= 0) { ended = 1; how = PC_END_CALLS; }
			while (!ended && !pending) {
				while (st == PC_ST_MARCH) st = pc_march_step(T, Pm, ph);
				if (st == PC_ST_EVENT) st = pc_event_pre(T, Pm, ph, h);
				if (st == PC_ST_REFLECT) { pending = 1; calls--; }
				else if (st == PC_ST_DONE) { ended = 1; how = (ph.rc == 1) ? PC_END_EXIT : PC_END_ERROR; calls--; }
			}
		}
		if (pending) {
			/* ---- reflection with leak bookkeeping: src/polycap-capil.c:596-887 */
			pending = 0;
			pc_refl_geom g;
			int r = pc_reflect_geom(ph, h.nx, h.ny, h.nz, g);     /* -1: alfa < 0 */
			double dtr = 0., qn = 0., rn = 0.;
			int wt = 0;
			if (r >= 0) {
				wt = pc_trace_wall(T, Pm, ph, dtr, qn, rn);
				if (wt <= 0) r = -1;
			}
			int leak_flag = 0, keep = 0;
			double *wl = nullptr;
			if (r >= 0) {
				/* the transmitted fractions go straight into the next frame's weight slots: they are the child's weights
				 * if one is spawned, and scratch otherwise */
				if (lvl + 1 >= cx.max_depth) { cx.stack_overflow = 1; r = -1; }
			}
			if (r >= 0) {
				wl = cx.frames + (long)(lvl + 1)*fstride + PC_LF_HDR;
				for (int e = 0; e < ne; e++) {
					double rtot, r_rough;
					if (pc_fresnel(cx.ec[e], g, rtot, r_rough) < 0) { r = -1; break; }
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
						const int zi = pc_last_node_le(T, nmax, lz);
						const double ce = ((T.ext[zi+1] - T.ext[zi])/(T.z[zi+1] - T.z[zi])) * (lz - T.z[zi]) + T.ext[zi];
						if (Pm.mono) { if (sqrt(lx*lx + ly*ly) >= ce) wt = 3; }
						else if (ce > 0. && pc_outside_hex(ce, lx, ly)) wt = 3;
					}
					if (wt == 3) pc_leak_emit(cx, PC_LEAK_EXT, lx, ly, lz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wl);
					if (wt == 2) pc_leak_emit(cx, PC_LEAK_INT, lx, ly, lz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wl);
					if (wt == 1 && lz < T.z[nmax]) {
						/* :711-803 suspend this photon, go on with the leaked fraction in capillary (qn, rn) */
						pc_frame_save(cx.frames + (long)lvl*fstride, ph, h, calls, keep, entrance);
						lvl++;
						w = wl;
						ph.wmem = w;
						ph.Px = lx; ph.Py = ly; ph.Pz = lz;
						ph.dtravel = ph.dtravel + dtr;
						if (Pm.mono) { ph.kx = 0.; ph.ky = 0.; }
						else { ph.ky = (3./2) * rn; ph.kx = (2.*qn + rn) * PC_COSPI_6; }
						ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
						pc_set_boundary_flag(Pm, ph);
						ph.i = pc_last_node_le(T, nmax + 1, lz);
						calls = nmax + 1 - ph.i;
						ph.rc = 0;
						entrance = 0;
						pc_trace_begin(ph);
						st = PC_ST_MARCH;
						continue;
					}
				}
			}
			/* ---- no child: finish the reflection (src/polycap-capil.c:1345-1355) */
			if (entrance) { ended = 1; how = PC_END_ABSORBED; entrance = 0; if (r < 0) how = PC_END_ERROR; }
			else if (r < 0) { ended = 1; how = PC_END_ERROR; }
			else if (r == 0) { ended = 1; how = PC_END_ABSORBED; }
			else {
				ph.dx = fma(-2.0*h.cosalfa, h.nx, ph.dx);
				ph.dy = fma(-2.0*h.cosalfa, h.ny, ph.dy);
				ph.dz = fma(-2.0*h.cosalfa, h.nz, ph.dz);
				ph.irefl++;
				ph.i = h.ix;
				pc_trace_begin(ph);
				st = PC_ST_MARCH;
				continue;
			}
		}

		/* ---- the photon being traced has ended */
		for (;;) {
			if (lvl == 0) {
				if (have_final == 1)                    /* entrance reflection: launch returns 2 whatever happened inside */
					{ if (how == PC_END_ERROR) pc_leak_emit(cx, PC_LEAK_VOID, 0,0,0, 0,0,0, 0,0,0, 0, nullptr); return 2; }
				if (how == PC_END_ERROR) { pc_leak_emit(cx, PC_LEAK_VOID, 0,0,0, 0,0,0, 0,0,0, 0, nullptr); ph.rc = -1; return -1; }
				if (have_final >= 2) {
					/* src/polycap-photon.c:808-870: a photon launched inside the glass ends as a leak event itself */
					if (have_final == 2 && (how == PC_END_CALLS || how == PC_END_EXIT)) {
						const double t = (T.z[nmax]-ph.Pz)/ph.dz;
						ph.Px = ph.Px + ph.dx * t; ph.Py = ph.Py + ph.dy * t; ph.Pz = ph.Pz + ph.dz * t;
						const int inside = (T.ext[nmax] <= 0.) ? -1 : (pc_outside_hex(T.ext[nmax], ph.Px, ph.Py) ? 0 : 1);
						if (inside == 0) pc_leak_emit(cx, PC_LEAK_EXT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
						else if (inside == 1) pc_leak_emit(cx, PC_LEAK_INT, ph.Px, ph.Py, ph.Pz, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, w);
					}
					for (int e = 0; e < ne; e++) w[e] = 0.;
					ph.Px = T.ext[nmax]+1.; ph.Py = T.ext[nmax]+1.; ph.Pz = T.z[nmax];
					ph.dx = sdx; ph.dy = sdy; ph.dz = sdz;
					ph.rc = 1;
					return 1;
				}
				ph.rc = (how == PC_END_ABSORBED) ? 0 : 1;
				return ph.rc;
			}
			/* a leaked fraction has ended: src/polycap-capil.c:810-880 */
			if (how == PC_END_ERROR) {
				/* reflect returns -2 -> the parent's trace returns -1 -> ... : the whole launch fails */
				lvl = 0;
				continue;
			}
			double cxl = 0., cyl = 0., czl = 0.;
			int final_kind = -2;
			if (how == PC_END_CALLS || how == PC_END_EXIT) {
				const double t = (T.z[nmax]-ph.Pz)/ph.dz;
				cxl = ph.Px + ph.dx * t; cyl = ph.Py + ph.dy * t; czl = ph.Pz + ph.dz * t;
				const int inside = (T.ext[nmax] <= 0.) ? -1 : (pc_outside_hex(T.ext[nmax], cxl, cyl) ? 0 : 1);
				final_kind = (inside == 0) ? PC_LEAK_EXT : ((inside == 1) ? PC_LEAK_INT : -2);
			}
			const double *wchild = w;
			lvl--;
			int keep;
			pc_frame_load(cx.frames + (long)lvl*fstride, ph, h, calls, keep, entrance);
			w = cx.frames + (long)lvl*fstride + PC_LF_HDR;
			ph.wmem = w;
			/* its last state is one more event, with THIS photon's direction, electric vector and reflection count */
			if (final_kind != -2)
				pc_leak_emit(cx, final_kind, cxl, cyl, czl, ph.dx, ph.dy, ph.dz, ph.ex, ph.ey, ph.ez, ph.irefl, wchild);
			/* resume the suspended reflection where it stopped */
			if (entrance) { how = PC_END_ABSORBED; entrance = 0; continue; }
			if (keep == 0) { how = PC_END_ABSORBED; continue; }
			ph.dx = fma(-2.0*h.cosalfa, h.nx, ph.dx);
			ph.dy = fma(-2.0*h.cosalfa, h.ny, ph.dy);
			ph.dz = fma(-2.0*h.cosalfa, h.nz, ph.dz);
			ph.irefl++;
			ph.i = h.ix;
			ph.rc = 0;
			pc_trace_begin(ph);
			st = PC_ST_MARCH;
			break;
		}
	}
}

#endif /* PC_LEAK_H */
