/*
 * pc_device.h -- per-photon device logic of the MI355X trace path (fp64).
 *
 * One lane owns one photon.  The logic is written as small state-transition functions so the
 * kernels (pc_kernels.hip) can schedule the three phases of a photon's life wave-wide:
 *   MARCH  : walk profile nodes with a 6-FMA "still strictly inside the capillary" certificate
 *   EVENT  : the reference's full segment quadratic + wall hit + Fresnel reflection
 *   NEW    : source sampling (Philox4x32-10) + entrance tests
 *
 * Reference functions restated here (reference v1.2, file:line):
 *   polycap_source_get_photon          src/polycap-source.c:23-144      -> pc_sample_photon
 *   polycap_photon_launch              src/polycap-photon.c:390-955     -> pc_launch_init + kernel loop
 *   polycap_capil_trace                src/polycap-capil.c:1197-1361    -> pc_march_ok / pc_event
 *   polycap_capil_segment              src/polycap-capil.c:52-255       -> pc_segment
 *   polycap_capil_reflect              src/polycap-capil.c:565-655      -> pc_reflect
 *   polycap_refl_polar                 src/polycap-capil.c:444-563      -> pc_reflect (energy loop)
 *   polycap_photon_within_pc_boundary  src/polycap-photon.c:139-169     -> pc_outside_hex
 *
 * The functions are PC_HD so that tests can also compile this header as host code and drive it
 * photon by photon (tests/emul/, never part of libpolycap).
 */
#ifndef PC_DEVICE_H
#define PC_DEVICE_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define PC_HD __host__ __device__ __forceinline__
#else
#define PC_HD inline
#endif

#define PC_COSPI_6 0.86602540378443864676
#define PC_PI 3.14159265358979323846

/* photon life-cycle states (per lane) */
enum { PC_ST_IDLE = 0, PC_ST_NEW = 1, PC_ST_MARCH = 2, PC_ST_EVENT = 3, PC_ST_DONE = 4 };

/* per-energy constants, precomputed on the host from (E, density, scatf, amu):
 * n = (1-alfa) + i*beta  (src/polycap-capil.c:497-499); ninv2 = (1/n)^2; rough_c = 1.01358*E*sig_rough */
struct pc_energy_const {
	double n_re, n_im;
	double ninv2_re, ninv2_im;
	double rough_c;
	double valid;   /* 0 -> polycap_refl_polar would reject its arguments (returns -1) */
	/* FORM 3 (pc_fresnel3: the weight sweeps of runs whose weights live in memory): n^2 = n2_re + i n2_im, d2 = 1 - n2_re
	 * = alfa (2 - alfa) + beta^2 without the cancellation, zi2 = max(n2_im^2, 2^-200), rough_k2 = rough_c^2 */
	double d2, n2_re, n2_im, zi2, rough_k2;
};

struct pc_params {
	int nmax;
	int mono;           /* n_shells == 0 */
	int n_energies;
	int literal;        /* 1: no certificate, every segment takes the full quadratic */
	int uniform_illum;  /* src_sigx < 0 || src_sigy < 0 */
	int generic_src;    /* src_x != src_y: elliptical source, libm sampling path */
	int form3;          /* the run's weights live in memory (more than 8 energies, or several on a profile of more than 1024 points):
	                     * its reflections evaluate the Fresnel factor in FORM 3 (pc_fresnel3) */
	double n_shells;
	double hexscale;    /* 2*cos(pi/6)*(n_shells+1) */
	double inv_hexscale; /* 1/hexscale, for certificates (outcomes divide by hexscale like the reference) */
	double adj;         /* certificate margin: see pc_march_ok */
	double two_rmax;    /* 2 * max_i cap[i] (block certificates) */
	float adjf, two_rmaxf;  /* the same two, rounded up to float and inflated by PC_MARGIN_INFLATE: pc_march_ok decides in float */
	double bnd_thresh;  /* max_i cap[i]/ext[i] (+slack): capillaries closer than this to the hexagon edge are "boundary" */
	double z_end, ext_end;
	double d_source, src_x, src_y, src_sigx, src_sigy, src_shiftx, src_shifty, frac_hor_pol;
	double cap0, ext0;
};

/* block-certificate margins of one start node (pc_march_ok): base and chord deviation of zh for strides PC_L1 and PC_L2;
 * +inf where the stride does not fit before the end of the profile */
/* Block-certificate data of one start node, one 16-byte read per march step: mb12 = margin bases of strides PC_L1 (high half)
 * and PC_L2 (low half) as the upper 16 bits of a float, rounded up; md1, md2 = chord deviations of zh over the two blocks;
 * r2 = twice the largest capillary radius of the PC_L2 block (which contains the PC_L1 block).  All rounded up and, except the
 * deviations, inflated by PC_MARGIN_INFLATE at build time (pc_problem.h). */
struct pc_marg4 { unsigned int mb12; float md1, md2, r2; };
/* leak path: chord deviations of cap over the PC_L1 / PC_L2 block that starts at a node, rounded up (infinite where the block
 * does not fit) */
struct pc_drdev { float d1, d2; };

PC_HD float pc_bits_as_float(unsigned int u) { return __builtin_bit_cast(float, u); }

/* profile tables.  z/cap/zh/cap2 are the MARCH tables (LDS on the device), ext is only read on events.  Every kernel sets the
 * tables its path reads; the rest stay null (a forgotten table then faults in the host compile of the tests instead of reading
 * whatever the register held). */
struct pc_tables {
	const double *z = nullptr;
	const double *cap = nullptr;
	const double *zh = nullptr;    /* ext[i] / hexscale: capillary axis = (kx, ky) * zh[i]  (src/polycap-photon.c:624-627) */
	const double *cap2 = nullptr;  /* cap[i]^2 */
	const double *hexd = nullptr;  /* sqrt(ext^2 - (ext/2)^2): centre-to-edge distance of the outer hexagon at node i */
	const double *idz = nullptr;   /* 1 / (z[i+1] - z[i]) */
	const double *ext = nullptr;
	const double *stp = nullptr;   /* leak path: cap[i] / 10, the step of the wall search in segment i (src/polycap-capil.c:1019) */
	const double *istp = nullptr;  /* leak path: 10 / cap[i] (candidates for step counts only) */
	const struct pc_drdev *dr = nullptr;   /* leak path: chord deviations of cap */
	/* block certificates (see pc_march_ok): for stride L1 / L2, margin base and chord deviation of zh, rounded up */
	const struct pc_marg4 *mg = nullptr;   /* margin base and chord deviation of zh of both strides, packed per start node: one 16-byte read */
};

#ifndef PC_L1
#define PC_L1 5     /* strides of the two block-certificate levels, in segments (flights between two reflections span 23
                     * segments on average, half of them fewer than 8: scripts/analysis/flight_stats.py, scripts/ab_strides.sh) */
#endif
#ifndef PC_L2
#define PC_L2 25
#endif
#define PC_MARGIN_INFLATE 1.0000038f   /* 1 + 2^-18 */
#ifndef PC_LV_LATER
#ifndef PC_CREEP
#define PC_CREEP 1     /* a failed probe at stride PC_L1 goes to the EVENT phase, which walks the last segments itself (pc_event_pre) */
#endif
#define PC_LV_LATER 2  /* widest level a flight after a reflection starts with */
#endif

template <int NE>
struct pc_photon {
	double Px, Py, Pz;      /* exit_coords: last interaction point */
	double dx, dy, dz;      /* exit_direction (unit) */
	double ex, ey, ez;      /* exit_electric_vector */
	double kx, ky;          /* capillary axis scale factors */
	double sx, sy, ox, oy;  /* ray of the current trace call: p(z) = o + s*z */
	double idzd;            /* 1 / dz of the current direction */
	double C0;              /* |p - axis|^2 - cap^2 at node i (certificate chain) */
	double dtravel;
	double kn;              /* |(kx, ky)| */
	double w[NE > 0 ? NE : 1]; /* NE > 0: one weight per energy in registers */
	double *wmem;              /* NE == 0: n_energies weights in memory at wmem[e*wstride], valid once wset != 0 */
	long wstride;
	int i;                  /* segment [z_i, z_i+1] to be visited next */
	int irefl;
	int first;              /* segment i is the first of a trace call (last hit lies inside it) */
	int bnd;                /* boundary capillary (or mono-capillary): hexagon tests are done at every node */
	int wset;               /* NE == 0: the in-memory weights have been written (before that every weight is 1) */
	int lv;                 /* widest block-certificate stride still allowed on this flight: 0 single segments, 1 L1, 2 L2 */
	int rc;                 /* final polycap_photon_launch return code once DONE */
	int qr;                 /* hexagonal (q, r) indices of the capillary, (q + 32768) << 16 | (r + 32768): kx, ky, kn follow from them */
};

/* ------------------------------------------------------------------ small helpers */

PC_HD void pc_norm3(double &x, double &y, double &z)
{
	double inv = 1.0 / sqrt(x*x + y*y + z*z);
	x *= inv; y *= inv; z *= inv;
}

/* hexagon test of polycap_photon_within_pc_boundary given the centre-to-edge distance d;
 * returns 1 when OUTSIDE (NaN coordinates compare false -> inside, as in the reference) */
PC_HD int pc_outside_hexd(double d, double x, double y)
{
	double dp1 = fabs(y);
	double dp2 = fabs(PC_COSPI_6*x + 0.5*y);
	double dp3 = fabs(PC_COSPI_6*x - 0.5*y);
	return (dp1 > d || dp2 > d || dp3 > d) ? 1 : 0;
}

/* same from the circum-radius (radius <= 0 -> the reference returns -1, which its callers treat like "inside") */
PC_HD int pc_outside_hex(double radius, double x, double y)
{
	if (radius <= 0.) return 0;
	double half = radius * 0.5;
	return pc_outside_hexd(sqrt(radius*radius - half*half), x, y);
}

/* ------------------------------------------------------------------ Philox4x32-10 */

PC_HD void pc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
	for (int r = 0; r < 10; r++) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c0;
		uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		uint32_t n1 = (uint32_t)p1;
		uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		uint32_t n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream (seed, slot, attempt): uniform #d, two 53-bit uniforms per Philox block */
struct pc_rng {
	uint64_t seed, slot;
	uint32_t attempt, d;
	uint32_t buf[4];
};

PC_HD void pc_rng_init(pc_rng &g, uint64_t seed, uint64_t slot, uint32_t attempt)
{
	g.seed = seed; g.slot = slot; g.attempt = attempt; g.d = 0;
}

PC_HD double pc_rng_uniform(pc_rng &g)
{
	uint32_t d = g.d++;
	if ((d & 1u) == 0u)
		pc_philox4x32_10((uint32_t)g.slot, (uint32_t)(g.slot >> 32), g.attempt, d >> 1,
		                 (uint32_t)g.seed, (uint32_t)(g.seed >> 32), g.buf);
	uint64_t w = (d & 1u) ? (((uint64_t)g.buf[3] << 32) | g.buf[2]) : (((uint64_t)g.buf[1] << 32) | g.buf[0]);
	return (double)(w >> 11) * (1.0/9007199254740992.0);
}

/* ------------------------------------------------------------------ source sampling */

struct pc_start {
	double x, y, z;        /* start_coords */
	double dx, dy, dz;     /* start_direction (unit) */
	double ex, ey, ez;     /* start_electric_vector (unit, perpendicular to the direction) */
	double srcx, srcy;     /* src_start_coords */
};

/* sin and cos on [0, pi/2] without the generic argument reduction (fdlibm-style kernels on [0, pi/4],
 * Cody-Waite pi/2 = hi + lo); < 1 ulp.  Keeps libm's large-argument paths out of the trace kernel. */
PC_HD void pc_sincos_quadrant(double x, double &sn, double &cs)
{
	const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
	const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
	             S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
	const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
	             C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
	const int swap = x > 0.78539816339744830962;
	const double y = swap ? ((pio2_hi - x) + pio2_lo) : x;
	const double z = y*y;
	const double ps = S1 + z*(S2 + z*(S3 + z*(S4 + z*(S5 + z*S6))));
	const double pcs = C1 + z*(C2 + z*(C3 + z*(C4 + z*(C5 + z*C6))));
	const double s = y + y*z*ps;
	const double hz = 0.5*z;
	const double c = (1.0 - hz) + z*z*pcs;
	sn = swap ? c : s;
	cs = swap ? s : c;
}

/* src/polycap-source.c:23-144.  GENERIC=false is the circular source (src_x == src_y), where
 * atan(src_y/src_x*tan(x)) == x up to rounding; GENERIC=true evaluates the elliptical formula with libm. */
template <bool GENERIC>
PC_HD void pc_sample_photon(const pc_params &Pm, uint64_t seed, uint64_t slot, uint32_t attempt, pc_start &s)
{
	pc_rng g;
	pc_rng_init(g, seed, slot, attempt);
	double r = pc_rng_uniform(g);
	double cphi, sphi;
	if (GENERIC) {
		double phi = atan(Pm.src_y/Pm.src_x * tan(2.0*PC_PI*r/4.));
		r = pc_rng_uniform(g);
		if ((r >= 0.25) && (r < 0.5)) phi = PC_PI - phi;
		if ((r >= 0.5) && (r < 0.75)) phi = PC_PI + phi;
		if (r >= 0.75) phi = -1.0 * phi;
		cphi = cos(phi); sphi = sin(phi);
	} else {
		double phi = 2.0*PC_PI*r/4.;
		pc_sincos_quadrant(phi, sphi, cphi);
		r = pc_rng_uniform(g);
		/* phi -> pi-phi, pi+phi, -phi : :57-62 */
		if ((r >= 0.25) && (r < 0.75)) cphi = -cphi;
		if (r >= 0.5) sphi = -sphi;
	}
	double max_rad = Pm.src_x*Pm.src_y / sqrt((Pm.src_y*cphi)*(Pm.src_y*cphi) + (Pm.src_x*sphi)*(Pm.src_x*sphi));
	r = pc_rng_uniform(g);
	double sr = sqrt(r);
	s.srcx = sr * max_rad * cphi + Pm.src_shiftx;
	s.srcy = sr * max_rad * sphi + Pm.src_shifty;
	if (Pm.uniform_illum) {
		/* :74-96 */
		if (Pm.mono) {
			r = pc_rng_uniform(g);
			s.x = (2.*r-1.) * Pm.cap0;
			r = pc_rng_uniform(g);
			s.y = (2.*r-1.) * Pm.cap0;
		} else {
			int outside;
			do {
				r = pc_rng_uniform(g);
				s.x = (2.*r-1.) * Pm.ext0;
				r = pc_rng_uniform(g);
				s.y = (2.*r-1.) * Pm.ext0;
				outside = pc_outside_hex(Pm.ext0, s.x, s.y);
			} while (outside && g.d < 4096u); /* the reference loops unbounded; acceptance is ~65 % per try */
		}
		s.dx = s.x - s.srcx;
		s.dy = s.y - s.srcy;
		s.dz = Pm.d_source;
	} else {
		/* :97-108 */
		r = pc_rng_uniform(g);
		s.dx = Pm.src_sigx * (1.-2.*fabs(r));
		r = pc_rng_uniform(g);
		s.dy = Pm.src_sigy * (1.-2.*fabs(r));
		s.dz = 1.;
		s.x = s.srcx + s.dx * Pm.d_source / s.dz;
		s.y = s.srcy + s.dy * Pm.d_source / s.dz;
	}
	s.z = 0.;
	pc_norm3(s.dx, s.dy, s.dz);
	/* :114-137 polarisation */
	r = pc_rng_uniform(g);
	double e0x, e0y;
	if (fabs(r) <= Pm.frac_hor_pol) { e0x = 1.; e0y = 0.; } else { e0x = 0.; e0y = 1.; }
	double cosalpha = e0x*s.dx + e0y*s.dy;
	/* c_ae = 1/sin(acos(c)), c_be = -c_ae*c */
	double c_ae = 1.0 / sqrt(1.0 - cosalpha*cosalpha);
	double c_be = -1.*c_ae*cosalpha;
	s.ex = e0x * c_ae + s.dx * c_be;
	s.ey = e0y * c_ae + s.dy * c_be;
	s.ez = s.dz * c_be;
	pc_norm3(s.ex, s.ey, s.ez);
}

/* ------------------------------------------------------------------ launch entrance tests */

PC_HD int pc_last_node_le(const pc_tables &T, int upto, double zval)
{
	int idx = 0;
	for (int j = 0; j < upto; j++)
		if (T.z[j] <= zval) idx = j;
	return idx;
}

/* capillary axis scale factors from the hexagonal indices: src/polycap-photon.c:624-627 */
template <int NE>
PC_HD void pc_axis_setup(pc_photon<NE> &ph, double q_i, double r_i)
{
	ph.ky = r_i * (3./2);
	ph.kx = (2.*q_i + r_i) * PC_COSPI_6;
	ph.kn = sqrt(ph.kx*ph.kx + ph.ky*ph.ky);
}

/* begin a polycap_capil_trace call at segment ph.i: src/polycap-capil.c:1236-1243 */
template <int NE>
PC_HD void pc_ray_setup(pc_photon<NE> &ph)
{
	double idz = 1.0 / ph.dz;
	ph.idzd = idz;
	ph.sx = ph.dx * idz;
	ph.sy = ph.dy * idz;
	ph.ox = ph.Px - ph.sx * ph.Pz;
	ph.oy = ph.Py - ph.sy * ph.Pz;
}

template <int NE>
PC_HD void pc_trace_begin(pc_photon<NE> &ph)
{
	pc_ray_setup(ph);
	ph.first = 1;
	/* long first flight: start with the widest stride; after a reflection flights are short */
	ph.lv = ph.bnd ? 0 : ((ph.irefl == 0) ? 2 : PC_LV_LATER);
}

/* src/polycap-photon.c:458-645 + 888-906.  Returns PC_ST_MARCH when the photon entered a capillary,
 * else PC_ST_DONE with ph.rc in {2, -2}. */
template <int NE>
PC_HD int pc_launch_init(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph,
                         double x, double y, double z, double dx, double dy, double dz,
                         double ex, double ey, double ez)
{
	const int nmax = Pm.nmax;
	pc_norm3(dx, dy, dz);
	ph.Px = x; ph.Py = y; ph.Pz = z;
	ph.dx = dx; ph.dy = dy; ph.dz = dz;
	/* polycap_refl_polar normalises the electric vector in place before its first use (src/polycap-capil.c:492-494);
	 * the reflection algebra below relies on |E| = 1, so it is done here */
	{
		const double en = sqrt(ex*ex + ey*ey + ez*ez);
		if (en != 1.) { ex /= en; ey /= en; ez /= en; }
	}
	ph.ex = ex; ph.ey = ey; ph.ez = ez;
	ph.irefl = 0; ph.dtravel = 0.; ph.rc = 0; ph.C0 = 0.; ph.bnd = 0; ph.wset = 0;
	/* NE == 0: weights live in memory and are initialised lazily (a photon without reflections has weight 1) */
	if (NE > 0) {
#pragma unroll
		for (int e = 0; e < (NE > 0 ? NE : 1); e++) ph.w[e] = 1.;
	}

	/* :507-512 */
	int z_id = (z > 0) ? pc_last_node_le(T, nmax, z) : 0;
	double cur_ext = ((T.ext[z_id] - T.ext[z_id+1]) / (T.z[z_id] - T.z[z_id+1])) * (z - T.z[z_id]) + T.ext[z_id];
	double q_i = 0., r_i = 0.;
	if (Pm.mono) {
		if (sqrt(x*x + y*y) > cur_ext) { ph.rc = -2; return PC_ST_DONE; }
	} else {
		/* :540-552 axial hex coordinates + cube rounding */
		double zz = cur_ext / Pm.hexscale;
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
		if (pc_outside_hex(cur_ext, x, y)) { ph.rc = -2; return PC_ST_DONE; }
	}
	/* :624-627 */
	pc_axis_setup(ph, q_i, r_i);
	ph.qr = (((int)q_i + 32768) << 16) | ((int)r_i + 32768);
	ph.i = (z > 0) ? pc_last_node_le(T, nmax + 1, z) : 0;
	/* :629-645 */
	double cur_rad, cur_cx, cur_cy;
	if (z > 0) {
		double dzseg = T.z[z_id+1] - T.z[z_id];
		double cx0 = ph.kx*T.zh[z_id], cx1 = ph.kx*T.zh[z_id+1];
		double cy0 = ph.ky*T.zh[z_id], cy1 = ph.ky*T.zh[z_id+1];
		cur_rad = ((T.cap[z_id+1] - T.cap[z_id])/dzseg) * (z - T.z[z_id]) + T.cap[z_id];
		cur_cx = ((cx1 - cx0)/dzseg) * (z - T.z[z_id]) + cx0;
		cur_cy = ((cy1 - cy0)/dzseg) * (z - T.z[z_id]) + cy0;
	} else {
		cur_rad = T.cap[0];
		cur_cx = ph.kx*T.zh[0];
		cur_cy = ph.ky*T.zh[0];
	}
	double d_ph_capcen = sqrt((x-cur_cx)*(x-cur_cx) + (y-cur_cy)*(y-cur_cy));
	if (d_ph_capcen > cur_rad) { ph.rc = 2; return PC_ST_DONE; }

	/* boundary capillary?  The capillary circle at node i stays inside the outer hexagon iff
	 *   hexd_i - max_j|n_j.axis_i| - cap_i > 0,  hexd_i = ext_i*sqrt(3)/2,  axis_i = (kx,ky)*ext_i/hexscale
	 * <=> sqrt(3)/2 - M/hexscale > cap_i/ext_i  with  M = max_j |n_j.(kx,ky)|.  bnd_thresh = max_i cap_i/ext_i (+slack). */
	if (!Pm.mono) {
		double m1 = fabs(ph.ky);
		double m2 = fabs(PC_COSPI_6*ph.kx + 0.5*ph.ky);
		double m3 = fabs(PC_COSPI_6*ph.kx - 0.5*ph.ky);
		double M = fmax(m1, fmax(m2, m3));
		ph.bnd = (PC_COSPI_6 - M/Pm.hexscale > Pm.bnd_thresh) ? 0 : 1;
	} else {
		ph.bnd = 1; /* mono-capillary: the reference's hexagon tests still run (against ext); keep them literal */
	}
	pc_trace_begin(ph);
	return PC_ST_MARCH;
}

/* ------------------------------------------------------------------ MARCH certificate
 *
 * Within segment i both the ray-to-axis offset q(u) and the capillary radius R(u) are linear in
 * u=(z-z_i)/(z_i+1-z_i), so g(u) = |q(u)|^2 - R(u)^2 = A u^2 + B u + C with A = |dq|^2 - dR^2 >= -dR^2.
 * A quadratic deviates from its chord by at most |A|/4 on an interval of length <= 1, hence
 *     g(u) <= max(g(u0), g(1)) + dR^2/4   for all u in [u0, 1].
 * If g at both ends is below -(dRmax^2/4 + m) the photon is strictly inside the capillary over that whole
 * stretch: the reference's quadratic (src/polycap-capil.c:119-157) has no admissible root there (it returns
 * -2/-3/-4/-5), the capillary lies inside the outer hexagon (non-boundary capillaries), so its hexagon tests
 * (src/polycap-capil.c:1263,1301) pass as well: the visit is a certain "miss" and is skipped.
 * adj = dRmax^2/4 + m with m ~ 1e6 x the rounding error of g.  Anything else goes to pc_event().
 * Cost: 6 FMA + 3 LDS reads per node.
 */
template <int NE>
PC_HD int pc_march_ok(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph)
{
	/* Block certificate (stride L > 1).  Over nodes i..i+L let zh_c, R_c be the chords of zh and cap between the two
	 * end nodes, Dzh and DR the largest deviations of the (piecewise linear) tables from those chords, dR = cap[i+L]-cap[i].
	 * With q = q_c - k*dev_zh and R = R_c + dev_R:
	 *     g <= g_c + 2 R_blk (|k| Dzh + DR) + (|k| Dzh)^2,     g_c <= max(g(z_i), g(z_i+L)) + dR^2/4
	 * (|q_c| < R_c <= R_blk wherever g_c < 0, R_blk = the largest radius of the block: near the narrow end of a tapered optic
	 * a third of the global maximum), so g < 0 on all L segments when both end values are below
	 * -(mb + |k| md (2 R_blk + |k| md)), with mb = dR^2/4 + 2 R_blk DR + m, md = Dzh and 2 R_blk tabulated per start node
	 * (pc_marg4, rounded up).  L = 1 is the plain single-segment certificate with margin adj. */
	/* ph.lv caps the stride for this flight (lowered after a failed probe); take the widest stride whose margin the
	 * current node already satisfies */
	const int cap = ph.lv;
	const int i0 = ph.i;
	int lv = 0;
	/* The certificate VALUES are fp64 (they decide nothing by themselves: a node that is not certified is visited
	 * literally); the comparison against the margins is made in single precision with every margin inflated by
	 * PC_MARGIN_INFLATE (2^-18, far above the three float roundings involved), so that "Cf < -mf" implies C < -m for the
	 * exact values.  Margins of both strides are then 3 float operations each instead of 2 conversions + 3 fp64 operations. */
	float marg = Pm.adjf;
	const float C0f = (float)ph.C0;
	{
		/* a stride that does not fit before the end of the profile has an infinite margin: no index test needed */
		const pc_marg4 g = T.mg[i0];
		const float knf = (float)ph.kn * PC_MARGIN_INFLATE;
		const float kd1 = knf * g.md1, kd2 = knf * g.md2;
		const float m1 = fmaf(kd1, g.r2 + kd1, pc_bits_as_float(g.mb12 & 0xffff0000u));
		const float m2 = fmaf(kd2, g.r2 + kd2, pc_bits_as_float(g.mb12 << 16));
		if (cap >= 1 && C0f < -m1) { lv = 1; marg = m1; }
		if (cap >= 2 && C0f < -m2) { lv = 2; marg = m2; }
	}
	const int L = (lv == 0) ? 1 : ((lv == 1) ? PC_L1 : PC_L2);
	const int i1 = i0 + L;
	double z1 = T.z[i1], zh1 = T.zh[i1], c2 = T.cap2[i1];
	double qx = fma(-ph.kx, zh1, fma(ph.sx, z1, ph.ox));
	double qy = fma(-ph.ky, zh1, fma(ph.sy, z1, ph.oy));
	double C1 = fma(qx, qx, fma(qy, qy, -c2));
	int ok = (C0f < -marg) & ((float)C1 < -marg);
	if (ph.bnd) {
		/* boundary capillary (or mono-capillary; always level 0): the hexagon tests of the visit are not implied, do them:
		 * axis at both nodes (src/polycap-capil.c:1263) and the ray at z_i (:1296-1308) */
		double z0 = T.z[i0], zh0 = T.zh[i0], h0 = T.hexd[i0], h1 = T.hexd[i1];
		double px = fma(ph.sx, z0, ph.ox), py = fma(ph.sy, z0, ph.oy);
		ok &= !pc_outside_hexd(h0, ph.kx*zh0, ph.ky*zh0);
		ok &= !pc_outside_hexd(h1, ph.kx*zh1, ph.ky*zh1);
		ok &= !pc_outside_hexd(h0, px, py);
	}
	if (ok) { ph.C0 = C1; ph.i = i1; return 1; }
	if (lv > 0) {
#if PC_CREEP
		/* The node PC_L1 segments ahead is not certified: the wall is (nearly always) within those segments.  Walking up to it
		 * one certified segment per step costs 1.8 steps per flight plus this one; pc_event_pre does that walk in straight-line
		 * code in front of its literal visit instead (ph.lv == 3 asks for it). */
		if (lv == 1) { ph.lv = 3; return 0; }
#endif
		ph.lv = lv - 1;                             /* far node not certified: shorter strides for the rest of this flight */
		return 1;
	}
	return 0;
}

/* First segment of a trace call: the last interaction point P lies inside [z_i, z_i+1] and the reference only
 * admits roots beyond P.z + 1e-5 (src/polycap-capil.c:134-171).  The certificate is evaluated on
 * [z_lo, z_i+1], z_lo = max(z_i, P.z + 1e-5), with zh and cap interpolated linearly to z_lo.  The reference
 * also tests the ray at z_i -- behind the photon -- against the outer hexagon (:1296-1308); that point can lie
 * just outside the capillary, so the test is done explicitly here. */
template <int NE>
PC_HD int pc_march_first_ok(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph)
{
	const int i0 = ph.i, i1 = ph.i + 1;
	double z0 = T.z[i0], z1 = T.z[i1], zh0 = T.zh[i0], zh1 = T.zh[i1];
	double R0 = T.cap[i0], R1 = T.cap[i1], h0 = T.hexd[i0];
	double qx1 = fma(-ph.kx, zh1, fma(ph.sx, z1, ph.ox));
	double qy1 = fma(-ph.ky, zh1, fma(ph.sy, z1, ph.oy));
	double C1 = fma(qx1, qx1, fma(qy1, qy1, -R1*R1));
	double zlo = fmax(z0, ph.Pz + 1.e-5);
	int ok;
	if (zlo >= z1) {
		ok = 1;                     /* no admissible root can lie in this segment */
	} else {
		double t = (zlo - z0) * T.idz[i0];
		double zhl = fma(t, zh1 - zh0, zh0);
		double Rl = fma(t, R1 - R0, R0);
		double qx = fma(-ph.kx, zhl, fma(ph.sx, zlo, ph.ox));
		double qy = fma(-ph.ky, zhl, fma(ph.sy, zlo, ph.oy));
		double Cl = fma(qx, qx, fma(qy, qy, -Rl*Rl));
		ok = (Cl < -Pm.adj) & (C1 < -Pm.adj);
	}
	/* ray at z_i inside the optic; boundary capillaries also test the axis at both nodes */
	double px = fma(ph.sx, z0, ph.ox), py = fma(ph.sy, z0, ph.oy);
	ok &= !(h0 > 0. && pc_outside_hexd(h0, px, py));
	if (ph.bnd) {
		double h1 = T.hexd[i1];
		ok &= !pc_outside_hexd(h0, ph.kx*zh0, ph.ky*zh0);
		ok &= !pc_outside_hexd(h1, ph.kx*zh1, ph.ky*zh1);
	}
	ok &= (ph.dz >= 0.);            /* backwards-flying photons keep the reference's literal path */
	if (ok) { ph.C0 = C1; ph.i = i1; ph.first = 0; }
	return ok;
}

/* certificate value at node idx for the current ray */
template <int NE>
PC_HD double pc_node_C(const pc_tables &T, const pc_photon<NE> &ph, int idx)
{
	double z1 = T.z[idx], zh1 = T.zh[idx], c2 = T.cap2[idx];
	double qx = fma(-ph.kx, zh1, fma(ph.sx, z1, ph.ox));
	double qy = fma(-ph.ky, zh1, fma(ph.sy, z1, ph.oy));
	return fma(qx, qx, fma(qy, qy, -c2));
}

/* ------------------------------------------------------------------ full segment (reference quadratic)
 * src/polycap-capil.c:52-255.  Returns the reference's status; on 1, (hx,hy,hz) is the hit and (nx,ny,nz)
 * the unit surface normal.  (p0x,p0y) = ray at z_i (phot_coord0 of src/polycap-capil.c:1256-1258).
 * Same quadratic, same root selection and guards as the reference; divisions by per-trace / per-segment
 * constants are hoisted (1/dz of the direction, 1/(z_i+1 - z_i) from the table) and the normal is normalised
 * once instead of four times. */
template <int NE>
PC_HD int pc_segment(const pc_tables &T, const pc_photon<NE> &ph, int i,
                     double &p0x, double &p0y, double &hx, double &hy, double &hz,
                     double &nx, double &ny, double &nz)
{
	double z0 = T.z[i], z1 = T.z[i+1];
	double R0 = T.cap[i], R1 = T.cap[i+1];
	double c0x = ph.kx*T.zh[i], c0y = ph.ky*T.zh[i];
	double c1x = ph.kx*T.zh[i+1], c1y = ph.ky*T.zh[i+1];
	double t0 = z0 - ph.Pz;
	p0x = fma(ph.sx, t0, ph.Px);
	p0y = fma(ph.sy, t0, ph.Py);
	nx = 0.; ny = 0.; nz = 0.;
	hx = hy = hz = 0.;
	if (ph.dz < 0) return -1;          /* :85-88 */
	double cdx = c1x - c0x, cdy = c1y - c0y, cdz = z1 - z0;
	double icdz = T.idz[i];
	double ddx = fma(-cdx, icdz, ph.sx);
	double ddy = fma(-cdy, icdz, ph.sy);
	double dR = R1 - R0;
	double rr = dR*icdz;
	double qx = p0x - c0x, qy = p0y - c0y;
	double a = fma(ddx, ddx, fma(ddy, ddy, -rr*rr));
	double b = 2.*fma(qx, ddx, fma(qy, ddy, -R0*rr));
	double c = fma(qx, qx, fma(qy, qy, -R0*R0));
	double discr = fma(b, b, -4.*a*c);
	if (discr < 0) return -2;
	double last = ph.Pz;
	const double i2a = 1.0/(2.*a);
	if (discr == 0) {
		hz = z0 + (-1.*b)*i2a;
	} else {
		double sq = sqrt(discr);
		double zr1 = z0 + (-1.*b + sq)*i2a;
		double zr2 = z0 + (-1.*b - sq)*i2a;
		/* written as negated "valid" tests so NaN behaves as in the reference (all comparisons false) */
		int bad1 = (zr1 < z0) || (zr1 - last < 1.e-5) || (zr1 > z1);
		int bad2 = (zr2 < z0) || (zr2 - last < 1.e-5) || (zr2 > z1);
		if (bad1) {
			if (bad2) return -3;
			hz = zr2;
		} else if (bad2) {
			hz = zr1;
		} else {
			hz = (zr2 - last < zr1 - last) ? zr2 : zr1;
		}
	}
	if (hz > z1) return -4;
	if (hz < z0 || hz - last < 1.e-5) return -5;
	double d_proj = (hz - z0) * ph.idzd;
	if (d_proj < 1.e-10) return -6;
	hx = fma(d_proj, ph.dx, p0x);
	hy = fma(d_proj, ph.dy, p0y);
	/* :225-246 surface normal: radial unit vector tilted by the wall angle gamma, tan(gamma) = (R0-R1)/|cap_dir| */
	double s1 = fma(qx, cdx, qy*cdy);                              /* (phot0-cap0).cap_dir, z component is 0 */
	double s2 = fma(ph.dx, cdx, fma(ph.dy, cdy, ph.dz*cdz));       /* photon_dir.cap_dir */
	double s3 = fma(cdx, cdx, fma(cdy, cdy, cdz*cdz));             /* |cap_dir|^2 */
	double is3 = 1.0 / s3;
	double tpar = fma(d_proj, s2, s1) * is3;
	double inx = hx - fma(tpar, cdx, c0x);
	double iny = hy - fma(tpar, cdy, c0y);
	double inz = hz - fma(tpar, cdz, z0);
	double idci = 1.0 / sqrt(fma(inx, inx, fma(iny, iny, inz*inz)));
	double tg = -dR * is3;                                          /* tan(gamma)/|cap_dir| */
	/* n ~ in/|in| + tan(gamma) * cap_dir/|cap_dir|, normalised once (= cos(gamma) in/|in| + sin(gamma) cap_dir/|cap_dir|
	 * of the reference up to its own final normalisation) */
	nx = fma(inx, idci, tg*cdx);
	ny = fma(iny, idci, tg*cdy);
	nz = fma(inz, idci, tg*cdz);
	/* |n|^2 = 1 + eps with eps ~ tan^2(gamma) + 2 tan(gamma) u.c: 1/sqrt(1+eps) by its series when eps is tiny
	 * (truncation < eps^4), by sqrt otherwise */
	double eps = fma(nx, nx, fma(ny, ny, nz*nz)) - 1.0;
	double f;
	if (fabs(eps) < 1e-4)
		f = fma(eps, fma(eps, fma(eps, -0.3125, 0.375), -0.5), 1.0);
	else
		f = 1.0 / sqrt(1.0 + eps);
	nx *= f; ny *= f; nz *= f;
	return 1;
}

/* ------------------------------------------------------------------ reflection
 * src/polycap-capil.c:565-655 with polycap_refl_polar (:444-563) inlined: the geometry of the s/p split is
 * energy independent and hoisted; per energy only the complex Fresnel amplitudes remain.
 * cos(theta)=n.d and sin^2(theta)=1-cos^2 replace cos/sin(acos(.)); |r|^2 = |num|^2/|den|^2 replaces the
 * complex reciprocal + cabs; |n x d| = sin(theta) normalises the s direction.  The reference's new electric
 * vector is |E_k| * f / |(|E| f)| for a common factor f (:546-559), i.e. the component-wise absolute value of
 * the (unit) vector.  Returns 1 keep, 0 absorbed, -1 error. */
/* energy-independent part of a reflection */
struct pc_refl_geom {
	double alfa;   /* cos(theta) = n.d, also the roughness argument (:599) */
	double st2;    /* sin^2(theta) */
	double es2;    /* (E . (n x d))^2   : frac_s = es2/sd2 */
	double ep2;    /* sd2 - es2         : frac_p = ep2/sd2 */
	double sd2;    /* |n x d|^2 */
};

/* returns -1 when the reference rejects the geometry (alfa < 0, :599-602), else 1 */
template <int NE>
PC_HD int pc_reflect_geom(const pc_photon<NE> &ph, double nx, double ny, double nz, pc_refl_geom &g)
{
	g.alfa = fma(ph.dx, nx, fma(ph.dy, ny, ph.dz*nz));
	if (g.alfa < 0.) return -1;
	g.st2 = fma(-g.alfa, g.alfa, 1.0);
	/* :520-537: s = (n x d)/|n x d|, |n x d|^2 = sin^2(theta); frac_s = (E.s)^2 */
	double sdx = fma(ny, ph.dz, -ph.dy*nz);
	double sdy = fma(nz, ph.dx, -ph.dz*nx);
	double sdz = fma(nx, ph.dy, -ph.dx*ny);
	double es = fma(ph.ex, sdx, fma(ph.ey, sdy, ph.ez*sdz));
	g.es2 = es*es;
	g.sd2 = fma(sdx, sdx, fma(sdy, sdy, sdz*sdz));
	g.ep2 = g.sd2 - g.es2;
	return 1;
}

/* Fresnel reflectivity rtot (polycap_refl_polar, src/polycap-capil.c:497-545) and roughness factor r_rough (:626-627) of
 * one energy.  Returns -1 on the reference's error exits, else 0. */
/* FORM 0: csq from (t, u = |wi|/(2t)) -- one more division, what single-energy runs use (their kernels are bound by
 * registers: the other form costs them 2-5 %).  FORM 1: rtot is a ratio of products that are homogeneous in (csr, csi, cos):
 * everything is carried multiplied by 2t, so that 2t*t = |w| + |wr| and 2t*u = |wi| need neither the division nor a second
 * use of the root -- what runs with several energies use (one fp64 division less per energy and reflection: 6-11 % of
 * their kernels).  The two differ in the last bits; which one a run uses depends on its number of energies only. */
template <int FORM>
PC_HD int pc_fresnel_f(const pc_energy_const &ec, const pc_refl_geom &g, double &rtot, double &r_rough)
{
	if (ec.valid == 0.) return -1;
	const double ct = g.alfa;
	/* tmp = n_inv^2 * sin^2 ; csq = csqrt(1 - tmp)   (:503-505) */
	double wr = fma(-ec.ninv2_re, g.st2, 1.0);
	double wi = -ec.ninv2_im*g.st2;
	/* principal complex square root without cancellation: t = sqrt((|w|+|wr|)/2), u = |wi|/(2t);
	 * (csr, |csi|) = (t, u) for wr >= 0 and (u, t) for wr < 0; the imaginary part takes the sign of wi */
	double mag = sqrt(fma(wr, wr, wi*wi));
	double q2 = mag + fabs(wr);                  /* 2 t^2 */
	double tt = sqrt(0.5*q2);
	double big, small, cts;
	if (FORM == 0) {
		big = tt;
		small = (tt > 0.) ? fabs(wi)/(2.*tt) : 0.;
		cts = ct;
	} else {
		const double sc = (tt > 0.) ? 2.*tt : 1.0;   /* the common factor (w == 0: csq = 0, q2 = |wi| = 0, any factor will do) */
		big = q2; small = fabs(wi); cts = ct*sc;
	}
	double csr = (wr >= 0.) ? big : small;
	double csi = copysign((wr >= 0.) ? small : big, wi);
	/* r_s = (cos - n*csq)/(cos + n*csq)   (:507-510) */
	double tr = fma(ec.n_re, csr, -ec.n_im*csi);
	double ti = fma(ec.n_re, csi, ec.n_im*csr);
	double nr = cts - tr, dr = cts + tr;
	double Ns = fma(nr, nr, ti*ti), Ds = fma(dr, dr, ti*ti);
	/* r_p = (csq - n*cos)/(csq + n*cos)   (:512-515) */
	double ur = ec.n_re*cts, ui = ec.n_im*cts;
	double pr = csr - ur, pi_ = csi - ui, er = csr + ur, ei = csi + ui;
	double Np = fma(pr, pr, pi_*pi_), Dp = fma(er, er, ei*ei);
	/* rtot = R_s frac_s + R_p frac_p = (es2 Ns Dp + ep2 Np Ds) / (sd2 Ds Dp): one division */
	rtot = fma(g.es2*Ns, Dp, g.ep2*Np*Ds) / (g.sd2*Ds*Dp);
	if (rtot < 0. || rtot > 1.) return -1;                      /* :633-637 */
	double cons1 = ec.rough_c*g.alfa;                           /* (1.01358*E)*alfa*sig_rough, :626 */
	r_rough = (ec.rough_c == 0.) ? 1.0 : exp(-1.*cons1*cons1);
	return 0;
}

/* `single`: the run has one energy (FORM 0), else FORM 1 */
PC_HD int pc_fresnel(const pc_energy_const &ec, const pc_refl_geom &g, double &rtot, double &r_rough, int single)
{
	return single ? pc_fresnel_f<0>(ec, g, rtot, r_rough) : pc_fresnel_f<1>(ec, g, rtot, r_rough);
}

/* one energy of src/polycap-capil.c:625-645: w *= rtot * r_rough.  Returns -1 on the reference's error exits,
 * else 1 when the new weight is still >= 1e-4, else 0. */
template <int FORM>
PC_HD int pc_reflect_energy_f(const pc_energy_const &ec, const pc_refl_geom &g, double &w)
{
	double rtot, r_rough;
	if (pc_fresnel_f<FORM>(ec, g, rtot, r_rough) < 0) return -1;
	w = w * rtot * r_rough;
	return (w >= 1.e-4) ? 1 : 0;
}

PC_HD int pc_reflect_energy(const pc_energy_const &ec, const pc_refl_geom &g, double &w, int single)
{
	return single ? pc_reflect_energy_f<0>(ec, g, w) : pc_reflect_energy_f<1>(ec, g, w);
}

/* ------------------------------------------------------------------ FORM 3: the sweeps of the any-n_energies kernels
 * Runs whose weights live in memory (more than 8 energies, or several on a long profile) spend their time in the Fresnel
 * factor: 291 energies x 21 reflections per started photon.  FORM 3 is the same reflectivity (polycap_refl_polar,
 * src/polycap-capil.c:497-545) written for that loop:
 *   - g = n csq = sqrt(n^2 - sin^2) is formed directly: z = g^2 = (cos^2 - d2) + i n2_im with d2 = 1 - Re n^2 kept as a
 *     constant of its own (no 1 - (1 + 2 delta)(1 - cos^2): the reference's cancellation is not reproduced, the result is
 *     closer to the exact value than the reference's own), Im z is a constant >= 0, so no sign handling and |z|^2 is one
 *     fma; the complex product n * csq of FORMs 0-2 disappears;
 *   - everything is carried multiplied by S = 2 max(Re g, Im g) = sqrt(2Q), Q = |z| + |Re z|: S g = (Q, Im z) or (Im z, Q);
 *   - r_s = (cos - g)/(cos + g), r_p = (g - n^2 cos)/(g + n^2 cos) (:507-515 multiplied by n);
 *   - the polarisation fractions fs = es2/sd2, fp = ep2/sd2 come per reflection (two IEEE divisions in the lane that
 *     reflects), so rtot = (fs Ns Dp + fp Np Ds)/(Ds Dp): one reciprocal;
 *   - on the device the two roots come from v_rsq_f64 and the quotient from v_rcp_f64 (2^-24, scripts/analysis/fp64_rates.hip)
 *     with one Newton step each (~4e-15 relative); |z|^2 >= zi2 >= 2^-200 keeps them finite.  The host compile evaluates
 *     the same expressions with IEEE sqrt and division.
 * 45 instructions per energy and reflection (FORM 2, round 3: 57).  The weights differ from FORMs 0/1 by the reference's own
 * rounding noise (~1e-10 relative near the critical angle, where 1 - sin^2/n^2 cancels); the trajectory does not depend on
 * them.  Callers guarantee ec.valid != 0 (runs with an invalid energy keep FORM 1). */
#if defined(__HIP_DEVICE_COMPILE__)
#define PC_FAST_MATH_DEVICE 1
#else
#define PC_FAST_MATH_DEVICE 0
#endif

/* sqrt(x), x > 0 and normal */
PC_HD double pc_sqrt_fast(double x)
{
#if PC_FAST_MATH_DEVICE
	const double y = __builtin_amdgcn_rsq(x);
	const double g = x*y, h = 0.5*y;
	const double r = fma(-h, g, 0.5);
	return fma(g, r, g);
#else
	return sqrt(x);
#endif
}

/* a / b */
PC_HD double pc_div_fast(double a, double b)
{
#if PC_FAST_MATH_DEVICE
	const double y = __builtin_amdgcn_rcp(b);
	const double e = fma(-b, y, 1.0);
	return a*fma(y, e, y);
#else
	return a / b;
#endif
}

/* exp(x) for x <= 0 (the roughness factor exp(-(c alfa)^2)): 2^k e^r with |r| <= ln2/2 and a degree-11 polynomial, 6e-15 */
PC_HD double pc_exp_neg_fast(double x)
{
#if PC_FAST_MATH_DEVICE
	const double k = rint(x*1.4426950408889634074);
	double r = fma(-k, 6.93147180369123816490e-01, x);
	r = fma(-k, 1.90821492927058770002e-10, r);
	double p = 2.50521083854417187751e-08;                 /* 1/11! */
	p = fma(p, r, 2.75573192239858906526e-07);
	p = fma(p, r, 2.75573192239858906526e-06);
	p = fma(p, r, 2.48015873015873015873e-05);
	p = fma(p, r, 1.98412698412698412698e-04);
	p = fma(p, r, 1.38888888888888888889e-03);
	p = fma(p, r, 8.33333333333333333333e-03);
	p = fma(p, r, 4.16666666666666666667e-02);
	p = fma(p, r, 1.66666666666666666667e-01);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return __builtin_amdgcn_ldexp(p, (int)k);                /* underflows to 0 by itself */
#else
	return exp(x);
#endif
}

/* what FORM 3 takes from a reflection's geometry: c2 = cos^2, fs = (E.s)^2 / |n x d|^2, fp = 1 - fs as the reference
 * forms it (frac_p from the p component: ep2 / sd2) */
PC_HD void pc_refl_geom3(const pc_refl_geom &g, double &c2, double &fs, double &fp)
{
	c2 = g.alfa*g.alfa;
	fs = g.es2/g.sd2;
	fp = g.ep2/g.sd2;
}
#define PC_SQRT2 1.41421356237309504880
/* cos(theta) sqrt(2): FORM 3 carries everything multiplied by S = sqrt(2Q) = sqrt(2) sqrt(Q); the constant factor goes to the
 * cosine once per reflection instead of to Q once per energy */
PC_HD double pc_refl_cr2(double c) { return c*PC_SQRT2; }

/* rtot of one energy: d2, n2r, n2i, zi2 from pc_energy_const; cr2 = cos(theta) sqrt(2) (pc_refl_cr2), c2, fs, fp from pc_refl_geom3 */
PC_HD double pc_fresnel3(double d2, double n2r, double n2i, double zi2, double cr2, double c2, double fs, double fp)
{
	const double zr = c2 - d2;
	const double mag = pc_sqrt_fast(fma(zr, zr, zi2));
	const double Q = mag + fabs(zr);
	const double cS = cr2*pc_sqrt_fast(Q);                   /* cos(theta) S, S = sqrt(2Q) */
	const bool up = zr >= 0.;
	const double Gr = up ? Q : n2i, Gi = up ? n2i : Q;
	const double nr = cS - Gr, dr = cS + Gr;
	const double Gi2 = Gi*Gi;
	const double Ns = fma(nr, nr, Gi2), Ds = fma(dr, dr, Gi2);
	const double A = n2r*cS, B = n2i*cS;
	const double pr = A - Gr, pi_ = B - Gi, er = A + Gr, ei = B + Gi;
	const double Np = fma(pr, pr, pi_*pi_), Dp = fma(er, er, ei*ei);
	return pc_div_fast(fma(fs*Ns, Dp, (fp*Np)*Ds), Ds*Dp);
}

/* FORM 3 with the polarisation weights as the geometry gives them (es2, ep2, sd2: no quotients of their own): what the
 * register-weight kernels of SOURCE runs multiply with (pc_reflect<NE, true>).  A photon's trajectory does not depend on its
 * weights, so these runs trace the same photons as with FORMs 0/1 -- counters and image planes bit for bit -- and their weights
 * differ from the IEEE forms by the reference's own rounding noise (1e-10 relative near the critical angle) and from the host
 * compile of this form by the device's reciprocal / reciprocal square root + Newton (4e-15 per factor).  47 instructions
 * against ~100 of FORM 0 with its two correctly rounded square roots and two divisions.  polycap_photon_launch (explicit
 * photons, where callers compare per photon) keeps FORMs 0/1. */
PC_HD double pc_fresnel3s(double d2, double n2r, double n2i, double zi2, double cr2, double c2, double es2, double ep2, double sd2)
{
	const double zr = c2 - d2;
	const double mag = pc_sqrt_fast(fma(zr, zr, zi2));
	const double Q = mag + fabs(zr);
	const double cS = cr2*pc_sqrt_fast(Q);
	const bool up = zr >= 0.;
	const double Gr = up ? Q : n2i, Gi = up ? n2i : Q;
	const double nr = cS - Gr, dr = cS + Gr;
	const double Gi2 = Gi*Gi;
	const double Ns = fma(nr, nr, Gi2), Ds = fma(dr, dr, Gi2);
	const double A = n2r*cS, B = n2i*cS;
	const double pr = A - Gr, pi_ = B - Gi, er = A + Gr, ei = B + Gi;
	const double Np = fma(pr, pr, pi_*pi_), Dp = fma(er, er, ei*ei);
	return pc_div_fast(fma(es2*Ns, Dp, (ep2*Np)*Ds), sd2*(Ds*Dp));
}

/* one energy of one reflection with pc_fresnel3s; same return values as pc_reflect_energy_f */
PC_HD int pc_reflect_energy_fast(const pc_energy_const &ec, const pc_refl_geom &g, double &w)
{
	if (ec.valid == 0.) return -1;
	const double rt = pc_fresnel3s(ec.d2, ec.n2_re, ec.n2_im, ec.zi2, pc_refl_cr2(g.alfa), g.alfa*g.alfa, g.es2, g.ep2, g.sd2);
	if (rt < 0. || rt > 1.) return -1;                          /* src/polycap-capil.c:633-637 */
	double f = rt;
	if (ec.rough_c != 0.) {
		const double c1 = ec.rough_c*g.alfa;
		f = rt*pc_exp_neg_fast(-c1*c1);
	}
	w = w*f;
	return (w >= 1.e-4) ? 1 : 0;
}

/* N reflections of one energy at once (the FAST loop of pc_trace_log_kernel): the same operations as N calls of pc_fresnel3,
 * written step by step for all of them so that the device compiler issues the N dependent chains alternately (its scheduler
 * would run them one after the other; a scheduling barrier after every step keeps the order written here).  in[k] = {cr2, c2,
 * fs, fp} of reflection k.  Bit-identical to the single evaluation. */
#if PC_FAST_MATH_DEVICE
#define PC_CHAIN_STEP() __builtin_amdgcn_sched_barrier(0)
#else
#define PC_CHAIN_STEP() ((void)0)
#endif
template <int N>
PC_HD void pc_fresnel3xN(double d2, double n2r, double n2i, double zi2, const double (&in)[N][4], double (&rt)[N])
{
#if PC_FAST_MATH_DEVICE
#define PC_EACH(expr) _Pragma("unroll") for (int k = 0; k < N; k++) { expr; } PC_CHAIN_STEP()
	double zr[N], m2[N], y[N], g[N], h[N], r[N], Q[N], S[N], cS[N], Gr[N], Gi[N], Ns[N], Ds[N], A[N], B[N], Np[N], Dp[N], t1[N], t2[N];
	PC_EACH(zr[k] = in[k][1] - d2);
	PC_EACH(m2[k] = fma(zr[k], zr[k], zi2));
	PC_EACH(y[k] = __builtin_amdgcn_rsq(m2[k]));
	PC_EACH(g[k] = m2[k]*y[k]);
	PC_EACH(h[k] = 0.5*y[k]);
	PC_EACH(r[k] = fma(-h[k], g[k], 0.5));
	PC_EACH(g[k] = fma(g[k], r[k], g[k]));                   /* |z| */
	PC_EACH(Q[k] = g[k] + fabs(zr[k]));
	PC_EACH(y[k] = __builtin_amdgcn_rsq(Q[k]));
	PC_EACH(g[k] = Q[k]*y[k]);
	PC_EACH(h[k] = 0.5*y[k]);
	PC_EACH(r[k] = fma(-h[k], g[k], 0.5));
	PC_EACH(S[k] = fma(g[k], r[k], g[k]));
	PC_EACH(cS[k] = in[k][0]*S[k]);
	PC_EACH(Gr[k] = (zr[k] >= 0.) ? Q[k] : n2i);
	PC_EACH(Gi[k] = (zr[k] >= 0.) ? n2i : Q[k]);
	PC_EACH(g[k] = cS[k] - Gr[k]);                           /* nr */
	PC_EACH(h[k] = cS[k] + Gr[k]);                           /* dr */
	PC_EACH(r[k] = Gi[k]*Gi[k]);
	PC_EACH(Ns[k] = fma(g[k], g[k], r[k]));
	PC_EACH(Ds[k] = fma(h[k], h[k], r[k]));
	PC_EACH(A[k] = n2r*cS[k]);
	PC_EACH(B[k] = n2i*cS[k]);
	PC_EACH(g[k] = A[k] - Gr[k]);                            /* pr */
	PC_EACH(h[k] = B[k] - Gi[k]);                            /* pi */
	PC_EACH(A[k] = A[k] + Gr[k]);                            /* er */
	PC_EACH(B[k] = B[k] + Gi[k]);                            /* ei */
	PC_EACH(h[k] = h[k]*h[k]);
	PC_EACH(Np[k] = fma(g[k], g[k], h[k]));
	PC_EACH(B[k] = B[k]*B[k]);
	PC_EACH(Dp[k] = fma(A[k], A[k], B[k]));
	PC_EACH(t1[k] = in[k][2]*Ns[k]);
	PC_EACH(t2[k] = in[k][3]*Np[k]);
	PC_EACH(t2[k] = t2[k]*Ds[k]);
	PC_EACH(t1[k] = fma(t1[k], Dp[k], t2[k]));               /* numerator */
	PC_EACH(m2[k] = Ds[k]*Dp[k]);                            /* denominator */
	PC_EACH(y[k] = __builtin_amdgcn_rcp(m2[k]));
	PC_EACH(r[k] = fma(-m2[k], y[k], 1.0));
	PC_EACH(y[k] = fma(y[k], r[k], y[k]));
	PC_EACH(rt[k] = t1[k]*y[k]);
#undef PC_EACH
#else
	for (int k = 0; k < N; k++) rt[k] = pc_fresnel3(d2, n2r, n2i, zi2, in[k][0], in[k][1], in[k][2], in[k][3]);
#endif
}

/* one energy of one reflection in FORM 3, the roughness factor per reflection as the reference applies it (:626-627).
 * Same return values as pc_reflect_energy_f. */
PC_HD int pc_reflect_energy3(const pc_energy_const &ec, double c, double c2, double fs, double fp, double &w)
{
	const double rt = pc_fresnel3(ec.d2, ec.n2_re, ec.n2_im, ec.zi2, pc_refl_cr2(c), c2, fs, fp);
	if (rt < 0. || rt > 1.) return -1;                          /* src/polycap-capil.c:633-637 */
	double f = rt;
	if (ec.rough_c != 0.) {
		const double c1 = ec.rough_c*c;
		f = rt*pc_exp_neg_fast(-c1*c1);
	}
	w = w*f;
	return (w >= 1.e-4) ? 1 : 0;
}

/* whole reflection for one lane: geometry, all energies in order (stopping at the first error like the reference),
 * new electric vector.  Returns 1 keep, 0 absorbed, -1 error. */
template <int NE, bool FASTF = false>
PC_HD int pc_reflect(const pc_params &Pm, const pc_energy_const *EC, pc_photon<NE> &ph,
                     double nx, double ny, double nz)
{
	pc_refl_geom g;
	if (pc_reflect_geom(ph, nx, ny, nz, g) < 0) return -1;
	int keep = 0;
	const int ne = (NE > 0) ? NE : Pm.n_energies;
	if (NE == 0 && Pm.form3) {
		/* what the kernels with weights in memory evaluate (serial here: the host compile of the tests) */
		bool all_valid = true;
		for (int e = 0; e < ne; e++) all_valid = all_valid && EC[e].valid != 0.;
		if (all_valid) {
			double c2, fs, fp;
			pc_refl_geom3(g, c2, fs, fp);
			for (int e = 0; e < ne; e++) {
				double we = ph.wset ? ph.wmem[e*ph.wstride] : 1.0;
				const int r = pc_reflect_energy3(EC[e], g.alfa, c2, fs, fp, we);
				if (r < 0) return -1;
				ph.wmem[e*ph.wstride] = we;
				keep |= r;
			}
			ph.wset = 1;
			ph.ex = fabs(ph.ex); ph.ey = fabs(ph.ey); ph.ez = fabs(ph.ez);
			return keep;
		}
	}
	for (int e = 0; e < ne; e++) {
		double we = (NE > 0) ? ph.w[NE > 0 ? e : 0] : (ph.wset ? ph.wmem[e*ph.wstride] : 1.0);
		int r = (FASTF && NE > 0) ? pc_reflect_energy_fast(EC[e], g, we)
		      : ((NE == 1) ? pc_reflect_energy_f<0>(EC[e], g, we) : ((NE > 1) ? pc_reflect_energy_f<1>(EC[e], g, we) : pc_reflect_energy(EC[e], g, we, ne == 1)));
		if (r < 0) return -1;
		if (NE > 0) ph.w[NE > 0 ? e : 0] = we; else ph.wmem[e*ph.wstride] = we;
		keep |= r;
	}
	ph.wset = 1;
	ph.ex = fabs(ph.ex); ph.ey = fabs(ph.ey); ph.ez = fabs(ph.ez);
	return keep;
}

/* ------------------------------------------------------------------ EVENT: one literal segment visit
 * src/polycap-capil.c:1246-1358 for segment ph.i, split so that kernels can run the per-energy loop of the
 * reflection cooperatively: pc_event_pre() does everything up to the reflection and returns PC_ST_REFLECT with the
 * hit data when one is due; pc_event_post() applies its outcome. */
#define PC_ST_REFLECT 6

struct pc_hit {
	double nx, ny, nz, cosalfa;
	int ix;
};

template <int NE>
PC_HD int pc_event_pre(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph, pc_hit &h)
{
	int i = ph.i;
	const int nmax = Pm.nmax;
	if (ph.lv == 3) {
		/* sent here by a failed probe at stride PC_L1 (pc_march_ok): certified single segments up to the first one that is
		 * not -- the same certificate, values and order as single march steps -- then the literal visit of that one.
		 * Nodes i+1 .. i+PC_L1 exist: the probe looked at the last of them (which may still pass the single-segment margin). */
		ph.lv = 0;
		const float adjf = Pm.adjf;
#pragma unroll
		for (int k = 1; k <= PC_L1; k++) {
			const double C = pc_node_C(T, ph, i + 1);
			if (!(((float)ph.C0 < -adjf) & ((float)C < -adjf))) break;
			ph.C0 = C;
			i++;
		}
		ph.i = i;
	}
	if (i >= nmax) { ph.rc = 1; return PC_ST_DONE; }           /* :1312-1313 -> launch returns 1 */

	/* :1263 capillary axis inside the optic at both ends of the segment (only boundary capillaries can fail) */
	if (ph.bnd) {
		if ((T.hexd[i] > 0. && pc_outside_hexd(T.hexd[i], ph.kx*T.zh[i], ph.ky*T.zh[i])) ||
		    (T.hexd[i+1] > 0. && pc_outside_hexd(T.hexd[i+1], ph.kx*T.zh[i+1], ph.ky*T.zh[i+1]))) {
			ph.rc = -1; return PC_ST_DONE;
		}
	}
	double p0x, p0y, hx, hy, hz;
	int iesc = pc_segment(T, ph, i, p0x, p0y, hx, hy, hz, h.nx, h.ny, h.nz);
	double cosalfa = fma(h.nx, ph.dx, fma(h.ny, ph.dy, h.nz*ph.dz));
	if (cosalfa < 0.) iesc = -5;                                /* acos(cosalfa) > pi/2, :1270-1273 */
	h.cosalfa = cosalfa;

	if (iesc != 1) {
		/* :1296-1308 the ray at z_i must still be inside the optic (trace -3 -> launch -1) */
		if (T.hexd[i] > 0. && pc_outside_hexd(T.hexd[i], p0x, p0y)) { ph.rc = -1; return PC_ST_DONE; }
		ph.i = i + 1;
		ph.first = 0;
		ph.C0 = pc_node_C(T, ph, i + 1);
		return PC_ST_MARCH;
	}

	/* :1277-1294 hit: still inside the optic?  (implied for non-boundary capillaries: the wall lies inside the hexagon) */
	if (ph.bnd) {
		double cur_ext = ((T.ext[i] - T.ext[i+1])/(T.z[i] - T.z[i+1])) * (hz - T.z[i+1]) + T.ext[i+1];
		if (Pm.mono) {
			if (sqrt(hx*hx + hy*hy) >= cur_ext) { ph.rc = -1; return PC_ST_DONE; }
		} else {
			if (pc_outside_hex(cur_ext, hx, hy)) { ph.rc = -1; return PC_ST_DONE; }
		}
	}
	/* :1315-1324 */
	/* |hit - P| along a unit direction = (hz - P.z)/dz: the reference's square root (:1315-1318) without the root */
	ph.dtravel += (hz - ph.Pz) * ph.idzd;
	ph.Px = hx; ph.Py = hy; ph.Pz = hz;
	if (fabs(cosalfa) > 1.0) { ph.rc = -1; return PC_ST_DONE; } /* :1325-1327 */
	/* :1330-1333 rescan: last node index < nmax with z <= hit z (z strictly increasing, z_i <= hz <= z_i+1) */
	int ix = (hz >= T.z[i+1] && i + 1 < nmax) ? i + 1 : i;
	h.ix = ix;
	/* :1334-1343 */
	if (ph.bnd) {
		double cur_ext = ((T.ext[ix+1] - T.ext[ix])/(T.z[ix+1] - T.z[ix])) * (hz - T.z[ix]) + T.ext[ix];
		if (Pm.mono) {
			if (hx*hx + hy*hy >= cur_ext) { ph.rc = -1; return PC_ST_DONE; }
		} else {
			if (pc_outside_hex(cur_ext, hx, hy)) { ph.rc = -1; return PC_ST_DONE; }
		}
	}
	return PC_ST_REFLECT;
}

/* r = outcome of the reflection (1 keep, 0 absorbed, -1 error): :1345-1355 and the launch loop bound */
template <int NE>
PC_HD int pc_event_post(const pc_params &Pm, pc_photon<NE> &ph, const pc_hit &h, int r)
{
	if (r == 0) { ph.rc = 0; return PC_ST_DONE; }
	if (r != 1) { ph.rc = -1; return PC_ST_DONE; }
	/* mirror reflection of a unit vector about a unit normal stays unit: the reference's re-normalisation is a no-op up to rounding */
	ph.dx = fma(-2.0*h.cosalfa, h.nx, ph.dx);
	ph.dy = fma(-2.0*h.cosalfa, h.ny, ph.dy);
	ph.dz = fma(-2.0*h.cosalfa, h.nz, ph.dz);
	ph.irefl++;
	/* src/polycap-photon.c:912-919: at most nmax+1 trace calls; every call that returns 1 is one reflection */
	if (ph.irefl > Pm.nmax) { ph.rc = 1; return PC_ST_DONE; }
	ph.i = h.ix;
	pc_trace_begin(ph);
	return PC_ST_MARCH;
}

/* FASTF: the weights of a register-weight kernel are multiplied with pc_fresnel3s (source runs) instead of FORMs 0/1 */
template <int NE, bool FASTF = false>
PC_HD int pc_event(const pc_tables &T, const pc_params &Pm, const pc_energy_const *EC, pc_photon<NE> &ph)
{
	pc_hit h;
	int st = pc_event_pre(T, Pm, ph, h);
	if (st != PC_ST_REFLECT) return st;
	int r = pc_reflect<NE, FASTF>(Pm, EC, ph, h.nx, h.ny, h.nz);
	return pc_event_post(Pm, ph, h, r);
}

/* MARCH step wrapper: returns the next state (MARCH to keep going, EVENT, or DONE at the end of the optic) */
template <int NE>
PC_HD int pc_march_step(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph)
{
	if (ph.i >= Pm.nmax) { ph.rc = 1; return PC_ST_DONE; }
	if (Pm.literal) return PC_ST_EVENT;
	if (ph.first) return pc_march_first_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT;
	return pc_march_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT;
}

/* MARCH step of the tight loop: the lane is known not to sit on the first segment of a trace call */
template <int NE>
PC_HD int pc_march_step_hot(const pc_tables &T, const pc_params &Pm, pc_photon<NE> &ph)
{
	if (ph.i >= Pm.nmax) { ph.rc = 1; return PC_ST_DONE; }
	return pc_march_ok(T, Pm, ph) ? PC_ST_MARCH : PC_ST_EVENT;
}

/* ------------------------------------------------------------------ exit window + image record
 * src/polycap-source.c:762-777: extrapolate to z[nmax] and test the exit hexagon (mono: circle). */
template <int NE>
PC_HD int pc_in_exit_window(const pc_params &Pm, const pc_photon<NE> &ph)
{
	double t = (Pm.z_end - ph.Pz) / ph.dz;
	double tx = ph.Px + ph.dx * t;
	double ty = ph.Py + ph.dy * t;
	if (Pm.mono)
		return (sqrt(tx*tx + ty*ty) > Pm.ext_end) ? 0 : 1;
	return pc_outside_hex(Pm.ext_end, tx, ty) ? 0 : 1;
}

#endif /* PC_DEVICE_H */
