/*
 * polycap_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see polycap_oracle.h).
 *
 * Plain-C fp64 restatement of the reference's per-photon trace path, same operation
 * order as the reference so that the reference's own known answers pin it.
 * Compile with -ffp-contract=off (see Makefile) so no FMA contraction changes rounding.
 *
 * Parity status: pinned by the reference's known-answer tests only (no compiled
 * reference available in this image -- it needs meson's config.h, GSL and xraylib).
 */
#include "polycap_oracle.h"
#include <math.h>
#include <complex.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* constants: include/polycap.h:47-49, src/polycap-private.h:30-38 */
#define ORC_HC     1.23984193E-7
#define ORC_N_AVOG 6.022098e+23
#define ORC_R0     2.8179403227e-13
#define ORC_COSPI_6 0.86602540378443864676
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ helpers */

/* src/polycap-photon.c:365-376 */
void orc_norm(orc_vec3 *v)
{
	double sum = sqrt(v->x*v->x + v->y*v->y + v->z*v->z);
	v->x /= sum;
	v->y /= sum;
	v->z /= sum;
}

/* src/polycap-photon.c:379-386 */
double orc_scalar(orc_vec3 a, orc_vec3 b)
{
	return a.x*b.x + a.y*b.y + a.z*b.z;
}

/* src/polycap-photon.c:139-169: 1 inside, 0 outside, -1 bad radius; NaN coordinates compare false -> 1 */
int orc_within_pc_boundary(double polycap_radius, orc_vec3 c)
{
	if (polycap_radius <= 0.)
		return -1;
	double d_cen2hexedge = sqrt((polycap_radius * polycap_radius) - ((polycap_radius/2.) * (polycap_radius/2.)));
	double dp1 = fabs(0*c.x + 1*c.y);
	double dp2 = fabs(ORC_COSPI_6*c.x + 0.5*c.y);
	double dp3 = fabs(ORC_COSPI_6*c.x + -0.5*c.y);
	if (dp1 > d_cen2hexedge || dp2 > d_cen2hexedge || dp3 > d_cen2hexedge)
		return 0;
	return 1;
}

/* src/polycap-photon.c:483 */
double orc_n_shells(int64_t n_cap)
{
	return round(sqrt(12. * n_cap - 3.)/6.-0.5);
}

/* src/polycap-description.c:214-216 */
double orc_open_area(const orc_optic *o)
{
	double n_cap_temp = (round(sqrt(12. * o->n_cap - 3.)/6.-0.5)+0.5)*6.;
	n_cap_temp = (n_cap_temp*n_cap_temp+3)/12;
	return (o->cap[0]*o->cap[0]*M_PI)*n_cap_temp/(3.*sin(M_PI/3)*o->ext[0]*o->ext[0]);
}

/* src/polycap-profile.c:66-207 (CONICAL :142-148, ELLIPSOIDAL :171-196) */
int orc_profile_new(int type, double length, double rad_ext_upstream, double rad_ext_downstream,
                    double rad_int_upstream, double rad_int_downstream,
                    double focal_dist_upstream, double focal_dist_downstream,
                    int nmax, double *z, double *cap, double *ext)
{
	int i;
	double slope, b, k, a;
	if (type == 0) {
		for (i = 0; i <= nmax; i++) {
			z[i] = length/nmax*i;
			cap[i] = (rad_int_downstream-rad_int_upstream)/length*z[i] + rad_int_upstream;
			ext[i] = (rad_ext_downstream-rad_ext_upstream)/length*z[i] + rad_ext_upstream;
		}
		return 0;
	}
	if (type == 2) {
		if (rad_ext_downstream < rad_ext_upstream) {
			slope = rad_ext_downstream / focal_dist_downstream;
			b = (-1.*(rad_ext_downstream-rad_ext_upstream)*(rad_ext_downstream-rad_ext_upstream)-slope*length*(rad_ext_downstream-rad_ext_upstream)) / (slope*length+2.*(rad_ext_downstream-rad_ext_upstream));
			k = rad_ext_upstream - b;
			a = sqrt((b*b*length)/(slope*(rad_ext_downstream-k)));
			for (i = 0; i <= nmax; i++) {
				z[i] = length/nmax*i;
				cap[i] = (rad_int_downstream-rad_int_upstream)/length*z[i] + rad_int_upstream;
				ext[i] = sqrt(b*b-(b*b*z[i]*z[i])/(a*a))+k;
			}
		} else {
			slope = rad_ext_upstream / focal_dist_upstream;
			b = (-1.*(rad_ext_upstream-rad_ext_downstream)*(rad_ext_upstream-rad_ext_downstream)-slope*length*(rad_ext_upstream-rad_ext_downstream)) / (slope*length+2.*(rad_ext_upstream-rad_ext_downstream));
			k = rad_ext_downstream - b;
			a = sqrt(fabs((b*b*length)/(slope*(rad_ext_upstream-k))));
			for (i = 0; i <= nmax; i++) {
				z[i] = length/nmax*i;
				cap[i] = (rad_int_downstream-rad_int_upstream)/length*z[i] + rad_int_upstream;
			}
			for (i = 0; i <= nmax; i++)
				ext[i] = sqrt(b*b-(b*b*z[nmax-i]*z[nmax-i])/(a*a))+k;
		}
		return 0;
	}
	return -1;
}

/* ------------------------------------------------------------------ segment */

/* src/polycap-capil.c:52-255.  Return codes: 1 hit, -1 bad argument, -2 no real root,
 * -3 neither root valid, -4 beyond segment, -5 before segment / not past last hit, -6 too close. */
int orc_segment(orc_vec3 cap_coord0, orc_vec3 cap_coord1, double cap_rad0, double cap_rad1,
                orc_vec3 phot_coord0, orc_vec3 phot_coord1, orc_vec3 photon_dir,
                orc_vec3 *photon_coord, orc_vec3 *surface_norm)
{
	double d_proj, d_cap_inter, d_cap_coord, tga, sga, cga, gam;
	double a, b, c, discr, dist1, dist2;
	orc_vec3 cap_coord, interact_coord, interact_norm, cap_dir, photon_coord_rel;

	/* :65-100 argument checks */
	if (cap_coord0.z < 0.) return -1;
	if (cap_coord1.z < 0) return -1;
	if (cap_rad0 < 0.) return -1;
	if (cap_rad1 < 0.) return -1;
	if (photon_coord == NULL) return -1;
	if (photon_dir.z < 0) return -1;
	if (surface_norm == NULL) return -1;
	if (cap_coord0.z != phot_coord0.z || cap_coord1.z != phot_coord1.z) return -1;
	if (cap_coord1.z <= cap_coord0.z) return -1;

	/* :102-105 */
	surface_norm->x = 0.0;
	surface_norm->y = 0.0;
	surface_norm->z = 0.0;
	orc_norm(&photon_dir);

	/* :109-112 */
	cap_dir.x = cap_coord1.x - cap_coord0.x;
	cap_dir.y = cap_coord1.y - cap_coord0.y;
	cap_dir.z = cap_coord1.z - cap_coord0.z;
	d_cap_coord = sqrt(orc_scalar(cap_dir, cap_dir));

	/* :119-124 quadratic in (z - z0) */
	a = ( ((photon_dir.x/photon_dir.z)-(cap_dir.x/cap_dir.z))*((photon_dir.x/photon_dir.z)-(cap_dir.x/cap_dir.z)) +
		((photon_dir.y/photon_dir.z)-(cap_dir.y/cap_dir.z))*((photon_dir.y/photon_dir.z)-(cap_dir.y/cap_dir.z)) -
		((cap_rad1-cap_rad0)/(cap_coord1.z-cap_coord0.z))*((cap_rad1-cap_rad0)/(cap_coord1.z-cap_coord0.z)) );
	b = (2.*(phot_coord0.x-cap_coord0.x)*((photon_dir.x/photon_dir.z)-(cap_dir.x/cap_dir.z)) + 2.*(phot_coord0.y-cap_coord0.y)*((photon_dir.y/photon_dir.z)-(cap_dir.y/cap_dir.z)) - 2.*cap_rad0*((cap_rad1-cap_rad0)/(cap_coord1.z-cap_coord0.z)));
	c = ( (phot_coord0.x-cap_coord0.x)*(phot_coord0.x-cap_coord0.x) + (phot_coord0.y-cap_coord0.y)*(phot_coord0.y-cap_coord0.y) - cap_rad0*cap_rad0);
	discr = b*b - 4.*a*c;
	/* :125-157 root selection */
	if (discr < 0)
		return -2;
	if (discr == 0) {
		dist1 = (-1.*b)/(2.*a);
		interact_coord.z = phot_coord0.z + dist1;
	} else {
		dist1 = (-1.*b + sqrt(discr))/(2.*a);
		dist2 = (-1.*b - sqrt(discr))/(2.*a);
		if (phot_coord0.z + dist1 < cap_coord0.z || phot_coord0.z + dist1 - photon_coord->z < 1.e-5 || phot_coord0.z + dist1 > cap_coord1.z) {
			if (phot_coord0.z + dist2 < cap_coord0.z || phot_coord0.z + dist2 - photon_coord->z < 1.e-5 || phot_coord0.z + dist2 > cap_coord1.z) {
				return -3;
			} else interact_coord.z = phot_coord0.z + dist2;
		} else {
			if (phot_coord0.z + dist2 < cap_coord0.z || phot_coord0.z + dist2 - photon_coord->z < 1.e-5 || phot_coord0.z + dist2 > cap_coord1.z) {
				interact_coord.z = phot_coord0.z + dist1;
			} else {
				if (phot_coord0.z + dist2 - photon_coord->z < phot_coord0.z + dist1 - photon_coord->z) {
					interact_coord.z = phot_coord0.z + dist2;
				} else {
					interact_coord.z = phot_coord0.z + dist1;
				}
			}
		}
	}

	/* :168-171 */
	if (interact_coord.z > cap_coord1.z)
		return -4;
	if (interact_coord.z < cap_coord0.z || interact_coord.z - photon_coord->z < 1.e-5)
		return -5;

	/* :175-179 */
	d_proj = (interact_coord.z - phot_coord0.z) / photon_dir.z;
	if (d_proj < 1.e-10)
		return -6;
	interact_coord.x = phot_coord0.x + d_proj * photon_dir.x;
	interact_coord.y = phot_coord0.y + d_proj * photon_dir.y;

	/* :225-230 axis point at the interaction */
	photon_coord_rel.x = phot_coord0.x - cap_coord0.x;
	photon_coord_rel.y = phot_coord0.y - cap_coord0.y;
	photon_coord_rel.z = phot_coord0.z - cap_coord0.z;
	cap_coord.x = cap_coord0.x + ((d_proj+(orc_scalar(photon_coord_rel,cap_dir)/orc_scalar(photon_dir,cap_dir)))/(orc_scalar(cap_dir,cap_dir)/orc_scalar(photon_dir,cap_dir)))*cap_dir.x;
	cap_coord.y = cap_coord0.y + ((d_proj+(orc_scalar(photon_coord_rel,cap_dir)/orc_scalar(photon_dir,cap_dir)))/(orc_scalar(cap_dir,cap_dir)/orc_scalar(photon_dir,cap_dir)))*cap_dir.y;
	cap_coord.z = cap_coord0.z + ((d_proj+(orc_scalar(photon_coord_rel,cap_dir)/orc_scalar(photon_dir,cap_dir)))/(orc_scalar(cap_dir,cap_dir)/orc_scalar(photon_dir,cap_dir)))*cap_dir.z;

	/* :233-246 radial unit vector tilted by the wall angle */
	interact_norm.x = interact_coord.x - cap_coord.x;
	interact_norm.y = interact_coord.y - cap_coord.y;
	interact_norm.z = interact_coord.z - cap_coord.z;
	d_cap_inter = sqrt(orc_scalar(interact_norm, interact_norm));
	tga = (cap_rad0 - cap_rad1)/d_cap_coord;
	gam = atan(tga);
	sga = sin(gam);
	cga = cos(gam);
	surface_norm->x = cga * interact_norm.x / d_cap_inter + sga * cap_dir.x / d_cap_coord;
	surface_norm->y = cga * interact_norm.y / d_cap_inter + sga * cap_dir.y / d_cap_coord;
	surface_norm->z = cga * interact_norm.z / d_cap_inter + sga * cap_dir.z / d_cap_coord;
	orc_norm(surface_norm);

	/* :250-252 */
	photon_coord->x = interact_coord.x;
	photon_coord->y = interact_coord.y;
	photon_coord->z = interact_coord.z;
	return 1;
}

/* ------------------------------------------------------------------ Fresnel */

/* src/polycap-capil.c:444-563 (HAVE_PROPER_COMPLEX_H branch, :27-38: C99 complex arithmetic) */
double orc_refl_polar(double e, double density, double scatf, double lin_abs_coeff,
                      orc_vec3 surface_norm, orc_photon *photon, orc_vec3 *electric_vector)
{
	double alfa, beta;
	double complex n, r_s, r_p, n_inv, our_csqrt, tmp;
	orc_vec3 s_dir, p_dir;
	double frac_s, frac_p, angle_a, angle_b, angle_c;
	double cos_theta, sin_theta, theta;
	double r_s_double, r_p_double, rtot;

	/* :463-482 */
	if (e < 1. || e > 100.) return -1;
	if (density <= 0.) return -1;
	if (scatf < 0.) return -1;
	if (lin_abs_coeff < 0.) return -1;
	if (photon == NULL) return -1;

	/* :486-494 */
	if (sqrt(surface_norm.x*surface_norm.x+surface_norm.y*surface_norm.y+surface_norm.z*surface_norm.z) != 1)
		orc_norm(&surface_norm);
	theta = acos(orc_scalar(surface_norm, photon->exit_direction));
	if (theta < 0.) return -1;
	if (sqrt(photon->exit_electric_vector.x*photon->exit_electric_vector.x+photon->exit_electric_vector.y*photon->exit_electric_vector.y+photon->exit_electric_vector.z*photon->exit_electric_vector.z) != 1)
		orc_norm(&photon->exit_electric_vector);

	/* :497-499 */
	alfa = (ORC_HC/e)*(ORC_HC/e)*((ORC_N_AVOG*ORC_R0*density)/(2*M_PI)) * scatf;
	beta = (ORC_HC)/(4.*M_PI) * (lin_abs_coeff/e);
	n = (1.0 - alfa) + I * beta;

	/* :501-515 */
	cos_theta = cos(theta);
	sin_theta = sin(theta);
	n_inv = 1.0/n;
	tmp = (n_inv * n_inv) * (sin_theta * sin_theta);
	our_csqrt = csqrt((1.0 - creal(tmp)) + I * (-1.0 * cimag(tmp)));

	tmp = n * our_csqrt;
	r_s = ((cos_theta - creal(tmp)) + I * (-1.0 * cimag(tmp))) * (1.0/((cos_theta + creal(tmp)) + I * cimag(tmp)));
	r_s_double = cabs(r_s);
	r_s_double *= r_s_double;

	tmp = n * cos_theta;
	r_p = ((creal(our_csqrt) - creal(tmp)) + I * (cimag(our_csqrt) - cimag(tmp))) * (1.0/((creal(our_csqrt) + creal(tmp)) + I * (cimag(our_csqrt) + cimag(tmp))));
	r_p_double = cabs(r_p);
	r_p_double *= r_p_double;

	/* :520-529 s and p directions */
	s_dir.x = surface_norm.y*photon->exit_direction.z - photon->exit_direction.y*surface_norm.z;
	s_dir.y = surface_norm.z*photon->exit_direction.x - photon->exit_direction.z*surface_norm.x;
	s_dir.z = surface_norm.x*photon->exit_direction.y - photon->exit_direction.x*surface_norm.y;
	orc_norm(&s_dir);
	p_dir.x = photon->exit_direction.y*s_dir.z - s_dir.y*photon->exit_direction.z;
	p_dir.y = photon->exit_direction.z*s_dir.x - s_dir.z*photon->exit_direction.x;
	p_dir.z = photon->exit_direction.x*s_dir.y - s_dir.x*photon->exit_direction.y;
	orc_norm(&p_dir);

	/* :537-543 */
	angle_a = orc_scalar(photon->exit_electric_vector, s_dir);
	frac_s = angle_a*angle_a;
	frac_p = 1.-frac_s;
	rtot = r_s_double * frac_s + r_p_double * frac_p;

	/* :546-558 new electric vector (component signs are lost, as in the reference) */
	angle_b = orc_scalar(photon->exit_electric_vector, surface_norm);
	angle_c = orc_scalar(photon->exit_electric_vector, p_dir);
	{
		orc_vec3 ev = photon->exit_electric_vector;
		electric_vector->x = sqrt( (ev.x*angle_a*frac_s)*(ev.x*angle_a*frac_s) +
			(ev.x*angle_b*frac_p)*(ev.x*angle_b*frac_p) +
			(ev.x*angle_c*frac_p)*(ev.x*angle_c*frac_p) );
		electric_vector->y = sqrt( (ev.y*angle_a*frac_s)*(ev.y*angle_a*frac_s) +
			(ev.y*angle_b*frac_p)*(ev.y*angle_b*frac_p) +
			(ev.y*angle_c*frac_p)*(ev.y*angle_c*frac_p) );
		electric_vector->z = sqrt( (ev.z*angle_a*frac_s)*(ev.z*angle_a*frac_s) +
			(ev.z*angle_b*frac_p)*(ev.z*angle_b*frac_p) +
			(ev.z*angle_c*frac_p)*(ev.z*angle_c*frac_p) );
	}
	orc_norm(electric_vector);
	return rtot;
}

/* src/polycap-capil.c:565-655 + 889-891, leak_calc=false: 1 keep, 0 absorbed, -1 error */
int orc_reflect(const orc_optic *optic, orc_photon *photon, orc_vec3 surface_norm)
{
	size_t i;
	int weight_flag = 0;
	double cons1, r_rough, rtot, alfa;
	orc_vec3 electric_vector = {0., 0., 0.};

	if (photon == NULL || optic == NULL) return -1;
	if (photon->leak_calc)
		return orc_reflect_leak(optic, photon, surface_norm);   /* polycap_oracle_leak.c */
	/* :596-602 */
	orc_norm(&surface_norm);
	orc_norm(&photon->exit_direction);
	alfa = orc_scalar(photon->exit_direction, surface_norm);
	if (alfa < 0.) return -1;

	/* :625-645 */
	for (i = 0; i < photon->n_energies; i++) {
		cons1 = (1.01358e0*photon->energies[i])*alfa*optic->sig_rough;
		r_rough = exp(-1.*cons1*cons1);
		rtot = orc_refl_polar(photon->energies[i], optic->density, photon->scatf[i], photon->amu[i], surface_norm, photon, &electric_vector);
		if (rtot < 0. || rtot > 1.)
			return -1;
		photon->weight[i] = photon->weight[i] * rtot * r_rough;
		if (photon->weight[i] >= 1.e-4) weight_flag = 1;
	}
	/* :648-654 */
	photon->exit_electric_vector.x = electric_vector.x;
	photon->exit_electric_vector.y = electric_vector.y;
	photon->exit_electric_vector.z = electric_vector.z;
	return weight_flag == 1 ? 1 : 0;
}

/* ------------------------------------------------------------------ trace */

/* src/polycap-capil.c:1197-1361.  1 reflected, 0 absorbed, -1 error, -2 flew out the end, -3 left the optic */
int orc_trace(const orc_optic *optic, int *ix, orc_photon *photon, const double *cap_x, const double *cap_y)
{
	int i, iesc = 0;
	double cap_rad0, cap_rad1;
	orc_vec3 cap_coord0, cap_coord1, phot_coord0, phot_coord1, photon_coord, photon_dir;
	orc_vec3 surface_norm = {0., 0., 0.}; /* reference leaves it uninitialised; only read after segment() wrote it or when iesc!=1 */
	orc_vec3 photon_coord_rel, temp_phot;
	double cosalfa = 0., d_travel, current_polycap_ext, n_shells;
	const double *z = optic->z, *cap = optic->cap, *ext = optic->ext;
	const int nmax = optic->nmax;

	if (ix == NULL || photon == NULL || optic == NULL || cap_x == NULL || cap_y == NULL) return -1;

	/* :1236-1243 */
	orc_norm(&photon->exit_direction);
	orc_norm(&photon->start_direction);
	photon_coord = photon->exit_coords;
	photon_dir = photon->exit_direction;

	n_shells = orc_n_shells(optic->n_cap);
	/* :1246-1310 march */
	for (i = *ix; i < nmax; i++) {
		cap_coord0.x = cap_x[i];
		cap_coord0.y = cap_y[i];
		cap_coord0.z = z[i];
		cap_rad0 = cap[i];
		cap_coord1.x = cap_x[i+1];
		cap_coord1.y = cap_y[i+1];
		cap_coord1.z = z[i+1];
		cap_rad1 = cap[i+1];
		phot_coord0.x = photon->exit_coords.x + photon->exit_direction.x * (z[i]-photon->exit_coords.z)/photon->exit_direction.z;
		phot_coord0.y = photon->exit_coords.y + photon->exit_direction.y * (z[i]-photon->exit_coords.z)/photon->exit_direction.z;
		phot_coord0.z = z[i];
		phot_coord1.x = photon->exit_coords.x + photon->exit_direction.x * (z[i+1]-photon->exit_coords.z)/photon->exit_direction.z;
		phot_coord1.y = photon->exit_coords.y + photon->exit_direction.y * (z[i+1]-photon->exit_coords.z)/photon->exit_direction.z;
		phot_coord1.z = z[i+1];
		/* :1263 */
		if ((orc_within_pc_boundary(ext[i], cap_coord0) == 0) || (orc_within_pc_boundary(ext[i+1], cap_coord1) == 0))
			return -1;
		/* :1268-1273 */
		iesc = orc_segment(cap_coord0, cap_coord1, cap_rad0, cap_rad1, phot_coord0, phot_coord1, photon_dir, &photon_coord, &surface_norm);
		cosalfa = orc_scalar(surface_norm, photon_dir);
		if (acos(cosalfa) > M_PI/2. || acos(cosalfa) < 0.)
			iesc = -5;

		if (iesc == 1) {
			/* :1277-1294 */
			current_polycap_ext = ((ext[i] - ext[i+1])/(z[i] - z[i+1])) * (photon_coord.z - z[i+1]) + ext[i+1];
			if (n_shells == 0.) {
				if (sqrt(photon_coord.x*photon_coord.x + photon_coord.y*photon_coord.y) >= current_polycap_ext)
					return -3;
			} else {
				if (orc_within_pc_boundary(current_polycap_ext, photon_coord) == 0)
					return -3;
			}
			*ix = i+1;
			break;
		} else {
			/* :1296-1308 */
			temp_phot.x = photon->exit_coords.x + photon->exit_direction.x * (z[i]-photon->exit_coords.z)/photon->exit_direction.z;
			temp_phot.y = photon->exit_coords.y + photon->exit_direction.y * (z[i]-photon->exit_coords.z)/photon->exit_direction.z;
			temp_phot.z = z[i];
			if (orc_within_pc_boundary(ext[i], temp_phot) == 0)
				return -3;
		}
	}

	if (iesc != 1) {
		iesc = -2; /* :1312-1313 */
	} else {
		/* :1315-1324 */
		photon_coord_rel.x = photon_coord.x - photon->exit_coords.x;
		photon_coord_rel.y = photon_coord.y - photon->exit_coords.y;
		photon_coord_rel.z = photon_coord.z - photon->exit_coords.z;
		d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
		photon->d_travel += d_travel;
		photon->exit_coords = photon_coord;
		if (fabs(cosalfa) > 1.0) {
			iesc = -1; /* :1325-1327 */
		} else {
			/* :1330-1333 O(nmax) rescan */
			for (i = 0; i < nmax; i++) {
				if (z[i] <= photon->exit_coords.z)
					*ix = i;
			}
			/* :1334-1343 */
			current_polycap_ext = ((ext[(*ix)+1] - ext[(*ix)])/(z[(*ix)+1] - z[(*ix)])) * (photon_coord.z - z[(*ix)]) + ext[(*ix)];
			if (n_shells == 0 && photon->exit_coords.x*photon->exit_coords.x+photon->exit_coords.y*photon->exit_coords.y >= current_polycap_ext) {
				iesc = -3;
			} else if (n_shells > 0 && orc_within_pc_boundary(current_polycap_ext, photon->exit_coords) == 0) {
				iesc = -3;
			} else {
				/* :1345-1355 */
				iesc = orc_reflect(optic, photon, surface_norm);
				if (iesc == 1) {
					photon->exit_direction.x = photon->exit_direction.x - 2.0*cosalfa * surface_norm.x;
					photon->exit_direction.y = photon->exit_direction.y - 2.0*cosalfa * surface_norm.y;
					photon->exit_direction.z = photon->exit_direction.z - 2.0*cosalfa * surface_norm.z;
					orc_norm(&photon->exit_direction);
					photon->i_refl++;
				} else if (iesc == -1 || iesc == -2) {
					iesc = -1;
				}
			}
		}
	}
	return iesc;
}

/* ------------------------------------------------------------------ launch */

/* per-thread pair of axis arrays (see orc_launch); grown on demand and kept for the life of the thread */
static int orc_axis_scratch(int n, double **cap_x, double **cap_y)
{
	static __thread double *buf = NULL;
	static __thread int have = 0;
	if (have < n) {
		double *nb = realloc(buf, sizeof(double) * 2 * (size_t)n);
		if (nb == NULL) return -1;
		buf = nb;
		have = n;
	}
	*cap_x = buf;
	*cap_y = buf + n;
	return 0;
}

/* src/polycap-photon.c:390-955 (leak_calc = photon->leak_calc).
 * amu/scatf are supplied by the caller (the reference calls xraylib via polycap_photon_scatf
 * at :495 for every launch; the values depend only on composition and energy). */
int orc_launch(const orc_optic *optic, orc_photon *photon, size_t n_energies, const double *energies,
               const double *amu, const double *scatf, double *weights)
{
	int i, iesc = 0;
	double n_shells, q_i, r_i, zz;
	double *cap_x, *cap_y;
	int ix_val = 0;
	int *ix = &ix_val;
	double d_ph_capcen;
	int z_id = 0;
	double current_polycap_ext = 0, current_cap_rad = 0, current_cap_x, current_cap_y;
	const double *z = optic->z, *cap = optic->cap, *ext = optic->ext;
	const int nmax = optic->nmax;

	/* :410-431 */
	if (photon == NULL || energies == NULL || n_energies < 1 || weights == NULL) return -1;
	for (i = 0; i < (int)n_energies; i++)
		if (energies[i] < 1. || energies[i] > 100.) return -1;

	/* :458-479 */
	photon->n_energies = n_energies;
	photon->energies = energies;
	photon->weight = weights;
	photon->amu = amu;
	photon->scatf = scatf;
	for (i = 0; i < (int)n_energies; i++)
		photon->weight[i] = 1.;
	photon->i_refl = 0;

	/* :483-504 */
	n_shells = orc_n_shells(optic->n_cap);
	orc_norm(&photon->start_direction);
	photon->exit_coords = photon->start_coords;
	photon->exit_direction = photon->start_direction;
	orc_norm(&photon->exit_direction);

	/* :507-512 */
	if (photon->start_coords.z > 0) {
		for (i = 0; i < nmax; i++)
			if (z[i] <= photon->start_coords.z) z_id = i;
	} else z_id = 0;
	current_polycap_ext = ((ext[z_id] - ext[z_id+1]) / (z[z_id] - z[z_id+1])) * (photon->start_coords.z - z[z_id]) + ext[z_id];

	if (n_shells == 0.) {
		/* :514-537 */
		q_i = 0;
		r_i = 0;
		if (sqrt((photon->start_coords.x)*(photon->start_coords.x) + (photon->start_coords.y)*(photon->start_coords.y)) > current_polycap_ext)
			return -2;
	} else {
		/* :538-574 */
		zz = current_polycap_ext/(2.*ORC_COSPI_6*(n_shells+1));
		r_i = photon->start_coords.y * (2./3) / zz;
		q_i = (photon->start_coords.x/(2.*ORC_COSPI_6) - photon->start_coords.y/3) / zz;
		if (fabs(q_i - round(q_i)) > fabs(r_i - round(r_i)) && fabs(q_i - round(q_i)) > fabs(-1.*q_i-r_i - round(-1.*q_i-r_i))) {
			q_i = -1.*round(r_i) - round(-1.*q_i-r_i);
			r_i = round(r_i);
		} else if (fabs(r_i - round(r_i)) > fabs(-1.*q_i-r_i - round(-1.*q_i-r_i))) {
			r_i = -1.*round(q_i) - round(-1.*q_i-r_i);
			q_i = round(q_i);
		} else {
			q_i = round(q_i);
			r_i = round(r_i);
		}
		if (orc_within_pc_boundary(current_polycap_ext, photon->start_coords) == 0)
			return -2;
	}

	/* :578-627 capillary axis */
	/* the reference mallocs and frees the two axis arrays in every launch (:578-581, :930-945); the oracle keeps one pair
	 * per thread so that the OpenMP driver does not serialise on the allocator (same values, same fill loop) */
	if (orc_axis_scratch(nmax + 1, &cap_x, &cap_y) != 0) return -1;
	for (i = 0; i <= nmax; i++) {
		zz = ext[i]/(2.*ORC_COSPI_6*(n_shells+1));
		cap_y[i] = r_i * (3./2) * zz;
		cap_x[i] = (2.* q_i+r_i) * ORC_COSPI_6 * zz;
		if (z[i] <= photon->start_coords.z) *ix = i;
	}
	/* :629-645 */
	if (photon->start_coords.z > 0) {
		current_cap_rad = ((cap[z_id+1] - cap[z_id])/(z[z_id+1] - z[z_id])) * (photon->start_coords.z - z[z_id]) + cap[z_id];
		current_cap_x = ((cap_x[z_id+1] - cap_x[z_id])/(z[z_id+1] - z[z_id])) * (photon->start_coords.z - z[z_id]) + cap_x[z_id];
		current_cap_y = ((cap_y[z_id+1] - cap_y[z_id])/(z[z_id+1] - z[z_id])) * (photon->start_coords.z - z[z_id]) + cap_y[z_id];
	} else {
		current_cap_rad = cap[0];
		current_cap_x = cap_x[0];
		current_cap_y = cap_y[0];
	}
	d_ph_capcen = sqrt( (photon->start_coords.x-current_cap_x)*(photon->start_coords.x-current_cap_x) + (photon->start_coords.y-current_cap_y)*(photon->start_coords.y-current_cap_y) );
	if (d_ph_capcen > current_cap_rad) {
		if (photon->leak_calc) {
			/* :645-887 */
			int rc = orc_launch_in_wall_leak(optic, photon, cap_x, cap_y, ix);
			return rc;
		}
		/* :888-906 (leak_calc=false): photon hit the glass at the entrance */
		return 2;
	}

	/* :912-919 bounce loop, at most nmax+1 trace calls */
	for (i = 0; i <= nmax; i++) {
		iesc = orc_trace(optic, ix, photon, cap_x, cap_y);
		if (iesc != 1)
			break;
	}

	/* :947-954 */
	if ((iesc == -1) || (iesc == -3))
		return -1;
	if (iesc == 0)
		return 0;
	return 1;
}

static orc_vec3 v3(const double *p) { orc_vec3 v = {p[0], p[1], p[2]}; return v; }
static void st3(double *p, orc_vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

/* src/polycap-photon.c:98-136 (polycap_photon_new) + launch */
int orc_launch_one(const orc_optic *optic, size_t n_energies, const double *energies,
                   const double *amu, const double *scatf,
                   const double start_coords[3], const double start_dir[3], const double start_elecv[3],
                   double *weights, double exit_coords[3], double exit_dir[3], double exit_elecv[3],
                   int64_t *i_refl, double *d_travel)
{
	orc_photon ph;
	int rc;
	memset(&ph, 0, sizeof(ph));
	ph.start_coords = v3(start_coords);
	ph.exit_coords = ph.start_coords;
	ph.start_direction = v3(start_dir);
	ph.exit_direction = ph.start_direction;
	ph.start_electric_vector = v3(start_elecv);
	ph.exit_electric_vector = ph.start_electric_vector;
	ph.d_travel = 0;
	rc = orc_launch(optic, &ph, n_energies, energies, amu, scatf, weights);
	if (exit_coords) st3(exit_coords, ph.exit_coords);
	if (exit_dir) st3(exit_dir, ph.exit_direction);
	if (exit_elecv) st3(exit_elecv, ph.exit_electric_vector);
	if (i_refl) *i_refl = ph.i_refl;
	if (d_travel) *d_travel = ph.d_travel;
	return rc;
}

void orc_launch_batch(const orc_optic *optic, size_t n_energies, const double *energies,
                      const double *amu, const double *scatf, int64_t n,
                      const double *start_coords, const double *start_dir, const double *start_elecv,
                      int *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                      int64_t *i_refl, double *d_travel, int n_threads)
{
	int64_t j;
#ifdef _OPENMP
	if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 64)
#endif
	for (j = 0; j < n; j++) {
		rc[j] = orc_launch_one(optic, n_energies, energies, amu, scatf,
		                       start_coords + 3*j, start_dir + 3*j, start_elecv + 3*j,
		                       weights + (size_t)j*n_energies, exit_coords + 3*j, exit_dir + 3*j, exit_elecv + 3*j,
		                       i_refl + j, d_travel + j);
	}
	(void)n_threads;
}

/* ------------------------------------------------------------------ RNG */

/* Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11.
 * Replaces GSL/easyRNG mt19937 (src/polycap-rng.c:31-95); stream values are pinned by no reference test. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
	uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
	uint32_t k0 = key[0], k1 = key[1];
	int r;
	for (r = 0; r < 10; r++) {
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

/* counter = (slot_lo, slot_hi, attempt, d>>1), key = (seed_lo, seed_hi);
 * even d takes words (0,1), odd d words (2,3); u = (64-bit word >> 11) * 2^-53 in [0,1) */
double orc_uniform(uint64_t seed, uint64_t slot, uint32_t attempt, uint32_t d)
{
	uint32_t ctr[4] = { (uint32_t)slot, (uint32_t)(slot >> 32), attempt, d >> 1 };
	uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
	uint32_t o[4];
	uint64_t w;
	orc_philox4x32_10(ctr, key, o);
	w = (d & 1u) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
	return (double)(w >> 11) * (1.0/9007199254740992.0);
}

/* ------------------------------------------------------------------ source */

/* src/polycap-source.c:23-144 */
void orc_sample_photon(const orc_optic *optic, const orc_source *source,
                       uint64_t seed, uint64_t slot, uint32_t attempt, orc_photon *photon)
{
	double n_shells, r, phi, src_start_x, src_start_y, max_rad;
	double cosalpha, alpha, c_ae, c_be, frac_hor_pol;
	orc_vec3 start_coords, start_direction, start_electric_vector, src_start_coords;
	uint32_t d = 0;
	int boundary_check;

	/* :54-67 */
	r = orc_uniform(seed, slot, attempt, d++);
	phi = atan(source->src_y/source->src_x * tan(2.0*M_PI*r/4.));
	r = orc_uniform(seed, slot, attempt, d++);
	if ((r >= 0.25) && (r < 0.5))
		phi = M_PI - phi;
	if ((r >= 0.5) && (r < 0.75))
		phi = M_PI + phi;
	if (r >= 0.75)
		phi = -1.0 * phi;
	max_rad = source->src_x*source->src_y / sqrt((source->src_y*cos(phi))*(source->src_y*cos(phi)) + (source->src_x*sin(phi))*(source->src_x*sin(phi)));
	r = orc_uniform(seed, slot, attempt, d++);
	src_start_x = sqrt(r) * max_rad * cos(phi) + source->src_shiftx;
	src_start_y = sqrt(r) * max_rad * sin(phi) + source->src_shifty;
	src_start_coords.x = src_start_x;
	src_start_coords.y = src_start_y;
	src_start_coords.z = 0;

	if (source->src_sigx < 0. || source->src_sigy < 0.) {
		/* :74-96 uniform illumination of the entrance window */
		n_shells = orc_n_shells(optic->n_cap);
		if (n_shells == 0.) {
			r = orc_uniform(seed, slot, attempt, d++);
			start_coords.x = (2.*r-1.) * optic->cap[0];
			r = orc_uniform(seed, slot, attempt, d++);
			start_coords.y = (2.*r-1.) * optic->cap[0];
		} else {
			do {
				r = orc_uniform(seed, slot, attempt, d++);
				start_coords.x = (2.*r-1.) * optic->ext[0];
				r = orc_uniform(seed, slot, attempt, d++);
				start_coords.y = (2.*r-1.) * optic->ext[0];
				start_coords.z = 0.;
				boundary_check = orc_within_pc_boundary(optic->ext[0], start_coords);
			} while (boundary_check == 0);
		}
		start_direction.x = start_coords.x - src_start_x;
		start_direction.y = start_coords.y - src_start_y;
		start_direction.z = source->d_source;
	} else {
		/* :97-108 */
		r = orc_uniform(seed, slot, attempt, d++);
		start_direction.x = source->src_sigx * (1.-2.*fabs(r));
		r = orc_uniform(seed, slot, attempt, d++);
		start_direction.y = source->src_sigy * (1.-2.*fabs(r));
		start_direction.z = 1.;
		start_coords.x = src_start_coords.x + start_direction.x * source->d_source / start_direction.z;
		start_coords.y = src_start_coords.y + start_direction.y * source->d_source / start_direction.z;
	}
	start_coords.z = 0.;
	orc_norm(&start_direction);

	/* :114-126 */
	frac_hor_pol = (1. + source->hor_pol)/2.;
	r = orc_uniform(seed, slot, attempt, d++);
	if (fabs(r) <= frac_hor_pol) {
		start_electric_vector.x = 1.;
		start_electric_vector.y = 0.;
		start_electric_vector.z = 0.;
	} else {
		start_electric_vector.x = 0.;
		start_electric_vector.y = 1.;
		start_electric_vector.z = 0.;
	}
	/* :128-137 */
	cosalpha = orc_scalar(start_electric_vector, start_direction);
	alpha = acos(cosalpha);
	c_ae = 1./sin(alpha);
	c_be = -1.*c_ae*cosalpha;
	start_electric_vector.x = start_electric_vector.x * c_ae + start_direction.x * c_be;
	start_electric_vector.y = start_electric_vector.y * c_ae + start_direction.y * c_be;
	start_electric_vector.z = start_electric_vector.z * c_ae + start_direction.z * c_be;
	orc_norm(&start_electric_vector);

	/* :140-141 + polycap_photon_new src/polycap-photon.c:125-133 */
	memset(photon, 0, sizeof(*photon));
	photon->start_coords = start_coords;
	photon->exit_coords = start_coords;
	photon->start_direction = start_direction;
	photon->exit_direction = start_direction;
	photon->start_electric_vector = start_electric_vector;
	photon->exit_electric_vector = start_electric_vector;
	photon->d_travel = 0;
	photon->src_start_coords = src_start_coords;
}

void orc_sample_photon_flat(const orc_optic *optic, const orc_source *source,
                            uint64_t seed, uint64_t slot, uint32_t attempt, double out[12])
{
	orc_photon ph;
	orc_sample_photon(optic, source, seed, slot, attempt, &ph);
	st3(out, ph.start_coords);
	st3(out + 3, ph.start_direction);
	st3(out + 6, ph.start_electric_vector);
	st3(out + 9, ph.src_start_coords);
}

/* ------------------------------------------------------------------ driver */

/* src/polycap-source.c:744-966 for one exit-photon slot j; returns attempts used, or 0 if max_attempts ran out.
 * cnt[0..3] += {iexit, not_entered, not_transmitted, i_refl of the transmitted photon} */
uint32_t orc_one_slot(const orc_optic *optic, const orc_source *source,
                      size_t n_energies, const double *energies, const double *amu, const double *scatf,
                      uint64_t seed, int64_t j, uint32_t max_attempts,
                      double *w, int64_t cnt[4], double *img, orc_slot_leaks *sl)
{
	const double *z = optic->z, *ext = optic->ext;
	const int nmax = optic->nmax;
	orc_photon ph;
	orc_vec3 temp_vect;
	int iesc;
	uint32_t k;
	double cosalpha, alpha, c_ae, c_be;
	double ex, ey, ez;

	for (k = 0; k < max_attempts; k++) {
		/* :748-750 */
		orc_sample_photon(optic, source, seed, (uint64_t)j, k, &ph);
		ph.leak_calc = (sl != NULL);
		iesc = orc_launch(optic, &ph, n_energies, energies, amu, scatf, w);
		/* :758-777 */
		if (iesc == 0) cnt[2]++;
		if (iesc == 2) cnt[1]++;
		if (iesc == 1) {
			temp_vect.x = ph.exit_coords.x + ph.exit_direction.x * (z[nmax] - ph.exit_coords.z)/ph.exit_direction.z;
			temp_vect.y = ph.exit_coords.y + ph.exit_direction.y * (z[nmax] - ph.exit_coords.z)/ph.exit_direction.z;
			temp_vect.z = z[nmax];
			if (orc_n_shells(optic->n_cap) == 0.) {
				if (sqrt((temp_vect.x)*(temp_vect.x) + (temp_vect.y)*(temp_vect.y)) > ext[nmax])
					iesc = 0;
				else
					iesc = 1;
			} else {
				iesc = orc_within_pc_boundary(ext[nmax], temp_vect);
			}
		}
		if (sl != NULL)
			orc_slot_leaks_collect(sl, &ph, iesc, k);   /* :799-879, polycap_oracle_leak.c */
		if (iesc == 1)
			break;
	}
	if (k == max_attempts)
		return 0;

	/* :779-798, 900-923 */
	cnt[0]++;
	cnt[3] += ph.i_refl;
	if (img) {
		img[0] = ph.src_start_coords.x;
		img[1] = ph.src_start_coords.y;
		img[2] = ph.start_coords.x;
		img[3] = ph.start_coords.y;
		img[4] = ph.start_direction.x;
		img[5] = ph.start_direction.y;
		cosalpha = orc_scalar(ph.start_electric_vector, ph.start_direction);
		alpha = acos(cosalpha);
		c_ae = 1./sin(alpha);
		c_be = -1.*c_ae*cosalpha;
		temp_vect.x = ph.start_electric_vector.x * c_ae + ph.start_direction.x * c_be;
		temp_vect.y = ph.start_electric_vector.y * c_ae + ph.start_direction.y * c_be;
		temp_vect.z = ph.start_electric_vector.z * c_ae + ph.start_direction.z * c_be;
		orc_norm(&temp_vect);
		img[6] = round(temp_vect.x);
		img[7] = round(temp_vect.y);
		ex = ph.exit_coords.x + ph.exit_direction.x*(z[nmax] - ph.exit_coords.z)/ph.exit_direction.z;
		ey = ph.exit_coords.y + ph.exit_direction.y*(z[nmax] - ph.exit_coords.z)/ph.exit_direction.z;
		ez = ph.exit_coords.z + ph.exit_direction.z*(z[nmax] - ph.exit_coords.z)/ph.exit_direction.z;
		img[8] = ex;
		img[9] = ey;
		img[10] = ez;
		img[11] = ph.exit_direction.x;
		img[12] = ph.exit_direction.y;
		temp_vect.x = ph.exit_electric_vector.x * c_ae + ph.exit_direction.x * c_be;
		temp_vect.y = ph.exit_electric_vector.y * c_ae + ph.exit_direction.y * c_be;
		temp_vect.z = ph.exit_electric_vector.z * c_ae + ph.exit_direction.z * c_be;
		orc_norm(&temp_vect);
		img[13] = round(temp_vect.x);
		img[14] = round(temp_vect.y);
		img[15] = (double)ph.i_refl;
		img[16] = ph.d_travel + sqrt( (ex - ph.exit_coords.x)*(ex - ph.exit_coords.x) +
			(ey - ph.exit_coords.y)*(ey - ph.exit_coords.y) +
			(z[nmax] - ph.exit_coords.z)*(z[nmax] - ph.exit_coords.z));
	}
	return k + 1;
}

/* src/polycap-source.c:448-1087 restricted to leak_calc=false and slots [slot0, slot0+n_slots) */
int orc_transmission(const orc_optic *optic, const orc_source *source,
                     size_t n_energies, const double *energies, const double *amu, const double *scatf,
                     uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                     double *sum_weights, int64_t counters[4], double *img, double *exit_weights)
{
	return orc_transmission_fixed(optic, source, n_energies, energies, amu, scatf, seed, slot0, n_slots, n_threads,
	                              max_attempts, sum_weights, counters, img, exit_weights, NULL);
}

/* The same driver; sumw_fixed (2 x n_energies words, may be NULL) also receives sum_j floor(w_j * 2^62) per energy as
 * an exact 128-bit integer (lo, hi) -- the representation the GPU path accumulates in, so that totals of any number of
 * slots can be compared, split and added without a rounding that depends on the order of the sum. */
int orc_transmission_fixed(const orc_optic *optic, const orc_source *source,
                     size_t n_energies, const double *energies, const double *amu, const double *scatf,
                     uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                     double *sum_weights, int64_t counters[4], double *img, double *exit_weights, uint64_t *sumw_fixed)
{
	int failed = 0;
	size_t e;
	int64_t c;
	for (e = 0; e < n_energies; e++) sum_weights[e] = 0.;
	for (c = 0; c < 4; c++) counters[c] = 0;
#ifdef _OPENMP
	if (n_threads < 1) n_threads = omp_get_max_threads();
#else
	n_threads = 1;
#endif
	{
		/* per-thread partials combined in thread order (the reference uses an omp critical, :973-980) */
		double *part_w = calloc((size_t)n_threads * n_energies, sizeof(double));
		int64_t *part_c = calloc((size_t)n_threads * 4, sizeof(int64_t));
		unsigned __int128 *part_f = calloc((size_t)n_threads * n_energies, sizeof(unsigned __int128));
		int t;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
		{
#ifdef _OPENMP
			int tid = omp_get_thread_num();
#else
			int tid = 0;
#endif
			double *w = malloc(sizeof(double) * n_energies);
			double *pw = part_w + (size_t)tid * n_energies;
			unsigned __int128 *pf = part_f + (size_t)tid * n_energies;
			int64_t *pc = part_c + (size_t)tid * 4;
			int64_t j;
			size_t ee;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
			for (j = 0; j < n_slots; j++) {
				uint32_t used = orc_one_slot(optic, source, n_energies, energies, amu, scatf, seed, slot0 + j,
				                             max_attempts, w, pc, img ? img + 17*(size_t)j : NULL, NULL);
				if (used == 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
					failed = 1;
					if (exit_weights)
						for (ee = 0; ee < n_energies; ee++) exit_weights[(size_t)j*n_energies + ee] = 0.;
					continue;
				}
				for (ee = 0; ee < n_energies; ee++) {
					pw[ee] += w[ee];
					pf[ee] += (unsigned __int128)(uint64_t)(w[ee] * 4611686018427387904.0);
					if (exit_weights) exit_weights[(size_t)j*n_energies + ee] = w[ee];
				}
			}
			free(w);
		}
		for (t = 0; t < n_threads; t++) {
			for (e = 0; e < n_energies; e++) sum_weights[e] += part_w[(size_t)t*n_energies + e];
			for (c = 0; c < 4; c++) counters[c] += part_c[(size_t)t*4 + c];
		}
		if (sumw_fixed) {
			for (e = 0; e < n_energies; e++) {
				unsigned __int128 tot = 0;
				for (t = 0; t < n_threads; t++) tot += part_f[(size_t)t*n_energies + e];
				sumw_fixed[2*e] = (uint64_t)tot;
				sumw_fixed[2*e + 1] = (uint64_t)(tot >> 64);
			}
		}
		free(part_w);
		free(part_c);
		free(part_f);
	}
	return failed ? -1 : 0;
}

/* src/polycap-source.c:1066-1076: open_area_sim = (iexit+not_trans)/i_start; eff = sum_w/(iexit+not_trans)*open_area_sim */
void orc_efficiencies(size_t n_energies, const double *sum_weights, const int64_t counters[4], double *eff)
{
	size_t i;
	int64_t sum_iexit = counters[0], sum_not_entered = counters[1], sum_not_transmitted = counters[2];
	double open_area = (double)(sum_iexit+sum_not_transmitted)/(sum_iexit+sum_not_entered+sum_not_transmitted);
	for (i = 0; i < n_energies; i++)
		eff[i] = (sum_weights[i] / ((double)sum_iexit+(double)sum_not_transmitted)) * open_area;
}
