/*
 * pc_profile.c -- polycap_profile: the (z, cap, ext) tables that describe the optic's shape.
 *
 * Host-side boundary glue of the trace path; behaviour follows the reference's src/polycap-profile.c:
 *   polycap_profile_new             :66-207   analytic conical / paraboloidal / ellipsoidal shapes, 1000 points
 *   polycap_profile_new_from_file   :211-317  three ASCII tables (capillary radius, central axis, exterior)
 *   polycap_profile_validate        :321-423  outer-shell capillaries must stay inside the exterior
 *   polycap_profile_new_from_arrays :426-476, getters :479-520, free :523-533
 * The paraboloidal exterior is a least-squares parabola through four points; the reference calls GSL's
 * multifit for it (:24-62), here it is a 4x3 Householder QR.
 */
#include "pc_private.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* least-squares solution of the 4x3 system sum_j coeff[j]*x^j ~ y (Householder QR, no pivoting) */
static void pc_fit_parabola(const double x[4], const double y[4], double coeff[3])
{
	double a[4][3], b[4];
	for (int i = 0; i < 4; i++) {
		a[i][0] = 1.0;
		a[i][1] = x[i];
		a[i][2] = x[i]*x[i];
		b[i] = y[i];
	}
	for (int k = 0; k < 3; k++) {
		double norm = 0.0;
		for (int i = k; i < 4; i++)
			norm += a[i][k]*a[i][k];
		norm = sqrt(norm);
		if (norm == 0.0)
			continue;
		double alpha = (a[k][k] > 0.0) ? -norm : norm;
		double v[4] = {0.0, 0.0, 0.0, 0.0};
		v[k] = a[k][k] - alpha;
		for (int i = k + 1; i < 4; i++)
			v[i] = a[i][k];
		double vtv = 0.0;
		for (int i = k; i < 4; i++)
			vtv += v[i]*v[i];
		if (vtv == 0.0)
			continue;
		for (int j = k; j < 3; j++) {
			double dot = 0.0;
			for (int i = k; i < 4; i++)
				dot += v[i]*a[i][j];
			double f = 2.0*dot/vtv;
			for (int i = k; i < 4; i++)
				a[i][j] -= f*v[i];
		}
		double dot = 0.0;
		for (int i = k; i < 4; i++)
			dot += v[i]*b[i];
		double f = 2.0*dot/vtv;
		for (int i = k; i < 4; i++)
			b[i] -= f*v[i];
	}
	for (int k = 2; k >= 0; k--) {
		double s = b[k];
		for (int j = k + 1; j < 3; j++)
			s -= a[k][j]*coeff[j];
		coeff[k] = s/a[k][k];
	}
}

static polycap_profile *pc_profile_alloc(int nmax, const char *caller, polycap_error **error)
{
	polycap_profile *profile = calloc(1, sizeof(polycap_profile));
	if (profile != NULL) {
		profile->nmax = nmax;
		profile->z = malloc(sizeof(double)*((size_t)nmax + 1));
		profile->cap = malloc(sizeof(double)*((size_t)nmax + 1));
		profile->ext = malloc(sizeof(double)*((size_t)nmax + 1));
	}
	if (profile == NULL || profile->z == NULL || profile->cap == NULL || profile->ext == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for profile -> %s", caller, strerror(errno));
		polycap_profile_free(profile);
		return NULL;
	}
	return profile;
}

polycap_profile *polycap_profile_new(polycap_profile_type type, double length, double rad_ext_upstream, double rad_ext_downstream,
	double rad_int_upstream, double rad_int_downstream, double focal_dist_upstream, double focal_dist_downstream, polycap_error **error)
{
	const int nmax = 999;

	/* argument checks and messages of the reference, src/polycap-profile.c:75-110 */
	if (length <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: length must be greater than 0.0");
		return NULL;
	}
	if (rad_ext_upstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_ext_upstream must be greater than 0.0");
		return NULL;
	}
	if (rad_ext_downstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_ext_downstream must be greater than 0.0");
		return NULL;
	}
	if (rad_int_upstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_int_upstream must be greater than 0.0");
		return NULL;
	}
	if (rad_int_downstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_int_downstream must be greater than 0.0");
		return NULL;
	}
	if (rad_int_upstream >= rad_ext_upstream) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_ext_upstream must be greater than rad_int_upstream");
		return NULL;
	}
	if (rad_int_downstream >= rad_ext_downstream) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: rad_ext_downstream must be greater than rad_int_downstream");
		return NULL;
	}
	if (focal_dist_upstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: focal_dist_upstream must be greater than 0.0");
		return NULL;
	}
	if (focal_dist_downstream <= 0.0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: focal_dist_downstream must be greater than 0.0");
		return NULL;
	}
	if (type != POLYCAP_PROFILE_CONICAL && type != POLYCAP_PROFILE_PARABOLOIDAL && type != POLYCAP_PROFILE_ELLIPSOIDAL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new: invalid profile type detected");
		return NULL;
	}

	polycap_profile *profile = pc_profile_alloc(nmax, "polycap_profile_new", error);
	if (profile == NULL)
		return NULL;
	double *z = profile->z, *cap = profile->cap, *ext = profile->ext;

	/* the single capillary is always conical (:145,165,177,187); z runs from 0 to length */
	for (int i = 0; i <= nmax; i++) {
		z[i] = length/nmax*i;
		cap[i] = (rad_int_downstream-rad_int_upstream)/length*z[i] + rad_int_upstream;
	}

	if (type == POLYCAP_PROFILE_CONICAL) {
		for (int i = 0; i <= nmax; i++)
			ext[i] = (rad_ext_downstream-rad_ext_upstream)/length*z[i] + rad_ext_upstream;
	} else if (type == POLYCAP_PROFILE_PARABOLOIDAL) {
		/* :149-169: entrance and exit points plus one point on each focal line */
		double px[4], py[4], coeff[3];
		px[0] = 0.;
		py[0] = rad_ext_upstream;
		px[3] = length;
		py[3] = rad_ext_downstream;
		px[1] = (focal_dist_upstream <= length) ? focal_dist_upstream/10. : length/10.;
		py[1] = (rad_ext_upstream-0.)/(0.-(-1.*focal_dist_upstream)) * (px[1] - 0.) + rad_ext_upstream;
		px[2] = (focal_dist_downstream <= length) ? length-focal_dist_downstream/10. : length-length/10.;
		py[2] = (rad_ext_downstream-0.)/(length-(length+focal_dist_downstream)) * (px[2] - length) + rad_ext_downstream;
		pc_fit_parabola(px, py, coeff);
		for (int i = 0; i <= nmax; i++)
			ext[i] = coeff[0]+coeff[1]*z[i]+coeff[2]*z[i]*z[i];
	} else {
		/* :171-196 quarter ellipse: horizontal tangent at the wide end, pointing at the focus on the narrow end */
		double slope, b, k, a;
		if (rad_ext_downstream < rad_ext_upstream) {
			slope = rad_ext_downstream / focal_dist_downstream;
			b = (-1.*(rad_ext_downstream-rad_ext_upstream)*(rad_ext_downstream-rad_ext_upstream)-slope*length*(rad_ext_downstream-rad_ext_upstream)) / (slope*length+2.*(rad_ext_downstream-rad_ext_upstream));
			k = rad_ext_upstream - b;
			a = sqrt((b*b*length)/(slope*(rad_ext_downstream-k)));
			for (int i = 0; i <= nmax; i++)
				ext[i] = sqrt(b*b-(b*b*z[i]*z[i])/(a*a))+k;
		} else {
			slope = rad_ext_upstream / focal_dist_upstream;
			b = (-1.*(rad_ext_upstream-rad_ext_downstream)*(rad_ext_upstream-rad_ext_downstream)-slope*length*(rad_ext_upstream-rad_ext_downstream)) / (slope*length+2.*(rad_ext_upstream-rad_ext_downstream));
			k = rad_ext_downstream - b;
			a = sqrt(fabs((b*b*length)/(slope*(rad_ext_upstream-k))));
			for (int i = 0; i <= nmax; i++)
				ext[i] = sqrt(b*b-(b*b*z[nmax-i]*z[nmax-i])/(a*a))+k;
		}
	}
	return profile;
}

/* reads "n" then n+1 rows of `cols` numbers; column 0 -> z, column `keep` -> out (may be NULL) */
static int pc_read_table(FILE *fptr, int nmax, int cols, int keep, double *z, double *out)
{
	for (int i = 0; i <= nmax; i++) {
		double row[3] = {0., 0., 0.};
		for (int c = 0; c < cols; c++)
			if (fscanf(fptr, "%lf", &row[c]) != 1)
				return -1;
		z[i] = row[0];
		if (out != NULL)
			out[i] = row[keep];
	}
	return 0;
}

polycap_profile *polycap_profile_new_from_file(const char *single_cap_profile_file, const char *central_axis_file, const char *external_shape_file, polycap_error **error)
{
	if (single_cap_profile_file == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_file: single_cap_profile_file cannot be NULL");
		return NULL;
	}
	if (central_axis_file == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_file: central_axis_file cannot be NULL");
		return NULL;
	}
	if (external_shape_file == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_file: external_shape_file cannot be NULL");
		return NULL;
	}

	int n_tmp = 0;
	FILE *fptr = fopen(single_cap_profile_file, "r");
	if (fptr == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_profile_new_from_file: could not open %s -> %s", single_cap_profile_file, strerror(errno));
		return NULL;
	}
	if (fscanf(fptr, "%d", &n_tmp) != 1 || n_tmp <= 100) {
		fclose(fptr);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_file: n_tmp must be greater than 100");
		return NULL;
	}
	polycap_profile *profile = pc_profile_alloc(n_tmp, "polycap_profile_new_from_file", error);
	if (profile == NULL) {
		fclose(fptr);
		return NULL;
	}
	if (pc_read_table(fptr, profile->nmax, 2, 1, profile->z, profile->cap) != 0) {
		fclose(fptr);
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_profile_new_from_file: could not read %d rows from %s", n_tmp + 1, single_cap_profile_file);
		polycap_profile_free(profile);
		return NULL;
	}
	fclose(fptr);

	/* the central-axis table only contributes its z column (reference :288-290 discards sx, sy) */
	const char *names[2] = { central_axis_file, external_shape_file };
	for (int f = 0; f < 2; f++) {
		fptr = fopen(names[f], "r");
		if (fptr == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_profile_new_from_file: could not open %s -> %s", names[f], strerror(errno));
			polycap_profile_free(profile);
			return NULL;
		}
		if (fscanf(fptr, "%d", &n_tmp) != 1 || profile->nmax != n_tmp) {
			fclose(fptr);
			polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_profile_new_from_file: Number of intervals inconsistent: %s", names[f]);
			polycap_profile_free(profile);
			return NULL;
		}
		int bad = (f == 0) ? pc_read_table(fptr, profile->nmax, 3, 0, profile->z, NULL)
		                   : pc_read_table(fptr, profile->nmax, 2, 1, profile->z, profile->ext);
		fclose(fptr);
		if (bad) {
			polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_profile_new_from_file: could not read %d rows from %s", n_tmp + 1, names[f]);
			polycap_profile_free(profile);
			return NULL;
		}
	}
	return profile;
}

double pc_n_shells(int64_t n_cap)
{
	return round(sqrt(12. * n_cap - 3.)/6.-0.5);
}

/* hexagon test, reference src/polycap-photon.c:139-169: 1 inside, 0 outside, -1 invalid radius */
int polycap_photon_within_pc_boundary(double polycap_radius, polycap_vector3 photon_coord, polycap_error **error)
{
	if (polycap_radius <= 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_within_pc_boundary: polycap_radius must be greater than 0");
		return -1;
	}
	double d_cen2hexedge = sqrt((polycap_radius * polycap_radius) - ((polycap_radius/2.) * (polycap_radius/2.)));
	double dp1 = fabs(photon_coord.y);
	double dp2 = fabs(PC_COSPI_6*photon_coord.x + 0.5*photon_coord.y);
	double dp3 = fabs(PC_COSPI_6*photon_coord.x - 0.5*photon_coord.y);
	if (dp1 > d_cen2hexedge || dp2 > d_cen2hexedge || dp3 > d_cen2hexedge)
		return 0;
	return 1;
}

/* 1 = feasible, 0 = some capillary pokes out of the exterior, -1 = error (reference :321-423, full_check branch) */
int polycap_profile_validate(polycap_profile *profile, int64_t n_cap, polycap_error **error)
{
	if (profile == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_validate: profile cannot be NULL");
		return -1;
	}
	double n_shells = pc_n_shells(n_cap);
	if (n_shells == 0) {
		for (int i = 0; i <= profile->nmax; i++)
			if (profile->cap[i] >= profile->ext[i])
				return 0;
		return 1;
	}
	/* walk the outermost shell: start at the (-n, n) corner, n steps along each of the six edges */
	static const int q_dir[6] = {1, 1, 0, -1, -1, 0};
	static const int r_dir[6] = {0, -1, -1, 0, 1, 1};
	double q_i = -1.*n_shells, r_i = n_shells;
	for (int j = 0; j < 6; j++) {
		for (int k = 0; k < n_shells; k++) {
			q_i += q_dir[j];
			r_i += r_dir[j];
			for (int i = 0; i <= profile->nmax; i++) {
				polycap_vector3 coord;
				double zz = profile->ext[i]/(2.*PC_COSPI_6*(n_shells+1));
				coord.y = r_i * (3./2) * zz;
				coord.x = (2* q_i + r_i) * PC_COSPI_6 * zz;
				double angle = atan(coord.y/coord.x);
				coord.x += cos(angle)*profile->cap[i];
				coord.y += sin(angle)*profile->cap[i];
				coord.z = profile->z[i];
				int check = polycap_photon_within_pc_boundary(profile->ext[i], coord, error);
				if (check == 0)
					return 0;
				if (check == -1)
					return -1;
			}
		}
	}
	return 1;
}

polycap_profile *polycap_profile_new_from_arrays(int nid, double *ext, double *cap, double *z, polycap_error **error)
{
	if (ext == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_array: ext cannot be NULL");
		return NULL;
	}
	if (cap == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_array: cap cannot be NULL");
		return NULL;
	}
	if (z == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_array: z cannot be NULL");
		return NULL;
	}
	if (nid <= 1) {
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_profile_new_from_array: nid must be greater than 1");
		return NULL;
	}
	polycap_profile *profile = pc_profile_alloc(nid, "polycap_profile_new_from_array", error);
	if (profile == NULL)
		return NULL;
	memcpy(profile->ext, ext, sizeof(double) * ((size_t)nid + 1));
	memcpy(profile->cap, cap, sizeof(double) * ((size_t)nid + 1));
	memcpy(profile->z, z, sizeof(double) * ((size_t)nid + 1));
	return profile;
}

static bool pc_profile_get(polycap_profile *profile, const double *src, size_t *nid, double **out)
{
	if (profile == NULL || nid == NULL || out == NULL)
		return false;
	*nid = (size_t)profile->nmax;
	*out = malloc(sizeof(double) * ((size_t)profile->nmax + 1));
	if (*out == NULL)
		return false;
	memcpy(*out, src, sizeof(double) * ((size_t)profile->nmax + 1));
	return true;
}

bool polycap_profile_get_ext(polycap_profile *profile, size_t *nid, double **ext, polycap_error **error)
{
	(void)error;
	return pc_profile_get(profile, profile ? profile->ext : NULL, nid, ext);
}

bool polycap_profile_get_cap(polycap_profile *profile, size_t *nid, double **cap, polycap_error **error)
{
	(void)error;
	return pc_profile_get(profile, profile ? profile->cap : NULL, nid, cap);
}

bool polycap_profile_get_z(polycap_profile *profile, size_t *nid, double **z, polycap_error **error)
{
	(void)error;
	return pc_profile_get(profile, profile ? profile->z : NULL, nid, z);
}

void polycap_profile_free(polycap_profile *profile)
{
	if (profile == NULL)
		return;
	free(profile->z);
	free(profile->cap);
	free(profile->ext);
	free(profile);
}
