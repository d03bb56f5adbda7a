/*
 * polycap_oracle_leak.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see polycap_oracle.h).
 *
 * Plain-C fp64 restatement of the reference's leak ("halo") path, leak_calc=true:
 *   polycap_capil_trace_wall            src/polycap-capil.c:893-1194
 *   leak branch of polycap_capil_reflect src/polycap-capil.c:565-891
 *   polycap_photon_pc_intersect         src/polycap-photon.c:171-362
 *   entrance-wall branch of launch      src/polycap-photon.c:645-907
 *   leak bookkeeping of the driver      src/polycap-source.c:799-879, 925-1032
 * Same statement order as the reference (including its quirks, which are noted where they matter), so that the
 * reference's own known answers (tests/leaks.c) pin it: tests/test_oracle_leak_known_answers.py.
 */
#include "polycap_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_COSPI_6 0.86602540378443864676

/* ------------------------------------------------------------------ leak lists (src/polycap-photon.c:1015-1035) */

static void leak_append(orc_leak **list, int64_t *n, orc_vec3 coords, orc_vec3 dir, orc_vec3 elecv, int64_t n_refl,
                        size_t n_energies, const double *weights)
{
	*list = realloc(*list, sizeof(orc_leak) * (size_t)(*n + 1));
	orc_leak *l = &(*list)[*n];
	l->coords = coords;
	l->direction = dir;
	l->elecv = elecv;
	l->n_refl = n_refl;
	l->n_energies = n_energies;
	l->weight = malloc(sizeof(double) * n_energies);
	memcpy(l->weight, weights, sizeof(double) * n_energies);
	(*n)++;
}

void orc_leaks_free(orc_leak *list, int64_t n)
{
	for (int64_t k = 0; k < n; k++)
		free(list[k].weight);
	free(list);
}

void orc_photon_clear_leaks(orc_photon *photon)
{
	orc_leaks_free(photon->extleak, photon->n_extleak);
	orc_leaks_free(photon->intleak, photon->n_intleak);
	photon->extleak = photon->intleak = NULL;
	photon->n_extleak = photon->n_intleak = 0;
}

/* hexagonal (q, r) index of the capillary region containing (x, y): cube rounding, src/polycap-capil.c:956-969 */
static void hex_index(double x, double y, double zz, double *q_out, double *r_out)
{
	double r_i = y * (2./3) / zz;
	double q_i = (x/(2.*ORC_COSPI_6) - y/3) / zz;
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
	*q_out = q_i;
	*r_out = r_i;
}

/* ------------------------------------------------------------------ src/polycap-photon.c:171-362
 * Intersection of the photon path with the outer (hexagonal) wall of the optic, searched backwards from a point
 * outside it.  Returns 1 and *out, or 0 where the reference returns NULL. */
int orc_pc_intersect(const orc_optic *optic, orc_vec3 photon_coord, orc_vec3 photon_direction, orc_vec3 *out)
{
	const double *z = optic->z, *ext = optic->ext;
	const int nmax = optic->nmax;
	double hex_edge_norm1[2], hex_edge_norm2[2], hex_edge_norm3[2];
	double d_hexcen_beg, d_hexcen_end;
	double dp1b, dp2b, dp3b, dp1e, dp2e, dp3e;
	orc_vec3 phot_temp, phot_dir, phot_beg, phot_end;
	int i, z_id = 0, dir, broke = 0;
	double current_polycap_ext;
	double z1, z2, z3, z_fin;

	if (photon_direction.z == 0.)
		return 0;
	hex_edge_norm1[0] = 0;
	hex_edge_norm1[1] = 1;
	hex_edge_norm2[0] = ORC_COSPI_6;
	hex_edge_norm2[1] = 0.5;
	hex_edge_norm3[0] = ORC_COSPI_6;
	hex_edge_norm3[1] = -0.5;

	/* :201-205 */
	phot_dir.x = -1.*photon_direction.x;
	phot_dir.y = -1.*photon_direction.y;
	phot_dir.z = -1.*photon_direction.z;
	orc_norm(&phot_dir);

	/* :208-218 */
	for (i = 0; i < nmax; i++) {
		if (z[i] <= photon_coord.z)
			z_id = i;
	}
	current_polycap_ext = (ext[z_id+1]-ext[z_id])/(z[z_id+1]-z[z_id]) * (photon_coord.z - z[z_id]) + ext[z_id];
	if (orc_within_pc_boundary(current_polycap_ext, photon_coord) == 1)
		return 0;
	if (phot_dir.z < 0.) {
		z_id = z_id+1;
		dir = -1;
	} else {
		dir = 1;
	}
	/* :226-236.  The reference's loop condition lets z_id step to -1 or nmax+1 and reads the profile there (out of
	 * bounds); a search that runs off either end finds nothing, so it stops here before the invalid read. */
	do {
		z_id += dir;
		if (z_id < 0 || z_id > nmax)
			break;
		phot_temp.x = photon_coord.x + phot_dir.x * (z[z_id]-photon_coord.z)/phot_dir.z;
		phot_temp.y = photon_coord.y + phot_dir.y * (z[z_id]-photon_coord.z)/phot_dir.z;
		phot_temp.z = z[z_id];
		if (orc_within_pc_boundary(current_polycap_ext, photon_coord) != orc_within_pc_boundary(ext[z_id], phot_temp)) {
			broke = 1;
			break;
		}
	} while (z_id >= 0 && z_id <= nmax);
	if (broke == 0)
		return 0;
	if (z_id - dir < 0 || z_id - dir > nmax)
		return 0;

	/* :244-249 */
	phot_beg.x = photon_coord.x + phot_dir.x * (z[z_id]-photon_coord.z)/phot_dir.z;
	phot_beg.y = photon_coord.y + phot_dir.y * (z[z_id]-photon_coord.z)/phot_dir.z;
	phot_beg.z = photon_coord.z + phot_dir.z * (z[z_id]-photon_coord.z)/phot_dir.z;
	phot_end.x = photon_coord.x + phot_dir.x * (z[z_id-dir]-photon_coord.z)/phot_dir.z;
	phot_end.y = photon_coord.y + phot_dir.y * (z[z_id-dir]-photon_coord.z)/phot_dir.z;
	phot_end.z = photon_coord.z + phot_dir.z * (z[z_id-dir]-photon_coord.z)/phot_dir.z;

	/* :252-259 */
	d_hexcen_beg = sqrt((ext[z_id] * ext[z_id]) - ((ext[z_id]/2.) * (ext[z_id]/2.)));
	d_hexcen_end = sqrt((ext[z_id-dir] * ext[z_id-dir]) - ((ext[z_id-dir]/2.) * (ext[z_id-dir]/2.)));
	dp1b = fabs(hex_edge_norm1[0]*phot_beg.x + hex_edge_norm1[1]*phot_beg.y);
	dp2b = fabs(hex_edge_norm2[0]*phot_beg.x + hex_edge_norm2[1]*phot_beg.y);
	dp3b = fabs(hex_edge_norm3[0]*phot_beg.x + hex_edge_norm3[1]*phot_beg.y);
	dp1e = fabs(hex_edge_norm1[0]*phot_end.x + hex_edge_norm1[1]*phot_end.y);
	dp2e = fabs(hex_edge_norm2[0]*phot_end.x + hex_edge_norm2[1]*phot_end.y);
	dp3e = fabs(hex_edge_norm3[0]*phot_end.x + hex_edge_norm3[1]*phot_end.y);

	/* :262-264 (interpolates between the two EXT values, compared against z below: kept as written) */
	z1 = (dp1b - d_hexcen_beg) / (d_hexcen_beg-d_hexcen_end - dp1b+dp1e) * (ext[z_id]-ext[z_id-dir]) + ext[z_id];
	z2 = (dp2b - d_hexcen_beg) / (d_hexcen_beg-d_hexcen_end - dp2b+dp2e) * (ext[z_id]-ext[z_id-dir]) + ext[z_id];
	z3 = (dp3b - d_hexcen_beg) / (d_hexcen_beg-d_hexcen_end - dp3b+dp3e) * (ext[z_id]-ext[z_id-dir]) + ext[z_id];

	/* :267-347: among the solutions inside the segment take the one closest to its far end */
	{
		const double lo = (dir < 0) ? z[z_id] : z[z_id-dir];
		const double hi = (dir < 0) ? z[z_id-dir] : z[z_id];
		const int v1 = (z1 >= lo && z1 <= hi), v2 = (z2 >= lo && z2 <= hi), v3 = (z3 >= lo && z3 <= hi);
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
			else { *out = phot_end; return 1; }
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
			else { *out = phot_end; return 1; }
		}
	}
	/* :349-351 */
	phot_temp.x = photon_coord.x + phot_dir.x * (z_fin-photon_coord.z)/phot_dir.z;
	phot_temp.y = photon_coord.y + phot_dir.y * (z_fin-photon_coord.z)/phot_dir.z;
	phot_temp.z = photon_coord.z + phot_dir.z * (z_fin-photon_coord.z)/phot_dir.z;
	*out = phot_temp;
	return 1;
}

/* ------------------------------------------------------------------ src/polycap-capil.c:893-1194
 * From the last interaction point through the glass: 1 entered capillary (q_cntr, r_cntr) after d_travel,
 * 2 reached the exit plane inside the glass, 3 left the optic through its side, <= 0 nothing to trace. */
int orc_trace_wall(const orc_optic *optic, orc_photon *photon, double *d_travel, int *r_cntr, int *q_cntr)
{
	const double *z = optic->z, *cap = optic->cap, *ext = optic->ext;
	const int nmax = optic->nmax;
	int i, photon_pos_check = 0, iesc = 0;
	int z_id = 0;
	double current_polycap_ext = 0;
	orc_vec3 photon_coord_rel;
	double n_shells;
	double r_i, q_i, zz;
	orc_vec3 cap_coord0, cap_coord1, phot_coord0, phot_coord1, temp_phot, phot_inter;
	double rad0, rad1;
	orc_vec3 interact_coords, surface_norm = {0., 0., 0.};
	double q_new = 0, r_new = 0;
	double d_phot0;
	double dist = 0;

	*d_travel = 0.;
	*r_cntr = 0;
	*q_cntr = 0;

	orc_norm(&photon->exit_direction);
	/* :918-932 */
	if (photon->exit_coords.z >= z[nmax])
		return -2;
	for (i = 0; i < nmax; i++) {
		if (z[i] <= photon->exit_coords.z)
			z_id = i;
	}
	if (z[z_id] != photon->exit_coords.z) {
		current_polycap_ext = ((ext[z_id+1] - ext[z_id])/(z[z_id+1] - z[z_id])) * (photon->exit_coords.z - z[z_id]) + ext[z_id];
	} else {
		current_polycap_ext = ext[z_id];
	}
	interact_coords = photon->exit_coords;

	/* :942-956 */
	n_shells = orc_n_shells(optic->n_cap);
	if (n_shells == 0.) {
		if (sqrt((photon->exit_coords.x)*(photon->exit_coords.x) + (photon->exit_coords.y)*(photon->exit_coords.y)) > current_polycap_ext)
			return -2;
	} else {
		photon_pos_check = orc_within_pc_boundary(current_polycap_ext, photon->exit_coords);
		if (photon_pos_check == 0)
			return -2;
	}

	/* :959-971 */
	zz = current_polycap_ext/(2.*ORC_COSPI_6*(n_shells+1));
	hex_index(photon->exit_coords.x, photon->exit_coords.y, zz, &q_i, &r_i);

	if (n_shells == 0.) {
		/* :991-1011 mono-capillary */
		iesc = 0;
		do {
			rad0 = cap[z_id];
			rad1 = cap[z_id+1];
			phot_coord0.x = photon->exit_coords.x + photon->exit_direction.x * (z[z_id]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord0.y = photon->exit_coords.y + photon->exit_direction.y * (z[z_id]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord0.z = z[z_id];
			phot_coord1.x = photon->exit_coords.x + photon->exit_direction.x * (z[z_id+1]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord1.y = photon->exit_coords.y + photon->exit_direction.y * (z[z_id+1]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord1.z = z[z_id+1];
			cap_coord0.x = 0.;
			cap_coord0.y = 0.;
			cap_coord0.z = z[z_id];
			cap_coord1.x = 0.;
			cap_coord1.y = 0.;
			cap_coord1.z = z[z_id+1];
			iesc = orc_segment(cap_coord0, cap_coord1, rad0, rad1, phot_coord0, phot_coord1, photon->exit_direction, &interact_coords, &surface_norm);
			z_id++;
		} while (iesc != 1 && z_id < nmax-1);
	} else {
next_hexagon:
		/* :1016-1064 step through the glass in steps of cap/10 until the hexagon cell changes */
		do {
			dist += cap[z_id]/10.;
			phot_coord0.x = photon->exit_coords.x + dist*photon->exit_direction.x;
			phot_coord0.y = photon->exit_coords.y + dist*photon->exit_direction.y;
			phot_coord0.z = photon->exit_coords.z + dist*photon->exit_direction.z;
			for (i = 0; i < nmax; i++) {
				if (z[i] <= phot_coord0.z)
					z_id = i;
			}
			current_polycap_ext = ((ext[z_id+1] - ext[z_id])/(z[z_id+1] - z[z_id])) * (phot_coord0.z - z[z_id]) + ext[z_id];
			rad0 = ((cap[z_id+1] - cap[z_id])/(z[z_id+1] - z[z_id])) * (phot_coord0.z - z[z_id]) + cap[z_id];
			zz = current_polycap_ext/(2.*ORC_COSPI_6*(n_shells+1));
			hex_index(phot_coord0.x, phot_coord0.y, zz, &q_new, &r_new);
			/* :1043-1063 stumbled into the capillary (q_i, r_i) it started next to */
			zz = current_polycap_ext/(2.*ORC_COSPI_6*(n_shells+1));
			cap_coord0.y = r_i * (3./2) * zz;
			cap_coord0.x = (2.* q_i+r_i) * ORC_COSPI_6 * zz;
			d_phot0 = sqrt((phot_coord0.x-cap_coord0.x)*(phot_coord0.x-cap_coord0.x)+(phot_coord0.y-cap_coord0.y)*(phot_coord0.y-cap_coord0.y));
			if (d_phot0 < rad0 && fabs(q_i) <= n_shells && fabs(r_i) <= n_shells && fabs(-1.*q_i-r_i) <= n_shells) {
				photon_coord_rel.x = phot_coord0.x - photon->exit_coords.x;
				photon_coord_rel.y = phot_coord0.y - photon->exit_coords.y;
				photon_coord_rel.z = phot_coord0.z - photon->exit_coords.z;
				*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
				if (*d_travel > 1.e-5) {
					*r_cntr = r_i;
					*q_cntr = q_i;
					return 1;
				} else {
					*d_travel = 0.;
				}
			}
		} while (q_new == q_i && r_new == r_i && phot_coord0.z <= z[nmax]);

		/* :1068-1100 left the hexagon stacking, or flew past the exit plane */
		if (fabs(q_new) > n_shells || fabs(r_new) > n_shells || fabs(-1.*q_new-r_new) > n_shells || phot_coord0.z > z[nmax]) {
			temp_phot.x = photon->exit_coords.x + photon->exit_direction.x * (z[nmax]-photon->exit_coords.z)/photon->exit_direction.z;
			temp_phot.y = photon->exit_coords.y + photon->exit_direction.y * (z[nmax]-photon->exit_coords.z)/photon->exit_direction.z;
			temp_phot.z = z[nmax];
			*r_cntr = r_new;
			*q_cntr = q_new;
			if (orc_within_pc_boundary(ext[nmax], temp_phot) == 0) {
				if (!orc_pc_intersect(optic, temp_phot, photon->exit_direction, &phot_inter)) {
					photon_coord_rel.x = phot_coord0.x - photon->exit_coords.x;
					photon_coord_rel.y = phot_coord0.y - photon->exit_coords.y;
					photon_coord_rel.z = phot_coord0.z - photon->exit_coords.z;
				} else {
					photon_coord_rel.x = phot_inter.x - photon->exit_coords.x;
					photon_coord_rel.y = phot_inter.y - photon->exit_coords.y;
					photon_coord_rel.z = phot_inter.z - photon->exit_coords.z;
				}
				*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
				return 3;
			} else {
				photon_coord_rel.x = phot_coord0.x - photon->exit_coords.x;
				photon_coord_rel.y = phot_coord0.y - photon->exit_coords.y;
				photon_coord_rel.z = phot_coord0.z - photon->exit_coords.z;
				*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
				return 2;
			}
		}

		/* :1105-1136 wall of capillary (q_new, r_new), segment by segment from z_id */
		iesc = 0;
		do {
			rad0 = cap[z_id];
			rad1 = cap[z_id+1];
			phot_coord0.x = photon->exit_coords.x + photon->exit_direction.x * (z[z_id]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord0.y = photon->exit_coords.y + photon->exit_direction.y * (z[z_id]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord0.z = z[z_id];
			phot_coord1.x = photon->exit_coords.x + photon->exit_direction.x * (z[z_id+1]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord1.y = photon->exit_coords.y + photon->exit_direction.y * (z[z_id+1]-photon->exit_coords.z)/photon->exit_direction.z;
			phot_coord1.z = z[z_id+1];

			zz = ext[z_id]/(2.*ORC_COSPI_6*(n_shells+1));
			cap_coord0.y = r_new * (3./2) * zz;
			cap_coord0.x = (2.* q_new+r_new) * ORC_COSPI_6 * zz;
			cap_coord0.z = z[z_id];
			zz = ext[z_id+1]/(2.*ORC_COSPI_6*(n_shells+1));
			cap_coord1.y = r_new * (3./2) * zz;
			cap_coord1.x = (2.* q_new+r_new) * ORC_COSPI_6 * zz;
			cap_coord1.z = z[z_id+1];
			iesc = orc_segment(cap_coord0, cap_coord1, rad0, rad1, phot_coord0, phot_coord1, photon->exit_direction, &interact_coords, &surface_norm);
			z_id++;
		} while (iesc != 1 && z_id < nmax-1);
		if (z_id >= nmax && iesc != 0) {
			/* :1129-1135 nothing in this capillary: go on to the next hexagon cell */
			q_i = q_new;
			r_i = r_new;
			z_id = nmax-1;
			goto next_hexagon;
		}
	}

	/* :1142-1190 */
	*r_cntr = r_new;
	*q_cntr = q_new;
	if (iesc != 1) {
		temp_phot.x = photon->exit_coords.x + photon->exit_direction.x * (z[nmax]-photon->exit_coords.z)/photon->exit_direction.z;
		temp_phot.y = photon->exit_coords.y + photon->exit_direction.y * (z[nmax]-photon->exit_coords.z)/photon->exit_direction.z;
		temp_phot.z = z[nmax];
		if (orc_within_pc_boundary(ext[nmax], temp_phot) == 0) {
			if (!orc_pc_intersect(optic, temp_phot, photon->exit_direction, &phot_inter)) {
				photon_coord_rel.x = temp_phot.x - photon->exit_coords.x;
				photon_coord_rel.y = temp_phot.y - photon->exit_coords.y;
				photon_coord_rel.z = temp_phot.z - photon->exit_coords.z;
			} else {
				photon_coord_rel.x = phot_inter.x - photon->exit_coords.x;
				photon_coord_rel.y = phot_inter.y - photon->exit_coords.y;
				photon_coord_rel.z = phot_inter.z - photon->exit_coords.z;
			}
			*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
			return 3;
		} else {
			photon_coord_rel.x = temp_phot.x - photon->exit_coords.x;
			photon_coord_rel.y = temp_phot.y - photon->exit_coords.y;
			photon_coord_rel.z = temp_phot.z - photon->exit_coords.z;
			*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
			return 2;
		}
	} else {
		photon_coord_rel.x = interact_coords.x - photon->exit_coords.x;
		photon_coord_rel.y = interact_coords.y - photon->exit_coords.y;
		photon_coord_rel.z = interact_coords.z - photon->exit_coords.z;
		*d_travel = sqrt(orc_scalar(photon_coord_rel, photon_coord_rel));
		if (z_id >= nmax)
			return 2;
		else
			return 1;
	}
}

/* ------------------------------------------------------------------ src/polycap-capil.c:565-891, leak_calc=true
 * 1 reflected, 0 absorbed, -1 error, -2 error inside the photon traced through a neighbouring capillary. */
int orc_reflect_leak(const orc_optic *optic, orc_photon *photon, orc_vec3 surface_norm)
{
	const double *z = optic->z, *ext = optic->ext;
	const int nmax = optic->nmax;
	size_t ie;
	int i, iesc, wall_trace, iesc_temp = 0;
	double cons1, r_rough, rtot, alfa;
	double *w_leak;
	int r_cntr, q_cntr;
	double zz, d_travel;
	int leak_flag = 0, weight_flag = 0;
	orc_vec3 leak_coords;
	double n_shells;
	int z_id = 0;
	double current_polycap_ext;
	orc_vec3 electric_vector = {0., 0., 0.};

	/* :596-602 */
	orc_norm(&surface_norm);
	orc_norm(&photon->exit_direction);
	alfa = orc_scalar(photon->exit_direction, surface_norm);
	if (alfa < 0.) return -1;
	w_leak = malloc(sizeof(double)*photon->n_energies);

	/* :612-619 */
	wall_trace = orc_trace_wall(optic, photon, &d_travel, &r_cntr, &q_cntr);
	if (wall_trace <= 0) {
		free(w_leak);
		return -1;
	}

	/* :625-645 */
	for (ie = 0; ie < photon->n_energies; ie++) {
		cons1 = (1.01358e0*photon->energies[ie])*alfa*optic->sig_rough;
		r_rough = exp(-1.*cons1*cons1);
		rtot = orc_refl_polar(photon->energies[ie], optic->density, photon->scatf[ie], photon->amu[ie], surface_norm, photon, &electric_vector);
		if (rtot < 0. || rtot > 1.) {
			free(w_leak);
			return -1;
		}
		w_leak[ie] = (1.-rtot * r_rough) * photon->weight[ie] * exp(-1.*d_travel*photon->amu[ie]);
		if (w_leak[ie] >= 1.e-4) leak_flag = 1;
		photon->weight[ie] = photon->weight[ie] * rtot * r_rough;
		if (photon->weight[ie] >= 1.e-4) weight_flag = 1;
	}
	iesc = (weight_flag != 1) ? 0 : 1;
	photon->exit_electric_vector = electric_vector;
	n_shells = orc_n_shells(optic->n_cap);

	if (leak_flag == 1) {
		/* :660-665 */
		leak_coords.x = photon->exit_coords.x + (d_travel / sqrt(orc_scalar(photon->exit_direction, photon->exit_direction))) * photon->exit_direction.x;
		leak_coords.y = photon->exit_coords.y + (d_travel / sqrt(orc_scalar(photon->exit_direction, photon->exit_direction))) * photon->exit_direction.y;
		leak_coords.z = photon->exit_coords.z + (d_travel / sqrt(orc_scalar(photon->exit_direction, photon->exit_direction))) * photon->exit_direction.z;

		/* :668-685 */
		if (wall_trace == 1) {
			for (i = 0; i < nmax; i++) {
				if (z[i] <= leak_coords.z)
					z_id = i;
			}
			current_polycap_ext = ((ext[z_id+1] - ext[z_id])/(z[z_id+1] - z[z_id])) * (leak_coords.z - z[z_id]) + ext[z_id];
			if (n_shells == 0.) {
				if (sqrt(leak_coords.x*leak_coords.x + leak_coords.y*leak_coords.y) >= current_polycap_ext)
					wall_trace = 3;
			} else {
				if (orc_within_pc_boundary(current_polycap_ext, leak_coords) == 0)
					wall_trace = 3;
			}
		}
		/* :687-710 */
		if (wall_trace == 3)
			leak_append(&photon->extleak, &photon->n_extleak, leak_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, w_leak);
		if (wall_trace == 2)
			leak_append(&photon->intleak, &photon->n_intleak, leak_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, w_leak);
		/* :711-881 the leaked fraction goes on as a photon of its own in the neighbouring capillary */
		if (wall_trace == 1 && leak_coords.z < z[nmax]) {
			orc_photon pt;
			double *capx_temp, *capy_temp;
			int ix_val_temp = 0;
			int *ix_temp = &ix_val_temp;
			memset(&pt, 0, sizeof(pt));
			/* polycap_photon_new (src/polycap-photon.c:98-136) */
			pt.start_coords = pt.exit_coords = leak_coords;
			pt.start_direction = pt.exit_direction = photon->exit_direction;
			pt.start_electric_vector = pt.exit_electric_vector = photon->exit_electric_vector;
			pt.leak_calc = 1;
			pt.i_refl = photon->i_refl;
			pt.d_travel = photon->d_travel + d_travel;
			pt.n_energies = photon->n_energies;
			pt.energies = photon->energies;
			pt.amu = photon->amu;
			pt.scatf = photon->scatf;
			pt.weight = malloc(sizeof(double)*pt.n_energies);
			for (ie = 0; ie < photon->n_energies; ie++)
				pt.weight[ie] = w_leak[ie];
			capx_temp = malloc(sizeof(double)*(nmax+1));
			capy_temp = malloc(sizeof(double)*(nmax+1));
			/* :779-790 */
			for (i = 0; i <= nmax; i++) {
				if (z[i] <= pt.exit_coords.z) *ix_temp = i;
				if (n_shells == 0.) {
					capx_temp[i] = 0.;
					capy_temp[i] = 0.;
				} else {
					zz = ext[i]/(2.*ORC_COSPI_6*(n_shells+1));
					capy_temp[i] = (3./2) * r_cntr * zz;
					capx_temp[i] = (2.* q_cntr + r_cntr) * ORC_COSPI_6 * zz;
				}
			}
			/* :795-803 */
			for (i = *ix_temp; i <= nmax; i++) {
				iesc_temp = orc_trace(optic, ix_temp, &pt, capx_temp, capy_temp);
				if (iesc_temp != 1)
					break;
			}
			free(capx_temp);
			free(capy_temp);
			/* :810-816 */
			if (iesc_temp == -1 || iesc_temp == -3) {
				orc_photon_clear_leaks(&pt);
				free(pt.weight);
				free(w_leak);
				return -2;
			}
			/* :821-850 its leaks become this photon's leaks */
			for (int64_t k = 0; k < pt.n_extleak; k++)
				leak_append(&photon->extleak, &photon->n_extleak, pt.extleak[k].coords, pt.extleak[k].direction, pt.extleak[k].elecv, pt.extleak[k].n_refl, photon->n_energies, pt.extleak[k].weight);
			for (int64_t k = 0; k < pt.n_intleak; k++)
				leak_append(&photon->intleak, &photon->n_intleak, pt.intleak[k].coords, pt.intleak[k].direction, pt.intleak[k].elecv, pt.intleak[k].n_refl, photon->n_energies, pt.intleak[k].weight);
			/* :854-880 what is left of it at the end of the optic is one more leak event; coordinates extrapolated to the
			 * exit plane, direction / electric vector / reflection count taken from THIS photon (as the reference does) */
			if (iesc_temp == 1 || iesc_temp == -2) {
				leak_coords.x = pt.exit_coords.x + pt.exit_direction.x * ((z[nmax]-pt.exit_coords.z)/pt.exit_direction.z);
				leak_coords.y = pt.exit_coords.y + pt.exit_direction.y * ((z[nmax]-pt.exit_coords.z)/pt.exit_direction.z);
				leak_coords.z = pt.exit_coords.z + pt.exit_direction.z * ((z[nmax]-pt.exit_coords.z)/pt.exit_direction.z);
				iesc_temp = orc_within_pc_boundary(ext[nmax], leak_coords);
				if (iesc_temp == 0)
					leak_append(&photon->extleak, &photon->n_extleak, leak_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, pt.weight);
				else if (iesc_temp == 1)
					leak_append(&photon->intleak, &photon->n_intleak, leak_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, pt.weight);
			}
			orc_photon_clear_leaks(&pt);
			free(pt.weight);
		}
	}
	free(w_leak);
	return iesc;
}

/* ------------------------------------------------------------------ src/polycap-photon.c:645-907
 * The photon starts inside the glass (d_ph_capcen > current_cap_rad) and leak_calc is on.  Returns the launch
 * return code.  cap_x/cap_y are the launch's axis arrays (rewritten for the capillary the photon leaks into). */
int orc_launch_in_wall_leak(const orc_optic *optic, orc_photon *photon, double *cap_x, double *cap_y, int *ix)
{
	const double *z = optic->z, *ext = optic->ext;
	const int nmax = optic->nmax;
	const double n_shells = orc_n_shells(optic->n_cap);
	int i, iesc = 0, wall_trace, r_cntr, q_cntr;
	double d_travel, zz;
	size_t ie;

	if (photon->start_coords.z == 0) {
		/* :647-672: reflection off the entrance face, normal = optic axis; the return code of reflect is ignored */
		orc_vec3 central_axis = {0., 0., 1.};
		orc_reflect_leak(optic, photon, central_axis);
		return 2;
	}
	if (!(photon->start_coords.z > 0))
		return 2;
	/* :674-697 */
	wall_trace = orc_trace_wall(optic, photon, &d_travel, &r_cntr, &q_cntr);
	if (wall_trace <= 0)
		return -1;
	/* :698-706 */
	for (ie = 0; ie < photon->n_energies; ie++)
		photon->weight[ie] = photon->weight[ie] * exp(-1.*d_travel*photon->amu[ie]);
	{
		/* the three components are updated one after the other with the same factor (|dir| is 1 here) */
		double f = d_travel / sqrt(orc_scalar(photon->exit_direction, photon->exit_direction));
		photon->exit_coords.x = photon->exit_coords.x + f * photon->exit_direction.x;
		photon->exit_coords.y = photon->exit_coords.y + f * photon->exit_direction.y;
		photon->exit_coords.z = photon->exit_coords.z + f * photon->exit_direction.z;
	}
	if (wall_trace == 3)
		leak_append(&photon->extleak, &photon->n_extleak, photon->exit_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, photon->weight);
	if (wall_trace == 2)
		leak_append(&photon->intleak, &photon->n_intleak, photon->exit_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, photon->weight);
	if (wall_trace == 1) {
		/* :759-858 */
		photon->d_travel = photon->d_travel + d_travel;
		for (i = 0; i <= nmax; i++) {
			zz = ext[i]/(2.*ORC_COSPI_6*(n_shells+1));
			cap_y[i] = r_cntr * (3./2) * zz;
			cap_x[i] = (2.* q_cntr+r_cntr) * ORC_COSPI_6 * zz;
			if (z[i] <= photon->exit_coords.z) *ix = i;
		}
		for (i = 0; i <= nmax; i++) {
			iesc = orc_trace(optic, ix, photon, cap_x, cap_y);
			if (iesc != 1)
				break;
		}
		if (iesc == -1 || iesc == -3)
			return -1;
		if (iesc == 1 || iesc == -2) {
			photon->exit_coords.x = photon->exit_coords.x + photon->exit_direction.x * ((z[nmax]-photon->exit_coords.z)/photon->exit_direction.z);
			photon->exit_coords.y = photon->exit_coords.y + photon->exit_direction.y * ((z[nmax]-photon->exit_coords.z)/photon->exit_direction.z);
			photon->exit_coords.z = photon->exit_coords.z + photon->exit_direction.z * ((z[nmax]-photon->exit_coords.z)/photon->exit_direction.z);
			iesc = orc_within_pc_boundary(ext[nmax], photon->exit_coords);
			if (iesc == 0)
				leak_append(&photon->extleak, &photon->n_extleak, photon->exit_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, photon->weight);
			else if (iesc == 1)
				leak_append(&photon->intleak, &photon->n_intleak, photon->exit_coords, photon->exit_direction, photon->exit_electric_vector, photon->i_refl, photon->n_energies, photon->weight);
		}
	}
	/* :860-870: the photon itself counts as absorbed and is parked outside the exit window */
	for (ie = 0; ie < photon->n_energies; ie++)
		photon->weight[ie] = 0.;
	photon->exit_coords.x = ext[nmax]+1.;
	photon->exit_coords.y = ext[nmax]+1.;
	photon->exit_coords.z = z[nmax];
	photon->exit_direction = photon->start_direction;
	orc_norm(&photon->exit_direction);
	return 1;
}

/* ------------------------------------------------------------------ flat interfaces for the tests */

void orc_free(void *p)
{
	free(p);
}

static double *records_from(const orc_leak *list, int64_t n, size_t n_energies, const int64_t *slot, const uint32_t *attempt)
{
	const size_t head = (slot != NULL) ? 2 : 0, stride = head + 10 + n_energies;
	double *out = malloc(sizeof(double) * stride * (size_t)(n > 0 ? n : 1));
	for (int64_t k = 0; k < n; k++) {
		double *r = out + (size_t)k * stride;
		if (slot != NULL) {
			r[0] = (double)slot[k];
			r[1] = (double)attempt[k];
		}
		r += head;
		r[0] = list[k].coords.x; r[1] = list[k].coords.y; r[2] = list[k].coords.z;
		r[3] = list[k].direction.x; r[4] = list[k].direction.y; r[5] = list[k].direction.z;
		r[6] = list[k].elecv.x; r[7] = list[k].elecv.y; r[8] = list[k].elecv.z;
		r[9] = (double)list[k].n_refl;
		memcpy(r + 10, list[k].weight, sizeof(double) * n_energies);
	}
	return out;
}

int orc_launch_one_leak(const orc_optic *optic, size_t n_energies, const double *energies,
                        const double *amu, const double *scatf,
                        const double start_coords[3], const double start_dir[3], const double start_elecv[3],
                        double *weights, double exit_coords[3], double exit_dir[3], double exit_elecv[3],
                        int64_t *i_refl, double *d_travel,
                        double **ext_records, int64_t *n_ext, double **int_records, int64_t *n_int)
{
	orc_photon ph;
	int rc;
	memset(&ph, 0, sizeof(ph));
	ph.start_coords.x = start_coords[0]; ph.start_coords.y = start_coords[1]; ph.start_coords.z = start_coords[2];
	ph.start_direction.x = start_dir[0]; ph.start_direction.y = start_dir[1]; ph.start_direction.z = start_dir[2];
	ph.start_electric_vector.x = start_elecv[0]; ph.start_electric_vector.y = start_elecv[1]; ph.start_electric_vector.z = start_elecv[2];
	ph.exit_coords = ph.start_coords;
	ph.exit_direction = ph.start_direction;
	ph.exit_electric_vector = ph.start_electric_vector;
	ph.leak_calc = 1;
	rc = orc_launch(optic, &ph, n_energies, energies, amu, scatf, weights);
	if (exit_coords) { exit_coords[0] = ph.exit_coords.x; exit_coords[1] = ph.exit_coords.y; exit_coords[2] = ph.exit_coords.z; }
	if (exit_dir) { exit_dir[0] = ph.exit_direction.x; exit_dir[1] = ph.exit_direction.y; exit_dir[2] = ph.exit_direction.z; }
	if (exit_elecv) { exit_elecv[0] = ph.exit_electric_vector.x; exit_elecv[1] = ph.exit_electric_vector.y; exit_elecv[2] = ph.exit_electric_vector.z; }
	if (i_refl) *i_refl = ph.i_refl;
	if (d_travel) *d_travel = ph.d_travel;
	*ext_records = records_from(ph.extleak, ph.n_extleak, n_energies, NULL, NULL);
	*n_ext = ph.n_extleak;
	*int_records = records_from(ph.intleak, ph.n_intleak, n_energies, NULL, NULL);
	*n_int = ph.n_intleak;
	orc_photon_clear_leaks(&ph);
	return rc;
}

/* ------------------------------------------------------------------ driver bookkeeping, src/polycap-source.c:799-879 */

static void move_leaks(orc_leak **dst, uint32_t **dst_att, int64_t *n_dst, orc_leak *src, const uint32_t *src_att, uint32_t att, int64_t n_src)
{
	if (n_src <= 0)
		return;
	*dst = realloc(*dst, sizeof(orc_leak) * (size_t)(*n_dst + n_src));
	*dst_att = realloc(*dst_att, sizeof(uint32_t) * (size_t)(*n_dst + n_src));
	for (int64_t k = 0; k < n_src; k++) {
		(*dst)[*n_dst + k] = src[k];            /* ownership of weight[] moves with the struct */
		(*dst_att)[*n_dst + k] = src_att ? src_att[k] : att;
	}
	*n_dst += n_src;
}

/* iesc = launch return code after the exit-window check.  0 / 2: the photon's events wait for the next transmitted
 * photon of this slot (:801-839); 1: they are stored, followed by the waiting ones (:841-877); otherwise (-1, -2)
 * the photon is freed with its events. */
void orc_slot_leaks_collect(orc_slot_leaks *sl, orc_photon *photon, int iesc, uint32_t attempt)
{
	if (iesc == 0 || iesc == 2) {
		move_leaks(&sl->ext_temp, &sl->ext_temp_attempt, &sl->n_ext_temp, photon->extleak, NULL, attempt, photon->n_extleak);
		move_leaks(&sl->int_temp, &sl->int_temp_attempt, &sl->n_int_temp, photon->intleak, NULL, attempt, photon->n_intleak);
	} else if (iesc == 1) {
		move_leaks(&sl->ext, &sl->ext_attempt, &sl->n_ext, photon->extleak, NULL, attempt, photon->n_extleak);
		move_leaks(&sl->intl, &sl->int_attempt, &sl->n_int, photon->intleak, NULL, attempt, photon->n_intleak);
		move_leaks(&sl->ext, &sl->ext_attempt, &sl->n_ext, sl->ext_temp, sl->ext_temp_attempt, 0, sl->n_ext_temp);
		move_leaks(&sl->intl, &sl->int_attempt, &sl->n_int, sl->int_temp, sl->int_temp_attempt, 0, sl->n_int_temp);
		free(sl->ext_temp); free(sl->ext_temp_attempt); free(sl->int_temp); free(sl->int_temp_attempt);
		sl->ext_temp = sl->int_temp = NULL;
		sl->ext_temp_attempt = sl->int_temp_attempt = NULL;
		sl->n_ext_temp = sl->n_int_temp = 0;
	} else {
		orc_leaks_free(photon->extleak, photon->n_extleak);
		orc_leaks_free(photon->intleak, photon->n_intleak);
	}
	if (iesc == 0 || iesc == 2 || iesc == 1) {
		free(photon->extleak);
		free(photon->intleak);
	}
	photon->extleak = photon->intleak = NULL;
	photon->n_extleak = photon->n_intleak = 0;
}

int orc_transmission_leak(const orc_optic *optic, const orc_source *source,
                          size_t n_energies, const double *energies, const double *amu, const double *scatf,
                          uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                          double *sum_weights, int64_t counters[4], double *img, double *exit_weights,
                          double **ext_records, int64_t *n_ext, double **int_records, int64_t *n_int)
{
	int failed = 0;
	orc_slot_leaks *sl = calloc((size_t)(n_slots > 0 ? n_slots : 1), sizeof(orc_slot_leaks));
	double *slot_w = malloc(sizeof(double) * n_energies * (size_t)(n_slots > 0 ? n_slots : 1));
	int64_t *slot_c = calloc((size_t)(n_slots > 0 ? n_slots : 1) * 4, sizeof(int64_t));
	uint32_t *used = calloc((size_t)(n_slots > 0 ? n_slots : 1), sizeof(uint32_t));
	int64_t j;
#ifdef _OPENMP
	if (n_threads < 1) n_threads = omp_get_max_threads();
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
#endif
	for (j = 0; j < n_slots; j++)
		used[j] = orc_one_slot(optic, source, n_energies, energies, amu, scatf, seed, slot0 + j, max_attempts,
		                       slot_w + (size_t)j*n_energies, slot_c + 4*j, img ? img + 17*(size_t)j : NULL, &sl[j]);

	/* totals in slot order (deterministic for any thread count) */
	for (size_t e = 0; e < n_energies; e++) sum_weights[e] = 0.;
	for (int c = 0; c < 4; c++) counters[c] = 0;
	int64_t tot_ext = 0, tot_int = 0;
	for (j = 0; j < n_slots; j++) {
		for (int c = 0; c < 4; c++) counters[c] += slot_c[4*j + c];
		if (used[j] == 0) {
			failed = 1;
			if (exit_weights)
				for (size_t e = 0; e < n_energies; e++) exit_weights[(size_t)j*n_energies + e] = 0.;
		} else {
			for (size_t e = 0; e < n_energies; e++) {
				sum_weights[e] += slot_w[(size_t)j*n_energies + e];
				if (exit_weights) exit_weights[(size_t)j*n_energies + e] = slot_w[(size_t)j*n_energies + e];
			}
		}
		tot_ext += sl[j].n_ext;
		tot_int += sl[j].n_int;
	}
	{
		orc_leak *all_ext = malloc(sizeof(orc_leak) * (size_t)(tot_ext > 0 ? tot_ext : 1));
		orc_leak *all_int = malloc(sizeof(orc_leak) * (size_t)(tot_int > 0 ? tot_int : 1));
		int64_t *slot_ext = malloc(sizeof(int64_t) * (size_t)(tot_ext > 0 ? tot_ext : 1));
		int64_t *slot_int = malloc(sizeof(int64_t) * (size_t)(tot_int > 0 ? tot_int : 1));
		uint32_t *att_ext = malloc(sizeof(uint32_t) * (size_t)(tot_ext > 0 ? tot_ext : 1));
		uint32_t *att_int = malloc(sizeof(uint32_t) * (size_t)(tot_int > 0 ? tot_int : 1));
		int64_t ke = 0, ki = 0;
		for (j = 0; j < n_slots; j++) {
			for (int64_t k = 0; k < sl[j].n_ext; k++, ke++) { all_ext[ke] = sl[j].ext[k]; slot_ext[ke] = slot0 + j; att_ext[ke] = sl[j].ext_attempt[k]; }
			for (int64_t k = 0; k < sl[j].n_int; k++, ki++) { all_int[ki] = sl[j].intl[k]; slot_int[ki] = slot0 + j; att_int[ki] = sl[j].int_attempt[k]; }
		}
		*ext_records = records_from(all_ext, tot_ext, n_energies, slot_ext, att_ext);
		*int_records = records_from(all_int, tot_int, n_energies, slot_int, att_int);
		*n_ext = tot_ext;
		*n_int = tot_int;
		for (ke = 0; ke < tot_ext; ke++) free(all_ext[ke].weight);
		for (ki = 0; ki < tot_int; ki++) free(all_int[ki].weight);
		free(all_ext); free(all_int); free(slot_ext); free(slot_int); free(att_ext); free(att_int);
	}
	for (j = 0; j < n_slots; j++) {
		free(sl[j].ext); free(sl[j].intl); free(sl[j].ext_attempt); free(sl[j].int_attempt);
		orc_leaks_free(sl[j].ext_temp, sl[j].n_ext_temp);
		orc_leaks_free(sl[j].int_temp, sl[j].n_int_temp);
		free(sl[j].ext_temp_attempt); free(sl[j].int_temp_attempt);
	}
	free(sl); free(slot_w); free(slot_c); free(used);
	return failed ? -1 : 0;
}
