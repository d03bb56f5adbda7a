/*
 * pc_photon.c -- polycap_photon: one photon and its trace through the optic.
 *
 * API and error behaviour of the reference's src/polycap-photon.c: polycap_photon_new :98-136,
 * polycap_photon_launch :390-955, getters :958-1013, leak getters :1037-1121, free.
 * polycap_photon_launch does not trace on the CPU: it hands the photon to the HIP kernel through
 * pc_hip_launch_photons (a batch of one) and copies the final state back into the photon.
 * leak_calc = true runs the leak ("halo") kernel (pc_hip_launch_photons_leak); its events become photon->extleak / intleak.
 */
#include "pc_private.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

polycap_photon *polycap_photon_new(polycap_description *description, polycap_vector3 start_coords,
	polycap_vector3 start_direction, polycap_vector3 start_electric_vector, polycap_error **error)
{
	if (description == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_new: description cannot be NULL");
		return NULL;
	}
	if (start_coords.z < 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_new: start_coords.z must be greater than 0");
		return NULL;
	}
	if (start_direction.z < 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_new: start_direction.z must be greater than 0");
		return NULL;
	}
	polycap_photon *photon = calloc(1, sizeof(polycap_photon));
	if (photon == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_photon_new: could not allocate memory for photon -> %s", strerror(errno));
		return NULL;
	}
	photon->description = description;
	photon->start_coords = start_coords;
	photon->exit_coords = start_coords;
	photon->start_direction = start_direction;
	photon->exit_direction = start_direction;
	photon->start_electric_vector = start_electric_vector;
	photon->exit_electric_vector = start_electric_vector;
	photon->d_travel = 0;
	return photon;
}

int polycap_photon_launch(polycap_photon *photon, size_t n_energies, double *energies, double **weights, bool leak_calc, polycap_error **error)
{
	/* argument checks and messages of the reference, src/polycap-photon.c:410-431 */
	if (photon == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: photon cannot be NULL");
		return -1;
	}
	if (energies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: energies cannot be NULL");
		return -1;
	}
	if (n_energies < 1) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: n_energies must be greater than 0");
		return -1;
	}
	for (size_t i = 0; i < n_energies; i++) {
		if (energies[i] < 1. || energies[i] > 100.) {
			polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: energies[i] must be greater than 1 and less than 100");
			return -1;
		}
	}
	if (weights == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: weights cannot be NULL");
		return -1;
	}
	/* events of an earlier launch of the same photon are dropped (reference :434-450) */
	pc_leak_list_free(photon->extleak, photon->n_extleak);
	pc_leak_list_free(photon->intleak, photon->n_intleak);
	photon->extleak = photon->intleak = NULL;
	photon->n_extleak = photon->n_intleak = 0;
	polycap_description *description = photon->description;
	if (description == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_launch: description cannot be NULL");
		return -1;
	}

	*weights = malloc(sizeof(double)*n_energies);
	if (*weights == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_photon_launch: could not allocate memory for weights -> %s", strerror(errno));
		return -1;
	}
	for (size_t i = 0; i < n_energies; i++)
		(*weights)[i] = 1.;
	photon->n_energies = n_energies;
	photon->i_refl = 0;
	photon->n_extleak = 0;
	photon->n_intleak = 0;

	pc_hip_ctx *ctx = pc_ctx_for(&description->cache, description, n_energies, energies, NULL, "polycap_photon_launch", error);
	if (ctx == NULL)
		return -1;

	double start[3] = { photon->start_coords.x, photon->start_coords.y, photon->start_coords.z };
	double dir[3] = { photon->start_direction.x, photon->start_direction.y, photon->start_direction.z };
	double elecv[3] = { photon->start_electric_vector.x, photon->start_electric_vector.y, photon->start_electric_vector.z };
	double exit_coords[3], exit_dir[3], exit_elecv[3], d_travel = 0.;
	int32_t rc = -1;
	int64_t i_refl = 0;
	int status = leak_calc
		? pc_hip_launch_photons_leak(ctx, 1, start, dir, elecv, &rc, *weights, exit_coords, exit_dir, exit_elecv, &i_refl, &d_travel)
		: pc_hip_launch_photons(ctx, 1, start, dir, elecv, &rc, *weights, exit_coords, exit_dir, exit_elecv, &i_refl, &d_travel);
	if (status != PC_HIP_OK) {
		pc_set_hip_error(error, "polycap_photon_launch", status);
		return -1;
	}
	if (leak_calc) {
		if (pc_fetch_leaks(ctx, 0, n_energies, &photon->extleak, &photon->n_extleak, NULL, "polycap_photon_launch", error) != 0 ||
		    pc_fetch_leaks(ctx, 1, n_energies, &photon->intleak, &photon->n_intleak, NULL, "polycap_photon_launch", error) != 0)
			return -1;
	}

	/* the reference normalises start_direction in place (src/polycap-photon.c:493) */
	double norm = sqrt(dir[0]*dir[0] + dir[1]*dir[1] + dir[2]*dir[2]);
	photon->start_direction.x = dir[0]/norm;
	photon->start_direction.y = dir[1]/norm;
	photon->start_direction.z = dir[2]/norm;
	if (rc == -2) {
		/* src/polycap-photon.c:517-537, 553-573 */
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, pc_n_shells(description->n_cap) == 0.
			? "polycap_photon_launch: photon_pos_check: photon not within monocapillary boundaries"
			: "polycap_photon_launch: photon_pos_check: photon not within optic boundaries");
	}
	photon->exit_coords.x = exit_coords[0]; photon->exit_coords.y = exit_coords[1]; photon->exit_coords.z = exit_coords[2];
	photon->exit_direction.x = exit_dir[0]; photon->exit_direction.y = exit_dir[1]; photon->exit_direction.z = exit_dir[2];
	photon->exit_electric_vector.x = exit_elecv[0]; photon->exit_electric_vector.y = exit_elecv[1]; photon->exit_electric_vector.z = exit_elecv[2];
	photon->i_refl = i_refl;
	photon->d_travel += d_travel;   /* launch never resets d_travel in the reference */
	return rc;
}

double polycap_photon_get_dtravel(polycap_photon *photon) { return photon->d_travel; }
int64_t polycap_photon_get_irefl(polycap_photon *photon) { return photon->i_refl; }
polycap_vector3 polycap_photon_get_start_coords(polycap_photon *photon) { return photon->start_coords; }
polycap_vector3 polycap_photon_get_start_direction(polycap_photon *photon) { return photon->start_direction; }
polycap_vector3 polycap_photon_get_start_electric_vector(polycap_photon *photon) { return photon->start_electric_vector; }
polycap_vector3 polycap_photon_get_exit_coords(polycap_photon *photon) { return photon->exit_coords; }
polycap_vector3 polycap_photon_get_exit_direction(polycap_photon *photon) { return photon->exit_direction; }
polycap_vector3 polycap_photon_get_exit_electric_vector(polycap_photon *photon) { return photon->exit_electric_vector; }

/* One polycap_leak per event record of include/polycap-hip.h (slot, attempt, coords, direction, elecv, n_refl, weights).
 * slots (optional) receives the slot column.  Returns 0, or -1 with *error set. */
int pc_fetch_leaks(pc_hip_ctx *ctx, int kind, size_t n_energies, polycap_leak ***list, int64_t *n, int64_t **slots,
	const char *caller, polycap_error **error)
{
	int64_t n_ext = 0, n_int = 0;
	*list = NULL;
	*n = 0;
	if (slots) *slots = NULL;
	int status = pc_hip_leak_counts(ctx, &n_ext, &n_int);
	if (status != PC_HIP_OK) {
		pc_set_hip_error(error, caller, status);
		return -1;
	}
	const int64_t count = (kind == 0) ? n_ext : n_int;
	if (count == 0)
		return 0;
	const size_t stride = PC_HIP_LEAK_HDR + n_energies;
	double *rec = malloc(sizeof(double) * stride * (size_t)count);
	polycap_leak **out = calloc((size_t)count, sizeof(polycap_leak *));
	int64_t *sl = slots ? malloc(sizeof(int64_t) * (size_t)count) : NULL;
	if (rec == NULL || out == NULL || (slots && sl == NULL)) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for leak events -> %s", caller, strerror(errno));
		free(rec); free(out); free(sl);
		return -1;
	}
	status = pc_hip_leak_events(ctx, kind, 0, count, rec);
	if (status != PC_HIP_OK) {
		pc_set_hip_error(error, caller, status);
		free(rec); free(out); free(sl);
		return -1;
	}
	for (int64_t k = 0; k < count; k++) {
		const double *r = rec + (size_t)k * stride;
		polycap_leak *l = malloc(sizeof(polycap_leak));
		double *w = malloc(sizeof(double) * n_energies);
		if (l == NULL || w == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for a leak event -> %s", caller, strerror(errno));
			free(l); free(w);
			pc_leak_list_free(out, k);
			free(rec); free(sl);
			return -1;
		}
		l->coords.x = r[2]; l->coords.y = r[3]; l->coords.z = r[4];
		l->direction.x = r[5]; l->direction.y = r[6]; l->direction.z = r[7];
		l->elecv.x = r[8]; l->elecv.y = r[9]; l->elecv.z = r[10];
		l->n_refl = (int64_t)r[11];
		l->n_energies = n_energies;
		memcpy(w, r + PC_HIP_LEAK_HDR, sizeof(double) * n_energies);
		l->weight = w;
		out[k] = l;
		if (sl) sl[k] = (int64_t)r[0];
	}
	free(rec);
	*list = out;
	*n = count;
	if (slots) *slots = sl;
	return 0;
}

void pc_leak_list_free(polycap_leak **list, int64_t n)
{
	if (list == NULL)
		return;
	for (int64_t k = 0; k < n; k++)
		polycap_leak_free(list[k]);
	free(list);
}

/* deep copy for the getters (reference :1037-1121): caller owns the result */
bool pc_leak_list_copy(polycap_leak **src, int64_t n_src, polycap_leak ***leaks, int64_t *n_leaks, const char *caller, const char *what,
	polycap_error **error)
{
	*n_leaks = n_src;
	if (n_src == 0) {
		*leaks = NULL;
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "%s: no %s events in photon", caller, what);
		return false;
	}
	*leaks = malloc(sizeof(polycap_leak *) * (size_t)n_src);
	if (*leaks == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for leaks -> %s", caller, strerror(errno));
		return false;
	}
	for (int64_t i = 0; i < n_src; i++) {
		(*leaks)[i] = malloc(sizeof(polycap_leak));
		if ((*leaks)[i] == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for (*leaks)[i] -> %s", caller, strerror(errno));
			return false;
		}
		memcpy((*leaks)[i], src[i], sizeof(polycap_leak));
		(*leaks)[i]->weight = malloc(sizeof(double) * src[i]->n_energies);
		if ((*leaks)[i]->weight == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for (*leaks[i])->weight -> %s", caller, strerror(errno));
			return false;
		}
		memcpy((*leaks)[i]->weight, src[i]->weight, sizeof(double) * src[i]->n_energies);
	}
	return true;
}

bool polycap_photon_get_extleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
{
	if (photon == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_get_extleak_data: photon cannot be NULL");
		return false;
	}
	return pc_leak_list_copy(photon->extleak, photon->n_extleak, leaks, n_leaks, "polycap_photon_get_extleak_data", "extleak", error);
}

bool polycap_photon_get_intleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
{
	if (photon == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_photon_get_intleak_data: photon cannot be NULL");
		return false;
	}
	return pc_leak_list_copy(photon->intleak, photon->n_intleak, leaks, n_leaks, "polycap_photon_get_intleak_data", "intleak", error);
}

void polycap_leak_free(polycap_leak *leak)
{
	if (leak == NULL)
		return;
	free(leak->weight);
	free(leak);
}

void polycap_photon_free(polycap_photon *photon)
{
	if (photon == NULL)
		return;
	free(photon->energies);
	free(photon->weight);
	free(photon->amu);
	free(photon->scatf);
	pc_leak_list_free(photon->extleak, photon->n_extleak);
	pc_leak_list_free(photon->intleak, photon->n_intleak);
	free(photon);
}
