/*
 * pc_source.c -- polycap_source: X-ray source + optic + energy grid, and the photon-loop driver.
 *
 * API and error behaviour of the reference's src/polycap-source.c:
 *   polycap_source_new :147-225, polycap_source_new_from_file :228-445 (legacy positional .inp deck),
 *   polycap_source_get_photon :23-144, polycap_source_get_transmission_efficiencies :448-1087,
 *   polycap_source_free / polycap_source_get_description :1090-1109.
 * The photon loop (the reference's OpenMP region :697-1049) runs on the GPU: this file only validates,
 * uploads the problem once, enqueues pc_hip_transmission_run for the n_photons exit-photon slots and copies
 * the image planes back.  Photon streams are Philox(seed, slot, attempt): the reference seeds one mt19937 per
 * OpenMP thread from /dev/urandom, so its runs are not reproducible; here POLYCAP_SEED fixes the key.
 */
#define _GNU_SOURCE
#include "pc_private.h"

#include <errno.h>
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

polycap_source *polycap_source_new(polycap_description *description, double d_source, double src_x, double src_y,
	double src_sigx, double src_sigy, double src_shiftx, double src_shifty, double hor_pol,
	size_t n_energies, double *energies, polycap_error **error)
{
	if (description == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: description cannot be NULL");
		return NULL;
	}
	if (d_source <= 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: d_source must be greater than 0");
		return NULL;
	}
	if (src_x <= 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: src_x must be greater than 0");
		return NULL;
	}
	if (src_y <= 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: src_y must be greater than 0");
		return NULL;
	}
	if (fabs(hor_pol) > 1.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: hor_pol must be greater than or equal to -1 and smaller than or equal to 1");
		return NULL;
	}
	if (n_energies <= 0.) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: n_energies must be greater than 0");
		return NULL;
	}
	if (energies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: energies cannot be NULL");
		return NULL;
	}
	for (size_t i = 0; i < n_energies; i++) {
		if (energies[i] < 1. || energies[i] > 100.) {
			polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: energies must be greater than 1 and smaller than 100");
			return NULL;
		}
	}
	if (polycap_profile_validate(description->profile, description->n_cap, error) != 1) {
		polycap_clear_error(error);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: description->profile is faulty. Some capillary coordinates are outside of the external radius.");
		return NULL;
	}

	polycap_source *source = calloc(1, sizeof(polycap_source));
	if (source != NULL)
		source->energies = malloc(sizeof(double)*n_energies);
	if (source == NULL || source->energies == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_new: could not allocate memory for source -> %s", strerror(errno));
		polycap_source_free(source);
		return NULL;
	}
	source->d_source = d_source;
	source->src_x = src_x;
	source->src_y = src_y;
	source->src_sigx = src_sigx;
	source->src_sigy = src_sigy;
	source->src_shiftx = src_shiftx;
	source->src_shifty = src_shifty;
	source->hor_pol = hor_pol;
	source->n_energies = n_energies;
	memcpy(source->energies, energies, sizeof(double)*n_energies);
	source->rng = polycap_rng_new();
	/* the source keeps its own deep copy of the description (tests free the original right away) */
	source->description = polycap_description_new(description->profile, description->sig_rough, description->n_cap,
		description->nelem, description->iz, description->wi, description->density, NULL);
	if (source->description == NULL) {
		polycap_clear_error(error);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new: description->profile is faulty. Some capillary coordinates are outside of the external radius.");
		polycap_source_free(source);
		return NULL;
	}
	return source;
}

/* opens `name`; a relative name that does not exist in the working directory is retried next to the .inp deck */
static char *pc_resolve_near(const char *deck, const char *name)
{
	FILE *probe = fopen(name, "r");
	if (probe != NULL) {
		fclose(probe);
		return strdup(name);
	}
	const char *slash = strrchr(deck, '/');
	if (name[0] == '/' || slash == NULL)
		return strdup(name);
	size_t dirlen = (size_t)(slash - deck) + 1;
	char *joined = malloc(dirlen + strlen(name) + 1);
	if (joined == NULL)
		return NULL;
	memcpy(joined, deck, dirlen);
	strcpy(joined + dirlen, name);
	return joined;
}

polycap_source *polycap_source_new_from_file(const char *filename, polycap_error **error)
{
	if (filename == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new_from_file: filename cannot be NULL");
		return NULL;
	}
	polycap_description *description = calloc(1, sizeof(polycap_description));
	polycap_source *source = calloc(1, sizeof(polycap_source));
	if (description == NULL || source == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_new_from_file: could not allocate memory for source -> %s", strerror(errno));
		free(description);
		free(source);
		return NULL;
	}
	source->description = description;
	source->rng = polycap_rng_new();

	FILE *fptr = fopen(filename, "r");
	if (fptr == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_source_new_from_file: could not open %s -> %s", filename, strerror(errno));
		polycap_source_free(source);
		return NULL;
	}

	/* positional deck, reference src/polycap-source.c:273-369 */
	double e_start = 0., e_final = 0., delta_e = 1.;
	int nphotons = 0, type = 0;
	int ok = 1;
	ok &= fscanf(fptr, "%lf", &description->sig_rough) == 1;
	ok &= fscanf(fptr, "%lf", &source->d_source) == 1;
	ok &= fscanf(fptr, "%lf %lf", &source->src_x, &source->src_y) == 2;
	ok &= fscanf(fptr, "%lf %lf", &source->src_sigx, &source->src_sigy) == 2;
	ok &= fscanf(fptr, "%lf %lf", &source->src_shiftx, &source->src_shifty) == 2;
	ok &= fscanf(fptr, "%lf", &source->hor_pol) == 1;
	ok &= fscanf(fptr, "%u", &description->nelem) == 1;
	if (!ok || description->nelem < 1 || description->nelem > 111) {
		fclose(fptr);
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_source_new_from_file: could not read the source/composition header of %s", filename);
		polycap_source_free(source);
		return NULL;
	}
	description->iz = malloc(sizeof(int)*description->nelem);
	description->wi = malloc(sizeof(double)*description->nelem);
	if (description->iz == NULL || description->wi == NULL) {
		fclose(fptr);
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_new_from_file: could not allocate memory for description->iz -> %s", strerror(errno));
		polycap_source_free(source);
		return NULL;
	}
	for (unsigned int i = 0; i < description->nelem; i++) {
		ok &= fscanf(fptr, "%d %lf", &description->iz[i], &description->wi[i]) == 2;
		description->wi[i] /= 100.0;
	}
	ok &= fscanf(fptr, "%lf", &description->density) == 1;
	ok &= fscanf(fptr, "%lf %lf %lf", &e_start, &e_final, &delta_e) == 3;
	if (!ok) {
		fclose(fptr);
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_source_new_from_file: could not read composition/energies from %s", filename);
		polycap_source_free(source);
		return NULL;
	}
	source->n_energies = (size_t)((e_final-e_start)/delta_e + 1);
	if (source->n_energies <= 0. || source->n_energies > 100000000u) {
		fclose(fptr);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new_from_file: source->n_energies must be greater than 0");
		polycap_source_free(source);
		return NULL;
	}
	source->energies = malloc(sizeof(double)*source->n_energies);
	if (source->energies == NULL) {
		fclose(fptr);
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_new_from_file: could not allocate memory for source->energies -> %s", strerror(errno));
		polycap_source_free(source);
		return NULL;
	}
	for (size_t i = 0; i < source->n_energies; i++)
		source->energies[i] = e_start + i*delta_e;
	ok &= fscanf(fptr, "%d", &nphotons) == 1;   /* read and ignored, as in the reference (:343) */
	ok &= fscanf(fptr, "%d", &type) == 1;
	if (ok && (type == 0 || type == 1 || type == 2)) {
		double length, rad_ext_upstream, rad_ext_downstream, rad_int_upstream, rad_int_downstream, focal_dist_upstream, focal_dist_downstream;
		if (fscanf(fptr, "%lf %lf %lf %lf %lf %lf %lf", &length, &rad_ext_upstream, &rad_ext_downstream, &rad_int_upstream,
		           &rad_int_downstream, &focal_dist_upstream, &focal_dist_downstream) != 7) {
			fclose(fptr);
			polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_source_new_from_file: could not read the profile shape from %s", filename);
			polycap_source_free(source);
			return NULL;
		}
		description->profile = polycap_profile_new((polycap_profile_type)type, length, rad_ext_upstream, rad_ext_downstream,
			rad_int_upstream, rad_int_downstream, focal_dist_upstream, focal_dist_downstream, error);
	} else if (ok) {
		fgetc(fptr); /* rest of the "type" line */
		char *names[3] = { NULL, NULL, NULL };
		for (int k = 0; k < 3; k++) {
			char *raw = polycap_read_input_line(fptr, NULL);
			names[k] = raw ? pc_resolve_near(filename, raw) : NULL;
			free(raw);
		}
		if (names[0] && names[1] && names[2])
			description->profile = polycap_profile_new_from_file(names[0], names[1], names[2], error);
		for (int k = 0; k < 3; k++)
			free(names[k]);
	}
	if (!ok || description->profile == NULL) {
		fclose(fptr);
		if (error != NULL && *error == NULL)
			polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_source_new_from_file: could not read the profile section of %s", filename);
		polycap_source_free(source);
		return NULL;
	}
	if (fscanf(fptr, "%" SCNd64, &description->n_cap) != 1)
		description->n_cap = 0;
	fclose(fptr);

	polycap_description_check_weight(description->nelem, description->wi, error);

	double n_cap = (pc_n_shells(description->n_cap)+0.5)*6.;
	n_cap = (n_cap*n_cap+3)/12;
	description->open_area = (description->profile->cap[0]*description->profile->cap[0]*M_PI)*n_cap/(3.*sin(M_PI/3)*description->profile->ext[0]*description->profile->ext[0]);

	/* sanity checks of the reference, :380-437 */
	const char *problem = NULL;
	if (source->d_source < 0.0) problem = "polycap_source_new_from_file: source_temp->d_source must be greater than 0.0";
	else if (source->src_x < 0.0) problem = "polycap_source_new_from_file: source_temp->src_x must be greater than 0.0";
	else if (source->src_y < 0.0) problem = "polycap_source_new_from_file: source_temp->src_y must be greater than 0.0";
	else if (description->n_cap < 1) problem = "polycap_source_new_from_file: description->n_cap must be greater than 1";
	else if (description->open_area < 0 || description->open_area > 1) problem = "polycap_source_new_from_file: description->open_area must be greater than 0 and less than 1";
	else if (description->density < 0.0) problem = "polycap_source_new_from_file: description->density must be greater than 0.0";
	for (size_t i = 0; problem == NULL && i < source->n_energies; i++)
		if (source->energies[i] < 1. || source->energies[i] > 100.)
			problem = "polycap_source_new_from_file: source->energies must be greater than 1 and smaller than 100";
	for (unsigned int i = 0; problem == NULL && i < description->nelem; i++)
		if (description->iz[i] < 1 || description->iz[i] > 94)
			problem = "polycap_source_new_from_file: description->iz[i] must be greater than 0 and less than 94";
	if (problem != NULL) {
		polycap_clear_error(error);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, problem);
		polycap_source_free(source);
		return NULL;
	}
	if (polycap_profile_validate(description->profile, description->n_cap, error) != 1) {
		polycap_clear_error(error);
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_new_from_file: description->profile is faulty. Some capillary coordinates are outside of the external radius.");
		polycap_source_free(source);
		return NULL;
	}
	return source;
}

/* One photon of the stream (rng->seed, photon index rng->counter++), sampled by the device sampler. */
polycap_photon *polycap_source_get_photon(polycap_source *source, polycap_rng *rng, polycap_error **error)
{
	if (source == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_photon: source cannot be NULL");
		return NULL;
	}
	polycap_description *description = source->description;
	if (description == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_photon: description cannot be NULL");
		return NULL;
	}
	if (rng == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_photon: rng cannot be NULL");
		return NULL;
	}
	pc_hip_ctx *ctx = pc_ctx_for(&source->cache, description, source->n_energies, source->energies, source, "polycap_source_get_photon", error);
	if (ctx == NULL)
		return NULL;
	int64_t slot = (int64_t)(rng->counter++ & 0x7fffffffffffffffull);
	uint32_t attempt = 0;
	double out[12];
	int status = pc_hip_sample_photons(ctx, rng->seed, 1, &slot, &attempt, out);
	if (status != PC_HIP_OK) {
		pc_set_hip_error(error, "polycap_source_get_photon", status);
		return NULL;
	}
	polycap_vector3 start_coords = { out[0], out[1], out[2] };
	polycap_vector3 start_direction = { out[3], out[4], out[5] };
	polycap_vector3 start_electric_vector = { out[6], out[7], out[8] };
	polycap_photon *photon = polycap_photon_new(description, start_coords, start_direction, start_electric_vector, error);
	if (photon == NULL)
		return NULL;
	photon->src_start_coords.x = out[9];
	photon->src_start_coords.y = out[10];
	photon->src_start_coords.z = 0.;
	return photon;
}

static uint64_t pc_env_u64(const char *name, uint64_t fallback, int *present)
{
	const char *env = getenv(name);
	if (present) *present = 0;
	if (env == NULL || *env == '\0')
		return fallback;
	char *end = NULL;
	unsigned long long v = strtoull(env, &end, 0);
	if (end == env)
		return fallback;
	if (present) *present = 1;
	return (uint64_t)v;
}

/* POLYCAP_HIP_DEVICES = "all" or a comma-separated list of device indices (an index may repeat).  *n = 0 when unset. */
static int pc_env_devices(int devices[64], int *n, polycap_error **error)
{
	*n = 0;
	const char *env = getenv("POLYCAP_HIP_DEVICES");
	if (env == NULL || *env == '\0')
		return 0;
	if (strcmp(env, "all") == 0) {
		int count = pc_hip_device_count();
		if (count < 1) {
			polycap_set_error_literal(error, POLYCAP_ERROR_RUNTIME, "polycap_source_get_transmission_efficiencies: POLYCAP_HIP_DEVICES=all but no HIP device is visible (the trace path has no CPU fallback)");
			return -1;
		}
		if (count > 64) count = 64;
		for (int k = 0; k < count; k++) devices[k] = k;
		*n = count;
		return 0;
	}
	const char *p = env;
	while (*p != '\0') {
		char *end = NULL;
		long v = strtol(p, &end, 10);
		if (end == p || v < 0 || *n >= 64 || (*end != ',' && *end != '\0')) {
			polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: cannot parse POLYCAP_HIP_DEVICES=%s (all, or up to 64 comma-separated device indices)", env);
			return -1;
		}
		devices[(*n)++] = (int)v;
		p = (*end == ',') ? end + 1 : end;
	}
	return 0;
}

/* POLYCAP_TIMING=1: stage times of polycap_source_get_transmission_efficiencies on stderr */
static double pc_now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec*1e3 + ts.tv_nsec*1e-6;
}

polycap_transmission_efficiencies *polycap_source_get_transmission_efficiencies(polycap_source *source, int max_threads, int n_photons,
	bool leak_calc, polycap_progress_monitor *progress_monitor, polycap_error **error)
{
	(void)max_threads; /* host-thread cap in the reference (:492-493); the photon loop runs on the GPU here */
	if (source == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: source cannot be NULL");
		return NULL;
	}
	if (progress_monitor != NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: progress_monitor must be NULL as polycap_progress_monitor currently has no implementation");
		return NULL;
	}
	polycap_description *description = source->description;
	if (description == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: description cannot be NULL");
		return NULL;
	}
	if (source->n_energies < 1) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: source->n_energies must be greater than or equal to 1");
		return NULL;
	}
	if (source->energies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: source->energies cannot be NULL");
		return NULL;
	}
	for (size_t i = 0; i < source->n_energies; i++) {
		if (source->energies[i] < 1. || source->energies[i] > 100.) {
			polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: source->energies[i] must be greater than 1 and less than 100");
			return NULL;
		}
	}
	if (n_photons < 1) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: n_photons must be greater than 1");
		return NULL;
	}

	const size_t ne = source->n_energies;
	const int timing = getenv("POLYCAP_TIMING") != NULL;
	/* Extensions of the reference call, all through the environment so that the signature stays the reference's:
	 *   POLYCAP_HIP_DEVICES=all | i,j,...  the photon loop is sharded over these devices from this one process (the
	 *       reference's OpenMP team, :697-745, becomes a team of GPUs); totals are summed by one RCCL all-reduce (:973-980)
	 *   POLYCAP_IMAGES=0                   histogram-only result: efficiencies and counts, no per-photon planes (at 1e8
	 *       photons x 291 energies the weight plane alone is 233 GB); the start/exit getters then report no events */
	int devices[64], n_devices = 0;
	if (pc_env_devices(devices, &n_devices, error) != 0)
		return NULL;
	const char *img_env = getenv("POLYCAP_IMAGES");
	const int keep_images = !(img_env != NULL && strcmp(img_env, "0") == 0);
	if (leak_calc && !keep_images) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_transmission_efficiencies: POLYCAP_IMAGES=0 cannot be combined with leak_calc (leak events are per-photon data)");
		return NULL;
	}
	double t_stage[8];
	t_stage[0] = pc_now_ms();
	polycap_transmission_efficiencies *eff = pc_transeff_alloc(source, keep_images ? (size_t)n_photons : 0, 0, "polycap_source_get_transmission_efficiencies", error);
	double *sum_weights = malloc(sizeof(double)*ne);
	if (eff == NULL || sum_weights == NULL) {
		if (eff != NULL)
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_get_transmission_efficiencies: could not allocate memory for efficiencies -> %s", strerror(errno));
		free(sum_weights);
		polycap_transmission_efficiencies_free(eff);
		return NULL;
	}

	pc_hip_ctx *ctx = NULL;
	pc_hip_group *group = NULL;
	if (n_devices > 1 || (n_devices > 0 && !leak_calc))      /* leak runs are sharded like plain runs (reference :744-884, 925-1032) */
		group = pc_group_for(&source->cache, description, ne, source->energies, source, n_devices, devices, "polycap_source_get_transmission_efficiencies", error);
	else      /* a one-entry list selects the device of a leak run; none: POLYCAP_HIP_DEVICE, default 0 */
		ctx = pc_ctx_for_device(&source->cache, description, ne, source->energies, source, n_devices >= 1 ? devices[0] : -1,
		                        "polycap_source_get_transmission_efficiencies", error);
	if (ctx == NULL && group == NULL) {
		free(sum_weights);
		polycap_transmission_efficiencies_free(eff);
		return NULL;
	}
	t_stage[1] = pc_now_ms();
	int have_seed = 0;
	uint64_t seed = pc_env_u64("POLYCAP_SEED", 0, &have_seed);
	if (!have_seed)
		seed = source->rng->seed + 0x9E3779B97F4A7C15ull * source->run_index;
	source->run_index++;
	uint32_t max_attempts = (uint32_t)pc_env_u64("POLYCAP_MAX_ATTEMPTS", 1u << 20, NULL);

	int64_t counters[6] = {0, 0, 0, 0, 0, 0};
	int status;
	/* big plain runs are traced in four parts so that the images of a finished part cross PCIe while the next part runs */
	const int parts = (!leak_calc && keep_images && n_photons >= 2000000) ? (int)pc_env_u64("POLYCAP_RUN_PARTS", 4, NULL) : 1;
	/* Plain runs store their exit photons in the order of completion (option "compact_images": coalesced plane stores, blocks
	 * copied to the host while the kernel runs); the reference's own order is the order in which randomly seeded threads fill
	 * the arrays.  POLYCAP_COMPACT=0 keeps every photon at the position of its slot (reproducible order for a given POLYCAP_SEED). */
	const char *compact_env = getenv("POLYCAP_COMPACT");
	const int compact = !(compact_env != NULL && strcmp(compact_env, "0") == 0) && !leak_calc;
	if (group != NULL) {
		status = pc_hip_group_set_option(group, "run_parts", parts);
		if (status == PC_HIP_OK)
			status = pc_hip_group_set_option(group, "compact_images", compact);
		if (status == PC_HIP_OK)
			status = pc_hip_group_set_option(group, "plane_images", leak_calc ? 0 : 1);
		if (status == PC_HIP_OK)
			status = leak_calc ? pc_hip_group_run_leak(group, seed, n_photons, max_attempts, 1)
			                   : pc_hip_group_run(group, seed, n_photons, max_attempts, keep_images);
	} else {
		status = pc_hip_set_option(ctx, "run_parts", parts);
		if (status == PC_HIP_OK)
			status = pc_hip_set_option(ctx, "compact_images", compact);
		if (status == PC_HIP_OK && getenv("POLYCAP_BLOCK_SHIFT") != NULL)
			status = pc_hip_set_option(ctx, "block_shift", (int64_t)pc_env_u64("POLYCAP_BLOCK_SHIFT", 18, NULL));
		if (status == PC_HIP_OK)
			status = pc_hip_set_option(ctx, "plane_images", leak_calc ? 0 : 1);   /* the result object wants planes: let the kernel write them */
		if (status == PC_HIP_OK)
			status = leak_calc ? pc_hip_transmission_run_leak(ctx, seed, 0, n_photons, max_attempts, 1)
			                   : pc_hip_transmission_run(ctx, seed, 0, n_photons, max_attempts, keep_images);
	}
	t_stage[2] = pc_now_ms();
	t_stage[3] = t_stage[2];
	if (status == PC_HIP_OK && keep_images) {
		pc_transeff_prefault(eff, (size_t)n_photons);    /* the kernel is running: fault the result pages in meanwhile */
		t_stage[3] = pc_now_ms();
		pc_hip_images dst;
		pc_transeff_plane_pointers(eff, &dst);
		/* a result in one slab stays pinned while it lives and in the pool after it (pc_transeff.c); planes of their own are
		 * pinned for the call only */
		const int keep_pinned = (group == NULL && eff->images->slab != NULL);
		if (keep_pinned)
			(void)pc_hip_set_option(ctx, "keep_pinned", 1);
		status = (group != NULL) ? pc_hip_group_images(group, &dst)
		                         : pc_hip_transmission_images(ctx, 0, n_photons, &dst);    /* block by block behind the kernel (a leak run: after it) */
		if (keep_pinned) {
			(void)pc_hip_set_option(ctx, "keep_pinned", 0);
			pc_transeff_planes_pinned(eff);      /* also after a failure: what the fetch pinned stays pinned until the slab is freed */
		}
	}
	t_stage[4] = pc_now_ms();
	int reduced_by = 0;
	if (status == PC_HIP_OK) {
		if (group != NULL) {
			const char *r = getenv("POLYCAP_RCCL");      /* 0: host sum, 1: RCCL or fail; default: RCCL when possible */
			status = pc_hip_group_totals(group, (r != NULL && *r != '\0') ? atoi(r) : -1, sum_weights, counters, NULL, &reduced_by, NULL);
		} else {
			status = pc_hip_transmission_wait(ctx, NULL);
			if (status == PC_HIP_OK)
				status = pc_hip_transmission_totals(ctx, sum_weights, counters, NULL);
		}
	}
	t_stage[5] = pc_now_ms();
	if (timing)
		fprintf(stderr, "polycap timing [ms]: alloc+context %.1f, enqueue %.1f, prefault %.1f, images (incl. waiting for the kernel) %.1f, totals %.1f%s\n",
			t_stage[1] - t_stage[0], t_stage[2] - t_stage[1], t_stage[3] - t_stage[2], t_stage[4] - t_stage[3], t_stage[5] - t_stage[4],
			group != NULL ? (reduced_by ? " (devices summed by RCCL all-reduce)" : " (devices summed on the host)") : "");
	if (status == PC_HIP_OK && leak_calc)
		status = pc_transeff_fetch_leaks(eff, ctx, group);      /* reference :925-1032 */
	if (status != PC_HIP_OK) {
		pc_set_hip_error(error, "polycap_source_get_transmission_efficiencies", status);
		free(sum_weights);
		polycap_transmission_efficiencies_free(eff);
		return NULL;
	}

	/* totals, summary lines and efficiency formula of the reference, :1055-1076 */
	int64_t sum_iexit = counters[0], sum_not_entered = counters[1], sum_not_transmitted = counters[2], sum_irefl = counters[3];
	printf("Average number of reflections: %lf, Simulated photons: %" PRId64 "\n", (double)sum_irefl/n_photons, sum_iexit+sum_not_entered+sum_not_transmitted);
	printf("Open area Calculated: %lf, Simulated: %lf\n",
		((pc_n_shells(description->n_cap)+0.5)*6.)*((pc_n_shells(description->n_cap)+0.5)*6.)/12.*(description->profile->cap[0]*description->profile->cap[0]*M_PI)/(3.*sin(M_PI/3)*description->profile->ext[0]*description->profile->ext[0]),
		(double)(sum_iexit+sum_not_transmitted)/(sum_iexit+sum_not_entered+sum_not_transmitted));
	printf("iexit: %" PRId64 ", no enter: %" PRId64 ", no trans: %" PRId64 "\n", sum_iexit, sum_not_entered, sum_not_transmitted);

	pc_transeff_finish(eff, sum_weights, counters);
	eff->synthetic_constants = source->cache.synthetic;
	if (!keep_images)
		eff->images->i_exit = 0;     /* no per-photon planes were kept: the exit/start getters report no events */
	free(sum_weights);
	return eff;
}

void polycap_source_free(polycap_source *source)
{
	if (source == NULL)
		return;
	pc_ctx_cache_clear(&source->cache);
	polycap_description_free(source->description);
	polycap_rng_free(source->rng);
	free(source->energies);
	free(source);
}

const polycap_description *polycap_source_get_description(polycap_source *source)
{
	return source->description;
}

/* Fills *p with the plain-array view of `source` (pointers borrowed from the source, valid while it lives);
 * amu/scatf are computed by the optical-constants provider into caller-owned arrays of n_energies doubles.
 * Used by the Python layer to build problems from .inp decks with the one C parser. Returns 0 on success. */
POLYCAP_EXTERN int pc_source_problem(polycap_source *source, pc_hip_problem *p, double *amu, double *scatf, int *synthetic, polycap_error **error);
int pc_source_problem(polycap_source *source, pc_hip_problem *p, double *amu, double *scatf, int *synthetic, polycap_error **error)
{
	if (source == NULL || p == NULL || source->description == NULL || source->description->profile == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "pc_source_problem: source and p cannot be NULL");
		return -1;
	}
	polycap_description *d = source->description;
	memset(p, 0, sizeof(*p));
	p->nmax = d->profile->nmax;
	p->z = d->profile->z; p->cap = d->profile->cap; p->ext = d->profile->ext;
	p->sig_rough = d->sig_rough; p->n_cap = d->n_cap; p->density = d->density;
	p->n_energies = source->n_energies; p->energies = source->energies;
	p->d_source = source->d_source; p->src_x = source->src_x; p->src_y = source->src_y;
	p->src_sigx = source->src_sigx; p->src_sigy = source->src_sigy;
	p->src_shiftx = source->src_shiftx; p->src_shifty = source->src_shifty; p->hor_pol = source->hor_pol;
	if (amu != NULL && scatf != NULL) {
		if (pc_optconst_scatf(d->nelem, d->iz, d->wi, d->density, source->n_energies, source->energies, amu, scatf, synthetic, error) != 0)
			return -1;
		p->amu = amu; p->scatf = scatf;
	}
	return 0;
}
