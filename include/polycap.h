/*
 * polycap.h -- public C API of libpolycap (MI355X build).
 *
 * This is the drop-in boundary: every POLYCAP_EXTERN symbol of the reference's
 * include/ headers (v1.2) is declared here with the same name, argument order, types and
 * error behaviour.  The reference splits the declarations over nine headers
 * (polycap-error.h, -profile.h, -description.h, -photon.h, -rng.h, -source.h,
 * -transmission-efficiencies.h, -progress-monitor.h, polycap.h); the per-module
 * headers next to this file simply include this one, so both
 * `#include <polycap.h>` and `#include <polycap-photon.h>` keep working.
 *
 * Each block cites the reference header it replaces (file:line, reference v1.2).
 * The trace path behind polycap_photon_launch() and
 * polycap_source_get_transmission_efficiencies() runs on the GPU through the thin
 * C-ABI layer declared in polycap-hip.h; there is no CPU implementation of it in
 * this library.
 */
#ifndef POLYCAP_H
#define POLYCAP_H

#include <stdbool.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>

#ifndef POLYCAP_EXTERN
#define POLYCAP_EXTERN __attribute__((visibility("default"))) extern
#endif

#if defined(__GNUC__)
#define POLYCAP_PRINTF(fmt_idx, arg_idx) __attribute__((__format__(__printf__, fmt_idx, arg_idx)))
#else
#define POLYCAP_PRINTF(fmt_idx, arg_idx)
#endif

/* reference include/polycap.h:30-50 */
#define POLYCAP_VERSION_MAJOR 1
#define POLYCAP_VERSION_MINOR 2
#define HC 1.23984193E-7       /* h*c [keV*cm] */
#define N_AVOG 6.022098e+23    /* Avogadro constant */
#define R0 2.8179403227e-13    /* classical electron radius [cm] */
#define EPSILON 1.0e-30

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors
 * reference include/polycap-error.h:66-173 (GLib GError convention: `polycap_error **error`
 * may be NULL; it is only written when *error == NULL). */
enum polycap_error_code {
	POLYCAP_ERROR_MEMORY,
	POLYCAP_ERROR_INVALID_ARGUMENT,
	POLYCAP_ERROR_IO,
	POLYCAP_ERROR_OPENMP,
	POLYCAP_ERROR_TYPE,
	POLYCAP_ERROR_UNSUPPORTED,
	POLYCAP_ERROR_RUNTIME,
};

typedef struct {
	enum polycap_error_code code;
	char *message;
} polycap_error;

POLYCAP_EXTERN polycap_error *polycap_error_new(enum polycap_error_code code, const char *format, ...) POLYCAP_PRINTF(2, 3);
POLYCAP_EXTERN polycap_error *polycap_error_new_literal(enum polycap_error_code code, const char *message);
POLYCAP_EXTERN polycap_error *polycap_error_new_valist(enum polycap_error_code code, const char *format, va_list args) POLYCAP_PRINTF(2, 0);
POLYCAP_EXTERN void polycap_error_free(polycap_error *error);
POLYCAP_EXTERN polycap_error *polycap_error_copy(const polycap_error *error);
POLYCAP_EXTERN bool polycap_error_matches(const polycap_error *error, enum polycap_error_code code);
POLYCAP_EXTERN void polycap_set_error(polycap_error **err, enum polycap_error_code code, const char *format, ...) POLYCAP_PRINTF(3, 4);
POLYCAP_EXTERN void polycap_set_error_literal(polycap_error **err, enum polycap_error_code code, const char *message);
POLYCAP_EXTERN void polycap_propagate_error(polycap_error **dest, polycap_error *src);
POLYCAP_EXTERN void polycap_clear_error(polycap_error **err);

/* ------------------------------------------------------------------ profile
 * reference include/polycap-profile.h:36-123 */
typedef enum {
	POLYCAP_PROFILE_CONICAL,
	POLYCAP_PROFILE_PARABOLOIDAL,
	POLYCAP_PROFILE_ELLIPSOIDAL,
} polycap_profile_type;

struct _polycap_profile;
typedef struct _polycap_profile polycap_profile;

POLYCAP_EXTERN polycap_profile *polycap_profile_new(polycap_profile_type type, double length,
	double rad_ext_upstream, double rad_ext_downstream, double rad_int_upstream, double rad_int_downstream,
	double focal_dist_upstream, double focal_dist_downstream, polycap_error **error);
POLYCAP_EXTERN polycap_profile *polycap_profile_new_from_file(const char *single_cap_profile_file,
	const char *central_axis_file, const char *external_shape_file, polycap_error **error);
/* declared without POLYCAP_EXTERN in the reference (include/polycap-profile.h:107) but used by its tests */
POLYCAP_EXTERN int polycap_profile_validate(polycap_profile *profile, int64_t n_cap, polycap_error **error);
POLYCAP_EXTERN polycap_profile *polycap_profile_new_from_arrays(int nid, double *ext, double *cap, double *z, polycap_error **error);
POLYCAP_EXTERN bool polycap_profile_get_ext(polycap_profile *profile, size_t *nid, double **ext, polycap_error **error);
POLYCAP_EXTERN bool polycap_profile_get_cap(polycap_profile *profile, size_t *nid, double **cap, polycap_error **error);
POLYCAP_EXTERN bool polycap_profile_get_z(polycap_profile *profile, size_t *nid, double **z, polycap_error **error);
POLYCAP_EXTERN void polycap_profile_free(polycap_profile *profile);

/* ------------------------------------------------------------------ description
 * reference include/polycap-description.h:36-76 */
struct _polycap_description;
typedef struct _polycap_description polycap_description;

POLYCAP_EXTERN polycap_description *polycap_description_new(polycap_profile *profile, double sig_rough, int64_t n_cap,
	unsigned int nelem, int iz[], double wi[], double density, polycap_error **error);
POLYCAP_EXTERN const polycap_profile *polycap_description_get_profile(polycap_description *description);
POLYCAP_EXTERN void polycap_description_free(polycap_description *description);

/* ------------------------------------------------------------------ photon
 * reference include/polycap-photon.h:34-203 */
typedef struct {
	double x;
	double y;
	double z;
} polycap_vector3;

struct _polycap_photon;
typedef struct _polycap_photon polycap_photon;

typedef struct {
	polycap_vector3 coords;
	polycap_vector3 direction;
	polycap_vector3 elecv;
	size_t n_energies;
	double *weight;
	int64_t n_refl;
} polycap_leak;

POLYCAP_EXTERN polycap_photon *polycap_photon_new(polycap_description *description, polycap_vector3 start_coords,
	polycap_vector3 start_direction, polycap_vector3 start_electric_vector, polycap_error **error);
/* returns 1 reached the end, 0 absorbed, 2 hit the glass at the entrance, -2 outside the optic, -1 error
 * (reference include/polycap-photon.h:89) */
POLYCAP_EXTERN int polycap_photon_launch(polycap_photon *photon, size_t n_energies, double *energies, double **weights,
	bool leak_calc, polycap_error **error);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_start_coords(polycap_photon *photon);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_start_direction(polycap_photon *photon);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_start_electric_vector(polycap_photon *photon);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_exit_coords(polycap_photon *photon);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_exit_direction(polycap_photon *photon);
POLYCAP_EXTERN polycap_vector3 polycap_photon_get_exit_electric_vector(polycap_photon *photon);
POLYCAP_EXTERN double polycap_photon_get_dtravel(polycap_photon *photon);
POLYCAP_EXTERN int64_t polycap_photon_get_irefl(polycap_photon *photon);
POLYCAP_EXTERN bool polycap_photon_get_extleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error);
POLYCAP_EXTERN bool polycap_photon_get_intleak_data(polycap_photon *photon, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error);
POLYCAP_EXTERN void polycap_photon_free(polycap_photon *photon);
POLYCAP_EXTERN void polycap_leak_free(polycap_leak *leak);

/* ------------------------------------------------------------------ rng
 * reference include/polycap-rng.h:36-60.  The generator behind it is Philox4x32-10
 * (counter based) instead of GSL's mt19937; no reference test pins stream values. */
struct _polycap_rng;
typedef struct _polycap_rng polycap_rng;

POLYCAP_EXTERN polycap_rng *polycap_rng_new(void);
POLYCAP_EXTERN polycap_rng *polycap_rng_new_with_seed(unsigned long int seed);
POLYCAP_EXTERN void polycap_rng_free(polycap_rng *rng);

/* ------------------------------------------------------------------ results
 * reference include/polycap-transmission-efficiencies.h:36-120 */
struct _polycap_transmission_efficiencies;
typedef struct _polycap_transmission_efficiencies polycap_transmission_efficiencies;

POLYCAP_EXTERN void polycap_transmission_efficiencies_free(polycap_transmission_efficiencies *efficiencies);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_write_hdf5(polycap_transmission_efficiencies *efficiencies, const char *filename, polycap_error **error);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_get_data(polycap_transmission_efficiencies *efficiencies, size_t *n_energies,
	double **energies_arr, double **efficiencies_arr, polycap_error **error);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_get_extleak_data(polycap_transmission_efficiencies *efficiencies,
	polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_get_intleak_data(polycap_transmission_efficiencies *efficiencies,
	polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_get_start_data(polycap_transmission_efficiencies *efficiencies,
	int64_t *n_start, int64_t *n_exit, polycap_vector3 **start_coords, polycap_vector3 **start_direction,
	polycap_vector3 **start_elecv, polycap_vector3 **src_start_coords, polycap_error **error);
POLYCAP_EXTERN bool polycap_transmission_efficiencies_get_exit_data(polycap_transmission_efficiencies *efficiencies,
	int64_t *n_exit, polycap_vector3 **exit_coords, polycap_vector3 **exit_direction, polycap_vector3 **exit_elecv,
	int64_t **n_refl, double **d_travel, size_t *n_energies, double ***exit_weights, polycap_error **error);

/* ------------------------------------------------------------------ progress monitor
 * reference include/polycap-progress-monitor.h:23-27 (opaque, no implementation) */
struct _polycap_progress_monitor;
typedef struct _polycap_progress_monitor polycap_progress_monitor;

/* ------------------------------------------------------------------ source
 * reference include/polycap-source.h:36-128 */
struct _polycap_source;
typedef struct _polycap_source polycap_source;

POLYCAP_EXTERN polycap_source *polycap_source_new(polycap_description *description, double d_source, double src_x, double src_y,
	double src_sigx, double src_sigy, double src_shiftx, double src_shifty, double hor_pol,
	size_t n_energies, double *energies, polycap_error **error);
POLYCAP_EXTERN void polycap_source_free(polycap_source *source);
POLYCAP_EXTERN polycap_photon *polycap_source_get_photon(polycap_source *source, polycap_rng *rng, polycap_error **error);
POLYCAP_EXTERN polycap_source *polycap_source_new_from_file(const char *filename, polycap_error **error);
/* The hot path.  max_threads is accepted for source compatibility and ignored (the photon loop runs on
 * the GPU selected by POLYCAP_HIP_DEVICE, default 0); n_photons = exit photons, as in the reference. */
POLYCAP_EXTERN polycap_transmission_efficiencies *polycap_source_get_transmission_efficiencies(polycap_source *source,
	int max_threads, int n_photons, bool leak_calc, polycap_progress_monitor *progress_monitor, polycap_error **error);
POLYCAP_EXTERN const polycap_description *polycap_source_get_description(polycap_source *source);

/* reference include/polycap.h:61-62 */
POLYCAP_EXTERN void polycap_free(void *data);

#ifdef __cplusplus
}
#endif

#include "polycap-hip.h"

#endif /* POLYCAP_H */
