/*
 * pc_private.h -- internal structures of libpolycap's host side (C11).
 *
 * Field meaning follows the reference's src/polycap-private.h:88-181 (profile, description, source,
 * photon, transmission_efficiencies, images); the HIP context handles are additions of this build.
 */
#ifndef PC_PRIVATE_H
#define PC_PRIVATE_H

#include "polycap.h"

#include <stdio.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define PC_COSPI_6 0.86602540378443864676

struct _polycap_profile {
	int nmax;          /* arrays hold nmax+1 points */
	double *z;
	double *cap;
	double *ext;
};

/* cached device context for one (description, energy grid, source) combination */
typedef struct {
	pc_hip_ctx *ctx;
	pc_hip_group *group;      /* POLYCAP_HIP_DEVICES: one context per listed device instead of `ctx` */
	int n_devices;
	int devices[64];
	int device;               /* device of `ctx` */
	int synthetic;            /* the optical constants of this context come from the built-in tables away from their pinned points */
	size_t n_energies;
	double *energies;
	int has_source;
	double src[8];
} pc_ctx_cache;

struct _polycap_description {
	double sig_rough;
	int64_t n_cap;
	double open_area;
	unsigned int nelem;
	int *iz;
	double *wi;
	double density;
	polycap_profile *profile;
	pc_ctx_cache cache;   /* used by polycap_photon_launch */
};

struct _polycap_rng {
	uint64_t seed;     /* Philox key */
	uint64_t counter;  /* next photon index of this stream */
};

struct _polycap_source {
	polycap_description *description;
	polycap_rng *rng;
	double d_source;
	double src_x;
	double src_y;
	double src_sigx;
	double src_sigy;
	double src_shiftx;
	double src_shifty;
	double hor_pol;
	size_t n_energies;
	double *energies;
	pc_ctx_cache cache;   /* used by get_photon / get_transmission_efficiencies */
	uint64_t run_index;   /* successive get_transmission_efficiencies calls use distinct Philox keys */
};

struct _polycap_photon {
	polycap_description *description;
	polycap_leak **extleak;
	polycap_leak **intleak;
	int64_t n_extleak;
	int64_t n_intleak;
	polycap_vector3 start_coords;
	polycap_vector3 start_direction;
	polycap_vector3 start_electric_vector;
	polycap_vector3 exit_coords;
	polycap_vector3 exit_direction;
	polycap_vector3 exit_electric_vector;
	polycap_vector3 src_start_coords;
	size_t n_energies;
	double *energies;
	double *weight;
	double *amu;
	double *scatf;
	int64_t i_refl;
	double d_travel;
};

struct _polycap_images {
	/* The 17 planes + the weights of a result with images are ONE allocation ("slab", pc_transeff.c): plane k at
	 * slab + k*slab_stride bytes, in the order of pc_hip_images, the weights last -- so that the device's planes cross PCIe as
	 * one pitched copy per group of blocks and the whole result is pinned in one piece.  NULL: every plane on its own. */
	void *slab;
	size_t slab_stride;
	int64_t i_start;
	int64_t i_exit;
	double *src_start_coords[2];
	double *pc_start_coords[2];
	double *pc_start_dir[2];
	double *pc_start_elecv[2];
	double *pc_exit_coords[3];
	double *pc_exit_dir[2];
	double *pc_exit_elecv[2];
	int64_t *pc_exit_nrefl;
	double *pc_exit_dtravel;
	double *exit_coord_weights;
	/* leak_calc=true: events that left the optic through its side (extleak) or reached the exit plane inside the glass
	 * (intleak); planes of i_extleak / i_intleak entries, weights row-major by event (reference :168-180) */
	int64_t i_extleak;
	int64_t i_intleak;
	double *extleak_coords[3];
	double *extleak_dir[2];
	int64_t *extleak_n_refl;
	double *extleak_coord_weights;
	double *intleak_coords[3];
	double *intleak_dir[2];
	double *intleak_elecv[2];
	int64_t *intleak_n_refl;
	double *intleak_coord_weights;
};

struct _polycap_transmission_efficiencies {
	size_t n_energies;
	double *energies;
	double *efficiencies;
	struct _polycap_images *images;
	polycap_source *source;
	int synthetic_constants;   /* extension: see pc_transmission_efficiencies_synthetic */
};

/* internal helpers */
char *polycap_read_input_line(FILE *fptr, polycap_error **error);
void polycap_description_check_weight(size_t nelem, double wi[], polycap_error **error);
double pc_n_shells(int64_t n_cap);
int polycap_photon_within_pc_boundary(double polycap_radius, polycap_vector3 photon_coord, polycap_error **error);

/* optical constants (what polycap_photon_scatf computes in the reference, src/polycap-photon.c:22-94) */
POLYCAP_EXTERN int pc_optconst_scatf(unsigned int nelem, const int *iz, const double *wi, double density,
	size_t n_energies, const double *energies, double *amu, double *scatf, int *synthetic, polycap_error **error);
POLYCAP_EXTERN const char *pc_optconst_provider(void);
POLYCAP_EXTERN const char *pc_optconst_library(void);
/* 1 when the efficiencies were computed with optical constants from the built-in tables away from the points the reference's
 * tests pin (no xraylib on the machine); 0 with xraylib or at the pinned points */
POLYCAP_EXTERN int pc_transmission_efficiencies_synthetic(const polycap_transmission_efficiencies *efficiencies);
/* name of the HDF5 shared library bound at run time by the result writer, or "none" */
POLYCAP_EXTERN const char *pc_hdf5_provider(void);

/* result object construction (pc_transeff.c) */
/* zeroed: planes zeroed like calloc's (0: the caller overwrites every entry; planes reused from the pool keep old contents) */
polycap_transmission_efficiencies *pc_transeff_alloc(polycap_source *source, size_t np, int zeroed, const char *caller, polycap_error **error);
void pc_transeff_planes_pinned(polycap_transmission_efficiencies *eff);
POLYCAP_EXTERN void pc_host_pool_clear(void);
POLYCAP_EXTERN int pc_transmission_efficiencies_slab(const polycap_transmission_efficiencies *efficiencies, void **base, size_t *stride);
void pc_transeff_prefault(polycap_transmission_efficiencies *eff, size_t np);
void pc_transeff_plane_pointers(polycap_transmission_efficiencies *eff, pc_hip_images *dst);
void pc_transeff_finish(polycap_transmission_efficiencies *eff, const double *sum_weights, const int64_t counters[6]);
int pc_transeff_fetch_leaks(polycap_transmission_efficiencies *eff, pc_hip_ctx *ctx, pc_hip_group *group);   /* one of ctx / group; returns a pc_hip_status */

/* leak events of the last leak_calc run of `ctx` as polycap_leak lists (pc_photon.c); *list is malloc'd, NULL when empty */
int pc_fetch_leaks(pc_hip_ctx *ctx, int kind, size_t n_energies, polycap_leak ***list, int64_t *n, int64_t **slots,
	const char *caller, polycap_error **error);
void pc_leak_list_free(polycap_leak **list, int64_t n);
bool pc_leak_list_copy(polycap_leak **src, int64_t n_src, polycap_leak ***leaks, int64_t *n_leaks, const char *caller, const char *what,
	polycap_error **error);

/* device context management */
void pc_ctx_cache_clear(pc_ctx_cache *c);
pc_hip_ctx *pc_ctx_for(pc_ctx_cache *c, polycap_description *description, size_t n_energies, const double *energies,
	const polycap_source *source, const char *caller, polycap_error **error);
/* the same on an explicit device (< 0: POLYCAP_HIP_DEVICE, default 0); the cache is keyed on the device too */
pc_hip_ctx *pc_ctx_for_device(pc_ctx_cache *c, polycap_description *description, size_t n_energies, const double *energies,
	const polycap_source *source, int device, const char *caller, polycap_error **error);
/* the same for a device list (POLYCAP_HIP_DEVICES): a group of contexts, one per entry */
pc_hip_group *pc_group_for(pc_ctx_cache *c, polycap_description *description, size_t n_energies, const double *energies,
	const polycap_source *source, int n_devices, const int *devices, const char *caller, polycap_error **error);
void pc_set_hip_error(polycap_error **error, const char *caller, int status);

#endif
