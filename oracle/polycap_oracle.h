/*
 * polycap_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, fp64) of the reference's per-photon trace path
 *   polycap_source_get_transmission_efficiencies -> polycap_source_get_photon
 *   -> polycap_photon_launch -> polycap_capil_trace / _segment / _reflect / polycap_refl_polar
 * Each function cites the reference file:line it follows (paths relative to the
 * reference checkout, revision v1.2).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this library, and only as the checker / CPU baseline.  The product path
 * (polycap_amd/csrc) never links, imports or calls anything in oracle/.
 *
 * Pinning: the reference itself cannot be compiled in this image without writing
 * stand-ins for config.h (meson-generated), GSL and xraylib (both absent), so no
 * oracle/_ref build exists.  This restatement is pinned by the reference's own
 * known-answer tests (tests/capil.c, tests/photon.c, tests/source.c) -- see
 * tests/test_oracle_known_answers.py and tests/golden/reference_known_answers.json.
 */
#ifndef POLYCAP_ORACLE_H
#define POLYCAP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } orc_vec3;

/* geometry + glass of one optic: src/polycap-private.h:88-108 (_polycap_profile, _polycap_description) */
typedef struct {
	int nmax;            /* profile arrays hold nmax+1 points */
	const double *z;
	const double *cap;
	const double *ext;
	double sig_rough;
	int64_t n_cap;
	double density;
} orc_optic;

/* X-ray source: src/polycap-private.h:110-122 (_polycap_source) */
typedef struct {
	double d_source, src_x, src_y, src_sigx, src_sigy, src_shiftx, src_shifty, hor_pol;
} orc_source;

/* one leak event: include/polycap-photon.h:40-47 (struct _polycap_leak) */
typedef struct {
	orc_vec3 coords, direction, elecv;
	int64_t n_refl;
	size_t n_energies;
	double *weight;
} orc_leak;

/* mutable photon state: src/polycap-private.h:124-145 (_polycap_photon) */
typedef struct {
	orc_vec3 start_coords, start_direction, start_electric_vector;
	orc_vec3 exit_coords, exit_direction, exit_electric_vector;
	orc_vec3 src_start_coords;
	size_t n_energies;
	const double *energies;
	double *weight;
	const double *amu;
	const double *scatf;
	int64_t i_refl;
	double d_travel;
	/* leak_calc=true state (zero otherwise): the flag the reference passes as an argument, and the event lists */
	int leak_calc;
	orc_leak *extleak, *intleak;
	int64_t n_extleak, n_intleak;
} orc_photon;

/* ---- helpers: src/polycap-photon.c:139-169, 365-386 ---- */
void   orc_norm(orc_vec3 *v);
double orc_scalar(orc_vec3 a, orc_vec3 b);
int    orc_within_pc_boundary(double polycap_radius, orc_vec3 coord);
double orc_n_shells(int64_t n_cap);
double orc_open_area(const orc_optic *optic);

/* ---- profile generator: src/polycap-profile.c:66-207 (conical=0, ellipsoidal=2; paraboloidal unsupported -> -1) ---- */
int orc_profile_new(int type, double length, double rad_ext_upstream, double rad_ext_downstream,
                    double rad_int_upstream, double rad_int_downstream,
                    double focal_dist_upstream, double focal_dist_downstream,
                    int nmax, double *z, double *cap, double *ext);

/* ---- kernel functions: src/polycap-capil.c ---- */
int    orc_segment(orc_vec3 cap_coord0, orc_vec3 cap_coord1, double cap_rad0, double cap_rad1,
                   orc_vec3 phot_coord0, orc_vec3 phot_coord1, orc_vec3 photon_dir,
                   orc_vec3 *photon_coord, orc_vec3 *surface_norm);
double orc_refl_polar(double e, double density, double scatf, double lin_abs_coeff,
                      orc_vec3 surface_norm, orc_photon *photon, orc_vec3 *electric_vector);
int    orc_reflect(const orc_optic *optic, orc_photon *photon, orc_vec3 surface_norm);
int    orc_trace(const orc_optic *optic, int *ix, orc_photon *photon, const double *cap_x, const double *cap_y);

/* ---- leak ("halo") path, polycap_oracle_leak.c.  orc_reflect / orc_launch take these branches when
 * photon->leak_calc != 0 ---- */
int  orc_pc_intersect(const orc_optic *optic, orc_vec3 photon_coord, orc_vec3 photon_direction, orc_vec3 *out);
int  orc_trace_wall(const orc_optic *optic, orc_photon *photon, double *d_travel, int *r_cntr, int *q_cntr);
int  orc_reflect_leak(const orc_optic *optic, orc_photon *photon, orc_vec3 surface_norm);
int  orc_launch_in_wall_leak(const orc_optic *optic, orc_photon *photon, double *cap_x, double *cap_y, int *ix);
void orc_leaks_free(orc_leak *list, int64_t n);
void orc_photon_clear_leaks(orc_photon *photon);
void orc_free(void *p);

/* ---- launch: src/polycap-photon.c:390-955.
 * photon must have start_* set; weights[n_energies] is filled. Returns {1,0,2,-2,-1}. ---- */
int orc_launch(const orc_optic *optic, orc_photon *photon, size_t n_energies, const double *energies,
               const double *amu, const double *scatf, double *weights);

/* flat-array convenience for ctypes-based tests: one photon */
int orc_launch_one(const orc_optic *optic, size_t n_energies, const double *energies,
                   const double *amu, const double *scatf,
                   const double start_coords[3], const double start_dir[3], const double start_elecv[3],
                   double *weights, double exit_coords[3], double exit_dir[3], double exit_elecv[3],
                   int64_t *i_refl, double *d_travel);

/* same with leak_calc=true: *ext_records / *int_records receive malloc'd arrays (free with orc_free) of
 * n x (10 + n_energies) doubles: coords(3), direction(3), elecv(3), n_refl, weights[n_energies] */
int orc_launch_one_leak(const orc_optic *optic, size_t n_energies, const double *energies,
                        const double *amu, const double *scatf,
                        const double start_coords[3], const double start_dir[3], const double start_elecv[3],
                        double *weights, double exit_coords[3], double exit_dir[3], double exit_elecv[3],
                        int64_t *i_refl, double *d_travel,
                        double **ext_records, int64_t *n_ext, double **int_records, int64_t *n_int);

/* batch of explicit photons, SoA in (start_xyz[3*n], dir[3*n], elecv[3*n]) AoS-by-photon layout;
 * outputs rc[n], weights[n*n_energies], exit_coords[3n], exit_dir[3n], exit_elecv[3n], i_refl[n], d_travel[n] */
void orc_launch_batch(const orc_optic *optic, size_t n_energies, const double *energies,
                      const double *amu, const double *scatf, int64_t n,
                      const double *start_coords, const double *start_dir, const double *start_elecv,
                      int *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
                      int64_t *i_refl, double *d_travel, int n_threads);

/* ---- counter-based RNG (Philox4x32-10, Salmon et al. SC'11) replacing GSL mt19937 (src/polycap-rng.c) ---- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* d-th uniform in [0,1) of stream (seed, slot, attempt): 53-bit, two per Philox block */
double orc_uniform(uint64_t seed, uint64_t slot, uint32_t attempt, uint32_t d);

/* ---- source sampling: src/polycap-source.c:23-144, RNG draws replaced by orc_uniform(seed,slot,attempt,d++) ---- */
void orc_sample_photon(const orc_optic *optic, const orc_source *source,
                       uint64_t seed, uint64_t slot, uint32_t attempt, orc_photon *photon);
/* flat version: out[12] = start(3), dir(3), elecv(3), src_start(3) */
void orc_sample_photon_flat(const orc_optic *optic, const orc_source *source,
                            uint64_t seed, uint64_t slot, uint32_t attempt, double out[12]);

/* ---- driver: src/polycap-source.c:448-1087 (leak_calc=false) for slots [slot0, slot0+n_slots).
 * sum_weights[n_energies]; counters[4] = {iexit, not_entered, not_transmitted, sum_irefl};
 * img (optional, may be NULL): 17 doubles per slot in the order
 *   src_start xy, pc_start xy, pc_start_dir xy, pc_start_elecv xy, pc_exit xyz, pc_exit_dir xy,
 *   pc_exit_elecv xy, nrefl, dtravel   (nrefl stored as double)
 * exit_weights (optional): n_slots*n_energies.
 * max_attempts bounds the retry loop (reference loops forever); returns 0 ok, -1 if a slot ran out. */
int orc_transmission(const orc_optic *optic, const orc_source *source,
                     size_t n_energies, const double *energies, const double *amu, const double *scatf,
                     uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                     double *sum_weights, int64_t counters[4], double *img, double *exit_weights);
/* + sumw_fixed[2*n_energies] (may be NULL): per energy the exact 128-bit integer sum_j floor(w_j * 2^62) as (lo, hi) */
int orc_transmission_fixed(const orc_optic *optic, const orc_source *source,
                     size_t n_energies, const double *energies, const double *amu, const double *scatf,
                     uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                     double *sum_weights, int64_t counters[4], double *img, double *exit_weights, uint64_t *sumw_fixed);

/* same driver with leak_calc=true (src/polycap-source.c:799-879, 925-1032).  Leak events come back as malloc'd arrays
 * (orc_free) of n x (12 + n_energies) doubles: slot, attempt, then the record of orc_launch_one_leak; ordered by slot,
 * and inside a slot as the reference orders them: events of the transmitted photon first, then those of the earlier
 * attempts (launch return 0 or 2, or 1 outside the exit window) in attempt order. */
int orc_transmission_leak(const orc_optic *optic, const orc_source *source,
                          size_t n_energies, const double *energies, const double *amu, const double *scatf,
                          uint64_t seed, int64_t slot0, int64_t n_slots, int n_threads, uint32_t max_attempts,
                          double *sum_weights, int64_t counters[4], double *img, double *exit_weights,
                          double **ext_records, int64_t *n_ext, double **int_records, int64_t *n_int);

/* internals shared by the two driver files */
typedef struct {
	orc_leak *ext, *intl;            /* events of this slot in final order */
	uint32_t *ext_attempt, *int_attempt;
	int64_t n_ext, n_int;
	orc_leak *ext_temp, *int_temp;   /* events of attempts that did not reach the exit window (:801-839) */
	uint32_t *ext_temp_attempt, *int_temp_attempt;
	int64_t n_ext_temp, n_int_temp;
} orc_slot_leaks;
void orc_slot_leaks_collect(orc_slot_leaks *sl, orc_photon *photon, int iesc, uint32_t attempt);
uint32_t orc_one_slot(const orc_optic *optic, const orc_source *source,
                      size_t n_energies, const double *energies, const double *amu, const double *scatf,
                      uint64_t seed, int64_t j, uint32_t max_attempts,
                      double *w, int64_t cnt[4], double *img, orc_slot_leaks *sl);

/* efficiency formula src/polycap-source.c:1066-1076 */
void orc_efficiencies(size_t n_energies, const double *sum_weights, const int64_t counters[4], double *eff);

#ifdef __cplusplus
}
#endif
#endif
