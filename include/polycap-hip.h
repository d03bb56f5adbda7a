/*
 * polycap-hip.h -- the thin C-ABI between libpolycap's host C code and its HIP (gfx950) kernels.
 *
 * Plain C: pointers, sizes and ints only; nothing HIP- or torch-typed crosses this boundary, so the
 * same entry points can be bound from C, ctypes, Cython or cgo.  The reference has no such layer (it
 * has no GPU code); each entry point names the reference function whose work it takes over.
 *
 * All functions return 0 on success or a negative pc_hip_status; pc_hip_last_error() gives the text.
 * There is no CPU fallback behind any of them: without a usable HIP device they fail with
 * PC_HIP_ERR_NO_DEVICE.
 */
#ifndef POLYCAP_HIP_H
#define POLYCAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifndef POLYCAP_EXTERN
#define POLYCAP_EXTERN __attribute__((visibility("default"))) extern
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	PC_HIP_OK = 0,
	PC_HIP_ERR_NO_DEVICE = -1,   /* no HIP device / runtime not usable */
	PC_HIP_ERR_INVALID = -2,     /* bad argument or unsupported problem shape */
	PC_HIP_ERR_RUNTIME = -3,     /* a HIP API call or a kernel failed */
	PC_HIP_ERR_MEMORY = -4,      /* host or device allocation failed */
	PC_HIP_ERR_ATTEMPTS = -5     /* some slot used max_attempts launches without a transmitted photon */
} pc_hip_status;

/* One simulation problem: optic geometry + glass + energy tables + X-ray source.
 * Plain-array mirror of the reference's _polycap_profile / _polycap_description / _polycap_source
 * (src/polycap-private.h:88-122).  amu[]/scatf[] are what polycap_photon_scatf()
 * (src/polycap-photon.c:22-94) computes per launch; here they are computed once by the host.
 * Everything is copied at pc_hip_ctx_create(); the caller keeps ownership. */
typedef struct {
	int32_t nmax;                 /* profile arrays hold nmax+1 points, z strictly increasing, z[0] >= 0 */
	const double *z, *cap, *ext;
	double sig_rough;
	int64_t n_cap;
	double density;
	size_t n_energies;
	const double *energies, *amu, *scatf;
	double d_source, src_x, src_y, src_sigx, src_sigy, src_shiftx, src_shifty, hor_pol;
} pc_hip_problem;

/* Host destination planes for per-exit-photon "images": the SoA of struct _polycap_images
 * (src/polycap-private.h:156-181), leak fields excluded.  Any pointer may be NULL to skip that plane. */
typedef struct {
	double *src_start_coords[2];
	double *pc_start_coords[2];
	double *pc_start_dir[2];
	double *pc_start_elecv[2];
	double *pc_exit_coords[3];
	double *pc_exit_dir[2];
	double *pc_exit_elecv[2];
	int64_t *pc_exit_nrefl;
	double *pc_exit_dtravel;
	double *exit_coord_weights;   /* [count * n_energies], row-major by photon */
} pc_hip_images;

typedef struct pc_hip_ctx pc_hip_ctx;

POLYCAP_EXTERN int pc_hip_device_count(void);
POLYCAP_EXTERN const char *pc_hip_last_error(void);

/* Uploads the problem to `device` (tables resident in HBM, staged into LDS by every workgroup). */
POLYCAP_EXTERN int pc_hip_ctx_create(const pc_hip_problem *problem, int device, pc_hip_ctx **ctx);
POLYCAP_EXTERN void pc_hip_ctx_destroy(pc_hip_ctx *ctx);

/* Tuning / test switches (none of them changes a result):
 *   "literal_march"    1 = visit every segment with the reference's full quadratic, 0 = certified skipping (default)
 *   "event_threshold", "march_stop", "new_threshold", "march_burst", "blocks_per_cu", "block_size"   scheduler / launch shape:
 *                      a MARCH burst starts when event_threshold lanes march (48) and goes on while march_stop do (8)
 *   "lds_ec"           many energies: per-energy constants staged in LDS (default 1)
 *   "producer"         single-energy source runs with a launching wave per workgroup (pc_producer_kernel.h): 1 always, 0 never,
 *                      -1 (default) when photons live long enough (see pc_hip_last_kernel)
 *   "pool", "pool_refill", "pool_march_min", "pool_event_min", "pool_new_min"   single-energy source runs: the kernel
 *                      that parks 64 more photons per wave in LDS (pc_pool_kernel.h); "pool" 0 by default = one photon per lane
 *   "run_parts"        a transmission run that keeps images is traced as this many consecutive launches on two streams,
 *                      so that pc_hip_transmission_images can fetch finished parts while later ones run (default 1;
 *                      polycap_source_get_transmission_efficiencies uses 4 from 2e6 photons on)
 *   "plane_images"     1 = a run that keeps images stores the planes of pc_hip_images itself (no records): pc_hip_transmission_images
 *                      is then a copy-engine transfer into the caller's planes, pinned for the duration of the call;
 *                      pc_hip_transmission_records is not available for such a run.  0 (default) = one record per slot,
 *                      turned into planes on the device when pc_hip_transmission_images asks for them
 *   "compact_images"   with "plane_images": 1 = exit photons are stored in the order in which they leave the optic instead of
 *                      at the position of their slot -- the photons a wave finalises together are one coalesced run per plane
 *                      (the reference's order of photons in its arrays is as arbitrary: it is the order in which OpenMP
 *                      threads with random seeds happen to fill them) -- and the planes are published in blocks of
 *                      2^"block_shift" positions (default 16) while the kernel runs: pc_hip_transmission_images copies the
 *                      blocks that are complete, as many at a time as have piled up.  The set of photons is the same as with 0 (default), bit for bit.
 *   "keep_pinned"      pc_hip_transmission_images leaves the destination planes it pinned (hipHostRegister) pinned: the
 *                      caller reuses them for later runs and unpins them with pc_hip_host_unregister before freeing them
 *   "slot_ids"         compact runs also record which slot sits at which position (pc_hip_transmission_slot_ids)
 *   "leak_order"       leak_calc source runs with two to five slots per lane hand out their slots heaviest first, predicted by
 *                      a plain pre-pass of the same slots, the heaviest n/400 to lanes of their own (default 1; 0 = slot order)
 *   "leak_heavy_lanes", "leak_heavy_every"   which lanes the heaviest slots go to: lanes 0 .. n-1 of every m-th wave (defaults 1, 1)
 *   "leak_slot_units"  leak_calc source runs keep the units of work per slot (pc_hip_leak_slot_units)
 *   "batch_reflections" source runs with more than 8 energies: 1 (default) reflections are logged (24 B each) and a photon's
 *                      weights swept once per log (any energy count whose sums and constants leave room in LDS for a log per
 *                      wave: to ~1400), 0 every reflection sweeps the weights at once
 *   "log_cap"          reflections per log of the logging kernel (1..255; default 0 = 64 from 64 energies on, 32 below, halved
 *                      while a log per wave does not fit beside the constants of more than ~450 energies)
 *   "log_min_energies" fewest energies of a source run that logs its reflections (default 9 = every run whose weights are
 *                      not in registers; >= 9)
 *   "sweep_skip"       histogram-only runs of the logging kernel stop multiplying a weight once it is below 2^-64 (it adds
 *                      nothing to the exact sums any more; default 1)
 *   "flush_max"        the logging kernel lets up to this many finished photons of a wave wait for a common sweep (default 8; the
 *                      count is chosen so that the last pass of a sweep is nearly full: 3 at 291 energies)
 *   "sweep_fuse"       histogram-only runs of the logging kernel: the sweep of a photon that has left the optic adds its weights
 *                      to the sums itself, no weight row is written (default 1; 2: also for photons whose proxy energies are
 *                      dead, which exercises the exact take-back pass; 0: off)
 *   "fetch_threads"    host threads of the staging fallback of the image fetch (0 = min(16, cores))
 *   leak runs: "leak_max_depth" (stack frames per lane = walls one photon may cross), "leak_stack_mb" (HBM for those
 *                      stacks), "leak_capacity" (leak record buffer, 0 = automatic; a run that outgrows it is repeated). */
POLYCAP_EXTERN int pc_hip_set_option(pc_hip_ctx *ctx, const char *name, int64_t value);

/* polycap_photon_launch (src/polycap-photon.c:390-955, leak_calc=false) for n explicit photons.
 * Inputs [3*n] xyz-interleaved host arrays; outputs rc[n] in {1,0,2,-2,-1}, weights[n*n_energies],
 * exit_*[3*n] (state at the last interaction, as polycap_photon_get_exit_*), i_refl[n], d_travel[n]. */
POLYCAP_EXTERN int pc_hip_launch_photons(pc_hip_ctx *ctx, int64_t n,
	const double *start_coords, const double *start_dir, const double *start_elecv,
	int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
	int64_t *i_refl, double *d_travel);

/* polycap_source_get_photon (src/polycap-source.c:23-144) evaluated on the device for the given
 * (slot, attempt) pairs of the Philox stream `seed`; out[12*n] = start(3), dir(3), elecv(3), src_start(3). */
POLYCAP_EXTERN int pc_hip_sample_photons(pc_hip_ctx *ctx, uint64_t seed, int64_t n,
	const int64_t *slots, const uint32_t *attempts, double *out);

/* polycap_source_get_transmission_efficiencies (src/polycap-source.c:448-1087, leak_calc=false) for the
 * exit-photon slots [slot0, slot0+n_slots): enqueue on the context's stream; results stay in HBM.
 * keep_images=0 is the histogram-only mode (no per-photon planes are allocated or written). */
POLYCAP_EXTERN int pc_hip_transmission_run(pc_hip_ctx *ctx, uint64_t seed, int64_t slot0, int64_t n_slots,
	uint32_t max_attempts, int keep_images);
/* Waits for the stream; *kernel_ms (optional) = duration of the trace kernel between two HIP events
 * recorded on that stream. */
POLYCAP_EXTERN int pc_hip_transmission_wait(pc_hip_ctx *ctx, float *kernel_ms);
/* Totals of the last run: sum_weights[n_energies]; counters = {iexit, not_entered, not_transmitted,
 * sum_irefl, failed_slots, launches}; sumw_fixed (optional) [2*n_energies] = exact 128-bit fixed-point sums
 * (lo, hi) in units of 2^-62, which add exactly across devices. */
POLYCAP_EXTERN int pc_hip_transmission_totals(pc_hip_ctx *ctx, double *sum_weights, int64_t counters[6], uint64_t *sumw_fixed);
/* Copies image planes of slots [first, first+count) (relative to slot0 of the last run) to the host.  May be called
 * before pc_hip_transmission_wait: with "run_parts" > 1 it waits for the run part by part and copies the finished parts
 * while the later ones are traced (pinned staging, host threads build the planes).  NULL planes are skipped. */
POLYCAP_EXTERN int pc_hip_transmission_images(pc_hip_ctx *ctx, int64_t first, int64_t count, const pc_hip_images *dst);
/* The same data as the device keeps it: one record of PC_HIP_N_PLANES + n_energies doubles per slot, the planes of
 * pc_hip_images in their order (the reflection count as int64 bits in its place), then the slot's weights.
 * records: [count][PC_HIP_N_PLANES + n_energies]. */
#define PC_HIP_N_PLANES 17
POLYCAP_EXTERN int pc_hip_transmission_records(pc_hip_ctx *ctx, int64_t first, int64_t count, double *records);
/* Unpins host memory that a fetch with option "keep_pinned" left pinned. */
POLYCAP_EXTERN void pc_hip_host_unregister(void *ptr);
/* Slot (relative to slot0 of the last run) of the photon stored at positions [first, first+count) of the image planes:
 * the identity unless the run was compact ("compact_images" with "slot_ids"). */
POLYCAP_EXTERN int pc_hip_transmission_slot_ids(pc_hip_ctx *ctx, int64_t first, int64_t count, int64_t *slots);
/* leak_calc runs (reference: the static split of the photon loop over OpenMP threads, src/polycap-source.c:744): the order in
 * which the next source runs of exactly n slots hand out their slots -- a permutation of 0 .. n-1, heaviest slot first; the
 * first n_heavy of them are traced by a few lanes of every fourth wave (options leak_heavy_lanes, leak_heavy_every).  n = 0
 * restores slot order.  Results do not depend on the order.  pc_hip_leak_slot_units: units of work the last leak run (option
 * leak_slot_units = 1) spent on each slot. */
POLYCAP_EXTERN int pc_hip_leak_set_order(pc_hip_ctx *ctx, const uint32_t *order, int64_t n, int64_t n_heavy);
POLYCAP_EXTERN int pc_hip_leak_slot_units(pc_hip_ctx *ctx, int64_t first, int64_t count, uint32_t *units);

/* ---- leak_calc = true ("halo" photons): src/polycap-capil.c:610-619, 657-1194, src/polycap-photon.c:171-362, 645-907,
 * src/polycap-source.c:799-879, 925-1032.  Same calls with the fraction of every reflection that is transmitted through
 * the glass followed as well; the leak events of the run are kept by the context until the next run.
 * One event = PC_HIP_LEAK_HDR + n_energies doubles: slot (photon index for pc_hip_launch_photons_leak), attempt,
 * coords xyz, direction xyz, electric vector xyz, n_refl, weights[n_energies] (struct _polycap_leak,
 * include/polycap-photon.h:40-47).  Order = the reference's lists: by slot; inside a slot the events of the transmitted
 * photon first, then those of the earlier attempts. */
#define PC_HIP_LEAK_HDR 12
POLYCAP_EXTERN int pc_hip_launch_photons_leak(pc_hip_ctx *ctx, int64_t n,
	const double *start_coords, const double *start_dir, const double *start_elecv,
	int32_t *rc, double *weights, double *exit_coords, double *exit_dir, double *exit_elecv,
	int64_t *i_refl, double *d_travel);
POLYCAP_EXTERN int pc_hip_transmission_run_leak(pc_hip_ctx *ctx, uint64_t seed, int64_t slot0, int64_t n_slots,
	uint32_t max_attempts, int keep_images);
/* number of extleak (left the optic through its side) / intleak (reached the exit plane inside the glass) events */
POLYCAP_EXTERN int pc_hip_leak_counts(pc_hip_ctx *ctx, int64_t *n_ext, int64_t *n_int);
/* events [first, first+count) of kind 0 (extleak) or 1 (intleak) into records[count * (PC_HIP_LEAK_HDR + n_energies)] */
POLYCAP_EXTERN int pc_hip_leak_events(pc_hip_ctx *ctx, int kind, int64_t first, int64_t count, double *records);
/* the same without a copy: *records points at the context's own (pinned) list of *count events of that kind, in the reference's
 * list order (put into it on the device); valid until the context's next leak run or its destruction */
POLYCAP_EXTERN int pc_hip_leak_events_view(pc_hip_ctx *ctx, int kind, const double **records, int64_t *count);

/* Waits until the context's device is idle (hipDeviceSynchronize): every stream, not only the context's own. */
POLYCAP_EXTERN int pc_hip_device_synchronize(pc_hip_ctx *ctx);

/* ---- several devices from one process: the OpenMP team of the reference (src/polycap-source.c:697-745) becomes a group of
 * device contexts, one per entry of `devices` (an index may repeat: the contexts then share that GPU).  A run shards the
 * exit-photon slots [0, n_slots) into contiguous ranges, one per member, enqueued from the calling thread; photons are
 * keyed by (seed, global slot), so the result does not depend on the partition.  The totals of the members -- counters and
 * the exact 128-bit fixed-point weight sums, split into 32-bit limbs -- are added by ONE all-reduce over RCCL
 * (librccl bound at run time, ncclCommInitAll over the group's devices: the reference's omp critical sum, :973-980) when
 * the devices are distinct and RCCL is present, else limb by limb on the host: the same bits either way. */
typedef struct pc_hip_group pc_hip_group;
POLYCAP_EXTERN int pc_hip_group_create(const pc_hip_problem *problem, int n_devices, const int *devices, pc_hip_group **group);
POLYCAP_EXTERN void pc_hip_group_destroy(pc_hip_group *group);
POLYCAP_EXTERN int pc_hip_group_size(const pc_hip_group *group);
POLYCAP_EXTERN int pc_hip_group_set_option(pc_hip_group *group, const char *name, int64_t value);
POLYCAP_EXTERN int pc_hip_group_run(pc_hip_group *group, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images);
/* leak_calc = true over the group: every member traces its contiguous slot range and orders its events on its own device; the
 * group's event lists are the members' lists one after the other (= slot order, as one device produces them).  Images and
 * totals through pc_hip_group_images / pc_hip_group_totals. */
POLYCAP_EXTERN int pc_hip_group_run_leak(pc_hip_group *group, uint64_t seed, int64_t n_slots, uint32_t max_attempts, int keep_images);
POLYCAP_EXTERN int pc_hip_group_leak_counts(pc_hip_group *group, int64_t *n_ext, int64_t *n_int);
POLYCAP_EXTERN int pc_hip_group_leak_events(pc_hip_group *group, int kind, int64_t first, int64_t count, double *records);
/* kernel that traced member k's share of the last run (as pc_hip_last_kernel) */
POLYCAP_EXTERN int pc_hip_group_last_kernel(pc_hip_group *group, int k);
/* image planes of all n_slots slots of the last run (one host thread per member copies its range into dst) */
POLYCAP_EXTERN int pc_hip_group_images(pc_hip_group *group, const pc_hip_images *dst);
/* totals as pc_hip_transmission_totals; *reduced_by (optional) = 1 when the sum was made by RCCL, 0 on the host;
 * *kernel_ms (optional) = the longest member kernel.  reduce: -1 automatic, 0 host, 1 RCCL (fails when it cannot) */
POLYCAP_EXTERN int pc_hip_group_totals(pc_hip_group *group, int reduce, double *sum_weights, int64_t counters[6], uint64_t *sumw_fixed,
	int *reduced_by, float *kernel_ms);

/* Scheduler statistics of the last transmission run (diagnostics): {march steps, march lane-steps, event phases,
 * event lanes, new phases, new lanes}, summed over all waves; lanes/phases = average active lanes per phase. */
POLYCAP_EXTERN int pc_hip_phase_stats(pc_hip_ctx *ctx, int64_t stats[6]);
/* Which kernel traced the last source run: 0 one photon per lane (pc_trace_kernel), 1 LDS photon pool (option "pool"),
 * 2 launching wave per workgroup (option "producer"; by default chosen when the photons of the context's last run made at
 * least 4 segment visits (reflections, mostly; absorbed photons included) per launch -- a first run of 2e6 slots or more is preceded by a 32768-slot probe),
 * 3 one wave per photon (experiment builds only), 4 logged reflections (pc_trace_log_kernel: source runs with more than 8
 * energies, option "batch_reflections" 1), 5 leak_calc runs (pc_leak_kernel; explicit-photon leak launches included).  -1: none yet. */
POLYCAP_EXTERN int pc_hip_last_kernel(pc_hip_ctx *ctx);
/* The weight sweeps of the last run when pc_trace_log_kernel traced it: stats = {wave-level passes over 64 (photon, energy)
 * pairs, wave-level (pass, reflection) iterations, sum over the waves of their lifetimes in shader clock ticks, the longest
 * lifetime (mean / longest = how evenly the waves finished)}; *ct_tame (optional) = the grazing cosine above which the host certified
 * every energy's reflectivity inside [0, 1 - 1e-11] (-1: no log run yet); proxies (optional) = the one or two energy indices
 * every lane follows itself (-1: none) */
POLYCAP_EXTERN int pc_hip_sweep_stats(pc_hip_ctx *ctx, int64_t stats[4], double *ct_tame, int proxies[2]);

/* 1 when the host address lies in memory registered with (pinned by) the HIP runtime */
POLYCAP_EXTERN int pc_hip_host_is_pinned(const void *p);

/* efficiency formula of src/polycap-source.c:1066-1076 from (summed) totals */
POLYCAP_EXTERN void pc_hip_efficiencies(size_t n_energies, const double *sum_weights, const int64_t counters[6], double *efficiencies);
/* exact fixed-point (lo,hi) pair -> double */
POLYCAP_EXTERN double pc_hip_fixed_to_double(uint64_t lo, uint64_t hi);


/* Host-side helper with no reference counterpart: builds the polycap_transmission_efficiencies result object from
 * totals and (optional) image planes produced elsewhere -- summed over several GPUs / ranks by the caller -- so the
 * getters and the HDF5 writer serve a sharded run exactly as a single-device one.  `source` is a polycap_source*
 * (it must outlive the result, as in the reference); counters as pc_hip_transmission_totals(); planes hold n_exit
 * entries each (NULL planes read as zeros); the efficiency formula is that of src/polycap-source.c:1066-1076.
 * Returns a polycap_transmission_efficiencies* or NULL with *error (a polycap_error**) set. */
POLYCAP_EXTERN void *pc_transmission_efficiencies_from_totals(void *source, int64_t n_exit, const double *sum_weights,
	const int64_t counters[6], const pc_hip_images *planes, void *error);

#ifdef __cplusplus
}
#endif
#endif
