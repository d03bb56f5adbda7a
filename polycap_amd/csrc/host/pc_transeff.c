/*
 * pc_transeff.c -- polycap_transmission_efficiencies: the result object of the photon loop.
 *
 * Layout and getters follow the reference's src/polycap-transmission-efficiencies.c:782-1172:
 * SoA image planes indexed by exit-photon slot, exit_coord_weights row-major by photon; getters hand out
 * malloc'd copies (caller frees with polycap_free).  After a leak_calc run the leak planes hold the events of the run
 * (extleak: coordinates, direction x/y, reflections, weights; intleak: the electric vector x/y as well).
 * The HDF5 writer (reference :38-780) lives in pc_hdf5.c.
 */
#define _GNU_SOURCE
#include "pc_private.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

/* Image planes: zeroed like calloc's.  Large planes are private anonymous mappings aligned to 2 MB and marked
 * MADV_HUGEPAGE, so that faulting in 1.4 GB of results (1e7 photons) takes hundreds of huge-page faults instead of
 * 350 000 small ones; a 64-byte header in front of every plane tells pc_plane_free how it was made. */
#define PC_PLANE_MAGIC 0x706c616e65733031ull
#define PC_HUGE_PAGE ((size_t)2 << 20)
struct pc_plane_hdr { uint64_t magic; void *base; size_t len; uint64_t mapped; uint64_t pad[4]; };

/* Freed planes of 4 MB and more are kept in a small pool (at most PC_POOL_PLANES planes, PC_POOL_BYTES bytes in all: one result
 * of 1e7 photons) and handed out again to the next result of the same size: their pages are faulted in and -- once a fetch
 * has pinned them for the copy engine (pc_hip_host_register) -- pinned, which saves the next call ~3 ms of hipHostRegister and
 * the page faults of 1.4 GB before its first copy can start.  A pooled plane keeps its old contents: `zeroed` asks for the
 * calloc semantics (memset), callers that overwrite every entry do without.  POLYCAP_HOST_POOL=0 disables the pool. */
#define PC_POOL_PLANES 24
#define PC_POOL_BYTES ((size_t)3 << 30)
struct pc_pool_entry { void *user; size_t bytes; };
static struct pc_pool_entry g_pool[PC_POOL_PLANES];
static int g_pool_n = 0;
static size_t g_pool_bytes = 0;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

static int pc_pool_enabled(void)
{
	const char *e = getenv("POLYCAP_HOST_POOL");
	return !(e != NULL && strcmp(e, "0") == 0);
}

static void *pc_plane_calloc(size_t n, size_t size, int zeroed)
{
	if (size != 0 && n > (SIZE_MAX - 2*PC_HUGE_PAGE)/size) { errno = ENOMEM; return NULL; }
	const size_t bytes = n*size;
	struct pc_plane_hdr h = { PC_PLANE_MAGIC, NULL, 0, 0, {0, 0, 0, 0} };
	char *user;
	if (bytes >= 2*PC_HUGE_PAGE) {
		void *pooled = NULL;
		pthread_mutex_lock(&g_pool_mu);
		for (int k = 0; k < g_pool_n; k++) {
			if (g_pool[k].bytes == bytes) {
				pooled = g_pool[k].user;
				g_pool_bytes -= bytes;
				g_pool[k] = g_pool[--g_pool_n];
				break;
			}
		}
		pthread_mutex_unlock(&g_pool_mu);
		if (pooled != NULL) {
			if (zeroed) memset(pooled, 0, bytes);
			return pooled;
		}
		h.len = bytes + sizeof(h) + PC_HUGE_PAGE;
		h.base = mmap(NULL, h.len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (h.base == MAP_FAILED) return NULL;
		h.mapped = 1;
		h.pad[1] = bytes;
		const uintptr_t first = (uintptr_t)h.base + sizeof(h);
		user = (char *)((first + PC_HUGE_PAGE - 1) & ~(uintptr_t)(PC_HUGE_PAGE - 1));
		(void)madvise(user, bytes, MADV_HUGEPAGE);     /* advisory: without THP the mapping is simply small pages */
	} else {
		h.base = calloc(1, bytes + sizeof(h));
		if (h.base == NULL) return NULL;
		user = (char *)h.base + sizeof(h);
	}
	memcpy(user - sizeof(h), &h, sizeof(h));
	return user;
}

/* header word 0 of a mapped plane: pinned by pc_hip_host_register (stays so while the plane lives or waits in the pool) */
static void pc_plane_set_pinned(void *plane, int pinned)
{
	if (plane == NULL) return;
	struct pc_plane_hdr *h = (struct pc_plane_hdr *)((char *)plane - sizeof(struct pc_plane_hdr));
	if (h->magic == PC_PLANE_MAGIC && h->mapped) h->pad[0] = pinned ? 1 : 0;
}

static int pc_plane_pooled_ready(const void *plane)
{
	if (plane == NULL) return 0;
	const struct pc_plane_hdr *h = (const struct pc_plane_hdr *)((const char *)plane - sizeof(struct pc_plane_hdr));
	return h->magic == PC_PLANE_MAGIC && h->mapped && h->pad[0] == 1;
}

static void pc_plane_free(void *plane)
{
	if (plane == NULL) return;
	struct pc_plane_hdr h;
	memcpy(&h, (char *)plane - sizeof(h), sizeof(h));
	if (h.magic != PC_PLANE_MAGIC) return;      /* not one of ours: leave it alone rather than guess */
	if (h.mapped) {
		const size_t bytes = (size_t)h.pad[1];
		if (pc_pool_enabled()) {
			int kept = 0;
			pthread_mutex_lock(&g_pool_mu);
			if (g_pool_n < PC_POOL_PLANES && g_pool_bytes + bytes <= PC_POOL_BYTES) {
				g_pool[g_pool_n].user = plane; g_pool[g_pool_n].bytes = bytes; g_pool_n++;
				g_pool_bytes += bytes;
				kept = 1;
			}
			pthread_mutex_unlock(&g_pool_mu);
			if (kept) return;
		}
		if (h.pad[0]) pc_hip_host_unregister(plane);
		munmap(h.base, h.len);
	} else {
		free(h.base);
	}
}

/* empties the pool (tests; POLYCAP_HOST_POOL=0 stops refilling it) */
void pc_host_pool_clear(void)
{
	pthread_mutex_lock(&g_pool_mu);
	while (g_pool_n > 0) {
		void *plane = g_pool[--g_pool_n].user;
		struct pc_plane_hdr h;
		memcpy(&h, (char *)plane - sizeof(h), sizeof(h));
		if (h.pad[0]) pc_hip_host_unregister(plane);
		munmap(h.base, h.len);
	}
	g_pool_bytes = 0;
	pthread_mutex_unlock(&g_pool_mu);
}

static void pc_images_free(struct _polycap_images *images)
{
	if (images == NULL)
		return;
	if (images->slab != NULL) {
		/* the planes are parts of the slab */
		pc_plane_free(images->slab);
		memset(images->src_start_coords, 0, sizeof(images->src_start_coords)); memset(images->pc_start_coords, 0, sizeof(images->pc_start_coords));
		memset(images->pc_start_dir, 0, sizeof(images->pc_start_dir)); memset(images->pc_start_elecv, 0, sizeof(images->pc_start_elecv));
		memset(images->pc_exit_coords, 0, sizeof(images->pc_exit_coords)); memset(images->pc_exit_dir, 0, sizeof(images->pc_exit_dir));
		memset(images->pc_exit_elecv, 0, sizeof(images->pc_exit_elecv));
		images->pc_exit_nrefl = NULL; images->pc_exit_dtravel = NULL; images->exit_coord_weights = NULL;
	}
	for (int k = 0; k < 2; k++) {
		pc_plane_free(images->src_start_coords[k]);
		pc_plane_free(images->pc_start_coords[k]);
		pc_plane_free(images->pc_start_dir[k]);
		pc_plane_free(images->pc_start_elecv[k]);
		pc_plane_free(images->pc_exit_dir[k]);
		pc_plane_free(images->pc_exit_elecv[k]);
	}
	for (int k = 0; k < 3; k++)
		pc_plane_free(images->pc_exit_coords[k]);
	pc_plane_free(images->pc_exit_nrefl);
	pc_plane_free(images->pc_exit_dtravel);
	pc_plane_free(images->exit_coord_weights);
	for (int k = 0; k < 3; k++) {
		free(images->extleak_coords[k]);
		free(images->intleak_coords[k]);
	}
	for (int k = 0; k < 2; k++) {
		free(images->extleak_dir[k]);
		free(images->intleak_dir[k]);
		free(images->intleak_elecv[k]);
	}
	free(images->extleak_n_refl);
	free(images->extleak_coord_weights);
	free(images->intleak_n_refl);
	free(images->intleak_coord_weights);
	free(images);
}

/* tests: where a result's image planes live (1: one slab, *base its address, *stride the bytes between two planes) */
int pc_transmission_efficiencies_slab(const polycap_transmission_efficiencies *efficiencies, void **base, size_t *stride)
{
	if (efficiencies == NULL || efficiencies->images == NULL || efficiencies->images->slab == NULL)
		return 0;
	if (base != NULL) *base = efficiencies->images->slab;
	if (stride != NULL) *stride = efficiencies->images->slab_stride;
	return 1;
}

int pc_transmission_efficiencies_synthetic(const polycap_transmission_efficiencies *efficiencies)
{
	return efficiencies != NULL ? efficiencies->synthetic_constants : 0;
}

void polycap_transmission_efficiencies_free(polycap_transmission_efficiencies *efficiencies)
{
	if (efficiencies == NULL)
		return;
	free(efficiencies->energies);
	free(efficiencies->efficiencies);
	pc_images_free(efficiencies->images);
	free(efficiencies);
}

/* result object with every plane allocated for np exit photons (reference: src/polycap-source.c:556-681) */
polycap_transmission_efficiencies *pc_transeff_alloc(polycap_source *source, size_t np, int zeroed, const char *caller, polycap_error **error)
{
	const size_t ne = source->n_energies;
	const size_t nalloc = np ? np : 1;
	polycap_transmission_efficiencies *eff = calloc(1, sizeof(*eff));
	int alloc_ok = (eff != NULL);
	if (alloc_ok) {
		eff->images = calloc(1, sizeof(*eff->images));
		alloc_ok = (eff->images != NULL);
	}
	if (alloc_ok) {
		struct _polycap_images *im = eff->images;
		eff->energies = malloc(sizeof(double)*ne);
		eff->efficiencies = malloc(sizeof(double)*ne);
		/* the planes in the order of pc_hip_images (pc_transeff_plane_pointers): the order the device keeps them in */
		double **planes[] = { &im->src_start_coords[0], &im->src_start_coords[1], &im->pc_start_coords[0], &im->pc_start_coords[1],
			&im->pc_start_dir[0], &im->pc_start_dir[1], &im->pc_start_elecv[0], &im->pc_start_elecv[1],
			&im->pc_exit_coords[0], &im->pc_exit_coords[1], &im->pc_exit_coords[2],
			&im->pc_exit_dir[0], &im->pc_exit_dir[1], &im->pc_exit_elecv[0], &im->pc_exit_elecv[1], NULL /* nrefl */, &im->pc_exit_dtravel };
		const size_t nplanes = sizeof(planes)/sizeof(planes[0]);
		alloc_ok = (eff->energies != NULL && eff->efficiencies != NULL);
		if (nalloc*sizeof(double) >= 2*PC_HUGE_PAGE) {
			/* big result: one slab, every plane at a multiple of 2 MB */
			const size_t stride = (nalloc*sizeof(double) + PC_HUGE_PAGE - 1) & ~(PC_HUGE_PAGE - 1);
			const size_t wbytes = nalloc*ne*sizeof(double);
			char *slab = pc_plane_calloc(nplanes*stride + wbytes, 1, zeroed);
			alloc_ok = alloc_ok && slab != NULL;
			if (slab != NULL) {
				im->slab = slab; im->slab_stride = stride;
				for (size_t k = 0; k < nplanes; k++) {
					if (planes[k] != NULL) *planes[k] = (double *)(slab + k*stride);
					else im->pc_exit_nrefl = (int64_t *)(slab + k*stride);
				}
				im->exit_coord_weights = (double *)(slab + nplanes*stride);
			}
		} else {
			for (size_t k = 0; k < nplanes; k++) {
				if (planes[k] != NULL) { *planes[k] = pc_plane_calloc(nalloc, sizeof(double), zeroed); alloc_ok = alloc_ok && (*planes[k] != NULL); }
				else { im->pc_exit_nrefl = pc_plane_calloc(nalloc, sizeof(int64_t), zeroed); alloc_ok = alloc_ok && im->pc_exit_nrefl != NULL; }
			}
			im->exit_coord_weights = pc_plane_calloc(nalloc*ne, sizeof(double), zeroed);
			alloc_ok = alloc_ok && im->exit_coord_weights != NULL;
		}
	}
	if (!alloc_ok) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for efficiencies -> %s", caller, strerror(errno));
		polycap_transmission_efficiencies_free(eff);
		return NULL;
	}
	eff->source = source;
	eff->n_energies = ne;
	memcpy(eff->energies, source->energies, sizeof(double)*ne);
	return eff;
}

void pc_transeff_plane_pointers(polycap_transmission_efficiencies *eff, pc_hip_images *dst)
{
	const struct _polycap_images *im = eff->images;
	memset(dst, 0, sizeof(*dst));
	for (int k = 0; k < 2; k++) {
		dst->src_start_coords[k] = im->src_start_coords[k];
		dst->pc_start_coords[k] = im->pc_start_coords[k];
		dst->pc_start_dir[k] = im->pc_start_dir[k];
		dst->pc_start_elecv[k] = im->pc_start_elecv[k];
		dst->pc_exit_dir[k] = im->pc_exit_dir[k];
		dst->pc_exit_elecv[k] = im->pc_exit_elecv[k];
	}
	for (int k = 0; k < 3; k++)
		dst->pc_exit_coords[k] = im->pc_exit_coords[k];
	dst->pc_exit_nrefl = im->pc_exit_nrefl;
	dst->pc_exit_dtravel = im->pc_exit_dtravel;
	dst->exit_coord_weights = im->exit_coord_weights;
}

/* Touch every page of the image planes from several host threads.  The planes come zeroed from calloc, i.e. as untouched
 * mappings: their first write costs a page fault + a zeroed page each (350 000 of them for 1e7 photons), which the copy
 * of the results would otherwise pay on one thread.  polycap_source_get_transmission_efficiencies calls this while the
 * GPU traces, so the faults are off the critical path. */
struct pc_touch_job { char *base; size_t bytes; };
struct pc_touch_arg { const struct pc_touch_job *jobs; int njobs, tid, nthreads; };

static void *pc_touch_thread(void *p)
{
	const struct pc_touch_arg *a = p;
	const size_t page = 4096;
	for (int j = 0; j < a->njobs; j++) {
		const size_t npages = (a->jobs[j].bytes + page - 1)/page;
		const size_t lo = npages*(size_t)a->tid/(size_t)a->nthreads, hi = npages*(size_t)(a->tid + 1)/(size_t)a->nthreads;
		volatile char *b = a->jobs[j].base;
		for (size_t k = lo; k < hi; k++)
			b[k*page] = 0;
	}
	return NULL;
}

void pc_transeff_prefault(polycap_transmission_efficiencies *eff, size_t np)
{
	if (eff == NULL || eff->images == NULL || np*sizeof(double) < ((size_t)4 << 20))
		return;
	pc_hip_images d;
	pc_transeff_plane_pointers(eff, &d);
	struct pc_touch_job jobs[18];
	int nj = 0;
	if (eff->images->slab != NULL) {
		if (pc_plane_pooled_ready(eff->images->slab))
			return;          /* from the pool: faulted in and pinned already */
		jobs[0].base = eff->images->slab;
		jobs[0].bytes = 17*eff->images->slab_stride + np*eff->n_energies*sizeof(double);
		nj = 1;
	}
	void *planes[] = { d.src_start_coords[0], d.src_start_coords[1], d.pc_start_coords[0], d.pc_start_coords[1],
		d.pc_start_dir[0], d.pc_start_dir[1], d.pc_start_elecv[0], d.pc_start_elecv[1],
		d.pc_exit_coords[0], d.pc_exit_coords[1], d.pc_exit_coords[2], d.pc_exit_dir[0], d.pc_exit_dir[1],
		d.pc_exit_elecv[0], d.pc_exit_elecv[1], d.pc_exit_nrefl, d.pc_exit_dtravel };
	for (size_t k = 0; k < sizeof(planes)/sizeof(planes[0]) && eff->images->slab == NULL; k++)
		if (planes[k] != NULL && !pc_plane_pooled_ready(planes[k])) { jobs[nj].base = planes[k]; jobs[nj].bytes = np*sizeof(double); nj++; }
	if (eff->images->slab == NULL && d.exit_coord_weights != NULL && !pc_plane_pooled_ready(d.exit_coord_weights)) { jobs[nj].base = (char *)d.exit_coord_weights; jobs[nj].bytes = np*eff->n_energies*sizeof(double); nj++; }
	if (nj == 0)
		return;          /* every plane comes from the pool: faulted in and pinned already */
	long cores = sysconf(_SC_NPROCESSORS_ONLN);
	int nt = (int)(cores < 1 ? 1 : (cores > 16 ? 16 : cores));
	pthread_t th[16];
	struct pc_touch_arg args[16];
	int started = 0;
	for (int t = 0; t < nt; t++) {
		args[t].jobs = jobs; args[t].njobs = nj; args[t].tid = t; args[t].nthreads = nt;
		if (t > 0 && pthread_create(&th[t], NULL, pc_touch_thread, &args[t]) != 0) {
			/* no thread: the caller touches this slice itself */
			pc_touch_thread(&args[t]);
			continue;
		}
		if (t > 0) started |= 1 << t;
	}
	pc_touch_thread(&args[0]);
	for (int t = 1; t < nt; t++)
		if (started & (1 << t)) pthread_join(th[t], NULL);
}

/* the image fetch may have pinned the (large) planes and left them pinned (option "keep_pinned"): remember in their header what
 * the runtime says about them now -- a fetch that fell back to pinning plane by plane, or whose registration was refused, leaves
 * the slab unpinned, and the pool and the free path must not treat it as registered */
void pc_transeff_planes_pinned(polycap_transmission_efficiencies *eff)
{
	if (eff == NULL || eff->images == NULL)
		return;
	if (eff->images->slab != NULL)
		pc_plane_set_pinned(eff->images->slab, pc_hip_host_is_pinned(eff->images->slab));
}

/* totals -> open area, counts and efficiencies (reference: src/polycap-source.c:1061-1076) */
void pc_transeff_finish(polycap_transmission_efficiencies *eff, const double *sum_weights, const int64_t counters[6])
{
	polycap_description *description = eff->source->description;
	const int64_t sum_iexit = counters[0], sum_not_entered = counters[1], sum_not_transmitted = counters[2];
	description->open_area = (double)(sum_iexit+sum_not_transmitted)/(sum_iexit+sum_not_entered+sum_not_transmitted);
	eff->images->i_start = sum_iexit+sum_not_entered+sum_not_transmitted;
	eff->images->i_exit = sum_iexit;
	for (size_t i = 0; i < eff->n_energies; i++)
		eff->efficiencies[i] = (sum_weights[i] / ((double)sum_iexit+(double)sum_not_transmitted)) * description->open_area;
}

void *pc_transmission_efficiencies_from_totals(void *source_, int64_t n_exit, const double *sum_weights, const int64_t counters[6],
	const pc_hip_images *planes, void *error_)
{
	polycap_source *source = source_;
	polycap_error **error = error_;
	if (source == NULL || source->description == NULL || source->energies == NULL || source->n_energies < 1) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "pc_transmission_efficiencies_from_totals: source must hold a description and energies");
		return NULL;
	}
	if (sum_weights == NULL || counters == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "pc_transmission_efficiencies_from_totals: sum_weights and counters cannot be NULL");
		return NULL;
	}
	if (n_exit < 0 || counters[0] != n_exit || counters[1] < 0 || counters[2] < 0 || counters[0] + counters[2] < 1) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "pc_transmission_efficiencies_from_totals: counters[0] must equal n_exit and at least one photon must have entered");
		return NULL;
	}
	polycap_transmission_efficiencies *eff = pc_transeff_alloc(source, (size_t)n_exit, 1, "pc_transmission_efficiencies_from_totals", error);
	if (eff == NULL)
		return NULL;
	if (planes != NULL && n_exit > 0) {
		pc_hip_images own;
		pc_transeff_plane_pointers(eff, &own);
		const size_t n = (size_t)n_exit, ne = source->n_energies;
#define PC_COPY(field, count) do { if (planes->field != NULL) memcpy(own.field, planes->field, sizeof(*own.field) * (count)); } while (0)
		for (int k = 0; k < 2; k++) {
			PC_COPY(src_start_coords[k], n); PC_COPY(pc_start_coords[k], n); PC_COPY(pc_start_dir[k], n);
			PC_COPY(pc_start_elecv[k], n); PC_COPY(pc_exit_dir[k], n); PC_COPY(pc_exit_elecv[k], n);
		}
		for (int k = 0; k < 3; k++)
			PC_COPY(pc_exit_coords[k], n);
		PC_COPY(pc_exit_nrefl, n); PC_COPY(pc_exit_dtravel, n); PC_COPY(exit_coord_weights, n * ne);
#undef PC_COPY
	}
	pc_transeff_finish(eff, sum_weights, counters);
	return eff;
}

bool polycap_transmission_efficiencies_get_data(polycap_transmission_efficiencies *efficiencies, size_t *n_energies,
	double **energies_arr, double **efficiencies_arr, polycap_error **error)
{
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_transmission_efficiencies_get_data: efficiencies cannot be NULL");
		return false;
	}
	if (n_energies)
		*n_energies = efficiencies->n_energies;
	if (energies_arr) {
		*energies_arr = malloc(sizeof(double) * efficiencies->n_energies);
		if (*energies_arr == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_transmission_efficiencies_get_data: could not allocate memory -> %s", strerror(errno));
			return false;
		}
		memcpy(*energies_arr, efficiencies->energies, sizeof(double) * efficiencies->n_energies);
	}
	if (efficiencies_arr) {
		*efficiencies_arr = malloc(sizeof(double) * efficiencies->n_energies);
		if (*efficiencies_arr == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_transmission_efficiencies_get_data: could not allocate memory -> %s", strerror(errno));
			return false;
		}
		memcpy(*efficiencies_arr, efficiencies->efficiencies, sizeof(double) * efficiencies->n_energies);
	}
	return true;
}

/* (x, y, sqrt(1 - x^2 - y^2)) from the two stored components, as the reference rebuilds unit vectors (:841-848) */
static polycap_vector3 pc_unit_from_xy(double x, double y)
{
	polycap_vector3 v = { x, y, sqrt(1. - x*x - y*y) };
	return v;
}

bool polycap_transmission_efficiencies_get_start_data(polycap_transmission_efficiencies *efficiencies, int64_t *n_start, int64_t *n_exit,
	polycap_vector3 **start_coords, polycap_vector3 **start_direction, polycap_vector3 **start_elecv, polycap_vector3 **src_start_coords,
	polycap_error **error)
{
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_start_data: efficiencies cannot be NULL");
		return false;
	}
	const struct _polycap_images *im = efficiencies->images;
	if (im == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_start_data: source->images cannot be NULL");
		return false;
	}
	*n_start = im->i_start;
	if (im->i_start == 0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_start_data: no photon start events in efficiencies");
		return false;
	}
	*n_exit = im->i_exit;
	if (im->i_exit == 0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_start_data: no photon exit events in efficiencies");
		return false;
	}
	const size_t n = (size_t)im->i_exit;
	*start_coords = malloc(sizeof(polycap_vector3) * n);
	*start_direction = malloc(sizeof(polycap_vector3) * n);
	*start_elecv = malloc(sizeof(polycap_vector3) * n);
	*src_start_coords = malloc(sizeof(polycap_vector3) * n);
	if (*start_coords == NULL || *start_direction == NULL || *start_elecv == NULL || *src_start_coords == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_get_start_data: could not allocate memory for start data -> %s", strerror(errno));
		return false;
	}
	for (size_t i = 0; i < n; i++) {
		(*start_coords)[i].x = im->pc_start_coords[0][i];
		(*start_coords)[i].y = im->pc_start_coords[1][i];
		(*start_coords)[i].z = 0.;
		(*start_direction)[i] = pc_unit_from_xy(im->pc_start_dir[0][i], im->pc_start_dir[1][i]);
		(*start_elecv)[i] = pc_unit_from_xy(im->pc_start_elecv[0][i], im->pc_start_elecv[1][i]);
		(*src_start_coords)[i].x = im->src_start_coords[0][i];
		(*src_start_coords)[i].y = im->src_start_coords[1][i];
		(*src_start_coords)[i].z = 0.;
	}
	return true;
}

bool polycap_transmission_efficiencies_get_exit_data(polycap_transmission_efficiencies *efficiencies, int64_t *n_exit,
	polycap_vector3 **exit_coords, polycap_vector3 **exit_direction, polycap_vector3 **exit_elecv, int64_t **n_refl, double **d_travel,
	size_t *n_energies, double ***exit_weights, polycap_error **error)
{
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_exit_data: efficiencies cannot be NULL");
		return false;
	}
	const struct _polycap_images *im = efficiencies->images;
	if (im == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_exit_data: source->images cannot be NULL");
		return false;
	}
	*n_exit = im->i_exit;
	*n_energies = efficiencies->n_energies;
	if (im->i_start == 0) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_exit_data: no photon start events in efficiencies");
		return false;
	}
	if (im->i_exit == 0) {
		/* only a histogram-only result (POLYCAP_IMAGES=0) has started photons and no exit planes */
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_exit_data: no photon exit events in efficiencies (histogram-only result)");
		return false;
	}
	const size_t n = (size_t)im->i_exit, ne = efficiencies->n_energies;
	*exit_coords = malloc(sizeof(polycap_vector3) * n);
	*exit_direction = malloc(sizeof(polycap_vector3) * n);
	*exit_elecv = malloc(sizeof(polycap_vector3) * n);
	*n_refl = malloc(sizeof(int64_t) * n);
	*d_travel = malloc(sizeof(double) * n);
	*exit_weights = calloc(n ? n : 1, sizeof(double *));
	if (*exit_coords == NULL || *exit_direction == NULL || *exit_elecv == NULL || *n_refl == NULL || *d_travel == NULL || *exit_weights == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_get_exit_data: could not allocate memory for exit data -> %s", strerror(errno));
		return false;
	}
	for (size_t i = 0; i < n; i++) {
		(*n_refl)[i] = im->pc_exit_nrefl[i];
		(*d_travel)[i] = im->pc_exit_dtravel[i];
		(*exit_coords)[i].x = im->pc_exit_coords[0][i];
		(*exit_coords)[i].y = im->pc_exit_coords[1][i];
		(*exit_coords)[i].z = im->pc_exit_coords[2][i];
		(*exit_direction)[i] = pc_unit_from_xy(im->pc_exit_dir[0][i], im->pc_exit_dir[1][i]);
		(*exit_elecv)[i] = pc_unit_from_xy(im->pc_exit_elecv[0][i], im->pc_exit_elecv[1][i]);
		(*exit_weights)[i] = malloc(sizeof(double) * ne);
		if ((*exit_weights)[i] == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_source_get_exit_data: could not allocate memory for (*exit_weights)[i] -> %s", strerror(errno));
			return false;
		}
		memcpy((*exit_weights)[i], im->exit_coord_weights + i*ne, sizeof(double) * ne);
	}
	return true;
}

/* Leak events of the finished leak_calc run of `ctx` -> the leak planes of eff->images (reference: src/polycap-source.c:
 * 982-1032).  Fetched in blocks of records (include/polycap-hip.h) and scattered into the planes. */
int pc_transeff_fetch_leaks(polycap_transmission_efficiencies *eff, pc_hip_ctx *ctx, pc_hip_group *group)
{
	struct _polycap_images *im = eff->images;
	const size_t ne = eff->n_energies, stride = PC_HIP_LEAK_HDR + ne;
	int64_t n_kind[2] = {0, 0};
	int status = (group != NULL) ? pc_hip_group_leak_counts(group, &n_kind[0], &n_kind[1]) : pc_hip_leak_counts(ctx, &n_kind[0], &n_kind[1]);
	if (status != PC_HIP_OK)
		return status;
	const int64_t block = 65536;
	double *rec = malloc(sizeof(double) * stride * (size_t)block);
	if (rec == NULL)
		return PC_HIP_ERR_MEMORY;
	for (int kind = 0; kind < 2 && status == PC_HIP_OK; kind++) {
		const size_t n = (size_t)n_kind[kind], nalloc = n ? n : 1;
		double *coords[3], *dir[2], *elecv[2] = { NULL, NULL }, *w;
		int64_t *nrefl;
		int ok = 1;
		for (int k = 0; k < 3; k++) { coords[k] = malloc(sizeof(double) * nalloc); ok = ok && coords[k] != NULL; }
		for (int k = 0; k < 2; k++) { dir[k] = malloc(sizeof(double) * nalloc); ok = ok && dir[k] != NULL; }
		if (kind == 1)
			for (int k = 0; k < 2; k++) { elecv[k] = malloc(sizeof(double) * nalloc); ok = ok && elecv[k] != NULL; }
		nrefl = malloc(sizeof(int64_t) * nalloc);
		w = malloc(sizeof(double) * nalloc * ne);
		ok = ok && nrefl != NULL && w != NULL;
		/* hand the planes to the images first so that a failure below is cleaned up by polycap_transmission_efficiencies_free */
		if (kind == 0) {
			for (int k = 0; k < 3; k++) im->extleak_coords[k] = coords[k];
			for (int k = 0; k < 2; k++) im->extleak_dir[k] = dir[k];
			im->extleak_n_refl = nrefl; im->extleak_coord_weights = w; im->i_extleak = (int64_t)n;
		} else {
			for (int k = 0; k < 3; k++) im->intleak_coords[k] = coords[k];
			for (int k = 0; k < 2; k++) { im->intleak_dir[k] = dir[k]; im->intleak_elecv[k] = elecv[k]; }
			im->intleak_n_refl = nrefl; im->intleak_coord_weights = w; im->i_intleak = (int64_t)n;
		}
		if (!ok) { status = PC_HIP_ERR_MEMORY; break; }
		for (int64_t first = 0; first < (int64_t)n && status == PC_HIP_OK; first += block) {
			const int64_t count = ((int64_t)n - first < block) ? (int64_t)n - first : block;
			status = (group != NULL) ? pc_hip_group_leak_events(group, kind, first, count, rec) : pc_hip_leak_events(ctx, kind, first, count, rec);
			if (status != PC_HIP_OK)
				break;
			for (int64_t k = 0; k < count; k++) {
				const double *r = rec + (size_t)k * stride;
				const size_t j = (size_t)(first + k);
				coords[0][j] = r[2]; coords[1][j] = r[3]; coords[2][j] = r[4];
				dir[0][j] = r[5]; dir[1][j] = r[6];
				if (kind == 1) { elecv[0][j] = r[8]; elecv[1][j] = r[9]; }
				nrefl[j] = (int64_t)r[11];
				memcpy(w + j*ne, r + PC_HIP_LEAK_HDR, sizeof(double) * ne);
			}
		}
	}
	free(rec);
	return status;
}

/* one malloc'd polycap_leak per stored event (reference :929-1058): direction.z and elecv.z are rebuilt from the two
 * stored components; extleak events carry no electric vector in the images (zeros here) */
static bool pc_transeff_leaks(polycap_transmission_efficiencies *efficiencies, int kind, polycap_leak ***leaks, int64_t *n_leaks,
	const char *caller, const char *what, polycap_error **error)
{
	const struct _polycap_images *im = efficiencies->images;
	const int64_t n = (kind == 0) ? im->i_extleak : im->i_intleak;
	const size_t ne = efficiencies->n_energies;
	*n_leaks = n;
	if (n == 0) {
		*leaks = NULL;
		polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "%s: no %s events in efficiencies", caller, what);
		return false;
	}
	*leaks = malloc(sizeof(polycap_leak *) * (size_t)n);
	if (*leaks == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for leaks -> %s", caller, strerror(errno));
		return false;
	}
	for (int64_t i = 0; i < n; i++) {
		polycap_leak *l = calloc(1, sizeof(polycap_leak));
		(*leaks)[i] = l;
		if (l != NULL)
			l->weight = malloc(sizeof(double) * ne);
		if (l == NULL || l->weight == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "%s: could not allocate memory for (*leaks)[i] -> %s", caller, strerror(errno));
			return false;
		}
		if (kind == 0) {
			l->coords.x = im->extleak_coords[0][i]; l->coords.y = im->extleak_coords[1][i]; l->coords.z = im->extleak_coords[2][i];
			l->direction = pc_unit_from_xy(im->extleak_dir[0][i], im->extleak_dir[1][i]);
			l->n_refl = im->extleak_n_refl[i];
			memcpy(l->weight, im->extleak_coord_weights + (size_t)i*ne, sizeof(double) * ne);
		} else {
			l->coords.x = im->intleak_coords[0][i]; l->coords.y = im->intleak_coords[1][i]; l->coords.z = im->intleak_coords[2][i];
			l->direction = pc_unit_from_xy(im->intleak_dir[0][i], im->intleak_dir[1][i]);
			l->elecv = pc_unit_from_xy(im->intleak_elecv[0][i], im->intleak_elecv[1][i]);
			l->n_refl = im->intleak_n_refl[i];
			memcpy(l->weight, im->intleak_coord_weights + (size_t)i*ne, sizeof(double) * ne);
		}
		l->n_energies = ne;
	}
	return true;
}

bool polycap_transmission_efficiencies_get_extleak_data(polycap_transmission_efficiencies *efficiencies, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
{
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_extleak_data: efficiencies cannot be NULL");
		return false;
	}
	if (efficiencies->images == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_extleak_data: source->images cannot be NULL");
		return false;
	}
	return pc_transeff_leaks(efficiencies, 0, leaks, n_leaks, "polycap_source_get_extleak_data", "extleak", error);
}

bool polycap_transmission_efficiencies_get_intleak_data(polycap_transmission_efficiencies *efficiencies, polycap_leak ***leaks, int64_t *n_leaks, polycap_error **error)
{
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_intleak_data: efficiencies cannot be NULL");
		return false;
	}
	if (efficiencies->images == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_source_get_intleak_data: source->images cannot be NULL");
		return false;
	}
	return pc_transeff_leaks(efficiencies, 1, leaks, n_leaks, "polycap_source_get_intleak_data", "intleak", error);
}
