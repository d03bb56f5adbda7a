/*
 * pc_hdf5.c -- polycap_transmission_efficiencies_write_hdf5: the result file of a transmission run.
 *
 * File layout = what the reference's writer produces (src/polycap-transmission-efficiencies.c:229-780,
 * leak_calc=false): every dataset is native fp64 with a fixed-length string attribute "Units";
 *
 *   /Energies [nE] keV                      /Transmission_Efficiencies [nE] a.u.
 *   /PC_Start/Coordinates|Direction|Electric_Vector [2, i_exit] "[cm,cm]"
 *   /PC_Exit/Coordinates [3, i_exit] "[cm,cm,cm]"   /PC_Exit/Direction|Electric_Vector [2, i_exit] "[cm,cm]"
 *   /PC_Exit/N_Reflections [i_exit] a.u.    /PC_Exit/D_Travel [i_exit] "[cm]"
 *   /PC_Exit/Weights [i_exit, nE] "[keV,a.u.]"
 *   /Source_Start_Coordinates [2, i_exit] "[cm,cm]"
 *   /Input/PC_Shape [2, nmax] (z, ext)   /Input/Cap_Shape [2, nmax] (z, cap)   "[cm,cm]"
 *   /Input/N_Capillaries a.u. | Surface_Roughness Angstrom | Open_Area a.u. | PC_Density g/cm3 | Src_PC_Dist cm   [1]
 *   /Input/PC_Composition [2, nelem] "[Z,w%]"
 *
 *   after a leak_calc run with events of the kind (:518-700), for <G> in ExternalLeaks, InternalLeaks:
 *   /<G>/Coordinates [3, n] "[cm,cm,cm]"   /<G>/Direction [2, n] "[cm,cm]"   /<G>/Weights [n, nE] "[keV,a.u.]"
 *   /<G>/Weight_Total [nE] a.u. (sum of the event weights / started photons)   /<G>/N_Reflections [n] a.u.
 *   /InternalLeaks/Electric_Vector [2, n] "[cm,cm]"
 *
 * libhdf5 is bound at run time (dlopen), like xraylib in pc_optconst.c, so libpolycap.so carries no link-time
 * dependency on it: hosts without HDF5 get POLYCAP_ERROR_UNSUPPORTED from this one function and nothing else changes.
 * POLYCAP_HDF5_LIB names the library explicitly.
 */
#include "pc_private.h"

#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* the handful of HDF5 1.10+ ABI types and constants used here (H5public.h, H5Ipublic.h, H5Fpublic.h, H5Spublic.h) */
typedef int64_t pc_hid;
typedef int pc_herr;
typedef unsigned long long pc_hsize;
#define PC_H5P_DEFAULT ((pc_hid)0)
#define PC_H5S_ALL ((pc_hid)0)
#define PC_H5F_ACC_TRUNC 0x0002u
#define PC_H5S_SCALAR 0
#define PC_H5E_DEFAULT ((pc_hid)0)

static struct {
	void *handle;
	int tried;
	pc_herr (*open)(void);
	pc_herr (*get_libversion)(unsigned *, unsigned *, unsigned *);
	pc_herr (*eset_auto2)(pc_hid, void *, void *);
	pc_hid (*fcreate)(const char *, unsigned, pc_hid, pc_hid);
	pc_herr (*fclose)(pc_hid);
	pc_hid (*gcreate2)(pc_hid, const char *, pc_hid, pc_hid, pc_hid);
	pc_herr (*gclose)(pc_hid);
	pc_hid (*screate_simple)(int, const pc_hsize *, const pc_hsize *);
	pc_hid (*screate)(int);
	pc_herr (*sclose)(pc_hid);
	pc_hid (*dcreate2)(pc_hid, const char *, pc_hid, pc_hid, pc_hid, pc_hid, pc_hid);
	pc_herr (*dwrite)(pc_hid, pc_hid, pc_hid, pc_hid, pc_hid, const void *);
	pc_herr (*dclose)(pc_hid);
	pc_hid (*tcopy)(pc_hid);
	pc_herr (*tset_size)(pc_hid, size_t);
	pc_herr (*tclose)(pc_hid);
	pc_hid (*acreate2)(pc_hid, const char *, pc_hid, pc_hid, pc_hid, pc_hid);
	pc_herr (*awrite)(pc_hid, pc_hid, const void *);
	pc_herr (*aclose)(pc_hid);
	pc_hid *native_double;   /* H5T_NATIVE_DOUBLE_g, valid after H5open() */
	pc_hid *c_s1;            /* H5T_C_S1_g */
	char name[256];
} h5;
static pthread_mutex_t h5_mutex = PTHREAD_MUTEX_INITIALIZER;

static int pc_h5_bind(void *handle)
{
#define PC_SYM(field, sym) do { *(void **)(&h5.field) = dlsym(handle, sym); if (h5.field == NULL) return 0; } while (0)
	PC_SYM(open, "H5open");
	PC_SYM(get_libversion, "H5get_libversion");
	PC_SYM(eset_auto2, "H5Eset_auto2");
	PC_SYM(fcreate, "H5Fcreate");
	PC_SYM(fclose, "H5Fclose");
	PC_SYM(gcreate2, "H5Gcreate2");
	PC_SYM(gclose, "H5Gclose");
	PC_SYM(screate_simple, "H5Screate_simple");
	PC_SYM(screate, "H5Screate");
	PC_SYM(sclose, "H5Sclose");
	PC_SYM(dcreate2, "H5Dcreate2");
	PC_SYM(dwrite, "H5Dwrite");
	PC_SYM(dclose, "H5Dclose");
	PC_SYM(tcopy, "H5Tcopy");
	PC_SYM(tset_size, "H5Tset_size");
	PC_SYM(tclose, "H5Tclose");
	PC_SYM(acreate2, "H5Acreate2");
	PC_SYM(awrite, "H5Awrite");
	PC_SYM(aclose, "H5Aclose");
	PC_SYM(native_double, "H5T_NATIVE_DOUBLE_g");
	PC_SYM(c_s1, "H5T_C_S1_g");
#undef PC_SYM
	unsigned maj = 0, min = 0, rel = 0;
	/* hid_t is 64-bit from HDF5 1.10 on; older libraries have a different ABI */
	if (h5.get_libversion(&maj, &min, &rel) < 0 || maj != 1 || min < 10)
		return 0;
	if (h5.open() < 0)
		return 0;
	h5.eset_auto2(PC_H5E_DEFAULT, NULL, NULL);   /* errors are reported through polycap_error, not on stderr */
	return 1;
}

static int pc_h5_load(void)
{
	pthread_mutex_lock(&h5_mutex);
	if (!h5.tried) {
		h5.tried = 1;
		const char *env = getenv("POLYCAP_HDF5_LIB");
		const char *candidates[] = { env, "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.310", "libhdf5.so.200", "libhdf5.so.103",
			"libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so" };
		for (size_t k = 0; k < sizeof(candidates)/sizeof(candidates[0]) && h5.handle == NULL; k++) {
			if (candidates[k] == NULL || candidates[k][0] == '\0')
				continue;
			void *handle = dlopen(candidates[k], RTLD_NOW | RTLD_LOCAL);
			if (handle == NULL)
				continue;
			if (pc_h5_bind(handle)) {
				h5.handle = handle;
				strncpy(h5.name, candidates[k], sizeof(h5.name) - 1);
			} else {
				dlclose(handle);
			}
		}
	}
	int ok = h5.handle != NULL;
	pthread_mutex_unlock(&h5_mutex);
	return ok;
}

const char *pc_hdf5_provider(void)
{
	return pc_h5_load() ? h5.name : "none";
}

/* one fp64 dataset + its "Units" attribute (reference :229-318) */
static bool pc_h5_dataset(pc_hid file, int rank, const pc_hsize *dim, const char *name, const double *data, const char *units,
	polycap_error **error)
{
	pc_hid space = -1, dset = -1, aspace = -1, atype = -1, attr = -1;
	bool ok = false;
	/* HDF5 has no zero-sized simple extents in this usage; an empty run writes one zero instead of failing */
	static const double zero = 0.;
	pc_hsize d[2] = { dim[0], rank > 1 ? dim[1] : 1 };
	if (d[0] * d[1] == 0) {
		d[0] = d[0] ? d[0] : 1;
		d[1] = d[1] ? d[1] : 1;
		if (d[0] * d[1] != 1) {
			polycap_set_error(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_transmission_efficiencies_write_hdf5: dataset %s is empty", name);
			return false;
		}
		data = &zero;
	}
	if ((space = h5.screate_simple(rank, d, NULL)) < 0) goto fail;
	if ((dset = h5.dcreate2(file, name, *h5.native_double, space, PC_H5P_DEFAULT, PC_H5P_DEFAULT, PC_H5P_DEFAULT)) < 0) goto fail;
	if (h5.dwrite(dset, *h5.native_double, PC_H5S_ALL, PC_H5S_ALL, PC_H5P_DEFAULT, data) < 0) goto fail;
	if ((aspace = h5.screate(PC_H5S_SCALAR)) < 0) goto fail;
	if ((atype = h5.tcopy(*h5.c_s1)) < 0) goto fail;
	if (h5.tset_size(atype, strlen(units)) < 0) goto fail;
	if ((attr = h5.acreate2(dset, "Units", atype, aspace, PC_H5P_DEFAULT, PC_H5P_DEFAULT)) < 0) goto fail;
	if (h5.awrite(attr, atype, units) < 0) goto fail;
	ok = true;
fail:
	if (attr >= 0 && h5.aclose(attr) < 0) ok = false;
	if (atype >= 0 && h5.tclose(atype) < 0) ok = false;
	if (aspace >= 0 && h5.sclose(aspace) < 0) ok = false;
	if (dset >= 0 && h5.dclose(dset) < 0) ok = false;
	if (space >= 0 && h5.sclose(space) < 0) ok = false;
	if (!ok)
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_transmission_efficiencies_write_hdf5: could not write dataset %s", name);
	return ok;
}

/* planes[k][0..n) stacked into one [nplanes, n] dataset */
static bool pc_h5_planes(pc_hid file, const char *name, int nplanes, double *const *planes, size_t n, const char *units, double *tmp,
	polycap_error **error)
{
	for (int k = 0; k < nplanes; k++)
		memcpy(tmp + (size_t)k * n, planes[k], sizeof(double) * n);
	pc_hsize dim[2] = { (pc_hsize)nplanes, (pc_hsize)n };
	return pc_h5_dataset(file, 2, dim, name, tmp, units, error);
}

static bool pc_h5_scalar(pc_hid file, const char *name, double value, const char *units, polycap_error **error)
{
	pc_hsize one = 1;
	return pc_h5_dataset(file, 1, &one, name, &value, units, error);
}

static bool pc_h5_group(pc_hid file, const char *name, polycap_error **error)
{
	pc_hid g = h5.gcreate2(file, name, PC_H5P_DEFAULT, PC_H5P_DEFAULT, PC_H5P_DEFAULT);
	if (g < 0 || h5.gclose(g) < 0) {
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_transmission_efficiencies_write_hdf5: could not create group %s", name);
		return false;
	}
	return true;
}

bool polycap_transmission_efficiencies_write_hdf5(polycap_transmission_efficiencies *efficiencies, const char *filename, polycap_error **error)
{
	if (filename == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_transmission_efficiencies_write_hdf5: filename cannot be NULL");
		return false;
	}
	if (efficiencies == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_transmission_efficiencies_write_hdf5: efficiencies cannot be NULL");
		return false;
	}
	const struct _polycap_images *im = efficiencies->images;
	const polycap_source *src = efficiencies->source;
	if (im == NULL || src == NULL || src->description == NULL || src->description->profile == NULL) {
		polycap_set_error_literal(error, POLYCAP_ERROR_INVALID_ARGUMENT, "polycap_transmission_efficiencies_write_hdf5: efficiencies hold no images or source");
		return false;
	}
	if (!pc_h5_load()) {
		polycap_set_error_literal(error, POLYCAP_ERROR_UNSUPPORTED, "polycap_transmission_efficiencies_write_hdf5: no HDF5 (>= 1.10) shared library found; set POLYCAP_HDF5_LIB or use the getters");
		return false;
	}
	const polycap_description *desc = src->description;
	const struct _polycap_profile *prof = desc->profile;
	const size_t n = (size_t)im->i_exit, ne = efficiencies->n_energies;
	const size_t nprof = (size_t)prof->nmax;          /* the reference writes nmax (not nmax+1) profile points (:709-737) */
	size_t tmp_len = 3 * (n ? n : 1);
	if (2 * nprof > tmp_len) tmp_len = 2 * nprof;
	if (2 * (size_t)desc->nelem > tmp_len) tmp_len = 2 * (size_t)desc->nelem;
	double *tmp = malloc(sizeof(double) * tmp_len);
	if (tmp == NULL) {
		polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_transmission_efficiencies_write_hdf5: could not allocate memory -> %s", strerror(errno));
		return false;
	}

	pthread_mutex_lock(&h5_mutex);     /* the writer makes no assumption about a thread-safe HDF5 build */
	bool ok = false;
	pc_hid file = h5.fcreate(filename, PC_H5F_ACC_TRUNC, PC_H5P_DEFAULT, PC_H5P_DEFAULT);
	if (file < 0) {
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_transmission_efficiencies_write_hdf5: unable to create file %s", filename);
		goto done;
	}
	pc_hsize dim[2];
	dim[0] = ne;
	if (!pc_h5_dataset(file, 1, dim, "/Energies", efficiencies->energies, "keV", error)) goto close;
	if (!pc_h5_dataset(file, 1, dim, "/Transmission_Efficiencies", efficiencies->efficiencies, "a.u.", error)) goto close;

	if (!pc_h5_group(file, "/PC_Start", error)) goto close;
	if (!pc_h5_planes(file, "/PC_Start/Coordinates", 2, im->pc_start_coords, n, "[cm,cm]", tmp, error)) goto close;
	if (!pc_h5_planes(file, "/PC_Start/Direction", 2, im->pc_start_dir, n, "[cm,cm]", tmp, error)) goto close;
	if (!pc_h5_planes(file, "/PC_Start/Electric_Vector", 2, im->pc_start_elecv, n, "[cm,cm]", tmp, error)) goto close;

	if (!pc_h5_group(file, "/PC_Exit", error)) goto close;
	if (!pc_h5_planes(file, "/PC_Exit/Coordinates", 3, im->pc_exit_coords, n, "[cm,cm,cm]", tmp, error)) goto close;
	for (size_t j = 0; j < n; j++)
		tmp[j] = (double)im->pc_exit_nrefl[j];
	dim[0] = n;
	if (!pc_h5_dataset(file, 1, dim, "/PC_Exit/N_Reflections", tmp, "a.u.", error)) goto close;
	if (!pc_h5_planes(file, "/PC_Exit/Direction", 2, im->pc_exit_dir, n, "[cm,cm]", tmp, error)) goto close;
	if (!pc_h5_planes(file, "/PC_Exit/Electric_Vector", 2, im->pc_exit_elecv, n, "[cm,cm]", tmp, error)) goto close;
	dim[0] = n; dim[1] = ne;
	if (!pc_h5_dataset(file, 2, dim, "/PC_Exit/Weights", im->exit_coord_weights, "[keV,a.u.]", error)) goto close;
	dim[0] = n;
	if (!pc_h5_dataset(file, 1, dim, "/PC_Exit/D_Travel", im->pc_exit_dtravel, "[cm]", error)) goto close;

	if (!pc_h5_planes(file, "/Source_Start_Coordinates", 2, im->src_start_coords, n, "[cm,cm]", tmp, error)) goto close;

	for (int kind = 0; kind < 2; kind++) {
		const size_t nl = (size_t)(kind == 0 ? im->i_extleak : im->i_intleak);
		if (nl == 0)
			continue;
		double *const *coords = kind == 0 ? im->extleak_coords : im->intleak_coords;
		double *const *dirs = kind == 0 ? im->extleak_dir : im->intleak_dir;
		const int64_t *nrefl = kind == 0 ? im->extleak_n_refl : im->intleak_n_refl;
		const double *lw = kind == 0 ? im->extleak_coord_weights : im->intleak_coord_weights;
		const char *g = kind == 0 ? "/ExternalLeaks" : "/InternalLeaks";
		char name[64];
		size_t need = 3*nl > ne ? 3*nl : ne;
		double *ltmp = malloc(sizeof(double) * need);
		if (ltmp == NULL) {
			polycap_set_error(error, POLYCAP_ERROR_MEMORY, "polycap_transmission_efficiencies_write_hdf5: could not allocate memory -> %s", strerror(errno));
			goto close;
		}
		bool lok = pc_h5_group(file, g, error);
		snprintf(name, sizeof name, "%s/Coordinates", g);
		lok = lok && pc_h5_planes(file, name, 3, coords, nl, "[cm,cm,cm]", ltmp, error);
		snprintf(name, sizeof name, "%s/Direction", g);
		lok = lok && pc_h5_planes(file, name, 2, dirs, nl, "[cm,cm]", ltmp, error);
		if (kind == 1)
			lok = lok && pc_h5_planes(file, "/InternalLeaks/Electric_Vector", 2, im->intleak_elecv, nl, "[cm,cm]", ltmp, error);
		dim[0] = nl; dim[1] = ne;
		snprintf(name, sizeof name, "%s/Weights", g);
		lok = lok && pc_h5_dataset(file, 2, dim, name, lw, "[keV,a.u.]", error);
		for (size_t j = 0; j < ne; j++) {
			ltmp[j] = 0.0;
			for (size_t k = 0; k < nl; k++)
				ltmp[j] += lw[k*ne + j];
			ltmp[j] = ltmp[j] / (double)im->i_start;
		}
		dim[0] = ne;
		snprintf(name, sizeof name, "%s/Weight_Total", g);
		lok = lok && pc_h5_dataset(file, 1, dim, name, ltmp, "a.u.", error);
		for (size_t j = 0; j < nl; j++)
			ltmp[j] = (double)nrefl[j];
		dim[0] = nl;
		snprintf(name, sizeof name, "%s/N_Reflections", g);
		lok = lok && pc_h5_dataset(file, 1, dim, name, ltmp, "a.u.", error);
		free(ltmp);
		if (!lok) goto close;
	}

	if (!pc_h5_group(file, "/Input", error)) goto close;
	{
		double *const shape_ext[2] = { prof->z, prof->ext }, *const shape_cap[2] = { prof->z, prof->cap };
		if (!pc_h5_planes(file, "/Input/PC_Shape", 2, shape_ext, nprof, "[cm,cm]", tmp, error)) goto close;
		if (!pc_h5_planes(file, "/Input/Cap_Shape", 2, shape_cap, nprof, "[cm,cm]", tmp, error)) goto close;
	}
	if (!pc_h5_scalar(file, "/Input/N_Capillaries", (double)desc->n_cap, "a.u.", error)) goto close;
	if (!pc_h5_scalar(file, "/Input/Surface_Roughness", desc->sig_rough, "Angstrom", error)) goto close;
	if (!pc_h5_scalar(file, "/Input/Open_Area", desc->open_area, "a.u.", error)) goto close;
	for (unsigned int j = 0; j < desc->nelem; j++) {
		tmp[j] = (double)desc->iz[j];
		tmp[j + desc->nelem] = desc->wi[j];
	}
	dim[0] = 2; dim[1] = desc->nelem;
	if (!pc_h5_dataset(file, 2, dim, "/Input/PC_Composition", tmp, "[Z,w%]", error)) goto close;
	if (!pc_h5_scalar(file, "/Input/PC_Density", desc->density, "g/cm3", error)) goto close;
	if (!pc_h5_scalar(file, "/Input/Src_PC_Dist", src->d_source, "cm", error)) goto close;
	ok = true;
close:
	if (h5.fclose(file) < 0 && ok) {
		polycap_set_error(error, POLYCAP_ERROR_IO, "polycap_transmission_efficiencies_write_hdf5: could not close file %s", filename);
		ok = false;
	}
done:
	pthread_mutex_unlock(&h5_mutex);
	free(tmp);
	return ok;
}
