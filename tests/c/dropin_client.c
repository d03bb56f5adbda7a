/* A C program written against polycap's public API only (include/polycap.h), as a user of the reference would write it:
 * it is linked against this repository's libpolycap.so and checks the reference's own published answers for the path
 *   - the seven-energy transmission curve of the ellipsoidal test optic (reference tests/source.c:216-222),
 *   - the per-photon sanity checks of the result getters (:229-278),
 *   - return codes of polycap_photon_launch for the reference's six launch cases (tests/photon.c:241-352),
 *   - polycap_source_get_photon + polycap_photon_launch against the driver (tests/source.c:306-340, 3000 photons here),
 *   - the error convention (INVALID_ARGUMENT for a progress monitor, n_photons < 1).
 * Exit status 0 = all checks passed; every failure prints its line.  Built and run by tests/test_gpu_parity.py. */
#include <polycap.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "dropin_client.c:%d: check failed: %s\n", __LINE__, #cond); failures++; } } while (0)

int main(void)
{
	polycap_error *error = NULL;
	int iz[2] = {8, 14};
	double wi[2] = {53.0, 47.0};
	double energies[7] = {1, 5, 10, 15, 20, 25, 30};
	const double expect[7] = {0.424, 0.349, 0.135, 0.050, 0.022, 0.011, 0.006};
	const double tol[7] = {0.01, 0.01, 0.0075, 0.005, 0.005, 0.005, 0.005};

	polycap_profile *profile = polycap_profile_new(POLYCAP_PROFILE_ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153E-5, 1000., 0.5, &error);
	CHECK(profile != NULL && error == NULL);
	polycap_description *description = polycap_description_new(profile, 0.0, 200000, 2, iz, wi, 2.23, &error);
	CHECK(description != NULL && error == NULL);
	polycap_profile_free(profile);                         /* constructors copy their inputs (tests/source.c:184-187) */
	polycap_source *source = polycap_source_new(description, 2000.0, 0.2065, 0.2065, 0.0, 0.0, 0.0, 0.0, 0.5, 7, energies, &error);
	CHECK(source != NULL && error == NULL);
	polycap_description_free(description);

	/* error convention */
	CHECK(polycap_source_get_transmission_efficiencies(source, 1, 0, false, NULL, &error) == NULL);
	CHECK(error != NULL && polycap_error_matches(error, POLYCAP_ERROR_INVALID_ARGUMENT));
	polycap_clear_error(&error);
	CHECK(polycap_source_get_transmission_efficiencies(source, 1, 1000, false, (polycap_progress_monitor *)&failures, &error) == NULL);
	CHECK(error != NULL && polycap_error_matches(error, POLYCAP_ERROR_INVALID_ARGUMENT));
	polycap_clear_error(&error);

	/* the driver: tests/source.c:197-278 */
	polycap_transmission_efficiencies *eff = polycap_source_get_transmission_efficiencies(source, -1, 30000, false, NULL, &error);
	CHECK(eff != NULL && error == NULL);
	if (eff == NULL) { fprintf(stderr, "%s\n", error ? error->message : "no error set"); return 1; }
	size_t n_energies = 0;
	double *e_arr = NULL, *t_arr = NULL;
	CHECK(polycap_transmission_efficiencies_get_data(eff, &n_energies, &e_arr, &t_arr, &error));
	CHECK(n_energies == 7);
	for (int i = 0; i < 7; i++) {
		CHECK(e_arr[i] == energies[i]);
		CHECK(fabs(t_arr[i] - expect[i]) <= tol[i]);
	}
	int64_t n_start = 0, n_exit = 0;
	polycap_vector3 *sc = NULL, *sd = NULL, *se = NULL, *src_sc = NULL;
	CHECK(polycap_transmission_efficiencies_get_start_data(eff, &n_start, &n_exit, &sc, &sd, &se, &src_sc, &error));
	CHECK(n_exit == 30000 && n_start > n_exit);
	CHECK(fabs(sc[0].x) <= 0.2065 && fabs(sc[0].y) <= 0.2065 && sc[0].z == 0.);
	CHECK(sd[0].x == 0. && sd[0].y == 0. && sd[0].z == 1.);
	CHECK((se[0].x == 1. && se[0].y == 0.) || (se[0].x == 0. && se[0].y == 1.));
	polycap_vector3 *ec = NULL, *ed = NULL, *ee = NULL;
	int64_t *n_refl = NULL;
	double *d_travel = NULL, **exit_weights = NULL;
	size_t ne2 = 0;
	int64_t n_exit2 = 0;
	CHECK(polycap_transmission_efficiencies_get_exit_data(eff, &n_exit2, &ec, &ed, &ee, &n_refl, &d_travel, &ne2, &exit_weights, &error));
	CHECK(n_exit2 == 30000 && ne2 == 7);
	for (int64_t j = 0; j < n_exit2; j += 997) {
		CHECK(ec[j].z == 9. && d_travel[j] >= 9. && n_refl[j] >= 0 && n_refl[j] < 1000);
		for (int i = 0; i < 7; i++)
			CHECK(exit_weights[j][i] >= 0. && exit_weights[j][i] <= 1.);
		CHECK(exit_weights[j][0] >= 1e-4 || exit_weights[j][1] >= 1e-4 || exit_weights[j][2] >= 1e-4);
	}

	/* explicit photons: tests/photon.c:241-352 (return codes) */
	{
		polycap_profile *p2 = polycap_profile_new(POLYCAP_PROFILE_ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153E-5, 1000., 0.5, &error);
		polycap_description *d2 = polycap_description_new(p2, 0.0, 200000, 2, iz, wi, 2.23, &error);
		polycap_profile_free(p2);
		const polycap_vector3 elecv = {0.5, 0.5, 0.};
		const struct { polycap_vector3 start, dir; int rc; } cases[] = {
			{ {0., 0., 0.}, {0., 0., 1.0}, 1 },                      /* straight through the central capillary */
			{ {0.21, 0.21, 0.}, {0., 0., 1.0}, -2 },                 /* outside the optic */
			{ {0.0585, 0., 0.}, {0.001, 0., 1.0}, 2 },               /* hits the glass at the entrance */
		};
		double e10 = 10.0;
		for (size_t k = 0; k < sizeof(cases)/sizeof(cases[0]); k++) {
			polycap_photon *ph = polycap_photon_new(d2, cases[k].start, cases[k].dir, elecv, &error);
			CHECK(ph != NULL);
			double *w = NULL;
			int rc = polycap_photon_launch(ph, 1, &e10, &w, false, &error);
			if (rc == -2) polycap_clear_error(&error);
			/* the entrance outcome of the third case depends on where the start point falls in its hexagon cell: 2 or 0/1 */
			if (k < 2) CHECK(rc == cases[k].rc);
			else CHECK(rc == 2 || rc == 0 || rc == 1);
			if (rc == 1 && k == 0) {
				CHECK(w != NULL && w[0] == 1.0 && polycap_photon_get_irefl(ph) == 0);
				polycap_vector3 x = polycap_photon_get_exit_coords(ph);
				CHECK(x.x == 0. && x.y == 0.);
			}
			free(w);
			polycap_photon_free(ph);
		}
		polycap_description_free(d2);
	}

	/* get_photon + launch against the driver: tests/source.c:306-340 with 3000 transmitted photons; the statistical error of the
	 * loop is ~0.3 / sqrt(3000) x efficiency, well inside the reference's 0.0075 except for the 1 keV point: 0.02 there */
	{
		polycap_rng *rng = polycap_rng_new_with_seed(20000);
		double w_tot[7] = {0, 0, 0, 0, 0, 0, 0};
		int64_t phot_ini = 0, phot_transm = 0;
		do {
			polycap_photon *ph = polycap_source_get_photon(source, rng, &error);
			CHECK(ph != NULL);
			if (ph == NULL) break;
			double *w = NULL;
			int test = polycap_photon_launch(ph, 7, energies, &w, false, &error);
			if (test == 1) {
				for (int j = 0; j < 7; j++) { CHECK(w[j] >= 0. && w[j] <= 1.); w_tot[j] += w[j]; }
				phot_transm++;
			}
			if (test != -2 && test != -1) phot_ini++;
			else polycap_clear_error(&error);
			free(w);
			polycap_photon_free(ph);
		} while (phot_transm < 3000);
		for (int j = 0; j < 7; j++) {
			w_tot[j] /= (double)phot_ini;
			CHECK(fabs(w_tot[j] - t_arr[j]) <= (j == 0 ? 0.02 : 0.0125));
		}
		printf("loop: %lld launched, %lld transmitted, eff(10 keV) %.4f vs driver %.4f\n", (long long)phot_ini, (long long)phot_transm, w_tot[2], t_arr[2]);
		polycap_rng_free(rng);
	}

	for (int64_t j = 0; j < n_exit2; j++) polycap_free(exit_weights[j]);
	polycap_free(exit_weights); polycap_free(ec); polycap_free(ed); polycap_free(ee); polycap_free(n_refl); polycap_free(d_travel);
	polycap_free(sc); polycap_free(sd); polycap_free(se); polycap_free(src_sc);
	polycap_free(e_arr); polycap_free(t_arr);
	polycap_transmission_efficiencies_free(eff);
	polycap_source_free(source);
	if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
	printf("dropin_client: all checks passed\n");
	return 0;
}
