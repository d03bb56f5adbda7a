/*
 * pc_main.c -- the `polycap` command-line program: input deck -> transmission efficiencies -> HDF5 result file.
 *
 * Same positional interface as the reference's src/main.c:24-96
 *     polycap input-file.inp [output.h5 [threads [leak_calc [n_photons]]]]
 * output defaults to polycap_out.h5, `threads` is accepted and ignored (the photon loop runs on the GPU selected by
 * POLYCAP_HIP_DEVICE), leak_calc = 1 switches the halo calculation on, 30000 exit photons unless a fifth argument
 * (an addition of this build) says otherwise.  Exit status 1 with the library's message on any failure.
 */
#include <polycap.h>

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char *argv[])
{
	polycap_error *error = NULL;
	const char *filename = "polycap_out.h5";
	int n_photons = 30000;
	bool leak_calc = false;

	if (argc <= 1) {
		printf("Usage: polycap input-file should be supplied.\n");
		return 0;
	}
	if (argc >= 3)
		filename = argv[2];
	if (argc >= 5 && atoi(argv[4]) == 1)
		leak_calc = true;
	if (argc >= 6 && atoi(argv[5]) > 0)
		n_photons = atoi(argv[5]);

	polycap_source *source = polycap_source_new_from_file(argv[1], &error);
	if (source == NULL) {
		fprintf(stderr, "%s\n", error ? error->message : "polycap: could not read the input file");
		return 1;
	}
	printf("Starting calculations...\n");
	polycap_transmission_efficiencies *efficiencies = polycap_source_get_transmission_efficiencies(source, -1, n_photons, leak_calc, NULL, &error);
	if (efficiencies == NULL) {
		fprintf(stderr, "%s\n", error ? error->message : "polycap: the calculation failed");
		polycap_source_free(source);
		return 1;
	}
	if (!polycap_transmission_efficiencies_write_hdf5(efficiencies, filename, &error)) {
		fprintf(stderr, "%s\n", error ? error->message : "polycap: could not write the result file");
		polycap_transmission_efficiencies_free(efficiencies);
		polycap_source_free(source);
		return 1;
	}
	polycap_transmission_efficiencies_free(efficiencies);
	polycap_source_free(source);
	return 0;
}
