/*
 * pc_rng.c -- polycap_rng on top of Philox4x32-10 (counter based).
 *
 * API of the reference's src/polycap-rng.c:31-95 (new / new_with_seed / free).  The reference wraps GSL's
 * mt19937; no reference test pins stream values (tests/source.c:50,313 only use seeded streams for range and
 * statistical checks), so the generator is replaced by the same counter-based Philox the device kernels use:
 * stream = (seed, photon index), which makes every photon reproducible and independent of scheduling.
 */
#include "pc_private.h"

#include <stdlib.h>
#include <sys/time.h>

polycap_rng *polycap_rng_new_with_seed(unsigned long int seed)
{
	polycap_rng *rng = calloc(1, sizeof(polycap_rng));
	if (rng == NULL)
		return NULL;
	rng->seed = (uint64_t)seed;
	rng->counter = 0;
	return rng;
}

/* seed from /dev/urandom, falling back to the clock (reference src/polycap-rng.c:50-71) */
polycap_rng *polycap_rng_new(void)
{
	unsigned long int seed = 0;
	int have = 0;
	FILE *random_device = fopen("/dev/urandom", "r");
	if (random_device != NULL) {
		have = fread(&seed, sizeof(seed), 1, random_device) == 1;
		fclose(random_device);
	}
	if (!have) {
		struct timeval tv;
		gettimeofday(&tv, NULL);
		seed = (unsigned long int)tv.tv_sec * 1000003ul + (unsigned long int)tv.tv_usec;
	}
	return polycap_rng_new_with_seed(seed);
}

void polycap_rng_free(polycap_rng *rng)
{
	free(rng);
}
