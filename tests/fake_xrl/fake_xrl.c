/* Test double for xraylib's libxrl (tests/test_xraylib_binding.py builds it as libxrl.so.11 into a temporary directory).
 * It exports the three entry points polycap's path uses with xraylib 4.x's signatures and the stand-in formulas of
 * SURVEY.md section 8(c), which reproduce the one point the reference's tests pin (tests/photon.c:75-76) for ANY composition:
 *     CS_Total == 42.544677 / 2.23,   Fi == 0,   AtomicWeight(Z) == Z / 0.503696
 * FAKE_XRL_FAIL=CS_Total|Fi|AtomicWeight makes that function report an error the way xraylib does (allocates an xrl_error and
 * returns 0); without it the error pointer must be left untouched (NULL). */
#include <stdlib.h>
#include <string.h>

typedef struct { int code; char *message; } xrl_error;

static int calls[3];

static double fail_or(const char *name, double value, xrl_error **error)
{
	const char *f = getenv("FAKE_XRL_FAIL");
	if (f != NULL && strcmp(f, name) == 0 && error != NULL) {
		xrl_error *e = malloc(sizeof(*e));
		e->code = 5;      /* XRL_ERROR_RUNTIME */
		e->message = strdup("fake_xrl: requested failure");
		*error = e;
		return 0.0;
	}
	return value;
}

__attribute__((visibility("default"))) double CS_Total(int Z, double E, xrl_error **error)
{
	(void)Z; (void)E;
	calls[0]++;
	return fail_or("CS_Total", 42.544677/2.23, error);
}

__attribute__((visibility("default"))) double Fi(int Z, double E, xrl_error **error)
{
	(void)Z; (void)E;
	calls[1]++;
	return fail_or("Fi", 0.0, error);
}

__attribute__((visibility("default"))) double AtomicWeight(int Z, xrl_error **error)
{
	calls[2]++;
	return fail_or("AtomicWeight", (double)Z/0.503696, error);
}

__attribute__((visibility("default"))) void xrl_error_free(xrl_error *error)
{
	if (error == NULL) return;
	free(error->message);
	free(error);
}

/* how often each entry point ran (the test checks that the values really passed through this library) */
__attribute__((visibility("default"))) int fake_xrl_calls(int which)
{
	return (which >= 0 && which < 3) ? calls[which] : -1;
}
