/*
 * pc_error.c -- polycap_error objects (GLib GError convention).
 * Behaviour follows the reference's src/polycap-error.c:22-182: errors are only stored into a NULL *err,
 * a second error over an existing one is reported on stderr and dropped.
 */
#define _GNU_SOURCE
#include "pc_private.h"

#include <stdlib.h>
#include <string.h>

static char *pc_vformat(const char *format, va_list args)
{
	va_list copy;
	va_copy(copy, args);
	int need = vsnprintf(NULL, 0, format, copy);
	va_end(copy);
	if (need < 0)
		return NULL;
	char *out = malloc((size_t)need + 1);
	if (out == NULL)
		return NULL;
	vsnprintf(out, (size_t)need + 1, format, args);
	return out;
}

static void pc_warn_overwrite(const char *message)
{
	fprintf(stderr, "polycap_error set over the top of a previous polycap_error or uninitialized memory.\n"
	        "This indicates a bug in someone's code. You must ensure an error is NULL before it's set.\n"
	        "The overwriting error message was: %s", message);
}

polycap_error *polycap_error_new_valist(enum polycap_error_code code, const char *format, va_list args)
{
	if (format == NULL) {
		fprintf(stderr, "polycap_error_new_valist: format cannot be NULL!\n");
		return NULL;
	}
	polycap_error *error = malloc(sizeof(polycap_error));
	if (error == NULL)
		return NULL;
	error->code = code;
	error->message = pc_vformat(format, args);
	return error;
}

polycap_error *polycap_error_new(enum polycap_error_code code, const char *format, ...)
{
	if (format == NULL) {
		fprintf(stderr, "polycap_error_new: format cannot be NULL!\n");
		return NULL;
	}
	va_list args;
	va_start(args, format);
	polycap_error *error = polycap_error_new_valist(code, format, args);
	va_end(args);
	return error;
}

polycap_error *polycap_error_new_literal(enum polycap_error_code code, const char *message)
{
	if (message == NULL) {
		fprintf(stderr, "polycap_error_new_literal: message cannot be NULL!\n");
		return NULL;
	}
	polycap_error *error = malloc(sizeof(polycap_error));
	if (error == NULL)
		return NULL;
	error->code = code;
	error->message = strdup(message);
	return error;
}

void polycap_error_free(polycap_error *error)
{
	if (error == NULL)
		return;
	free(error->message);
	free(error);
}

polycap_error *polycap_error_copy(const polycap_error *error)
{
	if (error == NULL)
		return NULL;
	polycap_error *copy = malloc(sizeof(polycap_error));
	if (copy == NULL)
		return NULL;
	copy->code = error->code;
	copy->message = error->message ? strdup(error->message) : NULL;
	return copy;
}

bool polycap_error_matches(const polycap_error *error, enum polycap_error_code code)
{
	return error && error->code == code;
}

void polycap_set_error(polycap_error **err, enum polycap_error_code code, const char *format, ...)
{
	if (err == NULL)
		return;
	va_list args;
	va_start(args, format);
	polycap_error *fresh = polycap_error_new_valist(code, format, args);
	va_end(args);
	if (*err == NULL) {
		*err = fresh;
	} else {
		pc_warn_overwrite(fresh ? fresh->message : "(null)");
		polycap_error_free(fresh);
	}
}

void polycap_set_error_literal(polycap_error **err, enum polycap_error_code code, const char *message)
{
	if (err == NULL)
		return;
	if (*err == NULL)
		*err = polycap_error_new_literal(code, message);
	else
		pc_warn_overwrite(message);
}

void polycap_propagate_error(polycap_error **dest, polycap_error *src)
{
	if (src == NULL) {
		fprintf(stderr, "polycap_propagate_error: src cannot be NULL");
		return;
	}
	if (dest == NULL) {
		polycap_error_free(src);
		return;
	}
	if (*dest != NULL) {
		pc_warn_overwrite(src->message);
		polycap_error_free(src);
	} else {
		*dest = src;
	}
}

void polycap_clear_error(polycap_error **err)
{
	if (err && *err) {
		polycap_error_free(*err);
		*err = NULL;
	}
}

void polycap_free(void *data)
{
	free(data);
}
