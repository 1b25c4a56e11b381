/* Prototype-only declarations of the few R C API names singlet_amd/r/singlet_hip_shim.c uses, so
 * that tests/test_abi.py can syntax-check the shim's calls against include/singlet_hip.h in an image
 * without R.  Nothing here is linked or run; a real build uses R's own headers. */
#ifndef R_STUB_H
#define R_STUB_H
#include <stddef.h>
void Rprintf(const char*, ...);
void Rf_error(const char*, ...) __attribute__((noreturn));
char* R_alloc(size_t, int);
#define ISNAN(x) ((x) != (x))
#endif
