/*
 * oracle/ref_loader.c -- TEST INFRASTRUCTURE. dlopen with LAZY binding for the reference-compiled libraries under
 * oracle/_ref/ that carry unresolved symbols of code paths nobody calls (arm_rfft_fast_* in libmfcc_f32_ref.so: their
 * tables are absent from the reference snapshot). Python's ctypes always adds RTLD_NOW, which refuses such a library.
 */
#include <dlfcn.h>
#include <stddef.h>

void *oracle_dl_open_lazy(const char *path) { return dlopen(path, RTLD_LAZY | RTLD_LOCAL); }
void *oracle_dl_sym(void *handle, const char *name) { return handle ? dlsym(handle, name) : NULL; }
const char *oracle_dl_error(void) { return dlerror(); }
