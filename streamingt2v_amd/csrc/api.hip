// Library-level entry points of libsvdhip.so (version + error text).
#include "svd_common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

extern "C" void svd_set_error_(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
extern "C" int svd_abi_version(void) { return 11; }
extern "C" const char* svd_last_error(void) { return g_err; }
