// Error plumbing shared by every entry point of libdesta_hip.so.
#include "common.h"
#include "desta_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void desta_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int desta_abi_version(void) { return DESTA_ABI_VERSION; }
extern "C" size_t desta_sizeof_desc(int which) {
    return which == 0 ? sizeof(desta_gemm_desc) : which == 1 ? sizeof(desta_attn_desc) : which == 2 ? sizeof(desta_opt_plan) : 0;
}
extern "C" const char* desta_last_error(void) { return g_err; }
