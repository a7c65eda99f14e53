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
extern "C" const char* desta_last_error(void) { return g_err; }
