// Error plumbing shared by every entry point of libdesta_hip.so.
#include "common.h"
#include "desta_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>

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

// ---- optional per-device context (desta_create / desta_destroy / desta_last_error(handle)) ----
// The data path is stateless (header, "Conventions"); what a context owns is what the library otherwise makes lazily at first
// use and keeps for the life of the process: the device's internal fork stream + events of the attention backward.
int desta_internal_reserve(void);
int desta_internal_release(void);
struct desta_context { int magic; int device; int compute_units; char arch[64]; char err[512]; };
static const int kCtxMagic = 0x44455354;   // "DEST"

extern "C" int desta_create(int device, desta_handle* out) {
    DESTA_CHECK_ARG(out, "desta_create: null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        desta_set_error("desta_create: device %d of %d visible", device, n);
        return DESTA_EINVAL;
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) { desta_set_error("desta_create: hipGetDeviceProperties failed"); return DESTA_ELAUNCH; }
    if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) {        // the code objects in this library are gfx950 only
        desta_set_error("desta_create: device %d is %s; this library holds gfx950 (MI355X) code only", device, pr.gcnArchName);
        return DESTA_EINVAL;
    }
    if (hipSetDevice(device) != hipSuccess) { desta_set_error("desta_create: hipSetDevice(%d) failed", device); return DESTA_ELAUNCH; }
    if (desta_internal_reserve() != DESTA_OK) { desta_set_error("desta_create: could not create the internal stream / events"); return DESTA_ELAUNCH; }
    desta_context* c = new (std::nothrow) desta_context();
    if (!c) { (void)desta_internal_release(); desta_set_error("desta_create: out of host memory"); return DESTA_ELAUNCH; }
    c->magic = kCtxMagic; c->device = device; c->compute_units = pr.multiProcessorCount;
    snprintf(c->arch, sizeof(c->arch), "%s", pr.gcnArchName);
    c->err[0] = 0;
    *out = c;
    return DESTA_OK;
}
extern "C" int desta_destroy(desta_handle h) {
    desta_context* c = (desta_context*)h;
    DESTA_CHECK_ARG(c && c->magic == kCtxMagic, "desta_destroy: not a live handle");
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    (void)desta_internal_release();
    (void)hipSetDevice(prev);
    c->magic = 0;
    delete c;
    return DESTA_OK;
}
extern "C" int desta_handle_info(desta_handle h, int* device, int* compute_units, char* arch, size_t arch_bytes) {
    desta_context* c = (desta_context*)h;
    DESTA_CHECK_ARG(c && c->magic == kCtxMagic, "desta_handle_info: not a live handle");
    if (device) *device = c->device;
    if (compute_units) *compute_units = c->compute_units;
    if (arch && arch_bytes) snprintf(arch, arch_bytes, "%s", c->arch);
    return DESTA_OK;
}
// the calling thread's last error text, copied into the handle (stays valid until the next call with this handle)
extern "C" const char* desta_handle_last_error(desta_handle h) {
    desta_context* c = (desta_context*)h;
    if (!c || c->magic != kCtxMagic) return "desta_handle_last_error: not a live handle";
    snprintf(c->err, sizeof(c->err), "%s", g_err);
    return c->err;
}
