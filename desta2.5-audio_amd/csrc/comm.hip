// Data-parallel gradient exchange behind the C ABI (desta_allreduce_grads; row A13): the mean over ranks of the
// flat fp32 gradient arena as ONE RCCL all-reduce on the caller's HIP stream.  What DDP's bucketed reducer does for the
// reference (accelerate `accelerator.py:1892`, torch Reducer), for a host that is not PyTorch.  The Python host layer uses
// torch.distributed (backend "nccl" = RCCL) for the same collective; this file is the same exchange without torch.
//
// RCCL is bound at FIRST USE with dlopen (librccl.so.1 of the ROCm install, or the copy a host process has already loaded):
// the library itself keeps linking against libamdhip64 only, so it loads — and every compute entry point works — on a
// box without RCCL.
#include "common.h"
#include "desta_hip.h"
#include <dlfcn.h>
#include <string.h>
#include <atomic>
#include <mutex>

namespace {
typedef int (*get_uid_fn)(void*);
typedef int (*init_rank_fn)(void**, int, desta_comm_unique_id, int);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*destroy_fn)(void*);
typedef const char* (*errstr_fn)(int);
struct Rccl { void* h = nullptr; get_uid_fn uid = nullptr; init_rank_fn init = nullptr; allreduce_fn ar = nullptr; destroy_fn destroy = nullptr; errstr_fn err = nullptr; };
Rccl g_rccl;
constexpr int kNcclFloat32 = 7, kNcclAvg = 4;               // rccl.h: ncclFloat32 = 7, ncclAvg = 4

std::mutex g_rccl_mu;
std::atomic<bool> g_rccl_ready{false};
// Thread-safe: the table is resolved into a LOCAL struct under a mutex and published whole (release store of the ready flag);
// readers take the acquire load first, so no thread ever sees a half-filled table.
int rccl_load() {
    if (g_rccl_ready.load(std::memory_order_acquire)) return DESTA_OK;
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_ready.load(std::memory_order_relaxed)) return DESTA_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }      // a copy the process already holds (torch's)
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { desta_set_error("desta_comm: cannot load librccl (%s)", dlerror()); return DESTA_ELAUNCH; }
    Rccl t;
    t.h = h;
    t.uid = (get_uid_fn)dlsym(h, "ncclGetUniqueId");
    t.init = (init_rank_fn)dlsym(h, "ncclCommInitRank");
    t.ar = (allreduce_fn)dlsym(h, "ncclAllReduce");
    t.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
    t.err = (errstr_fn)dlsym(h, "ncclGetErrorString");
    if (!t.uid || !t.init || !t.ar || !t.destroy) {
        desta_set_error("desta_comm: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return DESTA_ELAUNCH;
    }
    g_rccl = t;
    g_rccl_ready.store(true, std::memory_order_release);
    return DESTA_OK;
}
int rccl_check(int rc, const char* what) {
    if (rc == 0) return DESTA_OK;
    desta_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.err ? g_rccl.err(rc) : "?");
    return DESTA_ELAUNCH;
}
}  // namespace

extern "C" int desta_comm_get_unique_id(desta_comm_unique_id* id) {
    DESTA_CHECK_ARG(id, "desta_comm_get_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    return rccl_check(g_rccl.uid(id), "desta_comm_get_unique_id");
}

extern "C" int desta_comm_create(desta_comm* comm, int world_size, int rank, const desta_comm_unique_id* id) {
    DESTA_CHECK_ARG(comm && id && world_size >= 1 && rank >= 0 && rank < world_size, "desta_comm_create: bad argument");
    if (int rc = rccl_load()) return rc;
    void* c = nullptr;
    if (int rc = rccl_check(g_rccl.init(&c, world_size, *id, rank), "desta_comm_create")) return rc;
    *comm = c;
    return DESTA_OK;
}

extern "C" int desta_allreduce_grads(desta_comm comm, float* grads, int64_t n, void* stream) {
    DESTA_CHECK_ARG(comm && grads && n > 0, "desta_allreduce_grads: bad argument");
    if (int rc = rccl_load()) return rc;
    return rccl_check(g_rccl.ar(grads, grads, (size_t)n, kNcclFloat32, kNcclAvg, comm, (hipStream_t)stream), "desta_allreduce_grads");
}

extern "C" int desta_comm_destroy(desta_comm comm) {
    if (!comm) return DESTA_OK;
    if (int rc = rccl_load()) return rc;
    return rccl_check(g_rccl.destroy(comm), "desta_comm_destroy");
}
