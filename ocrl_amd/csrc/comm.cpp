// Gradient all-reduce over RCCL for hosts that drive the C ABI without torch.distributed (one SUM all-reduce of
// the flat fp32 gradient buffer per step; the 1/world mean is folded into ocrl_*_clip_adam's gscale).  librccl is opened
// lazily with dlopen, so the library has no link-time dependency on it and a process that already carries another copy of
// RCCL (PyTorch-ROCm) is not disturbed; the Python surface keeps using torch.distributed's "nccl" backend (= RCCL).
#include <dlfcn.h>
#include <string.h>

#include <new>

#include "../../include/ocrl_hip.h"
#include "common.h"

namespace {
typedef struct { char internal[128]; } UniqueId;                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
struct Api {
    void* lib = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init = nullptr;
    AllReduceFn allreduce = nullptr;
    CommDestroyFn destroy = nullptr;
    GetErrorStringFn errstr = nullptr;
};
Api g_api;
int load_api() {
    if (g_api.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    OCRL_REQUIRE(lib, "ocrl_comm: cannot open librccl (%s)", dlerror());
    g_api.get_id = (GetUniqueIdFn)dlsym(lib, "ncclGetUniqueId");
    g_api.init = (CommInitRankFn)dlsym(lib, "ncclCommInitRank");
    g_api.allreduce = (AllReduceFn)dlsym(lib, "ncclAllReduce");
    g_api.destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
    g_api.errstr = (GetErrorStringFn)dlsym(lib, "ncclGetErrorString");
    OCRL_REQUIRE(g_api.get_id && g_api.init && g_api.allreduce && g_api.destroy, "ocrl_comm: librccl lacks the expected entry points");
    g_api.lib = lib;
    return 0;
}
const char* err(int rc) { return g_api.errstr ? g_api.errstr(rc) : "rccl error"; }
}  // namespace

struct ocrl_comm {
    void* comm;
    int rank, world;
};

extern "C" {

int ocrl_comm_unique_id(void* out, size_t cap) {
    OCRL_REQUIRE(out && cap >= sizeof(UniqueId), "ocrl_comm_unique_id: need a %zu-byte buffer", sizeof(UniqueId));
    if (load_api()) return 1;
    UniqueId id;
    const int rc = g_api.get_id(&id);
    OCRL_REQUIRE(rc == 0, "ncclGetUniqueId: %s", err(rc));
    memcpy(out, &id, sizeof id);
    return 0;
}

int ocrl_comm_init(ocrl_comm** out, int rank, int world, const void* unique_id) {
    OCRL_REQUIRE(out && unique_id && world >= 1 && rank >= 0 && rank < world, "ocrl_comm_init: bad arguments");
    if (load_api()) return 1;
    UniqueId id;
    memcpy(&id, unique_id, sizeof id);
    void* comm = nullptr;
    const int rc = g_api.init(&comm, world, id, rank);      // uses the calling thread's current HIP device
    OCRL_REQUIRE(rc == 0, "ncclCommInitRank: %s", err(rc));
    ocrl_comm* c = new (std::nothrow) ocrl_comm;
    if (!c) { g_api.destroy(comm); ocrl_set_error("out of memory"); return 1; }
    c->comm = comm; c->rank = rank; c->world = world;
    *out = c;
    return 0;
}

int ocrl_comm_allreduce(ocrl_comm* c, float* buf, long long n, void* stream) {
    OCRL_REQUIRE(c && c->comm && buf && n > 0, "ocrl_comm_allreduce: bad arguments");
    const int rc = g_api.allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, static_cast<hipStream_t>(stream));
    OCRL_REQUIRE(rc == 0, "ncclAllReduce: %s", err(rc));
    return 0;
}

int ocrl_comm_world(const ocrl_comm* c) { return c ? c->world : -1; }

void ocrl_comm_destroy(ocrl_comm* c) {
    if (!c) return;
    if (c->comm && g_api.destroy) g_api.destroy(c->comm);
    delete c;
}

}  // extern "C"
