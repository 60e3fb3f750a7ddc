// comm.cpp -- the gradient exchange of the data-parallel step on RCCL, behind the C ABI (comm_init /
// allreduce_bucket / comm_destroy).  The reference wraps its model in nn.DataParallel (main_dgl.py:244); the MI355X form is
// one process per GPU whose flat gradient buckets (gdl/ddp.py) are summed over xGMI.  RCCL is bound at run time
// (dlopen of librccl.so.1: inside a PyTorch process that is the copy PyTorch itself loaded), so libgdl_hip.so carries no
// link-time dependency on it and every other entry point works where RCCL is absent.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace gdl {
namespace {
// the handful of RCCL (= NCCL API) symbols used; ncclUniqueId is 128 opaque bytes, ncclResult_t 0 = success,
// ncclFloat32 = 7, ncclSum = 0 (nccl.h)
struct NcclId {
    char internal[128];
};
typedef int (*GetUniqueIdFn)(NcclId*);
typedef int (*CommInitRankFn)(void**, int, NcclId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    int (*comm_count)(void*, int*) = nullptr;  // ncclCommCount: the communicator's own idea of its size
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return GDL_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        set_error("gdl_comm: RCCL not found (%s)", dlerror());
        return GDL_ERR_STATE;
    }
    Rccl r;
    r.lib = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    r.comm_count = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
        set_error("gdl_comm: librccl lacks the NCCL API symbols");
        dlclose(h);
        return GDL_ERR_STATE;
    }
    g_rccl = r;
    return GDL_OK;
}
int check_nccl(int rc, const char* what) {
    if (rc == 0) return GDL_OK;
    set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return GDL_ERR_HIP;
}
}  // namespace
}  // namespace gdl

using namespace gdl;

struct gdl_comm {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int gdl_comm_unique_id(void* id128) {
    GDL_REQUIRE(id128, "comm_unique_id: null");
    int rc = load_rccl();
    if (rc) return rc;
    NcclId id;
    rc = check_nccl(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id128, id.internal, sizeof(id.internal));
    return GDL_OK;
}

int gdl_comm_init(gdl_comm_t** out, int rank, int world, const void* id128) {
    GDL_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
    int rc = load_rccl();
    if (rc) return rc;
    NcclId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    gdl_comm* c = new gdl_comm();
    c->rank = rank;
    c->world = world;
    rc = check_nccl(g_rccl.comm_init_rank(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc) {
        delete c;
        return rc;
    }
    *out = c;
    return GDL_OK;
}

// the number of ranks AS RCCL REPORTS IT (ncclCommCount) -- what bench.py prints as comm.nranks, so that "did RCCL see N ranks" can be
// read off the line; falls back to the size the communicator was created with if the library lacks the symbol
int gdl_comm_world(const gdl_comm_t* c) {
    if (!c) return 0;
    int n = 0;
    if (g_rccl.comm_count && c->comm && g_rccl.comm_count(c->comm, &n) == 0 && n > 0) return n;
    return c->world;
}

int gdl_comm_allreduce_bucket(gdl_comm_t* c, float* grads, size_t count, void* stream) {
    GDL_REQUIRE(c && c->comm && (grads || count == 0), "comm_allreduce_bucket: bad arguments");
    if (count == 0) return GDL_OK;
    return check_nccl(g_rccl.all_reduce(grads, grads, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, (hipStream_t)stream),
                      "ncclAllReduce");
}

int gdl_comm_destroy(gdl_comm_t* c) {
    if (!c) return GDL_OK;
    int rc = GDL_OK;
    if (c->comm) rc = check_nccl(g_rccl.comm_destroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

}  // extern "C"
