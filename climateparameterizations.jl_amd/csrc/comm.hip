// comm.hip — the one exchange step of the path behind the C ABI: a SUM all-reduce of the result buffer
// [grad(n_params); 6 loss terms; total; 0] over RCCL (xGMI inside a node), so that a Julia (or any non-torch) host can shard
// columns over GPUs, one process per GPU.  RCCL is bound lazily (dlopen of librccl.so.1 at the first colnde_comm_* call):
// the library itself keeps no link-time dependency on it, loads on a box without RCCL, and fails loudly when asked to
// communicate there.  In a torch process the SONAME is already bound to torch's bundled copy, which is the one that gets used.
// The reference has no distributed code: MI355X-native addition, no counterpart to cite.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <new>
#include "../../include/colnde.h"

extern "C" int colnde_internal_set_error(const char* msg);   // api.hip: stores the thread-local message, returns 1

namespace {
typedef struct { char internal[128]; } nccl_uid;     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, nccl_comm, hipStream_t);
typedef int (*fn_destroy)(nccl_comm);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* so = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;

int bind_rccl() {
    if (g_rccl.so) return 0;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) {
        char b[512];
        snprintf(b, sizeof b, "RCCL is not loadable (%s): multi-GPU exchange unavailable on this box", dlerror());
        return colnde_internal_set_error(b);
    }
    Rccl r;
    r.so = so;
    r.get_uid = (fn_get_uid)dlsym(so, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(so, "ncclCommInitRank");
    r.allreduce = (fn_allreduce)dlsym(so, "ncclAllReduce");
    r.destroy = (fn_destroy)dlsym(so, "ncclCommDestroy");
    r.errstr = (fn_errstr)dlsym(so, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.allreduce || !r.destroy) return colnde_internal_set_error("librccl lacks an expected ncclXxx symbol");
    g_rccl = r;
    return 0;
}

int nccl_fail(const char* what, int rc) {
    char b[512];
    snprintf(b, sizeof b, "%s failed: %s (ncclResult %d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?", rc);
    return colnde_internal_set_error(b);
}
}  // namespace

struct colnde_comm {
    nccl_comm comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

extern "C" int colnde_comm_unique_id(void* out128) {
    if (!out128) return colnde_internal_set_error("null unique-id buffer");
    if (bind_rccl()) return 1;
    nccl_uid id;
    const int rc = g_rccl.get_uid(&id);
    if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, 128);
    return 0;
}

extern "C" int colnde_comm_create(int rank, int nranks, const void* unique_id128, int device, colnde_comm** out) {
    if (!out) return colnde_internal_set_error("null out pointer");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return colnde_internal_set_error("need 0 <= rank < nranks");
    if (!unique_id128) return colnde_internal_set_error("null unique id (rank 0 makes one with colnde_comm_unique_id and hands the 128 bytes to every rank)");
    if (bind_rccl()) return 1;
    if (hipSetDevice(device) != hipSuccess) return colnde_internal_set_error("hipSetDevice failed for the communicator's device");
    colnde_comm* c = new (std::nothrow) colnde_comm();
    if (!c) return colnde_internal_set_error("out of host memory");
    c->rank = rank; c->nranks = nranks; c->device = device;
    nccl_uid id;
    memcpy(id.internal, unique_id128, 128);
    const int rc = g_rccl.init_rank(&c->comm, nranks, id, rank);
    if (rc != 0) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return 0;
}

extern "C" void colnde_comm_destroy(colnde_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm);
    delete c;
}

extern "C" int colnde_comm_rank(const colnde_comm* c) { return c ? c->rank : -1; }
extern "C" int colnde_comm_size(const colnde_comm* c) { return c ? c->nranks : -1; }

// in-place all-reduce of n device floats on `stream` (enqueued, not synchronised); op 0 = sum, 1 = max
extern "C" int colnde_comm_allreduce_dev(colnde_comm* c, float* d_buf, int64_t n, int op, void* hip_stream) {
    if (!c || !c->comm) return colnde_internal_set_error("null communicator");
    if (!d_buf || n < 1) return colnde_internal_set_error("null buffer or n < 1");
    if (op != 0 && op != 1) return colnde_internal_set_error("op must be 0 (sum) or 1 (max)");
    if (hipSetDevice(c->device) != hipSuccess) return colnde_internal_set_error("hipSetDevice failed");
    const int rc = g_rccl.allreduce(d_buf, d_buf, (size_t)n, /*ncclFloat32*/ 7, op == 0 ? /*ncclSum*/ 0 : /*ncclMax*/ 2, c->comm,
                                    (hipStream_t)hip_stream);
    if (rc != 0) return nccl_fail("ncclAllReduce", rc);
    return 0;
}
