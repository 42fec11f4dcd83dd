// iaf_comm.cpp -- the data-parallel gradient exchange behind the C ABI: RCCL all-reduce(sum) over xGMI.
//
// Reference semantic: tf_utils/common.py:83-86 (average_grads: per-variable sum over the towers, times 1/N) under
// tf_train.py:124-147 (one tower per GPU).  Here every rank owns one GPU and one flat fp32 gradient buffer; the exchange is
// ncclAllReduce(sum) in place on caller-chosen [ptr, ptr + n) segments of it (the buckets of iaf_amd/parallel.py), and the
// 1/N rides in the fused Adamax + EMA launch (iaf_adamax_ema_step's grad_scale).  A non-Python host gets the whole DP step
// from this library: iaf_comm_unique_id on rank 0 -> the 128 bytes travel by whatever the host has (a file, MPI, a socket) ->
// iaf_comm_create on every rank -> iaf_allreduce_sum_f32 per bucket on a stream of its choice -> iaf_adamax_ema_step.
//
// librccl is bound at RUN time (dlopen, no link-time dependency): a process that already carries an RCCL -- PyTorch ships
// its own librccl.so and loads it with libtorch_hip -- gets THAT copy (one RCCL per process: one set of proxy threads, one
// topology detection); otherwise the system one (/opt/rocm/lib/librccl.so.1) is loaded.  Forward-only users never touch it.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>     // types and enums only: every entry point is resolved through dlsym below

#include "iaf_hip.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char origin[256] = {0};
};

RcclApi g_api;
std::once_flag g_once;
int g_load_rc = IAF_ERR_UNSUPPORTED;

void load_rccl() {
    static const char* const names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)          // a copy this process already carries (PyTorch's) first
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    for (const char* n : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    g_api.handle = h;
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.CommCount = (decltype(g_api.CommCount))dlsym(h, "ncclCommCount");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.AllReduce || !g_api.CommDestroy) return;
    Dl_info info;
    if (dladdr((void*)g_api.AllReduce, &info) && info.dli_fname) snprintf(g_api.origin, sizeof(g_api.origin), "%s", info.dli_fname);
    g_load_rc = IAF_OK;
}

int rccl() {
    std::call_once(g_once, load_rccl);
    return g_load_rc;
}

// RCCL status -> this ABI's int: 0 ok, else a positive code disjoint from hipError_t values a caller may also see
inline int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? IAF_OK : 10000 + (int)r; }

}  // namespace

struct iaf_comm {
    ncclComm_t comm;
    int rank, world, device;
};

extern "C" int iaf_comm_unique_id(void* id_out) {
    if (!id_out) return IAF_ERR_NULL;
    static_assert(sizeof(ncclUniqueId) == IAF_COMM_ID_BYTES, "IAF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    int rc = rccl();
    if (rc) return rc;
    ncclUniqueId id;
    if ((rc = nccl_rc(g_api.GetUniqueId(&id)))) return rc;
    memcpy(id_out, &id, sizeof(id));
    return IAF_OK;
}

extern "C" int iaf_comm_create(iaf_comm_t** out, const void* id, int rank, int world, int device) {
    if (!out || !id) return IAF_ERR_NULL;
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world || device < 0) return IAF_ERR_SHAPE;
    int rc = rccl();
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return IAF_ERR_SHAPE;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if ((rc = (int)hipSetDevice(device))) return rc;
    iaf_comm* c = new (std::nothrow) iaf_comm();
    if (!c) { (void)hipSetDevice(prev); return (int)hipErrorOutOfMemory; }
    c->rank = rank; c->world = world; c->device = device; c->comm = nullptr;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    rc = nccl_rc(g_api.CommInitRank(&c->comm, world, uid, rank));       // collective over the `world` ranks holding this id
    (void)hipSetDevice(prev);
    if (rc) { delete c; return rc; }
    *out = c;
    return IAF_OK;
}

extern "C" int iaf_comm_size(const iaf_comm_t* c, int* rank, int* world) {
    if (!c) return IAF_ERR_NULL;
    if (rank) *rank = c->rank;
    if (world) {
        *world = c->world;
        int n = 0;                                        // read back from the communicator where the library offers it
        if (g_api.CommCount && g_api.CommCount(c->comm, &n) == ncclSuccess) *world = n;
    }
    return IAF_OK;
}

extern "C" int iaf_allreduce_sum_f32(iaf_comm_t* c, float* buf, size_t n, void* stream) {
    if (!c || !buf) return IAF_ERR_NULL;
    if (n == 0) return IAF_OK;
    return nccl_rc(g_api.AllReduce(buf, buf, n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
}

extern "C" int iaf_comm_destroy(iaf_comm_t* c) {
    if (!c) return IAF_ERR_NULL;
    int rc = IAF_OK;
    if (c->comm) rc = nccl_rc(g_api.CommDestroy(c->comm));
    delete c;
    return rc;
}

extern "C" const char* iaf_comm_library(void) {
    return rccl() == IAF_OK ? g_api.origin : "";
}

// (called by iaf_error_string for codes >= 10000)
extern "C" const char* iaf_comm_error_string_(int code) {
    if (code < 10000 || rccl() != IAF_OK || !g_api.GetErrorString) return nullptr;
    return g_api.GetErrorString((ncclResult_t)(code - 10000));
}
