// comm_rccl.cpp -- the one exchange step of the sharded path through the C ABI: a sum-allreduce of the int64 count
// tables over RCCL (xGMI inside a node), for callers that shard reads across GPUs WITHOUT torch.distributed
// (kbbq_allreduce_tables(kbbq_comm*, ...); 8(e): reads shard, the tables add, nothing else is exchanged).
//
// librccl is half a gigabyte and torch ships its own copy: it is NOT a link-time dependency of libkbbq_hip.so.  The
// first kbbq_comm_* call binds the five entry points it needs with dlopen / dlsym -- a copy the process has already
// loaded (torch's, when torch.distributed runs beside this) is re-used, otherwise KBBQ_RCCL_LIB, then librccl.so.1 on the
// loader's path, then /opt/rocm/lib.  Types and enum values are RCCL's public ABI (rccl.h: ncclUniqueId is 128 opaque
// bytes, ncclInt64 = 4, ncclSum = 0).
#include "../../include/kbbq_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

int kbbq_set_error_(int code, const char* msg);      // kbbq_hip.hip
extern "C" void* kbbq_ctx_stream_(kbbq_ctx* c);      // kbbq_hip.hip: the context's launch stream
extern "C" int kbbq_ctx_device_(kbbq_ctx* c);

namespace {

struct UniqueId { char internal[KBBQ_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    char why[256] = {0};
};

std::mutex g_mutex;
Rccl g_rccl;
char g_path[1024] = {0};

bool bind_rccl()
{
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_rccl.handle) return true;
    const char* env = getenv("KBBQ_RCCL_LIB");
    const char* candidates[] = {g_path[0] ? g_path : nullptr, env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // a copy already in the process first (two RCCLs would each bring their own idea of the devices)
    for (const char* name : {"librccl.so.1", "librccl.so"})
        if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : candidates)
        if (!h && name && name[0]) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl not found (%s); set KBBQ_RCCL_LIB", dlerror()); return false; }
    g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
        snprintf(g_rccl.why, sizeof g_rccl.why, "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return false;
    }
    g_rccl.handle = h;
    return true;
}

int rccl_fail(const char* what, int rc)
{
    char buf[384];
    snprintf(buf, sizeof buf, "%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return kbbq_set_error_(KBBQ_E_HIP, buf);
}

}  // namespace

struct kbbq_comm {
    kbbq_ctx* ctx;
    Comm comm;
    int nranks, rank;
};

extern "C" {

int kbbq_comm_library(const char* path)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_rccl.handle) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_comm_library: RCCL is already bound");
    snprintf(g_path, sizeof g_path, "%s", path ? path : "");
    return KBBQ_OK;
}

int kbbq_comm_unique_id(void* id128)
{
    if (!id128) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_comm_unique_id: NULL");
    if (!bind_rccl()) return kbbq_set_error_(KBBQ_E_HIP, g_rccl.why);
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, KBBQ_COMM_ID_BYTES);
    return KBBQ_OK;
}

int kbbq_comm_create(kbbq_ctx* ctx, const void* id128, int nranks, int rank, kbbq_comm** out)
{
    if (!ctx || !id128 || !out) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_comm_create: NULL argument");
    *out = nullptr;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_comm_create: rank outside 0..nranks-1");
    if (!bind_rccl()) return kbbq_set_error_(KBBQ_E_HIP, g_rccl.why);
    if (hipSetDevice(kbbq_ctx_device_(ctx)) != hipSuccess) return kbbq_set_error_(KBBQ_E_HIP, "kbbq_comm_create: hipSetDevice failed");
    UniqueId id;
    memcpy(id.internal, id128, KBBQ_COMM_ID_BYTES);
    Comm comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&comm, nranks, id, rank);
    if (rc) return rccl_fail("ncclCommInitRank", rc);
    kbbq_comm* c = new kbbq_comm{ctx, comm, nranks, rank};
    *out = c;
    return KBBQ_OK;
}

int kbbq_allreduce_tables(kbbq_comm* comm, int64_t* d_buf, size_t n)
{
    if (!comm) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_allreduce_tables: comm is NULL");
    if (n && !d_buf) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_allreduce_tables: buffer is NULL");
    if (n == 0) return KBBQ_OK;
    if (hipSetDevice(kbbq_ctx_device_(comm->ctx)) != hipSuccess) return kbbq_set_error_(KBBQ_E_HIP, "kbbq_allreduce_tables: hipSetDevice failed");
    // in place, int64 sum, enqueued on the context's stream: ordered after K1 and before K3 without a host wait
    const int rc = g_rccl.all_reduce(d_buf, d_buf, n, /*ncclInt64*/ 4, /*ncclSum*/ 0, comm->comm,
                                     (hipStream_t)kbbq_ctx_stream_(comm->ctx));
    if (rc) return rccl_fail("ncclAllReduce", rc);
    return KBBQ_OK;
}

int kbbq_comm_destroy(kbbq_comm* comm)
{
    if (!comm) return KBBQ_OK;
    int rc = 0;
    if (g_rccl.comm_destroy && comm->comm) rc = g_rccl.comm_destroy(comm->comm);
    delete comm;
    return rc ? rccl_fail("ncclCommDestroy", rc) : KBBQ_OK;
}

}  // extern "C"
