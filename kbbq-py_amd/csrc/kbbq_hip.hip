// kbbq_hip.hip -- C ABI of libkbbq_hip.so (see include/kbbq_hip.h).
// gfx950 only.  Host side: launch geometry, device staging, status decoding.
#include "kbbq_kernels.h"
#include "kbbq_kernels_v3.h"
#include "kbbq_solve_kernels.h"
#include "kbbq_layout_kernels.h"
#include "kbbq_aligned_kernels.h"
#include "kbbq_k2_tile.h"
#include "../../include/kbbq_hip.h"
#include "host_threads.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

// used by fastq_host.cpp
int kbbq_set_error_(int code, const char* msg) { g_err = msg ? msg : ""; return code; }

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(KBBQ_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

struct kbbq_ctx {
    int device = 0;
    int cus = 0;
    int lds_bytes = 0;
    char name[128] = {0};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    u64* d_status = nullptr;          // [KBBQ_NSTATUS]
    int* d_stats = nullptr;           // [K7_NSTATS] scratch of kbbq_meta_stats_dev
    int* d_wgplan = nullptr;          // K2 (short-lived workgroups): first workgroup of every read group
    int wgplan_n = 0;
    void* d_ops4 = nullptr;           // K4: one 32-byte record of the first CIGAR operations per read (grown on demand)
    size_t ops4_bytes = 0;
    void* d_tally = nullptr;          // kbbq_tally_aligned_dev: [count | flags words of the reads | rows K4 still has to look at] (grown on demand)
    size_t tally_bytes = 0;
    // kbbq_accumulate / kbbq_apply (host buffers): two page-locked staging slabs and their device twins, a copy stream and
    // the events that order host copy -> upload -> kernel -> download slab by slab (grown on demand, kept for the next call)
    void* stage_host[2] = {nullptr, nullptr}; void* stage_dev[2] = {nullptr, nullptr}; size_t stage_bytes = 0;
    hipStream_t stage_stream = nullptr;      // uploads
    hipStream_t stage_down_stream = nullptr; // downloads (kbbq_apply): a stream of their own, so that slab k + 1 goes up while slab k comes back
    hipEvent_t stage_up[2] = {nullptr, nullptr}, stage_used[2] = {nullptr, nullptr}, stage_down[2] = {nullptr, nullptr};
    void* d_rowlut = nullptr;         // K2 on one-read-per-row planes: the LUT narrowed to the rows' pitch (grown on demand)
    size_t rowlut_bytes = 0;
    bool timing = false;
    // per-kernel event pairs recorded while timing is on
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];
    double ms_acc[2] = {0.0, 0.0};
    int64_t launches[2] = {0, 0};
};

#define KBBQ_NSTATUS 8
static const u64 ST_INIT[KBBQ_NSTATUS] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};

extern "C" {

int kbbq_abi_version(void) { return KBBQ_ABI_VERSION; }
const char* kbbq_last_error(void) { return g_err.c_str(); }

int kbbq_device_count(int* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(KBBQ_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return KBBQ_OK;
}

// host_affinity.cpp with the PCI address of HIP device `device`: a rank of a multi-GPU job binds its host threads to the
// NUMA node of ITS GPU (kbbq/parallel.py init_from_env, bench.py)
int kbbq_bind_host_to_device(int device, int* numa_node, int* ncpus)
{
    char busid[64] = {0};
    HIPCHK(hipDeviceGetPCIBusId(busid, (int)sizeof busid, device));
    return kbbq_bind_host_to_pci(busid, numa_node, ncpus);
}

int kbbq_ctx_create(int device, kbbq_ctx** out)
{
    if (!out) return fail(KBBQ_E_ARG, "kbbq_ctx_create: out is NULL");
    *out = nullptr;
    const bool dbg = getenv("KBBQ_CTX_DBG") != nullptr;         // where the context's first-use milliseconds go (stderr)
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[kbbq_ctx_create] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    HIPCHK(hipSetDevice(device));
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    lap("hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KBBQ_E_HIP, "libkbbq_hip is built for gfx950 only; device %d is %s", device, prop.gcnArchName);
    kbbq_ctx* c = new kbbq_ctx();
    c->device = device;
    c->cus = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (c->lds_bytes <= 0) c->lds_bytes = 160 * 1024;
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(KBBQ_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    lap("hipStreamCreate");
    e = hipMalloc((void**)&c->d_status, sizeof ST_INIT);
    if (e != hipSuccess) { (void)hipStreamDestroy(c->own_stream); delete c; return fail(KBBQ_E_HIP, "hipMalloc(status): %s", hipGetErrorString(e)); }
    e = hipMemcpy(c->d_status, ST_INIT, sizeof ST_INIT, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(c->d_status); (void)hipStreamDestroy(c->own_stream); delete c; return fail(KBBQ_E_HIP, "hipMemcpy(status): %s", hipGetErrorString(e)); }
    lap("hipMalloc + hipMemcpy");
    // allow the full 160 KiB of LDS as dynamic shared memory
    (void)hipFuncSetAttribute((const void*)k1_accumulate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    lap("first hipFuncSetAttribute");
    (void)hipFuncSetAttribute((const void*)k1_accumulate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k1v3_accumulate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k1v3_accumulate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<false, K1V3_DNREP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<true, K1V3_DNREP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
#define KBBQ_KM_ATTR(KJ_) \
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<false, K1V3_DNREP, true, KJ_>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes); \
    (void)hipFuncSetAttribute((const void*)(k1v3_accumulate<true, K1V3_DNREP, true, KJ_>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    KBBQ_KM_ATTR(13) KBBQ_KM_ATTR(19)
#undef KBBQ_KM_ATTR
    (void)hipFuncSetAttribute((const void*)(k2v3_apply<false>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k2v3_apply<true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k2t_apply<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k2t_apply<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k2t_bands, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned<false, K1V3_DNREP>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned<true, K1V3_DNREP>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned_ref<false, K1V3_DNREP>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned_ref<true, K1V3_DNREP>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned_ref<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_aligned_ref<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_bands<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_bands<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_bands<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)(k1v3_bands<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k2_apply<int16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipFuncSetAttribute((const void*)k2_apply<int8_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes);
    (void)hipGetLastError();
    lap("the other attributes");
    *out = c;
    return KBBQ_OK;
}

int kbbq_ctx_destroy(kbbq_ctx* c)
{
    if (!c) return KBBQ_OK;
    (void)hipSetDevice(c->device);
    for (int w = 0; w < 2; ++w)
        for (auto& pr : c->ev[w]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_ops4) (void)hipFree(c->d_ops4);
    if (c->d_rowlut) (void)hipFree(c->d_rowlut);
    if (c->d_tally) (void)hipFree(c->d_tally);
    for (int b = 0; b < 2; ++b) {
        if (c->stage_host[b]) (void)hipHostFree(c->stage_host[b]);
        if (c->stage_dev[b]) (void)hipFree(c->stage_dev[b]);
        if (c->stage_up[b]) (void)hipEventDestroy(c->stage_up[b]);
        if (c->stage_used[b]) (void)hipEventDestroy(c->stage_used[b]);
        if (c->stage_down[b]) (void)hipEventDestroy(c->stage_down[b]);
    }
    if (c->stage_stream) (void)hipStreamDestroy(c->stage_stream);
    if (c->stage_down_stream) (void)hipStreamDestroy(c->stage_down_stream);
    if (c->d_wgplan) (void)hipFree(c->d_wgplan);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return KBBQ_OK;
}

int kbbq_ctx_set_stream(kbbq_ctx* c, void* s)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    c->stream = (hipStream_t)s;      // NULL is the device's default (null) stream, e.g. torch's
    return KBBQ_OK;
}

// for comm_rccl.cpp (not part of the public header)
void* kbbq_ctx_stream_(kbbq_ctx* c) { return c ? (void*)c->stream : nullptr; }
int kbbq_ctx_device_(kbbq_ctx* c) { return c ? c->device : 0; }

int kbbq_ctx_sync(kbbq_ctx* c)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KBBQ_OK;
}

int kbbq_ctx_info(kbbq_ctx* c, int* cus, int* lds, char* name, int name_len)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (cus) *cus = c->cus;
    if (lds) *lds = c->lds_bytes;
    if (name && name_len > 0) { strncpy(name, c->name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    return KBBQ_OK;
}

int kbbq_ctx_status(kbbq_ctx* c, int64_t* read_index)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    u64 st[KBBQ_NSTATUS];
    HIPCHK(hipMemcpyAsync(st, c->d_status, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (read_index) *read_index = -1;
    if (st[0] == ~0ull && st[1] == ~0ull && st[2] == ~0ull && st[3] == ~0ull && st[ST_MEANQ] == ~0ull) return KBBQ_OK;
    HIPCHK(hipMemcpyAsync(c->d_status, ST_INIT, sizeof ST_INIT, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the reference stops at the FIRST offending read; within one read the dinucleotide
    // lookup (TypeError, recalibrate.py:94) runs before the table indexing (IndexError, :114)
    if (st[0] == ~0ull && st[1] == ~0ull && st[2] == ~0ull && st[ST_MEANQ] != ~0ull) {
        if (read_index) *read_index = (int64_t)st[ST_MEANQ];
        return fail(KBBQ_E_MEANQ, "read group %lld: meanq lies on a truncation boundary (or the group has no counted base): "
                    "solve through kbbq_solve_dev with the host's longdouble meanq", (long long)st[ST_MEANQ]);
    }
    if (st[0] == ~0ull && st[1] == ~0ull && st[2] == ~0ull)
        return fail(KBBQ_E_LUT, "the device-built LUT can leave 0..255 or does not fit int8: re-run kbbq_apply_dev in checked mode");
    int code = KBBQ_E_TYPE; u64 best = st[ST_TYPE];
    if (st[ST_INDEX] < best) { best = st[ST_INDEX]; code = KBBQ_E_INDEX; }
    if (st[ST_RANGE] < best) { best = st[ST_RANGE]; code = KBBQ_E_RANGE; }
    if (read_index) *read_index = (int64_t)best;
    const char* what = code == KBBQ_E_TYPE ? "base outside ACGTN in a dinucleotide (reference: TypeError)"
                     : code == KBBQ_E_INDEX ? "quality/read-group/cycle beyond the tables (reference: IndexError)"
                                            : "recalibrated quality + 33 outside 0..255";
    return fail(code, "read %lld: %s", (long long)best, what);
}

// diagnostic builds only (-DK4_DEBUG_COUNT): the raw status words
int kbbq_debug_status_words_(kbbq_ctx* c, uint64_t* out8)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out8, c->d_status, sizeof ST_INIT, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyAsync(c->d_status, ST_INIT, sizeof ST_INIT, hipMemcpyHostToDevice, c->stream));
    return KBBQ_OK;
}

int kbbq_dev_alloc(kbbq_ctx* c, size_t bytes, void** dptr)
{
    if (!c || !dptr) return fail(KBBQ_E_ARG, "kbbq_dev_alloc: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 16));
    return KBBQ_OK;
}

int kbbq_dev_free(kbbq_ctx* c, void* dptr)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (dptr) HIPCHK(hipFree(dptr));
    return KBBQ_OK;
}

int kbbq_dev_zero(kbbq_ctx* c, void* dptr, size_t bytes)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemsetAsync(dptr, 0, bytes, c->stream));
    return KBBQ_OK;
}

int kbbq_dev_upload(kbbq_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KBBQ_OK;
}

int kbbq_dev_download(kbbq_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KBBQ_OK;
}

int kbbq_dev_mem_info(kbbq_ctx* c, size_t* free_bytes, size_t* total_bytes)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return KBBQ_OK;
}

int kbbq_host_alloc(size_t bytes, void** hptr)
{
    if (!hptr) return fail(KBBQ_E_ARG, "kbbq_host_alloc: NULL argument");
    HIPCHK(hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return KBBQ_OK;
}

int kbbq_host_free(void* hptr)
{
    if (hptr) HIPCHK(hipHostFree(hptr));
    return KBBQ_OK;
}

int kbbq_dev_copy_async(kbbq_ctx* c, void* dst, const void* src, size_t bytes, int kind)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (kind < 1 || kind > 3) return fail(KBBQ_E_ARG, "kbbq_dev_copy_async: kind must be 1 (host to device), 2 (device to host) or 3 (device to device)");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
    return KBBQ_OK;
}

int kbbq_event_create(kbbq_ctx* c, void** ev)
{
    if (!c || !ev) return fail(KBBQ_E_ARG, "kbbq_event_create: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = (void*)e;
    return KBBQ_OK;
}

int kbbq_event_record(kbbq_ctx* c, void* ev)
{
    if (!c || !ev) return fail(KBBQ_E_ARG, "kbbq_event_record: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord((hipEvent_t)ev, c->stream));
    return KBBQ_OK;
}

int kbbq_event_sync(void* ev)
{
    if (!ev) return fail(KBBQ_E_ARG, "kbbq_event_sync: NULL event");
    HIPCHK(hipEventSynchronize((hipEvent_t)ev));
    return KBBQ_OK;
}

int kbbq_event_destroy(void* ev)
{
    if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return KBBQ_OK;
}

size_t kbbq_tables_count(int R, int S2)
{
    return 2 * (size_t)R * KQ * (size_t)S2 + 2 * (size_t)R * KQ * KND;
}

int kbbq_lut_row_stride(int S2) { return lut_row_stride(S2); }

static size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }
static size_t lut_full_offset(int R, int Qt, int S2) { return align16((size_t)R * Qt * lut_row_stride(S2) * 2); }
size_t kbbq_full_lut_bytes(int R, int Qt, int S2) { return (size_t)R * (33 + Qt) * (size_t)full_lut_row_bytes(S2); }
static size_t lut_compact8_offset(int R, int Qt, int S2) { return lut_full_offset(R, Qt, S2) + align16(kbbq_full_lut_bytes(R, Qt, S2)); }
static size_t lut_compact8_bytes(int R, int Qt, int S2) { return (size_t)R * Qt * lut_row_stride(S2); }
static size_t lut_flags_offset(int R, int Qt, int S2) { return lut_compact8_offset(R, Qt, S2) + align16(lut_compact8_bytes(R, Qt, S2)); }
size_t kbbq_lut_bytes(int R, int Qt, int S2) { return lut_flags_offset(R, Qt, S2) + 16; }

size_t kbbq_lut_count(int R, int Qt, int S2)
{
    return (size_t)R * Qt * (size_t)lut_row_stride(S2);     // even: K2 stages the LUT as 32-bit words
}

int kbbq_ctx_timing(kbbq_ctx* c, int enable)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    c->timing = enable != 0;
    return KBBQ_OK;
}

int kbbq_ctx_kernel_ms(kbbq_ctx* c, int which, double* total_ms, int64_t* launches, int reset)
{
    if (!c || which < 0 || which > 1) return fail(KBBQ_E_ARG, "kbbq_ctx_kernel_ms: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto& pr : c->ev[which]) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->ms_acc[which] += ms;
        c->launches[which] += 1;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    c->ev[which].clear();
    if (total_ms) *total_ms = c->ms_acc[which];
    if (launches) *launches = c->launches[which];
    if (reset) { c->ms_acc[which] = 0.0; c->launches[which] = 0; }
    return KBBQ_OK;
}

} // extern "C"

// ---- launch helpers ---------------------------------------------------
static int check_planes(const char* fn, int64_t nreads, int pitch, const void* a, const void* b, const void* c)
{
    if (nreads < 0) return fail(KBBQ_E_ARG, "%s: nreads < 0", fn);
    if (pitch <= 0 || (pitch & 15) || pitch > 65536) return fail(KBBQ_E_ARG, "%s: pitch must be a positive multiple of 16 (<= 65536), got %d", fn, pitch);
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) return fail(KBBQ_E_ARG, "%s: planes must be 16-byte aligned", fn);
    return KBBQ_OK;
}

static u32 magic_for(int cpr) { return cpr <= 1 ? 0u : (u32)(((1ull << 32) + (u64)cpr - 1) / (u64)cpr); }

struct Timed {
    kbbq_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    Timed(kbbq_ctx* c_, int w) : c(c_), which(w)
    {
        if (c->timing && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, c->stream);
        else a = b = nullptr;
    }
    ~Timed()
    {
        if (a && b) { (void)hipEventRecord(b, c->stream); c->ev[which].push_back({a, b}); }
    }
};

extern "C" {

int kbbq_accumulate_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq,
                        const uint8_t* d_qual, const uint32_t* d_meta,
                        int64_t nreads, int pitch, int R, int S2, int minscore, int64_t* d_tables)
{
    return kbbq_accumulate_ex_dev(c, d_seq, d_cseq, d_qual, d_meta, nreads, pitch, R, S2,
                                  minscore, minscore, d_tables);
}

// what accumulate_rows would launch (kernel parameters, LDS bytes, template variant): kbbq_accumulate_bands_dev collects
// these for all length bands and launches them as ONE kernel
struct K1Setup { K1v3Params q; int dn = 0; bool km = false; bool split = false; size_t lds = 0; int64_t iters = 0; int threads = K1V3_THREADS; };
static int accumulate_rows(kbbq_ctx* c, const char* who, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                           const uint32_t* d_meta, int64_t nrows, int pitch, int pairs, int R, int S2, int minscore,
                           int dinuc_minscore, const int64_t* d_seg, int64_t* d_tables, bool* fits, int S_band = 0, int S_min = 0, int nib = 0,
                           int twins = 0, K1Setup* setup_only = nullptr);

int kbbq_accumulate_ex_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq,
                           const uint8_t* d_qual, const uint32_t* d_meta,
                           int64_t nreads, int pitch, int R, int S2, int minscore,
                           int dinuc_minscore, int64_t* d_tables)
{
    return kbbq_accumulate_band_dev(c, d_seq, d_cseq, d_qual, d_meta, nreads, pitch, R, S2, 0, 0, minscore, dinuc_minscore, d_tables);
}

int kbbq_accumulate_band_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq,
                             const uint8_t* d_qual, const uint32_t* d_meta,
                             int64_t nreads, int pitch, int R, int S2, int S_band, int S_min, int minscore,
                             int dinuc_minscore, int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_planes("kbbq_accumulate_dev", nreads, pitch, d_seq, d_cseq, d_qual);
    if (rc) return rc;
    if (R <= 0 || R > 32767) return fail(KBBQ_E_ARG, "kbbq_accumulate_dev: R out of range (%d)", R);
    if (S2 <= 0 || (S2 & 1)) return fail(KBBQ_E_ARG, "kbbq_accumulate_dev: S2 must be positive and even (%d)", S2);
    if (minscore < 0 || minscore > KQ - 1) return fail(KBBQ_E_ARG, "kbbq_accumulate_dev: minscore out of range (%d)", minscore);
    if (dinuc_minscore < 0 || dinuc_minscore > 222) return fail(KBBQ_E_ARG, "kbbq_accumulate_dev: dinuc_minscore out of range (%d)", dinuc_minscore);
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));

    const int64_t nblocks = (nreads + 63) / 64;
    const bool split = dinuc_minscore > minscore;
    const char* force = getenv("KBBQ_K1");              // "v1" forces the first kernel (A/B timing)
    if (!(force && !strcmp(force, "v1"))) {
        bool fits = false;
        rc = accumulate_rows(c, "kbbq_accumulate_dev", d_seq, d_cseq, d_qual, d_meta, nreads, pitch, 0, R, S2, minscore,
                             dinuc_minscore, nullptr, d_tables, &fits, S_band, S_min);
        if (rc || fits) return rc;                      // launched (or a real error); otherwise the tables do not fit the LDS
    }
    K1Params p;
    p.seq = d_seq; p.cseq = d_cseq; p.qual = d_qual; p.meta = d_meta;
    p.nreads = nreads; p.pitch = pitch; p.cpr = pitch / 16; p.cpr_magic = magic_for(p.cpr);
    p.R = R; p.S2 = S2; p.minscore = dinuc_minscore;   // the TypeError rule follows the dinucleotide threshold
    p.qlo = 33u + (u32)minscore; p.dlo = 33u + (u32)dinuc_minscore;
    p.pos_stride = S2 | 1;                         // odd row stride spreads q rows over the banks
    p.tables = reinterpret_cast<u64*>(d_tables); p.status = c->d_status;
    const size_t lds = ((size_t)(((KQ * p.pos_stride) + 1) & ~1)) * 4 + (size_t)KQ * KND * 8;
    if (lds > (size_t)c->lds_bytes)
        return fail(KBBQ_E_ARG, "kbbq_accumulate_dev: reads of %d bases need %zu B of LDS (> %d)", S2 / 2, lds, c->lds_bytes);
    int per_cu = std::min<int>((int)(c->lds_bytes / lds), 2048 / K1_THREADS);
    per_cu = std::max(per_cu, 1);
    const int64_t iters = (nblocks + (K1_THREADS / 64) - 1) / (K1_THREADS / 64);
    int gx = (int)std::min<int64_t>(iters, std::max(1, c->cus * per_cu / R));
    dim3 grid((unsigned)gx, (unsigned)R, 1), block(K1_THREADS, 1, 1);
    {
        Timed t(c, 0);
        if (split) hipLaunchKernelGGL(k1_accumulate<true>, grid, block, lds, c->stream, p);
        else hipLaunchKernelGGL(k1_accumulate<false>, grid, block, lds, c->stream, p);
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// host twin of k3_fill_full_lut (same rules; the GPU test compares the two blobs byte for byte)
static int fill_full_lut_host(const int16_t* lut16, int rs16, int R, int Qt, int S2, int minscore, int8_t* full,
                              int8_t* compact8)
{
    const int rb = full_lut_row_bytes(S2);
    const int NR = 33 + Qt;
    int bad = 0;
    for (int r = 0; r < R; ++r)
        for (int qb = 0; qb < NR; ++qb) {
            int8_t* row = full + ((size_t)r * NR + qb) * rb;
            for (int x = 0; x < rb; ++x) {
                const int W = full_lut_width(S2);
                int v = 0;
                if (qb < 33 + minscore) { if (x < 2 * W) v = qb == 0 ? -33 : qb - 33; }
                else {
                    const int16_t* src = lut16 + ((size_t)r * Qt + (qb - 33)) * rs16;
                    if (x < S2) v = src[x];
                    else if (x >= W && x < W + S2) v = src[S2 - 1 - (x - W)];
                    else if (x >= 2 * W && x < 2 * W + 25) v = src[S2 + (x - 2 * W)];
                }
                if (v < -128 || v > 127) bad |= 1;
                row[x] = (int8_t)v;
            }
        }
    for (size_t i = 0; i < (size_t)R * Qt * rs16; ++i) {
        if (lut16[i] < -128 || lut16[i] > 127) bad |= 1;
        compact8[i] = (int8_t)lut16[i];
    }
    for (int r = 0; r < R; ++r)
        for (int q = minscore; q < Qt; ++q) {
            const int16_t* src = lut16 + ((size_t)r * Qt + q) * rs16;
            int lo1 = 32767, hi1 = -32768, lo2 = 32767, hi2 = -32768;
            for (int x = 0; x < S2; ++x) { lo1 = std::min<int>(lo1, src[x]); hi1 = std::max<int>(hi1, src[x]); }
            for (int x = 0; x < 25; ++x) { lo2 = std::min<int>(lo2, src[S2 + x]); hi2 = std::max<int>(hi2, src[S2 + x]); }
            if (lo1 + lo2 + 33 < 0 || hi1 + hi2 + 33 > 255) bad |= 2;
        }
    return bad;
}

int kbbq_build_lut(int R, int Qt, int S2, int D, int minscore, const int64_t* meanq, const int64_t* rgdq,
                   const int64_t* qdq, const int64_t* posdq, const int64_t* dinucdq, void* blob,
                   int* flags_out)
{
    // D = 17 is what get_delta_qs returns (applybqsr.py:98-101); D = 16 is accepted because the
    // reference's own apply test passes an unpadded table (index -1 then aliases column 15)
    if (R <= 0 || Qt <= 0 || Qt > 95 || S2 <= 0 || D < 16 || D > 17 || minscore < 0)
        return fail(KBBQ_E_ARG, "kbbq_build_lut: bad shape R=%d Qt=%d S2=%d D=%d minscore=%d", R, Qt, S2, D, minscore);
    memset(blob, 0, kbbq_lut_bytes(R, Qt, S2));
    int16_t* out = reinterpret_cast<int16_t*>(blob);
    const int rs = lut_row_stride(S2);
    for (int r = 0; r < R; ++r)
        for (int q = 0; q < Qt; ++q) {
            const size_t cell = (size_t)r * Qt + q;
            int16_t* row = out + cell * rs;
            const int64_t base = meanq[r] + rgdq[r] + qdq[cell];
            for (int s = 0; s < S2; ++s) {
                const int64_t v = base + posdq[cell * S2 + s];
                if (v < -32768 || v > 32767) return fail(KBBQ_E_RANGE, "kbbq_build_lut: value %lld does not fit the LUT", (long long)v);
                row[s] = (int16_t)v;
            }
            // dinucleotide index 5 * prev + cur, codes A0 T1 G2 C3 and 4 = N / no previous base;
            // anything involving code 4 is the reference's index -1, i.e. column D - 1
            for (int a = 0; a < 5; ++a)
                for (int b = 0; b < 5; ++b) {
                    const int64_t v = (a < 4 && b < 4) ? dinucdq[cell * D + 4 * a + b] : dinucdq[cell * D + (D - 1)];
                    if (v < -32768 || v > 32767) return fail(KBBQ_E_RANGE, "kbbq_build_lut: value %lld does not fit the LUT", (long long)v);
                    row[S2 + 5 * a + b] = (int16_t)v;
                }
        }
    int8_t* full = reinterpret_cast<int8_t*>(blob) + lut_full_offset(R, Qt, S2);
    int8_t* compact8 = reinterpret_cast<int8_t*>(blob) + lut_compact8_offset(R, Qt, S2);
    const int flags = fill_full_lut_host(out, rs, R, Qt, S2, std::min(minscore, Qt), full, compact8);
    *reinterpret_cast<int*>(reinterpret_cast<char*>(blob) + lut_flags_offset(R, Qt, S2)) = flags;
    if (flags_out) *flags_out = flags;
    return KBBQ_OK;
}

// K2 on one-read-per-row planes: when the rows are narrower than the tables' S2 columns, the LUT of the columns they
// can reach (k3_fill_row_lut) in context-owned scratch, written on the launch stream in front of the apply kernel.
// Returns the cycle width Sb the LUT was built for (S2: the blob's own full LUT is used, *lut untouched).
static int narrowed_row_lut(kbbq_ctx* c, const void* d_lut_blob, int R, int Qt, int S2, int pitch, int minscore, const int8_t** lut)
{
    const char* off = getenv("KBBQ_K2_ROWLUT");                                    // "0": the full LUT (A/B timing)
    const int Sb = std::min(pitch, S2);
    if (Sb >= S2 || (off && !strcmp(off, "0"))) return S2;
    const size_t need = (size_t)R * (33 + Qt) * full_lut_row_bytes(Sb);
    if (c->rowlut_bytes < need) {
        if (c->d_rowlut) { if (hipStreamSynchronize(c->stream) != hipSuccess) return S2; (void)hipFree(c->d_rowlut); c->d_rowlut = nullptr; c->rowlut_bytes = 0; }
        if (hipMalloc(&c->d_rowlut, need) != hipSuccess) { c->d_rowlut = nullptr; return S2; }
        c->rowlut_bytes = need;
    }
    RowLutParams f;
    f.lut16 = reinterpret_cast<const int16_t*>(d_lut_blob); f.rs16 = lut_row_stride(S2); f.R = R; f.Qt = Qt; f.S2 = S2; f.Sb = Sb;
    f.minscore = std::min(std::max(minscore, 0), Qt);
    f.out = reinterpret_cast<int8_t*>(c->d_rowlut);
    hipLaunchKernelGGL(k3_fill_row_lut, dim3((unsigned)(R * (33 + Qt))), dim3(256), 0, c->stream, f);
    *lut = f.out;
    return Sb;
}

static int apply_rows(kbbq_ctx* c, const char* who, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta,
                      int64_t nrows, int pitch, int pairs, int R, int S2, int minscore, const void* d_lut_blob,
                      const void* d_pair_lut, const int64_t* d_seg, uint8_t* d_out, int nib = 0, const int64_t* d_perm = nullptr);

int kbbq_apply_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta,
                   int64_t nreads, int pitch, int R, int Qt, int S2, int minscore,
                   const void* d_lut, int mode, uint8_t* d_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_planes("kbbq_apply_dev", nreads, pitch, d_seq, d_qual, d_out);
    if (rc) return rc;
    if (R <= 0 || R > 32767 || Qt <= 0 || Qt > 95 || S2 <= 0 || S2 > 65536)
        return fail(KBBQ_E_ARG, "kbbq_apply_dev: bad table shape R=%d Qt=%d S2=%d (Qt <= 95)", R, Qt, S2);
    if (minscore < 0 || minscore > 222) return fail(KBBQ_E_ARG, "kbbq_apply_dev: minscore out of range");
    if ((uintptr_t)d_lut & 15) return fail(KBBQ_E_ARG, "kbbq_apply_dev: LUT must be 16-byte aligned");
    if (mode != KBBQ_APPLY_CHECKED && mode != KBBQ_APPLY_FAST) return fail(KBBQ_E_ARG, "kbbq_apply_dev: bad mode %d", mode);
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    const int64_t nblocks = (nreads + 63) / 64;
    const char* force = getenv("KBBQ_K2");              // "v1" forces the first kernel (A/B timing)

    // the LUT of the columns rows of this pitch can reach (narrowed_row_lut): a band of short reads under wide tables
    const int Sb0 = (getenv("KBBQ_K2_ROWLUT") && !strcmp(getenv("KBBQ_K2_ROWLUT"), "0")) ? S2 : std::min(pitch, S2);
    const size_t full_bytes = (size_t)R * (33 + Qt) * full_lut_row_bytes(Sb0);
    if (mode == KBBQ_APPLY_FAST && R == 1 && Qt == KQ && !(S2 & 1) && S2 <= 65534 && !(force && !strcmp(force, "v1"))) {
        // one read group, rows as a caller holds them (one character row per read): the short-lived kernel (kbbq_k2_tile.h) when
        // apply_rows' conditions for it hold -- it falls through to the persistent kernel itself otherwise.  What it cannot serve
        // (a foreign letter, q > 42, a read longer than the LUT) it reports as KBBQ_E_LUT: the caller's KBBQ_APPLY_CHECKED run
        // decides, as for a LUT that is not range-safe.
        const char* tile = getenv("KBBQ_K2_TILE");
        const char* tile_chars = getenv("KBBQ_K2_TILE_CHARS");
        const int cpr = pitch / 16;
        const size_t rg_bytes = (size_t)(33 + KQ) * full_lut_row_bytes(Sb0);
        if (!(tile && !strcmp(tile, "0")) && !(tile_chars && !strcmp(tile_chars, "0")) && cpr >= 2 && cpr <= 4096
            && rg_bytes <= (size_t)(getenv("KBBQ_K2_TILE_LUT_KB") ? atoi(getenv("KBBQ_K2_TILE_LUT_KB")) : 52) << 10)
            return apply_rows(c, "kbbq_apply_dev", d_seq, d_qual, d_meta, nreads, pitch, 0, R, S2, minscore, d_lut, nullptr, nullptr, d_out, 0, nullptr);
    }
    if (mode == KBBQ_APPLY_FAST && full_bytes <= (size_t)c->lds_bytes && !(force && !strcmp(force, "v1"))) {
        K2v3Params q;
        q.seq = d_seq; q.qual = d_qual; q.meta = d_meta; q.nreads = nreads; q.pitch = pitch;
        q.cpr = pitch / 16; q.cpr_magic = magic_for(q.cpr);
        q.R = R; q.Qt = Qt; q.S2 = S2; q.minscore = minscore; q.qlo = 33u + (u32)minscore;
        q.lut16 = reinterpret_cast<const int16_t*>(d_lut); q.rs16 = lut_row_stride(S2);
        q.full = reinterpret_cast<const int8_t*>(d_lut) + lut_full_offset(R, Qt, S2);
        const int Sb = narrowed_row_lut(c, d_lut, R, Qt, S2, pitch, minscore, &q.full);
        if (Sb != Sb0) return fail(KBBQ_E_HIP, "kbbq_apply_dev: no scratch for the narrowed LUT");
        q.full_bytes = (int)full_bytes;
        q.rb = (u32)full_lut_row_bytes(Sb); q.W = (u32)full_lut_width(Sb); q.ctx_off = 2u * q.W;
        q.maxlen = Sb; q.pairs = 0; q.seg = nullptr; q.rpb = 64; q.perm = nullptr;
        { const char* x = getenv("KBBQ_K2_PARTS"); q.parts = x ? atoi(x) : 8; }      // fronts of the persistent traversal (0 / 1: one)
        q.out = d_out; q.status = c->d_status;
#ifdef K2V3_PER_CU
        int per_cu = K2V3_PER_CU;
#else
        // resident workgroups per CU: LDS copies of the LUT, and the register allocation's waves per SIMD
        int per_cu = std::max(1, std::min<int>((int)(c->lds_bytes / full_bytes), (K2V3_WAVES * 4 * 64) / K2V3_THREADS));
#endif
        const int64_t want = (nblocks + (K2V3_THREADS / 64) - 1) / (K2V3_THREADS / 64);
        int gx = (int)std::min<int64_t>(want, (int64_t)c->cus * per_cu);
        {
            Timed t(c, 1);
            hipLaunchKernelGGL(k2v3_apply<false>, dim3((unsigned)std::max(gx, 1)), dim3(K2V3_THREADS), full_bytes, c->stream, q);
        }
        HIPCHK(hipGetLastError());
        return KBBQ_OK;
    }

    K2Params p;
    p.seq = d_seq; p.qual = d_qual; p.meta = d_meta; p.nreads = nreads; p.pitch = pitch;
    p.cpr = pitch / 16; p.cpr_magic = magic_for(p.cpr);
    p.R = R; p.Qt = Qt; p.S2 = S2; p.minscore = minscore; p.qlo = 33u + (u32)minscore;
    p.lut = reinterpret_cast<const int16_t*>(d_lut); p.lut_count = (int)kbbq_lut_count(R, Qt, S2);
    p.out = d_out; p.status = c->d_status;
    // FAST (flags == 0): every value fits int8 -> stage the one-byte copy; CHECKED: the int16 LUT.
    // Too big for LDS either way: per-base exact path from global memory.
    const bool fast = mode == KBBQ_APPLY_FAST;
    const size_t stage_bytes = fast ? lut_compact8_bytes(R, Qt, S2) : (size_t)p.lut_count * 2;
    const bool in_lds = stage_bytes + 16 <= (size_t)c->lds_bytes;
    p.lut_in_lds = in_lds ? 1 : 0;
    p.stage = fast ? (const void*)(reinterpret_cast<const char*>(d_lut) + lut_compact8_offset(R, Qt, S2)) : d_lut;
    p.stage_bytes = (int)stage_bytes;
    const size_t lds = in_lds ? ((stage_bytes + 15) & ~(size_t)15) : 0;
    // 256-thread workgroups while >= 4 of them fit a CU, else 1024-thread ones (same waves per CU)
    int threads = 256, per_cu = 8;
    if (in_lds) {
        per_cu = (int)(c->lds_bytes / lds);
        if (per_cu < 4) { threads = 1024; per_cu = std::min(per_cu, 2); }
        per_cu = std::max(1, std::min(per_cu, 2048 / threads));
    }
    const int64_t want = (nblocks + (threads / 64) - 1) / (threads / 64);
    int gx = (int)std::min<int64_t>(want, (int64_t)c->cus * per_cu);
    dim3 grid((unsigned)std::max(gx, 1), 1, 1), block((unsigned)threads, 1, 1);
    {
        Timed t(c, 1);
        if (!in_lds) hipLaunchKernelGGL((k2_apply<int16_t, false, true>), grid, block, 0, c->stream, p);
        else if (fast) hipLaunchKernelGGL((k2_apply<int8_t, true, false>), grid, block, lds, c->stream, p);
        else hipLaunchKernelGGL((k2_apply<int16_t, true, true>), grid, block, lds, c->stream, p);
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_synth_dev(kbbq_ctx* c, uint8_t* d_seq, uint8_t* d_cseq, uint8_t* d_qual, uint32_t* d_meta,
                   int64_t first, int64_t nreads, int64_t total, int pitch, uint64_t seed,
                   int len_lo, int len_hi, int nrg, int qlo, int qhi, const uint32_t* thr43)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_planes("kbbq_synth_dev", nreads, pitch, d_seq, d_cseq, d_qual);
    if (rc) return rc;
    if (len_lo < 0 || len_hi < len_lo || len_hi > pitch || nrg <= 0 || nrg > 32767 || qlo < 0 || qhi < qlo || qhi >= KQ || total <= 0)
        return fail(KBBQ_E_ARG, "kbbq_synth_dev: bad shape");
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    KSParams p;
    p.seq = d_seq; p.cseq = d_cseq; p.qual = d_qual; p.meta = d_meta;
    p.first = first; p.nreads = nreads; p.total = total; p.pitch = pitch; p.cpr = pitch / 16;
    p.seed = seed; p.len_lo = len_lo; p.len_hi = len_hi; p.nrg = nrg; p.qlo = qlo; p.qhi = qhi;
    memcpy(p.thr, thr43, sizeof p.thr);
    const int64_t nchunks = nreads * p.cpr;
    int gx = (int)std::min<int64_t>((nchunks + 255) / 256, (int64_t)c->cus * 8);
    hipLaunchKernelGGL(ks_synth, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// ---- K3: model solve ------------------------------------------------------
static int load_consts(SolveConsts& c, const double* h_consts)
{
    if (!h_consts) return fail(KBBQ_E_ARG, "model constants are NULL");
    memcpy(c.prior, h_consts, sizeof c.prior);
    memcpy(c.logp, h_consts + KSOLVE_NQ, sizeof c.logp);
    memcpy(c.log1mp, h_consts + 2 * KSOLVE_NQ, sizeof c.log1mp);
    return KBBQ_OK;
}

int kbbq_delta_q_dev(kbbq_ctx* c, const int64_t* d_prior_q, const int64_t* d_errs, const int64_t* d_total,
                     const double* d_comb, int64_t ncells, const double* h_consts129, int64_t* d_dq)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (ncells < 0) return fail(KBBQ_E_ARG, "kbbq_delta_q_dev: ncells < 0");
    if (ncells == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    K3CellParams p;
    p.prior_q = (const long long*)d_prior_q; p.errs = (const long long*)d_errs; p.total = (const long long*)d_total;
    p.comb = d_comb; p.n = ncells; p.dq = (long long*)d_dq;
    int rc = load_consts(p.c, h_consts129);
    if (rc) return rc;
    int gx = (int)std::min<int64_t>((ncells + 255) / 256, (int64_t)c->cus * 8);
    hipLaunchKernelGGL(k3_delta_q, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_posterior_q_dev(kbbq_ctx* c, const double* d_prior_q, const int64_t* d_errs, const int64_t* d_total,
                         const double* d_comb, int64_t ncells, const double* h_consts129, int64_t* d_post)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (ncells < 0) return fail(KBBQ_E_ARG, "kbbq_posterior_q_dev: ncells < 0");
    if (ncells == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    K3PostParams p;
    p.prior_q = d_prior_q; p.errs = (const long long*)d_errs; p.total = (const long long*)d_total;
    p.comb = d_comb; p.n = ncells; p.post = (long long*)d_post;
    int rc = load_consts(p.c, h_consts129);
    if (rc) return rc;
    int gx = (int)std::min<int64_t>((ncells + 255) / 256, (int64_t)c->cus * 8);
    hipLaunchKernelGGL(k3_posterior_q, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

size_t kbbq_solve_aux_count(int R, int S2) { return (size_t)R + (size_t)R * KQ + (size_t)R * KQ * S2 + (size_t)R * KQ * KND; }
size_t kbbq_solve_dq_count(int R, int S2) { return (size_t)R + (size_t)R * KQ + (size_t)R * KQ * S2 + (size_t)R * KQ * 17; }

static int launch_solve(kbbq_ctx* c, K3FusedParams& p, int minscore);

int kbbq_solve_dev(kbbq_ctx* c, const int64_t* d_tables, int R, int S2, int minscore, const int32_t* d_meanq,
                   const double* d_aux, const double* h_consts129, int32_t* d_post_q,
                   void* d_lut, int32_t* d_dq)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (R <= 0 || R > 32767 || S2 <= 0 || S2 > 65536) return fail(KBBQ_E_ARG, "kbbq_solve_dev: bad shape R=%d S2=%d", R, S2);
    if (!d_tables || !d_meanq || !d_aux || !d_post_q || !d_lut) return fail(KBBQ_E_ARG, "kbbq_solve_dev: NULL device pointer");
    HIPCHK(hipSetDevice(c->device));
    K3FusedParams p;
    p.tables = (const long long*)d_tables; p.R = R; p.S2 = S2; p.rs = lut_row_stride(S2);
    p.meanq = d_meanq; p.aux = d_aux; p.post_q = d_post_q; p.lut = reinterpret_cast<int16_t*>(d_lut); p.dq = d_dq;
    p.logtab = nullptr; p.meanq_out = nullptr; p.status = c->d_status;
    memset(p.perr, 0, sizeof p.perr);
    int rc = load_consts(p.c, h_consts129);
    if (rc) return rc;
    return launch_solve(c, p, minscore);
}

static int launch_solve(kbbq_ctx* c, K3FusedParams& p, int minscore)
{
    const int R = p.R, S2 = p.S2;
    void* d_lut = p.lut;
    hipLaunchKernelGGL(k3_levels_ab, dim3((unsigned)R), dim3(1024), 0, c->stream, p);
    const int64_t cells = (int64_t)R * KQ * ((int64_t)S2 + KND);              // one wave per cell
    int gx = (int)std::min<int64_t>((cells + 3) / 4, (int64_t)c->cus * 32);
    hipLaunchKernelGGL(k3_level_c, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    // derive the table-driven (int8) LUT and the flags the fast apply kernel relies on
    LutFillParams f;
    f.lut16 = p.lut; f.rs16 = p.rs; f.R = R; f.Qt = KQ; f.S2 = S2; f.minscore = std::min(std::max(minscore, 0), KQ);
    f.full = reinterpret_cast<int8_t*>(d_lut) + lut_full_offset(R, KQ, S2);
    f.compact8 = reinterpret_cast<int8_t*>(d_lut) + lut_compact8_offset(R, KQ, S2);
    f.flags = reinterpret_cast<int*>(reinterpret_cast<char*>(d_lut) + lut_flags_offset(R, KQ, S2));
    f.status = c->d_status;
    HIPCHK(hipMemsetAsync(f.flags, 0, 16, c->stream));
    hipLaunchKernelGGL(k3_fill_full_lut, dim3((unsigned)(R * (33 + KQ))), dim3(256), 0, c->stream, f);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_solve_device_dev(kbbq_ctx* c, const int64_t* d_tables, int R, int S2, int minscore, const double* d_logtab,
                          const double* h_consts172, int32_t* d_post_q, void* d_lut, int32_t* d_dq, int32_t* d_meanq_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (R <= 0 || R > 32767 || S2 <= 0 || S2 > 65536) return fail(KBBQ_E_ARG, "kbbq_solve_device_dev: bad shape R=%d S2=%d", R, S2);
    if (!d_tables || !d_logtab || !h_consts172 || !d_post_q || !d_lut) return fail(KBBQ_E_ARG, "kbbq_solve_device_dev: NULL pointer");
    HIPCHK(hipSetDevice(c->device));
    K3FusedParams p;
    p.tables = (const long long*)d_tables; p.R = R; p.S2 = S2; p.rs = lut_row_stride(S2);
    p.meanq = nullptr; p.aux = nullptr; p.post_q = d_post_q; p.lut = reinterpret_cast<int16_t*>(d_lut); p.dq = d_dq;
    p.logtab = d_logtab; p.meanq_out = d_meanq_out; p.status = c->d_status;
    memcpy(p.perr, h_consts172 + 3 * KSOLVE_NQ, sizeof p.perr);
    int rc = load_consts(p.c, h_consts172);
    if (rc) return rc;
    return launch_solve(c, p, minscore);
}

int kbbq_gammaln_dev(kbbq_ctx* c, const double* d_x, int64_t n, const double* d_logtab, double* d_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!d_x || !d_out)) || !d_logtab) return fail(KBBQ_E_ARG, "kbbq_gammaln_dev: bad argument");
    if (n == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    K3GammalnParams p; p.x = d_x; p.n = n; p.logtab = d_logtab; p.out = d_out;
    int gx = (int)std::min<int64_t>((n + 255) / 256, (int64_t)c->cus * 8);
    hipLaunchKernelGGL(k3_gammaln, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// workgroups of a streaming kernel with a grid-stride loop: `per_cu` per CU, or what the environment variable says (timing
// experiments; 0 = no bound, i.e. one work item per thread).  Measured in three interleaved rounds (scripts/gpu_grids.sh,
// DESIGN.md section 6): 64 per CU -- four generations of workgroups per CU slot -- beat the 8-16 these kernels
// started with by 6-13 % (K4 1.95 -> 1.83 ms, K5 1.12 -> 0.98, K6 2.66 -> 2.37 per 16 M reads), 256 did for the layout pass
// (9.46 -> 8.42 ms per 50 M reads); one item per thread loses again where a workgroup has set-up work (K4's queue, K5's
// LDS counters and their flush: 13 ms).
static int bounded_grid(int64_t want, const kbbq_ctx* c, int per_cu, const char* env)
{
    if (const char* v = getenv(env)) per_cu = atoi(v);
    const int64_t cap = per_cu > 0 ? (int64_t)c->cus * per_cu : (int64_t)0x7FFFFFFF;
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, cap));
}

// ---- mate-pair rows ---------------------------------------------------------
int kbbq_pair_pitch(int S2) { return pair_pitch(S2); }
size_t kbbq_pair_lut_bytes(int R, int Qt, int S2) { return (size_t)R * (33 + Qt) * pair_lut_row_bytes(S2); }

static int check_pairs(const char* who, int64_t npairs, int S2)
{
    if (npairs < 0) return fail(KBBQ_E_ARG, "%s: npairs < 0", who);
    if (S2 <= 0 || (S2 & 1) || S2 > 65534) return fail(KBBQ_E_ARG, "%s: S2 must be positive and even (%d)", who, S2);
    return KBBQ_OK;
}

int kbbq_pack_pairs_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                        const uint32_t* d_meta, int64_t npairs, int pitch, int S2,
                        uint8_t* d_pseq, uint8_t* d_pcseq, uint8_t* d_pqual, uint32_t* d_pmeta)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_pairs("kbbq_pack_pairs_dev", npairs, S2);
    if (rc) return rc;
    if (pitch <= 0 || (pitch & 15) || pitch < S2 / 2) return fail(KBBQ_E_ARG, "kbbq_pack_pairs_dev: bad pitch %d", pitch);
    if (!d_seq || !d_qual || !d_meta || !d_pseq || !d_pqual || !d_pmeta || (d_cseq && !d_pcseq))
        return fail(KBBQ_E_ARG, "kbbq_pack_pairs_dev: NULL pointer");
    if (((uintptr_t)d_pseq | (uintptr_t)d_pcseq | (uintptr_t)d_pqual) & 15) return fail(KBBQ_E_ARG, "kbbq_pack_pairs_dev: planes must be 16-byte aligned");
    if (npairs == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    PairPackParams p;
    p.src[0] = d_seq; p.src[1] = d_cseq; p.src[2] = d_qual;
    p.dst[0] = d_pseq; p.dst[1] = d_pcseq; p.dst[2] = d_pqual;
    p.fill[0] = 'N'; p.fill[1] = 'N'; p.fill[2] = 0;
    p.meta = d_meta; p.pmeta = d_pmeta; p.npairs = npairs; p.pitch = pitch; p.ppitch = pair_pitch(S2); p.S = S2 / 2; p.unpack = 0;
    const int64_t nchunks = npairs * (p.ppitch / 16);
    const int gx = bounded_grid((nchunks + 255) / 256, c, 256, "KBBQ_K7_GRID");
    hipLaunchKernelGGL(k7_pack_pairs, dim3((unsigned)gx), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_unpack_pairs_dev(kbbq_ctx* c, const uint8_t* d_pplane, int64_t npairs, int S2, int pitch, uint8_t* d_plane)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_pairs("kbbq_unpack_pairs_dev", npairs, S2);
    if (rc) return rc;
    if (pitch <= 0 || (pitch & 15) || pitch < S2 / 2) return fail(KBBQ_E_ARG, "kbbq_unpack_pairs_dev: bad pitch %d", pitch);
    if (!d_pplane || !d_plane || ((uintptr_t)d_plane & 15)) return fail(KBBQ_E_ARG, "kbbq_unpack_pairs_dev: bad pointer");
    if (npairs == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    PairPackParams p;
    p.src[0] = d_pplane; p.src[1] = nullptr; p.src[2] = nullptr; p.dst[0] = d_plane; p.dst[1] = nullptr; p.dst[2] = nullptr;
    p.fill[0] = p.fill[1] = p.fill[2] = 0;
    p.meta = nullptr; p.pmeta = nullptr; p.npairs = npairs; p.pitch = pitch; p.ppitch = pair_pitch(S2); p.S = S2 / 2; p.unpack = 1;
    const int64_t nchunks = 2 * npairs * (pitch / 16);
    const int gx = bounded_grid((nchunks + 255) / 256, c, 256, "KBBQ_K7_GRID");
    hipLaunchKernelGGL(k7_pack_pairs, dim3((unsigned)gx), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// K1 (table-driven kernel) on one-read-per-row or mate-pair rows, optionally grouped by read group
static int accumulate_rows(kbbq_ctx* c, const char* who, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                           const uint32_t* d_meta, int64_t nrows, int pitch, int pairs, int R, int S2, int minscore,
                           int dinuc_minscore, const int64_t* d_seg, int64_t* d_tables, bool* fits, int S_band, int S_min, int nib, int twins,
                           K1Setup* setup_only)
{
    if (fits) *fits = true;
    if (S_band < 0 || S_band > S2 / 2 || (pairs && S_band)) return fail(KBBQ_E_ARG, "%s: S_band out of range (%d)", who, S_band);
    if (S_min < 0 || S_min > (S_band ? S_band : S2 / 2)) return fail(KBBQ_E_ARG, "%s: S_min out of range (%d)", who, S_min);
    int rc = check_planes(who, nrows, pitch, d_seq, d_cseq, d_qual);
    if (rc) return rc;
    if (R <= 0 || R > 32767) return fail(KBBQ_E_ARG, "%s: R out of range (%d)", who, R);
    if (S2 <= 0 || (S2 & 1) || S2 > 65534) return fail(KBBQ_E_ARG, "%s: S2 must be positive and even (%d)", who, S2);
    if (minscore < 0 || minscore > KQ - 1) return fail(KBBQ_E_ARG, "%s: minscore out of range (%d)", who, minscore);
    if (dinuc_minscore < 0 || dinuc_minscore > 222) return fail(KBBQ_E_ARG, "%s: dinuc_minscore out of range (%d)", who, dinuc_minscore);
    if (pairs && pitch != pair_pitch(S2)) return fail(KBBQ_E_ARG, "%s: mate-pair rows of %d-base reads have pitch %d, not %d", who, S2 / 2, pair_pitch(S2), pitch);
    if (nrows == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    const int S = S_band ? S_band : S2 / 2;          // the LDS tables are laid out for the longest read of THIS batch
    K1v3Params q;
    q.aflags = nullptr; q.aclip = nullptr; q.atrim = nullptr;
    q.seq = d_seq; q.cseq = d_cseq; q.qual = d_qual; q.meta = d_meta;
    q.nreads = nrows; q.pitch = pitch; q.cpr = pitch / 16; q.cpr_magic = magic_for(q.cpr);
    q.R = R; q.S = S; q.gS2 = S2; q.minscore = minscore; q.type_minscore = dinuc_minscore;
    q.qlo_m1 = 32u + (u32)minscore; q.dlo = 33u + (u32)dinuc_minscore;
    q.nrows = KQ + 1 - minscore;
    q.maxlen = pairs ? S2 + 1 : S; q.gap = pairs ? 1 : 0; q.twins = (pairs && twins) ? 1 : 0;
    q.seg = reinterpret_cast<const long long*>(d_seg);
    q.tables = reinterpret_cast<u64*>(d_tables); q.status = c->d_status;
    q.genome = nullptr; q.ag0 = nullptr;
    // LDS geometry.  A cycle row has one word per first-in-pair position [0, S) and per second-in-pair index
    // S + 2(S - len) + pos <= 3S - len - 1: 3S words serve any length.  When that does not fit (reads of ~200 bases
    // and more), a band whose shortest read is S_min needs only 3S - S_min words, and 8 copies of the context table
    // instead of 16 free the rest: reads of up to ~300 bases still run this kernel.
    const int trim = (!pairs && S_min > 0) ? std::min(S_min, S) : 0;
    int dn = 0; size_t lds3 = 0;
    // mate-pair rows of 19 or 13 chunks (2 x 150, 2 x 100 bp) on 4-bit planes: the chunk-position-major cycle table (kernel comment), the
    // 13-chunk form with several trash rows (K1v3Params::ntrash; 19 chunks were measured with 4 and 8 of them: nothing; KBBQ_K1_KM_NTRASH
    // overrides).  Measured for the other lengths instruments emit (scripts/time_km_widths.py): 2 x 100 bp 1.37 -> 1.31 ms per 3 Gbases; 2 x 50 bp
    // (7 chunks) is FASTER position-major with its copies of the cycle table (1.52 against 1.64 ms); 2 x 75 / 2 x 125 / 2 x 250 bp never
    // take mate-pair rows (2 S + 1 rounds up to twice the single read's pitch: no bytes saved).  KBBQ_K1_KM=0: none, =19: only 19 chunks.
    const char* km_off = getenv("KBBQ_K1_KM");
    const bool km_width = q.cpr == 19 || (q.cpr == 13 && !(km_off && !strcmp(km_off, "19")));
    const bool km = pairs && nib && km_width && !(km_off && !strcmp(km_off, "0"));
    bool km_ok = false; int km_ntrash = 1;                 // the form applies AND its tables fit
    if (km) {
        q.row_bytes = (u32)(((16 * q.cpr + 31) & ~31) * 4);
        q.minlen = 0; q.slack_bytes = 0; q.ntrash = 1; q.pos_copies = 1;
        q.dn_flush_iters = std::max(1, 65535 / ((K1V3_THREADS / K1V3_DNREP) * 16 * q.cpr));
        const char* kt = getenv("KBBQ_K1_KM_NTRASH");
        int want_t = kt ? atoi(kt) : (q.cpr == 19 ? 1 : 8);
        for (int nt = 8; nt >= 1; nt >>= 1) {
            if (nt > want_t) continue;
            const size_t rows = (size_t)q.nrows - 1 + nt;
            lds3 = rows * 128 * K1V3_DNREP + rows * q.row_bytes;
            if (lds3 <= (size_t)c->lds_bytes) { dn = K1V3_DNREP; km_ok = true; km_ntrash = nt; break; }
        }
    }
    // An experiment kept behind KBBQ_K1_TWO=1 (round 3; measured, not adopted): narrow rows (the short bands of a mixed-length
    // input) run K1 at half the rate of wide ones -- a 64-row block is a few steps of work behind two dependent loads -- and
    // their tables are small: with 8 copies of the context table TWO workgroups of 12 waves fit a CU's LDS and registers (24
    // waves per CU instead of 16; two 16-wave workgroups never fitted: 72 registers allow 7 waves per SIMD).  Config 5's K1:
    // 1.80-1.85 ms against 1.74-1.75 merged, 1.97 against 1.90 a launch per band -- the halved copies cost more than the waves bring.
    int threads = K1V3_THREADS, per_cu = 1;
    {
        const char* two_env = getenv("KBBQ_K1_TWO");
        const int cut = trim, words = (3 * S - cut) | 1;
        const size_t lds_two = (size_t)q.nrows * 128 * 8 + (size_t)q.nrows * words * 4 + (size_t)(S + 32) * 4;
        if (!km && !pairs && two_env && !strcmp(two_env, "1") && 2 * (lds_two + 1024) <= (size_t)c->lds_bytes
            && (768 / 8) * 16 * q.cpr <= 65535) {
            q.row_bytes = (u32)words * 4u; q.minlen = cut; q.slack_bytes = (u32)(S + 32) * 4u;
            q.dn_flush_iters = std::max(1, 65535 / ((768 / 8) * 16 * q.cpr));
            lds3 = lds_two; dn = 8; threads = 768; per_cu = 2;
        }
    }
    for (int attempt = 0; attempt < (trim ? 3 : 1) && !dn; ++attempt) {
        const int copies = attempt == 2 ? 8 : K1V3_DNREP, cut = attempt ? trim : 0;
        const int words = (3 * S - cut) | 1;
        q.row_bytes = (u32)words * 4u;
        q.minlen = cut;
        // bytes past a read's end (quality 0) land on the trash row, the last one, at indexes up to 3S - cut - 1 + 15
        q.slack_bytes = (u32)(S + 32) * 4u;
        q.dn_flush_iters = std::max(1, 65535 / ((K1V3_THREADS / copies) * 16 * q.cpr));
        lds3 = (size_t)q.nrows * 128 * copies + (size_t)q.nrows * q.row_bytes + q.slack_bytes;
        if (lds3 <= (size_t)c->lds_bytes && (K1V3_THREADS / copies) * 16 * q.cpr <= 65535) dn = copies;
    }
    if (!dn) {
        if (fits) { *fits = false; return KBBQ_OK; }     // the caller has another kernel for this shape
        return fail(KBBQ_E_LUT, "%s: %d-base reads with minscore %d do not fit the LDS tables; use plain one-read-per-row planes", who, S, minscore);
    }
    // copies of the cycle table for narrow rows (kernel comment at K1v3Params::pos_copies): as many of 4 / 2 as the LDS holds beside
    // the context table, for rows of up to 12 chunks (wider rows spread a wave's lanes over enough columns); KBBQ_K1_POSCOPIES=1: none
    // ... and several trash rows (K1v3Params::ntrash) first: the padding behind every read's last base is the hotter spot
    q.pos_copies = 1; q.ntrash = 1;
    q.pos_copy_bytes = (u32)q.nrows * q.row_bytes + q.slack_bytes;
    if (km_ok) { q.ntrash = km_ntrash; q.pos_copy_bytes = (u32)((size_t)(q.nrows - 1 + q.ntrash) * q.row_bytes); }
    if (!km_ok && per_cu == 1) {
        const char* nt_env = getenv("KBBQ_K1_NTRASH");
        const int most_t = nt_env ? atoi(nt_env) : 8;
        auto bytes_for = [&](int nt, int pc) {
            const size_t rows = (size_t)q.nrows - 1 + nt;
            return rows * 128 * dn + (size_t)pc * (rows * q.row_bytes + q.slack_bytes);
        };
        for (int nt = 8; nt >= 2; nt >>= 1)
            if (nt <= most_t && bytes_for(nt, 1) <= (size_t)c->lds_bytes) { q.ntrash = nt; break; }
        const char* pc_env = getenv("KBBQ_K1_POSCOPIES");
        const int most = pc_env ? atoi(pc_env) : 4;
        if (q.cpr <= 12)
            for (int pc = 4; pc >= 2; pc >>= 1)
                if (pc <= most && bytes_for(q.ntrash, pc) <= (size_t)c->lds_bytes) { q.pos_copies = pc; break; }
        q.pos_copy_bytes = (u32)((size_t)(q.nrows - 1 + q.ntrash) * q.row_bytes + q.slack_bytes);
        lds3 = bytes_for(q.ntrash, q.pos_copies);
    }
    const bool split = dinuc_minscore > minscore;
    const int64_t nblocks = (nrows + 63) / 64;
    const int64_t iters = (nblocks + (threads / 64) - 1) / (threads / 64);
    if (setup_only) {
        setup_only->q = q; setup_only->dn = dn; setup_only->km = km_ok; setup_only->split = split;
        setup_only->lds = lds3; setup_only->iters = iters; setup_only->threads = threads;
        return KBBQ_OK;
    }
    int gx = (int)std::min<int64_t>(iters, std::max(1, c->cus * per_cu / R));
    dim3 grid((unsigned)gx, (unsigned)R, 1), block((unsigned)threads, 1, 1);
    {
        Timed t(c, 0);
        if (km_ok) {
#define KBBQ_KM_LAUNCH(KJ_) case KJ_: \
                if (split) hipLaunchKernelGGL((k1v3_accumulate<true, K1V3_DNREP, true, KJ_>), grid, block, lds3, c->stream, q); \
                else hipLaunchKernelGGL((k1v3_accumulate<false, K1V3_DNREP, true, KJ_>), grid, block, lds3, c->stream, q); \
                break;
            switch (q.cpr) { KBBQ_KM_LAUNCH(13) KBBQ_KM_LAUNCH(19) }
#undef KBBQ_KM_LAUNCH
        } else if (nib) {
            if (dn == K1V3_DNREP) {
                if (split) hipLaunchKernelGGL((k1v3_accumulate<true, K1V3_DNREP, true>), grid, block, lds3, c->stream, q);
                else hipLaunchKernelGGL((k1v3_accumulate<false, K1V3_DNREP, true>), grid, block, lds3, c->stream, q);
            } else {
                if (split) hipLaunchKernelGGL((k1v3_accumulate<true, 8, true>), grid, block, lds3, c->stream, q);
                else hipLaunchKernelGGL((k1v3_accumulate<false, 8, true>), grid, block, lds3, c->stream, q);
            }
        } else if (dn == K1V3_DNREP) {
            if (split) hipLaunchKernelGGL((k1v3_accumulate<true, K1V3_DNREP>), grid, block, lds3, c->stream, q);
            else hipLaunchKernelGGL((k1v3_accumulate<false, K1V3_DNREP>), grid, block, lds3, c->stream, q);
        } else {
            if (split) hipLaunchKernelGGL((k1v3_accumulate<true, 8>), grid, block, lds3, c->stream, q);
            else hipLaunchKernelGGL((k1v3_accumulate<false, 8>), grid, block, lds3, c->stream, q);
        }
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_accumulate_pairs_dev(kbbq_ctx* c, const uint8_t* d_pseq, const uint8_t* d_pcseq, const uint8_t* d_pqual,
                              const uint32_t* d_pmeta, int64_t npairs, int R, int S2, int minscore,
                              int dinuc_minscore, int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_pairs("kbbq_accumulate_pairs_dev", npairs, S2);
    if (rc) return rc;
    return accumulate_rows(c, "kbbq_accumulate_pairs_dev", d_pseq, d_pcseq, d_pqual, d_pmeta, npairs, pair_pitch(S2), 1,
                           R, S2, minscore, dinuc_minscore, nullptr, d_tables, nullptr);
}

int kbbq_accumulate_grouped_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                                const uint32_t* d_meta, int64_t nrows, int pitch, int pairs, int R, int S2, int S_band,
                                int S_min, int minscore, int dinuc_minscore, const int64_t* d_seg, int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (!d_seg) return fail(KBBQ_E_ARG, "kbbq_accumulate_grouped_dev: d_seg is NULL");
    return accumulate_rows(c, "kbbq_accumulate_grouped_dev", d_seq, d_cseq, d_qual, d_meta, nrows, pitch, pairs ? 1 : 0,
                           R, S2, minscore, dinuc_minscore, d_seg, d_tables, nullptr, S_band, S_min);
}

int kbbq_pair_lut_dev(kbbq_ctx* c, const void* d_lut_blob, int R, int S2, int minscore, void* d_pair_lut)
{
    return kbbq_pair_lut_rows_dev(c, d_lut_blob, R, S2, minscore, KBBQ_ROWS_PAIRS, d_pair_lut);
}

int kbbq_pair_lut_rows_dev(kbbq_ctx* c, const void* d_lut_blob, int R, int S2, int minscore, int flags, void* d_pair_lut)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_pairs("kbbq_pair_lut_dev", 0, S2);
    if (rc) return rc;
    if (!(flags & KBBQ_ROWS_PAIRS) || (flags & ~(KBBQ_ROWS_PAIRS | KBBQ_ROWS_NIBBLES | KBBQ_ROWS_TWINS)))
        return fail(KBBQ_E_ARG, "kbbq_pair_lut_rows_dev: flags 0x%x are not those of mate-pair rows", flags);
    if (R <= 0 || R > 32767 || !d_lut_blob || !d_pair_lut) return fail(KBBQ_E_ARG, "kbbq_pair_lut_dev: bad argument");
    HIPCHK(hipSetDevice(c->device));
    PairLutParams f;
    f.lut16 = reinterpret_cast<const int16_t*>(d_lut_blob); f.rs16 = lut_row_stride(S2); f.R = R; f.Qt = KQ; f.S2 = S2;
    f.minscore = std::min(std::max(minscore, 0), KQ);
    f.twins = (flags & KBBQ_ROWS_TWINS) ? 1 : 0;
    f.out = reinterpret_cast<int8_t*>(d_pair_lut);
    hipLaunchKernelGGL(k3_fill_pair_lut, dim3((unsigned)(R * (33 + KQ))), dim3(256), 0, c->stream, f);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// K2 (table-driven kernel) on one-read-per-row or mate-pair rows, optionally grouped by read group
static int apply_rows(kbbq_ctx* c, const char* who, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta,
                      int64_t nrows, int pitch, int pairs, int R, int S2, int minscore, const void* d_lut_blob,
                      const void* d_pair_lut, const int64_t* d_seg, uint8_t* d_out, int nib, const int64_t* d_perm)
{
    int rc = check_planes(who, nrows, pitch, d_seq, d_qual, d_out);
    if (rc) return rc;
    if (R <= 0 || R > 32767 || !d_lut_blob || (pairs && (!d_pair_lut || ((uintptr_t)d_pair_lut & 15)))) return fail(KBBQ_E_ARG, "%s: bad argument", who);
    if (S2 <= 0 || (S2 & 1) || S2 > 65534) return fail(KBBQ_E_ARG, "%s: S2 must be positive and even (%d)", who, S2);
    if (minscore < 0 || minscore > 222) return fail(KBBQ_E_ARG, "%s: minscore out of range", who);
    if (pairs && pitch != pair_pitch(S2)) return fail(KBBQ_E_ARG, "%s: mate-pair rows of %d-base reads have pitch %d, not %d", who, S2 / 2, pair_pitch(S2), pitch);
    K2v3Params q;
    // one read per row: the LUT of the columns rows of this pitch can reach (narrowed_row_lut)
    const int Sb = pairs || (getenv("KBBQ_K2_ROWLUT") && !strcmp(getenv("KBBQ_K2_ROWLUT"), "0")) ? S2 : std::min(pitch, S2);
    q.rb = (u32)(pairs ? pair_lut_row_bytes(S2) : full_lut_row_bytes(Sb));
    const size_t rg_bytes = (size_t)(33 + KQ) * q.rb;
    const size_t all_bytes = rg_bytes * (size_t)R;
    const size_t lds = d_seg ? rg_bytes : all_bytes;
    if (lds > (size_t)c->lds_bytes)
        return fail(KBBQ_E_LUT, "%s: the apply LUT (%zu B) does not fit the LDS; group the rows by read group or use kbbq_apply_dev", who, lds);
    if (nrows == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    q.seq = d_seq; q.qual = d_qual; q.meta = d_meta; q.nreads = nrows; q.pitch = pitch;
    q.cpr = pitch / 16; q.cpr_magic = magic_for(q.cpr);
    q.R = R; q.Qt = KQ; q.S2 = S2; q.minscore = minscore; q.qlo = 33u + (u32)minscore;
    q.lut16 = reinterpret_cast<const int16_t*>(d_lut_blob); q.rs16 = lut_row_stride(S2);
    q.full = pairs ? reinterpret_cast<const int8_t*>(d_pair_lut)
                   : reinterpret_cast<const int8_t*>(d_lut_blob) + lut_full_offset(R, KQ, S2);
    q.full_bytes = (int)all_bytes;
    if (pairs) { q.W = 0u; q.ctx_off = (u32)pair_pitch(S2); q.maxlen = S2 + 1; }
    else {
        if (narrowed_row_lut(c, d_lut_blob, R, KQ, S2, pitch, minscore, &q.full) != Sb) return fail(KBBQ_E_HIP, "%s: no scratch for the narrowed LUT", who);
        q.W = (u32)full_lut_width(Sb); q.ctx_off = 2u * q.W; q.maxlen = Sb;
    }
    q.pairs = pairs; q.seg = reinterpret_cast<const long long*>(d_seg);
    q.perm = reinterpret_cast<const long long*>(d_perm);
    q.out = d_out; q.status = c->d_status;
    int per_cu = std::max(1, std::min<int>((int)(c->lds_bytes / lds), (K2V3_WAVES * 4 * 64) / K2V3_THREADS));
    { const char* x = getenv("KBBQ_K2_PARTS"); q.parts = x ? atoi(x) : 8; }      // fronts of the persistent traversal (0 / 1: one)
    q.rpb = 64;      // the persistent kernel's wave block; smaller blocks were measured slower (profiles/r01_traversal_microbench.md)
    const int64_t nblocks = (nrows + q.rpb - 1) / q.rpb;
    const int64_t want = (nblocks + (K2V3_THREADS / 64) - 1) / (K2V3_THREADS / 64);
    const int slices = d_seg ? R : 1;
    int gx = (int)std::min<int64_t>(want, std::max<int64_t>(1, (int64_t)c->cus * per_cu / slices));

    // mate-pair rows on 4-bit planes: short-lived workgroups (kbbq_k2_tile.h) unless KBBQ_K2_TILE=0 (A/B timing); everything
    // else, and LUTs of one group beyond a third of the LDS, keeps the persistent kernel
    const char* tile = getenv("KBBQ_K2_TILE");
    // (one read per row: while two workgroups' copies of the narrowed LUT fit a CU -- rows of up to ~300 bases; measured up to
    //  there, scripts/gpu_tilekb.sh: config 5's K2 0.651 -> 0.687 of roofline going from a 32 KB to a 52 KB limit)
    // character planes (round 4; KBBQ_K2_TILE_CHARS=0: the persistent kernel as before): the same kernel with 16-byte sequence loads
    const char* tile_chars = getenv("KBBQ_K2_TILE_CHARS");
    const bool planes_ok = nib || !(tile_chars && !strcmp(tile_chars, "0"));
    if (!(tile && !strcmp(tile, "0")) && planes_ok && (R == 1 || d_seg) && q.cpr >= 2 && q.cpr <= 4096
        && (pairs ? rg_bytes * 3 <= (size_t)c->lds_bytes : rg_bytes <= (size_t)(getenv("KBBQ_K2_TILE_LUT_KB") ? atoi(getenv("KBBQ_K2_TILE_LUT_KB")) : 52) << 10)) {
        K2tParams t;
        t.seq = d_seq; t.qual = d_qual; t.meta = d_meta; t.nchunks = nrows * q.cpr; t.cpr = q.cpr; t.cpr_magic = q.cpr_magic;
        t.Qt = KQ; t.S2 = S2; t.maxlen = q.maxlen; t.lut = q.full; t.lut_bytes = (int)((rg_bytes + 15) & ~(size_t)15);
        t.rb = q.rb; t.ctx_off = q.ctx_off; t.W = q.W; t.seg = reinterpret_cast<const long long*>(d_seg); t.R = R; t.wg_start = nullptr; t.order = nullptr;
        t.perm = reinterpret_cast<const long long*>(d_perm); t.pitch = pitch; t.out = d_out; t.status = c->d_status;
        { const char* x = getenv("KBBQ_K2_XCD_TILES"); t.xcd_tiles = x ? std::max(0, atoi(x)) : 1; }
        const int64_t per_wg = (int64_t)(K2T_THREADS / 64) * 64 * K2T_STEPS;
        int64_t gt = (t.nchunks + per_wg - 1) / per_wg;
        if (d_seg) {
            gt += R;                                     // every group rounds its last workgroup up
            const char* ord = getenv("KBBQ_K2_ORDER");  // "0": group after group (A/B timing)
            const bool interleave = d_perm && R > 1 && !(ord && !strcmp(ord, "0"));
            const int64_t ints = (int64_t)(R + 1) + 1 + (interleave ? 2 * gt : 0);
            if (ints > 0x7FFFFFFF) return fail(KBBQ_E_ARG, "%s: too many workgroups", who);
            if (c->wgplan_n < (int)ints) {
                if (c->d_wgplan) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_wgplan); c->d_wgplan = nullptr; c->wgplan_n = 0; }
                HIPCHK(hipMalloc((void**)&c->d_wgplan, (size_t)ints * sizeof(int)));
                c->wgplan_n = (int)ints;
            }
            K2tPlanParams pl; pl.seg = t.seg; pl.R = R; pl.cpr = q.cpr; pl.wg_start = c->d_wgplan;
            hipLaunchKernelGGL(k2t_plan, dim3(1), dim3(64), 0, c->stream, pl);
            t.wg_start = c->d_wgplan;
            if (interleave) {
                K2tOrderParams op; op.wg_start = c->d_wgplan; op.R = R;
                { const char* x = getenv("KBBQ_K2_XCD"); op.xcd = x ? atoi(x) : 8; }      // "0": ranks as they are (A/B timing)
                op.order = reinterpret_cast<int2*>(c->d_wgplan + ((R + 2) & ~1));           // 8-byte aligned behind the starts
                hipLaunchKernelGGL(k2t_order, dim3((unsigned)((gt + 255) / 256)), dim3(256), 0, c->stream, op);
                t.order = op.order;
            }
        }
        {
            Timed tm(c, 1);
            if (nib) hipLaunchKernelGGL(k2t_apply<true>, dim3((unsigned)gt), dim3(K2T_THREADS), (size_t)t.lut_bytes, c->stream, t);
            else hipLaunchKernelGGL(k2t_apply<false>, dim3((unsigned)gt), dim3(K2T_THREADS), (size_t)t.lut_bytes, c->stream, t);
        }
        HIPCHK(hipGetLastError());
        return KBBQ_OK;
    }
    {
        Timed t(c, 1);
        if (nib) hipLaunchKernelGGL(k2v3_apply<true>, dim3((unsigned)std::max(gx, 1), (unsigned)slices, 1), dim3(K2V3_THREADS), lds, c->stream, q);
        else hipLaunchKernelGGL(k2v3_apply<false>, dim3((unsigned)std::max(gx, 1), (unsigned)slices, 1), dim3(K2V3_THREADS), lds, c->stream, q);
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_apply_pairs_dev(kbbq_ctx* c, const uint8_t* d_pseq, const uint8_t* d_pqual, const uint32_t* d_pmeta,
                         int64_t npairs, int R, int S2, int minscore, const void* d_lut_blob, const void* d_pair_lut,
                         uint8_t* d_pout)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_pairs("kbbq_apply_pairs_dev", npairs, S2);
    if (rc) return rc;
    return apply_rows(c, "kbbq_apply_pairs_dev", d_pseq, d_pqual, d_pmeta, npairs, pair_pitch(S2), 1, R, S2, minscore,
                      d_lut_blob, d_pair_lut, nullptr, d_pout);
}

int kbbq_apply_grouped_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta,
                           int64_t nrows, int pitch, int pairs, int R, int S2, int minscore, const void* d_lut_blob,
                           const void* d_pair_lut, const int64_t* d_seg, uint8_t* d_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (!d_seg) return fail(KBBQ_E_ARG, "kbbq_apply_grouped_dev: d_seg is NULL");
    return apply_rows(c, "kbbq_apply_grouped_dev", d_seq, d_qual, d_meta, nrows, pitch, pairs ? 1 : 0, R, S2, minscore,
                      d_lut_blob, d_pair_lut, d_seg, d_out);
}


// ---- layouts: rows as handed over -> what K1 / K2 run fastest on ------------
static int layout_flags_ok(const char* who, int flags)
{
    if (flags & ~(KBBQ_ROWS_PAIRS | KBBQ_ROWS_NIBBLES | KBBQ_ROWS_TWINS)) return fail(KBBQ_E_ARG, "%s: unknown layout flags 0x%x", who, flags);
    if ((flags & KBBQ_ROWS_TWINS) && !(flags & KBBQ_ROWS_PAIRS)) return fail(KBBQ_E_ARG, "%s: KBBQ_ROWS_TWINS describes mate-pair rows (add KBBQ_ROWS_PAIRS)", who);
    return KBBQ_OK;
}

int kbbq_accumulate_rows_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                             const uint32_t* d_meta, int64_t nrows, int pitch, int flags, int R, int S2, int S_band,
                             int S_min, int minscore, int dinuc_minscore, const int64_t* d_seg, int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = layout_flags_ok("kbbq_accumulate_rows_dev", flags);
    if (rc) return rc;
    const int pairs = (flags & KBBQ_ROWS_PAIRS) ? 1 : 0;
    if (pairs) { rc = check_pairs("kbbq_accumulate_rows_dev", nrows, S2); if (rc) return rc; }
    return accumulate_rows(c, "kbbq_accumulate_rows_dev", d_seq, d_cseq, d_qual, d_meta, nrows, pitch, pairs, R, S2, minscore,
                           dinuc_minscore, d_seg, d_tables, nullptr, pairs ? 0 : S_band, pairs ? 0 : S_min,
                           (flags & KBBQ_ROWS_NIBBLES) ? 1 : 0, (flags & KBBQ_ROWS_TWINS) ? 1 : 0);
}

int kbbq_apply_rows_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta, int64_t nrows,
                        int pitch, int flags, int R, int S2, int minscore, const void* d_lut_blob, const void* d_pair_lut,
                        const int64_t* d_seg, const int64_t* d_perm, uint8_t* d_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = layout_flags_ok("kbbq_apply_rows_dev", flags);
    if (rc) return rc;
    const int pairs = (flags & KBBQ_ROWS_PAIRS) ? 1 : 0;
    if (pairs) { rc = check_pairs("kbbq_apply_rows_dev", nrows, S2); if (rc) return rc; }
    return apply_rows(c, "kbbq_apply_rows_dev", d_seq, d_qual, d_meta, nrows, pitch, pairs, R, S2, minscore, d_lut_blob,
                      d_pair_lut, d_seg, d_out, (flags & KBBQ_ROWS_NIBBLES) ? 1 : 0, d_perm);
}

// ---- all length bands of a mixed-length input in ONE launch per kernel -------------------------------------------
static int band_accumulate_alone(kbbq_ctx* c, const kbbq_band& b, int R, int S2, int minscore, int dinuc_minscore, int64_t* d_tables)
{
    if (b.flags || b.d_seg)
        return kbbq_accumulate_rows_dev(c, b.d_seq, b.d_cseq, b.d_qual, b.d_meta, b.nrows, b.pitch, b.flags, R, S2, b.S_band, b.S_min,
                                        minscore, dinuc_minscore, b.d_seg, d_tables);
    return kbbq_accumulate_band_dev(c, b.d_seq, b.d_cseq, b.d_qual, b.d_meta, b.nrows, b.pitch, R, S2, b.S_band, b.S_min, minscore,
                                    dinuc_minscore, d_tables);
}

// workgroups for every band of a merged launch: one each, the rest of `total` by largest remainder of weight / sum(weight),
// never more than a band can use (`cap`)
static void share_workgroups(const std::vector<double>& weight, const std::vector<int64_t>& cap, int total, std::vector<int>& out)
{
    const size_t n = weight.size();
    out.assign(n, 1);
    int left = total - (int)n;
    double sum = 0; for (double w : weight) sum += w;
    for (int round = 0; round < 4 && left > 0 && sum > 0; ++round) {            // a capped band's surplus goes round again
        std::vector<std::pair<double, size_t>> frac;
        int given = 0; double open_sum = 0;
        for (size_t i = 0; i < n; ++i) if (out[i] < cap[i]) open_sum += weight[i];
        if (open_sum <= 0) break;
        const int pool = left;
        for (size_t i = 0; i < n; ++i) {
            if (out[i] >= cap[i]) continue;
            const double want = pool * weight[i] / open_sum;
            int add = (int)std::min<int64_t>((int64_t)want, cap[i] - out[i]);
            out[i] += add; given += add;
            if (out[i] < cap[i]) frac.push_back({want - (int64_t)want, i});
        }
        left -= given;
        std::sort(frac.begin(), frac.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        for (auto& f : frac) { if (left <= 0) break; if (out[f.second] < cap[f.second]) { ++out[f.second]; --left; } }
    }
}

int kbbq_accumulate_bands_dev(kbbq_ctx* c, const kbbq_band* bands, int nbands, int R, int S2, int minscore, int dinuc_minscore,
                              int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nbands < 0 || (nbands > 0 && !bands)) return fail(KBBQ_E_ARG, "kbbq_accumulate_bands_dev: bad band list");
    // which bands the merged kernel (k1v3_bands) takes: the table-driven K1 fits their LDS geometry, position-major cycle
    // table (not the chunk-position-major form of 2 x 150 bp mate-pair rows), all on 4-bit planes or all on character planes
    std::vector<K1Setup> setups; std::vector<int> merged, alone;
    const char* off = getenv("KBBQ_K1_BANDS");                       // "0": a launch per band (A/B timing)
    int nib_of_merge = -1;
    for (int i = 0; i < nbands; ++i) {
        const kbbq_band& b = bands[i];
        if (b.nrows == 0) continue;
        int rc = layout_flags_ok("kbbq_accumulate_bands_dev", b.flags);
        if (rc) return rc;
        const int pairs = (b.flags & KBBQ_ROWS_PAIRS) ? 1 : 0, nib = (b.flags & KBBQ_ROWS_NIBBLES) ? 1 : 0;
        K1Setup su; bool fits = false;
        bool can = !(off && !strcmp(off, "0")) && (int)merged.size() < K1V3_MAX_BANDS;
        if (can) {
            if (pairs) { rc = check_pairs("kbbq_accumulate_bands_dev", b.nrows, S2); if (rc) return rc; }
            rc = accumulate_rows(c, "kbbq_accumulate_bands_dev", b.d_seq, b.d_cseq, b.d_qual, b.d_meta, b.nrows, b.pitch, pairs, R, S2, minscore,
                                 dinuc_minscore, b.d_seg, d_tables, &fits, pairs ? 0 : b.S_band, pairs ? 0 : b.S_min, nib,
                                 (b.flags & KBBQ_ROWS_TWINS) ? 1 : 0, &su);
            if (rc) return rc;
            can = fits && !su.km && (nib_of_merge < 0 || nib_of_merge == nib);
        }
        if (can) { nib_of_merge = nib; merged.push_back(i); setups.push_back(su); }
        else alone.push_back(i);
    }
    // two kinds of bands: narrow rows whose tables let two 12-wave workgroups share a CU (accumulate_rows), and the rest
    // (one 16-wave workgroup per CU); each kind with at least two bands is ONE launch, anything else goes alone
    std::vector<std::vector<size_t>> groups(2);
    for (size_t k = 0; k < merged.size(); ++k) groups[setups[k].threads == K1V3_THREADS ? 0 : 1].push_back(k);
    for (auto& grp : groups) {
        const int G = grp.empty() ? 1 : std::max(1, c->cus * (setups[grp[0]].threads == K1V3_THREADS ? 1 : 2) / std::max(R, 1));
        if (grp.size() < 2 || (int)grp.size() > G) { for (size_t k : grp) alone.push_back(merged[k]); grp.clear(); }
    }
    for (int i : alone) {
        int rc = band_accumulate_alone(c, bands[i], R, S2, minscore, dinuc_minscore, d_tables);
        if (rc) return rc;
    }
    HIPCHK(hipSetDevice(c->device));
    const char* rc_env = getenv("KBBQ_K1_BAND_ROWCOST");
    const double row_cost = rc_env ? std::max(0.0, atof(rc_env)) : 1.25;     // the knobs are clamped: a zero or negative weight would leave every band one workgroup
    // ... and a chunk of a WIDE row costs more than a chunk of a narrow one (larger tables to zero and flush, 8 instead of 16 copies
    // of the context table, more cycle columns for the same lanes): with equal cost per chunk the widest band of BASELINE config 5
    // ended 25 % after the narrowest (KBBQ_K1_BANDS_DBG=1 prints every band's workgroup end times): + 1 % per chunk of row width
    const char* sl_env = getenv("KBBQ_K1_BAND_SLOPE");
    const double slope = sl_env ? std::max(0.0, atof(sl_env)) : 0.01;
    const char* d8_env = getenv("KBBQ_K1_BAND_DN8");          // a band on 8 copies of the context table: twice the same-address atomics there
    const double dn8_cost = d8_env ? std::max(1.0, atof(d8_env)) : 1.06;
    for (auto& grp : groups) {
        if (grp.empty()) continue;
        const int threads = setups[grp[0]].threads;
        const int G = std::max(1, c->cus * (threads == K1V3_THREADS ? 1 : 2) / std::max(R, 1));
        K1BandsParams t;
        memset(&t, 0, sizeof t);
        t.nbands = (int)grp.size();
        // a band's share of the workgroups follows its work: chunks to bin plus a per-row term (a 64-row block pays the sidecar
        // -> row address -> first chunk latency whatever its width: narrow rows run at half the rate of wide ones)
        std::vector<double> weight; std::vector<int64_t> cap; std::vector<int> share;
        size_t lds = 0;
        for (size_t k : grp) {
            const K1Setup& su = setups[k];
            weight.push_back((double)su.q.nreads * (su.q.cpr + row_cost) * (1.0 + slope * su.q.cpr) * (su.dn == K1V3_DNREP ? 1.0 : dn8_cost));
            cap.push_back(std::max<int64_t>(1, su.iters));
            lds = std::max(lds, su.lds);
        }
        share_workgroups(weight, cap, G, share);
        int run = 0;
        for (size_t j = 0; j < grp.size(); ++j) {
            t.wg_start[j] = run; run += share[j];
            t.dn[j] = setups[grp[j]].dn; t.band[j] = setups[grp[j]].q;
        }
        t.wg_start[grp.size()] = run;
        const bool split = setups[grp[0]].split;
        dim3 grid((unsigned)run, (unsigned)R, 1), block((unsigned)threads, 1, 1);
        unsigned long long* dbg = nullptr;                    // KBBQ_K1_BANDS_DBG=1: when did every band's workgroups start and end?
        if (getenv("KBBQ_K1_BANDS_DBG")) { HIPCHK(hipMalloc((void**)&dbg, sizeof(unsigned long long) * 2 * (size_t)run)); t.dbg = dbg; }
        {
            Timed tm(c, 0);
            if (nib_of_merge) {
                if (split) hipLaunchKernelGGL((k1v3_bands<true, true>), grid, block, lds, c->stream, t);
                else hipLaunchKernelGGL((k1v3_bands<false, true>), grid, block, lds, c->stream, t);
            } else {
                if (split) hipLaunchKernelGGL((k1v3_bands<true, false>), grid, block, lds, c->stream, t);
                else hipLaunchKernelGGL((k1v3_bands<false, false>), grid, block, lds, c->stream, t);
            }
        }
        HIPCHK(hipGetLastError());
        if (dbg) {
            std::vector<unsigned long long> h(2 * (size_t)run);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(h.data(), dbg, h.size() * sizeof h[0], hipMemcpyDeviceToHost));
            (void)hipFree(dbg);
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < run; ++w) t0 = std::min(t0, h[(size_t)w]);
            for (size_t j = 0; j < grp.size(); ++j) {
                unsigned long long s1 = 0, e0 = ~0ull, e1 = 0;
                for (int w = t.wg_start[j]; w < t.wg_start[j + 1]; ++w) {
                    s1 = std::max(s1, h[(size_t)w] - t0); e0 = std::min(e0, h[(size_t)run + w] - t0); e1 = std::max(e1, h[(size_t)run + w] - t0);
                }
                fprintf(stderr, "[k1v3_bands] band %zu pitch %3d rows %9lld: %3d workgroups, last start %7.1f us, ends %7.1f .. %7.1f us\n", j, t.band[j].pitch,
                        (long long)t.band[j].nreads, t.wg_start[j + 1] - t.wg_start[j], s1 / 100.0, e0 / 100.0, e1 / 100.0);
            }
        }
    }
    return KBBQ_OK;
}

static int band_apply_alone(kbbq_ctx* c, const kbbq_band& b, int R, int S2, int minscore, const void* d_lut_blob)
{
    if (b.flags || b.d_seg)
        return kbbq_apply_rows_dev(c, b.d_seq, b.d_qual, b.d_meta, b.nrows, b.pitch, b.flags, R, S2, minscore, d_lut_blob, b.d_pair_lut,
                                   b.d_seg, b.d_perm, b.d_out);
    return kbbq_apply_dev(c, b.d_seq, b.d_qual, b.d_meta, b.nrows, b.pitch, R, KQ, S2, minscore, d_lut_blob, KBBQ_APPLY_FAST, b.d_out);
}

int kbbq_apply_bands_dev(kbbq_ctx* c, const kbbq_band* bands, int nbands, int R, int S2, int minscore, const void* d_lut_blob)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nbands < 0 || (nbands > 0 && !bands) || !d_lut_blob) return fail(KBBQ_E_ARG, "kbbq_apply_bands_dev: bad argument");
    if (R <= 0 || R > 32767 || S2 <= 0 || (S2 & 1) || S2 > 65534 || minscore < 0 || minscore > 222) return fail(KBBQ_E_ARG, "kbbq_apply_bands_dev: bad shape");
    // the merged kernel (k2t_bands) takes what the short-lived kernel takes of one-read-per-row 4-bit planes with one read
    // group: rows of 2..4096 chunks whose pitch-narrowed LUT is small (apply_rows' own test); the rest goes band by band
    const char* off = getenv("KBBQ_K2_BANDS");
    const char* tile = getenv("KBBQ_K2_TILE");
    const size_t lut_cap = (size_t)(getenv("KBBQ_K2_TILE_LUT_KB") ? atoi(getenv("KBBQ_K2_TILE_LUT_KB")) : 52) << 10;
    std::vector<int> merged, alone;
    for (int i = 0; i < nbands; ++i) {
        const kbbq_band& b = bands[i];
        if (b.nrows == 0) continue;
        int rc = layout_flags_ok("kbbq_apply_bands_dev", b.flags);
        if (rc) return rc;
        const int cpr = b.pitch / 16;
        const int Sb = std::min(b.pitch, S2);
        const bool can = !(off && !strcmp(off, "0")) && !(tile && !strcmp(tile, "0")) && (int)merged.size() < K2T_MAX_BANDS && R == 1
                         && b.flags == KBBQ_ROWS_NIBBLES && !b.d_seg && !b.d_perm && b.pitch > 0 && !(b.pitch & 15) && cpr >= 2 && cpr <= 4096
                         && (size_t)(33 + KQ) * full_lut_row_bytes(Sb) <= lut_cap
                         && !(((uintptr_t)b.d_seq | (uintptr_t)b.d_qual | (uintptr_t)b.d_out) & 15) && b.d_seq && b.d_qual && b.d_meta && b.d_out;
        (can ? merged : alone).push_back(i);
    }
    if (merged.size() < 2) { for (int i : merged) alone.push_back(i); merged.clear(); }
    for (int i : alone) {
        int rc = band_apply_alone(c, bands[i], R, S2, minscore, d_lut_blob);
        if (rc) return rc;
    }
    if (merged.empty()) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    // every band's LUT, narrowed to the columns rows of its pitch can reach, side by side in the context's scratch
    std::vector<size_t> lut_off; size_t need = 0;
    for (int i : merged) { lut_off.push_back(need); need += ((size_t)R * (33 + KQ) * full_lut_row_bytes(std::min(bands[i].pitch, S2)) + 255) & ~(size_t)255; }
    if (c->rowlut_bytes < need) {
        if (c->d_rowlut) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_rowlut); c->d_rowlut = nullptr; c->rowlut_bytes = 0; }
        HIPCHK(hipMalloc(&c->d_rowlut, need));
        c->rowlut_bytes = need;
    }
    K2tBandsParams t;
    memset(&t, 0, sizeof t);
    t.nbands = (int)merged.size();
    const int64_t per_wg = (int64_t)(K2T_THREADS / 64) * 64 * K2T_STEPS;
    int64_t run = 0; size_t lds = 0;
    for (size_t k = 0; k < merged.size(); ++k) {
        const kbbq_band& b = bands[merged[k]];
        const int Sb = std::min(b.pitch, S2);
        RowLutParams f;
        f.lut16 = reinterpret_cast<const int16_t*>(d_lut_blob); f.rs16 = lut_row_stride(S2); f.R = R; f.Qt = KQ; f.S2 = S2; f.Sb = Sb;
        f.minscore = std::min(std::max(minscore, 0), KQ);
        f.out = reinterpret_cast<int8_t*>(c->d_rowlut) + lut_off[k];
        hipLaunchKernelGGL(k3_fill_row_lut, dim3((unsigned)(R * (33 + KQ))), dim3(256), 0, c->stream, f);
        K2tParams& q = t.band[k];
        q.seq = b.d_seq; q.qual = b.d_qual; q.meta = b.d_meta; q.cpr = b.pitch / 16; q.cpr_magic = magic_for(q.cpr);
        q.nchunks = b.nrows * q.cpr; q.Qt = KQ; q.S2 = S2; q.maxlen = Sb;
        q.rb = (u32)full_lut_row_bytes(Sb); q.W = (u32)full_lut_width(Sb); q.ctx_off = 2u * q.W;
        q.lut = f.out; q.lut_bytes = (int)((((size_t)(33 + KQ) * q.rb) + 15) & ~(size_t)15);
        q.seg = nullptr; q.wg_start = nullptr; q.order = nullptr; q.R = R; q.perm = nullptr; q.pitch = b.pitch; q.out = b.d_out; q.status = c->d_status;
        { const char* x = getenv("KBBQ_K2_XCD_TILES"); q.xcd_tiles = x ? std::max(0, atoi(x)) : 1; }
        lds = std::max(lds, (size_t)q.lut_bytes);
        t.wg_start[k] = (int)run;
        run += (q.nchunks + per_wg - 1) / per_wg;
        if (run > 0x7FFFFFFF) return fail(KBBQ_E_ARG, "kbbq_apply_bands_dev: too many workgroups");
    }
    t.wg_start[merged.size()] = (int)run;
    {
        Timed tm(c, 1);
        hipLaunchKernelGGL(k2t_bands, dim3((unsigned)run), dim3(K2T_THREADS), lds, c->stream, t);
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_meta_stats_dev(kbbq_ctx* c, const uint32_t* d_meta, int64_t nreads, int32_t* h_stats8)
{
    if (!c || !h_stats8) return fail(KBBQ_E_ARG, "kbbq_meta_stats_dev: NULL argument");
    if (nreads < 0) return fail(KBBQ_E_ARG, "kbbq_meta_stats_dev: nreads < 0");
    static const int init[K7_NSTATS] = {0x7FFFFFFF, 0, 0, 0, 0, 0, 0, 0};
    memcpy(h_stats8, init, sizeof init);
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_stats) HIPCHK(hipMalloc((void**)&c->d_stats, sizeof init));
    HIPCHK(hipMemcpyAsync(c->d_stats, init, sizeof init, hipMemcpyHostToDevice, c->stream));
    MetaStatsParams p; p.meta = d_meta; p.n = nreads; p.stats = c->d_stats;
    const int64_t npairs = (nreads + 1) / 2;
    int gx = (int)std::min<int64_t>((npairs + 255) / 256, (int64_t)c->cus * 8);
    hipLaunchKernelGGL(k7_meta_stats, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_stats8, c->d_stats, sizeof init, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KBBQ_OK;
}

size_t kbbq_group_rows_work_bytes(int64_t nrows, int R)
{
    const int64_t nblocks = (nrows + K7_SORT_ROWS - 1) / K7_SORT_ROWS;
    return (size_t)std::max<int64_t>(nblocks, 1) * (size_t)std::max(R, 1) * sizeof(u32);
}

int kbbq_group_rows_dev(kbbq_ctx* c, const uint32_t* d_meta, int64_t nrows, int pairs, int R, void* d_work,
                        int64_t* d_perm, int64_t* d_seg)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nrows < 0 || nrows >= (1ll << 32)) return fail(KBBQ_E_ARG, "kbbq_group_rows_dev: nrows out of range");
    if (R <= 0 || R > K7_SORT_MAXR) return fail(KBBQ_E_ARG, "kbbq_group_rows_dev: 1 <= R <= %d (got %d)", K7_SORT_MAXR, R);
    if (!d_meta || !d_work || !d_perm || !d_seg) return fail(KBBQ_E_ARG, "kbbq_group_rows_dev: NULL pointer");
    HIPCHK(hipSetDevice(c->device));
    if (nrows == 0) { HIPCHK(hipMemsetAsync(d_seg, 0, (size_t)(R + 1) * 8, c->stream)); return KBBQ_OK; }
    RgSortParams p;
    p.meta = d_meta; p.nrows = nrows; p.pairs = pairs ? 1 : 0; p.R = R;
    p.nblocks = (nrows + K7_SORT_ROWS - 1) / K7_SORT_ROWS;
    p.hist = reinterpret_cast<u32*>(d_work); p.perm = reinterpret_cast<long long*>(d_perm); p.status = c->d_status;
    hipLaunchKernelGGL(k7_rg_sort<false>, dim3((unsigned)p.nblocks), dim3(K7_SORT_THREADS), 0, c->stream, p);
    RgScanParams s;
    s.hist = p.hist; s.count = p.nblocks * R; s.nblocks = p.nblocks; s.R = R; s.nrows = nrows; s.seg = reinterpret_cast<long long*>(d_seg);
    hipLaunchKernelGGL(k7_rg_scan, dim3(1), dim3(1024), 0, c->stream, s);
    hipLaunchKernelGGL(k7_rg_sort<true>, dim3((unsigned)p.nblocks), dim3(K7_SORT_THREADS), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_lay_out_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual, const uint32_t* d_meta,
                     int64_t nreads, int pitch, int flags, int S2, const int64_t* d_perm,
                     uint8_t* d_lseq, uint8_t* d_lcseq, uint8_t* d_lqual, uint32_t* d_lmeta)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = layout_flags_ok("kbbq_lay_out_dev", flags);
    if (rc) return rc;
    const int pairs = (flags & KBBQ_ROWS_PAIRS) ? 1 : 0, nib = (flags & KBBQ_ROWS_NIBBLES) ? 1 : 0;
    rc = check_planes("kbbq_lay_out_dev", nreads, pitch, d_seq, d_cseq, d_qual);
    if (rc) return rc;
    if (!d_seq || !d_qual || !d_meta || !d_lseq || !d_lqual || !d_lmeta || (d_cseq && !d_lcseq)) return fail(KBBQ_E_ARG, "kbbq_lay_out_dev: NULL pointer");
    if (((uintptr_t)d_lseq | (uintptr_t)d_lcseq | (uintptr_t)d_lqual) & 15) return fail(KBBQ_E_ARG, "kbbq_lay_out_dev: planes must be 16-byte aligned");
    if (pairs) {
        rc = check_pairs("kbbq_lay_out_dev", (nreads + 1) / 2, S2);
        if (rc) return rc;
        if ((nreads & 1) && !(flags & KBBQ_ROWS_TWINS)) return fail(KBBQ_E_ARG, "kbbq_lay_out_dev: mate-pair rows need an even number of reads");
        if (pitch < S2 / 2) return fail(KBBQ_E_ARG, "kbbq_lay_out_dev: pitch %d < read length %d", pitch, S2 / 2);
    }
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    LayOutParams p;
    p.src[0] = d_seq; p.src[1] = d_cseq; p.src[2] = d_qual; p.dst[0] = d_lseq; p.dst[1] = d_lcseq; p.dst[2] = d_lqual;
    p.meta = d_meta; p.dmeta = d_lmeta; p.perm = reinterpret_cast<const long long*>(d_perm);
    p.nrows = pairs ? (nreads + 1) / 2 : nreads; p.nsrc = nreads; p.pitch = pitch; p.dpitch = pairs ? pair_pitch(S2) : pitch; p.S = pairs ? S2 / 2 : 0;
    p.pairs = pairs; p.nib = nib; p.status = c->d_status;
    const int rpb7 = (p.dpitch / 16) <= 256 ? 256 / (p.dpitch / 16) : 1;      // destination rows per workgroup iteration
    const int gx = bounded_grid((p.nrows + rpb7 - 1) / rpb7, c, 256, "KBBQ_K7_GRID");
    { const char* x = getenv("KBBQ_K7_PARTS"); p.parts = x ? atoi(x) : 8; }
    hipLaunchKernelGGL(k7_lay_out, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_tables_add_dev(kbbq_ctx* c, int64_t* d_dst, const int64_t* d_src, size_t n)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (n && (!d_dst || !d_src)) return fail(KBBQ_E_ARG, "kbbq_tables_add_dev: NULL pointer");
    if (n == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    AddTablesParams p; p.dst = (long long*)d_dst; p.src = (const long long*)d_src; p.n = (long long)n;
    int gx = (int)std::min<size_t>((n + 255) / 256, (size_t)c->cus * 8);
    hipLaunchKernelGGL(k7_add_tables, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_unpack_nibbles_dev(kbbq_ctx* c, const uint8_t* d_nib, int64_t nbases, uint8_t* d_chars)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nbases < 0 || (nbases & 15) || ((uintptr_t)d_nib & 7) || ((uintptr_t)d_chars & 15)) return fail(KBBQ_E_ARG, "kbbq_unpack_nibbles_dev: bad size or alignment");
    if (nbases == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    UnNibParams p; p.src = d_nib; p.dst = d_chars; p.nchunks = nbases / 16;
    int gx = (int)std::min<int64_t>((p.nchunks + 255) / 256, (int64_t)c->cus * 16);
    hipLaunchKernelGGL(k7_unpack_nibbles, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// the first four operations of every read inline, one 32-byte record per read (context-owned scratch); with aflags_out also
// the classification kbbq_tally_aligned_dev needs (K4RecParams)
static int k4_records(kbbq_ctx* c, const K4Params& p, const u32* aflags_in, u32* aflags_out, u32* rows, u32* nrows, int pitch)
{
    const size_t need = (size_t)p.nreads * sizeof(K4Rec);
    if (c->ops4_bytes < need) {
        if (c->d_ops4) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_ops4); c->d_ops4 = nullptr; c->ops4_bytes = 0; }
        HIPCHK(hipMalloc(&c->d_ops4, need));
        c->ops4_bytes = need;
    }
    K4RecParams ip; ip.len = p.len; ip.ref_len = p.ref_len; ip.cig_off = p.cig_off; ip.cig_n = p.cig_n; ip.cigar = p.cigar;
    ip.nreads = p.nreads; ip.recs = (K4Rec*)c->d_ops4;
    ip.aflags_in = aflags_in; ip.aflags_out = aflags_out; ip.rows = rows; ip.nrows = nrows;
    ip.ref_start = p.ref_start; ip.genome_len = p.genome_len; ip.pitch = pitch;
    int gi = (int)std::min<int64_t>((p.nreads + 255) / 256, (int64_t)c->cus * 16);
    hipLaunchKernelGGL(k4_read_records, dim3((unsigned)std::max(gi, 1)), dim3(256), 0, c->stream, ip);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// ---- K4 / K5: benchmark path --------------------------------------------
int kbbq_find_errors_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint32_t* d_len, int64_t nreads, int pitch,
                         const int64_t* d_ref_start, const int32_t* d_ref_len,
                         const uint32_t* d_cig_off, const uint32_t* d_cig_n, const uint32_t* d_cigar,
                         const uint8_t* d_genome, const uint8_t* d_skipmask, int64_t genome_len,
                         const uint8_t* d_flip, uint8_t* d_err, uint8_t* d_skip)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nreads < 0 || pitch <= 0 || (pitch & 15) || genome_len < 0) return fail(KBBQ_E_ARG, "kbbq_find_errors_dev: bad nreads/pitch/genome_len");
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    K4Params p;
    p.seq = d_seq; p.len = d_len; p.nreads = nreads; p.pitch = pitch;
    p.ref_start = (const long long*)d_ref_start; p.ref_len = d_ref_len;
    p.cig_off = d_cig_off; p.cig_n = d_cig_n; p.cigar = d_cigar;
    p.genome = d_genome; p.skipmask = d_skipmask; p.genome_len = genome_len; p.flip = d_flip; p.err = d_err; p.skip = d_skip;
    p.status = c->d_status;
    if (((uintptr_t)d_seq | (uintptr_t)d_err | (uintptr_t)d_skip) & 15)
        return fail(KBBQ_E_ARG, "kbbq_find_errors_dev: planes must be 16-byte aligned");
    const int rpb4 = (pitch / 16) <= 256 ? 256 / (pitch / 16) : 1;              // reads per workgroup iteration
    const int gx = bounded_grid((nreads + rpb4 - 1) / rpb4, c, 64, "KBBQ_K4_GRID");
    const char* force = getenv("KBBQ_K4");              // "v1" forces the first form (A/B timing): needs both output planes
    if (force && !strcmp(force, "v1") && d_skip) {
        if (d_skipmask) hipLaunchKernelGGL(k4_find_errors<false>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
        else hipLaunchKernelGGL(k4_find_errors<true>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
        HIPCHK(hipGetLastError());
        return KBBQ_OK;
    }
    int rc4 = k4_records(c, p, nullptr, nullptr, nullptr, nullptr, 0);
    if (rc4) return rc4;
    K4v2Params q; q.base = p; q.recs = (const K4Rec*)c->d_ops4; q.idle16 = reinterpret_cast<const uint8_t*>(c->d_status);
    q.rows = nullptr; q.nrows = nullptr;
    if (d_skipmask) hipLaunchKernelGGL(k4v2_find_errors<false>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, q);
    else hipLaunchKernelGGL(k4v2_find_errors<true>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, q);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_canonical_reads_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_err,
                             const uint8_t* d_skip, const uint32_t* d_len, const uint32_t* d_clip,
                             const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads, int pitch, int S,
                             int minscore, int dinuc_minscore, uint8_t* d_out_seq, uint8_t* d_out_cseq,
                             uint8_t* d_out_qual, uint32_t* d_out_meta)
{
    return kbbq_canonical_reads_rows_dev(c, d_seq, d_oq, d_err, d_skip, d_len, d_clip, d_trim, d_flags, nreads, pitch, S,
                                         minscore, dinuc_minscore, 0, d_out_seq, d_out_cseq, d_out_qual, d_out_meta);
}

int kbbq_canonical_reads_rows_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_err,
                                  const uint8_t* d_skip, const uint32_t* d_len, const uint32_t* d_clip,
                                  const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads, int pitch, int S,
                                  int minscore, int dinuc_minscore, int layout, uint8_t* d_out_seq, uint8_t* d_out_cseq,
                                  uint8_t* d_out_qual, uint32_t* d_out_meta)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (layout & ~KBBQ_ROWS_NIBBLES) return fail(KBBQ_E_ARG, "kbbq_canonical_reads_rows_dev: the output is one read per row (layout 0 or KBBQ_ROWS_NIBBLES)");
    if (nreads < 0 || pitch <= 0 || (pitch & 15) || S <= 0 || S > pitch || S > 65535)
        return fail(KBBQ_E_ARG, "kbbq_canonical_reads_dev: bad nreads/pitch/S");
    if (minscore < 0 || minscore > 94 || dinuc_minscore < 0 || dinuc_minscore > 94)
        return fail(KBBQ_E_ARG, "kbbq_canonical_reads_dev: minscore out of range");
    if (nreads == 0) return KBBQ_OK;
    if (!d_seq || !d_oq || !d_err) return fail(KBBQ_E_ARG, "kbbq_canonical_reads_dev: NULL plane");
    if (((uintptr_t)d_seq | (uintptr_t)d_oq | (uintptr_t)d_err | (uintptr_t)d_skip | (uintptr_t)d_out_seq
         | (uintptr_t)d_out_cseq | (uintptr_t)d_out_qual) & 15)
        return fail(KBBQ_E_ARG, "kbbq_canonical_reads_dev: planes must be 16-byte aligned");
    HIPCHK(hipSetDevice(c->device));
    K6Params p;
    p.seq = d_seq; p.oq = d_oq; p.err = d_err; p.skip = d_skip; p.len = d_len; p.clip = d_clip; p.trim = d_trim;
    p.flags = d_flags; p.nreads = nreads; p.pitch = pitch; p.S = S;
    p.qlo = 33u + (u32)minscore; p.dlo = 33u + (u32)dinuc_minscore;
    p.out_seq = d_out_seq; p.out_cseq = d_out_cseq; p.out_qual = d_out_qual; p.out_meta = d_out_meta;
    p.status = c->d_status;
    const int rpb6 = (pitch / 16) <= 256 ? 256 / (pitch / 16) : 1;
    const int gx = bounded_grid((nreads + rpb6 - 1) / rpb6, c, 64, "KBBQ_K6_GRID");
    if (layout & KBBQ_ROWS_NIBBLES) hipLaunchKernelGGL(k6_canonical_reads<true>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL(k6_canonical_reads<false>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// K6 fused into K1 (k1v3_aligned): the BAM-sourced tally straight from the reads as aligned -- 3 B/base read (sequence, OQ,
// K4's plane of flags), nothing written but the count tables; kbbq_canonical_reads_rows_dev + kbbq_accumulate_rows_dev moved
// 3 + 2 + 2 B/base for the same tables.
// d_genome / d_g0 non-NULL: the REF form (k1v3_aligned_ref) -- reads whose d_flags word has bit 2 set are compared with the
// reference by the kernel itself (kbbq_tally_aligned_dev)
static int accumulate_aligned(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_flagplane,
                              const uint32_t* d_clip, const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads,
                              int pitch, int S, int R, int minscore, int dinuc_minscore, int64_t* d_tables,
                              const uint8_t* d_genome, const int64_t* d_g0, bool dry_run = false)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    int rc = check_planes("kbbq_accumulate_aligned_dev", nreads, pitch, d_seq, d_oq, d_flagplane);
    if (rc) return rc;
    if (S <= 0 || S > pitch || S > 32767 || pitch != ((S + 15) & ~15)) return fail(KBBQ_E_ARG, "kbbq_accumulate_aligned_dev: rows of S = %d bases have pitch %d (got %d)", S, (S + 15) & ~15, pitch);
    if (R <= 0 || R > 32767) return fail(KBBQ_E_ARG, "kbbq_accumulate_aligned_dev: R out of range (%d)", R);
    if (minscore < 0 || minscore > KQ - 1 || dinuc_minscore < 0 || dinuc_minscore > 94) return fail(KBBQ_E_ARG, "kbbq_accumulate_aligned_dev: minscore out of range");
    if (nreads == 0) return KBBQ_OK;
    if (!d_seq || !d_oq || !d_flagplane || !d_clip || !d_trim || !d_flags || !d_tables) return fail(KBBQ_E_ARG, "kbbq_accumulate_aligned_dev: NULL pointer");
    if (S < 32) return fail(KBBQ_E_LUT, "kbbq_accumulate_aligned_dev: reads of %d bases (< 32) are tallied through kbbq_canonical_reads_rows_dev", S);
    HIPCHK(hipSetDevice(c->device));
    K1v3Params q;
    memset(&q, 0, sizeof q);
    q.seq = d_seq; q.cseq = d_flagplane; q.qual = d_oq; q.meta = nullptr;
    q.aflags = d_flags; q.aclip = d_clip; q.atrim = d_trim;
    q.genome = d_genome; q.ag0 = reinterpret_cast<const long long*>(d_g0);
    q.nreads = nreads; q.pitch = pitch; q.cpr = pitch / 16; q.cpr_magic = magic_for(q.cpr);
    q.R = R; q.S = S; q.gS2 = 2 * S; q.minscore = minscore; q.type_minscore = dinuc_minscore;
    q.qlo_m1 = 32u + (u32)minscore; q.dlo = 33u + (u32)dinuc_minscore;
    q.nrows = KQ + 1 - minscore;
    q.maxlen = S; q.gap = 0; q.twins = 0; q.seg = nullptr;
    q.tables = reinterpret_cast<u64*>(d_tables); q.status = c->d_status;
    // LDS geometry as accumulate_rows: 3S words per cycle row (every read has length S: [0, S) read 1, [S, 2S) read 2 mirrored);
    // 16 copies of the context table, or 8 when that does not fit
    int dn = 0; size_t lds3 = 0;
    for (int copies : {K1V3_DNREP, 8}) {
        const int words = (3 * S) | 1;
        // minlen = S: the flush walks columns [0, 2S) only -- the words behind them take the uncounted bytes in front of a read's
        // first aligned base (kernel comment) and are never read
        q.row_bytes = (u32)words * 4u; q.minlen = S; q.slack_bytes = (u32)(S + 32) * 4u;
        q.dn_flush_iters = std::max(1, 65535 / ((K1V3_THREADS / copies) * 16 * q.cpr));
        lds3 = (size_t)q.nrows * 128 * copies + (size_t)q.nrows * q.row_bytes + q.slack_bytes;
        if (lds3 <= (size_t)c->lds_bytes && (K1V3_THREADS / copies) * 16 * q.cpr <= 65535) { dn = copies; break; }
    }
    if (!dn) return fail(KBBQ_E_LUT, "kbbq_accumulate_aligned_dev: %d-base reads with minscore %d do not fit the LDS tables; tally through kbbq_canonical_reads_rows_dev", S, minscore);
    if (dry_run) return KBBQ_OK;                      // the caller only asked whether this shape is served
    {   // several trash rows, then copies of the cycle table for short reads, as the LDS allows (accumulate_rows)
        q.pos_copies = 1; q.ntrash = 1;
        auto bytes_for = [&](int nt, int pc) {
            const size_t rows = (size_t)q.nrows - 1 + nt;
            return rows * 128 * dn + (size_t)pc * (rows * q.row_bytes + q.slack_bytes);
        };
        const char* nt_env = getenv("KBBQ_K1_NTRASH");
        const char* pc_env = getenv("KBBQ_K1_POSCOPIES");
        const int most_t = nt_env ? atoi(nt_env) : 8, most = pc_env ? atoi(pc_env) : 4;
        for (int nt = 8; nt >= 2; nt >>= 1)
            if (nt <= most_t && bytes_for(nt, 1) <= (size_t)c->lds_bytes) { q.ntrash = nt; break; }
        if (q.cpr <= 12)
            for (int pc = 4; pc >= 2; pc >>= 1)
                if (pc <= most && bytes_for(q.ntrash, pc) <= (size_t)c->lds_bytes) { q.pos_copies = pc; break; }
        q.pos_copy_bytes = (u32)((size_t)(q.nrows - 1 + q.ntrash) * q.row_bytes + q.slack_bytes);
        lds3 = bytes_for(q.ntrash, q.pos_copies);
    }
    const bool split = dinuc_minscore > minscore;
    const int64_t nblocks = (nreads + 63) / 64;
    const int64_t iters = (nblocks + (K1V3_THREADS / 64) - 1) / (K1V3_THREADS / 64);
    const int gx = (int)std::min<int64_t>(iters, std::max(1, c->cus / R));
    dim3 grid((unsigned)gx, (unsigned)R, 1), block(K1V3_THREADS, 1, 1);
    {
        Timed t(c, 0);
        if (d_genome) {
            if (dn == K1V3_DNREP) {
                if (split) hipLaunchKernelGGL((k1v3_aligned_ref<true, K1V3_DNREP>), grid, block, lds3, c->stream, q);
                else hipLaunchKernelGGL((k1v3_aligned_ref<false, K1V3_DNREP>), grid, block, lds3, c->stream, q);
            } else {
                if (split) hipLaunchKernelGGL((k1v3_aligned_ref<true, 8>), grid, block, lds3, c->stream, q);
                else hipLaunchKernelGGL((k1v3_aligned_ref<false, 8>), grid, block, lds3, c->stream, q);
            }
        } else if (dn == K1V3_DNREP) {
            if (split) hipLaunchKernelGGL((k1v3_aligned<true, K1V3_DNREP>), grid, block, lds3, c->stream, q);
            else hipLaunchKernelGGL((k1v3_aligned<false, K1V3_DNREP>), grid, block, lds3, c->stream, q);
        } else {
            if (split) hipLaunchKernelGGL((k1v3_aligned<true, 8>), grid, block, lds3, c->stream, q);
            else hipLaunchKernelGGL((k1v3_aligned<false, 8>), grid, block, lds3, c->stream, q);
        }
    }
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_accumulate_aligned_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_flagplane,
                                const uint32_t* d_clip, const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads,
                                int pitch, int S, int R, int minscore, int dinuc_minscore, int64_t* d_tables)
{
    return accumulate_aligned(c, d_seq, d_oq, d_flagplane, d_clip, d_trim, d_flags, nreads, pitch, S, R, minscore, dinuc_minscore,
                              d_tables, nullptr, nullptr);
}

// The whole BAM-sourced tally (gatk/bqsr.py:52-123: find_read_errors per read, then the covariate counts) with ONE pass over
// the reads for everything but the reads K4 has to walk.  kbbq_find_errors_dev + kbbq_accumulate_aligned_dev move read bytes
// and the reference in, a plane of flags out and read bytes, flags and OQ in again: 6 B/base.  Here k4_read_records sorts the
// reads: one M / = / X operation over all bases, every 16-byte reference window readable -> the tally kernel compares read and
// reference itself (k1v3_aligned_ref: read bytes, OQ, reference window = 3 B/base); every other read (indels, clips in the
// CIGAR, anything odd) is listed, k4v2_find_errors writes the rows of d_flagplane of just those reads, and the tally kernel
// reads them there.  d_flagplane: [nreads, pitch] scratch of the caller's, contents on entry irrelevant.  The reference must
// carry the site flags in bit 7 (d_skipmask == NULL in kbbq_find_errors_dev's terms).  Every read has S bases (d_len[i] == S
// is the caller's promise, as for kbbq_accumulate_aligned_dev).  Counts are what kbbq_find_errors_dev (no flip) followed by
// kbbq_accumulate_aligned_dev count; the same statuses are raised (KBBQ_E_LUT: use kbbq_canonical_reads_rows_dev).
int kbbq_tally_aligned_dev(kbbq_ctx* c, const uint8_t* d_seq, const uint8_t* d_oq, const uint32_t* d_len, int64_t nreads,
                           int pitch, int S, const int64_t* d_ref_start, const int32_t* d_ref_len,
                           const uint32_t* d_cig_off, const uint32_t* d_cig_n, const uint32_t* d_cigar,
                           const uint8_t* d_genome, int64_t genome_len,
                           const uint32_t* d_clip, const uint32_t* d_trim, const uint32_t* d_flags,
                           uint8_t* d_flagplane, int R, int minscore, int dinuc_minscore, int64_t* d_tables)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nreads < 0 || nreads > 0xFFFFFFFFll || pitch <= 0 || (pitch & 15) || genome_len < 0) return fail(KBBQ_E_ARG, "kbbq_tally_aligned_dev: bad nreads/pitch/genome_len");
    if (nreads == 0) return KBBQ_OK;
    if (!d_seq || !d_oq || !d_len || !d_ref_start || !d_ref_len || !d_cig_off || !d_cig_n || !d_cigar || !d_genome || !d_clip || !d_trim
        || !d_flags || !d_flagplane || !d_tables) return fail(KBBQ_E_ARG, "kbbq_tally_aligned_dev: NULL pointer");
    if (((uintptr_t)d_seq | (uintptr_t)d_oq | (uintptr_t)d_flagplane) & 15) return fail(KBBQ_E_ARG, "kbbq_tally_aligned_dev: planes must be 16-byte aligned");
    // what the tally kernel refuses it refuses before anything is launched (same checks, same codes)
    if (S <= 0 || S > pitch || S > 32767 || pitch != ((S + 15) & ~15)) return fail(KBBQ_E_ARG, "kbbq_tally_aligned_dev: rows of S = %d bases have pitch %d (got %d)", S, (S + 15) & ~15, pitch);
    if (S < 32) return fail(KBBQ_E_LUT, "kbbq_tally_aligned_dev: reads of %d bases (< 32) are tallied through kbbq_canonical_reads_rows_dev", S);
    int rc = accumulate_aligned(c, d_seq, d_oq, d_flagplane, d_clip, d_trim, d_flags, nreads, pitch, S, R, minscore, dinuc_minscore,
                                d_tables, d_genome, d_ref_start, true);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    const size_t need = 16 + 2 * (size_t)nreads * sizeof(u32);
    if (c->tally_bytes < need) {
        if (c->d_tally) { HIPCHK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_tally); c->d_tally = nullptr; c->tally_bytes = 0; }
        HIPCHK(hipMalloc(&c->d_tally, need));
        c->tally_bytes = need;
    }
    u32* count = reinterpret_cast<u32*>(c->d_tally);
    u32* aflags2 = count + 4;
    u32* rows = aflags2 + nreads;
    HIPCHK(hipMemsetAsync(count, 0, 16, c->stream));
    K4Params p;
    p.seq = d_seq; p.len = d_len; p.nreads = nreads; p.pitch = pitch;
    p.ref_start = (const long long*)d_ref_start; p.ref_len = d_ref_len;
    p.cig_off = d_cig_off; p.cig_n = d_cig_n; p.cigar = d_cigar;
    p.genome = d_genome; p.skipmask = nullptr; p.genome_len = genome_len; p.flip = nullptr; p.err = d_flagplane; p.skip = nullptr;
    p.status = c->d_status;
    rc = k4_records(c, p, d_flags, aflags2, rows, count, pitch);
    if (rc) return rc;
    {   // K4 over the listed reads only: their number is on the device, the grid is sized for a share of the reads (a workgroup
        // beyond the list's end finds nothing to do) -- KBBQ_TALLY_K4_SHARE: 1 / share of the reads' grid, default 8
        const int rpb4 = (pitch / 16) <= 256 ? 256 / (pitch / 16) : 1;
        const char* sh = getenv("KBBQ_TALLY_K4_SHARE");
        const int share = sh && atoi(sh) > 0 ? atoi(sh) : 8;
        const int gx = bounded_grid(((nreads + share - 1) / share + rpb4 - 1) / rpb4, c, 64, "KBBQ_K4_GRID");
        K4v2Params q; q.base = p; q.recs = (const K4Rec*)c->d_ops4; q.idle16 = reinterpret_cast<const uint8_t*>(c->d_status);
        q.rows = rows; q.nrows = count;
        hipLaunchKernelGGL(k4v2_find_errors<true>, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, q);
        HIPCHK(hipGetLastError());
    }
    return accumulate_aligned(c, d_seq, d_oq, d_flagplane, d_clip, d_trim, aflags2, nreads, pitch, S, R, minscore, dinuc_minscore,
                              d_tables, d_genome, d_ref_start);
}

int kbbq_count_q_dev(kbbq_ctx* c, const uint8_t* d_qual, const uint8_t* d_err, const uint8_t* d_skip,
                     const uint32_t* d_len, int64_t nreads, int pitch, int qoffset, int64_t* d_counts512)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (!d_qual || !d_err) return fail(KBBQ_E_ARG, "kbbq_count_q_dev: NULL plane");
    int rc = check_planes("kbbq_count_q_dev", nreads, pitch, d_qual, d_err, d_skip);
    if (rc) return rc;
    if (nreads == 0) return KBBQ_OK;
    HIPCHK(hipSetDevice(c->device));
    K5Params p;
    p.qual = d_qual; p.err = d_err; p.skip = d_skip; p.len = d_len; p.nreads = nreads; p.pitch = pitch;
    p.cpr = pitch / 16; p.cpr_magic = magic_for(p.cpr); p.qoffset = qoffset;
    p.counts = reinterpret_cast<u64*>(d_counts512); p.status = c->d_status;
    const int64_t nchunks = nreads * p.cpr;
    const int gx = bounded_grid((nchunks + 255) / 256, c, 64, "KBBQ_K5_GRID");
    hipLaunchKernelGGL(k5_count_q, dim3((unsigned)std::max(gx, 1)), dim3(256), 0, c->stream, p);
    HIPCHK(hipGetLastError());
    return KBBQ_OK;
}

// ---- host-buffer entry points: stage, run, fetch -------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};

// ---- host-buffer entry points: slabs of rows through page-locked staging -------------------------------------------------
// A caller's NumPy arrays are pageable memory: hipMemcpy from them goes through the runtime's own small staging buffers, one
// plane after the other, and the kernel starts when the last byte has arrived.  Here the rows travel in slabs: host threads
// copy slab k + 1 into a page-locked buffer while the copy engine uploads slab k and the kernel runs on slab k - 1 (apply: the
// new qualities of slab k - 2 come back meanwhile, PCIe being full duplex); device memory is two slabs, whatever the input's size.
static size_t stage_slab_rows(int pitch, int planes, int64_t nreads)
{
    const char* e = getenv("KBBQ_STAGE_MB");                       // staging bytes per slab (all planes), default 96 MB
    const size_t budget = (size_t)(e && atoi(e) > 0 ? atoi(e) : 96) << 20;
    size_t rows = budget / ((size_t)pitch * planes + 4);
    rows = std::max<size_t>(rows & ~(size_t)63, 64);
    return (size_t)std::min<int64_t>((int64_t)rows, std::max<int64_t>(nreads, 1));
}

static int stage_prepare(kbbq_ctx* c, size_t bytes)
{
    if (!c->stage_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->stage_stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->stage_down_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipEventCreateWithFlags(&c->stage_up[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->stage_used[b], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->stage_down[b], hipEventDisableTiming));
        }
    }
    if (c->stage_bytes < bytes) {
        HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipStreamSynchronize(c->stage_stream)); HIPCHK(hipStreamSynchronize(c->stage_down_stream));
        for (int b = 0; b < 2; ++b) {
            if (c->stage_host[b]) { (void)hipHostFree(c->stage_host[b]); c->stage_host[b] = nullptr; }
            if (c->stage_dev[b]) { (void)hipFree(c->stage_dev[b]); c->stage_dev[b] = nullptr; }
        }
        c->stage_bytes = 0;
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipHostMalloc(&c->stage_host[b], bytes, hipHostMallocDefault));
            HIPCHK(hipMalloc(&c->stage_dev[b], bytes));
        }
        c->stage_bytes = bytes;
    }
    return KBBQ_OK;
}

// copy `bytes` with all host threads (a slab is tens of MB: one thread moves ~10 GB/s, PCIe takes ~55)
static void threaded_copy(void* dst, const void* src, size_t bytes)
{
    const unsigned nt = kbbq_threads_for(bytes / 4);
    if (nt <= 1 || bytes < ((size_t)4 << 20)) { memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(bytes, (size_t)t * per), hi = std::min(bytes, lo + per);
        if (lo < hi) memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
    });
}

// Every exit of a pipelined run that leaves before its last slab -- a launch or a copy failed -- waits for what is still in
// flight: the staging buffers and their events belong to the context, and the next call's first two slabs do not wait on
// events of an earlier call.
static int stage_drain(kbbq_ctx* c, int rc)
{
    if (c->stage_stream) (void)hipStreamSynchronize(c->stage_stream);
    if (c->stage_down_stream) (void)hipStreamSynchronize(c->stage_down_stream);
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

// The status words hold the SMALLEST flagged read index of every kind over all launches since they were last read; slabs number
// their reads from 0, so after a pipelined run that flagged something the slabs are looked at again one by one (input errors
// are the rare case): the first slab that reproduces a status decides, its read index made file-wide.
static int first_offender(kbbq_ctx* c, int64_t nreads, size_t slab, const std::function<int(int64_t, int64_t)>& run_slab)
{
    for (int64_t lo = 0; lo < nreads; lo += (int64_t)slab) {
        const int64_t m = std::min<int64_t>((int64_t)slab, nreads - lo);
        int rc = run_slab(lo, m);
        if (rc) return rc;
        int64_t idx = -1;
        rc = kbbq_ctx_status(c, &idx);
        if (rc) {
            if (idx >= 0) {
                const std::string what = g_err;
                const size_t colon = what.find(": ");
                return fail(rc, "read %lld%s", (long long)(lo + idx), colon == std::string::npos ? "" : what.c_str() + colon);
            }
            return rc;
        }
    }
    return KBBQ_OK;
}

int kbbq_accumulate(kbbq_ctx* c, const uint8_t* seq, const uint8_t* cseq, const uint8_t* qual,
                    const uint32_t* meta, int64_t nreads, int pitch, int R, int S2, int minscore,
                    int64_t* pos_errs, int64_t* pos_total, int64_t* dinuc_errs, int64_t* dinuc_total)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nreads < 0 || pitch <= 0 || (pitch & 15)) return fail(KBBQ_E_ARG, "kbbq_accumulate: bad nreads/pitch");
    if (nreads > 0 && (!seq || !cseq || !qual || !meta)) return fail(KBBQ_E_ARG, "kbbq_accumulate: NULL plane");
    if (R <= 0 || S2 <= 0 || !pos_errs || !pos_total || !dinuc_errs || !dinuc_total) return fail(KBBQ_E_ARG, "kbbq_accumulate: bad tables");
    HIPCHK(hipSetDevice(c->device));
    const size_t npos = (size_t)R * KQ * S2, ndn = (size_t)R * KQ * KND;
    DevBuf dt;
    HIPCHK(dt.alloc(kbbq_tables_count(R, S2) * 8));
    HIPCHK(hipMemsetAsync(dt.p, 0, kbbq_tables_count(R, S2) * 8, c->stream));
    const size_t slab = stage_slab_rows(pitch, 3, nreads);
    const size_t plane = slab * (size_t)pitch, set = 3 * plane + slab * 4;
    int rc = stage_prepare(c, set);
    if (rc) return rc;
    auto launch = [&](int b, int64_t m) {
        const uint8_t* d = (const uint8_t*)c->stage_dev[b];
        return kbbq_accumulate_dev(c, d, d + plane, d + 2 * plane, (const uint32_t*)(d + 3 * plane), m, pitch, R, S2, minscore, (int64_t*)dt.p);
    };
    auto stage_in = [&](int b, int64_t lo, int64_t m) {
        uint8_t* h = (uint8_t*)c->stage_host[b];
        const size_t off = (size_t)lo * pitch, nb = (size_t)m * pitch;
        threaded_copy(h, seq + off, nb); threaded_copy(h + plane, cseq + off, nb); threaded_copy(h + 2 * plane, qual + off, nb);
        memcpy(h + 3 * plane, meta + lo, (size_t)m * 4);
    };
    auto pipelined = [&]() -> int {
        int64_t k = 0;
        for (int64_t lo = 0; lo < nreads; lo += (int64_t)slab, ++k) {
            const int b = (int)(k & 1);
            const int64_t m = std::min<int64_t>((int64_t)slab, nreads - lo);
            if (k >= 2) HIPCHK(hipEventSynchronize(c->stage_up[b]));              // the page-locked slab has been uploaded: free again
            stage_in(b, lo, m);
            if (k >= 2) HIPCHK(hipStreamWaitEvent(c->stage_stream, c->stage_used[b], 0));   // the device slab's kernel has run
            HIPCHK(hipMemcpyAsync(c->stage_dev[b], c->stage_host[b], set, hipMemcpyHostToDevice, c->stage_stream));
            HIPCHK(hipEventRecord(c->stage_up[b], c->stage_stream));
            HIPCHK(hipStreamWaitEvent(c->stream, c->stage_up[b], 0));
            const int r2 = launch(b, m);
            if (r2) return r2;
            HIPCHK(hipEventRecord(c->stage_used[b], c->stream));
        }
        return KBBQ_OK;
    };
    rc = pipelined();
    if (rc) return stage_drain(c, rc);
    rc = kbbq_ctx_status(c, nullptr);
    if (rc) {
        // something was flagged: which read of the WHOLE input comes first?  (nothing reaches the caller's tables)
        HIPCHK(hipStreamSynchronize(c->stage_stream));
        rc = first_offender(c, nreads, slab, [&](int64_t lo, int64_t m) {
            stage_in(0, lo, m);
            HIPCHK(hipMemcpyAsync(c->stage_dev[0], c->stage_host[0], set, hipMemcpyHostToDevice, c->stream));
            return launch(0, m);
        });
        return rc ? rc : fail(KBBQ_E_HIP, "kbbq_accumulate: a status reported by the pipelined run did not reproduce slab by slab");
    }
    std::vector<int64_t> h(kbbq_tables_count(R, S2));
    HIPCHK(hipMemcpyAsync(h.data(), dt.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < npos; ++i) { pos_errs[i] += h[i]; pos_total[i] += h[npos + i]; }
    for (size_t i = 0; i < ndn; ++i) { dinuc_errs[i] += h[2 * npos + i]; dinuc_total[i] += h[2 * npos + ndn + i]; }
    return KBBQ_OK;
}

int kbbq_apply(kbbq_ctx* c, const uint8_t* seq, const uint8_t* qual, const uint32_t* meta,
               int64_t nreads, int pitch, int R, int Qt, int S2, int D, int minscore,
               const int64_t* meanq, const int64_t* rgdq, const int64_t* qdq,
               const int64_t* posdq, const int64_t* dinucdq, uint8_t* qual_out)
{
    if (!c) return fail(KBBQ_E_ARG, "ctx is NULL");
    if (nreads < 0 || pitch <= 0 || (pitch & 15)) return fail(KBBQ_E_ARG, "kbbq_apply: bad nreads/pitch");
    if (nreads > 0 && (!seq || !qual || !meta || !qual_out)) return fail(KBBQ_E_ARG, "kbbq_apply: NULL plane");
    HIPCHK(hipSetDevice(c->device));
    std::vector<int64_t> lut((kbbq_lut_bytes(R, Qt, S2) + 7) / 8);      // 8-byte words: aligned storage
    int flags = 0;
    int rc = kbbq_build_lut(R, Qt, S2, D, minscore, meanq, rgdq, qdq, posdq, dinucdq, lut.data(), &flags);
    if (rc) return rc;
    DevBuf dl;
    HIPCHK(dl.alloc(lut.size() * 8));
    HIPCHK(hipMemcpyAsync(dl.p, lut.data(), lut.size() * 8, hipMemcpyHostToDevice, c->stream));
    // a slab: seq | qual | out | meta (the output plane travels back from the same device slab)
    const size_t slab = stage_slab_rows(pitch, 3, nreads);
    const size_t plane = slab * (size_t)pitch, set = 3 * plane + slab * 4;
    rc = stage_prepare(c, set);
    if (rc) return rc;
    const int mode = flags ? KBBQ_APPLY_CHECKED : KBBQ_APPLY_FAST;
    auto launch = [&](int b, int64_t m) {
        uint8_t* d = (uint8_t*)c->stage_dev[b];
        return kbbq_apply_dev(c, d, d + plane, (const uint32_t*)(d + 3 * plane), m, pitch, R, Qt, S2, minscore, dl.p, mode, d + 2 * plane);
    };
    auto stage_in = [&](int b, int64_t lo, int64_t m) {
        uint8_t* h = (uint8_t*)c->stage_host[b];
        const size_t off = (size_t)lo * pitch, nb = (size_t)m * pitch;
        threaded_copy(h, seq + off, nb); threaded_copy(h + plane, qual + off, nb);
        memcpy(h + 3 * plane, meta + lo, (size_t)m * 4);
    };
    auto upload = [&](int b, int64_t m, hipStream_t st) -> int {        // seq + qual, then the sidecars (the output plane lies between)
        HIPCHK(hipMemcpyAsync(c->stage_dev[b], c->stage_host[b], 2 * plane, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync((uint8_t*)c->stage_dev[b] + 3 * plane, (uint8_t*)c->stage_host[b] + 3 * plane, (size_t)m * 4, hipMemcpyHostToDevice, st));
        return KBBQ_OK;
    };
    struct Out { int64_t lo, m; };
    Out pending[2] = {{0, 0}, {0, 0}};                                   // downloads in flight, by buffer
    auto stage_out = [&](int b) -> int {                                 // wait for buffer b's download and hand its rows to the caller
        if (!pending[b].m) return KBBQ_OK;
        HIPCHK(hipEventSynchronize(c->stage_down[b]));
        threaded_copy(qual_out + (size_t)pending[b].lo * pitch, (uint8_t*)c->stage_host[b] + 2 * plane, (size_t)pending[b].m * pitch);
        pending[b].m = 0;
        return KBBQ_OK;
    };
    // Order of a slab k in buffers b = k & 1: [host] wait for slab k - 2's download (stage_out: everything of slab k - 2 on these
    // buffers has then finished, kernel included) -> host copy in -> upload (upload stream) -> kernel (launch stream, behind the
    // upload's event) -> download (DOWNLOAD stream, behind the kernel's event).  Uploads and downloads are on different streams,
    // so slab k + 1 goes up while slab k's kernel runs and slab k's (or k - 1's) new qualities come back: both directions of
    // PCIe busy at once (with both on one in-order stream the order was strictly U k, K k, D k, U k + 1).
    int64_t k = 0;
    auto pipelined = [&]() -> int {
        for (int64_t lo = 0; lo < nreads; lo += (int64_t)slab, ++k) {
            const int b = (int)(k & 1);
            const int64_t m = std::min<int64_t>((int64_t)slab, nreads - lo);
            int r2 = stage_out(b);                                            // slab k - 2 (same buffers): downloaded -> the caller's rows
            if (r2) return r2;
            stage_in(b, lo, m);
            r2 = upload(b, m, c->stage_stream);
            if (r2) return r2;
            HIPCHK(hipEventRecord(c->stage_up[b], c->stage_stream));
            HIPCHK(hipStreamWaitEvent(c->stream, c->stage_up[b], 0));
            r2 = launch(b, m);
            if (r2) return r2;
            HIPCHK(hipEventRecord(c->stage_used[b], c->stream));
            HIPCHK(hipStreamWaitEvent(c->stage_down_stream, c->stage_used[b], 0));
            HIPCHK(hipMemcpyAsync((uint8_t*)c->stage_host[b] + 2 * plane, (uint8_t*)c->stage_dev[b] + 2 * plane, (size_t)m * pitch, hipMemcpyDeviceToHost, c->stage_down_stream));
            HIPCHK(hipEventRecord(c->stage_down[b], c->stage_down_stream));
            pending[b] = {lo, m};
        }
        return KBBQ_OK;
    };
    rc = pipelined();
    if (rc) return stage_drain(c, rc);
    rc = kbbq_ctx_status(c, nullptr);
    if (rc) {
        (void)stage_drain(c, rc);
        rc = first_offender(c, nreads, slab, [&](int64_t lo, int64_t m) {
            stage_in(0, lo, m);
            int r2 = upload(0, m, c->stream);
            return r2 ? r2 : launch(0, m);
        });
        return rc ? rc : fail(KBBQ_E_HIP, "kbbq_apply: a status reported by the pipelined run did not reproduce slab by slab");
    }
    for (int b = 0; b < 2; ++b) { rc = stage_out((int)((k + b) & 1)); if (rc) return stage_drain(c, rc); }     // the older of the two first
    return KBBQ_OK;
}

} // extern "C"
