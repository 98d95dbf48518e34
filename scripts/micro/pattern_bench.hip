// Micro-benchmark: which traversal of three byte planes (2 read + 1 written, 16 bytes per lane)
// reaches the HBM rate?  Build: hipcc --offload-arch=gfx950 -O3 -o pattern_bench pattern_bench.hip
//   blocked : a wave owns `blk` consecutive KiB of each plane and walks them 1 KiB per step
//             (the K1/K2 read-block traversal: 64 reads x 160 B = 10 KiB)
//   flat    : at step t wave w touches KiB number t * nwaves + w (all waves sweep memory together)
//   tile    : non-persistent, one workgroup per tile of `blk` KiB per wave, all loads issued up front
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void blocked(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        uint4 x = a[base], y = b[base];
        for (int s = 0; s < blk; ++s) {
            uint4 nx = x, ny = y;
            if (s + 1 < blk) { nx = a[base + (s + 1) * 64]; ny = b[base + (s + 1) * 64]; }
            c[base + s * 64] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
            x = nx; y = ny;
        }
    }
}

// the blocked traversal with non-temporal (streaming) loads and / or stores
template <bool NTL, bool NTS>
__global__ __launch_bounds__(512) void blocked_nt(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    auto ld = [](const uint4* p) {
        if (!NTL) return *p;
        const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        uint4 x = ld(a + base), y = ld(b + base);
        for (int s = 0; s < blk; ++s) {
            uint4 nx = x, ny = y;
            if (s + 1 < blk) { nx = ld(a + base + (s + 1) * 64); ny = ld(b + base + (s + 1) * 64); }
            const uint4 r = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
            if (NTS) { v4u v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w; __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(c + base + s * 64)); }
            else c[base + s * 64] = r;
            x = nx; y = ny;
        }
    }
}

__global__ __launch_bounds__(512) void flat(const uint4* a, const uint4* b, uint4* c, long long nkib, int)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long stride = (long long)gridDim.x * nw;
    long long k = (long long)blockIdx.x * nw + wave;
    if (k >= nkib) return;
    uint4 x = a[k * 64 + lane], y = b[k * 64 + lane];
    for (; k < nkib; k += stride) {
        uint4 nx = x, ny = y;
        if (k + stride < nkib) { nx = a[(k + stride) * 64 + lane]; ny = b[(k + stride) * 64 + lane]; }
        c[k * 64 + lane] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
        x = nx; y = ny;
    }
}

// workgroup-contiguous: at step t workgroup g touches KiB [ (t * G + g) * nw, +nw ): consecutive waves of a
// workgroup read consecutive KiB, consecutive workgroups consecutive 8-KiB pieces
__global__ __launch_bounds__(512) void flat_wg(const uint4* a, const uint4* b, uint4* c, long long nkib, int)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long stride = (long long)gridDim.x * nw;
    long long k = (long long)blockIdx.x * nw + wave;
    for (; k < nkib; k += stride) {
        const uint4 x = a[k * 64 + lane], y = b[k * 64 + lane];
        c[k * 64 + lane] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    }
}

// persistent waves that take the next `blk` KiB from a global counter (in-order hand-out, no drift)
__global__ __launch_bounds__(512) void queue(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk, unsigned long long* counter)
{
    const int lane = threadIdx.x & 63;
    const long long nblocks = nkib / blk;
    for (;;) {
        unsigned long long B = 0;
        if (lane == 0) B = atomicAdd(counter, 1ull);
        B = __shfl(B, 0);
        if ((long long)B >= nblocks) break;
        const long long base = (long long)B * blk * 64 + lane;
        uint4 x = a[base], y = b[base];
        for (int s = 0; s < blk; ++s) {
            uint4 nx = x, ny = y;
            if (s + 1 < blk) { nx = a[base + (s + 1) * 64]; ny = b[base + (s + 1) * 64]; }
            c[base + s * 64] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
            x = nx; y = ny;
        }
    }
}

__global__ __launch_bounds__(256) void tile(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk)
{
    // one wave per `blk` KiB (blk <= 8): all loads first, then all stores
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long B = (long long)blockIdx.x * nw + wave;
    if (B * blk >= nkib) return;
    const long long base = B * blk * 64 + lane;
    uint4 x[8], y[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < blk) { x[s] = a[base + s * 64]; y[s] = b[base + s * 64]; }
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < blk) c[base + s * 64] = make_uint4(x[s].x ^ y[s].x, x[s].y ^ y[s].y, x[s].z ^ y[s].z, x[s].w ^ y[s].w);
}

// tile kernel with the tile order scrambled (tile = (wave id * mult) mod ntiles): same work per wave, no moving window
__global__ __launch_bounds__(256) void tile_scrambled(const uint4* a, const uint4* b, uint4* c, long long nkib, long long mult)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    long long B = (long long)blockIdx.x * nw + wave;
    if (B >= nkib) return;
    B = (B * mult) % nkib;
    const long long base = B * 64 + lane;
    const uint4 x = a[base], y = b[base];
    c[base] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
}

// persistent flat traversal with the waves of a workgroup kept in step by a barrier
__global__ __launch_bounds__(512) void flat_sync(const uint4* a, const uint4* b, uint4* c, long long nkib, int)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long stride = (long long)gridDim.x * nw;
    for (long long k0 = (long long)blockIdx.x * nw; k0 < nkib; k0 += stride) {
        const long long k = k0 + wave;
        if (k < nkib) {
            const uint4 x = a[k * 64 + lane], y = b[k * 64 + lane];
            c[k * 64 + lane] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
        }
        __syncthreads();
    }
}

// non-persistent: one wave per `parts` KiB taken far apart (B, B + n/parts, ...): several loop iterations per
// wave but no contiguous footprint beyond 1 KiB
__global__ __launch_bounds__(256) void tile_far(const uint4* a, const uint4* b, uint4* c, long long nkib, int parts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long per = nkib / parts;
    const long long B = (long long)blockIdx.x * nw + wave;
    if (B >= per) return;
    for (int s = 0; s < parts; ++s) {
        const long long base = (B + s * per) * 64 + lane;
        const uint4 x = a[base], y = b[base];
        c[base] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    }
}

// non-persistent, one wave per `blk` contiguous KiB, one KiB at a time (load, load, store; no loads up front)
__global__ __launch_bounds__(256) void tile_seq(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long B = (long long)blockIdx.x * nw + wave;
    if (B * blk >= nkib) return;
    for (int s = 0; s < blk; ++s) {
        const long long base = (B * blk + s) * 64 + lane;
        const uint4 x = a[base], y = b[base];
        c[base] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    }
}

int main(int argc, char** argv)
{
    const long long nkib = 20000000LL * 160 / 1024 / 80 * 80;       // the 20 M x 160 B planes of scripts/time_kernels.py
    uint4 *a, *b, *c;
    CHECK(hipMalloc(&a, nkib * 1024)); CHECK(hipMalloc(&b, nkib * 1024)); CHECK(hipMalloc(&c, nkib * 1024));
    CHECK(hipMemset(a, 1, nkib * 1024)); CHECK(hipMemset(b, 2, nkib * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    auto run = [&](const char* name, auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, 3.0 * nkib * 1024 / ms / 1e6); fflush(stdout);
    };
    char nm[128];
    for (int per_cu : {2, 4}) for (int blk : {1, 2, 4, 10, 80}) {
        snprintf(nm, sizeof nm, "blocked blk=%d KiB, %d x 512 thr per CU", blk, per_cu);
        run(nm, [&] { hipLaunchKernelGGL(blocked, dim3(cus * per_cu), dim3(512), 0, 0, a, b, c, nkib, blk); });
    }
    run("blocked blk=10, non-temporal loads", [&] { hipLaunchKernelGGL((blocked_nt<true, false>), dim3(cus * 2), dim3(512), 0, 0, a, b, c, nkib, 10); });
    run("blocked blk=10, non-temporal stores", [&] { hipLaunchKernelGGL((blocked_nt<false, true>), dim3(cus * 2), dim3(512), 0, 0, a, b, c, nkib, 10); });
    run("blocked blk=10, non-temporal loads + stores", [&] { hipLaunchKernelGGL((blocked_nt<true, true>), dim3(cus * 2), dim3(512), 0, 0, a, b, c, nkib, 10); });
    run("blocked blk=10, plain (same template)", [&] { hipLaunchKernelGGL((blocked_nt<false, false>), dim3(cus * 2), dim3(512), 0, 0, a, b, c, nkib, 10); });
    // the same kernel launched NON-persistently: one wave per block, grid covers everything
    for (int thr : {256, 512}) for (int blk : {2, 10}) {
        snprintf(nm, sizeof nm, "blocked blk=%d KiB, non-persistent grid, %d thr", blk, thr);
        const long long wgs = (nkib / blk + thr / 64 - 1) / (thr / 64);
        run(nm, [&] { hipLaunchKernelGGL(blocked, dim3((unsigned)wgs), dim3(thr), 0, 0, a, b, c, nkib, blk); });
    }
    for (int per_cu : {2, 4}) {
        snprintf(nm, sizeof nm, "flat (prefetch 1), %d x 512 thr per CU", per_cu);
        run(nm, [&] { hipLaunchKernelGGL(flat, dim3(cus * per_cu), dim3(512), 0, 0, a, b, c, nkib, 0); });
        snprintf(nm, sizeof nm, "flat (no prefetch), %d x 512 thr per CU", per_cu);
        run(nm, [&] { hipLaunchKernelGGL(flat_wg, dim3(cus * per_cu), dim3(512), 0, 0, a, b, c, nkib, 0); });
    }
    // the blocked traversal launched slice by slice (back-to-back kernels): waves cannot drift further apart than a slice
    for (int slices : {4, 16, 64, 256}) {
        snprintf(nm, sizeof nm, "blocked blk=10 KiB in %d back-to-back slices", slices);
        const long long per = nkib / slices / 10 * 10;
        run(nm, [&] { for (int i = 0; i < slices; ++i) hipLaunchKernelGGL(blocked, dim3(cus * 2), dim3(512), 0, 0, a + i * per * 64, b + i * per * 64, c + i * per * 64, per, 10); });
    }
    unsigned long long* counter; CHECK(hipMalloc(&counter, 8));
    for (int per_cu : {2}) for (int blk : {10}) {
        snprintf(nm, sizeof nm, "queue: wave takes %d KiB, %d x 512 thr per CU", blk, per_cu);
        run(nm, [&] { CHECK(hipMemsetAsync(counter, 0, 8, 0)); hipLaunchKernelGGL(queue, dim3(cus * per_cu), dim3(512), 0, 0, a, b, c, nkib, blk, counter); });
    }
    for (long long mult : {1LL, 17LL, 4099LL, 1000003LL}) {
        snprintf(nm, sizeof nm, "tile 1 KiB, order scrambled x%lld", mult);
        run(nm, [&] { hipLaunchKernelGGL(tile_scrambled, dim3((unsigned)((nkib + 3) / 4)), dim3(256), 0, 0, a, b, c, nkib, mult); });
    }
    for (int per_cu : {2, 4}) {
        snprintf(nm, sizeof nm, "flat + barrier per step, %d x 512 thr per CU", per_cu);
        run(nm, [&] { hipLaunchKernelGGL(flat_sync, dim3(cus * per_cu), dim3(512), 0, 0, a, b, c, nkib, 0); });
    }
    for (int parts : {2, 4, 8}) {
        snprintf(nm, sizeof nm, "tile: wave takes %d far-apart KiB", parts);
        run(nm, [&] { hipLaunchKernelGGL(tile_far, dim3((unsigned)((nkib / parts + 3) / 4)), dim3(256), 0, 0, a, b, c, nkib, parts); });
    }
    for (int blk : {2, 4, 8}) {
        snprintf(nm, sizeof nm, "tile: wave per %d contiguous KiB, one at a time", blk);
        run(nm, [&] { hipLaunchKernelGGL(tile_seq, dim3((unsigned)((nkib / blk + 3) / 4)), dim3(256), 0, 0, a, b, c, nkib, blk); });
    }
    for (int blk : {1, 2, 4, 8}) {
        snprintf(nm, sizeof nm, "tile: wave per %d KiB, non-persistent", blk);
        const long long waves = nkib / blk;
        run(nm, [&] { hipLaunchKernelGGL(tile, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, a, b, c, nkib, blk); });
    }
    return 0;
}
