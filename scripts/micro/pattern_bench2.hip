// Micro-benchmark 2: does mixing loads and stores in ONE wave's instruction stream cost bandwidth?
// Same planes as pattern_bench.hip (2 read + 1 written, 16 bytes per lane), persistent 10-KiB-block traversal.
//   readonly   : three planes read, nothing written (the K1 shape)
//   writeonly  : one plane written
//   burst<N>   : N chunks loaded back to back, then N results stored back to back (coarser read / write interleave)
//   split      : 2/3 of the waves only load (planes a, b), 1/3 only store (plane c): no wave mixes the two
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o pattern_bench2 pattern_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint4 x4(uint4 x, uint4 y) { return make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w); }

__global__ __launch_bounds__(512) void readonly(const uint4* a, const uint4* b, const uint4* c, uint4* sink, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        uint4 x = a[base], y = b[base], z = c[base];
        for (int s = 0; s < blk; ++s) {
            uint4 nx = x, ny = y, nz = z;
            if (s + 1 < blk) { nx = a[base + (s + 1) * 64]; ny = b[base + (s + 1) * 64]; nz = c[base + (s + 1) * 64]; }
            acc = x4(acc, x4(x, x4(y, z)));
            x = nx; y = ny; z = nz;
        }
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void writeonly(uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        for (int s = 0; s < blk; ++s) c[base + s * 64] = make_uint4((unsigned)B, s, lane, 7);
    }
}

template <int N>
__global__ __launch_bounds__(1024) void burst(const uint4* a, const uint4* b, uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        uint4 x[N], y[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { x[k] = a[base + k * 64]; y[k] = b[base + k * 64]; }
        for (int s = 0; s < blk; s += N) {
            uint4 r[N];
#pragma unroll
            for (int k = 0; k < N; ++k) r[k] = x4(x[k], y[k]);
            if (s + N < blk) {
#pragma unroll
                for (int k = 0; k < N; ++k) { x[k] = a[base + (s + N + k) * 64]; y[k] = b[base + (s + N + k) * 64]; }
            }
#pragma unroll
            for (int k = 0; k < N; ++k) c[base + (s + k) * 64] = r[k];
        }
    }
}

template <int NL, int NS>
__global__ __launch_bounds__((NL + NS) * 64) void split(const uint4* a, const uint4* b, uint4* c, uint4* sink, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;         // NL loader waves, NS storer waves
    const long long nblocks = nkib / blk;
    if (wave < NL) {
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (long long B = (long long)blockIdx.x * NL + wave; B < nblocks; B += (long long)gridDim.x * NL) {
            const long long base = B * blk * 64 + lane;
            uint4 x = a[base], y = b[base];
            for (int s = 0; s < blk; ++s) {
                uint4 nx = x, ny = y;
                if (s + 1 < blk) { nx = a[base + (s + 1) * 64]; ny = b[base + (s + 1) * 64]; }
                acc = x4(acc, x4(x, y));
                x = nx; y = ny;
            }
        }
        if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
    } else {
        for (long long B = (long long)blockIdx.x * NS + (wave - NL); B < nblocks; B += (long long)gridDim.x * NS) {
            const long long base = B * blk * 64 + lane;
            for (int s = 0; s < blk; ++s) c[base + s * 64] = make_uint4((unsigned)B, s, lane, 7);
        }
    }
}

int main()
{
    const long long nkib = 20000000LL * 160 / 1024 / 80 * 80;
    uint4 *a, *b, *c, *sink;
    CHECK(hipMalloc(&a, nkib * 1024)); CHECK(hipMalloc(&b, nkib * 1024)); CHECK(hipMalloc(&c, nkib * 1024)); CHECK(hipMalloc(&sink, 1 << 16));
    CHECK(hipMemset(a, 1, nkib * 1024)); CHECK(hipMemset(b, 2, nkib * 1024)); CHECK(hipMemset(c, 3, nkib * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    auto run = [&](const char* name, double planes, auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-52s %.3f ms  %.0f GB/s\n", name, ms, planes * nkib * 1024 / ms / 1e6); fflush(stdout);
    };
    char nm[128];
    for (int rep = 0; rep < 2; ++rep) {
        for (int thr : {256, 512, 768, 1024}) for (int per_cu : {1, 2}) {
            if (thr * per_cu > 2048) continue;
            snprintf(nm, sizeof nm, "mixed (burst<1>), %d WG x %d thr per CU", per_cu, thr);
            run(nm, 3, [&] { hipLaunchKernelGGL(burst<1>, dim3(cus * per_cu), dim3(thr), 0, 0, a, b, c, nkib, 10); });
        }
        run("readonly, 3 planes, 10 KiB, 2 x 512", 3, [&] { hipLaunchKernelGGL(readonly, dim3(cus * 2), dim3(512), 0, 0, a, b, c, sink, nkib, 10); });
        run("writeonly, 1 plane, 1 x 256", 1, [&] { hipLaunchKernelGGL(writeonly, dim3(cus), dim3(256), 0, 0, c, nkib, 10); });
        run("writeonly, 1 plane, 1 x 512", 1, [&] { hipLaunchKernelGGL(writeonly, dim3(cus), dim3(512), 0, 0, c, nkib, 10); });
        run("writeonly, 1 plane, 2 x 512", 1, [&] { hipLaunchKernelGGL(writeonly, dim3(cus * 2), dim3(512), 0, 0, c, nkib, 10); });
        run("split 8 load + 4 store waves, 1 WG per CU", 3, [&] { hipLaunchKernelGGL((split<8, 4>), dim3(cus), dim3(768), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 8 + 2", 3, [&] { hipLaunchKernelGGL((split<8, 2>), dim3(cus), dim3(640), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 8 + 1", 3, [&] { hipLaunchKernelGGL((split<8, 1>), dim3(cus), dim3(576), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 12 + 4", 3, [&] { hipLaunchKernelGGL((split<12, 4>), dim3(cus), dim3(1024), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 12 + 2", 3, [&] { hipLaunchKernelGGL((split<12, 2>), dim3(cus), dim3(896), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 6 + 2", 3, [&] { hipLaunchKernelGGL((split<6, 2>), dim3(cus), dim3(512), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 4 + 2", 3, [&] { hipLaunchKernelGGL((split<4, 2>), dim3(cus), dim3(384), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 4 + 4", 3, [&] { hipLaunchKernelGGL((split<4, 4>), dim3(cus), dim3(512), 0, 0, a, b, c, sink, nkib, 10); });
        run("split 8 + 8", 3, [&] { hipLaunchKernelGGL((split<8, 8>), dim3(cus), dim3(1024), 0, 0, a, b, c, sink, nkib, 10); });
    }
    return 0;
}
