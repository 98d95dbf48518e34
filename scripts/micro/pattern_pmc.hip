// The traversals of pattern_bench*.hip that matter, one launch each (x3), for rocprofv3 --pmc passes:
//   tile1k    : non-persistent, one wave per KiB (the 6.0 TB/s pattern)
//   blocked   : persistent 10-KiB blocks, 2 x 512 threads per CU (the K2 traversal, 5.25 TB/s)
//   readonly / writeonly : the same traversal with three planes read / one written
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o pattern_pmc pattern_pmc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint4 x4(uint4 x, uint4 y) { return make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w); }

__global__ __launch_bounds__(256) void tile1k(const uint4* a, const uint4* b, uint4* c, long long nkib)
{
    const long long k = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nkib) return;
    const long long i = k * 64 + (threadIdx.x & 63);
    c[i] = x4(a[i], b[i]);
}

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
            c[base + s * 64] = x4(x, y);
            x = nx; y = ny;
        }
    }
}

__global__ __launch_bounds__(512) void readonly(const uint4* a, const uint4* b, const uint4* c, uint4* sink, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        for (int s = 0; s < blk; ++s) acc = x4(acc, x4(a[base + s * 64], x4(b[base + s * 64], c[base + s * 64])));
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}

__global__ __launch_bounds__(512) void writeonly(uint4* c, long long nkib, int blk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long nblocks = nkib / blk;
    for (long long B = (long long)blockIdx.x * nw + wave; B < nblocks; B += (long long)gridDim.x * nw) {
        const long long base = B * blk * 64 + lane;
        for (int s = 0; s < blk; ++s) c[base + s * 64] = make_uint4((unsigned)B, s, lane, 7);
    }
}

int main()
{
    const long long nkib = 20000000LL * 160 / 1024 / 80 * 80;
    uint4 *a, *b, *c, *sink;
    CHECK(hipMalloc(&a, nkib * 1024)); CHECK(hipMalloc(&b, nkib * 1024)); CHECK(hipMalloc(&c, nkib * 1024)); CHECK(hipMalloc(&sink, 1 << 16));
    CHECK(hipMemset(a, 1, nkib * 1024)); CHECK(hipMemset(b, 2, nkib * 1024)); CHECK(hipMemset(c, 3, nkib * 1024));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(tile1k, dim3((unsigned)((nkib + 3) / 4)), dim3(256), 0, 0, a, b, c, nkib);
        hipLaunchKernelGGL(blocked, dim3(cus * 2), dim3(512), 0, 0, a, b, c, nkib, 10);
        hipLaunchKernelGGL(readonly, dim3(cus * 2), dim3(512), 0, 0, a, b, c, sink, nkib, 10);
        hipLaunchKernelGGL(writeonly, dim3(cus * 2), dim3(512), 0, 0, c, nkib, 10);
    }
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
