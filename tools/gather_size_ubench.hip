// How fast do random 128-byte table rows arrive as the table grows?  (Round 4: a wNAF recoding of the scalars would cut the
// bucket additions by 11...17 % but needs a window table per BIT position — 16x the rows, ~17 GB for the Spend circuit instead of
// 1.4 GB.)  One 96-byte row (6 x 16 B) per lane and iteration at a random index, tables of 0.5 ... 48 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256) k_gather(const uint8_t* __restrict__ tab, uint64_t nrows, int iters, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint64_t r = (((uint64_t)mix(x) << 32) | mix(x ^ 0x9e3779b9u)) % nrows;
        const uint4* row = reinterpret_cast<const uint4*>(tab + r * 128);
#pragma unroll
        for (int k = 0; k < 6; ++k) { uint4 v = row[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[t] = acc;
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 16, threads = 256, iters = 64;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    const double gib[] = {0.5, 1.5, 3, 8, 17, 32, 48};
    for (double g : gib) {
        const size_t bytes = (size_t)(g * (1ull << 30));
        uint8_t* tab;
        if (hipMalloc(&tab, bytes) != hipSuccess) { printf("%.1f GiB: hipMalloc failed\n", g); continue; }
        CHECK(hipMemset(tab, 1, bytes));
        const uint64_t nrows = bytes / 128;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, tab, nrows, iters, out);
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, tab, nrows, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double rows = 3.0 * blocks * threads * iters;
        printf("table %5.1f GiB: %7.2f G rows/s (%.2f TB/s of 128-byte lines)\n", g, rows / ms / 1e6, rows * 128 / ms / 1e9);
        CHECK(hipFree(tab));
    }
    return 0;
}
