// Is device memory that a process has allocated, used, FREED and allocated again as good as its first allocation?  (Round 6, the "slower
// second context": a prover context created after another one of the process — 175 GB of scratch with four slots — was closed proves
// 4 - 6 % slower, also on the plain host-to-host path, with every stream on a hardware queue of its own.)
// Random 128-byte rows gathered from a 2 GiB table (the size of the prover's window tables) and a 1 GiB-per-block streaming copy:
//   fresh process | after the process has allocated, touched and freed FILL_GB in 4 GiB pieces | ... and with FILL_GB allocated again around it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
static int blocks;
static uint32_t* out;
static void measure(const char* tag) {
    const size_t tb = (size_t)2 << 30, sb = (size_t)8 << 30;
    uint8_t *tab, *s0, *s1;
    CHECK(hipMalloc(&tab, tb)); CHECK(hipMalloc(&s0, sb)); CHECK(hipMalloc(&s1, sb));
    CHECK(hipMemset(tab, 1, tb)); CHECK(hipMemset(s0, 2, sb)); CHECK(hipMemset(s1, 3, sb));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, tb / 128, 64, out);
    hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4*)s0, (uint4*)s1, sb / 16);
    CHECK(hipDeviceSynchronize());
    float g_ms, s_ms;
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, tb / 128, 64, out);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&g_ms, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4*)s0, (uint4*)s1, sb / 16);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&s_ms, e0, e1);
    const double rows = 5.0 * blocks * 256 * 64;
    printf("%-58s gather %6.2f TB/s of 128-byte lines   stream copy %6.2f TB/s (read + write)   table at %p\n", tag, rows * 128 / g_ms / 1e9,
           5.0 * 2 * sb / s_ms / 1e9, (void*)tab);
    fflush(stdout);
    CHECK(hipFree(tab)); CHECK(hipFree(s0)); CHECK(hipFree(s1));
}
int main(int argc, char** argv) {
    const int fill_gb = argc > 1 ? atoi(argv[1]) : 176;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    blocks = p.multiProcessorCount * 16;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    measure("fresh process");
    measure("fresh process, again");
    for (int round = 1; round <= 2; ++round) {
        std::vector<uint8_t*> fill;
        // the shapes a prover context takes: a few tens of buffers between 1 and 34 GiB
        const size_t shapes[] = {34, 8, 2, 2, 1, 1};
        size_t got = 0;
        for (int i = 0; got < (size_t)fill_gb; ++i) {
            const size_t gb = shapes[i % 6];
            uint8_t* q;
            if (hipMalloc(&q, gb << 30) != hipSuccess) { (void)hipGetLastError(); break; }
            CHECK(hipMemset(q, i, gb << 30));
            fill.push_back(q);
            got += gb;
        }
        CHECK(hipDeviceSynchronize());
        char tag[128];
        snprintf(tag, sizeof tag, "round %d: with %zu GiB allocated next to it", round, got);
        measure(tag);
        for (uint8_t* q : fill) CHECK(hipFree(q));
        snprintf(tag, sizeof tag, "round %d: after those %zu GiB were freed", round, got);
        measure(tag);
    }
    return 0;
}
