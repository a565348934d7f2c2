// Why do random 128-byte rows arrive 4x slower from tables above ~4 GiB (profiles/r04_gather_rate_vs_table_size_ubench.txt), and can it be
// helped?  Three arms over the same gather loop:
//   malloc     hipMalloc'ed table, every workgroup gathers from all of it                      (the round-4 measurement, finer sizes)
//   xcd        the same table, workgroup w gathers only from eighth (w mod 8) of it: workgroups go to the 8 XCDs round-robin, each
//              XCD has its own L2 TLB — if the cliff is the reach of that TLB, an eighth per XCD brings the rate back
//   vmm<k>     table mapped through the virtual-memory API at a 2^k-byte aligned address (hipMemAddressReserve / hipMemCreate / hipMemMap):
//              the driver writes page-table fragments as large as the alignment of virtual AND physical address allows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256) k_gather(const uint8_t* __restrict__ tab, uint64_t nrows, int iters, uint32_t regions, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u, acc = 0;
    const uint64_t per = nrows / regions, base = (uint64_t)(blockIdx.x % regions) * per;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint64_t r = base + (((uint64_t)mix(x) << 32) | mix(x ^ 0x9e3779b9u)) % per;
        const uint4* row = reinterpret_cast<const uint4*>(tab + r * 128);
#pragma unroll
        for (int k = 0; k < 6; ++k) { uint4 v = row[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[t] = acc;
}
static void run(const char* arm, double g, const uint8_t* tab, size_t bytes, uint32_t regions, uint32_t* out, int blocks) {
    const int threads = 256, iters = 64;
    const uint64_t nrows = bytes / 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, tab, nrows, iters, regions, out);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, tab, nrows, iters, regions, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rows = 3.0 * blocks * threads * iters;
    printf("%-8s table %5.1f GiB: %7.2f G rows/s (%.2f TB/s of 128-byte lines)\n", arm, g, rows / ms / 1e6, rows * 128 / ms / 1e9);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 16;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const double gib[] = {3, 4, 5, 6, 8, 17};
    for (double g : gib) {
        const size_t bytes = (size_t)(g * (1ull << 30));
        uint8_t* tab;
        if (hipMalloc(&tab, bytes) != hipSuccess) { printf("%.1f GiB: hipMalloc failed\n", g); continue; }
        CHECK(hipMemset(tab, 1, bytes));
        printf("(hipMalloc gave %p)\n", (void*)tab);
        run("malloc", g, tab, bytes, 1, out, blocks);
        if (g >= 8) run("xcd", g, tab, bytes, 8, out, blocks);
        CHECK(hipFree(tab));
    }
    // the virtual-memory API
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("vmm granularity (recommended) %zu\n", gran);
    for (int alog : {21, 30, 31, 33}) {
        for (double g : {8.0, 16.0}) {
            const size_t bytes = (size_t)(g * (1ull << 30));
            void* va = nullptr;
            if (hipMemAddressReserve(&va, bytes, (size_t)1 << alog, nullptr, 0) != hipSuccess) { printf("reserve 2^%d failed\n", alog); (void)hipGetLastError(); continue; }
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); (void)hipGetLastError(); hipMemAddressFree(va, bytes); continue; }
            CHECK(hipMemMap(va, bytes, 0, h, 0));
            hipMemAccessDesc acc;
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CHECK(hipMemSetAccess(va, bytes, &acc, 1));
            CHECK(hipMemset(va, 1, bytes));
            char arm[32];
            snprintf(arm, sizeof arm, "vmm%d", alog);
            printf("(reserved %p)\n", va);
            run(arm, g, (const uint8_t*)va, bytes, 1, out, blocks);
            CHECK(hipMemUnmap(va, bytes));
            CHECK(hipMemRelease(h));
            CHECK(hipMemAddressFree(va, bytes));
        }
    }
    return 0;
}
