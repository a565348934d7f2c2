// Second arm of tools/gather_tlb_ubench.hip: a table mapped through HIP's virtual-memory API at an address aligned BY HAND (the reservation
// is one alignment unit longer than the table), as one physical allocation or as one allocation per alignment unit.  The amdgpu driver
// writes a page-directory entry as a 1 GiB translation when virtual and physical address are both 1 GiB-aligned and contiguous.
// usage: gather_vmm_ubench <GiB> <align_log2> <handles: 0 = one, 1 = one per 2^align_log2 bytes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
int main(int argc, char** argv) {
    const double g = argc > 1 ? atof(argv[1]) : 8.0;
    const int alog = argc > 2 ? atoi(argv[2]) : 30;
    const int per_unit = argc > 3 ? atoi(argv[3]) : 0;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 16;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const size_t unit = (size_t)1 << alog, bytes = ((size_t)(g * (1ull << 30)) + unit - 1) / unit * unit;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void* va = nullptr;
    CHECK(hipMemAddressReserve(&va, bytes + unit, 0, nullptr, 0));
    uint8_t* base = (uint8_t*)(((uintptr_t)va + unit - 1) & ~(uintptr_t)(unit - 1));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    if (per_unit) {
        for (size_t off = 0; off < bytes; off += unit) {
            hipMemGenericAllocationHandle_t h;
            CHECK(hipMemCreate(&h, unit, &prop, 0));
            CHECK(hipMemMap(base + off, unit, 0, h, 0));
            hs.push_back(h);
        }
    } else {
        hipMemGenericAllocationHandle_t h;
        CHECK(hipMemCreate(&h, bytes, &prop, 0));
        CHECK(hipMemMap(base, bytes, 0, h, 0));
        hs.push_back(h);
    }
    hipMemAccessDesc acc;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(base, bytes, &acc, 1));
    CHECK(hipMemset(base, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const uint64_t nrows = bytes / 128;
    const int iters = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, base, nrows, iters, out);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, base, nrows, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double rows = 3.0 * blocks * 256 * iters;
    printf("vmm align 2^%d %s: reserved %p mapped %p  table %5.1f GiB: %7.2f G rows/s (%.2f TB/s of 128-byte lines)\n", alog, per_unit ? "one handle per unit" : "one handle", va,
           (void*)base, bytes / 1073741824.0, rows / ms / 1e6, rows * 128 / ms / 1e9);
    return 0;
}
