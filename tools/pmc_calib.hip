// FETCH_SIZE calibration for the access pattern of k_msm_accumulate: every lane gathers ONE window-table row (96 bytes of
// payload = 6 x global_load_dwordx4) at a random row index from a table far larger than the 256 MiB Infinity Cache.
// MI355X_MICROARCH.md (§HBM): on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced stream; other access
// patterns must be calibrated on a known byte count — this is that known byte count.  Two kernels: rows at a 128-byte
// stride (the table layout since round 2: exactly one 128-B line per row) and at a 96-byte stride (round 1: rows straddle
// lines, 1.5 lines per row on average).  Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/_build/pmc_calib
// Run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- tools/_build/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int STRIDE>
__global__ void __launch_bounds__(64) k_calib_gather(const uint8_t* __restrict__ tab, uint32_t nrows, int iters, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint4* row = reinterpret_cast<const uint4*>(tab + (size_t)(x % nrows) * STRIDE);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uint4 v = row[k];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[t] = acc;
}

int main() {
    const size_t bytes = (size_t)3 << 30;                       // 3 GiB table: 12x the Infinity Cache
    const uint32_t lanes = 1u << 20;
    const int iters = 16;
    uint8_t* tab;
    uint32_t* out;
    CHECK(hipMalloc(&tab, bytes));
    CHECK(hipMemset(tab, 1, bytes));
    CHECK(hipMalloc(&out, lanes * 4));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_gather<128>, dim3(lanes / 64), dim3(64), 0, 0, tab, (uint32_t)(bytes / 128), iters, out);
        hipLaunchKernelGGL(k_calib_gather<96>, dim3(lanes / 64), dim3(64), 0, 0, tab, (uint32_t)(bytes / 96), iters, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("rows_gathered_per_launch %llu payload_bytes %llu lines128_stride128 %llu expected_lines_stride96 %.0f\n",
           (unsigned long long)lanes * iters, (unsigned long long)lanes * iters * 96, (unsigned long long)lanes * iters,
           (double)lanes * iters * 1.5);
    return 0;
}
