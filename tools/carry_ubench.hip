// How a column of a Montgomery product should take its carries on gfx950: the multiply-add chain of a column
//     acc(64) += a_i * b_j   (v_mad_u64_u32, carry-out in an SGPR pair)      c2 += carry   (v_addc_co_u32)
// in different instruction orders, at full occupancy and on a lone wave.  Every pattern does 8 multiply-adds into ONE
// accumulator (a dependent chain, as in a column) and folds the 8 carry-outs into one carry word.
//   P0  mad, addc, mad, addc, ...                    (the product's form: the addc reads the vcc its mad has just written)
//   P1  4 x mad (carries to four SGPR pairs), 4 x addc, twice
//   P2  8 x mad (eight SGPR pairs), 8 x addc
//   P3  mads only (no carry word): the bare dependent chain
//   P4  two columns interleaved: mad A, mad B, addc A, addc B, ...   (two accumulators: 16 mads per round, reported per 8)
//   P5  mad, addc with the PREVIOUS mad's carry (one SGPR pair of slack), ...
// Build: hipcc --offload-arch=gfx950 -O3 tools/carry_ubench.hip -o tools/_build/carry_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int P>
__global__ void __launch_bounds__(256) k_pat(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x + 1, c = 0, c2 = 0;
    uint64_t acc = a, acc2 = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (P == 0) {
                asm volatile(
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    "v_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                    : "+v"(acc), "+v"(c) : "v"(a), "v"(b) : "vcc");
            } else if (P == 1) {
                asm volatile(
                    "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_mad_u64_u32 %0, s[22:23], %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, s[24:25], %2, %2, %0\n\tv_mad_u64_u32 %0, s[26:27], %3, %3, %0\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_addc_co_u32_e64 %1, s[24:25], 0, %1, s[24:25]\n\tv_addc_co_u32_e64 %1, s[26:27], 0, %1, s[26:27]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_mad_u64_u32 %0, s[22:23], %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, s[24:25], %2, %2, %0\n\tv_mad_u64_u32 %0, s[26:27], %3, %3, %0\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_addc_co_u32_e64 %1, s[24:25], 0, %1, s[24:25]\n\tv_addc_co_u32_e64 %1, s[26:27], 0, %1, s[26:27]\n\t"
                    : "+v"(acc), "+v"(c) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            } else if (P == 2) {
                asm volatile(
                    "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_mad_u64_u32 %0, s[22:23], %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, s[24:25], %2, %2, %0\n\tv_mad_u64_u32 %0, s[26:27], %3, %3, %0\n\t"
                    "v_mad_u64_u32 %0, s[28:29], %2, %3, %0\n\tv_mad_u64_u32 %0, s[30:31], %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, s[32:33], %2, %2, %0\n\tv_mad_u64_u32 %0, s[34:35], %3, %3, %0\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_addc_co_u32_e64 %1, s[24:25], 0, %1, s[24:25]\n\tv_addc_co_u32_e64 %1, s[26:27], 0, %1, s[26:27]\n\t"
                    "v_addc_co_u32_e64 %1, s[28:29], 0, %1, s[28:29]\n\tv_addc_co_u32_e64 %1, s[30:31], 0, %1, s[30:31]\n\t"
                    "v_addc_co_u32_e64 %1, s[32:33], 0, %1, s[32:33]\n\tv_addc_co_u32_e64 %1, s[34:35], 0, %1, s[34:35]\n\t"
                    : "+v"(acc), "+v"(c) : "v"(a), "v"(b)
                    : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
            } else if (P == 3) {
                asm volatile(
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\t"
                    "v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\t"
                    : "+v"(acc), "+v"(c) : "v"(a), "v"(b) : "vcc");
            } else if (P == 4) {
                asm volatile(
                    "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %4, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %5, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %4, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %5, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %4, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %5, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %4, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n\tv_mad_u64_u32 %2, s[22:23], %5, %5, %2\n\t"
                    "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\tv_addc_co_u32_e64 %3, s[22:23], 0, %3, s[22:23]\n\t"
                    : "+v"(acc), "+v"(c), "+v"(acc2), "+v"(c2) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
            } else if (P == 5) {
                asm volatile(
                    "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\t"
                    "v_mad_u64_u32 %0, s[22:23], %3, %2, %0\n\tv_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %2, %2, %0\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[22:23], %3, %3, %0\n\tv_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[22:23], %3, %2, %0\n\tv_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
                    "v_mad_u64_u32 %0, s[20:21], %2, %2, %0\n\tv_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    "v_mad_u64_u32 %0, s[22:23], %3, %3, %0\n\tv_addc_co_u32_e64 %1, s[20:21], 0, %1, s[20:21]\n\t"
                    "v_addc_co_u32_e64 %1, s[22:23], 0, %1, s[22:23]\n\t"
                    : "+v"(acc), "+v"(c) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ c ^ (uint32_t)acc2 ^ (uint32_t)(acc2 >> 32) ^ c2;
}

template <int P>
static void run(const char* name, int blocks, int threads, int iters, uint32_t* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_pat<P>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pat<P>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mads = (double)blocks * threads * iters * 64.0 * (P == 4 ? 2 : 1);
    const double waves_per_simd = (double)blocks * threads / 64 / 1024;
    // cycles of a SIMD per wave-level multiply-add (+ its carry) at 2.4 GHz nominal
    const double cyc = waves_per_simd >= 1 ? ms * 1e-3 * 2.4e9 / (mads / 64 / 1024)
                                           : ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * (P == 4 ? 2 : 1));  // a lone wave: its own cycles
    printf("%-44s %3.0f waves/SIMD  %8.3f ms  %8.1f G mad/s  %5.2f SIMD-cycles@2.4GHz per wave mad\n", name, waves_per_simd, ms, mads / ms * 1e-6, cyc);
}

int main() {
    uint32_t* d;
    CHECK(hipMalloc(&d, 64u << 20));
    for (int occ : {1, 2, 4, 8}) {
        const int blocks = 256 * occ, threads = 256, iters = 2000 / occ;
        run<3>("P3 mads only (dependent chain)", blocks, threads, iters, d);
        run<0>("P0 mad, addc (vcc), ...", blocks, threads, iters, d);
        run<5>("P5 mad, addc of the previous carry, ...", blocks, threads, iters, d);
        run<1>("P1 4 mads, 4 addcs", blocks, threads, iters, d);
        run<2>("P2 8 mads, 8 addcs", blocks, threads, iters, d);
        run<4>("P4 two columns interleaved", blocks, threads, iters, d);
    }
    // a lone wave per SIMD-less chip: one workgroup of one wave
    run<3>("lone wave: P3", 1, 64, 20000, d);
    run<0>("lone wave: P0", 1, 64, 20000, d);
    run<5>("lone wave: P5", 1, 64, 20000, d);
    run<1>("lone wave: P1", 1, 64, 20000, d);
    run<2>("lone wave: P2", 1, 64, 20000, d);
    run<4>("lone wave: P4", 1, 64, 20000, d);
    return 0;
}
