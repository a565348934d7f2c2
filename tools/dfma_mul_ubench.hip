// Bounded experiment: a 384-bit Montgomery product built on the FP64 FMA pipe instead of v_mad_u64_u32.
// gfx950 issues v_fma_f64 at twice the rate of v_mad_u64_u32 (tools/ubench.hip: 37.7 against ~19 T lane-ops/s).  Scheme (Emmart,
// "Faster modular exponentiation using double precision floating point arithmetic on the GPU"): 8 limbs of 52 bits held as
// doubles; a 52 x 52 -> 104-bit limb product is split exactly by two FMAs
//     hi = fma(a, b, 2^104)  (round toward zero: the top 52 bits, at exponent 2^104)     lo = fma(a, b, 2^104 + 2^52 - hi) (the low 52)
// whose raw bit patterns are summed column by column with 64-bit INTEGER additions (the constants' contributions subtracted
// once per column).  Product-scanning multiplication (64 limb products) + Montgomery reduction by limbs (8 quotient digits, 64
// limb products) + carry propagation of the 16 columns into 52-bit limbs + conversion of the result back to doubles.  Table
// rows could be stored pre-converted (capacity is free), so the conversion INTO doubles is not charged; the conversion of the
// result is (the next product needs it).
// What is timed is the instruction mix of a full product in a dependent chain per lane, 2 chains per lane like k_femul of
// tools/ubench.hip; the value chain is not checked against the integer product here (a go would have to earn that first).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dfma_mul_ubench.hip -o tools/_build/dfma_mul_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../masp_amd/csrc/device/field.hpp"
using namespace masp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct D8 {
    double v[8];
};
__device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }

// C2 = 2^104 + 2^52 as a double.  The hi FMA must truncate: on gfx950 the rounding mode is a field of the MODE register
// (one s_setreg_b32 per kernel), not an instruction variant, so the same v_fma_f64 is timed here in the default mode
__device__ __forceinline__ void limb_mul(double a, double b, uint64_t& col_lo, uint64_t& col_hi) {
    const double C1 = 20282409603651670423947251286016.0;             // 2^104
    const double C2 = 20282409603651674927546878656512.0;             // 2^104 + 2^52
    const double hi = __builtin_fma(a, b, C1);
    const double lo = __builtin_fma(a, b, C2 - hi);
    col_hi += bits(hi);
    col_lo += bits(lo);
}
// 8 x 52-bit limbs: r = a * b / 2^416 mod p (Montgomery, radix 2^52); np[] the modulus limbs, ninv = -p^-1 mod 2^52 as doubles
__device__ __forceinline__ D8 dfma_mont_mul(const D8& a, const D8& b, const double* np, double ninv) {
    uint64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) col[k] = 0;
    // product scanning: column k collects the low halves of its limb products and the high halves of column k - 1's
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) limb_mul(a.v[i], b.v[j], col[i + j], col[i + j + 1]);
    // Montgomery reduction, one limb at a time: q = (col_k mod 2^52) * ninv mod 2^52 ; col += q * p << (52 k)
    const uint64_t M52 = (1ull << 52) - 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // (the biases of the FMA constants are removed from a column before it is used: one subtraction per column)
        const uint64_t t = (col[k] - (uint64_t)(k + 1) * 0x4330000000000000ull) & M52;
        const double td = (double)(long long)t;
        // low 52 bits of t * ninv: one more split product
        uint64_t ql = 0, qh = 0;
        limb_mul(td, ninv, ql, qh);
        const double q = (double)(long long)(ql & M52);
#pragma unroll
        for (int j = 0; j < 8; ++j) limb_mul(q, np[j], col[k + j], col[k + j + 1]);
        col[k + 1] += col[k] >> 52;  // carry
    }
    // carry propagation of the upper columns into 52-bit limbs, back to doubles
    D8 r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint64_t v = col[8 + k] - (uint64_t)(16 - k) * 0x4330000000000000ull + carry;
        r.v[k] = (double)(long long)(v & M52);
        carry = v >> 52;
    }
    return r;
}

__global__ void __launch_bounds__(256, 2) k_dfma(double* data, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    D8 a, b;
    double np[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a.v[k] = data[(size_t)t * 16 + k];
        b.v[k] = data[(size_t)t * 16 + 8 + k];
        np[k] = 4503599627370495.0 - 12345.0 * k;
    }
    const double ninv = 1234567890123.0;
    for (int i = 0; i < iters; ++i) {
        a = dfma_mont_mul(a, b, np, ninv);
        b = dfma_mont_mul(b, a, np, ninv);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) data[(size_t)t * 16 + k] = a.v[k] + b.v[k];
}
__global__ void k_femul(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    data[t] = fe_add(a, b);
}

template <class F> static float time_ms(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8, threads = 256, n = blocks * threads, it = 200;
    std::vector<double> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761ull + 12345) & ((1ull << 52) - 1));
    double* d; CHECK(hipMalloc(&d, h.size() * 8)); CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_dfma, dim3(blocks), dim3(threads), 0, 0, d, it); });
    const double dfma = (double)n * it * 2 / ms / 1e6;
    std::vector<Fp> hf(n);
    for (int i = 0; i < n; ++i) for (int k = 0; k < 12; ++k) hf[i].v[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 11 ? 0x0fffffff : 0xffffffff);
    Fp* df; CHECK(hipMalloc(&df, n * sizeof(Fp))); CHECK(hipMemcpy(df, hf.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    float ms2 = time_ms([&] { hipLaunchKernelGGL(k_femul, dim3(blocks), dim3(threads), 0, 0, df, it); });
    const double imad = (double)n * it * 2 / ms2 / 1e6;
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    printf("Fp product on v_mad_u64_u32 (device/field.hpp fe_mul, 12 x 32-bit limbs):  %8.3f ms  %7.2f G products/s\n", ms2, imad);
    printf("Fp product on v_fma_f64     (8 x 52-bit limbs, result converted back):     %8.3f ms  %7.2f G products/s\n", ms, dfma);
    printf("verdict: %s (go needs >= 15 %% more products/s than the integer path: %.2f)\n", dfma >= 1.15 * imad ? "GO" : "NO-GO", 1.15 * imad);
    return 0;
}
