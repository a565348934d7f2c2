// Bounded experiment (round 4): the 381-bit Montgomery product on 13 limbs of 30 bits (radix 2^390), next to the 14 x 28-bit form
// of tools/fp28_mul_ubench.hip and fe_mul (12 x 32 bits + a carry word).  A limb product is < 2^60: the <= 13 operand terms
// of a column sum to < 2^63.7 — they fit the 64-bit accumulator of v_mad_u64_u32 without a carry word, the 13 reduction terms
// of the same column do not fit on top of them.  So a column is accumulated in two halves: operand terms, the part above 30 bits
// set aside, reduction terms on the low 30 bits, both carries into the next column.  338 multiply-adds + 13 v_mul_lo instead of
// 392 + 14 (28-bit limbs) or 288 + 12 + 288 v_addc (32-bit limbs).  Inputs < 2^385 give a result < 2p: no final subtraction.
// Plain C++ (the compiler emits v_mad_u64_u32 for acc += (u64)a * b): no asm statements, so no s_nop padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../masp_amd/csrc/device/field.hpp"
using namespace masp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Fp30 { uint32_t v[13]; };
__device__ constexpr uint32_t P30[13] = {0x3fffaaabu, 0x27fbffffu, 0x153ffffbu, 0x2affffacu, 0x30f6241eu, 0x034a83dau, 0x112bf673u,
                                         0x12e13ce1u, 0x2cd76477u, 0x1ed90d2eu, 0x29a4b1bau, 0x3a8e5ff9u, 0x001a0111u};
static constexpr uint32_t INV30 = 0x3ffcfffdu, MASK30 = 0x3fffffffu;

template <int K, int I, int END>
__device__ __forceinline__ void col_ab(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
    if constexpr (I < END) {
        acc += (uint64_t)a[I] * b[K - I];
        col_ab<K, I + 1, END>(acc, a, b);
    }
}
template <int K, int I, int END>
__device__ __forceinline__ void col_mp(uint64_t& acc, const uint32_t* m) {
    if constexpr (I < END) {
        acc += (uint64_t)m[I] * P30[K - I];
        col_mp<K, I + 1, END>(acc, m);
    }
}
template <int K>
__device__ __forceinline__ void cols30(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* r) {
    if constexpr (K < 13) {
        col_ab<K, 0, K + 1>(acc, a, b);
        const uint64_t hi = acc >> 30;
        acc &= MASK30;
        col_mp<K, 0, K>(acc, m);
        m[K] = ((uint32_t)acc * INV30) & MASK30;
        acc += (uint64_t)m[K] * P30[0];
        acc = (acc >> 30) + hi;
        cols30<K + 1>(acc, a, b, m, r);
    } else if constexpr (K < 25) {
        col_ab<K, K - 12, 13>(acc, a, b);
        const uint64_t hi = acc >> 30;
        acc &= MASK30;
        col_mp<K, K - 12, 13>(acc, m);
        r[K - 13] = (uint32_t)acc & MASK30;
        acc = (acc >> 30) + hi;
        cols30<K + 1>(acc, a, b, m, r);
    }
}
__device__ __forceinline__ Fp30 fp30_mul(const Fp30& a, const Fp30& b) {
    uint32_t m[13];
    Fp30 r;
    uint64_t acc = 0;
    cols30<0>(acc, a.v, b.v, m, r.v);
    r.v[12] = (uint32_t)acc;
    return r;
}
// the square: operand terms a_i a_j (i < j) once with a doubled operand (2 a_i < 2^31: a column's <= 7 terms still < 2^64)
template <int K, int I>
__device__ __forceinline__ void col_sq(uint64_t& acc, const uint32_t* a, const uint32_t* a2) {
    constexpr int J = K - I;
    if constexpr (I <= 12 && J >= 0 && I < J) {
        if constexpr (J <= 12) acc += (uint64_t)a2[I] * a[J];
        col_sq<K, I + 1>(acc, a, a2);
    } else if constexpr (I == J && I <= 12) {
        acc += (uint64_t)a[I] * a[I];
    }
}
template <int K>
__device__ __forceinline__ void cols30_sq(uint64_t& acc, const uint32_t* a, const uint32_t* a2, uint32_t* m, uint32_t* r) {
    if constexpr (K < 13) {
        col_sq<K, 0>(acc, a, a2);
        const uint64_t hi = acc >> 30;
        acc &= MASK30;
        col_mp<K, 0, K>(acc, m);
        m[K] = ((uint32_t)acc * INV30) & MASK30;
        acc += (uint64_t)m[K] * P30[0];
        acc = (acc >> 30) + hi;
        cols30_sq<K + 1>(acc, a, a2, m, r);
    } else if constexpr (K < 25) {
        col_sq<K, K - 12>(acc, a, a2);
        const uint64_t hi = acc >> 30;
        acc &= MASK30;
        col_mp<K, K - 12, 13>(acc, m);
        r[K - 13] = (uint32_t)acc & MASK30;
        acc = (acc >> 30) + hi;
        cols30_sq<K + 1>(acc, a, a2, m, r);
    }
}
__device__ __forceinline__ Fp30 fp30_sqr(const Fp30& a) {
    uint32_t m[13], a2[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) a2[i] = a.v[i] << 1;
    Fp30 r;
    uint64_t acc = 0;
    cols30_sq<0>(acc, a.v, a2, m, r.v);
    r.v[12] = (uint32_t)acc;
    return r;
}

__global__ void k_mul30(Fp30* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp30 a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fp30_mul(a, b); b = fp30_mul(b, a); }
    data[t] = a;
    data[t].v[0] ^= b.v[0] & 0;
}
__global__ void k_sqr30(Fp30* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp30 a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fp30_sqr(a); b = fp30_sqr(b); }
    data[t] = a;
    data[t].v[0] ^= b.v[0] & 0;
}
__global__ void k_mul32(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    data[t] = fe_add(a, b);
}
__global__ void k_sqr32(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_sqr(a); b = fe_sqr(b); }
    data[t] = fe_add(a, b);
}
__global__ void k_once(const Fp30* x, Fp30* y, Fp30* ysq) {
    y[threadIdx.x] = fp30_mul(x[2 * threadIdx.x], x[2 * threadIdx.x + 1]);
    ysq[threadIdx.x] = fp30_sqr(x[2 * threadIdx.x]);
}

// ---- host-side check with schoolbook big integers (32-bit words, little endian)
static std::vector<uint32_t> P32 = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
static int cmp(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    for (int i = (int)a.size() - 1; i >= 0; --i) {
        uint32_t x = a[i], y = i < (int)b.size() ? b[i] : 0;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
static void sub_in(std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    int64_t br = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        int64_t d = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - br;
        br = d < 0;
        a[i] = (uint32_t)d;
    }
}
static std::vector<uint32_t> mod_p(std::vector<uint32_t> a) {
    std::vector<uint32_t> r(13, 0);
    for (int bit = (int)a.size() * 32 - 1; bit >= 0; --bit) {
        for (int i = 12; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
        r[0] = (r[0] << 1) | ((a[bit / 32] >> (bit % 32)) & 1);
        if (cmp(r, P32) >= 0) sub_in(r, P32);
    }
    r.resize(12);
    return r;
}
static std::vector<uint32_t> mul(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    std::vector<uint32_t> r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); ++i) {
        uint64_t c = 0;
        for (size_t j = 0; j < b.size(); ++j) {
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        r[i + b.size()] = (uint32_t)c;
    }
    return r;
}
static std::vector<uint32_t> from30(const Fp30& x) {
    std::vector<uint32_t> r(14, 0);
    for (int i = 0; i < 13; ++i) {
        int bit = 30 * i;
        uint64_t v = (uint64_t)x.v[i] << (bit % 32);
        uint64_t s = (uint64_t)r[bit / 32] + (uint32_t)v;
        r[bit / 32] = (uint32_t)s;
        uint64_t s2 = (uint64_t)r[bit / 32 + 1] + (uint32_t)(v >> 32) + (s >> 32);
        r[bit / 32 + 1] = (uint32_t)s2;
        if (bit / 32 + 2 < 14) r[bit / 32 + 2] += (uint32_t)(s2 >> 32);
    }
    return r;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8, threads = 256, n = blocks * threads, it = 200;
    std::vector<Fp30> h(n);
    // limbs < 2^30, top limb < 2^25: values < 2^385 (the bound the product allows), far above 2p for most
    for (int i = 0; i < n; ++i) for (int k = 0; k < 13; ++k) h[i].v[k] = ((uint32_t)(i * 2654435761u + k * 40503u + 77u) * 2246822519u & MASK30) >> (k == 12 ? 5 : 0);
    for (int k = 0; k < 12; ++k) { h[0].v[k] = MASK30; h[1].v[k] = MASK30; }   // an all-ones pair: the column-sum bound
    h[0].v[12] = h[1].v[12] = (1u << 25) - 1;
    Fp30* d; CHECK(hipMalloc(&d, n * sizeof(Fp30))); CHECK(hipMemcpy(d, h.data(), n * sizeof(Fp30), hipMemcpyHostToDevice));
    Fp30 *y, *ysq; CHECK(hipMalloc(&y, 64 * sizeof(Fp30))); CHECK(hipMalloc(&ysq, 64 * sizeof(Fp30)));
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, d, y, ysq);
    std::vector<Fp30> hy(64), hs(64);
    CHECK(hipMemcpy(hy.data(), y, 64 * sizeof(Fp30), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hs.data(), ysq, 64 * sizeof(Fp30), hipMemcpyDeviceToHost));
    int bad = 0;
    std::vector<uint32_t> R(14, 0); R[12] = 1u << 6;  // 2^390
    std::vector<uint32_t> P2 = P32; P2.push_back(0);
    { uint64_t c = 0; for (auto& w : P2) { uint64_t t = ((uint64_t)w << 1) | c; c = t >> 32; w = (uint32_t)t; } }   // 2p
    for (int t = 0; t < 64; ++t) {
        auto lhs = mod_p(mul(from30(hy[t]), R)), rhs = mod_p(mul(from30(h[2 * t]), from30(h[2 * t + 1])));
        if (lhs != rhs) { ++bad; continue; }
        auto lsq = mod_p(mul(from30(hs[t]), R)), rsq = mod_p(mul(from30(h[2 * t]), from30(h[2 * t])));
        if (lsq != rsq) ++bad;
        for (int k = 0; k < 13; ++k) if (hy[t].v[k] > MASK30 || hs[t].v[k] > MASK30) ++bad;
        if (cmp(from30(hy[t]), P2) >= 0 || cmp(from30(hs[t]), P2) >= 0) ++bad;      // results < 2p
    }
    printf("check: %d of 64 products / squares wrong\n", bad);
    auto time_ms = [&](auto f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mul30, dim3(blocks), dim3(threads), 0, 0, d, it); });
    printf("Fp product, 13 x 30-bit limbs, split columns:  %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_sqr30, dim3(blocks), dim3(threads), 0, 0, d, it); });
    printf("Fp square,  13 x 30-bit limbs:                 %8.3f ms  %8.2f Gsqr/s\n", ms, (double)n * it * 2 / ms / 1e6);
    Fp* d32; CHECK(hipMalloc(&d32, n * sizeof(Fp))); CHECK(hipMemset(d32, 1, n * sizeof(Fp)));
    ms = time_ms([&] { hipLaunchKernelGGL(k_mul32, dim3(blocks), dim3(threads), 0, 0, d32, it); });
    printf("Fp fe_mul,  12 x 32-bit limbs:                 %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_sqr32, dim3(blocks), dim3(threads), 0, 0, d32, it); });
    printf("Fp fe_sqr,  12 x 32-bit limbs:                 %8.3f ms  %8.2f Gsqr/s\n", ms, (double)n * it * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mul30, dim3(1), dim3(64), 0, 0, d, 2000); });
    printf("30-bit form, single-wave latency: %.3f us per product\n", ms * 1e3 / 4000);
    return bad != 0;
}
