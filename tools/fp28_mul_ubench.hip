// Bounded experiment: the 381-bit Montgomery product on 14 limbs of 28 bits (radix 2^392) instead of 12 limbs of 32.
// A limb product is < 2^56, so the 28 products of a column (operand and reduction terms) sum to < 2^61: the 64-bit column
// accumulator of v_mad_u64_u32 cannot overflow and the v_addc_co_u32 that follows EVERY multiply-add of fe_mul (40 % of its
// issue time) disappears — for 392 instead of 288 multiply-adds and a 64-bit shift per column.
// The kernel times a chain of products per lane over the whole chip (like tools/ubench.hip's "Fp fe_mul" row) and checks the
// result against fe_mul through the conversion  x_28form = x * 2^392 mod p.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../masp_amd/csrc/device/field.hpp"
using namespace masp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Fp28 { uint32_t v[14]; };
__device__ constexpr uint32_t P28[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                         0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
static constexpr uint32_t INV28 = 0xffcfffdu, MASK28 = 0x0fffffffu;

// (hipcc pads every asm statement with an s_nop: multiply-adds go four to a statement)
#define M28 "v_mad_u64_u32 %0, vcc, "
__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) {
    asm(M28 "%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mad2(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm(M28 "%1, %2, %0\n\t" M28 "%3, %4, %0" : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void mad4(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    asm(M28 "%1, %2, %0\n\t" M28 "%3, %4, %0\n\t" M28 "%5, %6, %0\n\t" M28 "%7, %8, %0"
        : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc");
}
__device__ __forceinline__ void mads(uint64_t& acc, uint32_t a, uint32_t k) {
    asm(M28 "%1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}
__device__ __forceinline__ void mads2(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) {
    asm(M28 "%1, %2, %0\n\t" M28 "%3, %4, %0" : "+v"(acc) : "v"(a0), "s"(k0), "v"(a1), "s"(k1) : "vcc");
}
__device__ __forceinline__ void mads4(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2, uint32_t a3, uint32_t k3) {
    asm(M28 "%1, %2, %0\n\t" M28 "%3, %4, %0\n\t" M28 "%5, %6, %0\n\t" M28 "%7, %8, %0"
        : "+v"(acc) : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3) : "vcc");
}
template <int K, int I, int END>
__device__ __forceinline__ void col_vv(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
    if constexpr (END - I >= 4) {
        mad4(acc, a[I], b[K - I], a[I + 1], b[K - I - 1], a[I + 2], b[K - I - 2], a[I + 3], b[K - I - 3]);
        col_vv<K, I + 4, END>(acc, a, b);
    } else if constexpr (END - I >= 2) {
        mad2(acc, a[I], b[K - I], a[I + 1], b[K - I - 1]);
        col_vv<K, I + 2, END>(acc, a, b);
    } else if constexpr (END - I == 1) {
        mad(acc, a[I], b[K - I]);
    }
}
template <int K, int I, int END>
__device__ __forceinline__ void col_vs(uint64_t& acc, const uint32_t* m) {
    if constexpr (END - I >= 4) {
        mads4(acc, m[I], P28[K - I], m[I + 1], P28[K - I - 1], m[I + 2], P28[K - I - 2], m[I + 3], P28[K - I - 3]);
        col_vs<K, I + 4, END>(acc, m);
    } else if constexpr (END - I >= 2) {
        mads2(acc, m[I], P28[K - I], m[I + 1], P28[K - I - 1]);
        col_vs<K, I + 2, END>(acc, m);
    } else if constexpr (END - I == 1) {
        mads(acc, m[I], P28[K - I]);
    }
}
template <int K>
__device__ __forceinline__ void cols(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* r) {
    if constexpr (K < 14) {
        col_vv<K, 0, K + 1>(acc, a, b);
        col_vs<K, 0, K>(acc, m);
        m[K] = ((uint32_t)acc * INV28) & MASK28;
        mads(acc, m[K], P28[0]);
        acc >>= 28;
        cols<K + 1>(acc, a, b, m, r);
    } else if constexpr (K < 27) {
        col_vv<K, K - 13, 14>(acc, a, b);
        col_vs<K, K - 13, 14>(acc, m);
        r[K - 14] = (uint32_t)acc & MASK28;
        acc >>= 28;
        cols<K + 1>(acc, a, b, m, r);
    }
}
// a, b < 2p (limbs < 2^28): a b / 2^392 mod p, result < 2p (no final subtraction: 4 p^2 / R + p < 2p)
__device__ __forceinline__ Fp28 fp28_mul(const Fp28& a, const Fp28& b) {
    uint32_t m[14];
    Fp28 r;
    uint64_t acc = 0;
    cols<0>(acc, a.v, b.v, m, r.v);
    r.v[13] = (uint32_t)acc;
    return r;
}
__global__ void k_mul28(Fp28* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp28 a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fp28_mul(a, b); b = fp28_mul(b, a); }
    data[t] = a;
    data[t].v[0] ^= b.v[0] & 0;  // (keep b alive)
}
__global__ void k_mul32(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    data[t] = fe_add(a, b);
}
// one product of each kind on the same integers, for the check on the host
__global__ void k_once(const Fp28* x28, Fp28* y28) { y28[threadIdx.x] = fp28_mul(x28[2 * threadIdx.x], x28[2 * threadIdx.x + 1]); }

typedef unsigned __int128 u128;
struct Big { std::vector<uint32_t> w; };  // little-endian 32-bit words, host-side schoolbook arithmetic for the check
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
static std::vector<uint32_t> mod_p(std::vector<uint32_t> a) {  // bitwise long division
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
static std::vector<uint32_t> from28(const Fp28& x) {
    std::vector<uint32_t> r(13, 0);
    for (int i = 0; i < 14; ++i) {
        int bit = 28 * i;
        uint64_t v = (uint64_t)x.v[i] << (bit % 32);
        r[bit / 32] += (uint32_t)v;  // limbs < 2^28 and disjoint bit ranges: no carries
        if (bit / 32 + 1 < 13) r[bit / 32 + 1] += (uint32_t)(v >> 32);
    }
    return r;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8, threads = 256, n = blocks * threads, it = 200;
    std::vector<Fp28> h(n);
    for (int i = 0; i < n; ++i) for (int k = 0; k < 14; ++k) h[i].v[k] = ((uint32_t)(i * 2654435761u + k * 40503u) & MASK28) >> (k == 13 ? 11 : 0);  // < 2^381 < p ... roughly: top limb < 2^17
    Fp28* d; CHECK(hipMalloc(&d, n * sizeof(Fp28))); CHECK(hipMemcpy(d, h.data(), n * sizeof(Fp28), hipMemcpyHostToDevice));
    // ---- check: y = a b / 2^392 mod p  <=>  y 2^392 = a b mod p
    Fp28* y; CHECK(hipMalloc(&y, 64 * sizeof(Fp28)));
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, d, y);
    std::vector<Fp28> hy(64); CHECK(hipMemcpy(hy.data(), y, 64 * sizeof(Fp28), hipMemcpyDeviceToHost));
    int bad = 0;
    std::vector<uint32_t> R(14, 0); R[12] = 1u << 8;  // 2^392
    for (int t = 0; t < 64; ++t) {
        auto lhs = mod_p(mul(from28(hy[t]), R)), rhs = mod_p(mul(from28(h[2 * t]), from28(h[2 * t + 1])));
        if (lhs != rhs) ++bad;
        for (int k = 0; k < 13; ++k) if (hy[t].v[k] > MASK28) ++bad;
    }
    printf("check: %d of 64 products wrong\n", bad);
    auto time_ms = [&](auto f) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mul28, dim3(blocks), dim3(threads), 0, 0, d, it); });
    printf("Fp product, 14 x 28-bit limbs, no carry word:   %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
    Fp* d32; CHECK(hipMalloc(&d32, n * sizeof(Fp))); CHECK(hipMemset(d32, 1, n * sizeof(Fp)));
    ms = time_ms([&] { hipLaunchKernelGGL(k_mul32, dim3(blocks), dim3(threads), 0, 0, d32, it); });
    printf("Fp fe_mul, 12 x 32-bit limbs (product path):   %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mul28, dim3(1), dim3(64), 0, 0, d, 2000); });
    printf("28-bit form, single-wave latency: %.3f us per product\n", ms * 1e3 / 4000);
    return bad != 0;
}
