// Instruction-rate micro-benchmarks on gfx950 for the integer ops a 384-bit Montgomery product is made of,
// plus end-to-end fe_mul / xyzz_madd throughput of masp_amd/csrc/device.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../masp_amd/csrc/device/curve.hpp"
using namespace masp;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void k_rate(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = seed ^ 0x9e3779b9u, d = a * 7 + 1;
    uint64_t x0 = a, x1 = b, x2 = c, x3 = d;
    double f0 = a, f1 = b, f2 = c, f3 = d;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == 0) {  // v_mad_u64_u32, 4 independent chains (inline asm: as plain C++ the compiler folds the chain away —
                            // the round-2 row of this tool printed 693 560 Gop/s for it)
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %6, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %5, %7, %2\n\tv_mad_u64_u32 %3, vcc, %6, %7, %3\n\t"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            } else if (OP == 1) {  // v_mul_lo_u32
                a = a * b + 1; b = b * c + 1; c = c * d + 1; d = d * a + 1;
            } else if (OP == 2) {  // v_mul_hi_u32
                a = __umulhi(a, b) + 3; b = __umulhi(b, c) + 3; c = __umulhi(c, d) + 3; d = __umulhi(d, a) + 3;
            } else if (OP == 3) {  // v_mad_u32_u24
                a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b;
            } else if (OP == 4) {  // fp64 fma
                f0 = fma(f0, 1.0000001, f1); f1 = fma(f1, 0.9999999, f2); f2 = fma(f2, 1.0000002, f3); f3 = fma(f3, 0.9999998, f0);
            } else if (OP == 5) {  // 32-bit add chain
                a += b; b += c; c += d; d += a;
            } else if (OP == 6) {  // 64-bit add (add_co + addc)
                x0 += x1; x1 += x2; x2 += x3; x3 += x0;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(x0 ^ x1 ^ x2 ^ x3) ^ (uint32_t)(f0 + f1 + f2 + f3);
}

__global__ void k_femul(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    data[t] = fe_add(a, b);
}
// An experiment kept here only: the product with TWO column accumulators (operand products / reduction products), i.e. two
// independent multiply-add chains for a lone wave.  Measured slower than fe_mul on a lone wave (1.56 vs 1.34 us): a lone wave
// already runs fe_mul at its issue time, the extra ~6 instructions per column only cost.
namespace masp {
template <int K, class C>
__device__ __forceinline__ void montl_columns_lo(uint64_t& accA, uint32_t& cA, const uint32_t* a, const uint32_t* b, uint32_t* m) {
    if constexpr (K < C::N) {
        uint64_t accB = 0;
        uint32_t cB = 0;
        macs_vv<0, K + 1, K, C>(accA, cA, a, b);
        macs_vs<0, K, K, C>(accB, cB, m);
        m[K] = ((uint32_t)accA + (uint32_t)accB) * C::INV;
        mac_vs(accB, cB, m[K], C::MOD[0]);
        const uint64_t t = accA + accB;
        const uint32_t c = cA + cB + (t < accA ? 1u : 0u);
        accA = (t >> 32) | ((uint64_t)c << 32);
        cA = 0;
        montl_columns_lo<K + 1, C>(accA, cA, a, b, m);
    }
}
template <int K, class C>
__device__ __forceinline__ void montl_columns_hi(uint64_t& accA, uint32_t& cA, const uint32_t* a, const uint32_t* b, const uint32_t* m, uint32_t* r) {
    if constexpr (K < 2 * C::N - 1) {
        uint64_t accB = 0;
        uint32_t cB = 0;
        macs_vv<K - C::N + 1, C::N, K, C>(accA, cA, a, b);
        macs_vs<K - C::N + 1, C::N, K, C>(accB, cB, m);
        const uint64_t t = accA + accB;
        const uint32_t c = cA + cB + (t < accA ? 1u : 0u);
        r[K - C::N] = (uint32_t)t;
        accA = (t >> 32) | ((uint64_t)c << 32);
        cA = 0;
        montl_columns_hi<K + 1, C>(accA, cA, a, b, m, r);
    }
}
template <class C>
__device__ __forceinline__ Fe<C> fe_mul_lat(const Fe<C>& a, const Fe<C>& b) {
    constexpr int N = C::N;
    uint32_t m[N];
    Fe<C> r;
    uint64_t acc = 0;
    uint32_t c2 = 0;
    montl_columns_lo<0, C>(acc, c2, a.v, b.v, m);
    montl_columns_hi<N, C>(acc, c2, a.v, b.v, m, r.v);
    r.v[N - 1] = (uint32_t)acc;
    fe_reduce_once(r);
    return r;
}
}  // namespace masp
__global__ void k_femul_lat(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul_lat(a, b); b = fe_mul_lat(b, a); }
    data[t] = fe_add(a, b);
}
__global__ void k_femul_call(Fp* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul_nc(a, b); b = fe_mul_nc(b, a); }
    data[t] = fe_add(a, b);
}
__global__ void k_frmul(Fr* data, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = data[t], b = data[t ^ 1];
    for (int i = 0; i < iters; ++i) { a = fe_mul(a, b); b = fe_mul(b, a); }
    data[t] = fe_add(a, b);
}
__global__ void __launch_bounds__(64) k_madd(G1Xyzz* acc, const G1Affine* pts, int npts, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    G1Xyzz a = acc[t];
    for (int i = 0; i < iters; ++i) xyzz_madd(a, pts[(t * 31 + i * 7) % npts], (i & 1) != 0);
    acc[t] = a;
}

template <class F> static float time_ms(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, CUs %d, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int blocks = p.multiProcessorCount * 8, threads = 256, iters = 2000;
    uint32_t* out; CHECK(hipMalloc(&out, blocks * threads * 4));
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_fma_f64", "v_add_u32", "add_u64(2 instr)"};
    auto run = [&](int op) {
        switch (op) {
            case 0: return time_ms([&] { hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            case 1: return time_ms([&] { hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            case 2: return time_ms([&] { hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            case 3: return time_ms([&] { hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            case 4: return time_ms([&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            case 5: return time_ms([&] { hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
            default: return time_ms([&] { hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(threads), 0, 0, out, 5u, iters); });
        }
    };
    for (int op = 0; op < 7; ++op) {
        float ms = run(op);
        double ops = (double)blocks * threads * iters * 16 * 4;
        printf("%-18s %8.3f ms  %8.2f Gop/s  (%.2f lane-ops/clk/CU @%dMHz)\n", names[op], ms, ops / ms / 1e6,
               ops / ms / 1e3 / p.multiProcessorCount / (p.clockRate / 1000.0) / 1e3 * 1e3 / 1e3, p.clockRate / 1000);
    }
    // field multiplication throughput
    {
        int n = blocks * threads;
        std::vector<Fp> h(n);
        for (int i = 0; i < n; ++i) for (int k = 0; k < 12; ++k) h[i].v[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 11 ? 0x0fffffff : 0xffffffff);
        Fp* d; CHECK(hipMalloc(&d, n * sizeof(Fp))); CHECK(hipMemcpy(d, h.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
        int it = 200;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_femul, dim3(blocks), dim3(threads), 0, 0, d, it); });
        printf("Fp  fe_mul          %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_frmul, dim3(blocks), dim3(threads), 0, 0, (Fr*)d, it); });
        printf("Fr  fe_mul          %8.3f ms  %8.2f Gmul/s\n", ms, (double)n * it * 2 / ms / 1e6);
        // single wave latency
        ms = time_ms([&] { hipLaunchKernelGGL(k_femul, dim3(1), dim3(64), 0, 0, d, 2000); });
        printf("Fp  fe_mul single-wave latency: %.3f us per mul\n", ms * 1e3 / 4000);
        ms = time_ms([&] { hipLaunchKernelGGL(k_femul_call, dim3(1), dim3(64), 0, 0, d, 2000); });
        printf("Fp  fe_mul_nc (call) single-wave latency: %.3f us per mul\n", ms * 1e3 / 4000);
        ms = time_ms([&] { hipLaunchKernelGGL(k_femul_lat, dim3(1), dim3(64), 0, 0, d, 2000); });
        printf("Fp  two-accumulator product single-wave latency: %.3f us per mul\n", ms * 1e3 / 4000);
        ms = time_ms([&] { hipLaunchKernelGGL(k_femul_lat, dim3(blocks), dim3(threads), 0, 0, d, it); });
        printf("Fp  two-accumulator product  %8.3f ms  %8.2f Gmul/s (full chip)\n", ms, (double)n * it * 2 / ms / 1e6);
    }
    {
        int waves_per_cu[] = {4, 8, 16};
        for (int w : waves_per_cu) {
            int nthreads = p.multiProcessorCount * w * 64;
            std::vector<G1Xyzz> hz(nthreads);
            memset(hz.data(), 0, hz.size() * sizeof(G1Xyzz));
            G1Xyzz* acc; CHECK(hipMalloc(&acc, nthreads * sizeof(G1Xyzz))); CHECK(hipMemcpy(acc, hz.data(), nthreads * sizeof(G1Xyzz), hipMemcpyHostToDevice));
            // arbitrary (not on-curve) "points": arithmetic cost is identical
            int npts = 1 << 16;
            std::vector<G1Affine> hp(npts);
            for (int i = 0; i < npts; ++i) for (int k = 0; k < 12; ++k) { hp[i].x.v[k] = (i + 1) * 2654435761u + k; hp[i].y.v[k] = (i + 7) * 40503u + k * 977; if (k == 11) { hp[i].x.v[k] &= 0x0fffffff; hp[i].y.v[k] &= 0x0fffffff; } }
            G1Affine* pts; CHECK(hipMalloc(&pts, npts * sizeof(G1Affine))); CHECK(hipMemcpy(pts, hp.data(), npts * sizeof(G1Affine), hipMemcpyHostToDevice));
            int it = 64;
            float ms = time_ms([&] { hipLaunchKernelGGL(k_madd, dim3(nthreads / 64), dim3(64), 0, 0, acc, pts, npts, it); });
            printf("xyzz_madd  %2d waves/CU: %8.3f ms  %8.2f Gmadd/s\n", w, ms, (double)nthreads * it / ms / 1e6);
            hipFree(acc); hipFree(pts);
        }
        G1Xyzz* acc; CHECK(hipMalloc(&acc, 64 * sizeof(G1Xyzz))); CHECK(hipMemset(acc, 0, 64 * sizeof(G1Xyzz)));
        G1Affine* pts; CHECK(hipMalloc(&pts, 1024 * sizeof(G1Affine))); CHECK(hipMemset(pts, 1, 1024 * sizeof(G1Affine)));
        float ms = time_ms([&] { hipLaunchKernelGGL(k_madd, dim3(1), dim3(64), 0, 0, acc, pts, 1024, 500); });
        printf("xyzz_madd single-wave latency: %.2f us per madd\n", ms * 1e3 / 500);
    }
    return 0;
}
