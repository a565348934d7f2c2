// Bounded experiment (round-1 review, item 10): can the int8 MFMA pipe take the constant half of a Montgomery product?
// Half of the 288 multiply-adds of a 384-bit Montgomery product multiply by a CONSTANT (the modulus): for a wave of 64
// independent elements, m x N is a (64 x 48 B) . (48 x 96 B) int8 GEMM — 24 v_mfma_i32_16x16x64_i8 — which issues beside the
// VALU.  What it costs around the MFMAs is what decides: the operands live one element per lane, the MFMA wants 16-row
// tiles (a transpose through LDS each way), and its output is 96 byte-position column sums per element that have to be
// swept back into 32-bit limbs with carries.  This file times, per wave of 64 elements,
//   A  the VALU path the product uses now: 144 v_mad_u64_u32 + v_addc (modulus limbs in SGPRs) + 12 v_mul_lo_u32
//   B  LDS transpose in -> 24 MFMA -> LDS transpose out -> carry sweep of 96 columns into 24 limbs
// B leaves out what the real thing would need on top (the quotient digits m must all exist BEFORE the GEMM, i.e. a second
// constant product T_lo x N' mod R instead of the 12 interleaved v_mul_lo; unsigned bytes through a signed-int8 MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_redc_ubench.hip -o tools/_build/mfma_redc_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../masp_amd/csrc/device/field.hpp"
using namespace masp;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: the reduction half of fe_mul on its own (columns of m x MOD), chained so that nothing is hoisted
__global__ void __launch_bounds__(64) k_valu(uint32_t* data, int iters) {
    uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t m[12], r[12];
    for (int i = 0; i < 12; ++i) m[i] = data[t * 12 + i];
    for (int it = 0; it < iters; ++it) {
        uint64_t acc = m[0];
        uint32_t c2 = 0;
        // low columns: digits and their products (the a x b half is absent: acc only carries the reduction terms)
        [&]<int... K>(std::integer_sequence<int, K...>) {
            (([&] {
                 macs_vs<0, K, K, FpCfg>(acc, c2, m);
                 uint32_t q = (uint32_t)acc * FpCfg::INV;
                 mac_vs(acc, c2, q, FpCfg::MOD[0]);
                 r[K] = q;
                 acc = (acc >> 32) | ((uint64_t)c2 << 32);
                 c2 = 0;
             }()),
             ...);
        }(std::make_integer_sequence<int, 12>{});
        [&]<int... K>(std::integer_sequence<int, K...>) {
            (([&] {
                 macs_vs<K + 1, 12, K + 12, FpCfg>(acc, c2, m);
                 m[K] = (uint32_t)acc ^ r[K];
                 acc = (acc >> 32) | ((uint64_t)c2 << 32);
                 c2 = 0;
             }()),
             ...);
        }(std::make_integer_sequence<int, 11>{});
        m[11] = (uint32_t)acc ^ r[11];
    }
    for (int i = 0; i < 12; ++i) data[t * 12 + i] = m[i];
}

// B: transpose in, 24 MFMAs, transpose out, carry sweep
__global__ void __launch_bounds__(64) k_mfma(uint32_t* data, int iters) {
    __shared__ uint32_t lds_in[64 * 16];      // 64 rows x 64 bytes (48 used)
    __shared__ uint32_t lds_out[96 * 64];     // [column][element]
    const uint32_t lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    uint32_t m[12];
    for (int i = 0; i < 12; ++i) m[i] = data[t * 12 + i];
    // B tiles: Toeplitz matrix of the modulus bytes, column tile j, this lane's 16 k-bytes of column (lane % 16)
    v4i Bt[6];
    {
        const uint8_t* nb = reinterpret_cast<const uint8_t*>(FpCfg::MOD);
        for (int j = 0; j < 6; ++j) {
            uint32_t w[4] = {0, 0, 0, 0};
            const int col = 16 * j + (lane & 15);
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * (lane >> 4) + kk, d = col - k;
                const uint32_t byte = (k < 48 && d >= 0 && d < 48) ? nb[d] : 0;
                w[kk >> 2] |= byte << (8 * (kk & 3));
            }
            Bt[j] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        }
    }
    for (int i = 0; i < 4; ++i) lds_in[lane * 16 + 12 + i] = 0;  // zero padding of K = 48..63
    for (int it = 0; it < iters; ++it) {
        // transpose in: element per lane -> 16-row tiles
        for (int i = 0; i < 12; ++i) lds_in[lane * 16 + i] = m[i];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        v4i C[4][6];
        for (int rt = 0; rt < 4; ++rt) {
            const uint32_t* src = &lds_in[(16 * rt + (lane & 15)) * 16 + 4 * (lane >> 4)];
            v4i A = v4i{(int)src[0], (int)src[1], (int)src[2], (int)src[3]};
            for (int j = 0; j < 6; ++j) C[rt][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bt[j], v4i{0, 0, 0, 0}, 0, 0, 0);
        }
        // transpose out: lane holds rows 4 (lane / 16) .. + 3 of column (lane % 16) of every tile
        for (int rt = 0; rt < 4; ++rt)
            for (int j = 0; j < 6; ++j) {
                const int col = 16 * j + (lane & 15), row = 16 * rt + 4 * (lane >> 4);
                uint32_t* dst = &lds_out[col * 64 + row];
                dst[0] = C[rt][j].x; dst[1] = C[rt][j].y; dst[2] = C[rt][j].z; dst[3] = C[rt][j].w;
            }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // carry sweep: 96 byte-position sums -> 24 limbs (the upper 12 are the reduced value's contribution)
        uint64_t acc = 0;
        uint32_t limb[24];
        for (int q = 0; q < 24; ++q) {
            for (int i = 0; i < 4; ++i) acc += (uint64_t)lds_out[(4 * q + i) * 64 + lane] << (8 * i);
            limb[q] = (uint32_t)acc;
            acc >>= 32;
        }
        for (int i = 0; i < 12; ++i) m[i] = limb[12 + i] ^ limb[i];
    }
    for (int i = 0; i < 12; ++i) data[t * 12 + i] = m[i];
}

template <class F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int waves = p.multiProcessorCount * 4 * 2, iters = 2000;      // two waves per SIMD, like the accumulation kernel
    uint32_t* d; CHECK(hipMalloc(&d, (size_t)waves * 64 * 12 * 4)); CHECK(hipMemset(d, 0x5a, (size_t)waves * 64 * 12 * 4));
    float a = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(waves), dim3(64), 0, 0, d, iters); });
    float b = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(waves), dim3(64), 0, 0, d, iters); });
    CHECK(hipDeviceSynchronize());
    const double elems = (double)waves * 64 * iters;
    printf("device %s, %d CUs; %d waves x %d iterations\n", p.name, p.multiProcessorCount, waves, iters);
    printf("A  VALU   (144 mad_u64_u32 + addc, 12 mul_lo)                    : %8.3f ms  %7.1f G reductions/s\n", a, elems / a / 1e6);
    printf("B  MFMA   (LDS in, 24 mfma_i32_16x16x64_i8, LDS out, carry sweep): %8.3f ms  %7.1f G reductions/s   B/A time = %.2f\n", b, elems / b / 1e6, b / a);
    printf("verdict: %s\n", b < 0.8 * a ? "GO (more than 20 %% faster)" : "NO-GO (not 20 %% faster than the VALU path, before the parts B leaves out)");
    return 0;
}
