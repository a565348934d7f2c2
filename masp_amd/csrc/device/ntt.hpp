// Radix-2 NTT over the BLS12-381 scalar field and the Groth16 quotient  h = (A*B - C)/Z  for gfx950.
//
// Six transforms, not bellperson's seven: with Z = x^m - 1 and C = A B mod Z (c = a o b on the domain), the product reduced
// on the coset g H is  A B mod (x^m - g^m) = (g^m - 1) h + C,  so  h = (icoset_fft(A B on the coset) - C) / (g^m - 1)  and C
// never has to be carried onto the coset and back (its coefficients come out of the first inverse transform anyway).
// Same field elements as the reference sequence: h is unique.
//
// Replaces bellperson's `EvaluationDomain::{ifft, coset_fft, mul_assign, sub_assign,
// divide_by_z_on_coset, icoset_fft}` sequence (nam-bellperson 0.26.6-nam.1, un-vendored; SURVEY.md A.3
// step 3, reached from /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252) and the
// `<Fr>_radix_fft` kernel of nam-ec-gpu-gen.
//
// A transform of 2^logm points is: one bit-reversal pass (fused with whatever pointwise work
// precedes it) and then ceil(logm / 10) LDS passes, each doing up to 10 butterfly stages on a
// 1024-element (32 KiB) tile held in LDS.  A lone transform (<= 4 MiB) stays in L2 / Infinity Cache
// between passes; in batch mode every pass runs over the whole batch (np x 4 MiB per buffer), which streams
// through HBM — see enqueue_quotient for how the batch is cut so that a sub-batch's working set stays in MALL.
#pragma once
#include <hip/hip_runtime.h>

#include "field.hpp"
#include "fr_io.hpp"
#include "ntt_geom.h"

namespace masp {


__device__ __forceinline__ uint32_t bitrev(uint32_t k, uint32_t logm) { return __brev(k) >> (32 - logm); }

// table[k] = scale * base^k  (Montgomery in, Montgomery out; `plain` strips the Montgomery factor)
__global__ void k_fr_powers(Fr* __restrict__ table, uint32_t n, Fr base, Fr scale, int plain) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t e[1] = {k};
    Fr r = fe_mul(fe_pow(base, e, 1), scale);
    if (plain) r = fe_from_mont(r);
    fr_store(table + k, r);
}

// Prove-time kernels take gridDim.y = proofs in the batch; proof p uses `ptr + p * stride` (strides in elements).
#define NTT_P (blockIdx.y)

// ---- bit-reversal passes with fused pointwise work ------------------------------------------------
// y[rev(k)] = to_mont(x[k]) for k < nrows, 0 above   (x canonical little-endian limbs)
__global__ void k_ntt_load_bitrev(const Fr* __restrict__ x, size_t x_stride, uint32_t nrows, Fr* __restrict__ y, uint32_t logm) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << logm)) return;
    x += NTT_P * x_stride;
    y += (size_t)NTT_P << logm;
    Fr v = fe_zero<FrCfg>();
    if (k < nrows) v = fe_to_mont(fr_load(x + k));
    fr_store(y + bitrev(k, logm), v);
}
// same but the input is already in Montgomery form
__global__ void k_ntt_copy_bitrev(const Fr* __restrict__ x, size_t x_stride, uint32_t nrows, Fr* __restrict__ y, uint32_t logm) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << logm)) return;
    x += NTT_P * x_stride;
    y += (size_t)NTT_P << logm;
    Fr v = fe_zero<FrCfg>();
    if (k < nrows) v = fr_load(x + k);
    fr_store(y + bitrev(k, logm), v);
}
// y[rev(k)] = x[k] * scale[k]
__global__ void k_ntt_scale_bitrev(const Fr* __restrict__ x, const Fr* __restrict__ scale, Fr* __restrict__ y, uint32_t logm) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << logm)) return;
    x += (size_t)NTT_P << logm;
    y += (size_t)NTT_P << logm;
    fr_store(y + bitrev(k, logm), fe_mul(fr_load(x + k), fr_load(scale + k)));
}
// y[rev(k)] = a[k] * b[k]
__global__ void k_ntt_ab_bitrev(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ y, uint32_t logm) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << logm)) return;
    a += (size_t)NTT_P << logm;
    b += (size_t)NTT_P << logm;
    y += (size_t)NTT_P << logm;
    fr_store(y + bitrev(k, logm), fe_mul(fr_load(a + k), fr_load(b + k)));
}
// y[k] = x[k] * scale[k] - c[k] * cscale   (plain-form scale factors: the result leaves Montgomery form)
__global__ void k_fr_scale_sub(const Fr* __restrict__ x, const Fr* __restrict__ scale, const Fr* __restrict__ c, Fr cscale, Fr* __restrict__ y,
                               uint32_t n, size_t y_stride) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    x += (size_t)NTT_P * n;
    c += (size_t)NTT_P * n;
    y += (size_t)NTT_P * y_stride;
    fr_store(y + k, fe_sub(fe_mul(fr_load(x + k), fr_load(scale + k)), fe_mul(fr_load(c + k), cscale)));
}
// y[k] = x[k] * scale[k]  (no permutation; with a plain-form scale table this also leaves Montgomery form)
__global__ void k_fr_scale(const Fr* __restrict__ x, const Fr* __restrict__ scale, Fr* __restrict__ y, uint32_t n, size_t y_stride) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    x += (size_t)NTT_P * n;
    y += (size_t)NTT_P * y_stride;
    fr_store(y + k, fe_mul(fr_load(x + k), fr_load(scale + k)));
}
__global__ void k_fr_from_mont(const Fr* __restrict__ x, Fr* __restrict__ y, uint32_t n) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    fr_store(y + k, fe_from_mont(fr_load(x + k)));
}

// ---- LDS pass: stages [s0, s0 + nst) of a decimation-in-time transform on bit-reversed input -------
// Index bits of an element: [0, s0) "lo" | [s0, s0+nst) "h" (the butterfly bits of this pass) | rest "top".
// A tile holds every h for `cols` = 2^(LT - nst) consecutive columns, column = top * 2^s0 + lo.
// tw[k] = w^k for k < m/2 (w = omega or omega^-1).
__global__ void __launch_bounds__(256) k_ntt_pass(Fr* __restrict__ data, const Fr* __restrict__ tw, uint32_t logm, uint32_t s0,
                                                  uint32_t nst) {
    __shared__ uint4 tile[2 << NTT_LT];  // 2 x uint4 per element, split planes to keep ds_read_b128 conflict-light
    const uint32_t lt = nst + (NTT_LT - nst < logm - nst ? NTT_LT - nst : logm - nst);  // log2 of this tile's size
    const uint32_t cols_log = lt - nst;
    const uint32_t cols = 1u << cols_log;
    const uint32_t tsize = 1u << lt;
    const uint32_t col0 = blockIdx.x << cols_log;
    const uint32_t lomask = (1u << s0) - 1u;
    data += (size_t)NTT_P << logm;
    // load
    for (uint32_t L = threadIdx.x; L < tsize; L += blockDim.x) {
        uint32_t h = L >> cols_log, col = col0 + (L & (cols - 1));
        uint32_t idx = ((col >> s0) << (s0 + nst)) | (h << s0) | (col & lomask);
        const uint4* q = reinterpret_cast<const uint4*>(data + idx);
        tile[L] = q[0];
        tile[tsize + L] = q[1];
    }
    __syncthreads();
    auto ld = [&](uint32_t L) {
        Fr r;
        uint4 a0 = tile[L], a1 = tile[tsize + L];
        r.v[0] = a0.x; r.v[1] = a0.y; r.v[2] = a0.z; r.v[3] = a0.w; r.v[4] = a1.x; r.v[5] = a1.y; r.v[6] = a1.z; r.v[7] = a1.w;
        return r;
    };
    auto st = [&](uint32_t L, const Fr& x) {
        tile[L] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        tile[tsize + L] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    };
    uint32_t q = 0;
    // two stages at a time: a lane holds the four elements that differ in bits q, q + 1 of the butterfly index and does both
    // stages in registers (same four products as two radix-2 stages, half the LDS round trips and barriers)
    for (; q + 1 < nst; q += 2) {
        const uint32_t s = s0 + q;
        for (uint32_t gidx = threadIdx.x; gidx < (tsize >> 2); gidx += blockDim.x) {
            const uint32_t cl = gidx & (cols - 1), r = gidx >> cols_log;
            const uint32_t hj = r & ((1u << q) - 1u), hg = r >> q;
            const uint32_t h00 = (hg << (q + 2)) | hj;
            const uint32_t L0 = (h00 << cols_log) | cl, d = cols << q;  // elements at L0, L0 + d, L0 + 2d, L0 + 3d
            const uint32_t lo = (col0 + cl) & lomask;
            const uint32_t j1 = (hj << s0) | lo;                          // stage s: both pairs
            const uint32_t j2a = j1, j2b = ((hj | (1u << q)) << s0) | lo;  // stage s + 1: pairs with bit q clear / set
            Fr x0 = ld(L0), x1 = ld(L0 + d), x2 = ld(L0 + 2 * d), x3 = ld(L0 + 3 * d);
            if (s == 0) {  // the first two stages of a transform: twiddles 1, 1 and (1, w^(m/4)) — one product instead of four
                const Fr y0 = fe_add(x0, x1), y1 = fe_sub(x0, x1), y2 = fe_add(x2, x3), y3 = fe_sub(x2, x3);
                st(L0, fe_add(y0, y2));
                st(L0 + 2 * d, fe_sub(y0, y2));
                const Fr t = fe_mul(y3, fr_load(tw + ((size_t)1 << (logm - 2))));
                st(L0 + d, fe_add(y1, t));
                st(L0 + 3 * d, fe_sub(y1, t));
                continue;
            }
            const Fr w1 = fr_load(tw + ((size_t)j1 << (logm - s - 1)));
            const Fr w2a = fr_load(tw + ((size_t)j2a << (logm - s - 2)));
            const Fr w2b = fr_load(tw + ((size_t)j2b << (logm - s - 2)));
            Fr t = fe_mul(x1, w1);
            Fr y0 = fe_add(x0, t), y1 = fe_sub(x0, t);
            t = fe_mul(x3, w1);
            Fr y2 = fe_add(x2, t), y3 = fe_sub(x2, t);
            t = fe_mul(y2, w2a);
            st(L0, fe_add(y0, t));
            st(L0 + 2 * d, fe_sub(y0, t));
            t = fe_mul(y3, w2b);
            st(L0 + d, fe_add(y1, t));
            st(L0 + 3 * d, fe_sub(y1, t));
        }
        __syncthreads();
    }
    for (; q < nst; ++q) {
        const uint32_t s = s0 + q;
        for (uint32_t b = threadIdx.x; b < (tsize >> 1); b += blockDim.x) {
            uint32_t cl = b & (cols - 1), hb = b >> cols_log;
            uint32_t hj = hb & ((1u << q) - 1u), hg = hb >> q;
            uint32_t L0 = (((hg << (q + 1)) | hj) << cols_log) | cl;
            uint32_t L1 = L0 + (cols << q);
            uint32_t col = col0 + cl;
            uint32_t jglob = (hj << s0) | (col & lomask);
            Fr w = fr_load(tw + ((size_t)jglob << (logm - s - 1)));
            Fr u = ld(L0), v = fe_mul(ld(L1), w);
            st(L0, fe_add(u, v));
            st(L1, fe_sub(u, v));
        }
        __syncthreads();
    }
    for (uint32_t L = threadIdx.x; L < tsize; L += blockDim.x) {
        uint32_t h = L >> cols_log, col = col0 + (L & (cols - 1));
        uint32_t idx = ((col >> s0) << (s0 + nst)) | (h << s0) | (col & lomask);
        uint4* q = reinterpret_cast<uint4*>(data + idx);
        q[0] = tile[L];
        q[1] = tile[tsize + L];
    }
}

}  // namespace masp
