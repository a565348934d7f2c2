// Groth16-specific device kernels around the NTT and MSM engines: witness -> (a, b, c) evaluation
// vectors from the static R1CS, query-scalar gathering by density, and the final proof assembly +
// zcash encoding.  Restates, for the GPU, bellperson's `ProvingAssignment::enforce` evaluation and
// `create_proof` tail (nam-bellperson 0.26.6-nam.1, un-vendored) and
// `Proof::write` (/root/reference/masp_proofs/src/prover.rs:190-193).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "subgroup.hpp"
#include "quad.hpp"
#include "fr_io.hpp"
#include "io.hpp"

namespace masp {


// Fixed-base tables for the per-circuit points that get multiplied by r, s, rs in every proof:
// tab[w*255 + d-1] = d * 2^(8w) * P in XYZZ form (no inversions to build), w < 32.  One lane per window,
// blockIdx.x selects the point.
template <class O>
__global__ void __launch_bounds__(64) k_fixed_table_xyzz(const Affine<O>* __restrict__ pts, Xyzz<O>* __restrict__ tabs) {
    const uint32_t w = threadIdx.x;
    if (w >= 32) return;
    Xyzz<O> base = xyzz_from_affine(pts[blockIdx.x]);
    for (uint32_t k = 0; k < 8 * w; ++k) base = xyzz_dbl(base);
    Xyzz<O>* tab = tabs + (size_t)blockIdx.x * 32 * 255 + w * 255;
    Xyzz<O> cur = base;
    for (uint32_t d = 1; d <= 255; ++d) {
        tab[d - 1] = cur;
        xyzz_add_nc(cur, base);
    }
}
template <class O>
__device__ __forceinline__ Xyzz<O> xyzz_fixed_mul(const Xyzz<O>* __restrict__ tab, const Fr& k) {
    Xyzz<O> acc = xyzz_inf<O>();
    for (int w = 0; w < 32; ++w) {
        uint32_t d = (k.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) xyzz_add_nc(acc, tab[w * 255 + d - 1]);
    }
    return acc;
}

// four lanes per point (xyzz_dbl_coop, xyzz_add_coop): device/quad.hpp

// Proof assembly:
//   g_a = r*delta1 + alpha1 + A
//   g_b = s*delta2 + beta2 + B2
//   g_c = (r s)*delta1 + s*alpha1 + r*beta1 + s*A + r*B1 + H + L
// as SIX small kernels (one workgroup per proof of the batch each) instead of one: every piece is a chain of dependent
// 384-bit products on one to four lanes, and as one kernel (100 000 instructions, 512 VGPRs + 1 166 spilled) the pieces
// could only start when the LAST multi-scalar multiplication of the proof was done.  Apart they start as soon as what they
// read exists — the fixed-base multiplications (which only read r and s) at the very beginning of a lone proof, s*A and
// r*B1 behind their own MSMs on their own streams — so that after the longest MSM chain only one finishing kernel is
// left.  part[6 p ..] = r*delta1, s*alpha1, r*beta1, s*A, r*B1, (r s)*delta1;  part2[p] = s*delta2.
// rs: 8 limbs r | 8 limbs s (canonical).  fb1: tables of delta1, alpha1, beta1 (in that order); fb2: table of delta2.

// value of lane (lane + d) of the wave, limb by limb
template <class O>
__device__ __forceinline__ Xyzz<O> xyzz_lane_down(const Xyzz<O>& p, int d) {
    Xyzz<O> r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Xyzz<O>) / 4; ++i) dst[i] = (uint32_t)__shfl_down((int)src[i], d, 64);
    return r;
}
// r*delta1, s*alpha1, r*beta1, (r s)*delta1 through the fixed-base tables: 128 lanes, 32 per product — lane w of a product
// fetches the table entry of its 8-bit window, a shuffle tree adds the 32 entries (5 dependent additions instead of 32)
__global__ void __launch_bounds__(128) k_groth16_fixed_g1(const G1Xyzz* __restrict__ fb1, const uint32_t* __restrict__ rs, size_t rs_stride,
                                                          G1Xyzz* __restrict__ part) {
    const uint32_t j = threadIdx.x >> 5, w = threadIdx.x & 31;
    rs += (size_t)blockIdx.x * rs_stride;
    part += (size_t)blockIdx.x * 6;
    Fr r, s;
    for (int i = 0; i < 8; ++i) {
        r.v[i] = rs[i];
        s.v[i] = rs[8 + i];
    }
    constexpr size_t TAB = 32 * 255;
    const G1Xyzz* tab = fb1 + (j == 1 ? TAB : j == 2 ? 2 * TAB : 0);
    Fr k = j == 1 ? s : r;
    if (j == 3) k = fe_mul(fe_to_mont(r), s);  // mont(r) * s = r*s mod q, canonical
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = (w >> 2) == (uint32_t)i ? k.v[i] : d;
    d = (d >> (8 * (w & 3))) & 0xffu;
    G1Xyzz acc = d ? tab[w * 255 + d - 1] : xyzz_inf<FpOps>();
    for (int dd = 16; dd >= 1; dd >>= 1) {
        const G1Xyzz other = xyzz_lane_down(acc, dd);
        if ((int)w < dd) xyzz_add_nc(acc, other);
    }
    if (w == 0) part[j == 3 ? 5 : j] = acc;
}
// s*delta2 on G2 the same way over 32 lane PAIRS (Fp2PairOps: half a coordinate per lane)
__global__ void __launch_bounds__(64) k_groth16_fixed_g2(const G2Xyzz* __restrict__ fb2, const uint32_t* __restrict__ rs, size_t rs_stride,
                                                         G2Xyzz* __restrict__ part2) {
    typedef Fp2PairOps O;
    const uint32_t w = threadIdx.x >> 1, h = threadIdx.x & 1u;
    rs += (size_t)blockIdx.x * rs_stride;
    const uint32_t d = (rs[8 + (w >> 2)] >> (8 * (w & 3))) & 0xffu;
    Xyzz<O> acc = xyzz_inf<O>();
    if (d) {
        const Fp* q = reinterpret_cast<const Fp*>(fb2 + w * 255 + d - 1);
        acc.X = q[h];
        acc.Y = q[2 + h];
        acc.ZZ = q[4 + h];
        acc.ZZZ = q[6 + h];
    }
    for (int dd = 16; dd >= 1; dd >>= 1) {
        const Xyzz<O> other = xyzz_lane_down(acc, 2 * dd);
        if ((int)w < dd) xyzz_add_nc(acc, other);
    }
    if (w == 0) {
        Fp* q = reinterpret_cast<Fp*>(part2 + blockIdx.x);
        q[h] = acc.X;
        q[2 + h] = acc.Y;
        q[4 + h] = acc.ZZ;
        q[6 + h] = acc.ZZZ;
    }
}
// k = q u^2 + rem, rem < u^2 < 2^128 (u^2 = FpCfg::U_SQR), by binary long division; q < 2^128 for every k < 2^255 (u^2 > 2^127)
__device__ __forceinline__ void endo_split(const uint32_t* k, uint32_t* q, uint32_t* rem) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) q[i] = 0;
    for (int i = 255; i >= 0; --i) {
        for (int j = 4; j > 0; --j) r[j] = (r[j] << 1) | (r[j - 1] >> 31);
        r[0] = (r[0] << 1) | ((k[i >> 5] >> (i & 31)) & 1u);
        // r >= u^2 ?  (r < 2 u^2 < 2^129: five words)
        bool ge = r[4] != 0;
        if (!ge) {
            ge = true;
            for (int j = 3; j >= 0; --j)
                if (r[j] != FpCfg::U_SQR[j]) {
                    ge = r[j] > FpCfg::U_SQR[j];
                    break;
                }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int j = 0; j < 4; ++j) {
                const uint64_t d = (uint64_t)r[j] - FpCfg::U_SQR[j] - borrow;
                r[j] = (uint32_t)d;
                borrow = (d >> 32) & 1u;
            }
            r[4] -= (uint32_t)borrow;
            if (i < 128) q[i >> 5] |= 1u << (i & 31);
        }
    }
    for (int i = 0; i < 4; ++i) rem[i] = r[i];
}
// WHICH = 0: s*A -> part[3];  1: r*B1 -> part[4].  Lanes 0..15 build the table d*P (d < 16) in LDS, then lanes 0..3 run
// 4-bit fixed windows, four lanes per point.
// endo != 0 (every CRS point that A and B1 are sums of lies in the prime-order subgroup: checked once, when the circuit is
// loaded — k_g1_subgroup_flag): k = q u^2 + rem and [u^2] P = -phi(P) = (beta x, -y) on the subgroup (subgroup.hpp), so
// [k] P = [rem] P + [q] (beta x, -y): two 128-bit scalars, 124 doublings and <= 32 additions each, on two quads of the same
// wave — the table of (beta x, -y) is the table of P with X scaled and Y negated.  This multiplication is the tail of the
// two longest G1 chains of a lone proof (2.1 ms of 252 dependent doublings before).
// endo == 0: 252 doublings + <= 64 additions, exact for ANY curve point — the reference reads the CRS unchecked
// (Parameters::read(_, false), /root/reference/masp_proofs/src/lib.rs:343-347) and a point of the curve outside the
// subgroup must give the bytes the reference's plain double-and-add gives.
// (gridDim.y = 2 runs both: WHICH = which0 + blockIdx.y)
__global__ void __launch_bounds__(64) k_groth16_var_mul(uint32_t which0, const G1Xyzz* __restrict__ msm_g1 /* H, L, A, B1 */,
                                                        const uint32_t* __restrict__ rs, size_t rs_stride, G1Xyzz* __restrict__ part, int endo) {
    __shared__ G1Xyzz wtab[32];
    const uint32_t tid = threadIdx.x, WHICH = which0 + blockIdx.y;
    msm_g1 += (size_t)blockIdx.x * 4;
    rs += (size_t)blockIdx.x * rs_stride + (WHICH == 0 ? 8 : 0);
    part += (size_t)blockIdx.x * 6;
    if (tid < 16) {  // wtab[d] = d * P by double-and-add over the 4 bits of d
        const G1Xyzz P = msm_g1[2 + WHICH];
        G1Xyzz t = xyzz_inf<FpOps>();
        for (int b = 3; b >= 0; --b) {
            t = xyzz_dbl(t);
            if ((tid >> b) & 1) xyzz_add_nc(t, P);
        }
        wtab[tid] = t;
        if (endo) {  // wtab[16 + d] = d * [u^2] P
            Fp beta;
            for (int i = 0; i < 12; ++i) beta.v[i] = FpCfg::ENDO_BETA[i];
            t.X = fe_mul(t.X, beta);
            t.Y = fe_neg(t.Y);
            wtab[16 + tid] = t;
        }
    }
    // same wave writes and reads the table: LDS is in order per wave, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // four lanes per point (quad.hpp).  endo: quad 0 runs [rem] P, quad 1 runs [q] [u^2] P — two chains of 124 doublings and
    // <= 32 additions side by side in the same wave — and quad 0 adds the two
    __shared__ G1Xyzz other;
    G1Xyzz acc = xyzz_inf<FpOps>();
    const uint32_t lig = tid & 3u, half = tid >> 2;
    if (tid < (endo ? 8u : 4u)) {
        uint32_t k[8];
        for (int i = 0; i < 8; ++i) k[i] = rs[i];
        if (endo) {
            uint32_t q[4], rem[4], mine[4];
            endo_split(k, q, rem);
            for (int i = 0; i < 4; ++i) mine[i] = half ? q[i] : rem[i];
            const G1Xyzz* tab = wtab + 16 * half;
            for (int w = 31; w >= 0; --w) {
                if (w != 31)
                    for (int j = 0; j < 4; ++j) acc = xyzz_dbl_coop(acc, lig);
                const uint32_t d = (mine[w >> 3] >> (4 * (w & 7))) & 15u;
                if (d) xyzz_add_coop(acc, tab[d], lig);
            }
            if (tid == 4) other = acc;
        } else {
            for (int w = 63; w >= 0; --w) {
                if (w != 63)
                    for (int j = 0; j < 4; ++j) acc = xyzz_dbl_coop(acc, lig);
                const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 15u;
                if (d) xyzz_add_coop(acc, wtab[d], lig);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid < 4) {
        if (endo) xyzz_add_coop(acc, other, lig);
        if (lig == 0) part[3 + WHICH] = acc;
    }
}
// *flag |= 1 if any of the n points (affine, `stride` bytes apart, infinity skipped) lies outside the prime-order subgroup
__global__ void __launch_bounds__(64) k_g1_subgroup_flag(const uint8_t* __restrict__ pts, size_t stride, uint32_t n, int* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = *reinterpret_cast<const G1Affine*>(pts + (size_t)i * stride);
    if (fe_is_zero(p.x) && fe_is_zero(p.y)) return;
    if (!g1_in_subgroup(p)) atomicOr(flag, 1);
}
// one lane: g_b = s*delta2 + beta2 + B2, normalised and encoded
__global__ void __launch_bounds__(64) k_groth16_finish_b(const VkDevice* __restrict__ vk, const G2Xyzz* __restrict__ part2,
                                                         const G2Xyzz* __restrict__ msm_g2, uint8_t* __restrict__ proof) {
    if (threadIdx.x != 0) return;
    G2Xyzz gb = part2[blockIdx.x];
    xyzz_madd_nc(gb, vk->beta_g2, false);
    xyzz_add_nc(gb, msm_g2[blockIdx.x]);
    g2_write_compressed(xyzz_to_affine<Fp2Ops, true>(gb), proof + (size_t)blockIdx.x * 192 + 48);
}
// two waves, one lane each: g_a and g_c, normalised and encoded
__global__ void __launch_bounds__(128) k_groth16_finish_ac(const VkDevice* __restrict__ vk, const G1Xyzz* __restrict__ part,
                                                           const G1Xyzz* __restrict__ msm_g1 /* H, L, A, B1 */, uint8_t* __restrict__ proof) {
    const uint32_t tid = threadIdx.x;
    part += (size_t)blockIdx.x * 6;
    msm_g1 += (size_t)blockIdx.x * 4;
    proof += (size_t)blockIdx.x * 192;
    if (tid == 0) {
        G1Xyzz ga = part[0];
        xyzz_madd_nc(ga, vk->alpha_g1, false);
        xyzz_add_nc(ga, msm_g1[2]);
        g1_write_compressed(xyzz_to_affine<FpOps, true>(ga), proof);
    } else if (tid == 64) {
        G1Xyzz gc = part[5];
        xyzz_add_nc(gc, part[1]);
        xyzz_add_nc(gc, part[2]);
        xyzz_add_nc(gc, part[3]);
        xyzz_add_nc(gc, part[4]);
        xyzz_add_nc(gc, msm_g1[0]);
        xyzz_add_nc(gc, msm_g1[1]);
        g1_write_compressed(xyzz_to_affine<FpOps, true>(gc), proof + 144);
    }
}

// The same in two steps for a lone proof, whose last chain to finish is quotient -> MSM h: everything that does not need H — g_a whole,
// and g_c's five assembly pieces + L summed into part[5] — runs behind the side chains while H is still being computed; what is left
// behind H is ONE addition, the normalisation and the encoding (0.33 -> 0.15 ms at the end of a 3.7 ms proof).
__global__ void __launch_bounds__(128) k_groth16_finish_ac_early(const VkDevice* __restrict__ vk, G1Xyzz* __restrict__ part,
                                                                 const G1Xyzz* __restrict__ msm_g1 /* H, L, A, B1 */, uint8_t* __restrict__ proof) {
    const uint32_t tid = threadIdx.x;
    part += (size_t)blockIdx.x * 6;
    msm_g1 += (size_t)blockIdx.x * 4;
    proof += (size_t)blockIdx.x * 192;
    if (tid == 0) {
        G1Xyzz ga = part[0];
        xyzz_madd_nc(ga, vk->alpha_g1, false);
        xyzz_add_nc(ga, msm_g1[2]);
        g1_write_compressed(xyzz_to_affine<FpOps, true>(ga), proof);
    } else if (tid == 64) {
        G1Xyzz gc = part[5];
        xyzz_add_nc(gc, part[1]);
        xyzz_add_nc(gc, part[2]);
        xyzz_add_nc(gc, part[3]);
        xyzz_add_nc(gc, part[4]);
        xyzz_add_nc(gc, msm_g1[1]);
        part[5] = gc;
    }
}
__global__ void __launch_bounds__(64) k_groth16_finish_c_late(const G1Xyzz* __restrict__ part, const G1Xyzz* __restrict__ msm_g1, uint8_t* __restrict__ proof) {
    if (threadIdx.x != 0) return;
    G1Xyzz gc = part[(size_t)blockIdx.x * 6 + 5];
    xyzz_add_nc(gc, msm_g1[(size_t)blockIdx.x * 4]);
    g1_write_compressed(xyzz_to_affine<FpOps, true>(gc), proof + (size_t)blockIdx.x * 192 + 144);
}

// single point XYZZ -> uncompressed bytes (building-block entry points)
__global__ void k_g1_export(const G1Xyzz* __restrict__ p, uint8_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g1_write_uncompressed(xyzz_to_affine(*p), out);
}
__global__ void k_g2_export(const G2Xyzz* __restrict__ p, uint8_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g2_write_uncompressed(xyzz_to_affine(*p), out);
}
__global__ void k_g1_import_one(const uint8_t* __restrict__ raw, G1Affine* __restrict__ out, int* __restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int st = g1_read_uncompressed(raw, *out);
        if (st) atomicOr(status, st);
    }
}
__global__ void k_g2_import_one(const uint8_t* __restrict__ raw, G2Affine* __restrict__ out, int* __restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int st = g2_read_uncompressed(raw, *out);
        if (st) atomicOr(status, st);
    }
}

}  // namespace masp
