// Groth16-specific device kernels around the NTT and MSM engines: witness -> (a, b, c) evaluation
// vectors from the static R1CS, query-scalar gathering by density, and the final proof assembly +
// zcash encoding.  Restates, for the GPU, bellperson's `ProvingAssignment::enforce` evaluation and
// `create_proof` tail (nam-bellperson 0.26.6-nam.1, un-vendored) and
// `Proof::write` (/root/reference/masp_proofs/src/prover.rs:190-193).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.cuh"
#include "fr_io.cuh"
#include "io.cuh"

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

// ---- four lanes, one point ------------------------------------------------------------------------------
// The only long serial chain of a proof is the pair of variable-base multiplications of the assembly (252 doublings
// each).  A lone wave spends 1.3 us per 384-bit product (its issue time: tools/ubench.hip), so the chain is cut by giving every
// point operation to FOUR adjacent lanes: all four hold the same point, each computes a different product of the same dependency
// level (one product site, different operands per lane), and the results are exchanged by DPP quad broadcasts.  A doubling is 3
// levels instead of 9 products, an addition 4 instead of 14.  `lig` = lane in group (0..3).
// lane `lig` of the group takes a_lig.  Written with lane masks (0 / ~0), not selects: the compiler turns a chain of selects
// over twelve limbs into divergent branches of moves (550 v_mov and 50 branches per product level, more than the product)
__device__ __forceinline__ Fp coop_pick(uint32_t lig, const Fp& a0, const Fp& a1, const Fp& a2, const Fp& a3) {
    const uint32_t m0 = 0u - (uint32_t)(lig == 0), m1 = 0u - (uint32_t)(lig == 1), m2 = 0u - (uint32_t)(lig == 2), m3 = 0u - (uint32_t)(lig == 3);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; ++i) r.v[i] = (a0.v[i] & m0) | (a1.v[i] & m1) | (a2.v[i] & m2) | (a3.v[i] & m3);
    return r;
}
// the value lane SRC of this 4-lane group holds: a DPP quad broadcast (a register move per limb; the LDS shuffle this replaced
// cost more than the product it fed)
template <int SRC>
__device__ __forceinline__ Fp coop_from(const Fp& v) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], SRC * 0x55 /* quad_perm [SRC, SRC, SRC, SRC] */, 0xf, 0xf, true);
    return r;
}
// dbl-2008-s-1, same case analysis as xyzz_dbl
__device__ __forceinline__ G1Xyzz xyzz_dbl_coop(const G1Xyzz& p, uint32_t lig) {
    if (xyzz_is_inf(p)) return p;
    Fp U = fe_dbl(p.Y);
    if (fe_is_zero(U)) return xyzz_inf<FpOps>();
    Fp t = fe_mul(coop_pick(lig, U, p.X, U, U), coop_pick(lig, U, p.X, U, U));  // V = U^2 | X^2
    const Fp V = coop_from<0>(t), X2 = coop_from<1>(t);
    const Fp M = fe_add(fe_dbl(X2), X2);
    t = fe_mul(coop_pick(lig, U, p.X, M, V), coop_pick(lig, V, V, M, p.ZZ));      // W = U V | S = X V | M^2 | ZZ' = V ZZ
    const Fp W = coop_from<0>(t), S = coop_from<1>(t), MM = coop_from<2>(t);
    G1Xyzz r;
    r.ZZ = coop_from<3>(t);
    r.X = fe_sub(MM, fe_dbl(S));
    t = fe_mul(coop_pick(lig, W, M, W, W), coop_pick(lig, p.Y, fe_sub(S, r.X), p.ZZZ, W));  // W Y | M (S - X') | ZZZ' = W ZZZ
    r.Y = fe_sub(coop_from<1>(t), coop_from<0>(t));
    r.ZZZ = coop_from<2>(t);
    return r;
}
// add-2008-s, same case analysis as xyzz_add (the rare P == +-Q cases are computed redundantly by the four lanes)
__device__ __forceinline__ void xyzz_add_coop(G1Xyzz& acc, const G1Xyzz& b, uint32_t lig) {
    if (xyzz_is_inf(b)) return;
    if (xyzz_is_inf(acc)) {
        acc = b;
        return;
    }
    Fp t = fe_mul(coop_pick(lig, acc.X, b.X, acc.Y, b.Y), coop_pick(lig, b.ZZ, acc.ZZ, b.ZZZ, acc.ZZZ));
    const Fp U1 = coop_from<0>(t), U2 = coop_from<1>(t), S1 = coop_from<2>(t), S2 = coop_from<3>(t);
    const Fp P = fe_sub(U2, U1), R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R))
            acc = xyzz_dbl(acc);
        else
            acc = xyzz_inf<FpOps>();
        return;
    }
    t = fe_mul(coop_pick(lig, P, R, acc.ZZ, acc.ZZZ), coop_pick(lig, P, R, b.ZZ, b.ZZZ));  // PP | R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const Fp PP = coop_from<0>(t), RR = coop_from<1>(t), Z12 = coop_from<2>(t), Z123 = coop_from<3>(t);
    t = fe_mul(coop_pick(lig, P, U1, Z12, P), PP);                                       // PPP | Q | ZZ3
    const Fp PPP = coop_from<0>(t), Q = coop_from<1>(t);
    acc.ZZ = coop_from<2>(t);
    acc.X = fe_sub(fe_sub(RR, PPP), fe_dbl(Q));
    t = fe_mul(coop_pick(lig, S1, R, Z123, S1), coop_pick(lig, PPP, fe_sub(Q, acc.X), PPP, PPP));  // S1 PPP | R (Q - X3) | ZZZ3
    acc.Y = fe_sub(coop_from<1>(t), coop_from<0>(t));
    acc.ZZZ = coop_from<2>(t);
}

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
// WHICH = 0: s*A -> part[3];  1: r*B1 -> part[4].  Lanes 0..15 build the table d*P (d < 16) in LDS, then lanes 0..3 run
// 4-bit fixed windows (252 doublings + <= 64 additions, four lanes per point; exact for any curve point: no endomorphism,
// the CRS is read unchecked like the reference's)
// (gridDim.y = 2 runs both: WHICH = which0 + blockIdx.y)
__global__ void __launch_bounds__(64) k_groth16_var_mul(uint32_t which0, const G1Xyzz* __restrict__ msm_g1 /* H, L, A, B1 */,
                                                        const uint32_t* __restrict__ rs, size_t rs_stride, G1Xyzz* __restrict__ part) {
    __shared__ G1Xyzz wtab[16];
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
    }
    // same wave writes and reads the table: LDS is in order per wave, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid < 4) {  // four lanes per point (see xyzz_dbl_coop)
        const uint32_t lig = tid;
        uint32_t k[8];
        for (int i = 0; i < 8; ++i) k[i] = rs[i];
        G1Xyzz acc = xyzz_inf<FpOps>();
        for (int w = 63; w >= 0; --w) {
            if (w != 63)
                for (int q = 0; q < 4; ++q) acc = xyzz_dbl_coop(acc, lig);
            const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 15u;
            if (d) xyzz_add_coop(acc, wtab[d], lig);
        }
        if (lig == 0) part[3 + WHICH] = acc;
    }
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
