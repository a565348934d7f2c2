// Four lanes, one G1 point: the serial chains of a lone proof (the variable-base multiplications of the assembly, the bucket
// tails of its MSMs) are strings of dependent 384-bit products on lanes that have a SIMD to themselves, and a lone wave cannot
// issue a product faster than its ~690 instructions take.  What it can do is run the INDEPENDENT products of one point
// operation side by side on adjacent lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"

namespace masp {

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

// FpOps over quads (field.hpp: FpQuadOps — O::LANES = 4, every lane of a quad holds the whole point): the bucket tails of a
// lone proof's G1 MSMs (device/msm.hpp) call these through the names they use for every other O
__device__ __forceinline__ uint32_t quad_lane() { return __lane_id() & 3u; }
__device__ __forceinline__ Xyzz<FpQuadOps> xyzz_dbl(const Xyzz<FpQuadOps>& p) {
    const G1Xyzz r = xyzz_dbl_coop(reinterpret_cast<const G1Xyzz&>(p), quad_lane());
    return reinterpret_cast<const Xyzz<FpQuadOps>&>(r);
}
__device__ __forceinline__ void xyzz_add_nc(Xyzz<FpQuadOps>& acc, const Xyzz<FpQuadOps>& b) {
    xyzz_add_coop(reinterpret_cast<G1Xyzz&>(acc), reinterpret_cast<const G1Xyzz&>(b), quad_lane());
}

}  // namespace masp
