// Eight lanes, one G2 point: the G2 twin of device/quad.hpp for the bucket tails of a lone proof's b_g2 MSM.
//
// What a lone proof waits for at the end is B2's chain of dependent G2 additions on waves that have a SIMD to themselves: over lane
// pairs (Fp2PairOps: one Fp2 per pair of lanes) an XYZZ addition is ~14 dependent Fp2 products of ~2.4 us each, and the heavy-bucket
// kernel alone strings twelve such additions together (1.4 of the 4.2 ms of a lone Spend proof, profiles/r04z_lone_proof_timeline.txt).
// Here FOUR pairs hold the same point and each computes a different product of the same dependency level, as the four lanes of a quad
// do for G1: an addition is 4 product levels instead of 14 products, a doubling 3 instead of 9.
// Lanes: group = 8 consecutive lanes (two quads of one 16-lane DPP row); pair `pig` = (lane >> 1) & 3; half = lane & 1 (c0 / c1).
// The results travel by DPP: inside a quad by quad_perm, between the two quads of a group by row_shr:4 / row_shl:4 under a bank mask.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "quad.hpp"

namespace masp {

struct OctLanes {
    static __device__ __forceinline__ uint32_t pig() { return (threadIdx.x >> 1) & 3u; }
    // the value pair SRC of this lane's group holds (same half): two DPP moves per limb
    template <int SRC>
    static __device__ __forceinline__ Fp from(const Fp& v) {
        constexpr int qp = (SRC & 1) ? 0xEE /* quad_perm [2, 3, 2, 3] */ : 0x44 /* quad_perm [0, 1, 0, 1] */;
        Fp r;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int t = __builtin_amdgcn_mov_dpp((int)v.v[i], qp, 0xf, 0xf, true);  // every quad: its own pair SRC & 1
            if constexpr ((SRC >> 1) == 0)  // the group's second quad (banks 1, 3 of the row) takes the first quad's: lane i <- lane i - 4
                r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x114 /* row_shr:4 */, 0xf, 0xA, false);
            else                            // the first quad (banks 0, 2) takes the second quad's: lane i <- lane i + 4
                r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        }
        return r;
    }
};

// add-2008-s over four pairs, same case analysis as xyzz_add (the rare P == +-Q cases are computed redundantly by the four pairs)
__device__ __forceinline__ Xyzz<Fp2OctOps> xyzz_dbl(const Xyzz<Fp2OctOps>& p);
__device__ __forceinline__ void xyzz_add_nc(Xyzz<Fp2OctOps>& acc, const Xyzz<Fp2OctOps>& b) {
    typedef Fp2PairOps P2;  // field operations on a pair's Fp2 (one half per lane)
    typedef Fp2PairCold K;  // the products by call (one copy of the 8 KB product in the kernel: inline at every site, an addition was 32 KB
                            // and a lone wave waited for its instructions — the heavy-bucket kernel ran 1.9 ms instead of 1.4 over pairs)
    if (P2::is_zero(b.ZZ)) return;
    if (P2::is_zero(acc.ZZ)) {
        acc = b;
        return;
    }
    const uint32_t g = OctLanes::pig();
    Fp t = K::mul(coop_pick(g, acc.X, b.X, acc.Y, b.Y), coop_pick(g, b.ZZ, acc.ZZ, b.ZZZ, acc.ZZZ));
    const Fp U1 = OctLanes::from<0>(t), U2 = OctLanes::from<1>(t), S1 = OctLanes::from<2>(t), S2 = OctLanes::from<3>(t);
    const Fp P = fe_sub(U2, U1), R = fe_sub(S2, S1);
    if (P2::is_zero(P)) {
        if (P2::is_zero(R))
            acc = xyzz_dbl(acc);
        else
            acc = xyzz_inf<Fp2OctOps>();
        return;
    }
    t = K::mul(coop_pick(g, P, R, acc.ZZ, acc.ZZZ), coop_pick(g, P, R, b.ZZ, b.ZZZ));  // PP | R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const Fp PP = OctLanes::from<0>(t), RR = OctLanes::from<1>(t), Z12 = OctLanes::from<2>(t), Z123 = OctLanes::from<3>(t);
    t = K::mul(coop_pick(g, P, U1, Z12, P), PP);                                          // PPP | Q | ZZ3
    const Fp PPP = OctLanes::from<0>(t), Q = OctLanes::from<1>(t);
    acc.ZZ = OctLanes::from<2>(t);
    acc.X = fe_sub(fe_sub(RR, PPP), fe_dbl(Q));
    t = K::mul(coop_pick(g, S1, R, Z123, S1), coop_pick(g, PPP, fe_sub(Q, acc.X), PPP, PPP));  // S1 PPP | R (Q - X3) | ZZZ3
    acc.Y = fe_sub(OctLanes::from<1>(t), OctLanes::from<0>(t));
    acc.ZZZ = OctLanes::from<2>(t);
}
// dbl-2008-s-1 over four pairs, same case analysis as xyzz_dbl
__device__ __forceinline__ Xyzz<Fp2OctOps> xyzz_dbl(const Xyzz<Fp2OctOps>& p) {
    typedef Fp2PairOps P2;
    typedef Fp2PairCold K;
    if (P2::is_zero(p.ZZ)) return p;
    const Fp U = fe_dbl(p.Y);
    if (P2::is_zero(U)) return xyzz_inf<Fp2OctOps>();
    const uint32_t g = OctLanes::pig();
    Fp t = K::mul(coop_pick(g, U, p.X, U, U), coop_pick(g, U, p.X, U, U));  // V = U^2 | X^2
    const Fp V = OctLanes::from<0>(t), X2 = OctLanes::from<1>(t);
    const Fp M = fe_add(fe_dbl(X2), X2);
    t = K::mul(coop_pick(g, U, p.X, M, V), coop_pick(g, V, V, M, p.ZZ));      // W = U V | S = X V | M^2 | ZZ' = V ZZ
    const Fp W = OctLanes::from<0>(t), S = OctLanes::from<1>(t), MM = OctLanes::from<2>(t);
    Xyzz<Fp2OctOps> r;
    r.ZZ = OctLanes::from<3>(t);
    r.X = fe_sub(MM, fe_dbl(S));
    t = K::mul(coop_pick(g, W, M, W, W), coop_pick(g, p.Y, fe_sub(S, r.X), p.ZZZ, W));  // W Y | M (S - X') | ZZZ' = W ZZZ
    r.Y = fe_sub(OctLanes::from<1>(t), OctLanes::from<0>(t));
    r.ZZZ = OctLanes::from<2>(t);
    return r;
}

}  // namespace masp
