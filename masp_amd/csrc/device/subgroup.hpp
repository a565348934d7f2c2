// Membership of a curve point in the prime-order subgroups of BLS12-381 (used by the verifier for proof points and, when a
// circuit is loaded, for the CRS points whose multiples the prover takes through the endomorphism: device/groth16.hpp).
#pragma once
#include "curve.hpp"

namespace masp {

// ---- subgroup membership (M. Scott, "A note on group membership tests for G1, G2 and GT on BLS pairing-friendly curves") ----
// `groth16::Proof::read` rejects points outside the prime-order subgroups (bellman `from_compressed`; the reference parses
// proofs with it at /root/reference/masp_proofs/src/sapling/verifier/batch.rs:85,125,154).  The pairing is blind to the cofactor
// part of a point, so a verifier that skipped this test would accept malleated proofs the reference refuses.
// G1: phi(x, y) = (beta x, y) is multiplication by -u^2 exactly on the subgroup: test [u^2] P + phi(P) = O (a 128-bit multiple).
// G2: psi = twist . Frobenius . untwist is multiplication by u on the subgroup: test psi(Q) = [u] Q (a 64-bit multiple), u < 0.
// Constants: tools/gen_device_consts.py (derived there and checked on the generators).  `p` on the curve, not infinity.
__device__ inline bool g1_in_subgroup(const G1Affine& p) {
    uint32_t k[8] = {FpCfg::U_SQR[0], FpCfg::U_SQR[1], FpCfg::U_SQR[2], FpCfg::U_SQR[3], 0, 0, 0, 0};
    const G1Xyzz m = xyzz_mul_scalar(xyzz_from_affine(p), k);
    if (xyzz_is_inf(m)) return false;
    Fp beta;
    for (int i = 0; i < 12; ++i) beta.v[i] = FpCfg::ENDO_BETA[i];
    // -m == phi(p)  <=>  m.X = beta x ZZ  and  m.Y = -y ZZZ
    return fe_eq(m.X, fe_mul_nc(fe_mul_nc(beta, p.x), m.ZZ)) && fe_eq(m.Y, fe_neg(fe_mul_nc(p.y, m.ZZZ)));
}
__device__ inline bool g2_in_subgroup(const G2Affine& q) {
    uint32_t k[8] = {FpCfg::U_ABS[0], FpCfg::U_ABS[1], 0, 0, 0, 0, 0, 0};
    const G2Xyzz m = xyzz_mul_scalar(xyzz_from_affine(q), k);   // [|u|] Q = -[u] Q
    if (xyzz_is_inf(m)) return false;
    Fp2 cx, cy;
    for (int i = 0; i < 12; ++i) {
        cx.c0.v[i] = FpCfg::PSI_CX0[i];
        cx.c1.v[i] = FpCfg::PSI_CX1[i];
        cy.c0.v[i] = FpCfg::PSI_CY0[i];
        cy.c1.v[i] = FpCfg::PSI_CY1[i];
    }
    const Fp2 px = Fp2Ops::mul(cx, Fp2{q.x.c0, fe_neg(q.x.c1)}), py = Fp2Ops::mul(cy, Fp2{q.y.c0, fe_neg(q.y.c1)});
    // psi(q) == -m  <=>  m.X = px ZZ  and  m.Y = -py ZZZ
    return Fp2Ops::eq(m.X, Fp2Ops::mul(px, m.ZZ)) && Fp2Ops::eq(m.Y, Fp2Ops::neg(Fp2Ops::mul(py, m.ZZZ)));
}

}  // namespace masp
