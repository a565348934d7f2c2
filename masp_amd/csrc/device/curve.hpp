// BLS12-381 G1 / G2 group law for gfx950, written once over a field-ops policy (FpOps / Fp2Ops).
//
// Accumulators use extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed addition costs 8M + 2S and needs no inversion; infinity is ZZ = 0.  Every exceptional
// case (P + P, P + (-P), infinity operands) is handled exactly, because proof bytes must be
// bit-identical to the CPU prover for *any* CRS, including ones with repeated bases.
// Replaces blst's point arithmetic used through `bls12_381::{G1Affine,G1Projective,G2...}`
// (nam-blstrs / nam-blst, /root/reference/Cargo.lock:1385-1411).  Curve: y^2 = x^3 + 4 over Fp,
// y^2 = x^3 + 4(1+u) over Fp2; a = 0 in both, which is all the formulas need.
#pragma once
#include "field.hpp"

namespace masp {

// Affine point.  Infinity is encoded as x = y = 0 (not on either curve), blst's convention.
template <class O>
struct Affine {
    typename O::T x, y;
};
template <class O>
struct Xyzz {
    typename O::T X, Y, ZZ, ZZZ;
};
// Row of an MSM window table: an affine point padded to whole 128-byte lines (96 -> 128 B on G1, 192 -> 256 B on G2), so
// that a gathered row never straddles two lines (at a 96-byte stride half of all rows did: ~1.5 lines fetched per row).
template <class O>
struct alignas(128) TabRow {
    Affine<O> p;
};

template <class O>
MASP_HD bool aff_is_inf(const Affine<O>& p) {
    return O::is_zero(p.x) && O::is_zero(p.y);
}
template <class O>
MASP_HD Xyzz<O> xyzz_inf() {
    Xyzz<O> r;
    r.X = O::zero();
    r.Y = O::zero();
    r.ZZ = O::zero();
    r.ZZZ = O::zero();
    return r;
}
template <class O>
MASP_HD bool xyzz_is_inf(const Xyzz<O>& p) {
    return O::is_zero(p.ZZ);
}
template <class O>
MASP_HD Xyzz<O> xyzz_from_affine(const Affine<O>& a) {
    if (aff_is_inf(a)) return xyzz_inf<O>();
    Xyzz<O> r;
    r.X = a.x;
    r.Y = a.y;
    r.ZZ = O::one();
    r.ZZZ = O::one();
    return r;
}
template <class O>
MASP_HD Xyzz<O> xyzz_neg(const Xyzz<O>& p) {
    Xyzz<O> r = p;
    r.Y = O::neg(p.Y);
    return r;
}

// dbl-2008-s-1 for XYZZ.  Reached from the rare P == Q branch of the additions and from serial tails: its products go
// through the Cold multiplier policy (register-argument calls, field.hpp), so inlining it costs ~1.5 KiB per site while
// the point itself never leaves the VGPRs.
template <class O>
MASP_HD Xyzz<O> xyzz_dbl(const Xyzz<O>& p) {
    if (xyzz_is_inf(p)) return p;
    typedef typename O::T F;
    typedef typename O::Cold K;
    F U = O::dbl(p.Y);
    if (O::is_zero(U)) return xyzz_inf<O>();  // order-2 point: cannot occur on these curves, kept for exactness
    F V = K::sqr(U);
    F W = K::mul(U, V);
    F S = K::mul(p.X, V);
    F X2 = K::sqr(p.X);
    F M = O::add(O::dbl(X2), X2);
    Xyzz<O> r;
    r.X = O::sub(K::sqr(M), O::dbl(S));
    r.Y = O::sub(K::mul(M, O::sub(S, r.X)), K::mul(W, p.Y));
    r.ZZ = K::mul(V, p.ZZ);
    r.ZZZ = K::mul(W, p.ZZZ);
    return r;
}
// doubling of an affine point (mdbl-2008-s-1)
template <class O>
MASP_HD Xyzz<O> xyzz_dbl_affine(const Affine<O>& p) {
    if (aff_is_inf(p)) return xyzz_inf<O>();
    typedef typename O::T F;
    typedef typename O::Cold K;
    F U = O::dbl(p.y);
    if (O::is_zero(U)) return xyzz_inf<O>();
    F V = K::sqr(U);
    F W = K::mul(U, V);
    F S = K::mul(p.x, V);
    F X2 = K::sqr(p.x);
    F M = O::add(O::dbl(X2), X2);
    Xyzz<O> r;
    r.X = O::sub(K::sqr(M), O::dbl(S));
    r.Y = O::sub(K::mul(M, O::sub(S, r.X)), K::mul(W, p.y));
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

// the two exceptional cases of a mixed addition (same x: P + P or P - P), out of line for G2: they practically never run, and
// inline their doubling widens the register allocation of every accumulation loop (see xyzz_madd)
template <class O>
MASP_NOINLINE void xyzz_madd_same_x(Xyzz<O>& acc, const typename O::T& bx, const typename O::T& by, bool same_y) {
    if (same_y) {
        Affine<O> t;
        t.x = bx;
        t.y = by;
        acc = xyzz_dbl_affine(t);
    } else {
        acc = xyzz_inf<O>();
    }
}

// acc += (negate ? -b : b), b affine (madd-2008-s)
template <class O, class M = O>
MASP_HD void xyzz_madd(Xyzz<O>& acc, const Affine<O>& b, bool negate) {
    typedef typename O::T F;
    if (aff_is_inf(b)) return;
    F by = negate ? O::neg(b.y) : b.y;
    if (xyzz_is_inf(acc)) {
        acc.X = b.x;
        acc.Y = by;
        acc.ZZ = O::one();
        acc.ZZZ = O::one();
        return;
    }
    F U2 = M::mul(b.x, acc.ZZ);
    F S2 = M::mul(by, acc.ZZZ);
    F P = O::sub(U2, acc.X);
    F R = O::sub(S2, acc.Y);
    if constexpr (sizeof(F) > 48) {
        // Fp2 (products are calls, the point lives in 256 architectural registers): the exceptional cases out of line and
        // every value consumed as early as the formulas allow (ZZ, ZZZ, X, Y are replaced as soon as PP / PPP exist) — the
        // accumulation loop then has no register spills (414 before), its launch went from 65.5 to 60.1 ms.  The same two
        // changes make the G1 loop (inline products, 228 registers) 13 % SLOWER: it keeps the plain form below.
        if (O::is_zero(P)) {
            xyzz_madd_same_x(acc, b.x, by, O::is_zero(R));
            return;
        }
        F PP = M::sqr(P);
        acc.ZZ = M::mul(acc.ZZ, PP);
        F PPP = M::mul(P, PP);
        acc.ZZZ = M::mul(acc.ZZZ, PPP);
        F Q = M::mul(acc.X, PP);
        F T = M::mul(acc.Y, PPP);
        acc.X = O::sub(O::sub(M::sqr(R), PPP), O::dbl(Q));
        acc.Y = O::sub(M::mul(R, O::sub(Q, acc.X)), T);
    } else {
        if (O::is_zero(P)) {
            if (O::is_zero(R)) {
                Affine<O> t;
                t.x = b.x;
                t.y = by;
                acc = xyzz_dbl_affine(t);
            } else {
                acc = xyzz_inf<O>();
            }
            return;
        }
        F PP = M::sqr(P);
        F PPP = M::mul(P, PP);
        F Q = M::mul(acc.X, PP);
        F X3 = O::sub(O::sub(M::sqr(R), PPP), O::dbl(Q));
        F Y3 = O::sub(M::mul(R, O::sub(Q, X3)), M::mul(acc.Y, PPP));
        acc.X = X3;
        acc.Y = Y3;
        acc.ZZ = M::mul(acc.ZZ, PP);
        acc.ZZZ = M::mul(acc.ZZZ, PPP);
    }
}

// the exceptional cases of a full addition (same x: doubling or the point at infinity), out of line for G2 (see xyzz_madd)
template <class O>
MASP_NOINLINE void xyzz_add_same_x(Xyzz<O>& acc, bool same_y) {
    if (same_y)
        acc = xyzz_dbl(acc);
    else
        acc = xyzz_inf<O>();
}

// acc += b, both XYZZ (add-2008-s)
template <class O, class M = O>
MASP_HD void xyzz_add(Xyzz<O>& acc, const Xyzz<O>& b) {
    typedef typename O::T F;
    if (xyzz_is_inf(b)) return;
    if (xyzz_is_inf(acc)) {
        acc = b;
        return;
    }
    if constexpr (sizeof(F) > 48) {
        // Fp2: as in xyzz_madd — every input is consumed as early as possible, the exceptional cases are a call
        F U1 = M::mul(acc.X, b.ZZ);
        F P = O::sub(M::mul(b.X, acc.ZZ), U1);
        F S1 = M::mul(acc.Y, b.ZZZ);
        F R = O::sub(M::mul(b.Y, acc.ZZZ), S1);
        if (O::is_zero(P)) {
            xyzz_add_same_x(acc, O::is_zero(R));
            return;
        }
        acc.ZZ = M::mul(acc.ZZ, b.ZZ);
        acc.ZZZ = M::mul(acc.ZZZ, b.ZZZ);
        F PP = M::sqr(P);
        acc.ZZ = M::mul(acc.ZZ, PP);
        F PPP = M::mul(P, PP);
        acc.ZZZ = M::mul(acc.ZZZ, PPP);
        F Q = M::mul(U1, PP);
        F T = M::mul(S1, PPP);
        acc.X = O::sub(O::sub(M::sqr(R), PPP), O::dbl(Q));
        acc.Y = O::sub(M::mul(R, O::sub(Q, acc.X)), T);
    } else {
        F U1 = M::mul(acc.X, b.ZZ);
        F U2 = M::mul(b.X, acc.ZZ);
        F S1 = M::mul(acc.Y, b.ZZZ);
        F S2 = M::mul(b.Y, acc.ZZZ);
        F P = O::sub(U2, U1);
        F R = O::sub(S2, S1);
        if (O::is_zero(P)) {
            if (O::is_zero(R))
                acc = xyzz_dbl(acc);
            else
                acc = xyzz_inf<O>();
            return;
        }
        F PP = M::sqr(P);
        F PPP = M::mul(P, PP);
        F Q = M::mul(U1, PP);
        F X3 = O::sub(O::sub(M::sqr(R), PPP), O::dbl(Q));
        F Y3 = O::sub(M::mul(R, O::sub(Q, X3)), M::mul(S1, PPP));
        acc.X = X3;
        acc.Y = Y3;
        acc.ZZ = M::mul(M::mul(acc.ZZ, b.ZZ), PP);
        acc.ZZZ = M::mul(M::mul(acc.ZZZ, b.ZZZ), PPP);
    }
}

// forms for cold kernels and serial tails: the group law inline (additions / subtractions, ~4 KiB), every field product
// a call with register arguments — small code, points stay in VGPRs
template <class O>
MASP_HD void xyzz_add_nc(Xyzz<O>& acc, const Xyzz<O>& b) {
    xyzz_add<O, typename O::Cold>(acc, b);
}
template <class O>
MASP_HD void xyzz_madd_nc(Xyzz<O>& acc, const Affine<O>& b, bool negate) {
    xyzz_madd<O, typename O::Cold>(acc, b, negate);
}
// [k]p for a 256-bit scalar given as 8 little-endian canonical limbs
template <class O>
MASP_HD Xyzz<O> xyzz_mul_scalar(const Xyzz<O>& p, const uint32_t* k) {
    Xyzz<O> r = xyzz_inf<O>();
    bool started = false;
    for (int i = 7; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) r = xyzz_dbl(r);
            if ((k[i] >> b) & 1) {
                xyzz_add_nc(r, p);
                started = true;
            }
        }
    return r;
}

// LONE: called by a single lane on a serial tail (see FpOps::inv_lone)
template <class O, bool LONE = false>
MASP_HD Affine<O> xyzz_to_affine(const Xyzz<O>& p) {
    Affine<O> r;
    if (xyzz_is_inf(p)) {
        r.x = O::zero();
        r.y = O::zero();
        return r;
    }
    // 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2 ... cheaper: zi3 = 1/ZZZ ; zi2 = (zi3 * ZZ)^2  (since ZZ^3 = ZZZ^2 => ZZ/ZZZ = 1/Z)
    typename O::T zi3 = LONE ? O::inv_lone(p.ZZZ) : O::inv(p.ZZZ);
    typename O::T zi = O::Cold::mul(zi3, p.ZZ);
    typename O::T zi2 = O::Cold::sqr(zi);
    r.x = O::Cold::mul(p.X, zi2);
    r.y = O::Cold::mul(p.Y, zi3);
    return r;
}

typedef Affine<FpOps> G1Affine;
typedef Affine<Fp2Ops> G2Affine;
typedef Xyzz<FpOps> G1Xyzz;
typedef Xyzz<Fp2Ops> G2Xyzz;

// the verifying-key points the prover itself uses (proof assembly)
struct VkDevice {
    G1Affine alpha_g1, beta_g1, delta_g1;
    G2Affine beta_g2, delta_g2;
};

}  // namespace masp
