// Host-side Groth16 verification for the prover's self-check:
//   `verify_proof(verifying_key, &proof, &public_input)` at /root/reference/masp_proofs/src/sapling/prover.rs:148,266
//   with `PreparedVerifyingKey` built at /root/reference/masp_proofs/src/lib.rs:391-393
// (nam-bellperson / pairing 0.23 / blst, un-vendored).  SURVEY.md §8 row a12: "CPU, ~ms; stays on host".
// BLS12-381 base field in 6 x 64-bit limbs, the Fp2/Fp6/Fp12 tower, zcash point decoding, a multi-pair ate Miller loop
// (G2 in homogeneous projective coordinates on the twist, sparse line evaluations, one shared squaring chain) and the
// final exponentiation  f^(3 (p^12-1)/r)  =  easy part (p^6-1)(p^2+1) by conjugation / inversion / Frobenius, hard part
//   3 (p^4-p^2+1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3      (x = -0xd201000000010000; identity checked in tools/)
// as five exponentiations by |x|.  The factor 3 is coprime to r, so "== 1" is unchanged.
// Product code (libmasp_host); independent of oracle/ (whose own slow pairing is what the tests compare against).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../device/consts.hpp"
#include "mont.h"

namespace masp_host {
namespace bls {

struct Fp {
    uint64_t l[6];
    static const uint64_t* P() {
        static const uint64_t m[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
        return m;
    }
    struct K {
        uint64_t r1[6], r2[6], inv;
    };
    static bool ge(const uint64_t* a, const uint64_t* b) {
        for (int i = 5; i >= 0; --i) {
            if (a[i] != b[i]) return a[i] > b[i];
        }
        return true;
    }
    static uint64_t addr(uint64_t* r, const uint64_t* a, const uint64_t* b) {
        u128 c = 0;
        for (int i = 0; i < 6; ++i) {
            c += (u128)a[i] + b[i];
            r[i] = (uint64_t)c;
            c >>= 64;
        }
        return (uint64_t)c;
    }
    static uint64_t subr(uint64_t* r, const uint64_t* a, const uint64_t* b) {
        uint64_t br = 0;
        for (int i = 0; i < 6; ++i) {
            u128 d = (u128)a[i] - b[i] - br;
            r[i] = (uint64_t)d;
            br = (uint64_t)(d >> 64) & 1;
        }
        return br;
    }
    static const K& k() {
        static K c = [] {
            K x;
            uint64_t v = 1;
            for (int i = 0; i < 7; ++i) v *= 2 - P()[0] * v;
            x.inv = (uint64_t)0 - v;
            uint64_t t[6] = {1, 0, 0, 0, 0, 0};
            for (int s = 0; s < 768; ++s) {
                uint64_t carry = addr(t, t, t);
                if (carry || ge(t, P())) subr(t, t, P());
                if (s == 383) memcpy(x.r1, t, 48);
            }
            memcpy(x.r2, t, 48);
            return x;
        }();
        return c;
    }
    static void mm(uint64_t* out, const uint64_t* a, const uint64_t* b) { mont_mul_n<6>(out, a, b, P(), k().inv); }
    static Fp zero() {
        Fp r;
        memset(r.l, 0, 48);
        return r;
    }
    static Fp one() {
        Fp r;
        memcpy(r.l, k().r1, 48);
        return r;
    }
    static Fp from_u64(uint64_t v) {
        uint64_t t[6] = {v, 0, 0, 0, 0, 0};
        Fp r;
        mm(r.l, t, k().r2);
        return r;
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3] | l[4] | l[5]) == 0; }
    bool operator==(const Fp& o) const { return memcmp(l, o.l, 48) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }
    Fp operator+(const Fp& o) const {
        Fp r;
        uint64_t c = addr(r.l, l, o.l);
        if (c || ge(r.l, P())) subr(r.l, r.l, P());
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r;
        if (subr(r.l, l, o.l)) addr(r.l, r.l, P());
        return r;
    }
    Fp neg() const { return zero() - *this; }
    Fp dbl() const { return *this + *this; }
    Fp operator*(const Fp& o) const {
        Fp r;
        mm(r.l, l, o.l);
        return r;
    }
    Fp sq() const { return *this * *this; }
    Fp pow(const uint64_t* e, int n) const {
        Fp r = one();
        for (int i = n - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                r = r.sq();
                if ((e[i] >> b) & 1) r = r * *this;
            }
        return r;
    }
    Fp inv() const {  // 0 -> 0, like a Fermat power
        static const ModInv<6> mi(P());
        uint64_t r[6], t[6];
        if (!mi.invert(r, l)) return zero();
        Fp o;
        mm(t, r, k().r2);  // (aR)^-1 R^2 R^-1 = a^-1
        mm(o.l, t, k().r2);
        return o;
    }
    // big-endian 48 bytes, canonical
    static bool from_be(Fp& out, const uint8_t* b) {
        uint64_t v[6];
        for (int i = 0; i < 6; ++i) {
            uint64_t x = 0;
            for (int k2 = 0; k2 < 8; ++k2) x = (x << 8) | b[8 * (5 - i) + k2];
            v[i] = x;
        }
        if (ge(v, P())) return false;
        mm(out.l, v, k().r2);
        return true;
    }
    void canon(uint64_t* v) const {
        uint64_t o[6] = {1, 0, 0, 0, 0, 0};
        mm(v, l, o);
    }
    bool lex_largest() const {  // value > (p-1)/2
        uint64_t a[6], b[6];
        canon(a);
        neg().canon(b);
        for (int i = 5; i >= 0; --i)
            if (a[i] != b[i]) return a[i] > b[i];
        return false;
    }
    bool sqrt(Fp& out) const {  // p = 3 mod 4
        uint64_t e[6], one_[6] = {1, 0, 0, 0, 0, 0};
        addr(e, P(), one_);
        for (int i = 0; i < 6; ++i) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
        Fp r = pow(e, 6);
        if (r.sq() != *this) return false;
        out = r;
        return true;
    }
};

struct Fp2 {
    Fp a, b;  // a + b u, u^2 = -1
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
    bool operator==(const Fp2& o) const { return a == o.a && b == o.b; }
    Fp2 operator+(const Fp2& o) const { return {a + o.a, b + o.b}; }
    Fp2 operator-(const Fp2& o) const { return {a - o.a, b - o.b}; }
    Fp2 neg() const { return {a.neg(), b.neg()}; }
    Fp2 operator*(const Fp2& o) const {
        Fp t0 = a * o.a, t1 = b * o.b;
        return {t0 - t1, (a + b) * (o.a + o.b) - t0 - t1};
    }
    Fp2 sq() const { return {(a + b) * (a - b), (a * b).dbl()}; }
    Fp2 xi() const { return {a - b, a + b}; }  // * (1 + u)
    Fp2 inv() const {
        Fp n = (a.sq() + b.sq()).inv();
        return {a * n, (b * n).neg()};
    }
    bool sqrt(Fp2& out) const {
        if (is_zero()) {
            out = *this;
            return true;
        }
        if (b.is_zero()) {
            Fp s;
            if (a.sqrt(s)) {
                out = {s, Fp::zero()};
                return true;
            }
            if (a.neg().sqrt(s)) {
                out = {Fp::zero(), s};
                return true;
            }
            return false;
        }
        Fp n;
        if (!(a.sq() + b.sq()).sqrt(n)) return false;
        Fp half = Fp::from_u64(2).inv();
        Fp d = (a + n) * half, x0;
        if (!d.sqrt(x0)) {
            d = (a - n) * half;
            if (!d.sqrt(x0)) return false;
        }
        Fp x1 = b * x0.dbl().inv();
        out = {x0, x1};
        return out.sq() == *this;
    }
};
struct Fp6 {
    Fp2 a, b, c;  // a + b v + c v^2, v^3 = xi
    static Fp6 zero() { return {Fp2::zero(), Fp2::zero(), Fp2::zero()}; }
    static Fp6 one() { return {Fp2::one(), Fp2::zero(), Fp2::zero()}; }
    bool operator==(const Fp6& o) const { return a == o.a && b == o.b && c == o.c; }
    Fp6 operator+(const Fp6& o) const { return {a + o.a, b + o.b, c + o.c}; }
    Fp6 operator-(const Fp6& o) const { return {a - o.a, b - o.b, c - o.c}; }
    Fp6 neg() const { return {a.neg(), b.neg(), c.neg()}; }
    Fp6 operator*(const Fp6& o) const {  // Karatsuba: 6 Fp2 products
        Fp2 v0 = a * o.a, v1 = b * o.b, v2 = c * o.c;
        return {v0 + ((b + c) * (o.b + o.c) - v1 - v2).xi(), (a + b) * (o.a + o.b) - v0 - v1 + v2.xi(), (a + c) * (o.a + o.c) - v0 - v2 + v1};
    }
    Fp6 mul_01(const Fp2& x0, const Fp2& x1) const {  // * (x0 + x1 v)
        Fp2 v0 = a * x0, v1 = b * x1;
        return {v0 + (c * x1).xi(), (a + b) * (x0 + x1) - v0 - v1, v1 + c * x0};
    }
    Fp6 mul_1(const Fp2& x1) const { return {(c * x1).xi(), a * x1, b * x1}; }  // * (x1 v)
    Fp6 mulv() const { return {c.xi(), a, b}; }
    Fp6 inv() const {
        Fp2 t0 = a.sq() - (b * c).xi(), t1 = c.sq().xi() - a * b, t2 = b.sq() - a * c;
        Fp2 d = (a * t0 + (c * t1 + b * t2).xi()).inv();
        return {t0 * d, t1 * d, t2 * d};
    }
};
struct Fp12 {
    Fp6 a, b;  // a + b w, w^2 = v
    static Fp12 one() { return {Fp6::one(), Fp6::zero()}; }
    bool operator==(const Fp12& o) const { return a == o.a && b == o.b; }
    Fp12 operator*(const Fp12& o) const {
        Fp6 t0 = a * o.a, t1 = b * o.b;
        return {t0 + t1.mulv(), (a + b) * (o.a + o.b) - t0 - t1};
    }
    Fp12 sq() const {  // (a + b w)^2 = (a^2 + b^2 v) + 2ab w  with two Fp6 products
        Fp6 ab = a * b;
        return {(a + b) * (a + b.mulv()) - ab - ab.mulv(), ab + ab};
    }
    // * (c0 + c1 v + c2 v w): the shape of a line evaluation
    Fp12 mul_line(const Fp2& c0, const Fp2& c1, const Fp2& c2) const {
        Fp6 t0 = a.mul_01(c0, c1), t1 = b.mul_1(c2);
        return {t0 + t1.mulv(), (a + b).mul_01(c0, c1 + c2) - t0 - t1};
    }
    Fp12 operator-(const Fp12& o) const { return {a - o.a, b - o.b}; }
    Fp12 operator+(const Fp12& o) const { return {a + o.a, b + o.b}; }
    Fp12 conj() const { return {a, b.neg()}; }  // = x^(p^6)
    Fp12 inv() const {
        Fp6 d = (a * a - (b * b).mulv()).inv();
        return {a * d, (b * d).neg()};
    }
    static Fp12 from_fp(const Fp& x) {
        Fp12 r = {Fp6::zero(), Fp6::zero()};
        r.a.a.a = x;
        return r;
    }
    static Fp12 from_fp2(const Fp2& x) {
        Fp12 r = {Fp6::zero(), Fp6::zero()};
        r.a.a = x;
        return r;
    }
    Fp12 pow(const std::vector<uint64_t>& e) const {
        Fp12 r = one();
        bool started = false;
        for (int i = (int)e.size() - 1; i >= 0; --i)
            for (int bit = 63; bit >= 0; --bit) {
                if (started) r = r.sq();
                if ((e[i] >> bit) & 1) {
                    r = started ? r * *this : *this;
                    started = true;
                }
            }
        return r;
    }
};

struct G1A {
    Fp x, y;
    bool inf;
};
struct G2A {
    Fp2 x, y;
    bool inf;
};
// Jacobian G1 for the public-input linear combination
struct G1J {
    Fp X, Y, Z;
    static G1J inf() { return {Fp::one(), Fp::one(), Fp::zero()}; }
    static G1J from(const G1A& a) { return a.inf ? inf() : G1J{a.x, a.y, Fp::one()}; }
    G1J dbl() const {
        if (Z.is_zero()) return *this;
        Fp A = X.sq(), B = Y.sq(), C = B.sq();
        Fp D = ((X + B).sq() - A - C).dbl();
        Fp E = A.dbl() + A;
        Fp X3 = E.sq() - D.dbl();
        return {X3, E * (D - X3) - C.dbl().dbl().dbl(), (Y * Z).dbl()};
    }
    G1J add(const G1J& o) const {
        if (Z.is_zero()) return o;
        if (o.Z.is_zero()) return *this;
        Fp z1 = Z.sq(), z2 = o.Z.sq();
        Fp u1 = X * z2, u2 = o.X * z1, s1 = Y * o.Z * z2, s2 = o.Y * Z * z1;
        if (u1 == u2) return s1 == s2 ? dbl() : inf();
        Fp H = u2 - u1, I = H.dbl().sq(), J = H * I, r = (s2 - s1).dbl(), V = u1 * I;
        Fp X3 = r.sq() - J - V.dbl();
        return {X3, r * (V - X3) - (s1 * J).dbl(), ((Z + o.Z).sq() - z1 - z2) * H};
    }
    G1J mul_le(const uint8_t* k32, int bits = 256) const {
        G1J r = inf();
        for (int i = bits - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k32[i / 8] >> (i % 8)) & 1) r = r.add(*this);
        }
        return r;
    }
    G1A affine() const {
        if (Z.is_zero()) return {Fp::zero(), Fp::zero(), true};
        Fp zi = Z.inv(), zi2 = zi.sq();
        return {X * zi2, Y * zi2 * zi, false};
    }
};

inline bool g1_uncompressed(G1A& p, const uint8_t* in) {
    if (in[0] & 0x80) return false;
    if (in[0] & 0x40) {
        p = {Fp::zero(), Fp::zero(), true};
        return true;
    }
    p.inf = false;
    return Fp::from_be(p.x, in) && Fp::from_be(p.y, in + 48);
}
inline bool g2_uncompressed(G2A& p, const uint8_t* in) {
    if (in[0] & 0x80) return false;
    if (in[0] & 0x40) {
        p = {Fp2::zero(), Fp2::zero(), true};
        return true;
    }
    p.inf = false;
    return Fp::from_be(p.x.b, in) && Fp::from_be(p.x.a, in + 48) && Fp::from_be(p.y.b, in + 96) && Fp::from_be(p.y.a, in + 144);
}
// ---- subgroup membership (same tests and constants as the device verifier, device/pairing.hpp) ------------------------
// `groth16::Proof::read` refuses points outside the prime-order subgroups (/root/reference/masp_proofs/src/sapling/verifier/
// batch.rs:85,125,154 parse with it); the pairing cannot see the cofactor part of a point, so a verifier without this test
// accepts malleated proofs.  G1: (beta x, y) = -[u^2] P.  G2: psi(Q) = [u] Q.  (M. Scott's membership tests.)
template <class F>
struct JacT {  // Jacobian coordinates over Fp or Fp2, a = 0
    F X, Y, Z;
    bool is_inf() const { return Z.is_zero(); }
    JacT dbl() const {
        if (Z.is_zero()) return *this;
        F A = X.sq(), B = Y.sq(), C = B.sq();
        F t = (X + B).sq() - A - C, D = t + t, E = A + A + A;
        F X3 = E.sq() - (D + D), C8 = C + C;
        C8 = C8 + C8;
        C8 = C8 + C8;
        F YZ = Y * Z;
        return {X3, E * (D - X3) - C8, YZ + YZ};
    }
    JacT add_affine(const F& x, const F& y) const {  // + (x, y), which is not infinity
        if (Z.is_zero()) return {x, y, F::one()};
        F z1 = Z.sq(), u2 = x * z1, s2 = y * Z * z1;
        if (u2 == X) return s2 == Y ? dbl() : JacT{F::one(), F::one(), F::zero()};
        F H = u2 - X, HH = H.sq(), I = HH + HH;
        I = I + I;
        F J = H * I, r = s2 - Y;
        r = r + r;
        F V = X * I, X3 = r.sq() - J - (V + V), YJ = Y * J;
        return {X3, r * (V - X3) - (YJ + YJ), (Z + H).sq() - z1 - HH};
    }
    static JacT mul_u64(const F& x, const F& y, const uint64_t* k, int limbs) {
        JacT r{F::one(), F::one(), F::zero()};
        for (int i = 64 * limbs - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) r = r.add_affine(x, y);
        }
        return r;
    }
};
inline Fp fp_from_mont_limbs32(const uint32_t* v) {  // device constants (32-bit limbs, radix 2^384) are the same bytes as ours
    Fp r;
    memcpy(r.l, v, 48);
    return r;
}
inline bool g1_in_subgroup(const G1A& p) {  // p on the curve, not infinity
    const uint64_t k[2] = {(uint64_t)masp::FpCfg::U_SQR[0] | ((uint64_t)masp::FpCfg::U_SQR[1] << 32),
                           (uint64_t)masp::FpCfg::U_SQR[2] | ((uint64_t)masp::FpCfg::U_SQR[3] << 32)};
    const JacT<Fp> m = JacT<Fp>::mul_u64(p.x, p.y, k, 2);
    if (m.is_inf()) return false;
    const Fp z2 = m.Z.sq(), z3 = z2 * m.Z, bx = fp_from_mont_limbs32(masp::FpCfg::ENDO_BETA) * p.x;
    return m.X == bx * z2 && m.Y == (p.y * z3).neg();  // [u^2] P = -phi(P)
}
inline bool g2_in_subgroup(const G2A& q) {
    const uint64_t k[1] = {(uint64_t)masp::FpCfg::U_ABS[0] | ((uint64_t)masp::FpCfg::U_ABS[1] << 32)};
    const JacT<Fp2> m = JacT<Fp2>::mul_u64(q.x, q.y, k, 1);  // [|u|] Q = -[u] Q
    if (m.is_inf()) return false;
    const Fp2 cx{fp_from_mont_limbs32(masp::FpCfg::PSI_CX0), fp_from_mont_limbs32(masp::FpCfg::PSI_CX1)};
    const Fp2 cy{fp_from_mont_limbs32(masp::FpCfg::PSI_CY0), fp_from_mont_limbs32(masp::FpCfg::PSI_CY1)};
    const Fp2 px = cx * Fp2{q.x.a, q.x.b.neg()}, py = cy * Fp2{q.y.a, q.y.b.neg()};
    const Fp2 z2 = m.Z.sq(), z3 = z2 * m.Z;
    return m.X == px * z2 && m.Y == (py * z3).neg();
}
// the point at infinity is the compression + infinity flags and nothing else (bellman rejects stray bits)
inline bool infinity_encoding_is_clean(const uint8_t* in, int len) {
    if (in[0] & 0x3f) return false;
    for (int i = 1; i < len; ++i)
        if (in[i]) return false;
    return true;
}

// zcash compressed encodings as `Proof::read` accepts them: canonical, on the curve AND in the prime-order subgroup
inline bool g1_compressed(G1A& p, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        p = {Fp::zero(), Fp::zero(), true};
        return infinity_encoding_is_clean(in, 48);
    }
    uint8_t t[48];
    memcpy(t, in, 48);
    bool big = t[0] & 0x20;
    t[0] &= 0x1f;
    if (!Fp::from_be(p.x, t)) return false;
    if (!(p.x.sq() * p.x + Fp::from_u64(4)).sqrt(p.y)) return false;
    if (p.y.lex_largest() != big) p.y = p.y.neg();
    p.inf = false;
    return g1_in_subgroup(p);
}
inline bool g2_compressed(G2A& p, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        p = {Fp2::zero(), Fp2::zero(), true};
        return infinity_encoding_is_clean(in, 96);
    }
    uint8_t t[96];
    memcpy(t, in, 96);
    bool big = t[0] & 0x20;
    t[0] &= 0x1f;
    if (!Fp::from_be(p.x.b, t) || !Fp::from_be(p.x.a, t + 48)) return false;
    Fp2 rhs = p.x.sq() * p.x + Fp2{Fp::from_u64(4), Fp::from_u64(4)};
    if (!rhs.sqrt(p.y)) return false;
    bool lg = p.y.b.is_zero() ? p.y.a.lex_largest() : p.y.b.lex_largest();
    if (lg != big) p.y = p.y.neg();
    p.inf = false;
    return g2_in_subgroup(p);
}

// ---- pairing ---------------------------------------------------------------------------------------------
struct PairingK {
    Fp2 gamma[6];  // gamma[i] = xi^(i (p-1)/6): (c w^i)^p = conj(c) gamma[i] w^i
};

inline const PairingK& pairing_k() {
    static PairingK k = [] {
        PairingK c;
        // (p - 1) / 6
        uint64_t e[6], one_[6] = {1, 0, 0, 0, 0, 0};
        Fp::subr(e, Fp::P(), one_);
        uint64_t rem = 0;
        for (int i = 5; i >= 0; --i) {
            u128 cur = ((u128)rem << 64) | e[i];
            e[i] = (uint64_t)(cur / 6);
            rem = (uint64_t)(cur % 6);
        }
        Fp2 xi = {Fp::one(), Fp::one()}, g = Fp2::one();
        for (int i = 5; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                g = g.sq();
                if ((e[i] >> b) & 1) g = g * xi;
            }
        c.gamma[0] = Fp2::one();
        for (int i = 1; i < 6; ++i) c.gamma[i] = c.gamma[i - 1] * g;
        return c;
    }();
    return k;
}

inline Fp2 fp2_conj(const Fp2& x) { return {x.a, x.b.neg()}; }
inline Fp2 fp2_scale(const Fp2& x, const Fp& k) { return {x.a * k, x.b * k}; }
// x -> x^p.  Coefficient of w^i: a.a w^0, b.a w^1, a.b w^2, b.b w^3, a.c w^4, b.c w^5
inline Fp12 frobenius(const Fp12& f) {
    const Fp2* g = pairing_k().gamma;
    return {{fp2_conj(f.a.a), fp2_conj(f.a.b) * g[2], fp2_conj(f.a.c) * g[4]},
            {fp2_conj(f.b.a) * g[1], fp2_conj(f.b.b) * g[3], fp2_conj(f.b.c) * g[5]}};
}

// G2 point in homogeneous projective coordinates on the twist  y^2 = x^3 + 4 xi
struct G2P {
    Fp2 X, Y, Z;
};
// One pair of the Miller loop.  A line through points of the twist, untwisted by (x', y') -> (x'/w^2, y'/w^3) and
// evaluated at P = (xp, yp), is  yp - (lam xp)/w + (lam x' - y')/w^3 ; multiplied by w^3 and by an Fp2 scale factor
// (both die in the final exponentiation) it is  c0 + c1 v + c2 v w  with the coefficients below.
struct MillerPair {
    Fp xp, yp;
    Fp2 xq, yq;
    G2P T;
    MillerPair(const G1A& P, const G2A& Q) : xp(P.x), yp(P.y), xq(Q.x), yq(Q.y), T{Q.x, Q.y, Fp2::one()} {}
    // T <- 2T ; f <- f * l_{T,T}(P)
    void dbl_step(Fp12& f) {
        Fp2 Y2 = T.Y.sq(), Z2 = T.Z.sq(), X2 = T.X.sq();
        Fp2 bz = Z2.xi();  // b' Z^2 / 4
        bz = bz + bz;
        bz = bz + bz;                   // b' Z^2
        Fp2 E = bz + bz + bz;           // 3 b' Z^2
        Fp2 YZ2 = T.Y * T.Z;
        YZ2 = YZ2 + YZ2;                // 2 Y Z
        Fp2 X23 = X2 + X2 + X2;         // 3 X^2
        f = f.mul_line(Y2 - E, fp2_scale(X23, xp).neg(), fp2_scale(YZ2, yp));
        Fp2 E3 = E + E + E, XY = T.X * T.Y;
        Fp2 EY = E * Y2, E2 = E.sq();
        Fp2 EY2 = EY + EY, EY6 = EY2 + EY2 + EY2;
        T.X = (XY + XY) * (Y2 - E3);
        T.Y = Y2.sq() + EY6 - (E2 + E2 + E2);
        Fp2 Y2Z = Y2 * YZ2;             // 2 Y^3 Z
        Y2Z = Y2Z + Y2Z;
        T.Z = Y2Z + Y2Z;                // 8 Y^3 Z
    }
    // T <- T + Q ; f <- f * l_{T,Q}(P)
    void add_step(Fp12& f) {
        Fp2 N = T.Y - yq * T.Z, D = T.X - xq * T.Z;
        f = f.mul_line(N * xq - D * yq, fp2_scale(N, xp).neg(), fp2_scale(D, yp));
        Fp2 D2 = D.sq(), D3 = D2 * D, xqZ = xq * T.Z;
        Fp2 A = N.sq() * T.Z - D2 * (T.X + xqZ);
        Fp2 Y3 = N * (xqZ * D2 - A) - yq * T.Z * D3;
        T.X = A * D;
        T.Y = Y3;
        T.Z = T.Z * D3;
    }
};
// prod_k f_{|x|,Q_k}(P_k), conjugated for x < 0 (equal to the inverse once final-exponentiated); pairs with a point at
// infinity contribute 1
inline Fp12 multi_miller(std::vector<MillerPair>& pairs) {
    Fp12 f = Fp12::one();
    const uint64_t xabs = 0xd201000000010000ull;
    for (int b = 62; b >= 0; --b) {
        if (b != 62) f = f.sq();
        for (auto& pr : pairs) pr.dbl_step(f);
        if ((xabs >> b) & 1)
            for (auto& pr : pairs) pr.add_step(f);
    }
    return f.conj();
}
inline Fp12 miller(const G1A& P, const G2A& Q) {
    std::vector<MillerPair> v;
    if (!P.inf && !Q.inf) v.emplace_back(P, Q);
    return multi_miller(v);
}
// f^x on the cyclotomic subgroup (inverse = conjugate there)
inline Fp12 pow_x(const Fp12& f) {
    const uint64_t xabs = 0xd201000000010000ull;
    Fp12 r = f;
    for (int b = 62; b >= 0; --b) {
        r = r.sq();
        if ((xabs >> b) & 1) r = r * f;
    }
    return r.conj();
}
// f^(3 (p^12 - 1)/r)
inline Fp12 final_exp(const Fp12& f) {
    Fp12 t = f.conj() * f.inv();            // ^(p^6 - 1)
    t = frobenius(frobenius(t)) * t;        // ^(p^2 + 1): t is now in the cyclotomic subgroup
    Fp12 a = pow_x(t) * t.conj();           // ^(x - 1)
    Fp12 b = pow_x(a) * a.conj();           // ^(x - 1)^2
    Fp12 c = pow_x(b) * frobenius(b);       // ^(x + p)
    Fp12 d = pow_x(pow_x(c)) * frobenius(frobenius(c)) * c.conj();  // ^(x^2 + p^2 - 1)
    return d * t.sq() * t;                  // + 3
}

}  // namespace bls
}  // namespace masp_host
