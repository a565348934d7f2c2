// Host-side Groth16 verification for the prover's self-check:
//   `verify_proof(verifying_key, &proof, &public_input)` at /root/reference/masp_proofs/src/sapling/prover.rs:148,266
//   with `PreparedVerifyingKey` built at /root/reference/masp_proofs/src/lib.rs:391-393
// (nam-bellperson / pairing 0.23 / blst, un-vendored).  SURVEY.md §8 row a12: "CPU, ~ms; stays on host".
// BLS12-381 base field in 6 x 64-bit limbs, the Fp2/Fp6/Fp12 tower, zcash point decoding, an ate Miller loop over
// E(Fp12) and the final exponentiation  f^((p^12-1)/r)  split as  (p^6-1)(p^2+1) * (p^4-p^2+1)/r  with the first two
// factors done by conjugation / Frobenius-free inversion and squaring tricks kept deliberately simple.
// Product code (libmasp_host); independent of oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace masp_host {
namespace bls {

typedef unsigned __int128 u128;

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
    static void mm(uint64_t* out, const uint64_t* a, const uint64_t* b) {
        const uint64_t* p = P();
        const uint64_t inv = k().inv;
        uint64_t t[8] = {0};
        for (int i = 0; i < 6; ++i) {
            u128 c = 0;
            for (int j = 0; j < 6; ++j) {
                u128 x = (u128)a[j] * b[i] + t[j] + c;
                t[j] = (uint64_t)x;
                c = x >> 64;
            }
            u128 x = (u128)t[6] + c;
            t[6] = (uint64_t)x;
            t[7] = (uint64_t)(x >> 64);
            uint64_t m = t[0] * inv;
            c = ((u128)m * p[0] + t[0]) >> 64;
            for (int j = 1; j < 6; ++j) {
                u128 y = (u128)m * p[j] + t[j] + c;
                t[j - 1] = (uint64_t)y;
                c = y >> 64;
            }
            x = (u128)t[6] + c;
            t[5] = (uint64_t)x;
            t[6] = t[7] + (uint64_t)(x >> 64);
        }
        if (t[6] || ge(t, p)) subr(t, t, p);
        memcpy(out, t, 48);
    }
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
    Fp inv() const {
        uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
        subr(e, P(), two);
        return pow(e, 6);
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
    Fp6 operator*(const Fp6& o) const {
        return {a * o.a + (b * o.c + c * o.b).xi(), a * o.b + b * o.a + (c * o.c).xi(), a * o.c + b * o.b + c * o.a};
    }
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
    Fp12 sq() const { return *this * *this; }
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
    G1J mul_le(const uint8_t* k32) const {
        G1J r = inf();
        for (int i = 255; i >= 0; --i) {
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
inline bool g1_compressed(G1A& p, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        p = {Fp::zero(), Fp::zero(), true};
        return true;
    }
    uint8_t t[48];
    memcpy(t, in, 48);
    bool big = t[0] & 0x20;
    t[0] &= 0x1f;
    if (!Fp::from_be(p.x, t)) return false;
    if (!(p.x.sq() * p.x + Fp::from_u64(4)).sqrt(p.y)) return false;
    if (p.y.lex_largest() != big) p.y = p.y.neg();
    p.inf = false;
    return true;
}
inline bool g2_compressed(G2A& p, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        p = {Fp2::zero(), Fp2::zero(), true};
        return true;
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
    return true;
}

// Miller loop f_{|x|,Q}(P) with Q untwisted into E(Fp12): (x', y') -> (x'/w^2, y'/w^3); x = -0xd201000000010000.
struct PairingK {
    Fp12 w2i, w3i;
    std::vector<uint64_t> hard;  // (p^4 - p^2 + 1) / r
};
const PairingK& pairing_k();  // host_api.cpp

inline Fp12 miller(const G1A& P, const G2A& Q) {
    if (P.inf || Q.inf) return Fp12::one();
    const PairingK& k = pairing_k();
    Fp12 xq = Fp12::from_fp2(Q.x) * k.w2i, yq = Fp12::from_fp2(Q.y) * k.w3i;
    Fp12 xp = Fp12::from_fp(P.x), yp = Fp12::from_fp(P.y);
    Fp12 xt = xq, yt = yq, f = Fp12::one();
    const uint64_t xabs = 0xd201000000010000ull;
    for (int b = 62; b >= 0; --b) {
        f = f.sq();
        Fp12 x2 = xt.sq();
        Fp12 lam = (x2 + x2 + x2) * (yt + yt).inv();
        f = f * ((yp - yt) - lam * (xp - xt));
        Fp12 x3 = lam.sq() - xt - xt;
        yt = lam * (xt - x3) - yt;
        xt = x3;
        if ((xabs >> b) & 1) {
            Fp12 l2 = (yq - yt) * (xq - xt).inv();
            f = f * ((yp - yt) - l2 * (xp - xt));
            Fp12 x4 = l2.sq() - xt - xq;
            yt = l2 * (xt - x4) - yt;
            xt = x4;
        }
    }
    return f.conj();  // x < 0: f^-1 up to factors killed by the final exponentiation (conj = inverse on the cyclotomic subgroup)
}
// f^((p^12 - 1)/r):  easy part f^(p^6 - 1) = conj(f)/f, then ^(p^2 + 1) by a plain power, then the hard exponent
inline Fp12 final_exp(const Fp12& f) {
    const PairingK& k = pairing_k();
    Fp12 t = f.conj() * f.inv();  // f^(p^6 - 1)
    // t^(p^2 + 1): p^2 + 1 as an exponent (762 bits)
    static const std::vector<uint64_t> p2p1 = [] {
        // (p * p + 1) in 64-bit limbs by schoolbook on the modulus
        const uint64_t* p = Fp::P();
        std::vector<uint64_t> r(12, 0);
        for (int i = 0; i < 6; ++i) {
            u128 c = 0;
            for (int j = 0; j < 6; ++j) {
                u128 x = (u128)p[i] * p[j] + r[i + j] + c;
                r[i + j] = (uint64_t)x;
                c = x >> 64;
            }
            r[i + 6] += (uint64_t)c;
        }
        for (int i = 0; i < 12; ++i)
            if (++r[i]) break;
        return r;
    }();
    t = t.pow(p2p1);
    return t.pow(k.hard);
}

}  // namespace bls
}  // namespace masp_host
