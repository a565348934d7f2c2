// ORACLE (test infrastructure, never linked by the product): BLS12-381 extension tower, G1/G2
// groups, zcash point encodings and a deliberately simple pairing.
//
// Restates what the reference obtains from `bls12_381::{G1Affine,G2Affine,Bls12}` =
// `nam-blstrs 0.7.1-nam.0` over `nam-blst 0.3.15-nam.0` (un-vendored; /root/reference/Cargo.lock:1385-1411)
// and `pairing 0.23.0` (Cargo.lock:1625).  Encodings: SURVEY.md Appendix A.5; curve equations A.4.
// The pairing is the textbook Miller loop over E(Fp12) with affine lines and a plain
// (p^12-1)/r exponentiation: slow, but with nothing to get subtly wrong.  It is used only to
// check the Groth16 verification equation (SURVEY.md §8c oracle 2), which is convention-free.
#pragma once
#include <vector>

#include "field.hpp"

namespace oracle {

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
struct Fp2 {
    Fp c0, c1;
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp2& o) const { return !(*this == o); }
    Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2 operator*(const Fp2& o) const {
        Fp a = c0 * o.c0, b = c1 * o.c1;
        Fp c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    Fp2 sqr() const { return *this * *this; }
    Fp2 mul_fp(const Fp& k) const { return {c0 * k, c1 * k}; }
    // multiply by xi = 1 + u
    Fp2 mul_xi() const { return {c0 - c1, c0 + c1}; }
    Fp2 inv() const {
        Fp n = (c0.sqr() + c1.sqr()).inv();
        return {c0 * n, (c1 * n).neg()};
    }
};

// ---------------------------------------------------------------- Fp6 = Fp2[v]/(v^3 - xi)
struct Fp6 {
    Fp2 c0, c1, c2;
    static Fp6 zero() { return {Fp2::zero(), Fp2::zero(), Fp2::zero()}; }
    static Fp6 one() { return {Fp2::one(), Fp2::zero(), Fp2::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    bool operator==(const Fp6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fp6 operator+(const Fp6& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fp6 operator-(const Fp6& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fp6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    Fp6 operator*(const Fp6& o) const {
        Fp2 r0 = c0 * o.c0 + (c1 * o.c2 + c2 * o.c1).mul_xi();
        Fp2 r1 = c0 * o.c1 + c1 * o.c0 + (c2 * o.c2).mul_xi();
        Fp2 r2 = c0 * o.c2 + c1 * o.c1 + c2 * o.c0;
        return {r0, r1, r2};
    }
    Fp6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    Fp6 inv() const {
        Fp2 t0 = c0.sqr() - (c1 * c2).mul_xi();
        Fp2 t1 = c2.sqr().mul_xi() - c0 * c1;
        Fp2 t2 = c1.sqr() - c0 * c2;
        Fp2 d = (c0 * t0 + (c2 * t1 + c1 * t2).mul_xi()).inv();
        return {t0 * d, t1 * d, t2 * d};
    }
};

// ---------------------------------------------------------------- Fp12 = Fp6[w]/(w^2 - v)
struct Fp12 {
    Fp6 c0, c1;
    static Fp12 zero() { return {Fp6::zero(), Fp6::zero()}; }
    static Fp12 one() { return {Fp6::one(), Fp6::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp12& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp12& o) const { return !(*this == o); }
    Fp12 operator+(const Fp12& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp12 operator-(const Fp12& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp12 neg() const { return {c0.neg(), c1.neg()}; }
    Fp12 dbl() const { return *this + *this; }
    Fp12 operator*(const Fp12& o) const {
        Fp6 a = c0 * o.c0, b = c1 * o.c1;
        return {a + b.mul_v(), c0 * o.c1 + c1 * o.c0};
    }
    Fp12 sqr() const { return *this * *this; }
    Fp12 inv() const {
        Fp6 d = (c0 * c0 - (c1 * c1).mul_v()).inv();
        return {c0 * d, (c1 * d).neg()};
    }
    static Fp12 from_fp(const Fp& x) {
        Fp12 r = zero();
        r.c0.c0.c0 = x;
        return r;
    }
    static Fp12 from_fp2(const Fp2& x) {
        Fp12 r = zero();
        r.c0.c0 = x;
        return r;
    }
    Fp12 pow(const std::vector<uint64_t>& e) const {
        Fp12 r = one();
        bool started = false;
        for (int i = (int)e.size() - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                if (started) r = r.sqr();
                if ((e[i] >> b) & 1) {
                    r = r * *this;
                    started = true;
                }
            }
        return r;
    }
};

// ---------------------------------------------------------------- short Weierstrass y^2 = x^3 + b, a = 0
template <class F>
struct Affine {
    F x, y;
    bool inf;
    static Affine infinity() { return {F::zero(), F::zero(), true}; }
    Affine neg() const { return {x, y.neg(), inf}; }
    bool operator==(const Affine& o) const {
        if (inf || o.inf) return inf == o.inf;
        return x == o.x && y == o.y;
    }
};

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac infinity() { return {F::one(), F::one(), F::zero()}; }
    static Jac from_affine(const Affine<F>& a) {
        if (a.inf) return infinity();
        return {a.x, a.y, F::one()};
    }
    bool is_inf() const { return Z.is_zero(); }
    Jac neg() const { return {X, Y.neg(), Z}; }
    // dbl-2009-l
    Jac dbl() const {
        if (is_inf()) return *this;
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = ((X + B).sqr() - A - C).dbl();
        F E = A.dbl() + A;
        F Fq = E.sqr();
        F X3 = Fq - D.dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        F Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    // add-2007-bl with the exceptional cases handled
    Jac add(const Jac& o) const {
        if (is_inf()) return o;
        if (o.is_inf()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return infinity();
        }
        F H = U2 - U1;
        F I = H.dbl().sqr();
        F J = H * I;
        F r = (S2 - S1).dbl();
        F V = U1 * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (S1 * J).dbl();
        F Z3 = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
        return {X3, Y3, Z3};
    }
    // madd-2007-bl (Z2 = 1) with the exceptional cases handled
    Jac add_affine(const Affine<F>& a) const {
        if (a.inf) return *this;
        if (is_inf()) return from_affine(a);
        F Z1Z1 = Z.sqr();
        F U2 = a.x * Z1Z1, S2 = a.y * Z * Z1Z1;
        if (U2 == X) {
            if (S2 == Y) return dbl();
            return infinity();
        }
        F H = U2 - X;
        F HH = H.sqr();
        F I = HH.dbl().dbl();
        F J = H * I;
        F r = (S2 - Y).dbl();
        F V = X * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (Y * J).dbl();
        F Z3 = (Z + H).sqr() - Z1Z1 - HH;
        return {X3, Y3, Z3};
    }
    // scalar as little-endian 64-bit limbs
    Jac mul(const uint64_t* e, int nlimbs) const {
        Jac r = infinity();
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                r = r.dbl();
                if ((e[i] >> b) & 1) r = r.add(*this);
            }
        return r;
    }
    Jac mul_fr(const Fr& k) const {
        uint64_t e[4];
        k.to_canonical(e);
        return mul(e, 4);
    }
    Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::infinity();
        F zi = Z.inv();
        F zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
};

typedef Affine<Fp> G1Affine;
typedef Affine<Fp2> G2Affine;
typedef Jac<Fp> G1;
typedef Jac<Fp2> G2;

// Batch conversion with one shared inversion (Montgomery's trick).
template <class F>
static void batch_to_affine(const std::vector<Jac<F>>& in, std::vector<Affine<F>>& out) {
    size_t n = in.size();
    out.resize(n);
    std::vector<F> pre(n);
    F acc = F::one();
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!in[i].is_inf()) acc = acc * in[i].Z;
    }
    F inv = acc.inv();
    for (size_t i = n; i-- > 0;) {
        if (in[i].is_inf()) {
            out[i] = Affine<F>::infinity();
            continue;
        }
        F zi = inv * pre[i];
        inv = inv * in[i].Z;
        F zi2 = zi.sqr();
        out[i] = {in[i].X * zi2, in[i].Y * zi2 * zi, false};
    }
}

// ---- generators (standard; their compressed encodings are KAT-checked in tests)
static inline Fp fp_from_hex(const char* hex) {
    uint8_t be[48];
    memset(be, 0, 48);
    size_t n = strlen(hex);
    for (size_t i = 0; i < n; ++i) {
        char ch = hex[n - 1 - i];
        int v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
        size_t byte = i / 2;
        be[47 - byte] |= (uint8_t)(v << (4 * (i & 1)));
    }
    Fp r;
    Fp::from_bytes_be(r, be);
    return r;
}
static inline G1Affine g1_generator() {
    return {fp_from_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
            fp_from_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"),
            false};
}
static inline G2Affine g2_generator() {
    Fp2 x = {fp_from_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
             fp_from_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")};
    Fp2 y = {fp_from_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
             fp_from_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")};
    return {x, y, false};
}
static inline Fp fp_b_g1() { return Fp::from_u64(4); }
static inline Fp2 fp2_b_g2() { return Fp2{Fp::from_u64(4), Fp::from_u64(4)}; }
static inline bool on_curve(const G1Affine& a) { return a.inf || a.y.sqr() == a.x.sqr() * a.x + fp_b_g1(); }
static inline bool on_curve(const G2Affine& a) { return a.inf || a.y.sqr() == a.x.sqr() * a.x + fp2_b_g2(); }

// ---- zcash encodings ----------------------------------------------------------
static inline bool fp_lex_largest(const Fp& y) {
    // y > (p-1)/2  <=>  y > -y as canonical integers (y != 0)
    uint64_t a[6], b[6];
    y.to_canonical(a);
    y.neg().to_canonical(b);
    return cmp_limbs<6>(a, b) > 0;
}
static inline bool fp2_lex_largest(const Fp2& y) {
    if (!y.c1.is_zero()) return fp_lex_largest(y.c1);
    return fp_lex_largest(y.c0);
}
static inline void g1_write_uncompressed(const G1Affine& a, uint8_t* out) {
    memset(out, 0, 96);
    if (a.inf) {
        out[0] = 0x40;
        return;
    }
    a.x.to_bytes_be(out);
    a.y.to_bytes_be(out + 48);
}
static inline void g2_write_uncompressed(const G2Affine& a, uint8_t* out) {
    memset(out, 0, 192);
    if (a.inf) {
        out[0] = 0x40;
        return;
    }
    a.x.c1.to_bytes_be(out);
    a.x.c0.to_bytes_be(out + 48);
    a.y.c1.to_bytes_be(out + 96);
    a.y.c0.to_bytes_be(out + 144);
}
static inline void g1_write_compressed(const G1Affine& a, uint8_t* out) {
    memset(out, 0, 48);
    if (a.inf) {
        out[0] = 0xc0;
        return;
    }
    a.x.to_bytes_be(out);
    out[0] |= 0x80;
    if (fp_lex_largest(a.y)) out[0] |= 0x20;
}
static inline void g2_write_compressed(const G2Affine& a, uint8_t* out) {
    memset(out, 0, 96);
    if (a.inf) {
        out[0] = 0xc0;
        return;
    }
    a.x.c1.to_bytes_be(out);
    a.x.c0.to_bytes_be(out + 48);
    out[0] |= 0x80;
    if (fp2_lex_largest(a.y)) out[0] |= 0x20;
}
// returns false on malformed input (unchecked decode: no subgroup / curve check, as
// `Parameters::read(_, false)` does, /root/reference/masp_proofs/src/lib.rs:336-341)
static inline bool g1_read_uncompressed(G1Affine& a, const uint8_t* in) {
    if (in[0] & 0x80) return false;
    if (in[0] & 0x40) {
        a = G1Affine::infinity();
        return true;
    }
    a.inf = false;
    return Fp::from_bytes_be(a.x, in) && Fp::from_bytes_be(a.y, in + 48);
}
static inline bool g2_read_uncompressed(G2Affine& a, const uint8_t* in) {
    if (in[0] & 0x80) return false;
    if (in[0] & 0x40) {
        a = G2Affine::infinity();
        return true;
    }
    a.inf = false;
    return Fp::from_bytes_be(a.x.c1, in) && Fp::from_bytes_be(a.x.c0, in + 48) &&
           Fp::from_bytes_be(a.y.c1, in + 96) && Fp::from_bytes_be(a.y.c0, in + 144);
}

// sqrt in Fp (p = 3 mod 4): a^((p+1)/4); returns false if a is not a square
static inline bool fp_sqrt(Fp& out, const Fp& a) {
    uint64_t e[6];
    uint64_t one_[6] = {1, 0, 0, 0, 0, 0};
    add_limbs<6>(e, Fp::ctx().p, one_);  // p+1 (no overflow: p < 2^381)
    for (int i = 0; i < 6; ++i) {
        e[i] >>= 2;
        if (i < 5) e[i] |= e[i + 1] << 62;
    }
    Fp r = a.pow(e, 6);
    if (r.sqr() != a) return false;
    out = r;
    return true;
}
static inline bool fp2_sqrt(Fp2& out, const Fp2& a) {
    if (a.is_zero()) {
        out = a;
        return true;
    }
    if (a.c1.is_zero()) {
        Fp s;
        if (fp_sqrt(s, a.c0)) {
            out = {s, Fp::zero()};
            return true;
        }
        if (fp_sqrt(s, a.c0.neg())) {
            out = {Fp::zero(), s};
            return true;
        }
        return false;
    }
    Fp alpha;
    if (!fp_sqrt(alpha, a.c0.sqr() + a.c1.sqr())) return false;
    Fp half = Fp::from_u64(2).inv();
    Fp delta = (a.c0 + alpha) * half;
    Fp x0;
    if (!fp_sqrt(x0, delta)) {
        delta = (a.c0 - alpha) * half;
        if (!fp_sqrt(x0, delta)) return false;
    }
    Fp x1 = a.c1 * (x0.dbl()).inv();
    out = {x0, x1};
    return out.sqr() == a;
}
static inline bool g1_read_compressed(G1Affine& a, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        a = G1Affine::infinity();
        return true;
    }
    uint8_t tmp[48];
    memcpy(tmp, in, 48);
    bool big = tmp[0] & 0x20;
    tmp[0] &= 0x1f;
    if (!Fp::from_bytes_be(a.x, tmp)) return false;
    Fp y;
    if (!fp_sqrt(y, a.x.sqr() * a.x + fp_b_g1())) return false;
    if (fp_lex_largest(y) != big) y = y.neg();
    a.y = y;
    a.inf = false;
    return true;
}
static inline bool g2_read_compressed(G2Affine& a, const uint8_t* in) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        a = G2Affine::infinity();
        return true;
    }
    uint8_t tmp[96];
    memcpy(tmp, in, 96);
    bool big = tmp[0] & 0x20;
    tmp[0] &= 0x1f;
    if (!Fp::from_bytes_be(a.x.c1, tmp) || !Fp::from_bytes_be(a.x.c0, tmp + 48)) return false;
    Fp2 y;
    if (!fp2_sqrt(y, a.x.sqr() * a.x + fp2_b_g2())) return false;
    if (fp2_lex_largest(y) != big) y = y.neg();
    a.y = y;
    a.inf = false;
    return true;
}

// ---- pairing ------------------------------------------------------------------------------------
// Untwist E'(Fp2) -> E(Fp12): (x', y') -> (x'/w^2, y'/w^3)   (w^6 = xi, so y^2 = x^3 + 4 holds).
struct PairingConsts {
    Fp12 w2_inv, w3_inv;
    std::vector<uint64_t> final_exp;  // (p^12 - 1) / r
};
const PairingConsts& pairing_consts();  // defined in groth16_oracle.cpp

static inline Fp12 miller_loop(const G1Affine& P, const G2Affine& Q) {
    if (P.inf || Q.inf) return Fp12::one();
    const PairingConsts& k = pairing_consts();
    Fp12 xq = Fp12::from_fp2(Q.x) * k.w2_inv, yq = Fp12::from_fp2(Q.y) * k.w3_inv;
    Fp12 xp = Fp12::from_fp(P.x), yp = Fp12::from_fp(P.y);
    Fp12 xt = xq, yt = yq;
    Fp12 f = Fp12::one();
    const uint64_t x_abs = 0xd201000000010000ull;  // |x| of BLS12-381
    bool t_inf = false;
    for (int b = 62; b >= 0; --b) {
        f = f.sqr();
        if (!t_inf) {
            // tangent at T
            Fp12 lam = (xt.sqr().dbl() + xt.sqr()) * (yt.dbl()).inv();
            f = f * ((yp - yt) - lam * (xp - xt));
            Fp12 x3 = lam.sqr() - xt.dbl();
            yt = lam * (xt - x3) - yt;
            xt = x3;
        }
        if ((x_abs >> b) & 1) {
            if (t_inf) {
                xt = xq;
                yt = yq;
                t_inf = false;
            } else if (xt == xq) {
                // vertical line (T = -Q) or tangent (T = Q): cannot occur for order-r Q inside the loop
                f = f * (xp - xt);
                t_inf = true;
            } else {
                Fp12 lam = (yq - yt) * (xq - xt).inv();
                f = f * ((yp - yt) - lam * (xp - xt));
                Fp12 x3 = lam.sqr() - xt - xq;
                yt = lam * (xt - x3) - yt;
                xt = x3;
            }
        }
    }
    return f.inv();  // x < 0
}
static inline Fp12 final_exponentiation(const Fp12& f) { return f.pow(pairing_consts().final_exp); }
static inline Fp12 pairing(const G1Affine& P, const G2Affine& Q) { return final_exponentiation(miller_loop(P, Q)); }

}  // namespace oracle
