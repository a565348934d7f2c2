// ORACLE (test infrastructure, never linked by the product): 64-bit-limb Montgomery prime
// fields for BLS12-381.
//
// Restates the field arithmetic the reference reaches through `bls12_381::Scalar` and the
// curve types of `nam-blstrs 0.7.1-nam.0` / `nam-blst 0.3.15-nam.0` (un-vendored; pinned at
// /root/reference/Cargo.lock:1385-1411, used at masp_proofs/Cargo.toml:22).  Moduli and
// encodings: SURVEY.md Appendix A.4/A.5.
//
// All derived Montgomery constants (R, R^2, -p^-1 mod 2^64) are computed at start-up from the
// modulus alone so that no hand-typed table can be wrong.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace oracle {

typedef unsigned __int128 u128;

template <int N>
struct MontCtx {
    uint64_t p[N];
    uint64_t r1[N];  // R mod p
    uint64_t r2[N];  // R^2 mod p
    uint64_t inv;    // -p^-1 mod 2^64
};

template <int N>
static inline int cmp_limbs(const uint64_t* a, const uint64_t* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
template <int N>
static inline uint64_t add_limbs(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    u128 c = 0;
    for (int i = 0; i < N; ++i) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
template <int N>
static inline uint64_t sub_limbs(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}

template <int N>
static void mont_ctx_init(MontCtx<N>& c, const uint64_t* modulus) {
    for (int i = 0; i < N; ++i) c.p[i] = modulus[i];
    // inv = -p^-1 mod 2^64 by Newton iteration
    uint64_t x = 1;
    for (int i = 0; i < 7; ++i) x *= 2 - c.p[0] * x;
    c.inv = (uint64_t)0 - x;
    // R = 2^(64N) mod p: double 1, 64N times.  R2: 64N more doublings.
    uint64_t t[N];
    for (int i = 0; i < N; ++i) t[i] = 0;
    t[0] = 1;
    for (int k = 0; k < 2 * 64 * N; ++k) {
        uint64_t carry = add_limbs<N>(t, t, t);
        if (carry || cmp_limbs<N>(t, c.p) >= 0) sub_limbs<N>(t, t, c.p);
        if (k == 64 * N - 1)
            for (int i = 0; i < N; ++i) c.r1[i] = t[i];
    }
    for (int i = 0; i < N; ++i) c.r2[i] = t[i];
}

// A prime field element in Montgomery form.  `Tag` supplies the (lazily built) context.
template <class Tag>
struct Mont {
    static constexpr int N = Tag::N;
    uint64_t l[N];

    static const MontCtx<N>& ctx() {
        static MontCtx<N> c = [] {
            MontCtx<N> x;
            mont_ctx_init<N>(x, Tag::modulus());
            return x;
        }();
        return c;
    }
    static Mont zero() {
        Mont r;
        for (int i = 0; i < N; ++i) r.l[i] = 0;
        return r;
    }
    static Mont one() {
        Mont r;
        for (int i = 0; i < N; ++i) r.l[i] = ctx().r1[i];
        return r;
    }
    bool is_zero() const {
        uint64_t a = 0;
        for (int i = 0; i < N; ++i) a |= l[i];
        return a == 0;
    }
    bool operator==(const Mont& o) const { return memcmp(l, o.l, sizeof(l)) == 0; }
    bool operator!=(const Mont& o) const { return !(*this == o); }

    Mont operator+(const Mont& o) const {
        Mont r;
        uint64_t carry = add_limbs<N>(r.l, l, o.l);
        if (carry || cmp_limbs<N>(r.l, ctx().p) >= 0) sub_limbs<N>(r.l, r.l, ctx().p);
        return r;
    }
    Mont operator-(const Mont& o) const {
        Mont r;
        if (sub_limbs<N>(r.l, l, o.l)) add_limbs<N>(r.l, r.l, ctx().p);
        return r;
    }
    Mont neg() const {
        if (is_zero()) return *this;
        Mont r;
        sub_limbs<N>(r.l, ctx().p, l);
        return r;
    }
    Mont dbl() const { return *this + *this; }

    // CIOS Montgomery product (a*b*R^-1 mod p), product and reduction interleaved in one pass per limb of b; valid
    // because both moduli leave the top bit of their top limb clear, so the running value never needs an extra limb.
    static void mont_mul(uint64_t* out, const uint64_t* a, const uint64_t* b) {
        const MontCtx<N>& c = ctx();
        const uint64_t* p = c.p;
        const uint64_t inv = c.inv;
        uint64_t t[N];
#pragma GCC unroll 8
        for (int j = 0; j < N; ++j) t[j] = 0;
#pragma GCC unroll 8
        for (int i = 0; i < N; ++i) {
            u128 x = (u128)a[0] * b[i] + t[0];
            uint64_t hi = (uint64_t)(x >> 64);
            const uint64_t m = (uint64_t)x * inv;
            uint64_t red = (uint64_t)(((u128)m * p[0] + (uint64_t)x) >> 64);
#pragma GCC unroll 8
            for (int j = 1; j < N; ++j) {
                x = (u128)a[j] * b[i] + t[j] + hi;
                hi = (uint64_t)(x >> 64);
                const u128 y = (u128)m * p[j] + (uint64_t)x + red;
                t[j - 1] = (uint64_t)y;
                red = (uint64_t)(y >> 64);
            }
            t[N - 1] = red + hi;
        }
        if (cmp_limbs<N>(t, p) >= 0) sub_limbs<N>(t, t, p);
        for (int i = 0; i < N; ++i) out[i] = t[i];
    }
    Mont operator*(const Mont& o) const {
        Mont r;
        mont_mul(r.l, l, o.l);
        return r;
    }
    Mont sqr() const { return *this * *this; }

    // exponent as little-endian 64-bit limbs
    Mont pow(const uint64_t* e, int nlimbs) const {
        Mont r = one();
        bool started = false;
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                if (started) r = r.sqr();
                if ((e[i] >> b) & 1) {
                    r = r * *this;
                    started = true;
                }
            }
        return r;
    }
    Mont pow_u64(uint64_t e) const { return pow(&e, 1); }
    // Fermat inverse; inv(0) = 0.
    Mont inv() const {
        uint64_t e[N];
        uint64_t two[N];
        for (int i = 0; i < N; ++i) two[i] = 0;
        two[0] = 2;
        sub_limbs<N>(e, ctx().p, two);
        return pow(e, N);
    }

    // canonical integer -> Montgomery (reduces values >= p by repeated subtraction of p is NOT
    // done: caller must pass < p unless `reduce` set)
    static Mont from_canonical(const uint64_t* v) {
        Mont r;
        mont_mul(r.l, v, ctx().r2);
        return r;
    }
    void to_canonical(uint64_t* v) const {
        uint64_t one_[N];
        for (int i = 0; i < N; ++i) one_[i] = 0;
        one_[0] = 1;
        mont_mul(v, l, one_);
    }
    static Mont from_u64(uint64_t x) {
        uint64_t v[N];
        for (int i = 0; i < N; ++i) v[i] = 0;
        v[0] = x;
        return from_canonical(v);
    }
    // little-endian canonical bytes (8N). Returns false if >= p.
    static bool from_bytes_le(Mont& out, const uint8_t* b) {
        uint64_t v[N];
        for (int i = 0; i < N; ++i) {
            uint64_t x = 0;
            for (int k = 7; k >= 0; --k) x = (x << 8) | b[8 * i + k];
            v[i] = x;
        }
        if (cmp_limbs<N>(v, ctx().p) >= 0) return false;
        out = from_canonical(v);
        return true;
    }
    void to_bytes_le(uint8_t* b) const {
        uint64_t v[N];
        to_canonical(v);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 8; ++k) b[8 * i + k] = (uint8_t)(v[i] >> (8 * k));
    }
    static bool from_bytes_be(Mont& out, const uint8_t* b) {
        uint8_t tmp[8 * N];
        for (int i = 0; i < 8 * N; ++i) tmp[i] = b[8 * N - 1 - i];
        return from_bytes_le(out, tmp);
    }
    void to_bytes_be(uint8_t* b) const {
        uint8_t tmp[8 * N];
        to_bytes_le(tmp);
        for (int i = 0; i < 8 * N; ++i) b[i] = tmp[8 * N - 1 - i];
    }
    // wide reduction of 64 little-endian bytes (used to derive pseudo-random field elements)
    static Mont from_bytes_wide_le(const uint8_t* b, int nbytes) {
        // Horner in base 256
        Mont acc = zero();
        Mont base = from_u64(256);
        for (int i = nbytes - 1; i >= 0; --i) acc = acc * base + from_u64(b[i]);
        return acc;
    }
};

// ---- BLS12-381 scalar field Fr -------------------------------------------------
struct FrTag {
    static constexpr int N = 4;
    static const uint64_t* modulus() {
        static const uint64_t m[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull,
                                      0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
        return m;
    }
};
// ---- BLS12-381 base field Fp ------------------------------------------------------------------
struct FpTag {
    static constexpr int N = 6;
    static const uint64_t* modulus() {
        static const uint64_t m[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull,
                                      0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull,
                                      0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
        return m;
    }
};
typedef Mont<FrTag> Fr;
typedef Mont<FpTag> Fp;

}  // namespace oracle
