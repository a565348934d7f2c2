// Host-side BLS12-381 scalar field (= the base field of Jubjub) for witness generation: 4 x 64-bit limbs,
// Montgomery form.  This is what `bls12_381::Scalar` is to the reference's circuits
// (/root/reference/masp_proofs/src/circuit/*.rs compute every witness value with it).
// Product code: independent of oracle/ (which has its own implementation used only as a checker).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

#include "mont.h"

namespace masp_host {

struct Fr {
    uint64_t l[4];

    static const uint64_t* modulus() {
        static const uint64_t m[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
        return m;
    }
    struct Consts {
        uint64_t r1[4], r2[4], inv;
    };
    static const Consts& k() {
        static Consts c = [] {
            Consts x;
            const uint64_t* p = modulus();
            uint64_t v = 1;
            for (int i = 0; i < 7; ++i) v *= 2 - p[0] * v;
            x.inv = (uint64_t)0 - v;
            uint64_t t[4] = {1, 0, 0, 0};
            for (int s = 0; s < 512; ++s) {
                uint64_t carry = 0;
                for (int i = 0; i < 4; ++i) {
                    uint64_t n = (t[i] << 1) | carry;
                    carry = t[i] >> 63;
                    t[i] = n;
                }
                if (carry || ge(t, p)) sub_raw(t, t, p);
                if (s == 255) memcpy(x.r1, t, 32);
            }
            memcpy(x.r2, t, 32);
            return x;
        }();
        return c;
    }
    static bool ge(const uint64_t* a, const uint64_t* b) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > b[i]) return true;
            if (a[i] < b[i]) return false;
        }
        return true;
    }
    static uint64_t add_raw(uint64_t* r, const uint64_t* a, const uint64_t* b) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)a[i] + b[i];
            r[i] = (uint64_t)c;
            c >>= 64;
        }
        return (uint64_t)c;
    }
    static uint64_t sub_raw(uint64_t* r, const uint64_t* a, const uint64_t* b) {
        uint64_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)a[i] - b[i] - borrow;
            r[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        return borrow;
    }
    static void mont_mul(uint64_t* out, const uint64_t* a, const uint64_t* b) { mont_mul_n<4>(out, a, b, modulus(), k().inv); }

    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    // 2^256 mod r (the Montgomery form of 1) as a constant: witness synthesis asks for it ~10^5 times per proof; k().r1 is
    // the same value computed from the modulus (compared once in the circuit set-up, host_api.cpp)
    static Fr one() { return Fr{{0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full}}; }
    static bool one_constant_ok() { return memcmp(one().l, k().r1, 32) == 0; }
    static Fr from_u64(uint64_t x) {
        uint64_t v[4] = {x, 0, 0, 0};
        Fr r;
        mont_mul(r.l, v, k().r2);
        return r;
    }
    // 4 little-endian 64-bit limbs, canonical value (< r); as `Scalar::from_u64s_le` in the reference constants
    static Fr from_limbs(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
        uint64_t v[4] = {a, b, c, d};
        Fr r;
        mont_mul(r.l, v, k().r2);
        return r;
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fr& o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
    Fr operator+(const Fr& o) const {
        Fr r;
        uint64_t c = add_raw(r.l, l, o.l);
        if (c || ge(r.l, modulus())) sub_raw(r.l, r.l, modulus());
        return r;
    }
    Fr operator-(const Fr& o) const {
        Fr r;
        if (sub_raw(r.l, l, o.l)) add_raw(r.l, r.l, modulus());
        return r;
    }
    Fr neg() const { return zero() - *this; }
    Fr dbl() const { return *this + *this; }
    Fr operator*(const Fr& o) const {
        Fr r;
        mont_mul(r.l, l, o.l);
        return r;
    }
    Fr square() const { return *this * *this; }
    Fr pow(const uint64_t* e, int n) const {
        Fr r = one();
        for (int i = n - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                r = r.square();
                if ((e[i] >> b) & 1) r = r * *this;
            }
        return r;
    }
    // returns false for zero (no inverse).  Batched-divstep inversion on the raw limbs (mont.h).
    bool invert(Fr& out) const {
        static const ModInv<4> inv(modulus());
        uint64_t r[4];  // (a R)^-1 mod p from the Montgomery representative a R
        if (!inv.invert(r, l)) return false;
        // to Montgomery form of a^-1:  (aR)^-1 * R^2 = a^-1 R : two Montgomery products by R^2
        uint64_t t[4];
        mont_mul(t, r, k().r2);
        mont_mul(out.l, t, k().r2);
        return true;
    }
    void to_canonical(uint64_t* v) const {
        uint64_t one_[4] = {1, 0, 0, 0};
        mont_mul(v, l, one_);
    }
    void to_bytes(uint8_t* out) const {  // 32 bytes little-endian canonical (`to_repr`)
        uint64_t v[4];
        to_canonical(v);
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(v[i] >> (8 * b));
    }
    static bool from_bytes(Fr& out, const uint8_t* in) {  // rejects non-canonical (`from_repr`)
        uint64_t v[4];
        for (int i = 0; i < 4; ++i) {
            uint64_t x = 0;
            for (int b = 7; b >= 0; --b) x = (x << 8) | in[8 * i + b];
            v[i] = x;
        }
        if (ge(v, modulus())) return false;
        mont_mul(out.l, v, k().r2);
        return true;
    }
    bool is_odd() const {
        uint64_t v[4];
        to_canonical(v);
        return v[0] & 1;
    }
    bool bit(int i) const {  // bit i of the canonical value
        uint64_t v[4];
        to_canonical(v);
        return (v[i / 64] >> (i % 64)) & 1;
    }
    // square root by Tonelli-Shanks (2-adicity 32, non-residue 7); false if not a square
    bool sqrt(Fr& out) const {
        if (is_zero()) {
            out = *this;
            return true;
        }
        // r - 1 = 2^32 * t
        uint64_t pm1[4], onev[4] = {1, 0, 0, 0};
        sub_raw(pm1, modulus(), onev);
        uint64_t t[4];
        for (int i = 0; i < 4; ++i) t[i] = (pm1[i] >> 32) | (i < 3 ? pm1[i + 1] << 32 : 0);
        uint64_t tp1h[4];  // (t + 1) / 2
        add_raw(tp1h, t, onev);
        for (int i = 0; i < 4; ++i) tp1h[i] = (tp1h[i] >> 1) | (i < 3 ? tp1h[i + 1] << 63 : 0);
        Fr c = from_u64(7).pow(t, 4);  // generator of the 2^32 subgroup
        Fr x = pow(tp1h, 4);
        Fr b = pow(t, 4);
        int m = 32;
        while (b != one()) {
            int i = 0;
            Fr b2 = b;
            while (b2 != one()) {
                b2 = b2.square();
                ++i;
                if (i == m) return false;
            }
            Fr g = c;
            for (int j = 0; j < m - i - 1; ++j) g = g.square();
            x = x * g;
            c = g.square();
            b = b * c;
            m = i;
        }
        if (x.square() != *this) return false;
        out = x;
        return true;
    }
};

}  // namespace masp_host
