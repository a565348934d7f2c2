// EXPERIMENT, not part of the product (measured and rejected in round 4: tools/fp28_field_ubench.hip).
// Fp of BLS12-381 on 14 limbs of 28 bits, Montgomery radix R' = 2^392, written as the field the G1 bucket tree's two passes
// would compute in — complete (products, square, carry-free differences, canonicalisation, conversions, an ops policy with the
// interface of FpOps) and checked against python integers on the host and on the device (tests/test_device_math_host.py).
//
// Why a second representation (measured, tools/valu_rate_ubench.hip / tools/carry_ubench.hip, profiles/r04e_*): on gfx950 every
// VOP3 instruction and every instruction that reads or writes a carry costs a wave ~4.5 SIMD cycles — a v_addc_co_u32 as much
// as a v_mad_u64_u32 — and only plain VOP2 add / and / shift / mov cost 2.45.  The 12 x 32-bit product (field.hpp) is 300
// multiply-adds + 219 carry words + a 36-instruction conditional subtraction: 555 slow instructions.  With 28-bit limbs a
// column of 28 limb products (< 2^56 each) cannot overflow the 64-bit accumulator: 406 multiply-adds + one v_alignbit per
// column, no carry word (433 slow instructions), the square 342 instead of 471; additions and subtractions need no carry
// chain at all — limbs have four spare bits, so they are 14 plain 32-bit adds, left unnormalised ("lazy") until a value is
// multiplied (the product takes limbs up to 2^30) or stored (fp28_canon).
//
// Forms of a value v = sum v[i] 2^(28 i):
//   canonical   limbs < 2^28, v < p                      what is stored (table rows, the tree's planes): zero tests and
//                                                        comparisons are limb-wise
//   N < 2p      limbs < 2^28, v < 2p                     what a product returns
//   lazy        limbs < 2^31, v < 8p                     sums / differences; fp28_canon brings them back
// Montgomery residues are x R' mod p.  A 12 x 32-bit residue s = x 2^384 (field.hpp) becomes one by reading it at an
// 8-bit offset: the integer s 2^8 < 2^392 is congruent to x R' (fp28_from_fp_lazy: no arithmetic); the way back costs a product
// by 2^384 (fp28_to_fp).
// Replaces nothing the reference has by name: it is the arithmetic under bellperson's multiexp (call
// sites /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252).
#pragma once
#include "../masp_amd/csrc/device/field.hpp"

namespace masp {

struct alignas(8) F28 {
    uint32_t v[14];
};

struct Fp28C {
    static constexpr uint32_t MASK = 0x0fffffffu, INV = 0xffcfffdu;  // INV = -p^-1 mod 2^28
    static constexpr uint32_t P[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                       0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x1a011u};
    static constexpr uint32_t ONE[14] = {0x347fcb8u, 0xd800000u, 0x2b119u,   0xcde6d2u,  0xc7212e0u, 0x83a2090u, 0x37669fu,
                                         0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x12ca7cu,  0x659fcfau, 0x577au};  // 2^392 mod p
    static constexpr uint32_t C384[14] = {0x2fffdu,   0x900000u,  0xc000276u, 0xbc40u,    0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                          0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x15f65u};  // 2^384 mod p
    // 2p with every limb but the top one lifted by 2^28 (borrowed from the limb above): K2[i] >= any canonical limb i, so
    // a[i] + K2[i] - b[i] never goes negative for a canonical b
    static constexpr uint32_t K2[14] = {0x1fff5556u, 0x1fdffffeu, 0x17ffff72u, 0x1fffd629u, 0x1c483d56u, 0x141ed61du, 0x1ece61a4u,
                                        0x1e70a256u, 0x18ee9708u, 0x19759aebu, 0x174f6c85u, 0x1cd34962u, 0x13d472feu, 0x34021u};
    static constexpr uint32_t TOP_D = 0x1a012u;      // P[13] + 1
    static constexpr uint32_t TOP_M = 40323u;        // floor(2^32 / TOP_D)
};

// ---- multiply-adds into a 64-bit column accumulator (no carry word) ------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define MASP_M28 "v_mad_u64_u32 %0, vcc, "
MASP_HD void m28_vv(uint64_t& acc, uint32_t a, uint32_t b) { asm(MASP_M28 "%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
MASP_HD void m28_vv2(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm(MASP_M28 "%1, %2, %0\n\t" MASP_M28 "%3, %4, %0" : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
MASP_HD void m28_vv4(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    asm(MASP_M28 "%1, %2, %0\n\t" MASP_M28 "%3, %4, %0\n\t" MASP_M28 "%5, %6, %0\n\t" MASP_M28 "%7, %8, %0"
        : "+v"(acc)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
        : "vcc");
}
MASP_HD void m28_vs(uint64_t& acc, uint32_t a, uint32_t k) { asm(MASP_M28 "%1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }
MASP_HD void m28_vs2(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) {
    asm(MASP_M28 "%1, %2, %0\n\t" MASP_M28 "%3, %4, %0" : "+v"(acc) : "v"(a0), "s"(k0), "v"(a1), "s"(k1) : "vcc");
}
MASP_HD void m28_vs4(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2, uint32_t a3, uint32_t k3) {
    asm(MASP_M28 "%1, %2, %0\n\t" MASP_M28 "%3, %4, %0\n\t" MASP_M28 "%5, %6, %0\n\t" MASP_M28 "%7, %8, %0"
        : "+v"(acc)
        : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3)
        : "vcc");
}
// the quotient digit: lo(acc) * INV mod 2^28 (a multiply-add: v_mul_lo_u32 costs three of them)
MASP_HD uint32_t m28_digit(uint64_t acc) {
    uint64_t t;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"((uint32_t)acc), "s"(Fp28C::INV) : "vcc");
    return (uint32_t)t & Fp28C::MASK;
}
// acc >>= 28 as v_alignbit_b32 + a 32-bit shift (v_lshrrev_b64 costs three multiply-adds)
MASP_HD void m28_shift(uint64_t& acc) {
    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    uint32_t nlo, nhi;  // (both in asm: left to the compiler, hi >> 28 becomes the high word of a v_lshrrev_b64 of the pair)
    asm("v_alignbit_b32 %0, %2, %3, 28\n\tv_lshrrev_b32 %1, 28, %2" : "=&v"(nlo), "=v"(nhi) : "v"(hi), "v"(lo));
    acc = ((uint64_t)nhi << 32) | nlo;
}
// q k mod 2^32 for a small q and a constant k (a multiply-add: v_mul_lo_u32 costs three)
MASP_HD uint32_t m28_mul_lo(uint32_t q, uint32_t k) {
    uint64_t t;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(q), "s"(k) : "vcc");
    return (uint32_t)t;
}
#else
MASP_HD void m28_vv(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
MASP_HD void m28_vv2(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) { acc += (uint64_t)a0 * b0 + (uint64_t)a1 * b1; }
MASP_HD void m28_vv4(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    acc += (uint64_t)a0 * b0 + (uint64_t)a1 * b1 + (uint64_t)a2 * b2 + (uint64_t)a3 * b3;
}
MASP_HD void m28_vs(uint64_t& acc, uint32_t a, uint32_t k) { acc += (uint64_t)a * k; }
MASP_HD void m28_vs2(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) { acc += (uint64_t)a0 * k0 + (uint64_t)a1 * k1; }
MASP_HD void m28_vs4(uint64_t& acc, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2, uint32_t k2, uint32_t a3, uint32_t k3) {
    acc += (uint64_t)a0 * k0 + (uint64_t)a1 * k1 + (uint64_t)a2 * k2 + (uint64_t)a3 * k3;
}
MASP_HD uint32_t m28_digit(uint64_t acc) { return ((uint32_t)acc * Fp28C::INV) & Fp28C::MASK; }
MASP_HD void m28_shift(uint64_t& acc) { acc >>= 28; }
MASP_HD uint32_t m28_mul_lo(uint32_t q, uint32_t k) { return q * k; }
#endif

// sum_{i = I}^{END - 1} a[i] b[K - i]
template <int K, int I, int END>
MASP_HD void m28_col_vv(uint64_t& acc, const uint32_t* a, const uint32_t* b) {
    if constexpr (END - I >= 4) {
        m28_vv4(acc, a[I], b[K - I], a[I + 1], b[K - I - 1], a[I + 2], b[K - I - 2], a[I + 3], b[K - I - 3]);
        m28_col_vv<K, I + 4, END>(acc, a, b);
    } else if constexpr (END - I >= 2) {
        m28_vv2(acc, a[I], b[K - I], a[I + 1], b[K - I - 1]);
        m28_col_vv<K, I + 2, END>(acc, a, b);
    } else if constexpr (END - I == 1) {
        m28_vv(acc, a[I], b[K - I]);
    }
}
// sum_{i = I}^{END - 1} m[i] P[K - i]
template <int K, int I, int END>
MASP_HD void m28_col_vs(uint64_t& acc, const uint32_t* m) {
    if constexpr (END - I >= 4) {
        m28_vs4(acc, m[I], Fp28C::P[K - I], m[I + 1], Fp28C::P[K - I - 1], m[I + 2], Fp28C::P[K - I - 2], m[I + 3], Fp28C::P[K - I - 3]);
        m28_col_vs<K, I + 4, END>(acc, m);
    } else if constexpr (END - I >= 2) {
        m28_vs2(acc, m[I], Fp28C::P[K - I], m[I + 1], Fp28C::P[K - I - 1]);
        m28_col_vs<K, I + 2, END>(acc, m);
    } else if constexpr (END - I == 1) {
        m28_vs(acc, m[I], Fp28C::P[K - I]);
    }
}
// the square's operand terms of column K: sum_{i < K - i} (2 a[i]) a[K - i]  (+ a[K/2]^2), i from I
template <int K, int I, int END>  // END = first i with i >= K - i
MASP_HD void m28_col_sq(uint64_t& acc, const uint32_t* a, const uint32_t* a2) {
    if constexpr (END - I >= 4) {
        m28_vv4(acc, a2[I], a[K - I], a2[I + 1], a[K - I - 1], a2[I + 2], a[K - I - 2], a2[I + 3], a[K - I - 3]);
        m28_col_sq<K, I + 4, END>(acc, a, a2);
    } else if constexpr (END - I >= 2) {
        m28_vv2(acc, a2[I], a[K - I], a2[I + 1], a[K - I - 1]);
        m28_col_sq<K, I + 2, END>(acc, a, a2);
    } else if constexpr (END - I == 1) {
        m28_vv(acc, a2[I], a[K - I]);
    }
}

template <int K, bool SQ>
MASP_HD void m28_cols(uint64_t& acc, const uint32_t* a, const uint32_t* b, uint32_t* m, uint32_t* r) {
    if constexpr (K < 14) {
        if constexpr (SQ) {
            m28_col_sq<K, 0, (K + 1) / 2>(acc, a, b);  // b = 2 a
            if constexpr (K % 2 == 0) m28_vv(acc, a[K / 2], a[K / 2]);
        } else {
            m28_col_vv<K, 0, K + 1>(acc, a, b);
        }
        m28_col_vs<K, 0, K>(acc, m);
        m[K] = m28_digit(acc);
        m28_vs(acc, m[K], Fp28C::P[0]);
        m28_shift(acc);
        m28_cols<K + 1, SQ>(acc, a, b, m, r);
    } else if constexpr (K < 27) {
        if constexpr (SQ) {
            m28_col_sq<K, K - 13, (K + 1) / 2>(acc, a, b);
            if constexpr (K % 2 == 0) m28_vv(acc, a[K / 2], a[K / 2]);
        } else {
            m28_col_vv<K, K - 13, 14>(acc, a, b);
        }
        m28_col_vs<K, K - 13, 14>(acc, m);
        r[K - 14] = (uint32_t)acc & Fp28C::MASK;
        m28_shift(acc);
        m28_cols<K + 1, SQ>(acc, a, b, m, r);
    }
}
// a b / R' mod p as N < 2p.  Operand limbs a[i] < 2^A, b[i] < 2^B with A + B <= 60 (14 2^60 + 14 2^56 + carry < 2^64); values
// with a b < p R' (any two lazy values: 8p 8p << 2^392 p)
MASP_HD F28 fp28_mul(const F28& a, const F28& b) {
    uint32_t m[14];
    F28 r;
    uint64_t acc = 0;
    m28_cols<0, false>(acc, a.v, b.v, m, r.v);
    r.v[13] = (uint32_t)acc;
    return r;
}
// a^2 / R' as N < 2p; limbs a[i] < 2^29
MASP_HD F28 fp28_sqr(const F28& a) {
    uint32_t m[14], a2[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) a2[i] = a.v[i] << 1;
    F28 r;
    uint64_t acc = 0;
    m28_cols<0, true>(acc, a.v, a2, m, r.v);
    r.v[13] = (uint32_t)acc;
    return r;
}

// ---- additions without carries ---------------------------------------------------------------------------------------
MASP_HD F28 fp28_zero() {
    F28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) r.v[i] = 0;
    return r;
}
MASP_HD F28 fp28_one() {
    F28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) r.v[i] = Fp28C::ONE[i];
    return r;
}
MASP_HD F28 fp28_add_lazy(const F28& a, const F28& b) {
    F28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + 2p for a CANONICAL b (a: any form whose limbs leave room for 2^29 more)
MASP_HD F28 fp28_sub_lazy(const F28& a, const F28& b) {
    F28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) r.v[i] = a.v[i] + (Fp28C::K2[i] - b.v[i]);
    return r;
}
// lazy (limbs < 2^31, value < 8p) -> canonical
MASP_HD F28 fp28_canon(const F28& a) {
    uint32_t t[14];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const uint32_t s = a.v[i] + c;
        t[i] = s & Fp28C::MASK;
        c = s >> 28;
    }
    t[13] = a.v[13] + c;  // value >> 364 < 8 (P[13] + 1) < 2^20
    // q = floor(t[13] / (P[13] + 1)) exactly: then 0 <= value - q p < p (1 + 9 / P[13])
    uint32_t q = (uint32_t)(((uint64_t)t[13] * Fp28C::TOP_M) >> 32);
    if (t[13] - m28_mul_lo(q, Fp28C::TOP_D) >= Fp28C::TOP_D) ++q;
    F28 r;
    int32_t b = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int32_t s = (int32_t)t[i] - (int32_t)m28_mul_lo(q, Fp28C::P[i]) + b;  // q <= 7: q P[i] < 2^31
        r.v[i] = (uint32_t)s & Fp28C::MASK;
        b = s >> 28;
    }
    // (the last limb is exact: the total is >= 0 and < 2^365, so no borrow leaves it and the mask takes nothing away)
    // now value < p (1 + 9 / P[13]): at or above p only if the top limb has reached P[13] — rare, and exact below
    if (r.v[13] >= Fp28C::P[13]) {
        uint32_t u[14];
        int32_t bb = 0;
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int32_t s = (int32_t)r.v[i] - (int32_t)Fp28C::P[i] + bb;
            u[i] = (uint32_t)s & Fp28C::MASK;
            bb = s >> 28;
        }
        if (bb == 0) {  // value >= p
#pragma unroll
            for (int i = 0; i < 14; ++i) r.v[i] = u[i];
        }
    }
    return r;
}
// canonical in, canonical out
MASP_HD bool fp28_is_zero(const F28& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) o |= a.v[i];
    return o == 0;
}
MASP_HD bool fp28_eq(const F28& a, const F28& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
MASP_HD F28 fp28_neg(const F28& a) {  // p - a, and 0 for 0
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) o |= a.v[i];
    const uint32_t nz = o ? 0xffffffffu : 0u;
    F28 r;
    int32_t b = 0;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int32_t s = (int32_t)Fp28C::P[i] - (int32_t)a.v[i] + b;
        r.v[i] = (uint32_t)s & Fp28C::MASK & nz;
        b = s >> 28;
    }
    return r;
}

// ---- to and from the 12 x 32-bit residues of field.hpp ----------------------------------------------------------------
// the integer s 2^8 in 28-bit limbs: congruent to x R' when s = x 2^384 — a lazy operand (limbs < 2^28, value < 2^392: a
// product with anything below 4p comes out below 2p)
MASP_HD F28 fp28_from_fp_lazy(const Fp& s) {
    F28 r;
    r.v[0] = (s.v[0] << 8) & Fp28C::MASK;
#pragma unroll
    for (int i = 1; i < 14; ++i) {
        const int o = 28 * i - 8, j = o / 32, sh = o % 32;  // limb i = bits [o, o + 28) of s
        const uint32_t lo = s.v[j], hi = j + 1 < 12 ? s.v[j + 1] : 0u;
        const uint32_t w = sh == 0 ? lo : (lo >> sh) | (hi << (32 - sh));
        r.v[i] = w & Fp28C::MASK;
    }
    return r;
}
// canonical x R' from canonical (or any) x 2^384
MASP_HD F28 fp28_from_fp(const Fp& s) {
    F28 one;
#pragma unroll
    for (int i = 0; i < 14; ++i) one.v[i] = Fp28C::ONE[i];
    return fp28_canon(fp28_mul(fp28_from_fp_lazy(s), one));  // s 2^8 R' / R'
}
// canonical x 2^384 in 12 words from x R' (N < 2p or canonical)
MASP_HD Fp fp28_to_fp(const F28& a) {
    F28 c;
#pragma unroll
    for (int i = 0; i < 14; ++i) c.v[i] = Fp28C::C384[i];
    const F28 t = fp28_canon(fp28_mul(a, c));  // a 2^384 / 2^392
    Fp r;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int lo = 32 * j / 28, sh = 32 * j - 28 * lo;  // word j = bits [32 j, 32 j + 32): limbs lo, lo + 1
        uint32_t w = t.v[lo] >> sh;
        if (lo + 1 < 14) w |= t.v[lo + 1] << (28 - sh);
        if (lo + 2 < 14 && 56 - sh < 32) w |= t.v[lo + 2] << (56 - sh);
        r.v[j] = w;
    }
    return r;
}

// ---- the ops policy of the tree's passes (device/msm_tree.hpp): elements are F28, what crosses into the rest of the MSM
// (lane totals for the shared inversion, the last level's points) is Fp --------------------------------------------------
struct Fp28Ops {
    typedef F28 T;
    typedef Fp Ext;
    typedef Fp28Ops Base;
    static constexpr uint32_t LANES = 1;
    static constexpr bool REPLICATED = false;
    static constexpr bool FP28 = true;
    static MASP_HD T zero() { return fp28_zero(); }
    static MASP_HD T one() { return fp28_one(); }
    // canonical in, canonical out (the rare paths: exceptional pairs)
    static MASP_HD T add(const T& a, const T& b) { return fp28_canon(fp28_add_lazy(a, b)); }
    static MASP_HD T sub(const T& a, const T& b) { return fp28_canon(fp28_sub_lazy(a, b)); }
    static MASP_HD T neg(const T& a) { return fp28_neg(a); }
    static MASP_HD T dbl(const T& a) { return fp28_canon(fp28_add_lazy(a, a)); }
    // products: lazy operands in, N < 2p out
    static MASP_HD T mul(const T& a, const T& b) { return fp28_mul(a, b); }
    static MASP_HD T mul_lazy(const T& a, const T& b) { return fp28_mul(a, b); }
    static MASP_HD T sqr(const T& a) { return fp28_sqr(a); }
    // the hot path: differences left lazy (the subtrahend canonical), canon() before a value is stored or compared
    static MASP_HD T sub_lazy(const T& a, const T& b) { return fp28_sub_lazy(a, b); }
    static MASP_HD T canon(const T& a) { return fp28_canon(a); }
    static MASP_HD bool is_zero(const T& a) { return fp28_is_zero(a); }
    static MASP_HD bool eq(const T& a, const T& b) { return fp28_eq(a, b); }
    static MASP_HD T from_ext(const Ext& s) { return fp28_from_fp_lazy(s); }
    static MASP_HD Ext to_ext(const T& a) { return fp28_to_fp(a); }
};

}  // namespace masp
