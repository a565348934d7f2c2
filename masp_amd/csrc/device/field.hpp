// BLS12-381 prime-field arithmetic for gfx950: 32-bit limbs, Montgomery form, everything in VGPRs.
//
// Replaces, on the device, what the reference reaches through `bls12_381::Scalar` / blst field
// types (nam-blstrs 0.7.1-nam.0 over nam-blst 0.3.15-nam.0, /root/reference/Cargo.lock:1385-1411;
// SURVEY.md §2b).  Montgomery radix 2^(32N) is blst's own, so limb arrays are bit-compatible with
// `blst_fp` / `blst_fr` memory.
//
// CDNA4 has no 64-bit integer multiplier in the vector ALU: a field product is built from
// 32x32->64 multiply-adds (v_mad_u64_u32).  Functions are __host__ __device__ so that the very
// same source is exercised on the CPU by tests/ (no GPU in the build container).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "consts.hpp"

#define MASP_HD __host__ __device__ __forceinline__
// Out-of-line variant.  NOTE (ROCm 7.2 / LLVM 22, gfx950): a non-kernel device function larger than the
// +-128 KiB SOPP branch range gets its long branches relaxed through `s_getpc_b64 s[30:31]`, which
// clobbers the return address and hangs the wave.  Every out-of-line function below is therefore kept
// far smaller than that (the 384-bit product is ~5 KiB), and everything big is inlined into kernels.
#define MASP_NOINLINE __host__ __device__ __noinline__

namespace masp {

template <class C>
struct Fe {
    uint32_t v[C::N];
};

template <class C>
MASP_HD Fe<C> fe_zero() {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = 0;
    return r;
}
template <class C>
MASP_HD Fe<C> fe_one() {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = C::R1[i];
    return r;
}
template <class C>
MASP_HD bool fe_is_zero(const Fe<C>& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) acc |= a.v[i];
    return acc == 0;
}
template <class C>
MASP_HD bool fe_eq(const Fe<C>& a, const Fe<C>& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
}

// ---- add / sub / neg / dbl / conditional subtraction ------------------------------------------------------------------
// Host forms (tests, tiny set-up computations): portable C++ with 64-bit intermediates.
// Device forms: carry chains written out (v_add_co / v_addc_co / v_sub_co / v_subb_co in their VOP3 encodings, the carry in
// an explicit SGPR pair so that a chain can span several asm statements of four limbs each).  Left to the compiler, the
// portable source became 64-bit additions with the carries rebuilt by shifts and moves: ~140 instructions for a 12-limb
// subtraction, ~180 for an addition, ~100 for the conditional subtraction that ends every product — a quarter of all
// instructions of the bucket-tree's additions pass (round 4; 37 / 48 / 36 now).  gfx9 VALU instructions read at most ONE
// scalar operand, and a carry-in is one: a modulus limb can therefore not be an SGPR or a literal inside a chain — it is moved
// into the destination register first (v_mov with a literal, inside the statement so that no register outlives it).
template <class C>
__host__ inline void fe_reduce_once(Fe<C>& a) {
    uint32_t t[C::N];
    uint64_t borrow = 0;
    for (int i = 0; i < C::N; ++i) {
        uint64_t d = (uint64_t)a.v[i] - C::MOD[i] - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    if (!borrow) {
        for (int i = 0; i < C::N; ++i) a.v[i] = t[i];
    }
}
template <class C>
__host__ inline Fe<C> fe_add(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r;
    uint64_t carry = 0;
    for (int i = 0; i < C::N; ++i) {
        uint64_t s = (uint64_t)a.v[i] + b.v[i] + carry;
        r.v[i] = (uint32_t)s;
        carry = s >> 32;
    }
    // both moduli leave a spare top bit, so a + b < 2p < 2^(32N): no carry out
    fe_reduce_once(r);
    return r;
}
template <class C>
__host__ inline Fe<C> fe_sub(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r;
    uint64_t borrow = 0;
    for (int i = 0; i < C::N; ++i) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    uint64_t carry = 0;
    for (int i = 0; i < C::N; ++i) {
        uint64_t s = (uint64_t)r.v[i] + (C::MOD[i] & mask) + carry;
        r.v[i] = (uint32_t)s;
        carry = s >> 32;
    }
    return r;
}
template <class C>
__host__ inline Fe<C> fe_neg(const Fe<C>& a) {
    return fe_sub(fe_zero<C>(), a);  // 0 - 0 = 0 stays canonical
}
template <class C>
__host__ inline Fe<C> fe_dbl(const Fe<C>& a) {
    return fe_add(a, a);
}

// four limbs of a chain; `c` = the carry / borrow (a wave-wide lane mask in an SGPR pair)
template <bool FIRST>
__device__ __forceinline__ void chain_add4(uint32_t* r, const uint32_t* b, uint64_t& c) {
    if constexpr (FIRST)
        asm("v_add_co_u32_e64 %0, %4, %0, %5\n\tv_addc_co_u32_e64 %1, %4, %1, %6, %4\n\tv_addc_co_u32_e64 %2, %4, %2, %7, %4\n\t"
            "v_addc_co_u32_e64 %3, %4, %3, %8, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "=&s"(c)
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    else
        asm("v_addc_co_u32_e64 %0, %4, %0, %5, %4\n\tv_addc_co_u32_e64 %1, %4, %1, %6, %4\n\tv_addc_co_u32_e64 %2, %4, %2, %7, %4\n\t"
            "v_addc_co_u32_e64 %3, %4, %3, %8, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+s"(c)
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
template <bool FIRST>
__device__ __forceinline__ void chain_sub4(uint32_t* r, const uint32_t* b, uint64_t& c) {  // r -= b
    if constexpr (FIRST)
        asm("v_sub_co_u32_e64 %0, %4, %0, %5\n\tv_subb_co_u32_e64 %1, %4, %1, %6, %4\n\tv_subb_co_u32_e64 %2, %4, %2, %7, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, %3, %8, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "=&s"(c)
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    else
        asm("v_subb_co_u32_e64 %0, %4, %0, %5, %4\n\tv_subb_co_u32_e64 %1, %4, %1, %6, %4\n\tv_subb_co_u32_e64 %2, %4, %2, %7, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, %3, %8, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+s"(c)
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
template <bool FIRST>
__device__ __forceinline__ void chain_neg4(uint32_t* r, uint64_t& c) {  // r = 0 - r
    if constexpr (FIRST)
        asm("v_sub_co_u32_e64 %0, %4, 0, %0\n\tv_subb_co_u32_e64 %1, %4, 0, %1, %4\n\tv_subb_co_u32_e64 %2, %4, 0, %2, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, 0, %3, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "=&s"(c));
    else
        asm("v_subb_co_u32_e64 %0, %4, 0, %0, %4\n\tv_subb_co_u32_e64 %1, %4, 0, %1, %4\n\tv_subb_co_u32_e64 %2, %4, 0, %2, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, 0, %3, %4"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+s"(c));
}
// t = s - (limbs K .. K+3 of the modulus)
template <class C, int K>
__device__ __forceinline__ void chain_submod4(uint32_t* t, const uint32_t* s, uint64_t& c) {
    if constexpr (K == 0)
        asm("v_mov_b32_e32 %0, %9\n\tv_sub_co_u32_e64 %0, %4, %5, %0\n\tv_mov_b32_e32 %1, %10\n\tv_subb_co_u32_e64 %1, %4, %6, %1, %4\n\t"
            "v_mov_b32_e32 %2, %11\n\tv_subb_co_u32_e64 %2, %4, %7, %2, %4\n\tv_mov_b32_e32 %3, %12\n\tv_subb_co_u32_e64 %3, %4, %8, %3, %4"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&s"(c)
            : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "i"(C::MOD[K]), "i"(C::MOD[K + 1]), "i"(C::MOD[K + 2]), "i"(C::MOD[K + 3]));
    else
        asm("v_mov_b32_e32 %0, %9\n\tv_subb_co_u32_e64 %0, %4, %5, %0, %4\n\tv_mov_b32_e32 %1, %10\n\tv_subb_co_u32_e64 %1, %4, %6, %1, %4\n\t"
            "v_mov_b32_e32 %2, %11\n\tv_subb_co_u32_e64 %2, %4, %7, %2, %4\n\tv_mov_b32_e32 %3, %12\n\tv_subb_co_u32_e64 %3, %4, %8, %3, %4"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "+s"(c)
            : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "i"(C::MOD[K]), "i"(C::MOD[K + 1]), "i"(C::MOD[K + 2]), "i"(C::MOD[K + 3]));
}
// t = c ? s : t  (per lane)
__device__ __forceinline__ void chain_sel4(uint32_t* t, const uint32_t* s, uint64_t c) {
    asm("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\tv_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
        : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "s"(c));
}
template <class C, int K>
__device__ __forceinline__ void chain_submod(uint32_t* t, const uint32_t* s, uint64_t& c) {
    if constexpr (K < C::N) {
        chain_submod4<C, K>(t + K, s + K, c);
        chain_submod<C, K + 4>(t, s, c);
    }
}
// a < 2p  ->  a - p if a >= p else a
template <class C>
__device__ __forceinline__ void fe_reduce_once(Fe<C>& a) {
    static_assert(C::N % 4 == 0, "limb chains are written four at a time");
    uint32_t t[C::N];
    uint64_t borrow;
    chain_submod<C, 0>(t, a.v, borrow);
#pragma unroll
    for (int k = 0; k < C::N; k += 4) chain_sel4(t + k, a.v + k, borrow);
#pragma unroll
    for (int i = 0; i < C::N; ++i) a.v[i] = t[i];
}
template <class C>
__device__ __forceinline__ Fe<C> fe_add(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r = a;
    uint64_t carry;
    chain_add4<true>(r.v, b.v, carry);
#pragma unroll
    for (int k = 4; k < C::N; k += 4) chain_add4<false>(r.v + k, b.v + k, carry);
    // both moduli leave a spare top bit, so a + b < 2p < 2^(32N): no carry out
    fe_reduce_once(r);
    return r;
}
// r += p where the lanes of `borrow` are set
template <class C>
__device__ __forceinline__ void fe_add_mod_if(Fe<C>& r, uint64_t borrow) {
    uint32_t mask;
    asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(mask) : "s"(borrow));
    uint32_t t[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) t[i] = C::MOD[i] & mask;
    uint64_t carry;
    chain_add4<true>(r.v, t, carry);
#pragma unroll
    for (int k = 4; k < C::N; k += 4) chain_add4<false>(r.v + k, t + k, carry);
}
template <class C>
__device__ __forceinline__ Fe<C> fe_sub(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r = a;
    uint64_t borrow;
    chain_sub4<true>(r.v, b.v, borrow);
#pragma unroll
    for (int k = 4; k < C::N; k += 4) chain_sub4<false>(r.v + k, b.v + k, borrow);
    fe_add_mod_if(r, borrow);
    return r;
}
template <class C>
__device__ __forceinline__ Fe<C> fe_neg(const Fe<C>& a) {
    Fe<C> r = a;
    uint64_t borrow;  // set iff a != 0: 0 - 0 = 0 stays canonical
    chain_neg4<true>(r.v, borrow);
#pragma unroll
    for (int k = 4; k < C::N; k += 4) chain_neg4<false>(r.v + k, borrow);
    fe_add_mod_if(r, borrow);
    return r;
}
template <class C>
__device__ __forceinline__ Fe<C> fe_dbl(const Fe<C>& a) {
    Fe<C> r;
    r.v[0] = a.v[0] << 1;
#pragma unroll
    for (int i = 1; i < C::N; ++i) r.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);   // 2a < 2p < 2^(32N)
    fe_reduce_once(r);
    return r;
}

// Montgomery product a*b*R^-1 mod p.
//
// Device form: product scanning (column by column) with the reduction folded in ("FIPS"): every
// limb product is ONE v_mad_u64_u32 into a 64-bit column accumulator plus ONE v_addc_co_u32 catching
// its carry in a third word — 2N^2 multiply-adds, N v_mul_lo_u32 for the quotient digits and nothing
// else in the inner loop.  (The plain-C++ CIOS form below costs the compiler two extra moves and a
// 64-bit add per product because v_mad_u64_u32 has no carry-in; measured 1292 vs ~700 instructions for
// N = 12.)  The modulus limbs ride in SGPRs: gfx950 VOP3 takes no 32-bit literals.
// Host form (tests, tiny set-up computations): CIOS in portable C++, selected by host/device overloading.
// (hipcc pads every asm statement with an s_nop, so the multiply-adds of a column go into as few statements
// as possible: groups of 4, 2, 1.)
#define MASP_MAC(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
// the first multiply-add of a column: the carry word starts from the carry alone (VOP3 form, both addends the constant 0), so
// no instruction has to clear it between columns
#define MASP_MAC0(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
template <bool FIRST = false>
__device__ __forceinline__ void mac_vv(uint64_t& acc, uint32_t& c2, uint32_t a, uint32_t b) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") : "+v"(acc), "=&v"(c2) : "v"(a), "v"(b) : "vcc");
    else
        asm(MASP_MAC("%2", "%3") : "+v"(acc), "+v"(c2) : "v"(a), "v"(b) : "vcc");
}
template <bool FIRST = false>
__device__ __forceinline__ void mac_vv2(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") MASP_MAC("%4", "%5") : "+v"(acc), "=&v"(c2) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
    else
        asm(MASP_MAC("%2", "%3") MASP_MAC("%4", "%5") : "+v"(acc), "+v"(c2) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
template <bool FIRST = false>
__device__ __forceinline__ void mac_vv4(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2,
                                        uint32_t b2, uint32_t a3, uint32_t b3) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") MASP_MAC("%4", "%5") MASP_MAC("%6", "%7") MASP_MAC("%8", "%9")
            : "+v"(acc), "=&v"(c2)
            : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
            : "vcc");
    else
        asm(MASP_MAC("%2", "%3") MASP_MAC("%4", "%5") MASP_MAC("%6", "%7") MASP_MAC("%8", "%9")
            : "+v"(acc), "+v"(c2)
            : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
            : "vcc");
}
template <bool FIRST = false>
__device__ __forceinline__ void mac_vs(uint64_t& acc, uint32_t& c2, uint32_t a, uint32_t k) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") : "+v"(acc), "=&v"(c2) : "v"(a), "s"(k) : "vcc");
    else
        asm(MASP_MAC("%2", "%3") : "+v"(acc), "+v"(c2) : "v"(a), "s"(k) : "vcc");
}
template <bool FIRST = false>
__device__ __forceinline__ void mac_vs2(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") MASP_MAC("%4", "%5") : "+v"(acc), "=&v"(c2) : "v"(a0), "s"(k0), "v"(a1), "s"(k1) : "vcc");
    else
        asm(MASP_MAC("%2", "%3") MASP_MAC("%4", "%5") : "+v"(acc), "+v"(c2) : "v"(a0), "s"(k0), "v"(a1), "s"(k1) : "vcc");
}
template <bool FIRST = false>
__device__ __forceinline__ void mac_vs4(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t k0, uint32_t a1, uint32_t k1, uint32_t a2,
                                        uint32_t k2, uint32_t a3, uint32_t k3) {
    if constexpr (FIRST)
        asm(MASP_MAC0("%2", "%3") MASP_MAC("%4", "%5") MASP_MAC("%6", "%7") MASP_MAC("%8", "%9")
            : "+v"(acc), "=&v"(c2)
            : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3)
            : "vcc");
    else
        asm(MASP_MAC("%2", "%3") MASP_MAC("%4", "%5") MASP_MAC("%6", "%7") MASP_MAC("%8", "%9")
            : "+v"(acc), "+v"(c2)
            : "v"(a0), "s"(k0), "v"(a1), "s"(k1), "v"(a2), "s"(k2), "v"(a3), "s"(k3)
            : "vcc");
}
// Multiply-adds that cannot carry out of the 64-bit accumulator, so that no carry word follows them.  A column starts from
// what the previous one carried over (below 2^40); a term x y with x < 2^32 and y < c is below c 2^32; so terms whose bounds c
// sum to less than 2^32 - 2^8 may come first in their column and skip the carry word.  Two kinds of terms have small bounds:
//  * those that hold a TOP limb — every operand of a product is below 2p, its top limb below 2 MOD[N-1] + 2 (0.203 x 2^32):
//    a[K-11] b[11] and a[11] b[K-11] in the columns 11 .. 22 of a 12-limb product;
//  * reduction terms m[i] MOD[j] with a small modulus limb: MOD[11] (0.10), MOD[3] (0.12), MOD[10] (0.22), MOD[8] (0.26),
//    MOD[9] (0.29) x 2^32 ..., taken smallest first while the column's budget lasts (NcTerms::vs_mask, at compile time).
// For a 12-limb product: 69 of 288 carry words less.  Fp only: Fr's top limb has one spare bit, and raw scalars reach a product
// before their range check has been acted on.
#define MASP_MACNC(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\t"
__device__ __forceinline__ void macnc_v(uint64_t& acc, uint32_t a, uint32_t b) { asm(MASP_MACNC("%1", "%2") : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void macnc_vv(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm(MASP_MACNC("%1", "%2") MASP_MACNC("%3", "%4") : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void macnc_s(uint64_t& acc, uint32_t m, uint32_t k) { asm(MASP_MACNC("%1", "%2") : "+v"(acc) : "v"(m), "s"(k) : "vcc"); }
__device__ __forceinline__ void macnc_s2(uint64_t& acc, uint32_t m0, uint32_t k0, uint32_t m1, uint32_t k1) {
    asm(MASP_MACNC("%1", "%2") MASP_MACNC("%3", "%4") : "+v"(acc) : "v"(m0), "s"(k0), "v"(m1), "s"(k1) : "vcc");
}
__device__ __forceinline__ void macnc_s3(uint64_t& acc, uint32_t m0, uint32_t k0, uint32_t m1, uint32_t k1, uint32_t m2, uint32_t k2) {
    asm(MASP_MACNC("%1", "%2") MASP_MACNC("%3", "%4") MASP_MACNC("%5", "%6") : "+v"(acc) : "v"(m0), "s"(k0), "v"(m1), "s"(k1), "v"(m2), "s"(k2) : "vcc");
}
__device__ __forceinline__ void macnc_s4(uint64_t& acc, uint32_t m0, uint32_t k0, uint32_t m1, uint32_t k1, uint32_t m2, uint32_t k2, uint32_t m3,
                                         uint32_t k3) {
    asm(MASP_MACNC("%1", "%2") MASP_MACNC("%3", "%4") MASP_MACNC("%5", "%6") MASP_MACNC("%7", "%8")
        : "+v"(acc)
        : "v"(m0), "s"(k0), "v"(m1), "s"(k1), "v"(m2), "s"(k2), "v"(m3), "s"(k3)
        : "vcc");
}
template <class C>
struct NcTerms {
    static constexpr bool ON = true;
    static constexpr bool TOPS = C::N == 12;   // top-limb terms: Fp only (Fr: one spare bit, and raw scalars reach a product unchecked)
    static constexpr uint64_t TOP = 2ull * C::MOD[C::N - 1] + 2;        // bound of an operand's top limb (operands below 2p)
    static constexpr uint64_t LIMIT = (1ull << 32) - (1ull << 8);       // budget of a column, in units of 2^32
    // the reduction terms m[i] MOD[K - i], i in [I0, I1), that go carry-free after terms worth `base`: bit i of the result
    static constexpr uint32_t vs_mask(int K, int I0, int I1, uint64_t base) {
        uint32_t mask = 0;
        uint64_t sum = base;
        for (;;) {
            int best = -1;
            for (int i = I0; i < I1; ++i)
                if (!((mask >> i) & 1u) && (best < 0 || C::MOD[K - i] < C::MOD[K - best])) best = i;
            if (best < 0 || sum + C::MOD[K - best] + 1 > LIMIT) break;
            sum += C::MOD[K - best] + 1;
            mask |= 1u << best;
        }
        return mask;
    }
    static constexpr int count(uint32_t mask) {
        int n = 0;
        for (; mask; mask &= mask - 1) ++n;
        return n;
    }
    static constexpr int nth(uint32_t mask, int n) {  // index of the n-th set bit
        for (int i = 0; i < 32; ++i)
            if ((mask >> i) & 1u) {
                if (!n) return i;
                --n;
            }
        return -1;
    }
    static constexpr bool any_left(int I0, int I1, uint32_t mask) {
        for (int i = I0; i < I1; ++i)
            if (!((mask >> i) & 1u)) return true;
        return false;
    }
};
// acc += sum_{i = I}^{END-1} x[i] * y[k - i]   (VV: both operand arrays in VGPRs; VS: y = modulus limbs).  FIRST: these are the
// first multiply-adds of their column (the range must not be empty): the carry word is written, not updated
template <int I, int END, int K, class C, bool FIRST = false>
__device__ __forceinline__ void macs_vv(uint64_t& acc, uint32_t& c2, const uint32_t* x, const uint32_t* y) {
    static_assert(!FIRST || END > I, "a column cannot start with an empty range");
    if constexpr (END - I >= 4) {
        mac_vv4<FIRST>(acc, c2, x[I], y[K - I], x[I + 1], y[K - I - 1], x[I + 2], y[K - I - 2], x[I + 3], y[K - I - 3]);
        macs_vv<I + 4, END, K, C>(acc, c2, x, y);
    } else if constexpr (END - I >= 2) {
        mac_vv2<FIRST>(acc, c2, x[I], y[K - I], x[I + 1], y[K - I - 1]);
        macs_vv<I + 2, END, K, C>(acc, c2, x, y);
    } else if constexpr (END - I == 1) {
        mac_vv<FIRST>(acc, c2, x[I], y[K - I]);
    }
}
template <int I, int END, int K, class C, bool FIRST = false>
__device__ __forceinline__ void macs_vs(uint64_t& acc, uint32_t& c2, const uint32_t* x) {
    static_assert(!FIRST || END > I, "a column cannot start with an empty range");
    if constexpr (END - I >= 4) {
        mac_vs4<FIRST>(acc, c2, x[I], C::MOD[K - I], x[I + 1], C::MOD[K - I - 1], x[I + 2], C::MOD[K - I - 2], x[I + 3], C::MOD[K - I - 3]);
        macs_vs<I + 4, END, K, C>(acc, c2, x);
    } else if constexpr (END - I >= 2) {
        mac_vs2<FIRST>(acc, c2, x[I], C::MOD[K - I], x[I + 1], C::MOD[K - I - 1]);
        macs_vs<I + 2, END, K, C>(acc, c2, x);
    } else if constexpr (END - I == 1) {
        mac_vs<FIRST>(acc, c2, x[I], C::MOD[K - I]);
    }
}
// the reduction terms of a column: those of MASK carry-free, the others with the carry word (FIRST: written by the first)
template <int I, int END, int K, class C, uint32_t MASK, int FROM = 0>
__device__ __forceinline__ void macsnc_vs(uint64_t& acc, const uint32_t* x) {  // (the bits of MASK lie in [I, END); as few asm statements as possible)
    typedef NcTerms<C> T;
    constexpr int LEFT = T::count(MASK) - FROM;
    if constexpr (LEFT >= 4) {
        constexpr int i0 = T::nth(MASK, FROM), i1 = T::nth(MASK, FROM + 1), i2 = T::nth(MASK, FROM + 2), i3 = T::nth(MASK, FROM + 3);
        macnc_s4(acc, x[i0], C::MOD[K - i0], x[i1], C::MOD[K - i1], x[i2], C::MOD[K - i2], x[i3], C::MOD[K - i3]);
        macsnc_vs<I, END, K, C, MASK, FROM + 4>(acc, x);
    } else if constexpr (LEFT == 3) {
        constexpr int i0 = T::nth(MASK, FROM), i1 = T::nth(MASK, FROM + 1), i2 = T::nth(MASK, FROM + 2);
        macnc_s3(acc, x[i0], C::MOD[K - i0], x[i1], C::MOD[K - i1], x[i2], C::MOD[K - i2]);
    } else if constexpr (LEFT == 2) {
        constexpr int i0 = T::nth(MASK, FROM), i1 = T::nth(MASK, FROM + 1);
        macnc_s2(acc, x[i0], C::MOD[K - i0], x[i1], C::MOD[K - i1]);
    } else if constexpr (LEFT == 1) {
        constexpr int i0 = T::nth(MASK, FROM);
        macnc_s(acc, x[i0], C::MOD[K - i0]);
    }
}
template <int I, int END, int K, class C, uint32_t MASK, bool FIRST>
__device__ __forceinline__ void macs_vs_sel(uint64_t& acc, uint32_t& c2, const uint32_t* x) {
    if constexpr (I < END) {
        if constexpr ((MASK >> I) & 1u) {
            macs_vs_sel<I + 1, END, K, C, MASK, FIRST>(acc, c2, x);
        } else if constexpr (END - I >= 4 && ((MASK >> I) & 0xfu) == 0) {
            mac_vs4<FIRST>(acc, c2, x[I], C::MOD[K - I], x[I + 1], C::MOD[K - I - 1], x[I + 2], C::MOD[K - I - 2], x[I + 3], C::MOD[K - I - 3]);
            macs_vs_sel<I + 4, END, K, C, MASK, false>(acc, c2, x);
        } else if constexpr (END - I >= 2 && ((MASK >> I) & 0x3u) == 0) {
            mac_vs2<FIRST>(acc, c2, x[I], C::MOD[K - I], x[I + 1], C::MOD[K - I - 1]);
            macs_vs_sel<I + 2, END, K, C, MASK, false>(acc, c2, x);
        } else {
            mac_vs<FIRST>(acc, c2, x[I], C::MOD[K - I]);
            macs_vs_sel<I + 1, END, K, C, MASK, false>(acc, c2, x);
        }
    }
}
// the quotient digit of a column: lo(acc) x INV mod 2^32.  As a v_mad_u64_u32 (30 T/s on gfx950) instead of the v_mul_lo_u32 the
// compiler picks (19 T/s); Fr's INV is -1: a negation
template <class C>
__device__ __forceinline__ uint32_t mont_digit(uint64_t acc) {
    if constexpr (C::INV == 0xffffffffu) {
        return 0u - (uint32_t)acc;
    } else {
        uint64_t t;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"((uint32_t)acc), "s"(C::INV) : "vcc");
        return (uint32_t)t;
    }
}
// end of a column: the low word has been consumed, the accumulator moves down one word (the carry word becomes its high word
// and is written afresh by the next column's first multiply-add)
__device__ __forceinline__ void mont_shift(uint64_t& acc, uint32_t c2) { acc = (acc >> 32) | ((uint64_t)c2 << 32); }
// the multiply-adds of column K before its quotient digit / output word, carry-free terms first (NcTerms).  VS_END: the end of
// the range of reduction terms m[i] MOD[K - i] (K in the low half, where m[K] is still to come, N in the high half)
template <int K, int VS_END, class C>
__device__ __forceinline__ void mont_column_nc(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, const uint32_t* m) {
    constexpr int N = C::N;
    constexpr bool TOPCOL = K >= N - 1;
    constexpr int L = TOPCOL ? K - (N - 1) : 0;                 // lowest index of the column
    constexpr int NTOP = !TOPCOL || !NcTerms<C>::TOPS ? 0 : (L == N - 1 ? 1 : 2);    // terms that hold a top limb (taken carry-free)
    constexpr uint32_t MASK = NcTerms<C>::vs_mask(K, L, VS_END, NTOP * NcTerms<C>::TOP);
    if constexpr (NTOP == 2) macnc_vv(acc, a[L], b[N - 1], a[N - 1], b[L]);
    if constexpr (NTOP == 1) macnc_v(acc, a[L], b[L]);
    macsnc_vs<L, VS_END, K, C, MASK>(acc, m);
    constexpr int V0 = NTOP ? L + 1 : L, V1 = NTOP == 2 ? N - 1 : (NTOP == 1 ? L : (TOPCOL ? N : K + 1));
    constexpr bool VV = V1 > V0, VS = NcTerms<C>::any_left(L, VS_END, MASK);
    if constexpr (VV) macs_vv<V0, V1, K, C, true>(acc, c2, a, b);
    if constexpr (VS) macs_vs_sel<L, VS_END, K, C, MASK, !VV>(acc, c2, m);
    if constexpr (!VV && !VS) c2 = 0;
}
template <int K, class C>
__device__ __forceinline__ void mont_columns_lo(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, uint32_t* m) {
    if constexpr (K < C::N) {
        if constexpr (NcTerms<C>::ON) {
            mont_column_nc<K, K, C>(acc, c2, a, b, m);
        } else {
            macs_vv<0, K + 1, K, C, true>(acc, c2, a, b);
            macs_vs<0, K, K, C>(acc, c2, m);
        }
        m[K] = mont_digit<C>(acc);
        mac_vs(acc, c2, m[K], C::MOD[0]);  // low word is now 0
        mont_shift(acc, c2);
        mont_columns_lo<K + 1, C>(acc, c2, a, b, m);
    }
}
template <int K, class C>
__device__ __forceinline__ void mont_columns_hi(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, const uint32_t* m,
                                                uint32_t* r) {
    if constexpr (K < 2 * C::N - 1) {
        if constexpr (NcTerms<C>::ON) {
            mont_column_nc<K, C::N, C>(acc, c2, a, b, m);
        } else {
            macs_vv<K - C::N + 1, C::N, K, C, true>(acc, c2, a, b);
            macs_vs<K - C::N + 1, C::N, K, C>(acc, c2, m);
        }
        r[K - C::N] = (uint32_t)acc;
        mont_shift(acc, c2);
        mont_columns_hi<K + 1, C>(acc, c2, a, b, m, r);
    }
}
template <class C>
__device__ __forceinline__ Fe<C> fe_mul(const Fe<C>& a, const Fe<C>& b) {
    constexpr int N = C::N;
    uint32_t m[N];
    Fe<C> r;
    uint64_t acc = 0;
    uint32_t c2 = 0;
    mont_columns_lo<0, C>(acc, c2, a.v, b.v, m);
    mont_columns_hi<N, C>(acc, c2, a.v, b.v, m, r.v);
    r.v[N - 1] = (uint32_t)acc;  // the value is < 2p < 2^(32N): nothing above
    fe_reduce_once(r);
    return r;
}
// The same product left in [0, 2p): (a b + m p) / R < a b / R + p, so any operands with a b < p R come out below 2p — and
// p R > 9.8 p^2 for the 381-bit modulus (R = 2^384), so operands below 2p (even 2p x 4p) qualify: a chain of products needs no
// conditional subtraction between its links.  Only for values that are multiplied again (by fe_mul / fe_sqr / this) and never
// compared, added or subtracted before a reducing product has made them canonical.  (Not for Fr: 4 q^2 > q 2^256.)
template <class C>
__device__ __forceinline__ Fe<C> fe_mul_lazy(const Fe<C>& a, const Fe<C>& b) {
    static_assert(C::N == 12, "only the 381-bit modulus leaves the three spare bits this needs");
    constexpr int N = C::N;
    uint32_t m[N];
    Fe<C> r;
    uint64_t acc = 0;
    uint32_t c2 = 0;
    mont_columns_lo<0, C>(acc, c2, a.v, b.v, m);
    mont_columns_hi<N, C>(acc, c2, a.v, b.v, m, r.v);
    r.v[N - 1] = (uint32_t)acc;
    return r;
}
// Montgomery form of a*b + z*w with ONE reduction: the two products are summed column by column before the quotient digit
// of the column is taken (3 N^2 multiply-adds instead of 4 N^2 for two products and an addition).  Needs 2 p^2 < p R, i.e.
// p < R / 2, to come out below 2p: true for both moduli (p < 2^381, q < 2^255).
// (the columns of a b + z w, carry-free terms first: four top-limb terms and the reduction's with MOD[N-1] fit the budget)
template <int K, int VS_END, class C>
__device__ __forceinline__ void mont2_column_nc(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, const uint32_t* z,
                                                const uint32_t* w, const uint32_t* m) {
    constexpr int N = C::N;
    constexpr bool TOPCOL = K >= N - 1;
    constexpr int L = TOPCOL ? K - (N - 1) : 0;
    constexpr int NTOP = !TOPCOL || !NcTerms<C>::TOPS ? 0 : (L == N - 1 ? 2 : 4);
    constexpr uint32_t MASK = NcTerms<C>::vs_mask(K, L, VS_END, NTOP * NcTerms<C>::TOP);
    if constexpr (NTOP == 4) {
        macnc_vv(acc, a[L], b[N - 1], a[N - 1], b[L]);
        macnc_vv(acc, z[L], w[N - 1], z[N - 1], w[L]);
    }
    if constexpr (NTOP == 2) macnc_vv(acc, a[L], b[L], z[L], w[L]);
    macsnc_vs<L, VS_END, K, C, MASK>(acc, m);
    constexpr int V0 = NTOP ? L + 1 : L, V1 = NTOP == 4 ? N - 1 : (NTOP == 2 ? L : (TOPCOL ? N : K + 1));
    constexpr bool VV = V1 > V0, VS = NcTerms<C>::any_left(L, VS_END, MASK);
    if constexpr (VV) {
        macs_vv<V0, V1, K, C, true>(acc, c2, a, b);
        macs_vv<V0, V1, K, C>(acc, c2, z, w);
    }
    if constexpr (VS) macs_vs_sel<L, VS_END, K, C, MASK, !VV>(acc, c2, m);
    if constexpr (!VV && !VS) c2 = 0;
}
template <int K, class C>
__device__ __forceinline__ void mont2_columns_lo(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, const uint32_t* z,
                                                 const uint32_t* w, uint32_t* m) {
    if constexpr (K < C::N) {
        if constexpr (NcTerms<C>::ON) {
            mont2_column_nc<K, K, C>(acc, c2, a, b, z, w, m);
        } else {
            macs_vv<0, K + 1, K, C, true>(acc, c2, a, b);
            macs_vv<0, K + 1, K, C>(acc, c2, z, w);
            macs_vs<0, K, K, C>(acc, c2, m);
        }
        m[K] = mont_digit<C>(acc);
        mac_vs(acc, c2, m[K], C::MOD[0]);  // low word is now 0
        mont_shift(acc, c2);
        mont2_columns_lo<K + 1, C>(acc, c2, a, b, z, w, m);
    }
}
template <int K, class C>
__device__ __forceinline__ void mont2_columns_hi(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b, const uint32_t* z,
                                                 const uint32_t* w, const uint32_t* m, uint32_t* r) {
    if constexpr (K < 2 * C::N - 1) {
        if constexpr (NcTerms<C>::ON) {
            mont2_column_nc<K, C::N, C>(acc, c2, a, b, z, w, m);
        } else {
            macs_vv<K - C::N + 1, C::N, K, C, true>(acc, c2, a, b);
            macs_vv<K - C::N + 1, C::N, K, C>(acc, c2, z, w);
            macs_vs<K - C::N + 1, C::N, K, C>(acc, c2, m);
        }
        r[K - C::N] = (uint32_t)acc;
        mont_shift(acc, c2);
        mont2_columns_hi<K + 1, C>(acc, c2, a, b, z, w, m, r);
    }
}
template <class C>
__device__ __forceinline__ Fe<C> fe_mul2(const Fe<C>& a, const Fe<C>& b, const Fe<C>& z, const Fe<C>& w) {
    constexpr int N = C::N;
    uint32_t m[N];
    Fe<C> r;
    uint64_t acc = 0;
    uint32_t c2 = 0;
    mont2_columns_lo<0, C>(acc, c2, a.v, b.v, z.v, w.v, m);
    mont2_columns_hi<N, C>(acc, c2, a.v, b.v, z.v, w.v, m, r.v);
    r.v[N - 1] = (uint32_t)acc;  // (2 p^2 + R p) / R < 2p < 2^(32N): nothing above
    fe_reduce_once(r);
    return r;
}
// CIOS.  Because 2p - 1 < 2^(32N) the running value fits in N+1 limbs (invariant t <= 2p - 1 after
// every outer iteration).
template <class C>
__host__ inline Fe<C> fe_mul(const Fe<C>& a, const Fe<C>& b) {
    constexpr int N = C::N;
    uint32_t t[N + 1];
    for (int i = 0; i <= N; ++i) t[i] = 0;
    for (int i = 0; i < N; ++i) {
        uint32_t bi = b.v[i];
        uint64_t c = 0;
        for (int j = 0; j < N; ++j) {
            uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        t[N] += (uint32_t)c;
        uint32_t m = t[0] * C::INV;
        c = ((uint64_t)m * C::MOD[0] + t[0]) >> 32;
        for (int j = 1; j < N; ++j) {
            uint64_t x = (uint64_t)m * C::MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)x;
        t[N] = (uint32_t)(x >> 32);
    }
    Fe<C> r;
    for (int i = 0; i < N; ++i) r.v[i] = t[i];
    fe_reduce_once(r);
    return r;
}
template <class C>
__host__ inline Fe<C> fe_mul_lazy(const Fe<C>& a, const Fe<C>& b) {
    return fe_mul(a, b);
}
// Montgomery square.  Device form: the same product scanning with the operand doubled ONCE up front (d = 2a < 2^(32N):
// both moduli leave spare top bits).  With B = 2^32 and P_j = a mod B^j:
//     2 sum_{i<j} a_i a_j B^(i+j) = sum_j a_j B^j (2 P_j),      2 P_j = sum_{i<j} d_i B^i + B^j msb(a_{j-1})
// (the limbs of d below j are those of 2 P_j except for the bit that the shift pushed out of limb j-1), so column k is
//     sum_{i < k-i} d_i a_{k-i}  +  [k = 2j] (a_j^2 + msb(a_{j-1}) a_j)  +  the reduction terms:
// N(N+1)/2 + (N-1) + N^2 multiply-adds instead of 2 N^2 (233 instead of 288 for N = 12).
// (the columns of a square, carry-free terms first: the cross term with a[N-1], the last column's square, small reduction terms)
template <int K, int VS_END, class C>
__device__ __forceinline__ void sqr_column_nc(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* a2, const uint32_t* m) {
    constexpr int N = C::N;
    constexpr bool TOPCOL = K >= N - 1;
    constexpr int L = TOPCOL ? K - (N - 1) : 0;
    constexpr bool LAST = NcTerms<C>::TOPS && L == N - 1;  // a[N-1]^2 + the shifted-out bit x a[N-1]: both small
    constexpr bool XTOP = NcTerms<C>::TOPS && TOPCOL && !LAST;   // the cross term a2[L] a[N-1]
    constexpr uint32_t MASK = NcTerms<C>::vs_mask(K, L, VS_END, (XTOP ? NcTerms<C>::TOP : 0) + (LAST ? NcTerms<C>::TOP + 2 : 0));
    if constexpr (XTOP) macnc_v(acc, a2[L], a[N - 1]);
    if constexpr (LAST) macnc_vv(acc, a[L], a[L], a[L - 1] >> 31, a[L]);
    macsnc_vs<L, VS_END, K, C, MASK>(acc, m);
    constexpr int X0 = L + (XTOP ? 1 : 0), X1 = (K + 1) / 2;
    constexpr bool CROSS = X1 > X0, SQ = K % 2 == 0 && !LAST, VS = NcTerms<C>::any_left(L, VS_END, MASK);
    if constexpr (CROSS) macs_vv<X0, X1, K, C, true>(acc, c2, a2, a);
    if constexpr (SQ) {
        mac_vv<!CROSS>(acc, c2, a[K / 2], a[K / 2]);
        if constexpr (K >= 2) mac_vv(acc, c2, a[K / 2 - 1] >> 31, a[K / 2]);
    }
    if constexpr (VS) macs_vs_sel<L, VS_END, K, C, MASK, !CROSS && !SQ>(acc, c2, m);
    if constexpr (!CROSS && !SQ && !VS) c2 = 0;
}
template <int K, class C>
__device__ __forceinline__ void sqr_columns_lo(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* a2, uint32_t* m) {
    if constexpr (K < C::N) {
        if constexpr (NcTerms<C>::ON) {
            sqr_column_nc<K, K, C>(acc, c2, a, a2, m);
        } else {
            constexpr bool CROSS = (K + 1) / 2 > 0;  // (column 0 has no cross terms: its square comes first)
            if constexpr (CROSS) macs_vv<0, (K + 1) / 2, K, C, true>(acc, c2, a2, a);
            if constexpr (K % 2 == 0) mac_vv<!CROSS>(acc, c2, a[K / 2], a[K / 2]);
            if constexpr (K % 2 == 0 && K >= 2) mac_vv(acc, c2, a[K / 2 - 1] >> 31, a[K / 2]);
            macs_vs<0, K, K, C>(acc, c2, m);
        }
        m[K] = mont_digit<C>(acc);
        mac_vs(acc, c2, m[K], C::MOD[0]);  // low word is now 0
        mont_shift(acc, c2);
        sqr_columns_lo<K + 1, C>(acc, c2, a, a2, m);
    }
}
template <int K, class C>
__device__ __forceinline__ void sqr_columns_hi(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* a2, const uint32_t* m, uint32_t* r) {
    if constexpr (K < 2 * C::N - 1) {
        if constexpr (NcTerms<C>::ON) {
            sqr_column_nc<K, C::N, C>(acc, c2, a, a2, m);
        } else {
            constexpr bool CROSS = (K + 1) / 2 > K - C::N + 1;  // (the last column has no cross terms)
            if constexpr (CROSS) macs_vv<K - C::N + 1, (K + 1) / 2, K, C, true>(acc, c2, a2, a);
            if constexpr (K % 2 == 0) {
                mac_vv<!CROSS>(acc, c2, a[K / 2], a[K / 2]);
                mac_vv(acc, c2, a[K / 2 - 1] >> 31, a[K / 2]);
            }
            macs_vs<K - C::N + 1, C::N, K, C>(acc, c2, m);
        }
        r[K - C::N] = (uint32_t)acc;
        mont_shift(acc, c2);
        sqr_columns_hi<K + 1, C>(acc, c2, a, a2, m, r);
    }
}
template <class C>
__device__ __forceinline__ Fe<C> fe_sqr(const Fe<C>& a) {
    constexpr int N = C::N;
    uint32_t m[N], a2[N];
    a2[0] = a.v[0] << 1;
#pragma unroll
    for (int i = 1; i < N; ++i) a2[i] = __funnelshift_l(a.v[i - 1], a.v[i], 1);
    Fe<C> r;
    uint64_t acc = 0;
    uint32_t c2 = 0;
    sqr_columns_lo<0, C>(acc, c2, a.v, a2, m);
    sqr_columns_hi<N, C>(acc, c2, a.v, a2, m, r.v);
    r.v[N - 1] = (uint32_t)acc;  // < 2p < 2^(32N)
    fe_reduce_once(r);
    return r;
}
template <class C>
__host__ inline Fe<C> fe_sqr(const Fe<C>& a) {
    return fe_mul(a, a);
}
// Out-of-line product: used where code size matters more than the call (G2, cold kernels, serial tails).
// On the device the 384-bit operands travel in VGPRs: as six 4-dword vectors they are register arguments of the AMDGPU
// calling convention (an aggregate passed by reference would be spilled to scratch by the caller and re-loaded with flat
// loads by the callee: ~36 dwords of scratch traffic and a full memory round trip per product).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct FpRegs {
    u32x4 q0, q1, q2;
};
__device__ __noinline__ FpRegs fp_mul_call(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
    Fe<FpCfg> a, b;
    a.v[0] = a0.x; a.v[1] = a0.y; a.v[2] = a0.z; a.v[3] = a0.w;
    a.v[4] = a1.x; a.v[5] = a1.y; a.v[6] = a1.z; a.v[7] = a1.w;
    a.v[8] = a2.x; a.v[9] = a2.y; a.v[10] = a2.z; a.v[11] = a2.w;
    b.v[0] = b0.x; b.v[1] = b0.y; b.v[2] = b0.z; b.v[3] = b0.w;
    b.v[4] = b1.x; b.v[5] = b1.y; b.v[6] = b1.z; b.v[7] = b1.w;
    b.v[8] = b2.x; b.v[9] = b2.y; b.v[10] = b2.z; b.v[11] = b2.w;
    Fe<FpCfg> r = fe_mul(a, b);
    FpRegs o;
    o.q0 = u32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    o.q1 = u32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
    o.q2 = u32x4{r.v[8], r.v[9], r.v[10], r.v[11]};
    return o;
}
__device__ __noinline__ FpRegs fp_sqr_call(u32x4 a0, u32x4 a1, u32x4 a2) {
    Fe<FpCfg> a;
    a.v[0] = a0.x; a.v[1] = a0.y; a.v[2] = a0.z; a.v[3] = a0.w;
    a.v[4] = a1.x; a.v[5] = a1.y; a.v[6] = a1.z; a.v[7] = a1.w;
    a.v[8] = a2.x; a.v[9] = a2.y; a.v[10] = a2.z; a.v[11] = a2.w;
    Fe<FpCfg> r = fe_sqr(a);
    FpRegs o;
    o.q0 = u32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    o.q1 = u32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
    o.q2 = u32x4{r.v[8], r.v[9], r.v[10], r.v[11]};
    return o;
}
// by-reference form for the other field (Fr: only in set-up code and the final assembly)
template <class C>
__device__ __noinline__ Fe<C> fe_mul_ref(const Fe<C>& a, const Fe<C>& b) {
    return fe_mul(a, b);
}
template <class C>
MASP_HD Fe<C> fe_mul_nc(const Fe<C>& a, const Fe<C>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (C::N == 12) {
        FpRegs o = fp_mul_call(u32x4{a.v[0], a.v[1], a.v[2], a.v[3]}, u32x4{a.v[4], a.v[5], a.v[6], a.v[7]},
                               u32x4{a.v[8], a.v[9], a.v[10], a.v[11]}, u32x4{b.v[0], b.v[1], b.v[2], b.v[3]},
                               u32x4{b.v[4], b.v[5], b.v[6], b.v[7]}, u32x4{b.v[8], b.v[9], b.v[10], b.v[11]});
        Fe<C> r;
        r.v[0] = o.q0.x; r.v[1] = o.q0.y; r.v[2] = o.q0.z; r.v[3] = o.q0.w;
        r.v[4] = o.q1.x; r.v[5] = o.q1.y; r.v[6] = o.q1.z; r.v[7] = o.q1.w;
        r.v[8] = o.q2.x; r.v[9] = o.q2.y; r.v[10] = o.q2.z; r.v[11] = o.q2.w;
        return r;
    } else {
        return fe_mul_ref(a, b);
    }
#else
    return fe_mul(a, b);
#endif
}

template <class C>
MASP_HD Fe<C> fe_sqr_nc(const Fe<C>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (C::N == 12) {
        FpRegs o = fp_sqr_call(u32x4{a.v[0], a.v[1], a.v[2], a.v[3]}, u32x4{a.v[4], a.v[5], a.v[6], a.v[7]}, u32x4{a.v[8], a.v[9], a.v[10], a.v[11]});
        Fe<C> r;
        r.v[0] = o.q0.x; r.v[1] = o.q0.y; r.v[2] = o.q0.z; r.v[3] = o.q0.w;
        r.v[4] = o.q1.x; r.v[5] = o.q1.y; r.v[6] = o.q1.z; r.v[7] = o.q1.w;
        r.v[8] = o.q2.x; r.v[9] = o.q2.y; r.v[10] = o.q2.z; r.v[11] = o.q2.w;
        return r;
    } else {
        return fe_mul_ref(a, a);
    }
#else
    return fe_mul(a, a);
#endif
}

// canonical integer limbs <-> Montgomery form
template <class C>
MASP_HD Fe<C> fe_to_mont(const Fe<C>& canonical) {
    Fe<C> r2;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r2.v[i] = C::R2[i];
    return fe_mul(canonical, r2);
}
template <class C>
MASP_HD Fe<C> fe_from_mont(const Fe<C>& a) {
    Fe<C> one;
#pragma unroll
    for (int i = 0; i < C::N; ++i) one.v[i] = 0;
    one.v[0] = 1;
    return fe_mul(a, one);
}
// a (canonical) >= p ?
template <class C>
MASP_HD bool fe_canonical_ge_mod(const Fe<C>& a) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        uint64_t d = (uint64_t)a.v[i] - C::MOD[i] - borrow;
        borrow = (d >> 32) & 1;
    }
    return !borrow;
}

// a^e for a public exponent given as N little-endian limbs (not unrolled: used off the hot path)
template <class C>
MASP_NOINLINE Fe<C> fe_pow(const Fe<C>& a, const uint32_t* e, int nlimbs) {
    Fe<C> r = fe_one<C>();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) r = fe_sqr_nc(r);
            if ((e[i] >> b) & 1) {
                r = started ? fe_mul_nc(r, a) : a;
                started = true;
            }
        }
    return r;
}
// Inverse (inv(0) = 0) by batched division steps (Bernstein-Yang "safegcd", variable time): 62 steps are decided on the
// low words and applied to the full-width values as one 2x2 integer matrix, ~10 rounds for 384 bits.  ~12x fewer
// instructions than the Fermat power (381 squarings + ~190 products) it replaces — it sits on the serial tail of every
// proof (three affine conversions in the assembly) and in the window-table precomputation at load time.
// Values are L signed limbs of 62 bits; 64 x 64 -> 128-bit products (`__int128` works in device code).
template <class C>
struct FeDivsteps {
    static constexpr int L = (32 * C::N + 61) / 62 + (((32 * C::N + 61) / 62) * 62 - 32 * C::N < 2 ? 1 : 0);  // room for a sign
    static constexpr uint64_t M62 = ~0ull >> 2;
    typedef __int128 i128;

    static MASP_HD void to62(int64_t* o, const uint32_t* a) {
        for (int i = 0; i < L; ++i) {
            uint64_t v = 0;
            for (int b = 0; b < 62; b += 1) {
                int bit = 62 * i + b;
                if (bit >= 32 * C::N) break;
                v |= (uint64_t)((a[bit >> 5] >> (bit & 31)) & 1u) << b;
            }
            o[i] = (int64_t)v;
        }
    }
    static MASP_HD void from62(uint32_t* o, const int64_t* a) {
        for (int w = 0; w < C::N; ++w) o[w] = 0;
        for (int i = 0; i < L; ++i)
            for (int b = 0; b < 62; ++b) {
                int bit = 62 * i + b;
                if (bit >= 32 * C::N) break;
                o[bit >> 5] |= (uint32_t)(((uint64_t)a[i] >> b) & 1u) << (bit & 31);
            }
    }
    static MASP_HD int ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __ffsll((unsigned long long)x) - 1;
#else
        return __builtin_ctzll(x);
#endif
    }
    // up to 62 division steps on the low words; eta = -delta.  2^62 (f', g') = [[u v] [q r]] (f, g)
    static MASP_HD int64_t divsteps(int64_t eta, uint64_t f, uint64_t g, int64_t* t) {
        uint64_t u = 1, v = 0, q = 0, r = 1;
        int i = 62;
        for (;;) {
            int zeros = ctz64(g | (~0ull << i));
            g >>= zeros;
            u <<= zeros;
            v <<= zeros;
            eta -= zeros;
            i -= zeros;
            if (i == 0) break;
            if (eta < 0) {
                uint64_t tmp;
                eta = -eta;
                tmp = f; f = g; g = (uint64_t)0 - tmp;
                tmp = u; u = q; q = (uint64_t)0 - tmp;
                tmp = v; v = r; r = (uint64_t)0 - tmp;
            }
            // cancel the low bits of g with a multiple of f (up to 4 at once): w = -g / f mod 2^k
            int limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            uint64_t m = (~0ull >> (64 - limit)) & 15u;
            uint64_t w = f + (((f + 1) & 4) << 1);  // f^-1 mod 16
            w = ((uint64_t)0 - w * g) & m;
            g += f * w;
            q += u * w;
            r += v * w;
        }
        t[0] = (int64_t)u;
        t[1] = (int64_t)v;
        t[2] = (int64_t)q;
        t[3] = (int64_t)r;
        return eta;
    }
    static MASP_HD void update_fg(int64_t* f, int64_t* g, const int64_t* t) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        i128 cf = (i128)u * f[0] + (i128)v * g[0];
        i128 cg = (i128)q * f[0] + (i128)r * g[0];
        cf >>= 62;
        cg >>= 62;
        for (int i = 1; i < L; ++i) {
            cf += (i128)u * f[i] + (i128)v * g[i];
            cg += (i128)q * f[i] + (i128)r * g[i];
            f[i - 1] = (int64_t)((uint64_t)cf & M62);
            g[i - 1] = (int64_t)((uint64_t)cg & M62);
            cf >>= 62;
            cg >>= 62;
        }
        f[L - 1] = (int64_t)cf;
        g[L - 1] = (int64_t)cg;
    }
    // (d, e) <- t (d, e) / 2^62 mod p, both kept in (-2p, p)
    static MASP_HD void update_de(int64_t* d, int64_t* e, const int64_t* t, const int64_t* p62, uint64_t p_inv62) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        const int64_t sd = d[L - 1] >> 63, se = e[L - 1] >> 63;
        int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
        i128 cd = (i128)u * d[0] + (i128)v * e[0];
        i128 ce = (i128)q * d[0] + (i128)r * e[0];
        md -= (int64_t)((p_inv62 * (uint64_t)cd + (uint64_t)md) & M62);
        me -= (int64_t)((p_inv62 * (uint64_t)ce + (uint64_t)me) & M62);
        cd += (i128)p62[0] * md;
        ce += (i128)p62[0] * me;
        cd >>= 62;
        ce >>= 62;
        for (int i = 1; i < L; ++i) {
            cd += (i128)u * d[i] + (i128)v * e[i] + (i128)p62[i] * md;
            ce += (i128)q * d[i] + (i128)r * e[i] + (i128)p62[i] * me;
            d[i - 1] = (int64_t)((uint64_t)cd & M62);
            e[i - 1] = (int64_t)((uint64_t)ce & M62);
            cd >>= 62;
            ce >>= 62;
        }
        d[L - 1] = (int64_t)cd;
        e[L - 1] = (int64_t)ce;
    }
    static MASP_HD void carry(int64_t* r) {
        for (int i = 0; i < L - 1; ++i) {
            r[i + 1] += r[i] >> 62;
            r[i] &= (int64_t)M62;
        }
    }
    // plain residues: out = x^-1 mod p (0 for x = 0)
    static MASP_HD void invert(uint32_t* out, const uint32_t* x) {
        int64_t f[L], g[L], d[L], e[L], p62[L];
        to62(p62, C::MOD);
        to62(g, x);
        uint64_t nz = 0;
        for (int i = 0; i < L; ++i) {
            f[i] = p62[i];
            d[i] = e[i] = 0;
            nz |= (uint64_t)g[i];
        }
        if (!nz) {
            for (int w = 0; w < C::N; ++w) out[w] = 0;
            return;
        }
        e[0] = 1;
        uint64_t pinv = 1;  // p^-1 mod 2^64 by Newton, then mod 2^62
        const uint64_t p0 = (uint64_t)C::MOD[0] | ((uint64_t)C::MOD[1] << 32);
        for (int i = 0; i < 6; ++i) pinv *= 2 - p0 * pinv;
        pinv &= M62;
        int64_t eta = -1;
        for (int round = 0; round < 64; ++round) {  // 12 rounds cover the 735-step bound for 384 bits
            int64_t t[4];
            eta = divsteps(eta, (uint64_t)f[0], (uint64_t)g[0], t);
            update_de(d, e, t, p62, pinv);
            update_fg(f, g, t);
            int64_t any = 0;
            for (int i = 0; i < L; ++i) any |= g[i];
            if (any == 0) break;
        }
        // f = +-1 ; x^-1 = d * f, brought into [0, p)
        const bool negate = f[L - 1] < 0;
        carry(d);
        if (d[L - 1] < 0) {
            for (int i = 0; i < L; ++i) d[i] += p62[i];
            carry(d);
        }
        if (negate) {
            for (int i = 0; i < L; ++i) d[i] = -d[i];
            carry(d);
        }
        for (int guard = 0; guard < 4 && d[L - 1] < 0; ++guard) {
            for (int i = 0; i < L; ++i) d[i] += p62[i];
            carry(d);
        }
        for (int guard = 0; guard < 4; ++guard) {  // subtract p while d >= p
            int64_t s[L];
            for (int i = 0; i < L; ++i) s[i] = d[i] - p62[i];
            carry(s);
            if (s[L - 1] < 0) break;
            for (int i = 0; i < L; ++i) d[i] = s[i];
        }
        from62(out, d);
    }
};
// Fermat inverse (inv(0) = 0): 381 squarings + ~190 products.  More instructions than the divsteps, but a straight
// chain of products — on a lone lane (the assembly's three affine conversions) it is the faster of the two (1.4 ms
// against 2.3 ms), while in bulk (table precomputation, every lane busy) the divsteps win 3-4x.
template <class C>
MASP_NOINLINE Fe<C> fe_inv_fermat(const Fe<C>& a) {
    uint32_t e[C::N];
    for (int i = 0; i < C::N; ++i) e[i] = C::PM2[i];
    return fe_pow(a, e, C::N);
}
template <class C>
MASP_NOINLINE Fe<C> fe_inv(const Fe<C>& a) {
    Fe<C> r, r2;
    FeDivsteps<C>::invert(r.v, a.v);  // (a R)^-1 from the Montgomery representative
#pragma unroll
    for (int i = 0; i < C::N; ++i) r2.v[i] = C::R2[i];
    return fe_mul_nc(fe_mul_nc(r, r2), r2);  // (aR)^-1 R^2 R^-1 = a^-1, once more: a^-1 R
}
// Inverse (inv(0) = 0) by the binary extended Euclid in 32-bit limbs, the form for a lane that inverts ONE value while its
// neighbours do the same (the shared inversions of the batch-affine bucket trees, device/msm_tree.hpp): ~1.45 log2 p rounds of
// ~150 full-rate integer instructions and not a single multiplication — on a lone wave ~6x sooner than the Fermat power
// (570 dependent 384-bit products), and free of the 64/128-bit arithmetic the divsteps above are written in.
// Invariants: a = u y, b = v y (mod p), b odd; a reaches 0 with b = gcd = 1 and v = 1 / y.  Branch-free inside a round
// (lanes of a wave hold different values); the loop itself ends per lane.
template <class C>
MASP_HD void fe_bingcd_inv(uint32_t* out, const uint32_t* y) {
    constexpr int N = C::N;
    uint32_t a[N], b[N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        a[i] = y[i];
        b[i] = C::MOD[i];
        u[i] = i == 0 ? 1u : 0u;
        v[i] = 0u;
    }
    for (int round = 0; round < 64 * N + 2; ++round) {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) nz |= a[i];
        if (!nz) break;
        const uint32_t odd = 0u - (a[0] & 1u);  // mask
        // d = a - b, borrow <=> a < b
        uint32_t d[N], w[N];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a[i] - b[i] - br;
            d[i] = (uint32_t)t;
            br = (t >> 32) & 1;
        }
        const uint32_t lt = (0u - (uint32_t)br) & odd;  // odd and a < b: the roles swap
        // w = u - v mod p
        br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)u[i] - v[i] - br;
            w[i] = (uint32_t)t;
            br = (t >> 32) & 1;
        }
        {
            const uint32_t m = 0u - (uint32_t)br;
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                uint64_t t = (uint64_t)w[i] + (C::MOD[i] & m) + c;
                w[i] = (uint32_t)t;
                c = t >> 32;
            }
        }
        // swap case: (a, b, u, v) <- (b - a, a, v - u, u) ; v - u = p - w unless w = 0.  |d| = -d when a < b.
        uint32_t wz = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) wz |= w[i];
        const uint32_t negw = lt & (0u - (uint32_t)(wz != 0));
        {
            // a <- odd ? |a - b| : a ;  b <- lt ? a : b
            uint64_t c = lt & 1u;  // two's complement negate of d under lt: (~d) + 1
            uint64_t brw = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t ai = a[i];
                uint64_t t = (uint64_t)(d[i] ^ lt) + c;
                c = t >> 32;
                a[i] = (ai & ~odd) | ((uint32_t)t & odd);
                b[i] = (b[i] & ~lt) | (ai & lt);
                // u <- odd ? (lt ? p - w : w) : u ;  v <- lt ? u : v
                const uint32_t ui = u[i];
                uint64_t pw = (uint64_t)C::MOD[i] - w[i] - brw;
                brw = (pw >> 32) & 1;
                const uint32_t wn = (w[i] & ~negw) | ((uint32_t)pw & negw);
                u[i] = (ui & ~odd) | (wn & odd);
                v[i] = (v[i] & ~lt) | (ui & lt);
            }
        }
        // a is even now: a >>= 1 ; u <- u / 2 mod p  (u + p if u is odd; u + p < 2^(32N) since p leaves a spare top bit)
        {
            const uint32_t uo = 0u - (u[0] & 1u);
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                uint64_t t = (uint64_t)u[i] + (C::MOD[i] & uo) + c;
                u[i] = (uint32_t)t;
                c = t >> 32;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                a[i] = (a[i] >> 1) | (i + 1 < N ? a[i + 1] << 31 : 0u);
                u[i] = (u[i] >> 1) | (i + 1 < N ? u[i + 1] << 31 : 0u);
            }
        }
    }
    // y = 0: a was 0 from the start, v = 0
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = v[i];
}
// The same inverse with the rounds taken 30 at a time (T. Pornin, "Optimized Binary GCD for Modular Inversion"): 30 rounds
// are decided on 64-bit approximations of a and b — their 30 low bits, which the parities depend on, and the 34 top bits of
// the larger, which the comparisons depend on — while a 2x2 matrix (f0 g0; f1 g1), |f| + |g| <= 2^30 per row, records what was
// done; then a, b, u, v are each replaced by their combination under that matrix divided by 2^30 (exactly for a and b, modulo p
// for u and v).  A wrong comparison near equality only makes a or b negative, which is repaired by negating a matrix row.
// ceil((2 len - 1) / 30) such steps always suffice: 26 for Fp, 18 for Fr; ~5x fewer instructions than one round at a time.
template <class C>
MASP_HD void fe_bingcd30_inv(uint32_t* out, const uint32_t* y) {
    constexpr int N = C::N, K = 30, STEPS = (2 * 32 * N + K - 1) / K;
    constexpr uint32_t KMASK = (1u << K) - 1u;
    const uint32_t minv = (0u - C::INV) & KMASK;  // p^-1 mod 2^30 (C::INV = -p^-1 mod 2^32)
    uint32_t a[N], b[N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        a[i] = y[i];
        b[i] = C::MOD[i];
        u[i] = i == 0 ? 1u : 0u;
        v[i] = 0u;
    }
    for (int step = 0; step < STEPS; ++step) {
        // ---- approximations: the three top limbs of max(a, b), normalised, and limb 0
        uint32_t a2 = 0, a1 = 0, a0 = 0, b2 = 0, b1 = 0, b0 = 0;
        bool found = false;
#pragma unroll
        for (int i = N - 1; i >= 2; --i) {
            const bool here = !found && (a[i] | b[i]) != 0;
            if (here) {
                a2 = a[i]; a1 = a[i - 1]; a0 = a[i - 2];
                b2 = b[i]; b1 = b[i - 1]; b0 = b[i - 2];
            }
            found = found || here;
        }
        uint64_t xa, xb;
        if (!found) {  // both below 2^64: exact
            xa = ((uint64_t)a[1] << 32) | a[0];
            xb = ((uint64_t)b[1] << 32) | b[0];
        } else {
            const uint32_t top = a2 | b2;
            int sh = 0;  // leading zeros of top (top != 0)
            for (uint32_t t = top; !(t & 0x80000000u); t <<= 1) ++sh;
            uint64_t ha = ((uint64_t)a2 << 32) | a1, hb = ((uint64_t)b2 << 32) | b1;
            if (sh) {
                ha = (ha << sh) | (a0 >> (32 - sh));
                hb = (hb << sh) | (b0 >> (32 - sh));
            }
            xa = ((ha >> K) << K) | (a[0] & KMASK);  // 34 top bits | 30 low bits
            xb = ((hb >> K) << K) | (b[0] & KMASK);
        }
        // ---- 30 rounds on the approximations
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int r = 0; r < K; ++r) {
            const uint64_t odd = 0ull - (xa & 1ull);
            const uint64_t sw = odd & (0ull - (uint64_t)(xa < xb));
            const uint32_t odd32 = (uint32_t)odd, sw32 = (uint32_t)sw;
            const uint64_t tx = (xa ^ xb) & sw;
            xa ^= tx;
            xb ^= tx;
            const uint32_t tf = (uint32_t)(f0 ^ f1) & sw32, tg = (uint32_t)(g0 ^ g1) & sw32;
            f0 = (int32_t)((uint32_t)f0 ^ tf);
            f1 = (int32_t)((uint32_t)f1 ^ tf);
            g0 = (int32_t)((uint32_t)g0 ^ tg);
            g1 = (int32_t)((uint32_t)g1 ^ tg);
            xa -= xb & odd;
            f0 -= (int32_t)((uint32_t)f1 & odd32);
            g0 -= (int32_t)((uint32_t)g1 & odd32);
            xa >>= 1;
            f1 = (int32_t)((uint32_t)f1 << 1);
            g1 = (int32_t)((uint32_t)g1 << 1);
        }
        // ---- (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30, made non-negative
        uint32_t ta[N + 1], tb[N + 1];
        {
            int64_t ca = 0, cb = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                ca += (int64_t)f0 * (int64_t)(uint64_t)a[i] + (int64_t)g0 * (int64_t)(uint64_t)b[i];
                cb += (int64_t)f1 * (int64_t)(uint64_t)a[i] + (int64_t)g1 * (int64_t)(uint64_t)b[i];
                ta[i] = (uint32_t)ca;
                tb[i] = (uint32_t)cb;
                ca >>= 32;
                cb >>= 32;
            }
            ta[N] = (uint32_t)ca;
            tb[N] = (uint32_t)cb;
            const uint32_t na = 0u - (uint32_t)(ca < 0), nb_ = 0u - (uint32_t)(cb < 0);
            uint64_t c1 = na & 1u, c2 = nb_ & 1u;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t sa = (ta[i] >> K) | (ta[i + 1] << (32 - K)), sb = (tb[i] >> K) | (tb[i + 1] << (32 - K));
                c1 += (uint64_t)(sa ^ na);
                c2 += (uint64_t)(sb ^ nb_);
                a[i] = (uint32_t)c1;
                b[i] = (uint32_t)c2;
                c1 >>= 32;
                c2 >>= 32;
            }
            // a negated value means the opposite matrix row was applied
            f0 = (int32_t)(((uint32_t)f0 ^ na) - na);
            g0 = (int32_t)(((uint32_t)g0 ^ na) - na);
            f1 = (int32_t)(((uint32_t)f1 ^ nb_) - nb_);
            g1 = (int32_t)(((uint32_t)g1 ^ nb_) - nb_);
        }
        // ---- (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^30 mod p: add the multiple of p that clears the 30 low bits, shift, and
        // bring the result (in (-p, 2p)) back into [0, p)
        {
            int64_t cu = 0, cv = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                cu += (int64_t)f0 * (int64_t)(uint64_t)u[i] + (int64_t)g0 * (int64_t)(uint64_t)v[i];
                cv += (int64_t)f1 * (int64_t)(uint64_t)u[i] + (int64_t)g1 * (int64_t)(uint64_t)v[i];
                ta[i] = (uint32_t)cu;
                tb[i] = (uint32_t)cv;
                cu >>= 32;
                cv >>= 32;
            }
            const uint32_t qu = ((0u - ta[0]) * minv) & KMASK, qv = ((0u - tb[0]) * minv) & KMASK;
            uint64_t du = 0, dv = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                du += (uint64_t)qu * C::MOD[i] + ta[i];
                dv += (uint64_t)qv * C::MOD[i] + tb[i];
                ta[i] = (uint32_t)du;
                tb[i] = (uint32_t)dv;
                du >>= 32;
                dv >>= 32;
            }
            cu += (int64_t)du;
            cv += (int64_t)dv;
            ta[N] = (uint32_t)cu;
            tb[N] = (uint32_t)cv;
            const bool negu = cu < 0, negv = cv < 0;  // (after the shift the sign sits in the bits above N limbs)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                u[i] = (ta[i] >> K) | (ta[i + 1] << (32 - K));
                v[i] = (tb[i] >> K) | (tb[i + 1] << (32 - K));
            }
            // negative: + p (the N-limb two's complement wraps to the right value); otherwise - p if >= p
            {
                const uint32_t mu = 0u - (uint32_t)negu, mv = 0u - (uint32_t)negv;
                uint64_t c1 = 0, c2 = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    c1 += (uint64_t)u[i] + (C::MOD[i] & mu);
                    c2 += (uint64_t)v[i] + (C::MOD[i] & mv);
                    u[i] = (uint32_t)c1;
                    v[i] = (uint32_t)c2;
                    c1 >>= 32;
                    c2 >>= 32;
                }
                uint32_t su[N], sv[N];
                uint64_t b1 = 0, b2 = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    uint64_t t1 = (uint64_t)u[i] - C::MOD[i] - b1, t2 = (uint64_t)v[i] - C::MOD[i] - b2;
                    su[i] = (uint32_t)t1;
                    sv[i] = (uint32_t)t2;
                    b1 = (t1 >> 32) & 1;
                    b2 = (t2 >> 32) & 1;
                }
                const uint32_t ku = 0u - (uint32_t)(b1 == 0), kv = 0u - (uint32_t)(b2 == 0);  // no borrow: value >= p
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    u[i] = (u[i] & ~ku) | (su[i] & ku);
                    v[i] = (v[i] & ~kv) | (sv[i] & kv);
                }
            }
        }
    }
    // a = 0, b = gcd = 1 (y = 0: b = p, v = 0): v = 1 / y
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = v[i];
}
// a^-1 in Montgomery form from a in Montgomery form: (aR)^-1 = a^-1 R^-1, times R^3 (as a Montgomery product) = a^-1 R
template <class C>
MASP_HD Fe<C> fe_inv_bingcd(const Fe<C>& a) {
    Fe<C> r, r3;
    fe_bingcd30_inv<C>(r.v, a.v);
#pragma unroll
    for (int i = 0; i < C::N; ++i) r3.v[i] = C::R3[i];
    return fe_mul(r, r3);
}
// out of line, for the serial tails (one lane converting one point): ~0.1 ms where the Fermat power takes ~1.1 ms
template <class C>
MASP_NOINLINE Fe<C> fe_inv_bingcd_nc(const Fe<C>& a) {
    return fe_inv_bingcd(a);
}
// canonical value > (p-1)/2 ?  (zcash "lexicographically largest")
template <class C>
MASP_HD bool fe_canonical_gt_half(const Fe<C>& canon) {
    // HALF - canon borrows  <=>  canon > HALF
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        uint64_t d = (uint64_t)C::HALF[i] - canon.v[i] - borrow;
        borrow = (d >> 32) & 1;
    }
    return borrow != 0;
}

typedef Fe<FpCfg> Fp;
typedef Fe<FrCfg> Fr;

// ---- Fp2 = Fp[u]/(u^2 + 1) ---------------------------------------------------------------------
struct Fp2 {
    Fp c0, c1;
};

// A uniform static interface so the curve code is written once for G1 (Fp) and G2 (Fp2).
// `Cold` names the multiplier policy to use in out-of-line / cold code: same field, products by call.
struct FpMulCold {
    typedef Fp T;
    static MASP_HD T mul(const T& a, const T& b) { return fe_mul_nc(a, b); }
    static MASP_HD T sqr(const T& a) { return fe_sqr_nc(a); }
};
struct FpOps {
    typedef Fp T;
    typedef FpMulCold Cold;
    typedef FpOps Base;                     // the ops of the stored element (see Fp2PairOps)
    static constexpr uint32_t LANES = 1;    // lanes that hold one element
    static constexpr uint32_t PARTS = 1;    // parts a STORED element is read in (one per lane of a group that shares it: see Fp2PairOps)
    static constexpr bool REPLICATED = false;   // see FpQuadOps
    static MASP_HD T zero() { return fe_zero<FpCfg>(); }
    static MASP_HD T one() { return fe_one<FpCfg>(); }
    static MASP_HD T add(const T& a, const T& b) { return fe_add(a, b); }
    static MASP_HD T sub(const T& a, const T& b) { return fe_sub(a, b); }
    static MASP_HD T neg(const T& a) { return fe_neg(a); }
    static MASP_HD T dbl(const T& a) { return fe_dbl(a); }
    static MASP_HD T mul(const T& a, const T& b) { return fe_mul(a, b); }
    static MASP_HD T mul_lazy(const T& a, const T& b) { return fe_mul_lazy(a, b); }  // result in [0, 2p): see fe_mul_lazy
    static MASP_HD T sqr(const T& a) { return fe_sqr(a); }
    static MASP_HD bool is_zero(const T& a) { return fe_is_zero(a); }
    static MASP_HD bool eq(const T& a, const T& b) { return fe_eq(a, b); }
    static MASP_HD T inv(const T& a) { return fe_inv(a); }
    static MASP_HD T inv_lone(const T& a) { return fe_inv_bingcd_nc(a); }  // for single-lane serial tails
    static MASP_HD T inv_gcd(const T& a) { return fe_inv_bingcd(a); }   // for a few thousand lanes each inverting one value
};
// FpOps for code written over O::LANES lanes per point, with FOUR lanes per G1 point: every lane of a quad holds the whole
// element (REPLICATED: loads read it four times, one lane stores it) and the point operations spread their independent
// products over the quad (device/quad.hpp).  Field operations on their own are FpOps's, done redundantly by the four lanes.
struct FpQuadOps : FpOps {
    static constexpr uint32_t LANES = 4;
    static constexpr bool REPLICATED = true;
};
struct Fp2Ops {
    typedef Fp2 T;
    typedef Fp2Ops Cold;  // already call-based
    typedef Fp2Ops Base;
    static constexpr uint32_t LANES = 1;
    static constexpr uint32_t PARTS = 1;
    static constexpr bool REPLICATED = false;   // see FpQuadOps
    static MASP_HD T zero() { return {fe_zero<FpCfg>(), fe_zero<FpCfg>()}; }
    static MASP_HD T one() { return {fe_one<FpCfg>(), fe_zero<FpCfg>()}; }
    static MASP_HD T add(const T& a, const T& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
    static MASP_HD T sub(const T& a, const T& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
    static MASP_HD T neg(const T& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
    static MASP_HD T dbl(const T& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
    // Karatsuba: 3 base-field products (out-of-line: keeps every G2 function small, see MASP_NOINLINE)
    static MASP_HD T mul(const T& a, const T& b) {
        Fp aa = fe_mul_nc(a.c0, b.c0), bb = fe_mul_nc(a.c1, b.c1);
        Fp cc = fe_mul_nc(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
        return {fe_sub(aa, bb), fe_sub(fe_sub(cc, aa), bb)};
    }
    static MASP_HD T mul_lazy(const T& a, const T& b) { return mul(a, b); }
    // (a0 + a1)(a0 - a1) + 2 a0 a1 u : 2 base-field products
    static MASP_HD T sqr(const T& a) {
        Fp s = fe_add(a.c0, a.c1), d = fe_sub(a.c0, a.c1);
        Fp m = fe_mul_nc(a.c0, a.c1);
        return {fe_mul_nc(s, d), fe_dbl(m)};
    }
    static MASP_HD bool is_zero(const T& a) { return fe_is_zero(a.c0) && fe_is_zero(a.c1); }
    static MASP_HD bool eq(const T& a, const T& b) { return fe_eq(a.c0, b.c0) && fe_eq(a.c1, b.c1); }
    static MASP_HD T inv(const T& a) {
        Fp n = fe_inv(fe_add(fe_mul_nc(a.c0, a.c0), fe_mul_nc(a.c1, a.c1)));
        return {fe_mul_nc(a.c0, n), fe_neg(fe_mul_nc(a.c1, n))};
    }
    static MASP_HD T inv_lone(const T& a) {
        Fp n = fe_inv_bingcd_nc(fe_add(fe_mul_nc(a.c0, a.c0), fe_mul_nc(a.c1, a.c1)));
        return {fe_mul_nc(a.c0, n), fe_neg(fe_mul_nc(a.c1, n))};
    }
    static MASP_HD T inv_gcd(const T& a) {
        Fp n = fe_inv_bingcd(fe_add(fe_mul_nc(a.c0, a.c0), fe_mul_nc(a.c1, a.c1)));
        return {fe_mul_nc(a.c0, n), fe_neg(fe_mul_nc(a.c1, n))};
    }
};

#if defined(__HIPCC__)
// Fp2 over a PAIR of adjacent lanes: the even lane holds c0, the odd lane c1, and the partner's half arrives through a DPP
// quad permutation (a register move, no LDS).  An element costs a lane 12 VGPRs instead of 24, so a kernel written over
// Fp2PairOps has the register footprint of its G1 twin (two waves per SIMD where the Fp2Ops form gets one), and the products
// cost what Karatsuba costs:  mul = one fused a b + z w per lane (3 N^2 multiply-adds x 2 lanes = 3 products' worth, one
// reduction each), sqr = one product per lane.  Both lanes of a pair must be active and take the same branches.
// A stored Fp2 (c0 | c1, 96 bytes) is reached as  reinterpret_cast<const Fp*>(ptr)[2 * index + half].
struct Fp2PairLanes {
    static __device__ __forceinline__ uint32_t half() { return threadIdx.x & 1u; }  // (one-dimensional workgroups)
    static __device__ __forceinline__ uint32_t swap32(uint32_t v) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
    }
    static __device__ __forceinline__ Fp partner(const Fp& a) {
        Fp r;
#pragma unroll
        for (int i = 0; i < 12; ++i) r.v[i] = swap32(a.v[i]);
        return r;
    }
    static __device__ __forceinline__ Fp pick(bool odd, const Fp& e, const Fp& o) {
        Fp r;
#pragma unroll
        for (int i = 0; i < 12; ++i) r.v[i] = odd ? o.v[i] : e.v[i];
        return r;
    }
    // even: a0 b0 - a1 b1 = a b + (-a') b';   odd: a0 b1 + a1 b0 = a' b + a b'     (' = the partner's half)
    static __device__ __forceinline__ Fp mul(const Fp& a, const Fp& b) {
        const bool odd = half();
        const Fp ap = partner(a), bp = partner(b);
        return fe_mul2(pick(odd, a, ap), b, pick(odd, fe_neg(ap), a), bp);
    }
    // even: (a0 + a1)(a0 - a1);   odd: 2 a0 a1
    static __device__ __forceinline__ Fp sqr(const Fp& a) {
        const bool odd = half();
        const Fp ap = partner(a);
        return fe_mul(pick(odd, fe_add(a, ap), fe_dbl(ap)), pick(odd, fe_sub(a, ap), a));
    }
};
// the same products out of line, operands in registers (see fp_mul_call): for the tails, where code size and registers count
__device__ __noinline__ FpRegs fp2pair_mul_call(u32x4 a0, u32x4 a1, u32x4 a2, u32x4 b0, u32x4 b1, u32x4 b2) {
    Fp a, b;
    a.v[0] = a0.x; a.v[1] = a0.y; a.v[2] = a0.z; a.v[3] = a0.w;
    a.v[4] = a1.x; a.v[5] = a1.y; a.v[6] = a1.z; a.v[7] = a1.w;
    a.v[8] = a2.x; a.v[9] = a2.y; a.v[10] = a2.z; a.v[11] = a2.w;
    b.v[0] = b0.x; b.v[1] = b0.y; b.v[2] = b0.z; b.v[3] = b0.w;
    b.v[4] = b1.x; b.v[5] = b1.y; b.v[6] = b1.z; b.v[7] = b1.w;
    b.v[8] = b2.x; b.v[9] = b2.y; b.v[10] = b2.z; b.v[11] = b2.w;
    const Fp r = Fp2PairLanes::mul(a, b);
    FpRegs o;
    o.q0 = u32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    o.q1 = u32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
    o.q2 = u32x4{r.v[8], r.v[9], r.v[10], r.v[11]};
    return o;
}
__device__ __noinline__ FpRegs fp2pair_sqr_call(u32x4 a0, u32x4 a1, u32x4 a2) {
    Fp a;
    a.v[0] = a0.x; a.v[1] = a0.y; a.v[2] = a0.z; a.v[3] = a0.w;
    a.v[4] = a1.x; a.v[5] = a1.y; a.v[6] = a1.z; a.v[7] = a1.w;
    a.v[8] = a2.x; a.v[9] = a2.y; a.v[10] = a2.z; a.v[11] = a2.w;
    const Fp r = Fp2PairLanes::sqr(a);
    FpRegs o;
    o.q0 = u32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
    o.q1 = u32x4{r.v[4], r.v[5], r.v[6], r.v[7]};
    o.q2 = u32x4{r.v[8], r.v[9], r.v[10], r.v[11]};
    return o;
}
struct Fp2PairCold {
    typedef Fp T;
    static __device__ __forceinline__ T mul(const T& a, const T& b) {
        FpRegs o = fp2pair_mul_call(u32x4{a.v[0], a.v[1], a.v[2], a.v[3]}, u32x4{a.v[4], a.v[5], a.v[6], a.v[7]}, u32x4{a.v[8], a.v[9], a.v[10], a.v[11]},
                                    u32x4{b.v[0], b.v[1], b.v[2], b.v[3]}, u32x4{b.v[4], b.v[5], b.v[6], b.v[7]}, u32x4{b.v[8], b.v[9], b.v[10], b.v[11]});
        T r;
        r.v[0] = o.q0.x; r.v[1] = o.q0.y; r.v[2] = o.q0.z; r.v[3] = o.q0.w;
        r.v[4] = o.q1.x; r.v[5] = o.q1.y; r.v[6] = o.q1.z; r.v[7] = o.q1.w;
        r.v[8] = o.q2.x; r.v[9] = o.q2.y; r.v[10] = o.q2.z; r.v[11] = o.q2.w;
        return r;
    }
    static __device__ __forceinline__ T sqr(const T& a) {
        FpRegs o = fp2pair_sqr_call(u32x4{a.v[0], a.v[1], a.v[2], a.v[3]}, u32x4{a.v[4], a.v[5], a.v[6], a.v[7]}, u32x4{a.v[8], a.v[9], a.v[10], a.v[11]});
        T r;
        r.v[0] = o.q0.x; r.v[1] = o.q0.y; r.v[2] = o.q0.z; r.v[3] = o.q0.w;
        r.v[4] = o.q1.x; r.v[5] = o.q1.y; r.v[6] = o.q1.z; r.v[7] = o.q1.w;
        r.v[8] = o.q2.x; r.v[9] = o.q2.y; r.v[10] = o.q2.z; r.v[11] = o.q2.w;
        return r;
    }
};
struct Fp2PairOps {
    typedef Fp T;
    typedef Fp2Ops Base;
    typedef Fp2PairCold Cold;
    static constexpr uint32_t LANES = 2;
    static constexpr uint32_t PARTS = 2;        // lane `half` reads half of every stored Fp2
    static constexpr bool REPLICATED = false;   // see FpQuadOps
    static __device__ __forceinline__ uint32_t half() { return Fp2PairLanes::half(); }
    static __device__ __forceinline__ T partner(const T& a) { return Fp2PairLanes::partner(a); }
    static __device__ __forceinline__ T zero() { return fe_zero<FpCfg>(); }
    static __device__ __forceinline__ T one() { return half() ? fe_zero<FpCfg>() : fe_one<FpCfg>(); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fe_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fe_sub(a, b); }
    static __device__ __forceinline__ T neg(const T& a) { return fe_neg(a); }
    static __device__ __forceinline__ T dbl(const T& a) { return fe_dbl(a); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return Fp2PairLanes::mul(a, b); }
    static __device__ __forceinline__ T mul_lazy(const T& a, const T& b) { return Fp2PairLanes::mul(a, b); }
    static __device__ __forceinline__ T sqr(const T& a) { return Fp2PairLanes::sqr(a); }
    static __device__ __forceinline__ bool both(bool mine) {
        const uint32_t f = mine ? 1u : 0u;
        return (f & Fp2PairLanes::swap32(f)) != 0u;
    }
    static __device__ __forceinline__ bool is_zero(const T& a) { return both(fe_is_zero(a)); }
    static __device__ __forceinline__ bool eq(const T& a, const T& b) { return both(fe_eq(a, b)); }
    // 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2): both lanes invert the same norm
    static __device__ __forceinline__ T inv_gcd(const T& a) {
        const T ap = partner(a);
        const T r = fe_mul(a, fe_inv_bingcd(fe_mul2(a, a, ap, ap)));
        return half() ? fe_neg(r) : r;
    }
};
// Fp2PairOps with FOUR pairs per G2 point (device/oct.hpp): eight lanes per point, every pair holds the whole element (its lanes one half
// each); the point operations spread their independent products over the four pairs.  Field operations on their own are Fp2PairOps's,
// done redundantly by the four pairs.  The bucket tails of a lone proof's b_g2 MSM.
struct Fp2OctOps : Fp2PairOps {
    static constexpr uint32_t LANES = 8;
    static constexpr bool REPLICATED = true;
};
#endif

}  // namespace masp
