// zcash / bellman wire encodings <-> device Montgomery limbs.
//   * uncompressed points as found in a `Parameters` file (read at /root/reference/masp_proofs/src/lib.rs:336-341,
//     unchecked decode) — big-endian coordinates, flag bits 0x80 compressed / 0x40 infinity / 0x20 sign;
//   * compressed points as written by `Proof::write` (/root/reference/masp_proofs/src/prover.rs:190-193);
//   * Fr as 32-byte little-endian canonical (`to_repr()`).
#pragma once
#include "curve.hpp"

namespace masp {

// 4*N big-endian bytes -> canonical limbs (little-endian limb order)
template <class C>
MASP_HD Fe<C> fe_load_be(const uint8_t* in) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        const uint8_t* b = in + 4 * (C::N - 1 - i);
        r.v[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
    return r;
}
template <class C>
MASP_HD void fe_store_be(const Fe<C>& canon, uint8_t* out) {
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        uint8_t* b = out + 4 * (C::N - 1 - i);
        uint32_t w = canon.v[i];
        b[0] = (uint8_t)(w >> 24);
        b[1] = (uint8_t)(w >> 16);
        b[2] = (uint8_t)(w >> 8);
        b[3] = (uint8_t)w;
    }
}
template <class C>
MASP_HD Fe<C> fe_load_le(const uint8_t* in) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        const uint8_t* b = in + 4 * i;
        r.v[i] = ((uint32_t)b[3] << 24) | ((uint32_t)b[2] << 16) | ((uint32_t)b[1] << 8) | b[0];
    }
    return r;
}
template <class C>
MASP_HD void fe_store_le(const Fe<C>& canon, uint8_t* out) {
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
        uint32_t w = canon.v[i];
        out[4 * i] = (uint8_t)w;
        out[4 * i + 1] = (uint8_t)(w >> 8);
        out[4 * i + 2] = (uint8_t)(w >> 16);
        out[4 * i + 3] = (uint8_t)(w >> 24);
    }
}

// status bits returned by the point readers
enum : int { PT_OK = 0, PT_BAD_FLAGS = 1, PT_NOT_CANONICAL = 2, PT_INFINITY = 4, PT_NOT_IN_SUBGROUP = 8 };

__host__ __device__ inline int g1_read_uncompressed(const uint8_t* in, G1Affine& p) {
    if (in[0] & 0x80) return PT_BAD_FLAGS;
    if (in[0] & 0x40) {
        p.x = fe_zero<FpCfg>();
        p.y = fe_zero<FpCfg>();
        return PT_INFINITY;
    }
    Fp x = fe_load_be<FpCfg>(in), y = fe_load_be<FpCfg>(in + 48);
    if (fe_canonical_ge_mod(x) || fe_canonical_ge_mod(y)) return PT_NOT_CANONICAL;
    p.x = fe_to_mont(x);
    p.y = fe_to_mont(y);
    return PT_OK;
}
__host__ __device__ inline int g2_read_uncompressed(const uint8_t* in, G2Affine& p) {
    if (in[0] & 0x80) return PT_BAD_FLAGS;
    if (in[0] & 0x40) {
        p.x = Fp2Ops::zero();
        p.y = Fp2Ops::zero();
        return PT_INFINITY;
    }
    Fp xc1 = fe_load_be<FpCfg>(in), xc0 = fe_load_be<FpCfg>(in + 48);
    Fp yc1 = fe_load_be<FpCfg>(in + 96), yc0 = fe_load_be<FpCfg>(in + 144);
    if (fe_canonical_ge_mod(xc0) || fe_canonical_ge_mod(xc1) || fe_canonical_ge_mod(yc0) || fe_canonical_ge_mod(yc1))
        return PT_NOT_CANONICAL;
    p.x.c0 = fe_to_mont(xc0);
    p.x.c1 = fe_to_mont(xc1);
    p.y.c0 = fe_to_mont(yc0);
    p.y.c1 = fe_to_mont(yc1);
    return PT_OK;
}
__host__ __device__ inline void g1_write_uncompressed(const G1Affine& p, uint8_t* out) {
    if (aff_is_inf(p)) {
        for (int i = 0; i < 96; ++i) out[i] = 0;
        out[0] = 0x40;
        return;
    }
    fe_store_be(fe_from_mont(p.x), out);
    fe_store_be(fe_from_mont(p.y), out + 48);
}
__host__ __device__ inline void g2_write_uncompressed(const G2Affine& p, uint8_t* out) {
    if (aff_is_inf(p)) {
        for (int i = 0; i < 192; ++i) out[i] = 0;
        out[0] = 0x40;
        return;
    }
    fe_store_be(fe_from_mont(p.x.c1), out);
    fe_store_be(fe_from_mont(p.x.c0), out + 48);
    fe_store_be(fe_from_mont(p.y.c1), out + 96);
    fe_store_be(fe_from_mont(p.y.c0), out + 144);
}
__host__ __device__ inline void g1_write_compressed(const G1Affine& p, uint8_t* out) {
    if (aff_is_inf(p)) {
        for (int i = 0; i < 48; ++i) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fe_store_be(fe_from_mont(p.x), out);
    out[0] |= 0x80;
    if (fe_canonical_gt_half(fe_from_mont(p.y))) out[0] |= 0x20;
}
__host__ __device__ inline void g2_write_compressed(const G2Affine& p, uint8_t* out) {
    if (aff_is_inf(p)) {
        for (int i = 0; i < 96; ++i) out[i] = 0;
        out[0] = 0xc0;
        return;
    }
    fe_store_be(fe_from_mont(p.x.c1), out);
    fe_store_be(fe_from_mont(p.x.c0), out + 48);
    out[0] |= 0x80;
    Fp y1 = fe_from_mont(p.y.c1);
    bool big = fe_is_zero(y1) ? fe_canonical_gt_half(fe_from_mont(p.y.c0)) : fe_canonical_gt_half(y1);
    if (big) out[0] |= 0x20;
}

}  // namespace masp
