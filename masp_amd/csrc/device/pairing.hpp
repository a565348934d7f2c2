// Device half of the GPU Groth16 batch verifier (bellman `verify_proofs_batch`, reached from
// /root/reference/masp_proofs/src/sapling/verifier/batch.rs:24-31,201-239):
//   k_verify_prepare   one lane per proof: decompress A, C (G1) and B (G2) from the 192 proof bytes (square roots in Fp /
//                      Fp2, zcash sign convention), z_i * A_i in affine form and z_i * C_i for the random 128-bit z_i
//   k_miller_pairs     one WAVE per pair (z_i A_i, B_i): the ate Miller loop as an interpreter of the levelled straight-line
//                      programs built on the host (host/pairing_prog.h): values in LDS slots, lane k executes operation k
//                      of the current step, so the ~135 dependent 384-bit products of an iteration become 3-4 product steps
//   k_fp12_product     the product of the pairs' Miller values (same interpreter, Fp12 multiplication program)
// The public-input combination, the two remaining pairs and the final exponentiation stay on the host (host/pairing.h).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "subgroup.hpp"
#include "io.hpp"

namespace masp {

// ---- square roots / decompression ---------------------------------------------------------------------------
// p = 3 mod 4: a^((p+1)/4) is a square root of a if one exists
__device__ inline bool fp_sqrt(const Fp& a, Fp& out) {
    uint32_t e[12];
    uint64_t carry = 1;  // (p + 1) / 4 = (p >> 2) + 1 because p = 3 mod 4
    for (int i = 0; i < 12; ++i) {
        uint32_t w = (FpCfg::MOD[i] >> 2) | (i < 11 ? FpCfg::MOD[i + 1] << 30 : 0u);
        carry += w;
        e[i] = (uint32_t)carry;
        carry >>= 32;
    }
    Fp r = fe_pow(a, e, 12);
    out = r;
    return fe_eq(fe_sqr_nc(r), a);
}
__device__ inline bool fp2_sqrt(const Fp2& v, Fp2& out) {  // host/pairing.h Fp2::sqrt
    if (Fp2Ops::is_zero(v)) {
        out = v;
        return true;
    }
    if (fe_is_zero(v.c1)) {
        Fp s;
        if (fp_sqrt(v.c0, s)) {
            out = {s, fe_zero<FpCfg>()};
            return true;
        }
        if (fp_sqrt(fe_neg(v.c0), s)) {
            out = {fe_zero<FpCfg>(), s};
            return true;
        }
        return false;
    }
    Fp n;
    if (!fp_sqrt(fe_add(fe_sqr_nc(v.c0), fe_sqr_nc(v.c1)), n)) return false;
    Fp two = fe_dbl(fe_one<FpCfg>());
    Fp half = fe_inv_fermat(two);
    Fp d = fe_mul_nc(fe_add(v.c0, n), half), x0;
    if (!fp_sqrt(d, x0)) {
        d = fe_mul_nc(fe_sub(v.c0, n), half);
        if (!fp_sqrt(d, x0)) return false;
    }
    Fp x1 = fe_mul_nc(v.c1, fe_inv_fermat(fe_dbl(x0)));
    out = {x0, x1};
    return Fp2Ops::eq(Fp2Ops::sqr(out), v);
}
// subgroup membership of G1 / G2 points (g1_in_subgroup, g2_in_subgroup): device/subgroup.hpp
// the encoding of the point at infinity: compression and infinity flags, nothing else (bellman rejects stray bits)
__device__ inline bool infinity_encoding_is_clean(const uint8_t* in, int len) {
    if ((in[0] & 0x3f) != 0) return false;
    uint32_t acc = 0;
    for (int i = 1; i < len; ++i) acc |= in[i];
    return acc == 0;
}

// zcash compressed encodings -> affine (Montgomery).  PT_* status; PT_BAD_FLAGS also for "not on the curve"
__device__ inline int g1_read_compressed(const uint8_t* in, G1Affine& p) {
    if (!(in[0] & 0x80)) return PT_BAD_FLAGS;
    if (in[0] & 0x40) {
        p.x = fe_zero<FpCfg>();
        p.y = fe_zero<FpCfg>();
        return infinity_encoding_is_clean(in, 48) ? PT_INFINITY : PT_BAD_FLAGS;
    }
    uint8_t t[48];
    for (int i = 0; i < 48; ++i) t[i] = in[i];
    const bool big = t[0] & 0x20;
    t[0] &= 0x1f;
    Fp xc = fe_load_be<FpCfg>(t);
    if (fe_canonical_ge_mod(xc)) return PT_NOT_CANONICAL;
    p.x = fe_to_mont(xc);
    Fp four = fe_dbl(fe_dbl(fe_one<FpCfg>()));
    Fp rhs = fe_add(fe_mul_nc(fe_sqr_nc(p.x), p.x), four);
    if (!fp_sqrt(rhs, p.y)) return PT_BAD_FLAGS;
    if (fe_canonical_gt_half(fe_from_mont(p.y)) != big) p.y = fe_neg(p.y);
    return PT_OK;
}
__device__ inline int g2_read_compressed(const uint8_t* in, G2Affine& p) {
    if (!(in[0] & 0x80)) return PT_BAD_FLAGS;
    if (in[0] & 0x40) {
        p.x = Fp2Ops::zero();
        p.y = Fp2Ops::zero();
        return infinity_encoding_is_clean(in, 96) ? PT_INFINITY : PT_BAD_FLAGS;
    }
    uint8_t t[96];
    for (int i = 0; i < 96; ++i) t[i] = in[i];
    const bool big = t[0] & 0x20;
    t[0] &= 0x1f;
    Fp x1 = fe_load_be<FpCfg>(t), x0 = fe_load_be<FpCfg>(t + 48);
    if (fe_canonical_ge_mod(x0) || fe_canonical_ge_mod(x1)) return PT_NOT_CANONICAL;
    p.x = {fe_to_mont(x0), fe_to_mont(x1)};
    Fp four = fe_dbl(fe_dbl(fe_one<FpCfg>()));
    Fp2 rhs = Fp2Ops::add(Fp2Ops::mul(Fp2Ops::sqr(p.x), p.x), Fp2{four, four});  // y^2 = x^3 + 4 (1 + u)
    if (!fp2_sqrt(rhs, p.y)) return PT_BAD_FLAGS;
    const Fp y1 = fe_from_mont(p.y.c1);
    const bool lg = fe_is_zero(y1) ? fe_canonical_gt_half(fe_from_mont(p.y.c0)) : fe_canonical_gt_half(y1);
    if (lg != big) p.y = Fp2Ops::neg(p.y);
    return PT_OK;
}

// ---- stage 1: decode + randomise ---------------------------------------------------------------------------------
// proofs: n x 192 B; z: n x 16 B (little-endian, bit 0 forced to 1 like the host verifier).  Outputs per proof: za = z A
// (affine), b = B (affine), zc = z C (XYZZ), status = PT_* bits of the three points (the host refuses every one of them, PT_INFINITY included: k_verify.hip).
__global__ void __launch_bounds__(64) k_verify_prepare(const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ z, uint32_t n,
                                                       G1Affine* __restrict__ za, G2Affine* __restrict__ b, G1Xyzz* __restrict__ zc,
                                                       int* __restrict__ status) {
    // gridDim.y = 5: y = 0 -> z A, 1 -> z C, 2 -> B and its subgroup test, 3 / 4 -> the subgroup tests of A / C (every wave does
    // one kind of work; B's square root + 64-bit multiple is the longest chain, the tests of A and C run next to their multiples)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, what = blockIdx.y;
    if (i >= n) return;
    const uint8_t* pr = proofs + 192 * (size_t)i;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 4; ++w)
        k[w] = (uint32_t)z[16 * i + 4 * w] | ((uint32_t)z[16 * i + 4 * w + 1] << 8) | ((uint32_t)z[16 * i + 4 * w + 2] << 16) |
               ((uint32_t)z[16 * i + 4 * w + 3] << 24);
    k[0] |= 1u;
    int st;
    if (what == 2) {
        G2Affine q;
        st = g2_read_compressed(pr + 48, q);
        if (st == PT_OK && !g2_in_subgroup(q)) st = PT_NOT_IN_SUBGROUP;
        b[i] = q;
    } else if (what >= 3) {
        G1Affine p;
        st = g1_read_compressed(pr + (what == 3 ? 0 : 144), p);
        st = st == PT_OK && !g1_in_subgroup(p) ? PT_NOT_IN_SUBGROUP : 0;   // (decoding errors are reported by the lanes of y = 0 / 1)
    } else {
        G1Affine p;
        st = g1_read_compressed(pr + (what == 0 ? 0 : 144), p);
        G1Xyzz m = (st & ~PT_INFINITY) ? xyzz_inf<FpOps>() : xyzz_mul_scalar(xyzz_from_affine(p), k);
        if (what == 0)
            za[i] = xyzz_to_affine<FpOps, true>(m);
        else
            zc[i] = m;
    }
    if (st) atomicOr(status + i, st);
}

// sum of the n points zc[] (one workgroup of 256 lanes: strided serial sums, then an LDS tree), written as an uncompressed
// affine point (96 bytes, bellman wire format) for the host
__global__ void __launch_bounds__(256) k_g1_sum_export(const G1Xyzz* __restrict__ zc, uint32_t n, uint8_t* __restrict__ out96) {
    __shared__ G1Xyzz sh[256];
    const uint32_t tid = threadIdx.x;
    G1Xyzz acc = xyzz_inf<FpOps>();
    for (uint32_t k = tid; k < n; k += 256) xyzz_add_nc(acc, zc[k]);
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        sh[tid] = acc;
        __syncthreads();
        if (tid < d) xyzz_add_nc(acc, sh[tid + d]);
        __syncthreads();
    }
    if (tid == 0) g1_write_uncompressed(xyzz_to_affine<FpOps, true>(acc), out96);
}

// ---- stage 2: Miller loops, one wave per pair -------------------------------------------------------------------------
// program word: op | dst << 2 | a << 12 | b << 22 ; steps[s] .. steps[s + 1] = the mutually independent operations of step s
struct PairingProgramDev {
    const uint32_t* ops;
    const uint32_t* steps;
    uint32_t n_steps;
};
__device__ __forceinline__ Fp slot_load(const uint32_t* sl, uint32_t s) {
    const uint4* q = reinterpret_cast<const uint4*>(sl + 12 * s);
    uint4 a = q[0], b = q[1], c = q[2];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w;
    return r;
}
__device__ __forceinline__ void slot_store(uint32_t* sl, uint32_t s, const Fp& r) {
    uint4* q = reinterpret_cast<uint4*>(sl + 12 * s);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
    q[2] = make_uint4(r.v[8], r.v[9], r.v[10], r.v[11]);
}
// the wave runs one program over its slots
__device__ __forceinline__ void run_program(const PairingProgramDev& P, uint32_t* sl, uint32_t lane) {
    uint32_t lo = P.steps[0];
    for (uint32_t s = 0; s < P.n_steps; ++s) {
        const uint32_t hi = P.steps[s + 1];
        for (uint32_t k = lo + lane; k < hi; k += 64) {
            const uint32_t w = P.ops[k], op = w & 3u, dst = (w >> 2) & 1023u, a = (w >> 12) & 1023u, b = (w >> 22) & 1023u;
            const Fp x = slot_load(sl, a);
            Fp r;
            if (op == 0u)
                r = fe_mul(x, slot_load(sl, b));
            else if (op == 1u)
                r = fe_add(x, slot_load(sl, b));
            else if (op == 2u)
                r = fe_sub(x, slot_load(sl, b));
            else
                r = x;
            slot_store(sl, dst, r);
        }
        // one wave: LDS operations execute in order; only the compiler must not move them across the step boundary
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lo = hi;
    }
}
// slot numbers: host/pairing_prog.h (SLOT_ZERO 0, SLOT_F 1..12, SLOT_T 13..18, SLOT_P 19..20, SLOT_Q 21..24)
__global__ void __launch_bounds__(64) k_miller_pairs(PairingProgramDev dbl, PairingProgramDev add, uint32_t n_slots, const G1Affine* __restrict__ Pp,
                                                     const G2Affine* __restrict__ Qp, Fp* __restrict__ out) {
    extern __shared__ uint4 pairing_lds[];
    uint32_t* sl = reinterpret_cast<uint32_t*>(pairing_lds);
    const uint32_t lane = threadIdx.x, pair = blockIdx.x;
    const G1Affine P = Pp[pair];
    const G2Affine Q = Qp[pair];
    Fp* o = out + 12 * (size_t)pair;
    const Fp one = fe_one<FpCfg>(), zero = fe_zero<FpCfg>();
    if (aff_is_inf(P) || aff_is_inf(Q)) {  // a pair with a point at infinity contributes 1
        if (lane < 12) o[lane] = lane == 0 ? one : zero;
        return;
    }
    for (uint32_t s = lane; s < n_slots; s += 64) slot_store(sl, s, zero);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        slot_store(sl, 1, one);  // f = 1
        slot_store(sl, 13, Q.x.c0); slot_store(sl, 14, Q.x.c1); slot_store(sl, 15, Q.y.c0); slot_store(sl, 16, Q.y.c1); slot_store(sl, 17, one);  // T = Q
        slot_store(sl, 19, P.x); slot_store(sl, 20, P.y);
        slot_store(sl, 21, Q.x.c0); slot_store(sl, 22, Q.x.c1); slot_store(sl, 23, Q.y.c0); slot_store(sl, 24, Q.y.c1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint64_t xabs = 0xd201000000010000ull;  // |x| of BLS12-381; the value is conjugated at the end because x < 0
    for (int bit = 62; bit >= 0; --bit) {
        run_program(dbl, sl, lane);  // f <- f^2 l_{T,T}(P), T <- 2T   (squaring f = 1 in the first round is harmless)
        if ((xabs >> bit) & 1) run_program(add, sl, lane);
    }
    if (lane < 12) {
        Fp v = slot_load(sl, 1 + lane);
        o[lane] = lane >= 6 ? fe_neg(v) : v;  // conj: the w-part negated
    }
}

// ---- stage 3: product of the Miller values ---------------------------------------------------------------------------------
// wave w multiplies vals[w], vals[w + stride], vals[w + 2 stride], ... (12 Fp each) into vals[w]
__global__ void __launch_bounds__(64) k_fp12_product(PairingProgramDev mul12, uint32_t n_slots, Fp* __restrict__ vals, uint32_t n, uint32_t stride) {
    extern __shared__ uint4 pairing_lds[];
    uint32_t* sl = reinterpret_cast<uint32_t*>(pairing_lds);
    const uint32_t lane = threadIdx.x, w = blockIdx.x;
    if (w >= n) return;
    const Fp zero = fe_zero<FpCfg>();
    for (uint32_t s = lane; s < n_slots; s += 64) slot_store(sl, s, zero);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 12) slot_store(sl, 1 + lane, vals[12 * (size_t)w + lane]);
    for (uint32_t k = w + stride; k < n; k += stride) {
        if (lane < 12) slot_store(sl, 13 + lane, vals[12 * (size_t)k + lane]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        run_program(mul12, sl, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 12) vals[12 * (size_t)w + lane] = slot_load(sl, 1 + lane);
}

}  // namespace masp
