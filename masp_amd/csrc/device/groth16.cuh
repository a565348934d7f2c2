// Groth16-specific device kernels around the NTT and MSM engines: witness -> (a, b, c) evaluation
// vectors from the static R1CS, query-scalar gathering by density, and the final proof assembly +
// zcash encoding.  Restates, for the GPU, bellperson's `ProvingAssignment::enforce` evaluation and
// `create_proof` tail (nam-bellperson 0.26.6-nam.1, un-vendored) and
// `Proof::write` (/root/reference/masp_proofs/src/prover.rs:190-193).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.cuh"
#include "io.cuh"
#include "ntt.cuh"

namespace masp {

// canonical -> Montgomery, n elements; flags any value >= r
__global__ void k_fr_to_mont(const Fr* __restrict__ x, size_t x_stride, Fr* __restrict__ y, uint32_t n, int* __restrict__ range_err) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    x += blockIdx.y * x_stride;
    y += (size_t)blockIdx.y * n;
    Fr v = fr_load(x + k);
    if (fe_canonical_ge_mod(v)) atomicOr(range_err, 1);
    fr_store(y + k, fe_to_mont(v));
}

// One CSR row per lane: out[row] = sum_t coef[t] * w[col[t]]  (all Montgomery).  Rows
// n_constraints .. n_constraints + n_inputs - 1 are bellperson's extra "Input(i) * 0 = 0" rows:
// a = input value, b = c = 0 (which == 0 selects matrix A).
// Row lengths of the MASP circuits range from 1 to several hundred terms (bit packings): `order` lists the constraint
// rows by decreasing length, so the 64 rows of a wave take about equally long.
__global__ void k_r1cs_eval(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ order, const uint32_t* __restrict__ col,
                            const Fr* __restrict__ coef, const Fr* __restrict__ w, uint32_t n_vars, uint32_t n_constraints, uint32_t n_inputs,
                            int which, Fr* __restrict__ out) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_constraints + n_inputs) return;
    w += (size_t)blockIdx.y * n_vars;
    out += (size_t)blockIdx.y * (n_constraints + n_inputs);
    Fr acc = fe_zero<FrCfg>();
    if (row < n_constraints) {
        row = order[row];
        uint32_t lo = rowptr[row], hi = rowptr[row + 1];
        for (uint32_t t = lo; t < hi; ++t) acc = fe_add(acc, fe_mul(fr_load(coef + t), fr_load(w + col[t])));
    } else if (which == 0) {
        acc = fr_load(w + (row - n_constraints));
    }
    fr_store(out + row, acc);
}

// dst[k] = src[idx[k]]  (32-byte scalars)
__global__ void k_gather_scalars(const Fr* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ idx, uint32_t n,
                                 Fr* __restrict__ dst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    src += blockIdx.y * src_stride;
    dst += (size_t)blockIdx.y * n;
    fr_store(dst + k, fr_load(src + idx[k]));
}

struct VkDevice {
    G1Affine alpha_g1, beta_g1, delta_g1;
    G2Affine beta_g2, delta_g2;
};

// Fixed-base tables for the per-circuit points that get multiplied by r, s, rs in every proof:
// tab[w*255 + d-1] = d * 2^(8w) * P in XYZZ form (no inversions to build), w < 32.  One lane per window,
// blockIdx.x selects the point.
template <class O>
__global__ void __launch_bounds__(64) k_fixed_table_xyzz(const Affine<O>* __restrict__ pts, Xyzz<O>* __restrict__ tabs) {
    const uint32_t w = threadIdx.x;
    if (w >= 32) return;
    Xyzz<O> base = xyzz_from_affine(pts[blockIdx.x]);
    for (uint32_t k = 0; k < 8 * w; ++k) base = xyzz_dbl(base);
    Xyzz<O>* tab = tabs + (size_t)blockIdx.x * 32 * 255 + w * 255;
    Xyzz<O> cur = base;
    for (uint32_t d = 1; d <= 255; ++d) {
        tab[d - 1] = cur;
        xyzz_add_nc(cur, base);
    }
}
template <class O>
__device__ __forceinline__ Xyzz<O> xyzz_fixed_mul(const Xyzz<O>* __restrict__ tab, const Fr& k) {
    Xyzz<O> acc = xyzz_inf<O>();
    for (int w = 0; w < 32; ++w) {
        uint32_t d = (k.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) xyzz_add_nc(acc, tab[w * 255 + d - 1]);
    }
    return acc;
}

// Proof assembly:
//   g_a = r*delta1 + alpha1 + A
//   g_b = s*delta2 + beta2 + B2
//   g_c = (r s)*delta1 + s*alpha1 + r*beta1 + s*A + r*B1 + H + L
// One workgroup of three waves so that the three differently-shaped jobs do not serialise inside a wave:
//   wave 0             : the two variable-base multiplications  s*A, r*B1: lanes 0..31 build the tables d*A, d*B1
//                        (d < 16) in LDS, then lanes 0..1 run 4-bit fixed windows (252 doublings + <= 64 additions each;
//                        exact for any curve point: no endomorphism, the CRS is read unchecked like the reference's)
//   wave 1, lane 0     : s*delta2 on G2 through its fixed-base table          (<= 32 additions)
//   wave 2, lanes 0..3 : r*delta1, s*alpha1, r*beta1, (r s)*delta1 through fixed-base tables
// then three lanes normalise and encode.  rs: 8 limbs r | 8 limbs s (canonical).
// fb1: tables of delta1, alpha1, beta1 (in that order); fb2: table of delta2.
__global__ void __launch_bounds__(192) k_groth16_assemble(const VkDevice* __restrict__ vk, const G1Xyzz* __restrict__ fb1,
                                                          const G2Xyzz* __restrict__ fb2, const G1Xyzz* __restrict__ msm_g1 /* H, L, A, B1 */,
                                                          const G2Xyzz* __restrict__ msm_g2, const uint32_t* __restrict__ rs, size_t rs_stride,
                                                          uint8_t* __restrict__ proof) {
    __shared__ G1Xyzz part[6];
    __shared__ G2Xyzz part2;
    __shared__ G1Xyzz wtab[2][16];
    const uint32_t tid = threadIdx.x;
    msm_g1 += (size_t)blockIdx.x * 4;  // one workgroup per proof of the batch
    msm_g2 += blockIdx.x;
    rs += (size_t)blockIdx.x * rs_stride;
    proof += (size_t)blockIdx.x * 192;
    Fr r, s;
    for (int i = 0; i < 8; ++i) {
        r.v[i] = rs[i];
        s.v[i] = rs[8 + i];
    }
    constexpr size_t TAB = 32 * 255;
    if (tid < 64) {
        if (tid < 32) {  // wtab[j][d] = d * P_j by double-and-add over the 4 bits of d
            const uint32_t j = tid >> 4, d = tid & 15;
            const G1Xyzz P = msm_g1[2 + j];
            G1Xyzz t = xyzz_inf<FpOps>();
            for (int b = 3; b >= 0; --b) {
                t = xyzz_dbl(t);
                if ((d >> b) & 1) xyzz_add_nc(t, P);
            }
            wtab[j][d] = t;
        }
        // same wave writes and reads the table: LDS is in order per wave, only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < 2) {
            const uint32_t* k = tid == 0 ? s.v : r.v;  // s*A, r*B1
            G1Xyzz acc = xyzz_inf<FpOps>();
            for (int w = 63; w >= 0; --w) {
                if (w != 63)
                    for (int q = 0; q < 4; ++q) acc = xyzz_dbl(acc);
                const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 15u;
                if (d) xyzz_add_nc(acc, wtab[tid][d]);
            }
            part[3 + tid] = acc;
        }
    } else if (tid == 64) {
        part2 = xyzz_fixed_mul<Fp2Ops>(fb2, s);
    } else if (tid >= 128 && tid < 132) {
        const uint32_t j = tid - 128;
        Fr rs_prod = fe_mul(fe_to_mont(r), s);  // mont(r) * s = r*s mod q, canonical
        const G1Xyzz* tab = fb1 + (j == 1 ? TAB : j == 2 ? 2 * TAB : 0);
        Fr k = j == 0 ? r : j == 1 ? s : j == 2 ? r : rs_prod;
        G1Xyzz v = xyzz_fixed_mul<FpOps>(tab, k);
        part[j == 3 ? 5 : j] = v;  // 0: r*delta1, 1: s*alpha1, 2: r*beta1, 5: (r s)*delta1
    }
    __syncthreads();
    if (tid == 0) {
        G1Xyzz ga = part[0];
        xyzz_madd_nc(ga, vk->alpha_g1, false);
        xyzz_add_nc(ga, msm_g1[2]);
        g1_write_compressed(xyzz_to_affine(ga), proof);
    } else if (tid == 128) {
        G1Xyzz gc = part[5];
        xyzz_add_nc(gc, part[1]);
        xyzz_add_nc(gc, part[2]);
        xyzz_add_nc(gc, part[3]);
        xyzz_add_nc(gc, part[4]);
        xyzz_add_nc(gc, msm_g1[0]);
        xyzz_add_nc(gc, msm_g1[1]);
        g1_write_compressed(xyzz_to_affine(gc), proof + 144);
    } else if (tid == 64) {
        G2Xyzz gb = part2;
        xyzz_madd_nc(gb, vk->beta_g2, false);
        xyzz_add_nc(gb, *msm_g2);
        g2_write_compressed(xyzz_to_affine(gb), proof + 48);
    }
}

// single point XYZZ -> uncompressed bytes (building-block entry points)
__global__ void k_g1_export(const G1Xyzz* __restrict__ p, uint8_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g1_write_uncompressed(xyzz_to_affine(*p), out);
}
__global__ void k_g2_export(const G2Xyzz* __restrict__ p, uint8_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) g2_write_uncompressed(xyzz_to_affine(*p), out);
}
__global__ void k_g1_import_one(const uint8_t* __restrict__ raw, G1Affine* __restrict__ out, int* __restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int st = g1_read_uncompressed(raw, *out);
        if (st) atomicOr(status, st);
    }
}
__global__ void k_g2_import_one(const uint8_t* __restrict__ raw, G2Affine* __restrict__ out, int* __restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int st = g2_read_uncompressed(raw, *out);
        if (st) atomicOr(status, st);
    }
}

}  // namespace masp
