// Groth16 parameter generation from explicit toxic waste, on the GPU.
//
// Mirrors bellperson's `groth16::generate_random_parameters` / `generate_parameters` (nam-bellperson
// 0.26.6-nam.1, un-vendored; called by the reference's benches at
// /root/reference/masp_proofs/benches/sapling.rs:24-36 and benches/convert.rs:19-29) with the semantics of
// QAP polynomials evaluated at tau through the Lagrange basis of the 2^k domain, queries
// h, l, ic, a, b_g1, b_g2 as fixed-base multiples of the standard generators, identities filtered out.
// The real MASP parameters come from an MPC and cannot be regenerated; this exists so that benches and
// tests have a CRS (no network).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "fr_io.hpp"
#include "io.hpp"

namespace masp {

// lag[k] = (Z(tau)/m) * w^k / (tau - w^k)      (Montgomery)
__global__ void k_setup_lagrange(Fr* __restrict__ lag, uint32_t nrows, Fr omega, Fr tau, Fr z_over_m) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrows) return;
    uint32_t e[1] = {k};
    Fr wk = fe_pow(omega, e, 1);
    Fr den = fe_inv(fe_sub(tau, wk));
    fr_store(lag + k, fe_mul(fe_mul(z_over_m, wk), den));
}
// per variable v: out[v] = sum over its column entries coef * lag[row]   (CSC; coef Montgomery)
__global__ void k_setup_qap(const uint32_t* __restrict__ colptr, const uint32_t* __restrict__ rowidx, const Fr* __restrict__ coef,
                            const Fr* __restrict__ lag, uint32_t nv, uint32_t n_inputs, uint32_t n_constraints, int is_a,
                            Fr* __restrict__ out) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    Fr acc = fe_zero<FrCfg>();
    for (uint32_t t = colptr[v]; t < colptr[v + 1]; ++t) acc = fe_add(acc, fe_mul(fr_load(coef + t), fr_load(lag + rowidx[t])));
    if (is_a && v < n_inputs) acc = fe_add(acc, fr_load(lag + n_constraints + v));  // the extra Input(i) * 0 = 0 rows
    fr_store(out + v, acc);
}
// k[v] = (beta * at[v] + alpha * bt[v] + ct[v]) * scale
__global__ void k_setup_lc(const Fr* __restrict__ at, const Fr* __restrict__ bt, const Fr* __restrict__ ct, uint32_t n, Fr alpha, Fr beta,
                           Fr scale, Fr* __restrict__ out) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    Fr x = fe_add(fe_add(fe_mul(beta, fr_load(at + v)), fe_mul(alpha, fr_load(bt + v))), fr_load(ct + v));
    fr_store(out + v, fe_mul(x, scale));
}
// flags[v] = value != 0
__global__ void k_setup_nonzero(const Fr* __restrict__ x, uint32_t n, uint8_t* __restrict__ flags) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    flags[v] = fe_is_zero(fr_load(x + v)) ? 0 : 1;
}

// Fixed-base tables: tab[w*255 + d-1] = d * 2^(8w) * G  (affine), w < 32.  One lane per window.
template <class O>
__global__ void k_setup_fixed_table(Affine<O> gen, Affine<O>* __restrict__ tab) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= 32) return;
    Xyzz<O> base = xyzz_from_affine(gen);
    for (uint32_t k = 0; k < 8 * w; ++k) base = xyzz_dbl(base);
    Affine<O> b = xyzz_to_affine(base);
    Xyzz<O> cur = xyzz_from_affine(b);
    for (uint32_t d = 1; d <= 255; ++d) {
        tab[w * 255 + d - 1] = xyzz_to_affine(cur);
        xyzz_madd_nc(cur, b, false);
    }
}
// out[i] = [k_i] G as uncompressed bytes; scalars Montgomery (mont != 0) or canonical
template <class O, int BYTES>
__global__ void __launch_bounds__(64) k_setup_fixed_mul(const Affine<O>* __restrict__ tab, const Fr* __restrict__ scalars, uint32_t n, int mont,
                                                        uint8_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = fr_load(scalars + i);
    if (mont) k = fe_from_mont(k);
    Xyzz<O> acc = xyzz_inf<O>();
    for (int w = 0; w < 32; ++w) {
        uint32_t d = (k.v[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) xyzz_madd_nc(acc, tab[w * 255 + d - 1], false);
    }
    Affine<O> p = xyzz_to_affine(acc);
    if constexpr (BYTES == 96)
        g1_write_uncompressed(p, out + (size_t)i * 96);
    else
        g2_write_uncompressed(p, out + (size_t)i * 192);
}
// dst[k] = src[idx[k]]
__global__ void k_setup_gather(const Fr* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, Fr* __restrict__ dst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    fr_store(dst + k, fr_load(src + idx[k]));
}
// h scalars: out[i] = tau^i * c    (c = Z(tau)/delta)
__global__ void k_setup_h_scalars(Fr tau, Fr c, uint32_t n, Fr* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e[1] = {i};
    fr_store(out + i, fe_mul(fe_pow(tau, e, 1), c));
}

}  // namespace masp
