// Witness -> (a, b, c) evaluation vectors from the static R1CS, range check and query-scalar gathering by density.
// Restates, for the GPU, bellperson's `ProvingAssignment::enforce` evaluation (nam-bellperson 0.26.6-nam.1, un-vendored;
// SURVEY.md A.3 step 2).
#pragma once
#include <hip/hip_runtime.h>

#include "../launch.h"
#include "fr_io.hpp"

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
// The same for an assignment whose elements from index `mont_from` on arrived as Montgomery residues (masp_hip_job::aux_form):
// those are range-checked as they are, copied to y, and REPLACED in x by their canonical value (the MSMs read canonical scalars).
__global__ void k_fr_split_forms(Fr* __restrict__ x, size_t x_stride, Fr* __restrict__ y, uint32_t n, uint32_t mont_from, int* __restrict__ range_err) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    x += blockIdx.y * x_stride;
    y += (size_t)blockIdx.y * n;
    Fr v = fr_load(x + k);
    if (fe_canonical_ge_mod(v)) atomicOr(range_err, 1);
    if (k < mont_from) {
        fr_store(y + k, fe_to_mont(v));
    } else {
        fr_store(y + k, v);
        fr_store(x + k, fe_from_mont(v));
    }
}

// out[row] = sum_t coef[t] * w[col[t]]  (all Montgomery), one CSR row per lane — except the long rows.  Rows
// n_constraints .. n_constraints + n_inputs - 1 are bellperson's extra "Input(i) * 0 = 0" rows:
// a = input value, b = c = 0 (which == 0 selects matrix A).
// Row lengths of the MASP circuits range from 1 to several hundred terms (Spend: 93 rows of A with 577 terms, 168 rows of C
// with 256: bit packings): `order` lists the constraint rows by decreasing length, so the 64 rows of a wave take about equally
// long, and the first n_long of them (>= R1CS_LONG_ROW terms) get a WAVE each — lanes stride over the terms, a shuffle tree adds
// the 64 partial sums.  (One lane per row made a lone proof wait 1 ms for 577 dependent load-multiply-add steps.)
// The three matrices in ONE launch: blockIdx.z selects A, B or C (R1csMatrices: launch.h).  Lanes: n_long x 64, then one per
// remaining row.
__global__ void k_r1cs_eval(R1csMatrices M, const Fr* __restrict__ w, uint32_t n_vars, uint32_t n_constraints, uint32_t n_inputs) {
    const int which = blockIdx.z;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x, nl = M.n_long[which];
    const uint32_t* __restrict__ rowptr = M.rowptr[which];
    const uint32_t* __restrict__ order = M.order[which];
    const uint32_t* __restrict__ col = M.col[which];
    const Fr* __restrict__ coef = M.coef[which];
    Fr* __restrict__ out = M.out[which];
    w += (size_t)blockIdx.y * n_vars;
    out += (size_t)blockIdx.y * (n_constraints + n_inputs);
    if (v < nl * 64u) {  // (whole waves: the branch is uniform)
        const uint32_t row = order[v >> 6], lane = v & 63u;
        const uint32_t lo = rowptr[row], hi = rowptr[row + 1];
        Fr acc = fe_zero<FrCfg>();
        for (uint32_t t = lo + lane; t < hi; t += 64) acc = fe_add(acc, fe_mul(fr_load(coef + t), fr_load(w + col[t])));
        for (int d = 32; d >= 1; d >>= 1) {
            Fr other;
#pragma unroll
            for (int i = 0; i < 8; ++i) other.v[i] = (uint32_t)__shfl_down((int)acc.v[i], d, 64);
            acc = fe_add(acc, other);
        }
        if (lane == 0) fr_store(out + row, acc);
        return;
    }
    uint32_t row = v - nl * 64u + nl;
    if (row >= n_constraints + n_inputs) return;
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

}  // namespace masp
