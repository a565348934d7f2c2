// Launch wrappers of the kernel families that live in their own translation units (k_ntt.hip, k_groth16.hip): what the
// host-side pipeline (prover.hip, k_setup.hip) calls instead of launching those kernels itself, so that every kernel
// family compiles once, side by side with the others.  All of them only enqueue on `s`.
#pragma once
#include <hip/hip_runtime.h>

#include "device/curve.hpp"

namespace masp {

// ---- k_ntt.hip: Fr NTT / quotient kernels (device/ntt.hpp) and the R1CS kernels (device/r1cs.hpp) ----
void launch_fr_powers(hipStream_t s, Fr* table, uint32_t n, const Fr& base, const Fr& scale, int plain);
// stages [s0, s0 + nst) of np transforms of 2^logm points at data + p * 2^logm
void launch_ntt_pass(hipStream_t s, Fr* data, const Fr* tw, uint32_t logm, uint32_t s0, uint32_t nst, uint32_t np);
void launch_ntt_load_bitrev(hipStream_t s, const Fr* x, size_t x_stride, uint32_t nrows, Fr* y, uint32_t logm, uint32_t np);
void launch_ntt_copy_bitrev(hipStream_t s, const Fr* x, size_t x_stride, uint32_t nrows, Fr* y, uint32_t logm, uint32_t np);
void launch_ntt_scale_bitrev(hipStream_t s, const Fr* x, const Fr* scale, Fr* y, uint32_t logm, uint32_t np);
void launch_ntt_ab_bitrev(hipStream_t s, const Fr* a, const Fr* b, Fr* y, uint32_t logm, uint32_t np);
void launch_fr_scale_sub(hipStream_t s, const Fr* x, const Fr* scale, const Fr* c, const Fr& cscale, Fr* y, uint32_t n, uint32_t np, size_t y_stride = 0);
// y_p = y + p * y_stride (0: n)
void launch_fr_scale(hipStream_t s, const Fr* x, const Fr* scale, Fr* y, uint32_t n, uint32_t np, size_t y_stride = 0);
void launch_fr_from_mont(hipStream_t s, const Fr* x, Fr* y, uint32_t n);
void launch_fr_to_mont(hipStream_t s, const Fr* x, size_t x_stride, Fr* y, uint32_t n, uint32_t np, int* range_err);
void launch_fr_split_forms(hipStream_t s, Fr* x, size_t x_stride, Fr* y, uint32_t n, uint32_t mont_from, uint32_t np, int* range_err);
struct R1csMatrices {  // the static R1CS of a circuit (CSR, rows by decreasing length) and where a, b, c go
    const uint32_t* rowptr[3];
    const uint32_t* order[3];
    const uint32_t* col[3];
    const Fr* coef[3];
    Fr* out[3];
    uint32_t n_long[3];  // the first n_long rows of `order` (>= R1CS_LONG_ROW terms) are summed by a wave each
};
static constexpr uint32_t R1CS_LONG_ROW = 64;
void launch_r1cs_eval(hipStream_t s, const R1csMatrices& M, const Fr* w, uint32_t n_vars, uint32_t n_constraints, uint32_t n_inputs, uint32_t np);
void launch_gather_scalars(hipStream_t s, const Fr* src, size_t src_stride, const uint32_t* idx, uint32_t n, Fr* dst, uint32_t np);

// ---- k_groth16.hip: proof assembly, point import / export, fixed-base tables (device/groth16.hpp) ----
void launch_groth16_fixed_g1(hipStream_t s, const G1Xyzz* fb1, const uint32_t* rs, size_t rs_stride, G1Xyzz* part, uint32_t np);
void launch_groth16_fixed_g2(hipStream_t s, const G2Xyzz* fb2, const uint32_t* rs, size_t rs_stride, G2Xyzz* part2, uint32_t np);
// endo: Circuit::g1_endo (the points behind A and B1 all lie in the prime-order subgroup: half as many doublings)
void launch_groth16_var_mul(hipStream_t s, int which, const G1Xyzz* msm_g1, const uint32_t* rs, size_t rs_stride, G1Xyzz* part, uint32_t np, bool endo);
// *flag |= 1 if one of n affine G1 points, stride_bytes apart, is outside the prime-order subgroup (infinity skipped)
void launch_g1_subgroup_flag(hipStream_t s, const void* pts, size_t stride_bytes, uint32_t n, int* flag);
void launch_groth16_finish_b(hipStream_t s, const VkDevice* vk, const G2Xyzz* part2, const G2Xyzz* msm_g2, uint8_t* proof, uint32_t np);
void launch_groth16_finish_ac(hipStream_t s, const VkDevice* vk, const G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np);
void launch_groth16_finish_ac_early(hipStream_t s, const VkDevice* vk, G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np);
void launch_groth16_finish_c_late(hipStream_t s, const G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np);
void launch_g1_export(hipStream_t s, const G1Xyzz* p, uint8_t* out);
void launch_g2_export(hipStream_t s, const G2Xyzz* p, uint8_t* out);
void launch_g1_import_one(hipStream_t s, const uint8_t* raw, G1Affine* out, int* status);
void launch_g2_import_one(hipStream_t s, const uint8_t* raw, G2Affine* out, int* status);
// tabs[k * 32 * 255 ...] = fixed-base table of pts[k], k < npts
void launch_fixed_table_g1(hipStream_t s, const G1Affine* pts, G1Xyzz* tabs, uint32_t npts);
void launch_fixed_table_g2(hipStream_t s, const G2Affine* pts, G2Xyzz* tabs, uint32_t npts);

}  // namespace masp
