// Fr NTT / quotient kernels and the static-R1CS kernels with their launch wrappers (launch.h).
#include <algorithm>
#include "device/ntt.hpp"
#include "device/r1cs.hpp"
#include "launch.h"
#include "util.h"

namespace masp {

void launch_fr_powers(hipStream_t s, Fr* table, uint32_t n, const Fr& base, const Fr& scale, int plain) {
    MASP_LAUNCH(k_fr_powers, dim3((n + 255) / 256), dim3(256), 0, s, table, n, base, scale, plain);
}
void launch_ntt_pass(hipStream_t s, Fr* data, const Fr* tw, uint32_t logm, uint32_t s0, uint32_t nst, uint32_t np) {
    const uint32_t lt = (uint32_t)NTT_LT < logm ? (uint32_t)NTT_LT : logm;
    MASP_LAUNCH(k_ntt_pass, dim3(1u << (logm - lt), np), dim3(256), 0, s, data, tw, logm, s0, nst);
}
void launch_ntt_load_bitrev(hipStream_t s, const Fr* x, size_t x_stride, uint32_t nrows, Fr* y, uint32_t logm, uint32_t np) {
    MASP_LAUNCH(k_ntt_load_bitrev, dim3(((1u << logm) + 255) / 256, np), dim3(256), 0, s, x, x_stride, nrows, y, logm);
}
void launch_ntt_copy_bitrev(hipStream_t s, const Fr* x, size_t x_stride, uint32_t nrows, Fr* y, uint32_t logm, uint32_t np) {
    MASP_LAUNCH(k_ntt_copy_bitrev, dim3(((1u << logm) + 255) / 256, np), dim3(256), 0, s, x, x_stride, nrows, y, logm);
}
void launch_ntt_scale_bitrev(hipStream_t s, const Fr* x, const Fr* scale, Fr* y, uint32_t logm, uint32_t np) {
    MASP_LAUNCH(k_ntt_scale_bitrev, dim3(((1u << logm) + 255) / 256, np), dim3(256), 0, s, x, scale, y, logm);
}
void launch_ntt_ab_bitrev(hipStream_t s, const Fr* a, const Fr* b, Fr* y, uint32_t logm, uint32_t np) {
    MASP_LAUNCH(k_ntt_ab_bitrev, dim3(((1u << logm) + 255) / 256, np), dim3(256), 0, s, a, b, y, logm);
}
void launch_fr_scale_sub(hipStream_t s, const Fr* x, const Fr* scale, const Fr* c, const Fr& cscale, Fr* y, uint32_t n, uint32_t np, size_t y_stride) {
    MASP_LAUNCH(k_fr_scale_sub, dim3((n + 255) / 256, np), dim3(256), 0, s, x, scale, c, cscale, y, n, y_stride ? y_stride : (size_t)n);
}
void launch_fr_scale(hipStream_t s, const Fr* x, const Fr* scale, Fr* y, uint32_t n, uint32_t np, size_t y_stride) {
    MASP_LAUNCH(k_fr_scale, dim3((n + 255) / 256, np), dim3(256), 0, s, x, scale, y, n, y_stride ? y_stride : (size_t)n);
}
void launch_fr_from_mont(hipStream_t s, const Fr* x, Fr* y, uint32_t n) {
    MASP_LAUNCH(k_fr_from_mont, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
}
void launch_fr_to_mont(hipStream_t s, const Fr* x, size_t x_stride, Fr* y, uint32_t n, uint32_t np, int* range_err) {
    MASP_LAUNCH(k_fr_to_mont, dim3((n + 255) / 256, np), dim3(256), 0, s, x, x_stride, y, n, range_err);
}
void launch_fr_split_forms(hipStream_t s, Fr* x, size_t x_stride, Fr* y, uint32_t n, uint32_t mont_from, uint32_t np, int* range_err) {
    MASP_LAUNCH(k_fr_split_forms, dim3((n + 255) / 256, np), dim3(256), 0, s, x, x_stride, y, n, mont_from, range_err);
}
void launch_r1cs_eval(hipStream_t s, const R1csMatrices& M, const Fr* w, uint32_t n_vars, uint32_t n_constraints, uint32_t n_inputs, uint32_t np) {
    // lanes: 64 per long row, one per remaining row (the matrix with the most long rows sizes the grid)
    const uint32_t nl = std::max(M.n_long[0], std::max(M.n_long[1], M.n_long[2])), lanes = n_constraints + n_inputs + nl * 63u;
    MASP_LAUNCH(k_r1cs_eval, dim3((lanes + 127) / 128, np, 3), dim3(128), 0, s, M, w, n_vars, n_constraints, n_inputs);
}
void launch_gather_scalars(hipStream_t s, const Fr* src, size_t src_stride, const uint32_t* idx, uint32_t n, Fr* dst, uint32_t np) {
    MASP_LAUNCH(k_gather_scalars, dim3((n + 255) / 256, np), dim3(256), 0, s, src, src_stride, idx, n, dst);
}

}  // namespace masp
