// Proof assembly + zcash encoding, point import / export and the per-circuit fixed-base tables, with their launch
// wrappers (launch.h).
#include "device/groth16.hpp"
#include "launch.h"
#include "util.h"

namespace masp {

void launch_groth16_fixed_g1(hipStream_t s, const G1Xyzz* fb1, const uint32_t* rs, size_t rs_stride, G1Xyzz* part, uint32_t np) {
    MASP_LAUNCH(k_groth16_fixed_g1, dim3(np), dim3(128), 0, s, fb1, rs, rs_stride, part);
}
void launch_groth16_fixed_g2(hipStream_t s, const G2Xyzz* fb2, const uint32_t* rs, size_t rs_stride, G2Xyzz* part2, uint32_t np) {
    MASP_LAUNCH(k_groth16_fixed_g2, dim3(np), dim3(64), 0, s, fb2, rs, rs_stride, part2);
}
// which = 0: s*A, 1: r*B1, 2: both in one launch
void launch_groth16_var_mul(hipStream_t s, int which, const G1Xyzz* msm_g1, const uint32_t* rs, size_t rs_stride, G1Xyzz* part, uint32_t np, bool endo) {
    MASP_LAUNCH(k_groth16_var_mul, dim3(np, which == 2 ? 2 : 1), dim3(64), 0, s, which == 1 ? 1u : 0u, msm_g1, rs, rs_stride, part, endo ? 1 : 0);
}
void launch_g1_subgroup_flag(hipStream_t s, const void* pts, size_t stride_bytes, uint32_t n, int* flag) {
    if (n) MASP_LAUNCH(k_g1_subgroup_flag, dim3((n + 63) / 64), dim3(64), 0, s, reinterpret_cast<const uint8_t*>(pts), stride_bytes, n, flag);
}
void launch_groth16_finish_b(hipStream_t s, const VkDevice* vk, const G2Xyzz* part2, const G2Xyzz* msm_g2, uint8_t* proof, uint32_t np) {
    MASP_LAUNCH(k_groth16_finish_b, dim3(np), dim3(64), 0, s, vk, part2, msm_g2, proof);
}
void launch_groth16_finish_ac(hipStream_t s, const VkDevice* vk, const G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np) {
    MASP_LAUNCH(k_groth16_finish_ac, dim3(np), dim3(128), 0, s, vk, part, msm_g1, proof);
}
void launch_groth16_finish_ac_early(hipStream_t s, const VkDevice* vk, G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np) {
    MASP_LAUNCH(k_groth16_finish_ac_early, dim3(np), dim3(128), 0, s, vk, part, msm_g1, proof);
}
void launch_groth16_finish_c_late(hipStream_t s, const G1Xyzz* part, const G1Xyzz* msm_g1, uint8_t* proof, uint32_t np) {
    MASP_LAUNCH(k_groth16_finish_c_late, dim3(np), dim3(64), 0, s, part, msm_g1, proof);
}
void launch_g1_export(hipStream_t s, const G1Xyzz* p, uint8_t* out) { MASP_LAUNCH(k_g1_export, dim3(1), dim3(1), 0, s, p, out); }
void launch_g2_export(hipStream_t s, const G2Xyzz* p, uint8_t* out) { MASP_LAUNCH(k_g2_export, dim3(1), dim3(1), 0, s, p, out); }
void launch_g1_import_one(hipStream_t s, const uint8_t* raw, G1Affine* out, int* status) {
    MASP_LAUNCH(k_g1_import_one, dim3(1), dim3(1), 0, s, raw, out, status);
}
void launch_g2_import_one(hipStream_t s, const uint8_t* raw, G2Affine* out, int* status) {
    MASP_LAUNCH(k_g2_import_one, dim3(1), dim3(1), 0, s, raw, out, status);
}
void launch_fixed_table_g1(hipStream_t s, const G1Affine* pts, G1Xyzz* tabs, uint32_t npts) {
    MASP_LAUNCH((k_fixed_table_xyzz<FpOps>), dim3(npts), dim3(64), 0, s, pts, tabs);
}
void launch_fixed_table_g2(hipStream_t s, const G2Affine* pts, G2Xyzz* tabs, uint32_t npts) {
    MASP_LAUNCH((k_fixed_table_xyzz<Fp2Ops>), dim3(npts), dim3(64), 0, s, pts, tabs);
}

}  // namespace masp
