// Proof assembly + zcash encoding, point import / export and the per-circuit fixed-base tables, with their launch
// wrappers (launch.h).
#include "device/groth16.cuh"
#include "launch.h"

namespace masp {

void launch_groth16_assemble(hipStream_t s, const VkDevice* vk, const G1Xyzz* fb1, const G2Xyzz* fb2, const G1Xyzz* msm_g1, const G2Xyzz* msm_g2,
                             const uint32_t* rs, size_t rs_stride, uint8_t* proof, uint32_t np) {
    hipLaunchKernelGGL(k_groth16_assemble, dim3(np), dim3(192), 0, s, vk, fb1, fb2, msm_g1, msm_g2, rs, rs_stride, proof);
}
void launch_g1_export(hipStream_t s, const G1Xyzz* p, uint8_t* out) { hipLaunchKernelGGL(k_g1_export, dim3(1), dim3(1), 0, s, p, out); }
void launch_g2_export(hipStream_t s, const G2Xyzz* p, uint8_t* out) { hipLaunchKernelGGL(k_g2_export, dim3(1), dim3(1), 0, s, p, out); }
void launch_g1_import_one(hipStream_t s, const uint8_t* raw, G1Affine* out, int* status) {
    hipLaunchKernelGGL(k_g1_import_one, dim3(1), dim3(1), 0, s, raw, out, status);
}
void launch_g2_import_one(hipStream_t s, const uint8_t* raw, G2Affine* out, int* status) {
    hipLaunchKernelGGL(k_g2_import_one, dim3(1), dim3(1), 0, s, raw, out, status);
}
void launch_fixed_table_g1(hipStream_t s, const G1Affine* pts, G1Xyzz* tabs, uint32_t npts) {
    hipLaunchKernelGGL((k_fixed_table_xyzz<FpOps>), dim3(npts), dim3(64), 0, s, pts, tabs);
}
void launch_fixed_table_g2(hipStream_t s, const G2Affine* pts, G2Xyzz* tabs, uint32_t npts) {
    hipLaunchKernelGGL((k_fixed_table_xyzz<Fp2Ops>), dim3(npts), dim3(64), 0, s, pts, tabs);
}

}  // namespace masp
