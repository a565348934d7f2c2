// G1 instantiation of the batch-affine pre-reduction of the bucket runs (device/msm_tree.hpp, msm_tree_impl.hpp).
#include "msm_tree_impl.hpp"

namespace masp {
template struct MsmTreeWs<FpOps>;
template int msm_tree_enqueue<FpOps, 96>(hipStream_t, const MsmBases<FpOps, 96>&, const MsmSortBuf&, MsmTreeWs<FpOps>&, uint32_t, uint32_t, uint32_t);
template void msm_launch_accumulate_pts<FpOps>(hipStream_t, const Fp*, const Fp*, size_t, const uint32_t*, uint32_t, uint32_t, Xyzz<FpOps>*, uint32_t);
}  // namespace masp
