// G2 instantiation of the batch-affine pre-reduction of the bucket runs (device/msm_tree.hpp, msm_tree_impl.hpp).
#include "msm_tree_impl.hpp"

namespace masp {
template struct MsmTreeWs<Fp2Ops>;
template int msm_tree_enqueue<Fp2Ops, 192>(hipStream_t, const MsmBases<Fp2Ops, 192>&, const MsmSortBuf&, MsmTreeWs<Fp2Ops>&, uint32_t, uint32_t, uint32_t);
template void msm_launch_accumulate_pts<Fp2Ops>(hipStream_t, const Fp2*, const Fp2*, size_t, const uint32_t*, uint32_t, uint32_t, Xyzz<Fp2Ops>*, uint32_t);
}  // namespace masp
