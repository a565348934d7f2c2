// The dominant kernel: bucket accumulation over G1 (k_msm_accumulate<FpOps>, multiplier inlined).
#include "msm_acc_impl.hpp"

namespace masp {
template void msm_launch_accumulate<FpOps>(hipStream_t, const TabRow<FpOps>*, const uint32_t*, size_t, const uint32_t*, uint32_t, uint32_t,
                                           Xyzz<FpOps>*, uint32_t);
}  // namespace masp
