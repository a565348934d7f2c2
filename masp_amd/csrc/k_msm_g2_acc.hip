// Bucket accumulation over G2 (k_msm_accumulate<Fp2Ops>).
#include "msm_acc_impl.hpp"

namespace masp {
template void msm_launch_accumulate<Fp2Ops>(hipStream_t, const TabRow<Fp2Ops>*, const uint32_t*, size_t, const uint32_t*, uint32_t, uint32_t,
                                            Xyzz<Fp2Ops>*, uint32_t);
}  // namespace masp
