// G1 instantiation of the MSM driver (window-table import / precomputation, bucket gather, heavy buckets, weighted sums).
#include "msm_impl.hpp"

namespace masp {
template struct MsmBases<FpOps, 96>;
template struct MsmWorkspace<FpOps>;
template int msm_reduce_enqueue<FpOps, 96>(hipStream_t, const MsmBases<FpOps, 96>&, const MsmSortBuf&, MsmWorkspace<FpOps>&, Xyzz<FpOps>*, size_t,
                                           MsmProfile*);
}  // namespace masp
