// G2 instantiation of the MSM driver (window-table import / precomputation, bucket gather, heavy buckets, weighted sums).
#include "msm_impl.hpp"

namespace masp {
template struct MsmBases<Fp2Ops, 192>;
template struct MsmWorkspace<Fp2Ops>;
template int msm_reduce_enqueue<Fp2Ops, 192>(hipStream_t, const MsmBases<Fp2Ops, 192>&, const MsmSortBuf&, MsmWorkspace<Fp2Ops>&, Xyzz<Fp2Ops>*, size_t,
                                             MsmProfile*);
}  // namespace masp
