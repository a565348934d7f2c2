// The G2 bucket tails of a lone proof over groups of four lane pairs (device/oct.hpp, Fp2OctOps): a translation unit — a code object —
// of their own, like the G1 quads (k_msm_g1_lone.hip): the batch path's tail kernels (k_msm_g2.hip) are compiled as they were.
#define MASP_TAILS_OCT_UNIT
#include "msm_impl.hpp"

namespace masp {
template void msm_tails_enqueue<Fp2Ops, Fp2OctOps>(hipStream_t, MsmWorkspace<Fp2Ops>&, const uint32_t*, uint32_t, uint32_t, uint32_t, bool, Xyzz<Fp2Ops>*, size_t);
}  // namespace masp
