// The G1 bucket tails of a lone proof over quads (device/quad.hpp, FpQuadOps): a translation unit — a code object — of their own,
// so that the batch path's tail kernels (k_msm_g1.hip) are compiled exactly as they were before these existed.
#define MASP_TAILS_QUAD_UNIT
#include "msm_impl.hpp"

namespace masp {
template void msm_tails_enqueue<FpOps, FpQuadOps>(hipStream_t, MsmWorkspace<FpOps>&, const uint32_t*, uint32_t, uint32_t, uint32_t, bool, Xyzz<FpOps>*, size_t);
}  // namespace masp
