// Window geometry of one MSM (shared by the kernels and the host-side drivers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace masp {

struct MsmGeom {
    int c;        // window width in bits (signed digits)
    int W;        // windows of a 256-bit scalar: ceil(256 / c) — also the number of tables of a base set (table j holds 2^(c j) P)
    int nb;       // buckets = 2^(c-1)  (|digit| in 1..2^(c-1))
};
static inline MsmGeom msm_geom(int c) {
    MsmGeom g;
    g.c = c;
    g.W = (256 + c - 1) / c;
    g.nb = 1 << (c - 1);
    return g;
}
// digits a scalar that is neither 0 nor 1 is expected to have (x 16: fixed point)
static inline uint32_t msm_mean_digits_x16(const MsmGeom& g) { return (uint32_t)g.W * 16; }

// entries per lane of the accumulation kernel (the gather / heavy-bucket kernels derive the same value)
__host__ __device__ static inline uint32_t msm_chunk_len(uint32_t total, uint32_t nchunks) {
    uint32_t k = (total + nchunks - 1) / nchunks;
    return k < 4 ? 4 : k;  // at least 4 additions per lane: fewer partials to gather
}

// weighted-sum geometry (k_msm_wsum_level): 128 lanes per workgroup, 2^G_LOG buckets per lane
static constexpr unsigned WSUM_L_LOG = 7, WSUM_L = 1u << WSUM_L_LOG;
static constexpr unsigned WSUM_G_LOG_MIN = 2;
// a lone proof's heavy buckets are shared by MSM_HEAVY_SPLIT workgroups each (device/msm.hpp k_msm_bucket_heavy): slots for their shares
static constexpr unsigned MSM_HEAVY_SPLIT = 8, MSM_HEAVY_SLOTS = 2048;
static constexpr unsigned MSM_SORT_THREADS = 1024;
// the entry that fills the gap behind a bucket's run when runs are aligned (MsmSortBuf::pad_log): the point at infinity
static constexpr uint32_t MSM_PAD_ENTRY = 0xffffffffu;

}  // namespace masp
