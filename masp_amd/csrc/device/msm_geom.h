// Window geometry of one MSM (shared by the kernels and the host-side drivers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace masp {

struct MsmGeom {
    int c;        // digit width in bits: a window (fixed windows) / w of the width-w non-adjacent form
    int W;        // most digits a scalar can have: ceil(256 / c) windows / floor(255 / w) + 1 non-zero NAF digits
    int nb;       // buckets = 2^(c-1)  (|digit| in 1..2^(c-1)) / 2^(c-2)  (|digit| odd, below 2^(c-1): bucket b holds |digit| = 2 b + 1)
    int naf;      // 0: signed fixed windows, table j holds 2^(c j) P;  1: width-c NAF, table t holds 2^t P for EVERY bit position t
    int tpos;     // tables of the base set: W (fixed windows) / 256 (NAF)
};
static inline MsmGeom msm_geom(int c) {
    MsmGeom g;
    g.c = c;
    g.W = (256 + c - 1) / c;
    g.nb = 1 << (c - 1);
    g.naf = 0;
    g.tpos = g.W;
    return g;
}
// Width-w non-adjacent form (w >= 3): every digit is odd and below 2^(w-1) in absolute value, two non-zero digits are at least w bit
// positions apart, and a uniform scalar has one every w + 1 positions on average — 255 / (w + 1) entries where fixed windows with the
// same number of buckets (c = w - 1) have 256 / (w - 1): 14.2 instead of 16 for the 32 768 buckets of h + l, 18.2 instead of 22 for the
// 2 048 of the witness queries.  The price is a table per BIT position (256 x n rows instead of W x n): HBM capacity for additions.
// Scalars are canonical (below r < 2^255), so the last digit sits at position 255 at most.
static inline MsmGeom msm_geom_naf(int w) {
    MsmGeom g;
    g.c = w;
    g.W = 255 / w + 1;
    g.nb = 1 << (w - 2);
    g.naf = 1;
    g.tpos = 256;
    return g;
}
// digits a scalar that is neither 0 nor 1 is expected to have (x 16: fixed point)
static inline uint32_t msm_mean_digits_x16(const MsmGeom& g) { return g.naf ? (uint32_t)(255 * 16 / (g.c + 1)) : (uint32_t)g.W * 16; }

// entries per lane of the accumulation kernel (the gather / heavy-bucket kernels derive the same value)
__host__ __device__ static inline uint32_t msm_chunk_len(uint32_t total, uint32_t nchunks) {
    uint32_t k = (total + nchunks - 1) / nchunks;
    return k < 4 ? 4 : k;  // at least 4 additions per lane: fewer partials to gather
}

// weighted-sum geometry (k_msm_wsum_level): 128 lanes per workgroup, 2^G_LOG buckets per lane
static constexpr unsigned WSUM_L_LOG = 7, WSUM_L = 1u << WSUM_L_LOG;
static constexpr unsigned WSUM_G_LOG_MIN = 2;
// a lone proof's heavy buckets are shared by MSM_HEAVY_SPLIT workgroups each (device/msm.cuh k_msm_bucket_heavy): slots for their shares
static constexpr unsigned MSM_HEAVY_SPLIT = 8, MSM_HEAVY_SLOTS = 2048;
static constexpr unsigned MSM_SORT_THREADS = 1024;
// the entry that fills the gap behind a bucket's run when runs are aligned (MsmSortBuf::pad_log): the point at infinity
static constexpr uint32_t MSM_PAD_ENTRY = 0xffffffffu;

}  // namespace masp
