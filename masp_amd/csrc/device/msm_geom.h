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
    int rg;       // regions of the table (1, or MSM_REGIONS): region r holds, table after table, the rows of the points i = r (mod rg) —
                  // one region per XCD, whose L2 TLB reaches ~3.5 GiB of a table that is far larger (see msm_row)
};
static constexpr int MSM_REGIONS = 8;  // = XCDs of an MI355X: workgroups go to them round-robin by their linear id
static inline MsmGeom msm_geom(int c) {
    MsmGeom g;
    g.c = c;
    g.W = (256 + c - 1) / c;
    g.nb = 1 << (c - 1);
    g.naf = 0;
    g.tpos = g.W;
    g.rg = 1;
    return g;
}
// Width-w non-adjacent form (w >= 3): every digit is odd and below 2^(w-1) in absolute value, two non-zero digits are at least w bit
// positions apart, and a uniform scalar has one every w + 1 positions on average — 255 / (w + 1) entries where fixed windows with the
// same number of buckets (c = w - 1) have 256 / (w - 1): 14.2 instead of 16 for the 32 768 buckets of h + l, 18.2 instead of 22 for the
// 2 048 of the witness queries.  The price is a table per BIT position (256 x n rows instead of W x n): HBM capacity for additions.
// Scalars are canonical (below r < 2^255), so the last digit sits at position 255 at most.
static inline MsmGeom msm_geom_naf(int w, int regions = 1) {
    MsmGeom g;
    g.c = w;
    g.W = 255 / w + 1;
    g.nb = 1 << (w - 2);
    g.naf = 1;
    g.tpos = 256;
    g.rg = regions;
    return g;
}
// Row of table t, point i, in a base set of n points.  With regions the table is laid out region after region: random 128-byte rows
// arrive at 6.5 TB/s from up to 3 GiB and at 1.8 TB/s from 4 GiB on — the reach of ONE XCD's L2 TLB — but at 6.3 TB/s again from 17 GiB
// if every XCD only gathers from its own eighth (profiles/r05_gather_rate_tlb_reach_and_xcd_partition_ubench.txt).  The sort keeps the
// entries of a bucket region by region, each region's share padded to an even length, so that the pairs of the bucket tree's level 0
// (entries 2q, 2q + 1) never mix regions, and that level's two passes hand every pair to a workgroup on the XCD of its region.
__host__ __device__ static inline uint32_t msm_region_rows(uint32_t n, int rg) { return (n + (uint32_t)rg - 1u) / (uint32_t)rg; }
__host__ __device__ static inline uint32_t msm_row(const MsmGeom& g, uint32_t n, uint32_t t, uint32_t i) {
    if (g.rg <= 1) return t * n + i;
    const uint32_t rg = (uint32_t)g.rg, per = msm_region_rows(n, g.rg);
    return ((i % rg) * (uint32_t)g.tpos + t) * per + i / rg;
}
__host__ __device__ static inline uint64_t msm_table_rows(const MsmGeom& g, uint32_t n) {
    return g.rg <= 1 ? (uint64_t)n * (uint32_t)g.tpos : (uint64_t)msm_region_rows(n, g.rg) * (uint32_t)g.rg * (uint32_t)g.tpos;
}
// the region a table row lies in (regions > 1)
__host__ __device__ static inline uint32_t msm_row_region(const MsmGeom& g, uint32_t n, uint32_t row) {
    return row / (msm_region_rows(n, g.rg) * (uint32_t)g.tpos);
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
