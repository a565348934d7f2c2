// Counting sort of the MSM digits: kernels (device/msm_sort.hpp) + their host-side enqueue.
#include <cstring>

#include "device/msm_sort.hpp"
#include "msm_host.h"

namespace masp {

int msm_sort_enqueue(hipStream_t s, uint32_t n, const MsmGeom& g, MsmSortBuf& sb, const uint32_t* d_scalars, size_t scalar_stride,
                                   uint32_t np, uint32_t pad_log) {
    static_assert((1u << 15) / MSM_SCAN_BLOCK <= 64, "k_msm_offsets_scan_b: a wave's lanes fetch the block totals");
    if (g.c < 2 || g.nb > (1 << 15)) {
        last_hip_error() = "MSM window width must be 2..16 bits: the bucket histogram lives in LDS";
        return MASP_HIP_E_INVALID_ARG;
    }
    if ((uint64_t)n * (uint32_t)g.W > 0x7ffffffeull) {
        last_hip_error() = "msm_sort_enqueue: the base set has more table rows than an entry's 31 bits can name";
        return MASP_HIP_E_INVALID_ARG;
    }
    if (pad_log > 12) {
        last_hip_error() = "msm_sort_enqueue: runs can be aligned to at most 2^12 entries";
        return MASP_HIP_E_INVALID_ARG;
    }
    int rc = sb.reserve(n, g, np, pad_log);
    if (rc) return rc;
    sb.n = n;
    sb.np = np;
    sb.g = g;
    sb.pad_log = pad_log;
    sb.ent_stride = MsmSortBuf::padded_entries(n, g, pad_log);
    const uint32_t ng = MsmSortBuf::ranges_for(n, np), nb = g.nb;
    // two-pass placement (runs instead of single scattered words): the first pass stages a tile's entries (a word and a byte each) in LDS
    // ... and the second pass is one workgroup per (proof, coarse bin): with too few of them (a lone proof's b_g2 on 8-bit windows: ONE,
    // 0.57 ms for 600 000 entries) the single-pass scatter over the scalar ranges is the shorter chain
    const uint32_t wide = MsmSortBuf::msm_rows_wide(n, g) ? 1u : 0u;  // the low bucket bits of an entry in `tmpf` instead of the entry word
    // LDS of the first pass: the bins' counters + per staged entry a word and a byte (its bin), wide: one more byte — of the 160 KiB of a CU
    const int part_entry = wide ? 6 : 5, part_fixed = 4 * (4 * 256 + 8), part_w_max = std::min(30, (160 * 1024 - part_fixed) / ((int)MSM_PART_TILE * part_entry));
    const bool two_pass = nb >= MSM_FINE && g.W <= part_w_max && (uint64_t)(nb >> MSM_FINE_LOG) * np >= 8;
    const int part_lds = part_fixed + part_entry * (int)MSM_PART_TILE * g.W;
    static PerDeviceOnce once;
    const bool lds_ok = once([] {
        int bytes = 4 << 15;
        const int part_bytes = 4 * (4 * 256 + 8) + 5 * (int)MSM_PART_TILE * 30;  // = the largest part_lds below (6 x 25 = 5 x 30)
        return hipFuncSetAttribute((const void*)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute((const void*)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute((const void*)k_msm_partition, hipFuncAttributeMaxDynamicSharedMemorySize, part_bytes) == hipSuccess;
    });
    if (!lds_ok) {
        last_hip_error() = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        return MASP_HIP_E_HIP;
    }
    MASP_LAUNCH(k_msm_hist, dim3(ng, np), dim3(MSM_SORT_THREADS), 4 * nb, s, d_scalars, scalar_stride, n, g, ng, sb.hist_wg);
    MASP_LAUNCH(k_msm_offsets_cols, dim3((nb + 255) / 256, np), dim3(256), 0, s, sb.hist_wg, ng, nb, sb.dense);
    const uint32_t scan_blocks = (nb + MSM_SCAN_BLOCK - 1) / MSM_SCAN_BLOCK;
    MASP_LAUNCH(k_msm_offsets_scan_a, dim3(scan_blocks, np), dim3(256), 0, s, nb, sb.start, sb.dense, pad_log, sb.btot);
    MASP_LAUNCH(k_msm_offsets_scan_b, dim3(scan_blocks, np), dim3(256), 0, s, nb, sb.start, sb.dense, sb.btot);
    if (two_pass) {
        const uint32_t nbins = nb >> MSM_FINE_LOG;
        const uint32_t cw = std::min(ng, 4u);  // waves per workgroup of k_msm_coarse: one per scalar range
        MASP_LAUNCH(k_msm_coarse, dim3(nbins, np, (ng + cw - 1) / cw), dim3(64 * cw), 0, s, sb.hist_wg, ng, nb, sb.crel);
        MASP_LAUNCH(k_msm_partition, dim3(ng, np), dim3(MSM_PART_TILE), part_lds, s, d_scalars, scalar_stride, n, g, ng, sb.crel, sb.dense, sb.tmp, sb.tmpf, wide);
        MASP_LAUNCH(k_msm_bucketize, dim3(nbins, np), dim3(1024), 0, s, sb.tmp, sb.tmpf, (size_t)n * g.W, sb.dense, sb.start, nb, sb.sorted, sb.ent_stride, wide);
    } else {
        // (the single-pass placement writes entries only: aligned runs get their padding from a fill first)
        if (pad_log) HIP_TRY(hipMemsetAsync(sb.sorted, 0xff, 4 * sb.ent_stride * np, s));
        MASP_LAUNCH(k_msm_scatter, dim3(ng, np), dim3(MSM_SORT_THREADS), 4 * nb, s, d_scalars, scalar_stride, n, g, ng, sb.hist_wg, sb.start, sb.sorted,
                        sb.ent_stride);
    }
    return launch_status();
}

}  // namespace masp
