// Definitions of the MSM driver functions that launch kernels (everything but the accumulation): included only by the
// translation units that instantiate them for one curve (k_msm_g1.hip, k_msm_g2.hip).
#pragma once
#include "device/msm.hpp"
#include "msm_host.h"
#include <type_traits>

namespace masp {

// the G2 bucket tails (gather, heavy buckets, weighted sums) run over lane pairs (Fp2PairOps, field.hpp); 0: one lane per point
#ifndef MASP_G2_PAIR_TAILS
#define MASP_G2_PAIR_TAILS 1
#endif
#ifndef MASP_G2_OCT_LONE
#define MASP_G2_OCT_LONE 1
#endif
template <class O>
struct TailLaneOps {
    typedef typename std::conditional<(MASP_G2_PAIR_TAILS) != 0 && std::is_same<O, Fp2Ops>::value, Fp2PairOps, O>::type type;
};

static inline uint32_t log2_ceil_u64(uint64_t n) {
    uint32_t k = 0;
    while ((1ull << k) < n) ++k;
    return k;
}

template <class O, int BYTES>
int MsmBases<O, BYTES>::load_device(const uint8_t* d_raw, uint32_t n_, hipStream_t s, uint32_t n_eff, int force_c) {
        release();
        n = n_;
        this->n_eff = std::min(n_eff, n_);
        g = force_c ? msm_geom(force_c) : pick_geom(std::min(n_eff, n_));
        if (n == 0) return MASP_HIP_OK;
        if ((uint64_t)n * (uint32_t)g.W > 0x7ffffffeull) {
            last_hip_error() = "MsmBases: more table rows than an entry's 31 bits can name";
            return MASP_HIP_E_INVALID_ARG;
        }
        HIP_TRY(dev_malloc(&tab, sizeof(TabRow<O>) * (size_t)g.W * n));
        int* d_status;
        HIP_TRY(dev_malloc(&d_status, sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), s));
        dim3 grid((n + 63) / 64), block(64);
        MASP_LAUNCH((k_msm_import<O, BYTES>), grid, block, 0, s, d_raw, tab, n, d_status);
        MASP_LAUNCH((k_msm_precompute<O>), grid, block, 0, s, tab, n, g.c, g.W);
        HIP_TRY(hipMemcpyAsync(&import_status, d_status, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        dev_free(d_status);
        return MASP_HIP_OK;
    }

template <class O>
void MsmWorkspace<O>::reduce_to_one(hipStream_t s, uint32_t np, const Xyzz<O>* src, size_t src_stride, uint32_t m, Xyzz<O>* dst, size_t dst_stride,
                                    size_t r_stride) {
        int flip = 0;
        const Xyzz<O>* cur = src;
        size_t cur_stride = src_stride;
        while (true) {
            uint32_t outn = (m + 255) / 256;
            Xyzz<O>* out = outn == 1 ? dst : R[flip];
            size_t out_stride = outn == 1 ? dst_stride : r_stride;
            MASP_LAUNCH((k_xyzz_reduce_block<O>), dim3(outn, np), dim3(256), 256 * sizeof(Xyzz<O>), s, cur, cur_stride, m, out, out_stride);
            if (outn == 1) break;
            cur = out;
            cur_stride = out_stride;
            m = outn;
            flip ^= 1;
        }
    }

template <class O>
template <class OT>
void MsmWorkspace<O>::reduce_to_one_lanes(hipStream_t s, uint32_t np, const Xyzz<O>* src, size_t src_stride, uint32_t m, Xyzz<O>* dst, size_t dst_stride,
                                          size_t r_stride) {
        int flip = 0;
        const Xyzz<O>* cur = src;
        size_t cur_stride = src_stride;
        constexpr uint32_t PTS = reduce_lanes_points<OT>();
        while (true) {
            uint32_t outn = (m + PTS - 1) / PTS;
            Xyzz<O>* out = outn == 1 ? dst : R[flip];
            size_t out_stride = outn == 1 ? dst_stride : r_stride;
            MASP_LAUNCH((k_xyzz_reduce_block_lanes<OT>), dim3(outn, np), dim3(PTS * OT::LANES), PTS * sizeof(Xyzz<O>), s, cur, cur_stride, m, out, out_stride);
            if (outn == 1) break;
            cur = out;
            cur_stride = out_stride;
            m = outn;
            flip ^= 1;
        }
    }

#ifndef MASP_LONE_HEAVY_THREADS
#define MASP_LONE_HEAVY_THREADS 256
#endif
// The bucket tails of an MSM — gather, heavy buckets, weighted sums by levels, block reductions, combine — over OT::LANES lanes
// per point (OT: O itself, its lane-pair form for G2, FpQuadOps for a lone proof's G1 MSMs).
template <class O, class OT>
void msm_tails_enqueue(hipStream_t s, MsmWorkspace<O>& ws, const uint32_t* start, uint32_t nb, uint32_t nchunks, uint32_t np, bool lone, Xyzz<O>* d_out,
                       size_t out_stride) {
    constexpr uint32_t LN = OT::LANES;
    if constexpr (OT::LANES <= 2) {
        if (!lone) {  // a batch: single partials copied, one lane per chunk boundary for the buckets that straddle one (k_msm_bucket_gather_split)
            const uint32_t nbw = (nb * LN + 63) / 64, ncw = (nchunks * LN + 63) / 64;
            MASP_LAUNCH((k_msm_bucket_gather_split<OT>), dim3(nbw + ncw, np), dim3(64), 0, s, ws.part, start, nb, nchunks, ws.bkt, ws.heavy, ws.n_heavy, 8u, nbw);
        } else {
            MASP_LAUNCH((k_msm_bucket_gather<OT>), dim3((nb * LN + 63) / 64, np), dim3(64), 0, s, ws.part, start, nb, nchunks, ws.bkt, ws.heavy, ws.n_heavy, 12u);
        }
    } else {
        MASP_LAUNCH((k_msm_bucket_gather<OT>), dim3((nb * LN + 63) / 64, np), dim3(64), 0, s, ws.part, start, nb, nchunks, ws.bkt, ws.heavy,
                           ws.n_heavy, lone ? 12u : 8u);
    }
    if (lone) {
        const uint32_t heavy_blocks = std::min<uint32_t>(std::max<uint32_t>(4096u / np, 16u), nb) | 1u;
        if constexpr (OT::REPLICATED) {
            // (the lone forms: every heavy bucket shared by MSM_HEAVY_SPLIT workgroups, then joined)
            MASP_LAUNCH((k_msm_bucket_heavy<OT, MASP_LONE_HEAVY_THREADS, MSM_HEAVY_SPLIT>), dim3(heavy_blocks * MSM_HEAVY_SPLIT, np), dim3(MASP_LONE_HEAVY_THREADS), 0, s,
                        ws.part, start, nb, nchunks, ws.bkt, ws.heavy, ws.n_heavy, ws.hparts);
            constexpr uint32_t PER = 64 / (LN * MSM_HEAVY_SPLIT);
            const uint32_t joinable = std::min<uint32_t>(nb, MSM_HEAVY_SLOTS / MSM_HEAVY_SPLIT);
            MASP_LAUNCH((k_msm_heavy_join<OT, MSM_HEAVY_SPLIT>), dim3((joinable + PER - 1) / PER, np), dim3(64), 0, s, ws.hparts, ws.heavy, ws.n_heavy, nb, ws.bkt);
        } else {
            MASP_LAUNCH((k_msm_bucket_heavy<OT, MASP_LONE_HEAVY_THREADS>), dim3(heavy_blocks, np), dim3(MASP_LONE_HEAVY_THREADS), 0, s, ws.part, start, nb,
                        nchunks, ws.bkt, ws.heavy, ws.n_heavy, (Xyzz<O>*)nullptr);
        }
    } else {
        const uint32_t heavy_blocks = std::min<uint32_t>(64u, nb) | 1u;
        MASP_LAUNCH((k_msm_bucket_heavy<OT, 64>), dim3(heavy_blocks, np), dim3(64), 0, s, ws.part, start, nb, nchunks, ws.bkt, ws.heavy,
                           ws.n_heavy, (Xyzz<O>*)nullptr);
    }
    // weighted sum by levels of (G x 128)-bucket workgroups.  Per lane the kernel costs 2 G additions for its buckets plus
    // ~19 for the two lane scans: a batch takes the largest G in {8 .. 64} that still fills one workgroup (2 048 buckets: 16,
    // the 32 768 of h+l: 64 = 2.3 instead of 3.2 additions per bucket) — with other batches in flight total work counts, not
    // the length of the chain; a lone proof takes 4 (shortest dependent chain).
    uint32_t g_log = nb <= WSUM_L ? 0 : WSUM_G_LOG_MIN;  // at most 128 buckets: one per lane, one workgroup, no second level
    if (!lone) {
        const uint32_t hi = 6u;
        g_log = 3;
        while (g_log < hi && (1u << (g_log + 1 + WSUM_L_LOG)) <= nb) ++g_log;
    }
    const uint32_t cs = 1u << (g_log + WSUM_L_LOG);
    const size_t st_stride = (nb + cs - 1) / cs;  // level-0 chunk count bounds every later level
    const Xyzz<O>* bk = ws.bkt;
    size_t bk_stride = nb;
    uint32_t m = nb, off = 1;
    int level = 0, flip = 0;
    do {
        uint32_t chunks = (m + cs - 1) / cs;
        const dim3 grid(chunks, np), block((1u << wsum_points_log<OT>()) * LN);
        switch (g_log) {
#define MASP_WSUM_CASE(GL) \
    case GL: MASP_LAUNCH((k_msm_wsum_level<OT, GL>), grid, block, 0, s, bk, bk_stride, m, off, ws.S[flip], ws.T, st_stride); break;
            MASP_WSUM_CASE(0) MASP_WSUM_CASE(2)
            default:
                if constexpr (OT::LANES <= 2) {  // (the replicated forms only serve lone proofs: g_log 0 or WSUM_G_LOG_MIN)
                    switch (g_log) { MASP_WSUM_CASE(3) MASP_WSUM_CASE(4) MASP_WSUM_CASE(5) MASP_WSUM_CASE(6) }
                }
                break;
#undef MASP_WSUM_CASE
        }
        if constexpr (OT::REPLICATED)
            ws.template reduce_to_one_lanes<OT>(s, np, ws.T, st_stride, chunks, ws.tsum + level, 32, st_stride);
        else
            ws.reduce_to_one(s, np, ws.T, st_stride, chunks, ws.tsum + level, 32, st_stride);
        bk = ws.S[flip];
        bk_stride = st_stride;
        flip ^= 1;
        m = chunks;
        off = 0;
        ++level;
    } while (m > 1);
    if constexpr (OT::REPLICATED)
        MASP_LAUNCH((k_msm_combine_lanes<OT>), dim3(1, np), dim3(64), 0, s, ws.tsum, level, (int)(g_log + WSUM_L_LOG), d_out, out_stride);
    else
        MASP_LAUNCH((k_msm_combine<O>), dim3(1, np), dim3(64), 0, s, ws.tsum, level, (int)(g_log + WSUM_L_LOG), d_out, out_stride);
}
#ifndef MASP_TAILS_QUAD_UNIT
extern template void msm_tails_enqueue<FpOps, FpQuadOps>(hipStream_t, MsmWorkspace<FpOps>&, const uint32_t*, uint32_t, uint32_t, uint32_t, bool, Xyzz<FpOps>*, size_t);
#endif
#ifndef MASP_TAILS_OCT_UNIT
extern template void msm_tails_enqueue<Fp2Ops, Fp2OctOps>(hipStream_t, MsmWorkspace<Fp2Ops>&, const uint32_t*, uint32_t, uint32_t, uint32_t, bool, Xyzz<Fp2Ops>*, size_t);
#endif

template <class O, int BYTES>
int msm_reduce_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, const MsmSortBuf& sb, MsmWorkspace<O>& ws, Xyzz<O>* d_out, size_t out_stride,
                       MsmProfile* prof) {
    const MsmGeom& g = B.g;
    const uint32_t n = B.n, np = sb.np;
    if (sb.n != n || sb.g.c != g.c) {
        last_hip_error() = "msm_reduce_enqueue: sort does not match the base set";
        return MASP_HIP_E_INVALID_ARG;
    }
    int rc = ws.reserve(n, g, np);
    if (rc) return rc;
    {
        // the LDS tree kernel keeps 256 XYZZ points per workgroup: 48 KiB (G1) / 96 KiB (G2) of the 160 KiB LDS
        static PerDeviceOnce once;
        const bool lds_ok = once([] {
            int bytes = 256 * (int)sizeof(Xyzz<O>);
            return hipFuncSetAttribute((const void*)k_xyzz_reduce_block<O>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        });
        if (!lds_ok) {
            last_hip_error() = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
            return MASP_HIP_E_HIP;
        }
    }
    const uint32_t nb = g.nb;
    const uint32_t total = n * g.W;
    const uint32_t nchunks = ws.nchunks_for(n, g, np);
    HIP_TRY(hipMemsetAsync(ws.n_heavy, 0, 4 * np, s));
    MsmProfile::Rec rec{};
    if (prof) {
        rec = prof->acquire();
        rec.alg_bytes = (uint64_t)np * n * (BYTES + 32);  // n x (affine base + scalar) per proof
        hipEventRecord(rec.e0, s);
        prof->mark(s, MsmProfile::PH_START);
    }
    const bool lone = np < 8;  // latency regime: short chains matter more than total work
    // the batch-affine tree in front (msm_tree_levels, msm_host.h): it needs the sort's runs padded to even lengths
    const uint32_t tree_T = sb.pad_log >= 1 ? msm_tree_levels(B.n_eff, g, np, ws.tree_levels) : 0;
    const uint32_t* start = sb.start;
    if (tree_T) {
        uint32_t sub = std::max(1u, std::min(ws.tree_sub, np));
        ws.tree.prof = prof;
        for (uint32_t p0 = 0; p0 < np;) {
            uint32_t q = std::min(sub, np - p0);
            rc = msm_tree_enqueue<O, BYTES>(s, B, sb, ws.tree, p0, q, tree_T);
            // the tree's scratch (~0.4 GB per Spend proof) does not fit — a smaller GPU, more slots, a second prover on the device:
            // halve the sub-batch (and keep it so: the next batch does not try again), below 8 proofs leave the rest of the
            // batch to the XYZZ accumulation over the digit list (it skips the padding entries)
            while (rc == MASP_HIP_E_TREE_SCRATCH && sub > 8) {
                sub = std::max(8u, sub / 2);
                ws.tree_sub = sub;
                q = std::min(sub, np - p0);
                rc = msm_tree_enqueue<O, BYTES>(s, B, sb, ws.tree, p0, q, tree_T);
            }
            if (rc == MASP_HIP_E_TREE_SCRATCH) {
                const uint32_t rest = np - p0;
                ws.tree_fallbacks += rest;
                msm_launch_accumulate<O>(s, B.tab, sb.sorted + (size_t)p0 * sb.ent_stride, sb.ent_stride, sb.start + (size_t)p0 * (nb + 1), nb, nchunks,
                                         ws.part + (size_t)p0 * ((size_t)nchunks + nb), rest);
                HIP_TRY(hipMemcpyAsync(ws.startT + (size_t)p0 * (nb + 1), sb.start + (size_t)p0 * (nb + 1), sizeof(uint32_t) * rest * (nb + 1),
                                       hipMemcpyDeviceToDevice, s));
                break;
            }
            if (rc) return rc;
            msm_launch_accumulate_pts<O>(s, ws.tree.points_x(), ws.tree.points_y(), ws.tree.point_stride(), ws.tree.plan_D(tree_T), nb, nchunks,
                                         ws.part + (size_t)p0 * ((size_t)nchunks + nb), q);
            if (prof) prof->mark(s, MsmProfile::PH_ACC);
            // the bucket tails run over the whole batch: keep this sub-batch's offsets (the next one overwrites the plan)
            HIP_TRY(hipMemcpyAsync(ws.startT + (size_t)p0 * (nb + 1), ws.tree.plan_D(tree_T), sizeof(uint32_t) * q * (nb + 1), hipMemcpyDeviceToDevice, s));
            if (prof) prof->mark(s, MsmProfile::PH_PLAN);
            p0 += q;
        }
        ws.tree.prof = nullptr;
        start = ws.startT;
    } else {
        msm_launch_accumulate<O>(s, B.tab, sb.sorted, sb.ent_stride, sb.start, nb, nchunks, ws.part, np);
        if (prof) prof->mark(s, MsmProfile::PH_ACC);
    }
    if (prof) {
        hipEventRecord(rec.e1, s);
        prof->recs.push_back(rec);
    }
    // A bucket whose entries are spread over `span` chunks or more leaves the gather lanes for a workgroup of its own
    // (k_msm_bucket_heavy: strided sums + shuffle tree).  Batches: span 8 and SINGLE-WAVE workgroups, 65 per proof — the G2
    // kernels hold 512 VGPRs, i.e. a whole SIMD per wave, and all but a handful of these workgroups find no work: at four
    // waves each the idle ones alone kept the chip busy (5.5 -> 2.3 ms for the G2 launch of a 128-proof batch, +3.5 % end to
    // end).  A lone proof keeps span 12 and four waves (shortest chain for its one big bucket).
    // Workgroups go to the 8 XCDs round-robin by linear id x + gridDim.x * y, so with gridDim.x a multiple of 8 every proof's
    // first working workgroup (x = 0: bucket 0) would land on the same XCD: keep gridDim.x odd.
    // a lone proof's G1 tails run over quads (device/quad.hpp): their chains of dependent additions are what it waits for.
    // (Those kernels live in a translation unit of their own, k_msm_g1_lone.hip.)
    if constexpr (std::is_same<O, FpOps>::value) {
        if (lone)
            msm_tails_enqueue<O, FpQuadOps>(s, ws, start, nb, nchunks, np, lone, d_out, out_stride);
        else
            msm_tails_enqueue<O, FpOps>(s, ws, start, nb, nchunks, np, lone, d_out, out_stride);
    } else {
        // ... and its G2 tails over groups of four lane pairs (device/oct.hpp)
        if (lone && (MASP_G2_OCT_LONE))
            msm_tails_enqueue<O, Fp2OctOps>(s, ws, start, nb, nchunks, np, lone, d_out, out_stride);
        else
            msm_tails_enqueue<O, typename TailLaneOps<O>::type>(s, ws, start, nb, nchunks, np, lone, d_out, out_stride);
    }
    return launch_status();  // (a launch the runtime refused: MASP_LAUNCH, util.h)
}


}  // namespace masp
