// Host driver of the batch-affine pre-reduction (device/msm_tree.hpp): instantiated per curve in k_msm_g1_tree.hip /
// k_msm_g2_tree.hip.
#pragma once
#include "device/msm_tree.hpp"
#include "msm_host.h"
#include <type_traits>

namespace masp {

// ---- geometry of a level (host side; the device works with the exact counts of D / Q) -----------------------------------
// E_ub: upper bound of a proof's digit-list length.  Points of level L <= E / 2^L + nb, pairs <= E / 2^(L+1) + nb / 2 + 1.
// (even: the planes keep points of even and odd index in separate halves, device/msm_tree.hpp)
static inline uint32_t tree_points_ub(uint64_t E_ub, uint32_t nb, uint32_t L) { return (uint32_t)(((E_ub >> L) + nb + 2) & ~(uint64_t)1); }
static inline uint32_t tree_pairs_ub(uint64_t E_ub, uint32_t nb, uint32_t L) { return (uint32_t)((E_ub >> (L + 1)) + nb / 2 + 1); }
// lanes per proof of the two passes: ~MSM_TREE_KP pairs per lane, whole workgroups
// G2 kernels that run over lane pairs (Fp2PairOps, field.hpp: half an Fp2 per lane, two waves per SIMD where the Fp2Ops form
// gets one).  Bits: 1 pass 2, 2 pass 1, 4 the shared inversions, 8 the accumulation of the level-T points.
#ifndef MASP_TREE_G2_PAIR
#define MASP_TREE_G2_PAIR 15
#endif
template <class O, int BIT>
struct TreeLaneOps {
    typedef typename std::conditional<((MASP_TREE_G2_PAIR) & BIT) != 0 && std::is_same<O, Fp2Ops>::value, Fp2PairOps, O>::type type;
};
#ifndef MASP_TREE_KP
#define MASP_TREE_KP 128   // pairs per lane of the two passes (16 / 32 / 64 / 128 / 256 measured: 955 / 981 / 997 / 1007 / 1009 proofs/s)
#endif
static constexpr uint32_t MSM_TREE_KP = MASP_TREE_KP;
static inline uint32_t tree_lanes(uint64_t E_ub, uint32_t nb, uint32_t L) {
    const uint32_t pairs = tree_pairs_ub(E_ub, nb, L);
    return std::max<uint32_t>(256u, ((pairs + MSM_TREE_KP - 1) / MSM_TREE_KP + 255u) & ~255u);
}

template <class O>
int MsmTreeWs<O>::reserve(uint64_t E_ub, uint32_t nb, uint32_t q, uint32_t T) {
    const size_t plan = (size_t)(T + 1) * q * (nb + 1);
    const size_t recs = (size_t)q * tree_pairs_ub(E_ub, nb, 1);
    const size_t NT0 = tree_lanes(E_ub, nb, 0);
    const size_t lanes = (size_t)q * NT0;
    // pre[(j q + p) NT + t], j < ceil(pairs / NT) <= KP (+1 for the rounding of NT): bounded by level 0
    const size_t pres = (size_t)((tree_pairs_ub(E_ub, nb, 0) + NT0 - 1) / NT0) * lanes;
    pre_cap = pres;
    const size_t pts1 = (size_t)q * tree_points_ub(E_ub, nb, 1), pts2 = T >= 2 ? (size_t)q * tree_points_ub(E_ub, nb, 2) : 0;
    const size_t m1 = (lanes + BINV_C - 1) / BINV_C;
    // carve: every view 256-byte aligned
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t oD = take(4 * plan), oQ = take(4 * plan), oRec = take(sizeof(uint2) * recs), oPre = take(sizeof(F) * pres), oTp = take(sizeof(F) * lanes),
                 oTinv = take(sizeof(F) * lanes), oBpre = take(sizeof(F) * lanes), oBtot = take(sizeof(F) * m1), oBitot = take(sizeof(F) * m1),
                 oBpre2 = take(sizeof(F) * m1), oX1 = take(sizeof(F) * pts1), oY1 = take(sizeof(F) * pts1), oX0 = take(sizeof(F) * pts2),
                 oY0 = take(sizeof(F) * pts2);
    int rc = arena->reserve(off);
    if (rc) return rc;
    uint8_t* b = arena->p;
    D = (uint32_t*)(b + oD);
    Q = (uint32_t*)(b + oQ);
    rec = (uint2*)(b + oRec);
    pre = (F*)(b + oPre);
    tp = (F*)(b + oTp);
    tinv = (F*)(b + oTinv);
    bpre = (F*)(b + oBpre);
    btot = (F*)(b + oBtot);
    bitot = (F*)(b + oBitot);
    bpre2 = (F*)(b + oBpre2);
    px[1] = (F*)(b + oX1);
    py[1] = (F*)(b + oY1);
    px[0] = (F*)(b + oX0);
    py[0] = (F*)(b + oY0);
    return MASP_HIP_OK;
}

// out[i] = 1 / in[i], i < n (no zeros among them), on stream s: chains of BINV_C products, then <= BINV_MID lanes with one
// inversion each, then the chains backwards
template <class O>
void MsmTreeWs<O>::batch_invert(hipStream_t s, const F* in, uint32_t n, F* out) {
    typedef typename TreeLaneOps<O, 4>::type OI;
    typedef typename OI::T FI;
    constexpr uint32_t LN = OI::LANES;
    if (n <= 4 * BINV_MID) {
        const uint32_t M = std::min<uint32_t>(n, BINV_MID);
        MASP_LAUNCH((k_binv_mid<OI>), dim3((M * LN + 63) / 64), dim3(64), 0, s, (const FI*)in, n, M, (FI*)bpre, (FI*)out);
        return;
    }
    const uint32_t M1 = (n + BINV_C - 1) / BINV_C, M2 = std::min<uint32_t>(M1, BINV_MID);
    MASP_LAUNCH((k_binv_fwd<OI>), dim3((M1 * LN + 255) / 256), dim3(256), 0, s, (const FI*)in, n, M1, (FI*)bpre, (FI*)btot);
    MASP_LAUNCH((k_binv_mid<OI>), dim3((M2 * LN + 63) / 64), dim3(64), 0, s, (const FI*)btot, M1, M2, (FI*)bpre2, (FI*)bitot);
    MASP_LAUNCH((k_binv_bwd<OI>), dim3((M1 * LN + 255) / 256), dim3(256), 0, s, (const FI*)in, n, M1, (const FI*)bpre, (const FI*)bitot, (FI*)out);
}

// T levels of pairwise affine additions over the digit lists of proofs [p0, p0 + q) of the sort `sb`.  Afterwards the points of
// proof p0 + i lie at points_x() / points_y() + i * point_stride(), bucket b of it at [DT[b], DT[b + 1]) with
// DT = plan_D(T) + i * (nb + 1).  No host synchronisation.
template <class O, int BYTES>
int msm_tree_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, const MsmSortBuf& sb, MsmTreeWs<O>& tw, uint32_t p0, uint32_t q, uint32_t T) {
    typedef typename O::T F;
    const uint32_t nb = (uint32_t)B.g.nb;
    if (sb.pad_log < 1) {
        last_hip_error() = "msm_tree_enqueue: the sort must pad every run to an even length (MsmSortBuf::pad_log >= 1)";
        return MASP_HIP_E_INVALID_ARG;
    }
    const uint64_t E_ub = sb.ent_stride;  // entries per proof, the padding included
    int rc = tw.reserve(E_ub, nb, q, T);
    if (rc) return rc;
    tw.q = q;
    tw.nb = nb;
    tw.T = T;
    tw.stride[1] = tree_points_ub(E_ub, nb, 1);
    tw.stride[0] = tree_points_ub(E_ub, nb, 2);
    const size_t ent_stride = E_ub;
    const uint32_t* sorted = sb.sorted + (size_t)p0 * ent_stride;
    const uint32_t* start = sb.start + (size_t)p0 * (nb + 1);
    MsmProfile* const prof = tw.prof;
    MASP_LAUNCH(k_tree_plan, dim3(T + 1, q), dim3(1024), 0, s, start, nb, tw.D, tw.Q);
    if (prof) prof->mark(s, MsmProfile::PH_PLAN);
    const size_t lvl = (size_t)q * (nb + 1);
    const size_t rec_stride = tree_pairs_ub(E_ub, nb, 1);
    for (uint32_t L = 0; L < T; ++L) {
        const uint32_t *Dl = tw.D + L * lvl, *Dn = tw.D + (L + 1) * lvl, *Ql = tw.Q + L * lvl;
        const uint32_t NT = tree_lanes(E_ub, nb, L), pairs_ub = tree_pairs_ub(E_ub, nb, L);
        const F *xi = tw.px[L & 1], *yi = tw.py[L & 1];
        F *xo = tw.px[(L + 1) & 1], *yo = tw.py[(L + 1) & 1];
        const size_t si = tw.stride[L & 1], so = tw.stride[(L + 1) & 1];
        const dim3 rgrid(std::min<uint32_t>((pairs_ub + 255) / 256, 4096u), q), grid(NT / 256, q), cgrid((nb + 255) / 256, q), block(256);
        typedef typename TreeLaneOps<O, 2>::type O1;
        typedef typename O1::T F1;
        const dim3 grid1(NT * O1::LANES / 256, q);
        // `pre` is a plane of tw.pre_cap curve elements (device/msm_tree.hpp): its capacity in the kernels' element types
        const size_t pre_cap1 = tw.pre_cap * sizeof(F) / sizeof(F1);
        const uint32_t out_whole = L + 1 == T;  // the last level's points as whole elements: what k_msm_accumulate_pts reads
        // level 0: the "records" are the digit list itself, two entries at a time (runs of even length: pair q = entries 2q, 2q + 1)
        const void* recL = L >= sb.pad_log ? (const void*)tw.rec : nullptr;  // levels >= 1: records, or none below pad_log
        const void* rec0 = (const void*)sorted;
        const size_t rec0_stride = ent_stride / 2;
        if (L == 0) {
            MASP_LAUNCH((k_tree_pass1<O1, true>), grid1, block, 0, s, B.tab, (const F1*)xi, (const F1*)yi, si, rec0, rec0_stride, Ql, nb, NT,
                               (F1*)tw.pre, pre_cap1, (F1*)tw.tp);
        } else {
            // (a level below the sort's pad_log has runs of even lengths only: no records, no odd points to copy)
            if (L >= sb.pad_log) {
                MASP_LAUNCH(k_tree_records, rgrid, block, 0, s, Dl, Dn, Ql, nb, tw.rec, rec_stride);
                if (prof) prof->mark(s, MsmProfile::PH_PLAN);
            }
            MASP_LAUNCH((k_tree_pass1<O1, false>), grid1, block, 0, s, B.tab, (const F1*)xi, (const F1*)yi, si, recL, rec_stride, Ql, nb,
                               NT, (F1*)tw.pre, pre_cap1, (F1*)tw.tp);
        }
        if (prof) prof->mark(s, MsmProfile::PH_PASS1);
        tw.batch_invert(s, tw.tp, q * NT, tw.tinv);
        if (prof) prof->mark(s, MsmProfile::PH_INV);
        typedef typename TreeLaneOps<O, 1>::type O2;
        typedef typename O2::T F2;
        const dim3 grid2(NT * O2::LANES / 256, q);
        const size_t pre_cap2 = tw.pre_cap * sizeof(F) / sizeof(F2);
        if (L == 0)
            MASP_LAUNCH((k_tree_pass2<O2, true>), grid2, block, 0, s, B.tab, (const F2*)xi, (const F2*)yi, si, rec0, rec0_stride, Ql, nb, NT,
                               (const F2*)tw.pre, pre_cap2, (const F2*)tw.tinv, (F2*)xo, (F2*)yo, so, out_whole);
        else
            MASP_LAUNCH((k_tree_pass2<O2, false>), grid2, block, 0, s, B.tab, (const F2*)xi, (const F2*)yi, si, recL, rec_stride, Ql,
                               nb, NT, (const F2*)tw.pre, pre_cap2, (const F2*)tw.tinv, (F2*)xo, (F2*)yo, so, out_whole);
        if (prof) prof->mark(s, MsmProfile::PH_PASS2);
        // (level 0 has no odd runs: a run of odd length met its padding entry as P + infinity)
        if (L >= sb.pad_log) {
            MASP_LAUNCH((k_tree_copy<O, false>), cgrid, block, 0, s, B.tab, sorted, ent_stride, xi, yi, si, Dl, Dn, nb, xo, yo, so, out_whole);
            if (prof) prof->mark(s, MsmProfile::PH_PLAN);
        }
    }
    return launch_status();
}

template <class O>
void msm_launch_accumulate_pts(hipStream_t s, const typename O::T* xs, const typename O::T* ys, size_t pt_stride, const uint32_t* start, uint32_t nb,
                               uint32_t nchunks, Xyzz<O>* part, uint32_t np) {
    typedef typename TreeLaneOps<O, 8>::type OA;
    typedef typename OA::T FA;
    MASP_LAUNCH((k_msm_accumulate_pts<OA>), dim3((nchunks * OA::LANES + 63) / 64, np), dim3(64), 0, s, (const FA*)xs, (const FA*)ys, pt_stride, start,
                       nb, nchunks, part);
}

}  // namespace masp
