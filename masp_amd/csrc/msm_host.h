// Host-side driver of the MSM kernels: owns the precomputed window tables of one base set and a reusable workspace, and
// enqueues one MSM on a HIP stream without any host sync.  This header holds types and declarations only; the functions
// that launch kernels are defined in msm_impl.hpp / msm_acc_impl.hpp and instantiated in their own translation units
// (k_msm_g1.hip, k_msm_g2.hip, k_msm_g1_acc.hip, k_msm_g2_acc.hip, k_msm_sort.hip), so that the kernel families compile side
// by side and a change to one kernel recompiles one unit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "device/curve.hpp"
#include "device/msm_geom.h"
#include "util.h"

// lanes of the accumulation of a lone proof's narrow-window MSM (B2 on 8-bit windows): A/B builds
#ifndef MASP_LONE_NARROW_LANES_LOG
#define MASP_LONE_NARROW_LANES_LOG 16
#endif
namespace masp {

// ---- base set: T[j][i] = 2^(c j) P_i -------------------------------------------------------------------------------------------
template <class O, int BYTES>
struct MsmBases {
    MsmGeom g{};
    uint32_t n = 0;
    uint32_t n_eff = 0;        // scalars expected to be neither 0 nor 1 (<= n): the mean length of a bucket's run follows from it
    TabRow<O>* tab = nullptr;  // g.W * n rows of 128 / 256 bytes
    int import_status = 0;     // PT_* bits seen while decoding

    ~MsmBases() { release(); }
    void release() {
        if (tab) dev_free(tab);
        tab = nullptr;
    }
    // Window width by the number of scalars expected to be neither 0 nor 1 (`n_eff`; the caller knows the witness
    // statistics of its circuit, a generic caller passes n): per non-trivial scalar the accumulation costs W = 256/c
    // mixed additions, per bucket the gather + weighted sum cost ~5.5 full additions.
    static MsmGeom pick_geom(uint32_t n_eff) {
        int c = n_eff >= (1u << 16) ? 16 : n_eff >= (1u << 12) ? 12 : n_eff >= (1u << 8) ? 10 : 7;
        return msm_geom(c);
    }
    // raw: device pointer to n uncompressed points (bellman wire format)          [msm_impl.hpp]
    // force_c: window width (0: pick_geom)
    int load_device(const uint8_t* d_raw, uint32_t n_, hipStream_t s, uint32_t n_eff = 0xffffffffu, int force_c = 0);
    int load_host(const uint8_t* raw, uint32_t n_, hipStream_t s, uint32_t n_eff = 0xffffffffu, int force_c = 0) {
        uint8_t* d_raw = nullptr;
        if (n_) {
            HIP_TRY(dev_malloc(&d_raw, (size_t)n_ * BYTES));
            HIP_TRY(hipMemcpyAsync(d_raw, raw, (size_t)n_ * BYTES, hipMemcpyHostToDevice, s));
        }
        int rc = load_device(d_raw, n_, s, n_eff, force_c);
        if (d_raw) dev_free(d_raw);
        return rc;
    }
};

// ---- sort buffers (curve-independent): one counting sort can feed several MSMs over the same scalars ---------
struct MsmSortBuf {
    size_t cap_ent = 0, cap_nb = 0, cap_np = 0, cap_hist = 0, cap_crel = 0;  // cap_hist, cap_crel: words
    uint32_t *sorted = nullptr, *hist_wg = nullptr, *start = nullptr;
    uint32_t *tmp = nullptr, *crel = nullptr;  // two-pass placement: entries grouped by coarse bin; per-range offsets of the bins
    uint8_t* tmpf = nullptr;                   // ... and the low 7 bits of every such entry's bucket, where the entry word has no room
                                               // for them (msm_rows_wide: more than 2^24 table rows); not allocated otherwise
    bool has_tmpf = false;
    static bool msm_rows_wide(uint32_t n_, const MsmGeom& g_) { return (uint64_t)n_ * (uint32_t)g_.W > (1u << 24); }
    uint32_t* dense = nullptr;                 // [np][nb + 1] offsets without padding (where a bin lies in `tmp`)
    uint2* btot = nullptr;                     // [np][ceil(nb / 1024)] the offsets scan's block totals (packed, aligned)
    // what the last msm_sort_enqueue produced (consumed by msm_reduce_enqueue)
    uint32_t n = 0, np = 0;
    MsmGeom g{};
    // Every bucket's run in `sorted` starts at a multiple of 2^pad_log entries; the gap behind a run holds MSM_PAD_ENTRY (the
    // point at infinity).  The batch-affine tree (device/msm_tree.hpp) asks for pad_log = 2 (MASP_TREE_PAD_LOG): every run's
    // length is a multiple of four, so pair q of level 0 is simply entries 2q, 2q + 1 and lands at point q of level 1, and the
    // same again for level 1 — no per-pair records for the two levels that hold three quarters of all pairs (1 / 2 / 3
    // measured: stage 54.1 / 52.7 / 54.2 ms per G1 MSM; padding to 2^levels costs more pair slots than records: DESIGN.md §6).
    // start[] counts the padding; ent_stride = entries per proof of `sorted`.
    uint32_t pad_log = 0;
    size_t ent_stride = 0;
    static size_t padded_entries(uint32_t n_, const MsmGeom& g_, uint32_t pad_log_) {
        const size_t unit = (size_t)1 << pad_log_;
        return ((size_t)n_ * g_.W + (size_t)g_.nb * (unit - 1) + unit - 1) / unit * unit;
    }

    ~MsmSortBuf() { release(); }
    void release() {
        void* ptrs[] = {sorted, hist_wg, start, tmp, crel, dense, tmpf, btot};
        for (void* p : ptrs)
            if (p) dev_free(p);
        sorted = hist_wg = start = tmp = crel = dense = nullptr;
        tmpf = nullptr;
        btot = nullptr;
        has_tmpf = false;
        cap_ent = cap_nb = cap_np = cap_hist = cap_crel = 0;
    }
    // scalar ranges (= sorting workgroups) per proof: enough to occupy the chip across the batch, not more
    static uint32_t ranges_for(uint32_t n, uint32_t np) {
        uint32_t ng = std::max(1u, 512u / std::max(np, 1u));
        ng = std::min(ng, 64u);
        return std::max(1u, std::min(ng, (n + 1023) / 1024));
    }
    int reserve(uint32_t n_, const MsmGeom& g_, uint32_t np_, uint32_t pad_log_ = 0) {
        // the per-workgroup histograms are addressed with the launch's own strides: np x ng x nb words, and np x ng <= 512
        // for every batch size (ranges_for) — a batch of 64 proofs (8 ranges each) fits what a batch of 256 (2 each) allocated
        const size_t need_ent = std::max(padded_entries(n_, g_, pad_log_), cap_ent), ng = ranges_for(n_, np_);
        const size_t bins = std::max<size_t>(g_.nb >> 7, 1);
        const size_t hist_need = (size_t)np_ * ng * g_.nb, crel_need = (size_t)np_ * ng * bins;
        const bool want_tmpf = has_tmpf || msm_rows_wide(n_, g_);
        if (need_ent <= cap_ent && (size_t)g_.nb <= cap_nb && np_ <= cap_np && hist_need <= cap_hist && crel_need <= cap_crel && want_tmpf == has_tmpf)
            return MASP_HIP_OK;
        const size_t need_nb = std::max<size_t>(g_.nb, cap_nb), need_np = std::max<size_t>(np_, cap_np);
        const size_t rows = std::max<size_t>(need_np, 512);
        const size_t need_hist = std::max(std::max(hist_need, cap_hist), rows * need_nb);
        const size_t need_crel = std::max(std::max(crel_need, cap_crel), rows * std::max<size_t>(need_nb >> 7, 1));
        release();  // (capacities are 0 from here on: a failed allocation below must not leave them claiming memory)
        auto alloc_all = [&]() -> int {
            HIP_TRY(dev_malloc(&sorted, need_np * 4 * std::max<size_t>(need_ent, 1)));
            HIP_TRY(dev_malloc(&hist_wg, 4 * need_hist));
            HIP_TRY(dev_malloc(&start, need_np * 4 * (need_nb + 1)));
            HIP_TRY(dev_malloc(&tmp, need_np * 4 * std::max<size_t>(need_ent, 1)));
            if (want_tmpf) HIP_TRY(dev_malloc(&tmpf, need_np * std::max<size_t>(need_ent, 1)));
            HIP_TRY(dev_malloc(&crel, 4 * need_crel));
            HIP_TRY(dev_malloc(&dense, need_np * 4 * (need_nb + 1)));
            HIP_TRY(dev_malloc(&btot, need_np * sizeof(uint2) * ((need_nb + 1023) / 1024)));
            return MASP_HIP_OK;
        };
        if (int rc = alloc_all()) {
            release();
            return rc;
        }
        has_tmpf = want_tmpf;
        cap_ent = need_ent;
        cap_nb = need_nb;
        cap_np = need_np;
        cap_hist = need_hist;
        cap_crel = need_crel;
        return MASP_HIP_OK;
    }
};

// ---- scratch of the batch-affine pre-reduction (device/msm_tree.hpp) for up to q proofs at a time ---------------------
// One byte arena serves every tree of a slot (its G1 and G2 MSMs run one after the other on the slot's stream): the views below
// are carved out of it anew by every msm_tree_enqueue.
// internal: the tree's scratch does not fit (device out of memory, or above masp_hip_options::bucket_tree_scratch_mb) — the caller
// halves the sub-batch or leaves the proofs to the XYZZ accumulation (msm_reduce_enqueue); never returned through the C ABI
static constexpr int MASP_HIP_E_TREE_SCRATCH = -100;
struct MsmTreeArena {
    uint8_t* p = nullptr;
    size_t cap = 0;
    size_t limit = 0;       // bytes this arena may take (0: whatever the device gives)
    uint64_t refused = 0;   // reservations that did not fit
    ~MsmTreeArena() { release(); }
    void release() {
        if (p) dev_free(p);
        p = nullptr;
        cap = 0;
    }
    int reserve(size_t bytes) {
        if (bytes <= cap) return MASP_HIP_OK;
        if (limit && bytes > limit) {
            ++refused;
            return MASP_HIP_E_TREE_SCRATCH;
        }
        release();  // (capacity is 0 from here on: a failed allocation must not leave it claiming memory)
        const hipError_t e = dev_malloc(&p, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            p = nullptr;
            ++refused;
            return MASP_HIP_E_TREE_SCRATCH;
        }
        HIP_TRY(e);
        cap = bytes;
        return MASP_HIP_OK;
    }
};
template <class O>
struct MsmTreeWs {
    typedef typename O::T F;
    static constexpr uint32_t BINV_C = 16, BINV_MID = 4096;  // chain length / lanes with an inversion of their own (batch_invert)
    MsmTreeArena own;
    MsmTreeArena* arena = &own;           // a slot points the trees of its workspaces at one arena
    uint32_t *D = nullptr, *Q = nullptr;  // [level][proof][nb + 1]
    uint2* rec = nullptr;  // pair records of the levels >= 1 (level 0 reads the digit list itself)
    F *pre = nullptr, *tp = nullptr, *tinv = nullptr, *bpre = nullptr, *btot = nullptr, *bitot = nullptr, *bpre2 = nullptr;
    F *px[2] = {nullptr, nullptr}, *py[2] = {nullptr, nullptr};  // points of the odd / even levels
    // what the last msm_tree_enqueue produced
    uint32_t q = 0, nb = 0, T = 0;
    size_t stride[2] = {0, 0};
    size_t pre_cap = 0;  // elements of the plane `pre`
    struct MsmProfile* prof = nullptr;  // set by msm_reduce_enqueue while a profiled MSM runs through the tree (MsmProfile::mark)

    int reserve(uint64_t E_ub, uint32_t nb, uint32_t q, uint32_t T);                 // [msm_tree_impl.hpp]
    void batch_invert(hipStream_t s, const F* in, uint32_t n, F* out);              // [msm_tree_impl.hpp]
    const F* points_x() const { return px[T & 1]; }
    const F* points_y() const { return py[T & 1]; }
    size_t point_stride() const { return stride[T & 1]; }
    const uint32_t* plan_D(uint32_t L) const { return D + (size_t)L * q * (nb + 1); }
};

// ---- workspace of the group arithmetic ---------------------------------------------------------------
#ifndef MASP_LONE_CHUNKS_PER_BUCKET
#define MASP_LONE_CHUNKS_PER_BUCKET 8
#endif
template <class O>
struct MsmWorkspace {
    static constexpr uint32_t CS_LOG = WSUM_G_LOG_MIN + WSUM_L_LOG;   // smallest weighted-sum chunk (buckets per workgroup): sizes S / T
    static constexpr uint32_t NCHUNKS = 1u << 18;     // lanes of the accumulation kernel (1024 waves x 4 per SIMD)

    MsmSortBuf sort;  // used unless the caller shares another workspace's sort
    // batch-affine pre-reduction of the bucket runs (batches only): tree_levels < 0 off, 0 = the default (4, fewer for very
    // short runs), else that many levels; tree_sub proofs go through the tree at a time (its scratch is ~0.4 GB per Spend proof)
    MsmTreeWs<O> tree;
    int tree_levels = 0;
    int tree_levels_shared = -1;  // levels of another workspace that reduces THIS workspace's sort (B2 over B1's): it is padded for both
    uint32_t tree_sub = 64;
    uint64_t tree_fallbacks = 0;  // proofs whose bucket runs went to the XYZZ accumulation because the tree's scratch did not fit
    uint32_t* startT = nullptr;  // [np][nb + 1] bucket offsets of the points the tree leaves
    size_t cap_nb = 0, cap_np = 0, cap_part = 0;  // cap_part: elements of `part` (np x (chunks + nb) of the largest launch)
    uint32_t *heavy = nullptr, *n_heavy = nullptr;
    Xyzz<O>*part = nullptr, *bkt = nullptr, *S[2] = {nullptr, nullptr}, *T = nullptr, *R[2] = {nullptr, nullptr};
    Xyzz<O>* tsum = nullptr;
    Xyzz<O>* hparts = nullptr;  // [min(np, 7)][MSM_HEAVY_SLOTS]: the shares of a lone proof's heavy buckets

    ~MsmWorkspace() { release(); }
    void release() {
        void* ptrs[] = {heavy, n_heavy, part, bkt, S[0], S[1], T, R[0], R[1], tsum, startT, hparts};
        for (void* p : ptrs)
            if (p) dev_free(p);
        heavy = n_heavy = startT = nullptr;
        part = bkt = S[0] = S[1] = T = R[0] = R[1] = tsum = hparts = nullptr;
        cap_nb = cap_np = cap_part = 0;
    }
    // lanes of the accumulation kernel per proof: ~2^18 across the whole batch.  Fewer, longer chunks mean fewer
    // partial sums to write and to gather (each extra partial costs a full XYZZ addition later).
    static uint32_t nchunks_for(uint32_t n, const MsmGeom& g, uint32_t np) {
        uint64_t ent = (uint64_t)n * g.W;
        // a batch: ~3 * 2^18 lanes across its proofs (six full rounds of the G1 kernel's 2 048 resident waves), between 3 072 and
        // 8 192 per proof (256 proofs per launch sequence: 3 072 measured +1.5 % over 8 192, 2 048 and 5 120 worse again)
        uint64_t lanes = np >= 8 ? std::min<uint64_t>(std::max<uint64_t>((3u << 18) / np, 3072), 1u << 13)
                                 : std::max<uint64_t>(NCHUNKS / std::max<uint32_t>(np, 1), 1u << 13);
        // lone proof: about eight chunks per bucket, so that a bucket's partials are few enough for one gather lane
        // (otherwise every bucket of a 12-bit-window MSM becomes a "heavy" bucket with a workgroup of its own)
        // ... except with so few buckets (a lone proof's B2 on 8-bit windows) that every bucket goes to the heavy-bucket
        // workgroups anyway: there a lane's chunk is a chain of dependent additions on an otherwise idle chip, so the digit
        // list is cut into one full round of waves
        if (np < 8) lanes = g.nb <= 256 ? std::max<uint64_t>((1u << MASP_LONE_NARROW_LANES_LOG) / np, 1u << 13) : std::min<uint64_t>(lanes, std::max<uint64_t>((uint64_t)(MASP_LONE_CHUNKS_PER_BUCKET) * g.nb, 1u << 13));
        return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(lanes, NCHUNKS), std::max<uint64_t>(ent, 1));
    }
    // room for `np` proofs of an n-point MSM with geometry g (every per-proof array is np-fold)
    int reserve(uint32_t n, const MsmGeom& g, uint32_t np) {
        // `part` is addressed with the launch's own stride (chunks + nb per proof), so what counts is the product np x (chunks +
        // nb): a batch of 64 proofs (8 192 chunks each) fits the buffer of a batch of 256 (3 072 each) without a new hipMalloc
        const size_t part_need = (size_t)np * ((size_t)nchunks_for(n, g, np) + g.nb);
        if ((size_t)g.nb <= cap_nb && np <= cap_np && part_need <= cap_part) return MASP_HIP_OK;
        size_t need_nb = std::max<size_t>(g.nb, cap_nb), need_np = std::max<size_t>(np, cap_np);
        // every batch shape of this base set: np x chunks <= 3 * 2^18 (nchunks_for), np x nb <= need_np x need_nb
        const size_t need_part = std::max(std::max(part_need, cap_part), need_np * need_nb + ((size_t)3 << 18));
        release();  // (capacities are 0 from here on: a failed allocation below must not leave them claiming memory)
        const size_t P = need_np;
        size_t chunks = (need_nb + (1u << CS_LOG) - 1) >> CS_LOG;
        auto alloc_all = [&]() -> int {
            HIP_TRY(dev_malloc(&heavy, P * 4 * need_nb));
            HIP_TRY(dev_malloc(&n_heavy, P * 4));
            HIP_TRY(dev_malloc(&part, sizeof(Xyzz<O>) * need_part));
            HIP_TRY(dev_malloc(&bkt, P * sizeof(Xyzz<O>) * need_nb));
            HIP_TRY(dev_malloc(&S[0], P * sizeof(Xyzz<O>) * chunks));
            HIP_TRY(dev_malloc(&S[1], P * sizeof(Xyzz<O>) * chunks));
            HIP_TRY(dev_malloc(&T, P * sizeof(Xyzz<O>) * chunks));
            HIP_TRY(dev_malloc(&R[0], P * sizeof(Xyzz<O>) * chunks));
            HIP_TRY(dev_malloc(&R[1], P * sizeof(Xyzz<O>) * chunks));
            HIP_TRY(dev_malloc(&tsum, P * sizeof(Xyzz<O>) * 32));
            HIP_TRY(dev_malloc(&startT, P * 4 * (need_nb + 1)));
            HIP_TRY(dev_malloc(&hparts, std::min<size_t>(P, 7) * sizeof(Xyzz<O>) * MSM_HEAVY_SLOTS));
            return MASP_HIP_OK;
        };
        if (int rc = alloc_all()) {
            release();
            return rc;
        }
        cap_nb = need_nb;
        cap_np = need_np;
        cap_part = need_part;
        return MASP_HIP_OK;
    }

    // per proof: reduce the `m` points at `src + p*src_stride` to one at `dst + p*dst_stride` (src must not be R[]).
    // Workgroup tree reductions (fan-in 256, 8 dependent additions per pass).                      [msm_impl.hpp]
    void reduce_to_one(hipStream_t s, uint32_t np, const Xyzz<O>* src, size_t src_stride, uint32_t m, Xyzz<O>* dst, size_t dst_stride,
                       size_t r_stride);
    // the same over OT::LANES lanes per point (FpQuadOps: a lone proof)
    template <class OT>
    void reduce_to_one_lanes(hipStream_t s, uint32_t np, const Xyzz<O>* src, size_t src_stride, uint32_t m, Xyzz<O>* dst, size_t dst_stride, size_t r_stride);
};

// Optional live timing of the dominant kernel (bucket accumulation) with HIP events on the launching stream.
struct MsmProfile {
    struct Rec {
        hipEvent_t e0, e1;
        uint64_t alg_bytes;
    };
    std::vector<Rec> recs;      // pending (recorded, not yet read)
    std::vector<Rec> pool;      // reusable event pairs
    double total_ms = 0;
    uint64_t launches = 0, alg_bytes = 0;
    // the stage by kernel group: an event behind every group of launches of the bucket stage (the tree's plan / records / copies,
    // its denominators pass, the shared inversions, its additions pass, the XYZZ accumulation of what is left); split_ms[g] = time
    // between the event in front of group g's launches and the one behind them, summed over the profiled launches
    enum { PH_START = -1, PH_PLAN = 0, PH_PASS1, PH_INV, PH_PASS2, PH_ACC, PH_N };
    struct Mark {
        hipEvent_t e;
        int tag;
    };
    std::vector<Mark> marks;
    std::vector<hipEvent_t> mark_pool;
    double split_ms[PH_N] = {0, 0, 0, 0, 0};
    void mark(hipStream_t s, int tag) {
        hipEvent_t e;
        if (!mark_pool.empty()) {
            e = mark_pool.back();
            mark_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            return;
        }
        hipEventRecord(e, s);
        marks.push_back({e, tag});
    }
    ~MsmProfile() {
        for (auto& r : recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        for (auto& r : pool) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        for (auto& m : marks) hipEventDestroy(m.e);
        for (auto& e : mark_pool) hipEventDestroy(e);
    }
    Rec acquire() {
        if (!pool.empty()) {
            Rec r = pool.back();
            pool.pop_back();
            return r;
        }
        Rec r;
        hipEventCreate(&r.e0);
        hipEventCreate(&r.e1);
        r.alg_bytes = 0;
        return r;
    }
    // call after the stream has been synchronised
    void collect() {
        for (auto& r : recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
                total_ms += ms;
                ++launches;
                alg_bytes += r.alg_bytes;
            }
            pool.push_back(r);
        }
        recs.clear();
        for (size_t i = 0; i < marks.size(); ++i) {
            float ms = 0;
            if (i && marks[i].tag >= 0 && hipEventElapsedTime(&ms, marks[i - 1].e, marks[i].e) == hipSuccess) split_ms[marks[i].tag] += ms;
            mark_pool.push_back(marks[i].e);
        }
        marks.clear();
    }
    void reset() {
        collect();
        total_ms = 0;
        launches = 0;
        alg_bytes = 0;
        for (double& v : split_ms) v = 0;
    }
};

// Counting sort of the signed window digits of `np` scalar vectors (n scalars each) by bucket, on stream `s`.
// scalars_p = d_scalars + p * scalar_stride (u32 units), n x 8 canonical LE limbs each.  No host synchronisation.   [k_msm_sort.hip]
// pad_log: see MsmSortBuf (0: packed runs).
int msm_sort_enqueue(hipStream_t s, uint32_t n, const MsmGeom& g, MsmSortBuf& sb, const uint32_t* d_scalars, size_t scalar_stride, uint32_t np,
                     uint32_t pad_log = 0);

// runs of the digit list padded to multiples of 2^MASP_TREE_PAD_LOG entries when a tree follows: that many levels need no records
#ifndef MASP_TREE_PAD_LOG
#define MASP_TREE_PAD_LOG 2
#endif
// levels of the batch-affine tree for an MSM of `np` proofs over a base set with n_eff non-trivial scalars (0: none).  The tree
// halves the bucket runs T times: 1 - 2^-T of the additions at 5.8 instead of 9.6 products.  Every level costs two passes over
// its points and a latency-bound shared inversion, so T stays small (measured on 256 Spend proofs, 4 batches in flight,
// sub-batches of 64: T = 3 +7.5 %, 4 +9 %, 5 +8.5 %, 6 / 8 +3 % over the XYZZ accumulation alone); runs shorter than 32 points
// (mean: n_eff W digits over nb buckets) get fewer levels.  opt: < 0 off, 0 automatic, else that many (at most 12).
inline uint32_t msm_tree_levels(uint32_t n_eff, const MsmGeom& g, uint32_t np, int opt) {
    if (np < 8 || opt < 0) return 0;  // a lone proof: latency regime, short chains matter more than total work
    if (opt > 0) return std::min<uint32_t>((uint32_t)opt, 12u);
    const uint64_t mean = std::max<uint64_t>((uint64_t)std::max(n_eff, 1u) * msm_mean_digits_x16(g) / 16 / g.nb, 1);
    uint32_t lg = 0;
    while (((uint64_t)1 << lg) < mean) ++lg;
    return std::min(4u, lg > 1 ? lg - 1 : 0u);
}

// The dominant kernel on its own: bucket accumulation of the sorted digit list (k_msm_accumulate<O>).   [msm_acc_impl.hpp]
template <class O>
void msm_launch_accumulate(hipStream_t s, const TabRow<O>* tab, const uint32_t* sorted, size_t ent_stride, const uint32_t* start, uint32_t nb,
                           uint32_t nchunks, Xyzz<O>* part, uint32_t np);

// Batch-affine pre-reduction (device/msm_tree.hpp) of proofs [p0, p0 + q) of the sort `sb`: T levels.   [msm_tree_impl.hpp]
template <class O, int BYTES>
int msm_tree_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, const MsmSortBuf& sb, MsmTreeWs<O>& tw, uint32_t p0, uint32_t q, uint32_t T);
// k_msm_accumulate over explicit points instead of gathered table rows                                   [msm_tree_impl.hpp]
template <class O>
void msm_launch_accumulate_pts(hipStream_t s, const typename O::T* xs, const typename O::T* ys, size_t pt_stride, const uint32_t* start, uint32_t nb,
                               uint32_t nchunks, Xyzz<O>* part, uint32_t np);

// Bucket accumulation + reduction of the MSM whose digits were sorted into `sb` (same n, geometry and batch size):
// result p at d_out + p * out_stride.  No host synchronisation.                                    [msm_impl.hpp]
template <class O, int BYTES>
int msm_reduce_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, const MsmSortBuf& sb, MsmWorkspace<O>& ws, Xyzz<O>* d_out, size_t out_stride,
                       MsmProfile* prof = nullptr);

// Enqueue, for each of `np` proofs p, sum_i scalars_p[i] * P_i on stream `s` — one launch per stage for the whole batch.
// scalars_p = d_scalars + p * scalar_stride (u32 units), n x 8 canonical LE limbs each; result p at d_out + p * out_stride.
// No host synchronisation.
template <class O, int BYTES>
int msm_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, MsmWorkspace<O>& ws, const uint32_t* d_scalars, size_t scalar_stride,
                Xyzz<O>* d_out, size_t out_stride, uint32_t np, MsmProfile* prof = nullptr) {
    if (np == 0) return MASP_HIP_OK;
    if (B.n == 0) {
        HIP_TRY(hipMemset2DAsync(d_out, out_stride * sizeof(Xyzz<O>), 0, sizeof(Xyzz<O>), np, s));  // infinity (ZZ = 0) for every proof
        return MASP_HIP_OK;
    }
    // the tree wants runs of even length
    const bool tree = msm_tree_levels(B.n_eff, B.g, np, ws.tree_levels) || msm_tree_levels(B.n_eff, B.g, np, ws.tree_levels_shared);
    int rc = msm_sort_enqueue(s, B.n, B.g, ws.sort, d_scalars, scalar_stride, np, tree ? (uint32_t)MASP_TREE_PAD_LOG : 0u);
    if (rc) return rc;
    return msm_reduce_enqueue(s, B, ws.sort, ws, d_out, out_stride, prof);
}

typedef MsmBases<FpOps, 96> BasesG1;
typedef MsmBases<Fp2Ops, 192> BasesG2;

}  // namespace masp
