// Host-side driver of the MSM kernels (device/msm.cuh): owns the precomputed window tables of one
// base set and a reusable workspace, and enqueues one MSM on a HIP stream without any host sync.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "device/msm.cuh"
#include "util.h"

namespace masp {

// ---- base set: T[j][i] = 2^(c j) P_i -------------------------------------------------------------
template <class O, int BYTES>
struct MsmBases {
    MsmGeom g{};
    uint32_t n = 0;
    Affine<O>* tab = nullptr;  // W * n rows
    int import_status = 0;     // PT_* bits seen while decoding

    ~MsmBases() { release(); }
    void release() {
        if (tab) hipFree(tab);
        tab = nullptr;
    }
    // Window width by the number of scalars expected to be neither 0 nor 1 (`n_eff`; the caller knows the witness
    // statistics of its circuit, a generic caller passes n): per non-trivial scalar the accumulation costs W = 256/c
    // mixed additions, per bucket the gather + weighted sum cost ~5.5 full additions.
    static MsmGeom pick_geom(uint32_t n_eff) {
        int c = n_eff >= (1u << 16) ? 16 : n_eff >= (1u << 12) ? 12 : n_eff >= (1u << 8) ? 10 : 7;
        const char* e = getenv("MASP_HIP_MSM_C");
        if (e) c = atoi(e);
        return msm_geom(c);
    }
    // raw: device pointer to n uncompressed points (bellman wire format)
    int load_device(const uint8_t* d_raw, uint32_t n_, hipStream_t s, uint32_t n_eff = 0xffffffffu, int force_c = 0) {
        release();
        n = n_;
        g = force_c ? msm_geom(force_c) : pick_geom(std::min(n_eff, n_));
        if (n == 0) return MASP_HIP_OK;
        HIP_TRY(hipMalloc(&tab, sizeof(Affine<O>) * (size_t)g.W * n));
        int* d_status;
        HIP_TRY(hipMalloc(&d_status, sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), s));
        dim3 grid((n + 63) / 64), block(64);
        hipLaunchKernelGGL((k_msm_import<O, BYTES>), grid, block, 0, s, d_raw, tab, n, d_status);
        hipLaunchKernelGGL((k_msm_precompute<O>), grid, block, 0, s, tab, n, g.c, g.W);
        HIP_TRY(hipMemcpyAsync(&import_status, d_status, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        hipFree(d_status);
        return MASP_HIP_OK;
    }
    int load_host(const uint8_t* raw, uint32_t n_, hipStream_t s, uint32_t n_eff = 0xffffffffu, int force_c = 0) {
        uint8_t* d_raw = nullptr;
        if (n_) {
            HIP_TRY(hipMalloc(&d_raw, (size_t)n_ * BYTES));
            HIP_TRY(hipMemcpyAsync(d_raw, raw, (size_t)n_ * BYTES, hipMemcpyHostToDevice, s));
        }
        int rc = load_device(d_raw, n_, s, n_eff, force_c);
        if (d_raw) hipFree(d_raw);
        return rc;
    }
};

// ---- sort buffers (curve-independent): one counting sort can feed several MSMs over the same scalars ---------
struct MsmSortBuf {
    size_t cap_ent = 0, cap_nb = 0, cap_np = 0, cap_ng = 0;
    uint32_t *sorted = nullptr, *hist_wg = nullptr, *start = nullptr;
    // what the last msm_sort_enqueue produced (consumed by msm_reduce_enqueue)
    uint32_t n = 0, np = 0;
    MsmGeom g{};

    ~MsmSortBuf() { release(); }
    void release() {
        void* ptrs[] = {sorted, hist_wg, start};
        for (void* p : ptrs)
            if (p) hipFree(p);
        sorted = hist_wg = start = nullptr;
        cap_ent = cap_nb = cap_np = cap_ng = 0;
    }
    // scalar ranges (= sorting workgroups) per proof: enough to occupy the chip across the batch, not more
    static uint32_t ranges_for(uint32_t n, uint32_t np) {
        uint32_t ng = std::max(1u, 512u / std::max(np, 1u));
        ng = std::min(ng, 64u);
        return std::max(1u, std::min(ng, (n + 1023) / 1024));
    }
    int reserve(uint32_t n_, const MsmGeom& g_, uint32_t np_) {
        size_t need_ent = (size_t)n_ * g_.W, need_ng = ranges_for(n_, np_);
        if (need_ent <= cap_ent && (size_t)g_.nb <= cap_nb && np_ <= cap_np && need_ng <= cap_ng) return MASP_HIP_OK;
        need_ent = std::max(need_ent, cap_ent);
        size_t need_nb = std::max<size_t>(g_.nb, cap_nb), need_np = std::max<size_t>(np_, cap_np);
        need_ng = std::max(need_ng, cap_ng);
        release();
        cap_ent = need_ent;
        cap_nb = need_nb;
        cap_np = need_np;
        cap_ng = need_ng;
        HIP_TRY(hipMalloc(&sorted, cap_np * 4 * std::max<size_t>(cap_ent, 1)));
        HIP_TRY(hipMalloc(&hist_wg, cap_np * 4 * cap_ng * cap_nb));
        HIP_TRY(hipMalloc(&start, cap_np * 4 * (cap_nb + 1)));
        return MASP_HIP_OK;
    }
};

// ---- workspace of the group arithmetic ---------------------------------------------------------------
template <class O>
struct MsmWorkspace {
    static constexpr uint32_t CS_LOG = WSUM_G_LOG_MIN + WSUM_L_LOG;   // smallest weighted-sum chunk (buckets per workgroup): sizes S / T
    static constexpr uint32_t NCHUNKS = 1u << 18;     // lanes of the accumulation kernel (1024 waves x 4 per SIMD)

    MsmSortBuf sort;  // used unless the caller shares another workspace's sort
    size_t cap_nb = 0, cap_np = 0, cap_chunks = 0;
    uint32_t *heavy = nullptr, *n_heavy = nullptr;
    Xyzz<O>*part = nullptr, *bkt = nullptr, *S[2] = {nullptr, nullptr}, *T = nullptr, *R[2] = {nullptr, nullptr};
    Xyzz<O>* tsum = nullptr;

    ~MsmWorkspace() { release(); }
    void release() {
        void* ptrs[] = {heavy, n_heavy, part, bkt, S[0], S[1], T, R[0], R[1], tsum};
        for (void* p : ptrs)
            if (p) hipFree(p);
        heavy = n_heavy = nullptr;
        part = bkt = S[0] = S[1] = T = R[0] = R[1] = tsum = nullptr;
        cap_nb = cap_np = cap_chunks = 0;
    }
    // lanes of the accumulation kernel per proof: ~2^18 across the whole batch.  Fewer, longer chunks mean fewer
    // partial sums to write and to gather (each extra partial costs a full XYZZ addition later).
    static uint32_t nchunks_for(uint32_t n, const MsmGeom& g, uint32_t np) {
        uint64_t ent = (uint64_t)n * g.W;
        uint64_t lanes = std::max<uint64_t>(NCHUNKS / std::max<uint32_t>(np, 1), 1u << 13);
        // lone proof: about eight chunks per bucket, so that a bucket's partials are few enough for one gather lane
        // (otherwise every bucket of a 12-bit-window MSM becomes a "heavy" bucket with a workgroup of its own)
        if (np < 8) lanes = std::min<uint64_t>(lanes, std::max<uint64_t>(8ull * g.nb, 1u << 13));
        static const int forced = [] {  // experiment knob, read once per process
            const char* e = getenv("MASP_HIP_MSM_CHUNKS");
            return e ? std::max(1, atoi(e)) : 0;
        }();
        if (forced) lanes = (uint64_t)forced;
        return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(lanes, NCHUNKS), std::max<uint64_t>(ent, 1));
    }
    // room for `np` proofs of an n-point MSM with geometry g (every per-proof array is np-fold)
    int reserve(uint32_t n, const MsmGeom& g, uint32_t np) {
        size_t need_chunks = nchunks_for(n, g, np);
        if ((size_t)g.nb <= cap_nb && np <= cap_np && need_chunks <= cap_chunks) return MASP_HIP_OK;
        need_chunks = std::max(need_chunks, cap_chunks);
        size_t need_nb = std::max<size_t>(g.nb, cap_nb), need_np = std::max<size_t>(np, cap_np);
        release();
        cap_nb = need_nb;
        cap_np = need_np;
        cap_chunks = need_chunks;
        const size_t P = cap_np;
        size_t chunks = (cap_nb + (1u << CS_LOG) - 1) >> CS_LOG;
        HIP_TRY(hipMalloc(&heavy, P * 4 * cap_nb));
        HIP_TRY(hipMalloc(&n_heavy, P * 4));
        HIP_TRY(hipMalloc(&part, P * sizeof(Xyzz<O>) * (cap_chunks + cap_nb)));
        HIP_TRY(hipMalloc(&bkt, P * sizeof(Xyzz<O>) * cap_nb));
        HIP_TRY(hipMalloc(&S[0], P * sizeof(Xyzz<O>) * chunks));
        HIP_TRY(hipMalloc(&S[1], P * sizeof(Xyzz<O>) * chunks));
        HIP_TRY(hipMalloc(&T, P * sizeof(Xyzz<O>) * chunks));
        HIP_TRY(hipMalloc(&R[0], P * sizeof(Xyzz<O>) * chunks));
        HIP_TRY(hipMalloc(&R[1], P * sizeof(Xyzz<O>) * chunks));
        HIP_TRY(hipMalloc(&tsum, P * sizeof(Xyzz<O>) * 32));
        return MASP_HIP_OK;
    }

    // per proof: reduce the `m` points at `src + p*src_stride` to one at `dst + p*dst_stride` (src must not be R[]).
    // Workgroup tree reductions (fan-in 256, 8 dependent additions per pass).
    void reduce_to_one(hipStream_t s, uint32_t np, const Xyzz<O>* src, size_t src_stride, uint32_t m, Xyzz<O>* dst, size_t dst_stride,
                       size_t r_stride) {
        int flip = 0;
        const Xyzz<O>* cur = src;
        size_t cur_stride = src_stride;
        while (true) {
            uint32_t outn = (m + 255) / 256;
            Xyzz<O>* out = outn == 1 ? dst : R[flip];
            size_t out_stride = outn == 1 ? dst_stride : r_stride;
            hipLaunchKernelGGL((k_xyzz_reduce_block<O>), dim3(outn, np), dim3(256), 256 * sizeof(Xyzz<O>), s, cur, cur_stride, m, out, out_stride);
            if (outn == 1) break;
            cur = out;
            cur_stride = out_stride;
            m = outn;
            flip ^= 1;
        }
    }
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
    ~MsmProfile() {
        for (auto& r : recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        for (auto& r : pool) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
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
    }
    void reset() {
        collect();
        total_ms = 0;
        launches = 0;
        alg_bytes = 0;
    }
};

// Counting sort of the signed window digits of `np` scalar vectors (n scalars each) by bucket, on stream `s`.
// scalars_p = d_scalars + p * scalar_stride (u32 units), n x 8 canonical LE limbs each.  No host synchronisation.
static inline int msm_sort_enqueue(hipStream_t s, uint32_t n, const MsmGeom& g, MsmSortBuf& sb, const uint32_t* d_scalars, size_t scalar_stride,
                                   uint32_t np) {
    if (g.c < 2 || g.c > 16) {
        last_hip_error() = "MSM window width must be 2..16 bits (the bucket histogram lives in LDS)";
        return MASP_HIP_E_INVALID_ARG;
    }
    int rc = sb.reserve(n, g, np);
    if (rc) return rc;
    sb.n = n;
    sb.np = np;
    sb.g = g;
    static bool lds_ok = [] {
        int bytes = 4 << 15;
        return hipFuncSetAttribute((const void*)k_msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute((const void*)k_msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    }();
    if (!lds_ok) {
        last_hip_error() = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
        return MASP_HIP_E_HIP;
    }
    const uint32_t ng = MsmSortBuf::ranges_for(n, np), nb = g.nb;
    hipLaunchKernelGGL(k_msm_hist, dim3(ng, np), dim3(MSM_SORT_THREADS), 4 * nb, s, d_scalars, scalar_stride, n, g, ng, sb.hist_wg);
    hipLaunchKernelGGL(k_msm_offsets, dim3(1, np), dim3(1024), 0, s, sb.hist_wg, ng, nb, sb.start);
    hipLaunchKernelGGL(k_msm_scatter, dim3(ng, np), dim3(MSM_SORT_THREADS), 4 * nb, s, d_scalars, scalar_stride, n, g, ng, sb.hist_wg, sb.start,
                       sb.sorted);
    return MASP_HIP_OK;
}

// Bucket accumulation + reduction of the MSM whose digits were sorted into `sb` (same n, geometry and batch size):
// result p at d_out + p * out_stride.  No host synchronisation.
template <class O, int BYTES>
int msm_reduce_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, const MsmSortBuf& sb, MsmWorkspace<O>& ws, Xyzz<O>* d_out, size_t out_stride,
                       MsmProfile* prof = nullptr) {
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
        static bool lds_ok = [] {
            int bytes = 256 * (int)sizeof(Xyzz<O>);
            return hipFuncSetAttribute((const void*)k_xyzz_reduce_block<O>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        }();
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
    }
    hipLaunchKernelGGL((k_msm_accumulate<O>), dim3((nchunks + 63) / 64, np), dim3(64), 0, s, B.tab, sb.sorted, (size_t)total, sb.start, nb,
                       nchunks, ws.part);
    if (prof) {
        hipEventRecord(rec.e1, s);
        prof->recs.push_back(rec);
    }
    const bool lone = np < 8;  // latency regime: short chains matter more than total work
    hipLaunchKernelGGL((k_msm_bucket_gather<O>), dim3((nb + 63) / 64, np), dim3(64), 0, s, ws.part, sb.start, nb, nchunks, ws.bkt, ws.heavy,
                       ws.n_heavy, lone ? 12u : 24u);
    // (with a lone proof the chunks are short and most buckets of a narrow-window MSM count as heavy: give them the chip)
    // Usually ONE workgroup per proof has work here (bucket 0).  Workgroups go to the 8 XCDs round-robin by linear id
    // x + gridDim.x * y, so with gridDim.x a multiple of 8 every proof's working workgroup (x = 0) would land on the same
    // XCD (measured: 9.9 ms instead of 4.1 ms for the G2 launch of a 64-proof batch): keep gridDim.x odd.
    const uint32_t heavy_blocks = std::min<uint32_t>(std::max<uint32_t>(4096u / np, 16u), nb) | 1u;
    hipLaunchKernelGGL((k_msm_bucket_heavy<O, 256>), dim3(heavy_blocks, np), dim3(256), 0, s, ws.part, sb.start, nb, nchunks, ws.bkt, ws.heavy,
                       ws.n_heavy);
    // weighted sum by levels of (G x 128)-bucket workgroups; G = 16 buckets per lane in a batch (least work per bucket:
    // the latency of the launches with few buckets hides behind the other batches in flight; choosing G = 4 for those
    // was measured 2 % slower), 4 for a lone proof (shortest dependent chain)
    const uint32_t g_log = lone ? WSUM_G_LOG_MIN : 4;
    const uint32_t cs = 1u << (g_log + WSUM_L_LOG);
    const size_t st_stride = (nb + cs - 1) / cs;  // level-0 chunk count bounds every later level
    const Xyzz<O>* bk = ws.bkt;
    size_t bk_stride = nb;
    uint32_t m = nb, off = 1;
    int level = 0, flip = 0;
    do {
        uint32_t chunks = (m + cs - 1) / cs;
        if (g_log == 4)
            hipLaunchKernelGGL((k_msm_wsum_level<O, 4>), dim3(chunks, np), dim3(WSUM_L), 0, s, bk, bk_stride, m, off, ws.S[flip], ws.T, st_stride);
        else
            hipLaunchKernelGGL((k_msm_wsum_level<O, WSUM_G_LOG_MIN>), dim3(chunks, np), dim3(WSUM_L), 0, s, bk, bk_stride, m, off, ws.S[flip],
                               ws.T, st_stride);
        ws.reduce_to_one(s, np, ws.T, st_stride, chunks, ws.tsum + level, 32, st_stride);
        bk = ws.S[flip];
        bk_stride = st_stride;
        flip ^= 1;
        m = chunks;
        off = 0;
        ++level;
    } while (m > 1);
    hipLaunchKernelGGL((k_msm_combine<O>), dim3(1, np), dim3(64), 0, s, ws.tsum, level, (int)(g_log + WSUM_L_LOG), d_out, out_stride);
    return MASP_HIP_OK;
}

// Enqueue, for each of `np` proofs p, sum_i scalars_p[i] * P_i on stream `s` — one launch per stage for the whole batch.
// scalars_p = d_scalars + p * scalar_stride (u32 units), n x 8 canonical LE limbs each; result p at d_out + p * out_stride.
// No host synchronisation.
template <class O, int BYTES>
int msm_enqueue(hipStream_t s, const MsmBases<O, BYTES>& B, MsmWorkspace<O>& ws, const uint32_t* d_scalars, size_t scalar_stride,
                Xyzz<O>* d_out, size_t out_stride, uint32_t np, MsmProfile* prof = nullptr) {
    if (np == 0) return MASP_HIP_OK;
    if (B.n == 0) {
        for (uint32_t p = 0; p < np; ++p) HIP_TRY(hipMemsetAsync(d_out + p * out_stride, 0, sizeof(Xyzz<O>), s));  // infinity (ZZ = 0)
        return MASP_HIP_OK;
    }
    int rc = msm_sort_enqueue(s, B.n, B.g, ws.sort, d_scalars, scalar_stride, np);
    if (rc) return rc;
    return msm_reduce_enqueue(s, B, ws.sort, ws, d_out, out_stride, prof);
}

}  // namespace masp
