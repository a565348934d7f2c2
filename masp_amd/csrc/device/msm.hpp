// Multi-scalar multiplication  sum_i k_i * P_i  over BLS12-381 G1 / G2 for gfx950.
//
// Replaces bellperson's `multiexp` (nam-bellperson 0.26.6-nam.1, un-vendored; the six G1 calls and
// two G2 calls per proof listed in SURVEY.md A.3 step 4, reached from
// /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252) and the `<G1>_multiexp` kernels of
// nam-ec-gpu-gen.  Not a port of either: the design spends HBM capacity (288 GB) to
// remove ALU work, which is what bounds this path on CDNA4.
//
//   load time   T[j][i] = 2^(c*j) * P_i  for every window j (affine, Montgomery), so that all windows
//               share ONE bucket set and no per-window Horner doublings exist at prove time.
//   prove time  (1) hist     : signed c-bit digits of every scalar, counted per bucket in LDS (zero scalars, 38 % of a
//                              Spend witness, vanish; scalars equal to 1, 33 %, are one entry of bucket 0).
//               (2) offsets  : prefix sums over buckets and scalar ranges.
//               (3) scatter  : counting sort of (table row, sign) by bucket, positions from LDS counters.
//               (4) accumulate: the SORTED list is cut into equal chunks, one lane per chunk, mixed XYZZ
//                              additions of gathered table rows; a lane flushes a partial sum whenever its
//                              chunk crosses a bucket boundary.  Every lane does the same number of
//                              additions whatever the digit distribution (the 1-bucket of a MASP witness
//                              holds ~33 000 points, a uniform bucket ~64).
//               (5) gather   : one lane per bucket adds the few partials of its bucket; buckets spread over
//                              many chunks are finished by one wave each (LDS tree).
//               (6) reduce   : sum_w w*B_w by a log-depth LDS suffix scan + tree per 256 buckets.
// Group arithmetic is exact, so the result (after affine normalisation) is bit-identical to any
// other evaluation order — which is what lets the sort be unstable and the atomics unordered.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "quad.hpp"
#include "oct.hpp"
#include "io.hpp"
#include "msm_geom.h"

namespace masp {

// Every prove-time kernel below is launched with gridDim.y = number of proofs in the batch: proof p works on
// `ptr + p * stride` of each per-proof array (the window tables are shared).  One launch per stage for the whole
// batch keeps the latency-bound reduction stages wide enough to fill the chip.
#define MSM_P (blockIdx.y)

// ---- load time ----------------------------------------------------------------------------------
// raw uncompressed bytes -> T[0][i]; status word collects PT_* bits (infinity is legal in a generic
// MSM and contributes nothing).
template <class O, int BYTES>
__global__ void k_msm_import(const uint8_t* __restrict__ raw, TabRow<O>* __restrict__ tab, uint32_t n, int* __restrict__ status) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<O> p;
    int st;
    if constexpr (BYTES == 96)
        st = g1_read_uncompressed(raw + (size_t)i * 96, p);
    else
        st = g2_read_uncompressed(raw + (size_t)i * 192, p);
    if (st & ~PT_INFINITY) {
        atomicOr(status, st);
        p.x = O::zero();
        p.y = O::zero();
    } else if (st & PT_INFINITY) {
        atomicOr(status, PT_INFINITY);
    }
    tab[i].p = p;
}
// T[j][i] = 2^c * T[j-1][i]
template <class O>
__global__ void __launch_bounds__(64) k_msm_precompute(TabRow<O>* __restrict__ tab, uint32_t n, int c, int W) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<O> p = tab[i].p;
    for (int j = 1; j < W; ++j) {
        Xyzz<O> q = xyzz_dbl_affine(p);
        for (int k = 1; k < c; ++k) q = xyzz_dbl(q);
        p = xyzz_to_affine(q);
        tab[(size_t)j * n + i].p = p;
    }
}

// ---- (4) accumulate: device/msm_acc.hpp (its own translation unit) -------------------------------------

// ---- (5) gather: bucket b = sum of its partials part[c + b], c over the chunks its entries touch ---------
// heavy_span: a bucket with at least this many partials is left to k_msm_bucket_heavy (24 in a batch, where work
// counts; 12 for a lone proof, where the longest serial chain counts too — but a wave per bucket costs 64 lanes)
#ifndef MASP_TAIL_MIN_WAVES
#define MASP_TAIL_MIN_WAVES 1
#endif
// The tail kernels are written over O::LANES lanes per point (1; 2 for Fp2PairOps: the even lane holds the c0 halves of the four
// coordinates, the odd lane the c1 halves — G2 at the register footprint of G1).  A stored point is Xyzz<O::Base>.
// (h = lane in the group of O::LANES lanes that share a point; a stored element is read in O::PARTS parts: 1 = whole — every lane of
// a replicated group reads it —, 2 = one half of every Fp2 per lane, whichever pair of the group the lane belongs to)
template <class O>
__device__ __forceinline__ Xyzz<O> xyzz_load(const Xyzz<typename O::Base>* __restrict__ p, uint32_t h) {
    if constexpr (O::PARTS == 1) {
        const Xyzz<typename O::Base> v = *p;
        return reinterpret_cast<const Xyzz<O>&>(v);
    } else {
        const typename O::T* q = reinterpret_cast<const typename O::T*>(p);
        const uint32_t part = h % O::PARTS;
        Xyzz<O> r;
        r.X = q[part];
        r.Y = q[O::PARTS + part];
        r.ZZ = q[2 * O::PARTS + part];
        r.ZZZ = q[3 * O::PARTS + part];
        return r;
    }
}
template <class O>
__device__ __forceinline__ void xyzz_store(Xyzz<typename O::Base>* __restrict__ p, const Xyzz<O>& v, uint32_t h) {
    if constexpr (O::PARTS == 1) {
        if (h == 0) *p = reinterpret_cast<const Xyzz<typename O::Base>&>(v);
    } else {
        if (h >= O::PARTS) return;  // (a replicated group: its first pair stores)
        typename O::T* q = reinterpret_cast<typename O::T*>(p);
        q[h] = v.X;
        q[O::PARTS + h] = v.Y;
        q[2 * O::PARTS + h] = v.ZZ;
        q[3 * O::PARTS + h] = v.ZZZ;
    }
}
// launch with 64 lanes per workgroup: 64 / LANES buckets
template <class O>
__global__ void __launch_bounds__(64, MASP_TAIL_MIN_WAVES)
k_msm_bucket_gather(const Xyzz<typename O::Base>* __restrict__ part, const uint32_t* __restrict__ start, uint32_t nb, uint32_t nchunks,
                    Xyzz<typename O::Base>* __restrict__ bkt, uint32_t* __restrict__ heavy, uint32_t* __restrict__ n_heavy, uint32_t heavy_span) {
    constexpr uint32_t LN = O::LANES;
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) / LN, h = threadIdx.x % LN;
    if (b >= nb) return;
    part += (size_t)MSM_P * ((size_t)nchunks + nb);
    start += (size_t)MSM_P * (nb + 1);
    bkt += (size_t)MSM_P * nb;
    heavy += (size_t)MSM_P * nb;
    n_heavy += MSM_P;
    const uint32_t s0 = start[b], s1 = start[b + 1];
    Xyzz<O> acc = xyzz_inf<O>();
    if (s1 > s0) {
        const uint32_t K = msm_chunk_len(start[nb], nchunks);
        const uint32_t c0 = s0 / K, c1 = (s1 - 1) / K;
        if (c1 - c0 >= heavy_span) {
            if (h == 0) heavy[atomicAdd(n_heavy, 1u)] = b;
            return;  // written by k_msm_bucket_heavy
        }
        acc = xyzz_load<O>(part + c0 + b, h);
        for (uint32_t c = c0 + 1; c <= c1; ++c) xyzz_add_nc(acc, xyzz_load<O>(part + c + b, h));
    }
    xyzz_store<O>(bkt + b, acc, h);
}
// The same for a batch, in two kinds of workgroup of ONE launch.  After the bucket tree a chunk holds ~50 points and a bucket ~5, so nine
// buckets in ten lie inside one chunk and have ONE partial — but among the 64 buckets of a wave some always straddle a chunk boundary, and
// the wave paid a whole group addition (exec-masked for the rest) per bucket-lane: 1.4e9 wave instructions per Spend batch for ~0.1
// additions per bucket.  Here workgroups [0, nbw) copy the single partials (one lane per bucket: no arithmetic; heavy buckets go to their
// list as before), workgroups [nbw, nbw + ncw) take one CHUNK BOUNDARY per lane: lane c (1 <= c < nchunks) owns the bucket that begins in
// chunk c - 1 and runs on into chunk c, and adds its partials c - 1 .. c1 — the additions now fill their waves.  grid (nbw + ncw, np).
template <class O>
__global__ void __launch_bounds__(64, MASP_TAIL_MIN_WAVES)
k_msm_bucket_gather_split(const Xyzz<typename O::Base>* __restrict__ part, const uint32_t* __restrict__ start, uint32_t nb, uint32_t nchunks,
                          Xyzz<typename O::Base>* __restrict__ bkt, uint32_t* __restrict__ heavy, uint32_t* __restrict__ n_heavy, uint32_t heavy_span,
                          uint32_t nbw) {
    constexpr uint32_t LN = O::LANES;
    const uint32_t h = threadIdx.x % LN;
    part += (size_t)MSM_P * ((size_t)nchunks + nb);
    start += (size_t)MSM_P * (nb + 1);
    bkt += (size_t)MSM_P * nb;
    heavy += (size_t)MSM_P * nb;
    n_heavy += MSM_P;
    const uint32_t total = start[nb];
    const uint32_t K = msm_chunk_len(total, nchunks);
    if (blockIdx.x < nbw) {
        const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) / LN;
        if (b >= nb) return;
        const uint32_t s0 = start[b], s1 = start[b + 1];
        if (s1 <= s0) {
            xyzz_store<O>(bkt + b, xyzz_inf<O>(), h);
            return;
        }
        const uint32_t c0 = s0 / K, c1 = (s1 - 1) / K;
        if (c1 - c0 >= heavy_span) {
            if (h == 0) heavy[atomicAdd(n_heavy, 1u)] = b;  // written by k_msm_bucket_heavy
            return;
        }
        if (c1 == c0) xyzz_store<O>(bkt + b, xyzz_load<O>(part + c0 + b, h), h);
        return;  // (c1 > c0: the lane of chunk boundary c0 + 1 below)
    }
    const uint32_t c = ((blockIdx.x - nbw) * blockDim.x + threadIdx.x) / LN + 1u;
    if (c >= nchunks || (uint64_t)c * K >= total) return;
    // the bucket that holds position c K - 1, the last of chunk c - 1: largest b with start[b] <= c K - 1
    const uint32_t pos = c * K - 1u;
    uint32_t b = 0, span = nb;
    while (span > 1) {
        const uint32_t half = span >> 1;
        if (start[b + half] <= pos) b += half;
        span -= half;
    }
    const uint32_t s0 = start[b], s1 = start[b + 1];
    if (s1 <= c * K) return;                 // the bucket ends with chunk c - 1: nothing straddles this boundary
    const uint32_t c0 = s0 / K, c1 = (s1 - 1) / K;
    if (c0 != c - 1u || c1 - c0 >= heavy_span) return;   // it began earlier (the lane of ITS first boundary adds it up), or it is a heavy bucket
    Xyzz<O> acc = xyzz_load<O>(part + c0 + b, h);
    for (uint32_t k = c; k <= c1; ++k) xyzz_add_nc(acc, xyzz_load<O>(part + k + b, h));
    xyzz_store<O>(bkt + b, acc, h);
}
// value of lane (lane + d) of the wave, limb by limb (a point is 36 / 96 dwords: noise next to one group addition)
template <class O>
__device__ __forceinline__ Xyzz<O> xyzz_shfl_down(const Xyzz<O>& p, int d) {
    Xyzz<O> r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(Xyzz<O>) / 4; ++i) dst[i] = (uint32_t)__shfl_down((int)src[i], d, 64);
    return r;
}
// A heavy bucket (the unit scalars of a witness put ~33 000 entries into bucket 0) is finished by one workgroup of
// THREADS / 64 waves: strided serial sums, a shuffle tree inside each wave, the waves' values through LDS.
// (THREADS = 256 measured faster than a single wave in both regimes: one bucket of ~650 partials per proof in a batch,
// thousands of buckets with ~80 partials each for a lone proof.)
// SPLIT > 1 (a lone proof): a heavy bucket is shared by SPLIT workgroups, each summing an equal share of its partials into
// hparts[hb * SPLIT + share]; k_msm_heavy_join adds the shares.  The unit scalars of a witness all sit in ONE bucket — 2 500 partials
// of a lone proof's b_g2 MSM where every other bucket has 480 — and as one workgroup's 78 dependent additions that bucket alone was 0.9
// of the 4 ms a lone Spend proof takes (profiles/r05_lone_b2_msm_kernels_in_isolation.txt).  Heavy buckets beyond MSM_HEAVY_SLOTS / SPLIT
// (never more than a handful in practice) are summed by their first workgroup alone, as before.
template <class O, uint32_t THREADS, uint32_t SPLIT = 1>
__global__ void __launch_bounds__(THREADS, MASP_TAIL_MIN_WAVES)
k_msm_bucket_heavy(const Xyzz<typename O::Base>* __restrict__ part, const uint32_t* __restrict__ start, uint32_t nb, uint32_t nchunks,
                   Xyzz<typename O::Base>* __restrict__ bkt, const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy,
                   Xyzz<typename O::Base>* __restrict__ hparts) {
    constexpr uint32_t LN = O::LANES, WE = 64 / LN, EL = THREADS / LN;  // points per wave / per workgroup
    __shared__ Xyzz<typename O::Base> sh[THREADS / 64];
    const uint32_t tid = threadIdx.x, e = tid / LN, h = tid % LN, le = (tid & 63) / LN, wid = tid >> 6;
    part += (size_t)MSM_P * ((size_t)nchunks + nb);
    start += (size_t)MSM_P * (nb + 1);
    bkt += (size_t)MSM_P * nb;
    heavy += (size_t)MSM_P * nb;
    n_heavy += MSM_P;
    if constexpr (SPLIT > 1) hparts += (size_t)MSM_P * MSM_HEAVY_SLOTS;
    const uint32_t nh = *n_heavy;
    const uint32_t K = msm_chunk_len(start[nb], nchunks);
    const uint32_t share = blockIdx.x % SPLIT, nwg = gridDim.x / SPLIT;
    for (uint32_t hb = blockIdx.x / SPLIT; hb < nh; hb += nwg) {
        const uint32_t b = heavy[hb];
        uint32_t c0 = start[b] / K, c1 = (start[b + 1] - 1) / K;
        const bool split = SPLIT > 1 && hb < MSM_HEAVY_SLOTS / SPLIT;
        if (split) {
            const uint32_t len = (c1 - c0 + SPLIT) / SPLIT;  // ceil(span / SPLIT)
            c0 += share * len;
            c1 = c0 + len - 1 < c1 ? c0 + len - 1 : c1;
        } else if (share != 0) {
            continue;
        }
        Xyzz<O> acc = xyzz_inf<O>();
        for (uint32_t c = c0 + e; c <= c1 && c1 + 1 > c0; c += EL) xyzz_add_nc(acc, xyzz_load<O>(part + c + b, h));
        const uint32_t span = c1 + 1 > c0 ? c1 - c0 + 1 : 0;  // points >= span hold infinity: skip the tree levels that only move infinities
        for (uint32_t d = WE / 2; d >= 1; d >>= 1) {
            if (d >= span) continue;
            Xyzz<O> other = xyzz_shfl_down(acc, (int)(d * LN));
            if (le < d) xyzz_add_nc(acc, other);
        }
        if constexpr (THREADS > 64) {
            if (le == 0) xyzz_store<O>(sh + wid, acc, h);
            __syncthreads();
            if (e == 0)
                for (uint32_t w = 1; w < THREADS / 64; ++w) xyzz_add_nc(acc, xyzz_load<O>(sh + w, h));
        }
        if (e == 0) xyzz_store<O>(split ? hparts + hb * SPLIT + share : bkt + b, acc, h);
        if constexpr (THREADS > 64) __syncthreads();
    }
}
// bkt[heavy[hb]] = sum of the SPLIT shares of heavy bucket hb (hb < MSM_HEAVY_SLOTS / SPLIT): SPLIT groups of lanes per bucket, a
// shuffle tree.  64 lanes per workgroup = 64 / (LANES x SPLIT) buckets.
template <class O, uint32_t SPLIT>
__global__ void __launch_bounds__(64, MASP_TAIL_MIN_WAVES)
k_msm_heavy_join(const Xyzz<typename O::Base>* __restrict__ hparts, const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ n_heavy, uint32_t nb,
                 Xyzz<typename O::Base>* __restrict__ bkt) {
    constexpr uint32_t LN = O::LANES, PER = 64 / (LN * SPLIT);
    static_assert(PER >= 1, "a bucket's shares fit one wave");
    const uint32_t e = threadIdx.x / LN, h = threadIdx.x % LN, share = e % SPLIT, hb = blockIdx.x * PER + e / SPLIT;
    hparts += (size_t)MSM_P * MSM_HEAVY_SLOTS;
    heavy += (size_t)MSM_P * nb;
    bkt += (size_t)MSM_P * nb;
    const uint32_t nh = n_heavy[MSM_P], lim = nh < MSM_HEAVY_SLOTS / SPLIT ? nh : MSM_HEAVY_SLOTS / SPLIT;
    if (hb >= lim) return;  // (all SPLIT groups of a bucket leave together)
    Xyzz<O> acc = xyzz_load<O>(hparts + hb * SPLIT + share, h);
    for (uint32_t d = SPLIT / 2; d >= 1; d >>= 1) {
        Xyzz<O> other = xyzz_shfl_down(acc, (int)(d * LN));
        if (share < d) xyzz_add_nc(acc, other);
    }
    if (share == 0) xyzz_store<O>(bkt + heavy[hb], acc, h);
}

// ---- (6) reductions -----------------------------------------------------------------------------
// One level of the weighted sum  V(B, off) = sum_k (k + off) * B[k].  A workgroup of WSUM_L points (WSUM_L x LANES lanes) owns
// a chunk of WSUM_CS = WSUM_G * WSUM_L elements and produces  S[ch] = sum_l B[ch*cs + l],  T[ch] = sum_l (l + off) * B[ch*cs + l];
// then V(B, off) = sum_ch T[ch] + cs * V(S, 0).
//   phase 1  every lane (pair) runs the classic running sum over its own WSUM_G consecutive elements (2G - 1 additions);
//   phase 2  the per-lane sums are combined with a log-depth suffix scan and one tree through LDS.
// ~4.5 additions per bucket in total (a pure log-depth scan costs 16) at a depth of 33 dependent additions:
// with a batch of proofs in flight the chip is throughput-bound here, so work counts, not just depth.
// G = 2^G_LOG is chosen by the host: 16 for batches (least work per bucket: ~2.9 additions), 4 for a lone proof
// (shortest dependent chain).
// (the G1 kernels of a batch are held to 256 registers = two waves per SIMD: 219 spilled, launch 3.8 -> 3.05 ms for h + l)
#ifndef MASP_WSUM_PAIR_WAVES
#define MASP_WSUM_PAIR_WAVES 2
#endif
// points (groups of O::LANES lanes) per workgroup of k_msm_wsum_level: WSUM_L, or half of it where eight lanes hold a point (1 024 lanes
// per workgroup would leave a lane 128 registers) — the chunk a workgroup owns stays (WSUM_L << G_LOG) buckets, a group takes twice as many
template <class O>
constexpr uint32_t wsum_points_log() {
    return O::LANES > 4 ? WSUM_L_LOG - 1 : WSUM_L_LOG;
}
template <class O, uint32_t G_LOG>
__global__ void __launch_bounds__((1u << wsum_points_log<O>()) * O::LANES, (O::LANES > 1 ? MASP_WSUM_PAIR_WAVES : sizeof(Xyzz<O>) > 200 || G_LOG < 3 ? MASP_TAIL_MIN_WAVES : 2))
k_msm_wsum_level(const Xyzz<typename O::Base>* __restrict__ B, size_t b_stride, uint32_t m, uint32_t off, Xyzz<typename O::Base>* __restrict__ S,
                 Xyzz<typename O::Base>* __restrict__ T, size_t st_stride) {
    constexpr uint32_t LP = 1u << wsum_points_log<O>(), GL = G_LOG + WSUM_L_LOG - wsum_points_log<O>(), G = 1u << GL, CS = G * LP, LN = O::LANES,
                       WE = 64 / LN, NW = LP / WE;  // points per workgroup, buckets per point, points per wave, waves
    __shared__ Xyzz<typename O::Base> sh[2][NW];
    const uint32_t tid = threadIdx.x, e = tid / LN, h = tid % LN, le = (tid & 63) / LN, wid = tid >> 6;
    B += MSM_P * b_stride;
    S += MSM_P * st_stride;
    T += MSM_P * st_stride;
    // phase 1: lane-local running sum, top element first:  run = sum B_l,  acc = sum (l + off) B_l  (l local)
    const uint32_t base = blockIdx.x * CS + e * G;
    Xyzz<O> run = xyzz_inf<O>(), acc = xyzz_inf<O>();
    for (uint32_t l = G; l-- > 0;) {
        if (base + l < m) xyzz_add_nc(run, xyzz_load<O>(B + base + l, h));
        if (l > 0 || off) xyzz_add_nc(acc, run);
    }
    // phase 2: lanes.  sum_e (e * G) * run_e = G * sum_{i >= 1} x_i  with x = inclusive suffix scan of run over the WSUM_L
    // points: a shuffle scan inside each wave, then every wave adds the totals of the waves behind it
    Xyzz<O> x = run;
    for (uint32_t d = 1; d < WE; d <<= 1) {
        Xyzz<O> other = xyzz_shfl_down(x, (int)(d * LN));
        if (le + d < WE) xyzz_add_nc(x, other);
    }
    // ... across the waves: wave 0 turns the waves' totals into "sum of the totals of the waves behind" with one more shuffle scan over
    // its first NW points (NW <= 8 <= points per wave), so that every wave adds ONE value (as a loop over the waves behind it, wave 0
    // of a lone proof's 8-wave workgroup ran 7 dependent additions here and 7 more at the end)
    static_assert(NW <= WE, "the waves' totals fit the points of one wave");
    if (le == 0) xyzz_store<O>(&sh[0][wid], x, h);
    __syncthreads();
    if constexpr (NW > 1) {
        if (wid == 0) {
            Xyzz<O> v = le < NW ? xyzz_load<O>(&sh[0][le], h) : xyzz_inf<O>();
            for (uint32_t d = 1; d < NW; d <<= 1) {
                Xyzz<O> other = xyzz_shfl_down(v, (int)(d * LN));
                if (le + d < NW) xyzz_add_nc(v, other);
            }
            Xyzz<O> behind = xyzz_shfl_down(v, (int)LN);  // inclusive suffix of point le + 1 = the totals of the waves behind wave le
            if (le + 1 < NW) xyzz_store<O>(&sh[0][le], behind, h);
        }
        __syncthreads();
        if (wid + 1 < NW) xyzz_add_nc(x, xyzz_load<O>(&sh[0][wid], h));
    }
    Xyzz<O> y = e > 0 ? x : xyzz_inf<O>();
    for (uint32_t k = 0; k < GL; ++k) y = xyzz_dbl(y);
    xyzz_add_nc(y, acc);
    for (uint32_t d = WE / 2; d >= 1; d >>= 1) {
        Xyzz<O> other = xyzz_shfl_down(y, (int)(d * LN));
        if (le < d) xyzz_add_nc(y, other);
    }
    if (le == 0) xyzz_store<O>(&sh[1][wid], y, h);
    __syncthreads();
    if (wid == 0) {
        if constexpr (NW > 1) {
            y = le < NW ? xyzz_load<O>(&sh[1][le], h) : xyzz_inf<O>();
            for (uint32_t d = NW / 2; d >= 1; d >>= 1) {
                Xyzz<O> other = xyzz_shfl_down(y, (int)(d * LN));
                if (le < d) xyzz_add_nc(y, other);
            }
        }
        if (e == 0) {
            xyzz_store<O>(S + blockIdx.x, x, h);
            xyzz_store<O>(T + blockIdx.x, y, h);
        }
    }
}
// out[b] = sum of in[b*256 .. min(n, b*256+256)) by an LDS tree (8 dependent additions)
template <class O>
__global__ void __launch_bounds__(256) k_xyzz_reduce_block(const Xyzz<O>* __restrict__ in, size_t in_stride, uint32_t n,
                                                           Xyzz<O>* __restrict__ out, size_t out_stride) {
    extern __shared__ uint4 wsum_lds[];
    Xyzz<O>* sh = reinterpret_cast<Xyzz<O>*>(wsum_lds);
    const uint32_t tid = threadIdx.x;
    in += MSM_P * in_stride;
    out += MSM_P * out_stride;
    const uint32_t k = blockIdx.x * 256 + tid;
    Xyzz<O> y = k < n ? in[k] : xyzz_inf<O>();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        sh[tid] = y;
        __syncthreads();
        if (tid < d) xyzz_add_nc(y, sh[tid + d]);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = y;
}
// V = T0 + cs*(T1 + cs*(T2 + ...)) ;  tsum[l] holds the fully reduced T of level l.
template <class O>
__global__ void __launch_bounds__(64) k_msm_combine(const Xyzz<O>* __restrict__ tsum, int levels, int cs_log, Xyzz<O>* __restrict__ out, size_t out_stride) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    tsum += (size_t)MSM_P * 32;
    out += MSM_P * out_stride;
    Xyzz<O> acc = xyzz_inf<O>();
    for (int l = levels - 1; l >= 0; --l) {
        for (int k = 0; k < cs_log; ++k) acc = xyzz_dbl(acc);
        xyzz_add_nc(acc, tsum[l]);
    }
    *out = acc;
}

// The same two kernels over O::LANES lanes per point (a lone proof's tails: FpQuadOps, Fp2OctOps) — 256 / 64 points per workgroup, one point.
template <class O>
constexpr uint32_t reduce_lanes_points() {
    return O::LANES > 4 ? 64u : 256u;  // (eight lanes per point: 512 lanes per workgroup, so that a lane keeps 256 registers)
}
template <class O>
__global__ void __launch_bounds__(reduce_lanes_points<O>() * O::LANES) k_xyzz_reduce_block_lanes(const Xyzz<typename O::Base>* __restrict__ in, size_t in_stride, uint32_t n,
                                                                            Xyzz<typename O::Base>* __restrict__ out, size_t out_stride) {
    extern __shared__ uint4 wsum_lds[];
    constexpr uint32_t PTS = reduce_lanes_points<O>();
    Xyzz<typename O::Base>* sh = reinterpret_cast<Xyzz<typename O::Base>*>(wsum_lds);
    const uint32_t e = threadIdx.x / O::LANES, h = threadIdx.x % O::LANES;
    in += MSM_P * in_stride;
    out += MSM_P * out_stride;
    const uint32_t k = blockIdx.x * PTS + e;
    Xyzz<O> y = k < n ? xyzz_load<O>(in + k, h) : xyzz_inf<O>();
    for (uint32_t d = PTS / 2; d >= 1; d >>= 1) {
        xyzz_store<O>(sh + e, y, h);
        __syncthreads();
        if (e < d) xyzz_add_nc(y, xyzz_load<O>(sh + e + d, h));
        __syncthreads();
    }
    if (e == 0) xyzz_store<O>(out + blockIdx.x, y, h);
}
template <class O>
__global__ void __launch_bounds__(64) k_msm_combine_lanes(const Xyzz<typename O::Base>* __restrict__ tsum, int levels, int cs_log,
                                                          Xyzz<typename O::Base>* __restrict__ out, size_t out_stride) {
    if (blockIdx.x != 0 || threadIdx.x >= O::LANES) return;
    const uint32_t h = threadIdx.x;
    tsum += (size_t)MSM_P * 32;
    out += MSM_P * out_stride;
    Xyzz<O> acc = xyzz_inf<O>();
    for (int l = levels - 1; l >= 0; --l) {
        for (int k = 0; k < cs_log; ++k) acc = xyzz_dbl(acc);
        xyzz_add_nc(acc, xyzz_load<O>(tsum + l, h));
    }
    xyzz_store<O>(out, acc, h);
}

}  // namespace masp
