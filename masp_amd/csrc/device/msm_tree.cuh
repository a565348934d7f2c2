// Shared-inversion ("batch-affine") pre-reduction of the bucket runs of an MSM: stage (3b) of device/msm.cuh's plan, between
// the counting sort and the XYZZ accumulation.
//
// The sorted digit list holds, bucket after bucket, the window-table rows a bucket has to sum.  k_msm_accumulate adds them
// one after the other in extended-Jacobian coordinates: 8M + 2S = 9.6 field products per addition, and the kernel runs at
// ~95 % of what the integer multiplier delivers.  Here every bucket's run is first halved T times by pairwise AFFINE additions
//     lambda = (y2 - y1) / (x2 - x1),  x3 = lambda^2 - x1 - x2,  y3 = lambda (x1 - x3) - y1
// whose inversions are shared by ALL pairs of a level through Montgomery's trick spread over the grid: 5M + 1S = 5.8 products
// per addition.  After T levels 1 - 2^-T of the additions are done; k_msm_accumulate_pts and the unchanged bucket tails finish.
//
// One level (input: points of level L, dense, bucket after bucket; D_L[b] = first point of bucket b; level 0 = the digit list):
//   k_tree_plan     (once, all levels)  D_L[], Q_L[] = exclusive scans of len_L = ceil(len_0 / 2^L) and of len_L >> 1
//   k_tree_records  (levels >= pad_log) pair q of the level -> its record (first input point, output point): a binary search in
//                   Q_L.  The sort pads every run to a multiple of 2^pad_log entries (MsmSortBuf::pad_log, 2 by default; the
//                   padding is the point at infinity), so on levels below pad_log pair q is simply entries 2q, 2q + 1 (level 0:
//                   of the digit list itself) and lands at point q of the next level: no records, no odd points to copy
//   k_tree_pass1    lane t of a proof takes pairs t, t + NT, t + 2 NT, ... (every access of a wave is contiguous): denominator
//                   of each pair, running product along the lane, prefixes to `pre`, the lane's product to `tp`
//   k_binv_*        tp -> 1 / tp for all lanes: chains of products, ~4 096 binary-gcd inversions in the middle
//   k_tree_pass2    the same lanes backwards: 1 / d from the prefixes, the affine addition, the point to its place in level L+1
//   k_tree_copy     (levels >= pad_log) the last point of an odd bucket passes through
// The price is memory: a level reads its points twice and keeps 48 bytes per pair in between; level 0 gathers every table row
// twice (measured: random 128-byte rows arrive at 6.5 TB/s, tools/batch_affine_ubench.hip).
// Exceptional pairs (P + P, P - P, the point at infinity as an operand) are handled exactly, like everywhere else: proof bytes
// must equal the CPU prover's for any CRS.  Infinity is x = y = 0 (curve.cuh).
// Replaces nothing the reference has by name: bellperson's multiexp (call sites
// /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252) sums buckets in projective coordinates on the CPU.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.cuh"
#include "msm_geom.h"

namespace masp {

#ifndef MSM_P
#define MSM_P (blockIdx.y)
#endif

// ---- plan -------------------------------------------------------------------------------------------------------
// grid (T + 1, np), 1024 threads.  Level L = blockIdx.x of proof p = blockIdx.y:
//   D[(L np + p)(nb + 1) + b] = sum_{b' < b} len_L(b')        (D[0] = start; D[L][nb] = points of level L)
//   Q[(L np + p)(nb + 1) + b] = sum_{b' < b} len_L(b') >> 1   (Q[L][nb] = pairs of level L)
// (curve-independent kernels are `static`: the header is compiled into one translation unit per curve)
static __global__ void __launch_bounds__(1024) k_tree_plan(const uint32_t* __restrict__ start, uint32_t nb, uint32_t* __restrict__ D, uint32_t* __restrict__ Q) {
    __shared__ uint32_t wsum[2][16];
    __shared__ uint32_t base[2];
    const uint32_t L = blockIdx.x, p = blockIdx.y, np = gridDim.y;
    start += (size_t)p * (nb + 1);
    uint32_t* Dl = D + ((size_t)L * np + p) * (nb + 1);
    uint32_t* Ql = Q + ((size_t)L * np + p) * (nb + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, round = (1u << L) - 1u;
    if (tid == 0) base[0] = base[1] = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += blockDim.x) {
        const uint32_t b = b0 + tid;
        const uint32_t len = b < nb ? (start[b + 1] - start[b] + round) >> L : 0u;
        const uint32_t v0 = len, v1 = len >> 1;
        uint32_t x0 = v0, x1 = v1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y0 = __shfl_up(x0, d, 64), y1 = __shfl_up(x1, d, 64);
            if ((int)lane >= d) {
                x0 += y0;
                x1 += y1;
            }
        }
        if (lane == 63) {
            wsum[0][wid] = x0;
            wsum[1][wid] = x1;
        }
        __syncthreads();
        uint32_t w0 = 0, w1 = 0;
        for (uint32_t k = 0; k < wid; ++k) {
            w0 += wsum[0][k];
            w1 += wsum[1][k];
        }
        const uint32_t bs0 = base[0], bs1 = base[1];
        if (b < nb) {
            Dl[b] = bs0 + w0 + x0 - v0;
            Ql[b] = bs1 + w1 + x1 - v1;
        }
        __syncthreads();
        if (tid == blockDim.x - 1) {
            base[0] = bs0 + w0 + x0;
            base[1] = bs1 + w1 + x1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        Dl[nb] = base[0];
        Ql[nb] = base[1];
    }
}

// pair q of a level >= 1 -> its record: uint2 (index of the first input point, index of the output point).  Dl / Dn / Ql: this
// level's D, the next level's D, this level's Q (per proof: stride nb + 1).  Grid-stride over the pairs the proof really has; a
// binary search in Q_L per pair (one lane per bucket writing its pairs in a loop was measured 2.8x slower: the one long bucket of
// a witness MSM).  Level 0 has no records: its pairs are the digit list read two entries at a time (see the top of the file).
static __global__ void __launch_bounds__(256)
k_tree_records(const uint32_t* __restrict__ Dl, const uint32_t* __restrict__ Dn, const uint32_t* __restrict__ Ql, uint32_t nb, uint2* __restrict__ rec,
               size_t rec_stride) {
    Dl += (size_t)MSM_P * (nb + 1);
    Dn += (size_t)MSM_P * (nb + 1);
    Ql += (size_t)MSM_P * (nb + 1);
    rec += (size_t)MSM_P * rec_stride;
    const uint32_t P = Ql[nb];
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
        uint32_t b = 0, span = nb;  // largest b with Ql[b] <= q (then Ql[b + 1] > q: bucket b holds pair q)
        while (span > 1) {
            uint32_t half = span >> 1;
            if (Ql[b + half] <= q) b += half;
            span -= half;
        }
        const uint32_t j = q - Ql[b];
        rec[q] = make_uint2(Dl[b] + 2u * j, Dn[b] + j);
    }
}

// ---- the two operands of a pair ------------------------------------------------------------------------------------
enum : int { TREE_ADD = 0, TREE_DBL = 1, TREE_FIRST = 2, TREE_SECOND = 3, TREE_INF = 4 };  // FIRST / SECOND: the result is that operand

// what a pair needs, from all four coordinates: the kind and (ADD, DBL) the denominator its slope divides by
template <class O>
__device__ __forceinline__ int tree_classify(const typename O::T& x1, const typename O::T& y1, const typename O::T& x2, const typename O::T& y2,
                                             typename O::T& denom) {
    const bool inf1 = O::is_zero(x1) && O::is_zero(y1), inf2 = O::is_zero(x2) && O::is_zero(y2);
    if (inf2) return inf1 ? TREE_INF : TREE_FIRST;
    if (inf1) return TREE_SECOND;
    denom = O::sub(x2, x1);
    if (!O::is_zero(denom)) return TREE_ADD;
    if (O::eq(y1, y2) && !O::is_zero(y1)) {
        denom = O::dbl(y1);
        return TREE_DBL;
    }
    return TREE_INF;  // P + (-P)  (or a point of order two added to itself)
}

// the operands of a pair from its record — level 0: the two digit-list words (table row | sign << 31, or the padding entry =
// the point at infinity), gathered from the table and negated if the digit is negative; deeper levels: (first input point,
// output point), read from the previous level's points
template <bool L0>
struct TreeRec {
    typedef uint2 type;
};
template <class O, bool L0>
struct TreeSrc {
    typedef typename O::T F;
    typedef typename TreeRec<L0>::type Rec;
    // O::LANES lanes hold one element (Fp2PairOps: 2): lane `h` of them reads part h of every stored element
    const TabRow<typename O::Base>* tab;
    const F *xs, *ys;  // this proof's points (deeper levels)
    uint32_t h;
    static __device__ __forceinline__ uint32_t lane_part() {
        if constexpr (O::LANES > 1)
            return O::half();
        else
            return 0u;
    }
    __device__ __forceinline__ F row_x(uint32_t w) const {
        if (w == MSM_PAD_ENTRY) return O::zero();
        return reinterpret_cast<const F*>(&tab[w & 0x7fffffffu].p.x)[h];
    }
    __device__ __forceinline__ F row_y(uint32_t w) const {
        if (w == MSM_PAD_ENTRY) return O::zero();
        return reinterpret_cast<const F*>(&tab[w & 0x7fffffffu].p.y)[h];
    }
    __device__ __forceinline__ size_t at(size_t i) const { return i * O::LANES + h; }
    __device__ __forceinline__ void load_x(const Rec& r, F& x1, F& x2) const {
        if constexpr (L0) {
            x1 = row_x(r.x);
            x2 = row_x(r.y);
        } else {
            x1 = xs[at(r.x)];
            x2 = xs[at(r.x + 1)];
        }
    }
    // the y coordinates as stored (no arithmetic on them here: a load that is consumed at once cannot be overlapped with the
    // previous pair's products) ...
    __device__ __forceinline__ void load_y_raw(const Rec& r, F& y1, F& y2) const {
        if constexpr (L0) {
            y1 = row_y(r.x);
            y2 = row_y(r.y);
        } else {
            y1 = ys[at(r.x)];
            y2 = ys[at(r.x + 1)];
        }
    }
    // ... and the signs of the digits applied (level 0; the padding entry stays (0, 0): -0 = 0)
    static __device__ __forceinline__ void fix_y(const Rec& r, F& y1, F& y2) {
        if constexpr (L0) {
            if (r.x >> 31) y1 = O::neg(y1);
            if (r.y >> 31) y2 = O::neg(y2);
        }
    }
    __device__ __forceinline__ void load_y(const Rec& r, F& y1, F& y2) const {
        load_y_raw(r, y1, y2);
        fix_y(r, y1, y2);
    }
    static __device__ __forceinline__ uint32_t out_index(const Rec& r, uint32_t q) {  // level 0: pair q -> point q of level 1
        if constexpr (L0)
            return q;
        else
            return r.y;
    }
};

// ---- pass 1: denominators and their running products -----------------------------------------------------------------
// grid (NT / 256, np).  pre[(j np + p) NT + t] = product of the denominators of lane (p, t)'s pairs 0 .. j; tp[p NT + t] = all of them.
template <class O, bool L0>
__global__ void __launch_bounds__(256)
k_tree_pass1(const TabRow<typename O::Base>* __restrict__ tab, const typename O::T* __restrict__ xs, const typename O::T* __restrict__ ys,
             size_t pt_stride, const void* __restrict__ rec_, size_t rec_stride, const uint32_t* __restrict__ Ql, uint32_t nb, uint32_t NT,
             typename O::T* __restrict__ pre, typename O::T* __restrict__ tp) {
    typedef typename O::T F;
    typedef typename TreeRec<L0>::type Rec;
    constexpr uint32_t LN = O::LANES;  // lanes per element (see k_tree_pass2)
    const uint32_t p = MSM_P, np = gridDim.y, t = (blockIdx.x * blockDim.x + threadIdx.x) / LN;
    if (t >= NT) return;
    const uint32_t P = Ql[(size_t)p * (nb + 1) + nb];
    // rec_ == nullptr: a level >= 1 whose runs all have even lengths (the sort padded to a multiple of 2^(level + 1)): pair q is
    // points 2q, 2q + 1 and lands at point q, like level 0 over the digit list
    const Rec* recs = reinterpret_cast<const Rec*>(rec_) + (size_t)p * rec_stride;
    const bool synth = rec_ == nullptr;
    auto rec_at = [&](uint32_t q) -> Rec { return synth ? make_uint2(2u * q, q) : recs[q]; };
    TreeSrc<O, L0> src;
    src.tab = tab;
    src.xs = xs + (size_t)p * pt_stride * LN;
    src.ys = ys + (size_t)p * pt_stride * LN;
    src.h = TreeSrc<O, L0>::lane_part();
    F chain = O::one();
    // two-stage software pipeline: the record of pair j + 2 and the operands of pair j + 1 are requested before pair j is
    // multiplied in — an operand is two dependent loads away (record, then row / point) and nothing else hides that
    Rec ra{}, rb{};
    F x1 = O::zero(), x2 = O::zero();
    if (t < P) {
        ra = rec_at(t);
        if (t + NT < P) rb = rec_at(t + NT);
        src.load_x(ra, x1, x2);
    }
    // (gfx9 counts loads and stores in ONE counter and stores may complete out of order, so a wait for loaded data drains every
    // store issued before it: the prefix of pair j is therefore stored at the top of iteration j + 1, right after that
    // iteration's wait and before its loads — by the next wait it has had a whole iteration to complete)
    uint32_t j = 0;
    for (uint32_t q = t; q < P; q += NT, ++j) {
        const Rec cr = ra;
        const F cx1 = x1, cx2 = x2;
        if (j) pre[src.at(((size_t)(j - 1) * np + p) * NT + t)] = chain;
        ra = rb;
        if (q + NT < P) src.load_x(ra, x1, x2);
        if (q + 2 * (uint64_t)NT < P) rb = rec_at(q + 2 * NT);
        F d = O::sub(cx2, cx1);
        if (O::is_zero(cx1) || O::is_zero(cx2) || O::is_zero(d)) {  // rare: needs the y coordinates to decide
            F y1, y2;
            src.load_y(cr, y1, y2);
            if (tree_classify<O>(cx1, y1, cx2, y2, d) > TREE_DBL) d = O::one();
        }
        chain = O::mul(chain, d);
    }
    if (j) pre[src.at(((size_t)(j - 1) * np + p) * NT + t)] = chain;
    tp[src.at((size_t)p * NT + t)] = chain;
}

// ---- pass 2: the additions --------------------------------------------------------------------------------------------
// tinv[p NT + t] = 1 / tp[p NT + t].  The lane walks its pairs backwards: 1 / d_j = (1 / (d_0 .. d_j)) (d_0 .. d_{j-1}).
template <class O, bool L0>
__global__ void __launch_bounds__(256, (sizeof(typename O::T) > 48 ? 1 : 2))   // two waves per SIMD (<= 256 VGPRs) where an element is 12 registers
k_tree_pass2(const TabRow<typename O::Base>* __restrict__ tab, const typename O::T* __restrict__ xs, const typename O::T* __restrict__ ys,
             size_t pt_stride, const void* __restrict__ rec_, size_t rec_stride, const uint32_t* __restrict__ Ql, uint32_t nb, uint32_t NT,
             const typename O::T* __restrict__ pre, const typename O::T* __restrict__ tinv, typename O::T* __restrict__ ox,
             typename O::T* __restrict__ oy, size_t out_stride) {
    typedef typename O::T F;
    typedef typename TreeRec<L0>::type Rec;
    constexpr uint32_t LN = O::LANES;  // lanes per element; every stride and index below counts ELEMENTS (LN values of F each)
    const uint32_t p = MSM_P, np = gridDim.y, t = (blockIdx.x * blockDim.x + threadIdx.x) / LN;
    if (t >= NT) return;
    const uint32_t P = Ql[(size_t)p * (nb + 1) + nb];
    if (t >= P) return;
    // rec_ == nullptr: a level >= 1 whose runs all have even lengths (the sort padded to a multiple of 2^(level + 1)): pair q is
    // points 2q, 2q + 1 and lands at point q, like level 0 over the digit list
    const Rec* recs = reinterpret_cast<const Rec*>(rec_) + (size_t)p * rec_stride;
    const bool synth = rec_ == nullptr;
    auto rec_at = [&](uint32_t q) -> Rec { return synth ? make_uint2(2u * q, q) : recs[q]; };
    ox += (size_t)p * out_stride * LN;
    oy += (size_t)p * out_stride * LN;
    TreeSrc<O, L0> src;
    src.tab = tab;
    src.xs = xs + (size_t)p * pt_stride * LN;
    src.ys = ys + (size_t)p * pt_stride * LN;
    src.h = TreeSrc<O, L0>::lane_part();
    F I = tinv[src.at((size_t)p * NT + t)];
    // two-stage software pipeline, backwards: the record of pair j - 2 and the operands of pair j - 1 are requested before pair
    // j is computed (see pass 1)
    struct Ops {
        F x1, y1, x2, y2;
    };
    auto fetch = [&](const Rec& r, Ops& o) {
        src.load_x(r, o.x1, o.x2);
        src.load_y_raw(r, o.y1, o.y2);
    };
    uint32_t j = (P - 1 - t) / NT;
    Rec ra = rec_at(t + j * NT), rb{};
    if (j) rb = rec_at(t + (j - 1) * NT);
    Ops nxt;
    fetch(ra, nxt);
    // (the result of a pair is stored at the top of the NEXT iteration, after that iteration's wait for its operands and before
    // its loads: see pass 1 — a store in flight would otherwise be drained by the wait)
    F hx = O::zero(), hy = O::zero();
    uint32_t hout = 0;
    bool held = false;
    for (;; --j) {
        Ops c = nxt;
        const Rec cr = ra;
        ra = rb;
        if (held) {
            ox[src.at(hout)] = hx;
            oy[src.at(hout)] = hy;
        }
        F pp = O::one();
        if (j) pp = pre[src.at(((size_t)(j - 1) * np + p) * NT + t)];
        if (j) fetch(ra, nxt);
        if (j > 1) rb = rec_at(t + (j - 2) * NT);
        const uint32_t out = TreeSrc<O, L0>::out_index(cr, t + j * NT);
        TreeSrc<O, L0>::fix_y(cr, c.y1, c.y2);
        F d = O::sub(c.x2, c.x1);
        int kind = TREE_ADD;
        if (O::is_zero(c.x1) || O::is_zero(c.x2) || O::is_zero(d)) kind = tree_classify<O>(c.x1, c.y1, c.x2, c.y2, d);
        F x3, y3;
        if (kind <= TREE_DBL) {
            const F Inext = O::mul(I, d);
            const F inv = j ? O::mul(I, pp) : I;
            I = Inext;
            F lam, xx = c.x2;
            if (kind == TREE_ADD) {
                lam = O::mul(O::sub(c.y2, c.y1), inv);
            } else {
                const F s = O::sqr(c.x1);
                lam = O::mul(O::add(O::dbl(s), s), inv);
                xx = c.x1;
            }
            x3 = O::sub(O::sub(O::sqr(lam), c.x1), xx);
            y3 = O::sub(O::mul(lam, O::sub(c.x1, x3)), c.y1);
        } else if (kind == TREE_FIRST) {
            x3 = c.x1;
            y3 = c.y1;
        } else if (kind == TREE_SECOND) {
            x3 = c.x2;
            y3 = c.y2;
        } else {
            x3 = O::zero();
            y3 = O::zero();
        }
        hx = x3;
        hy = y3;
        hout = out;
        held = true;
        if (!j) break;
    }
    ox[src.at(hout)] = hx;
    oy[src.at(hout)] = hy;
}

// the last point of a bucket with an odd number of points goes to the next level as it is.  grid (nb / 256, np)
template <class O, bool L0>
__global__ void __launch_bounds__(256)
k_tree_copy(const TabRow<O>* __restrict__ tab, const uint32_t* __restrict__ sorted, size_t ent_stride, const typename O::T* __restrict__ xs,
            const typename O::T* __restrict__ ys, size_t pt_stride, const uint32_t* __restrict__ Dl, const uint32_t* __restrict__ Dn, uint32_t nb,
            typename O::T* __restrict__ ox, typename O::T* __restrict__ oy, size_t out_stride) {
    const uint32_t p = MSM_P, b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    Dl += (size_t)p * (nb + 1);
    Dn += (size_t)p * (nb + 1);
    const uint32_t len = Dl[b + 1] - Dl[b];
    if (!(len & 1u)) return;
    const uint32_t in = Dl[b] + len - 1u, out = Dn[b] + (len >> 1);
    typename O::T x, y;
    if (L0) {
        const uint32_t w = sorted[(size_t)p * ent_stride + in];
        const Affine<O> pt = tab[w & 0x7fffffffu].p;
        x = pt.x;
        y = (w >> 31) ? O::neg(pt.y) : pt.y;
    } else {
        x = xs[(size_t)p * pt_stride + in];
        y = ys[(size_t)p * pt_stride + in];
    }
    ox[(size_t)p * out_stride + out] = x;
    oy[(size_t)p * out_stride + out] = y;
}

// ---- grid-wide batch inversion (Montgomery's trick as chains of products) ------------------------------------------------
// (all of these also run over lane pairs — O::LANES = 2, Fp2PairOps —: a chain is a string of dependent products, and a pair
// finishes one in half the instructions)
// forward: chain m < M takes elements m, m + M, m + 2 M, ... < n: pre[k M + m] = product of its first k + 1, tot[m] = all
template <class O>
__global__ void __launch_bounds__(256) k_binv_fwd(const typename O::T* __restrict__ in, uint32_t n, uint32_t M, typename O::T* __restrict__ pre,
                                                   typename O::T* __restrict__ tot) {
    typedef typename O::T F;
    constexpr uint32_t LN = O::LANES;
    const uint32_t m = (blockIdx.x * blockDim.x + threadIdx.x) / LN, h = threadIdx.x % LN;
    if (m >= M) return;
    F chain = O::one();
    uint32_t k = 0;
    for (uint32_t i = m; i < n; i += M, ++k) {
        chain = O::mul(chain, in[(size_t)i * LN + h]);
        pre[((size_t)k * M + m) * LN + h] = chain;
    }
    tot[(size_t)m * LN + h] = chain;
}
// backward: out[i] = 1 / in[i] given itot[m] = 1 / tot[m]
template <class O>
__global__ void __launch_bounds__(256) k_binv_bwd(const typename O::T* __restrict__ in, uint32_t n, uint32_t M, const typename O::T* __restrict__ pre,
                                                   const typename O::T* __restrict__ itot, typename O::T* __restrict__ out) {
    typedef typename O::T F;
    constexpr uint32_t LN = O::LANES;
    const uint32_t m = (blockIdx.x * blockDim.x + threadIdx.x) / LN, h = threadIdx.x % LN;
    if (m >= M || m >= n) return;
    F I = itot[(size_t)m * LN + h];
    for (uint32_t k = (n - 1 - m) / M + 1; k-- > 0;) {
        const size_t i = (size_t)k * M + m;
        const F v = in[i * LN + h];
        out[i * LN + h] = k ? O::mul(I, pre[((size_t)(k - 1) * M + m) * LN + h]) : I;
        I = O::mul(I, v);
    }
}
// the middle: chain m < M inverts elements m, m + M, ... < n of `in` with its own inversion (binary gcd)
template <class O>
__global__ void __launch_bounds__(64) k_binv_mid(const typename O::T* __restrict__ in, uint32_t n, uint32_t M, typename O::T* __restrict__ pre,
                                                  typename O::T* __restrict__ out) {
    typedef typename O::T F;
    constexpr uint32_t LN = O::LANES;
    const uint32_t m = (blockIdx.x * blockDim.x + threadIdx.x) / LN, h = threadIdx.x % LN;
    if (m >= M || m >= n) return;
    F chain = O::one();
    uint32_t k = 0;
    for (uint32_t i = m; i < n; i += M, ++k) {
        chain = O::mul(chain, in[(size_t)i * LN + h]);
        pre[((size_t)k * M + m) * LN + h] = chain;
    }
    F I = O::inv_gcd(chain);
    while (k-- > 0) {
        const size_t i = (size_t)k * M + m;
        const F v = in[i * LN + h];
        out[i * LN + h] = k ? O::mul(I, pre[((size_t)(k - 1) * M + m) * LN + h]) : I;
        I = O::mul(I, v);
    }
}

// ---- accumulation of explicit points (what the tree leaves) -------------------------------------------------------------
// k_msm_accumulate with the digit list and the table replaced by the level-T points: same chunks, same partial sums
template <class O>
__global__ void __launch_bounds__(64, 1)
k_msm_accumulate_pts(const typename O::T* __restrict__ xs, const typename O::T* __restrict__ ys, size_t pt_stride, const uint32_t* __restrict__ start,
                     uint32_t nb, uint32_t nchunks, Xyzz<typename O::Base>* __restrict__ part) {
    typedef typename O::T F;
    constexpr uint32_t LN = O::LANES;
    const uint32_t ch = (blockIdx.x * blockDim.x + threadIdx.x) / LN, h = threadIdx.x % LN;
    if (ch >= nchunks) return;
    xs += (size_t)MSM_P * pt_stride * LN;
    ys += (size_t)MSM_P * pt_stride * LN;
    start += (size_t)MSM_P * (nb + 1);
    part += (size_t)MSM_P * ((size_t)nchunks + nb);
    const uint32_t total = start[nb];
    const uint32_t K = msm_chunk_len(total, nchunks);
    const uint32_t lo = ch * K;
    if (lo >= total) return;
    const uint32_t hi = lo + K < total ? lo + K : total;
    uint32_t b = 0, span = nb;
    while (span > 1) {
        uint32_t half = span >> 1;
        if (start[b + half] <= lo) b += half;
        span -= half;
    }
    auto put = [&](uint32_t slot, const Xyzz<O>& v) {  // a stored XYZZ point is four elements of LN parts each
        F* q = reinterpret_cast<F*>(part + slot);
        q[h] = v.X;
        q[LN + h] = v.Y;
        q[2 * LN + h] = v.ZZ;
        q[3 * LN + h] = v.ZZZ;
    };
    uint32_t next = start[b + 1];
    Xyzz<O> acc = xyzz_inf<O>();
    for (uint32_t pos = lo; pos < hi; ++pos) {
        if (pos >= next) {
            put(ch + b, acc);
            acc = xyzz_inf<O>();
            do {
                ++b;
                next = start[b + 1];
            } while (pos >= next);
        }
        Affine<O> pt;
        pt.x = xs[(size_t)pos * LN + h];
        pt.y = ys[(size_t)pos * LN + h];
        xyzz_madd(acc, pt, false);
    }
    put(ch + b, acc);
}

}  // namespace masp
