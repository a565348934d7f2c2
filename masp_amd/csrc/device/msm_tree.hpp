// Shared-inversion ("batch-affine") pre-reduction of the bucket runs of an MSM: stage (3b) of device/msm.hpp's plan, between
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
// must equal the CPU prover's for any CRS.  Infinity is x = y = 0 (curve.hpp).
// Replaces nothing the reference has by name: bellperson's multiexp (call sites
// /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252) sums buckets in projective coordinates on the CPU.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "msm_geom.h"

namespace masp {

#ifndef MSM_P
#define MSM_P (blockIdx.y)
#endif

// ---- planes ------------------------------------------------------------------------------------------------------------
// The tree's big arrays — the points of a level (x[], y[]) and the running products `pre` — are PLANES of `cap` 48-byte field
// elements cut into three 16-byte slices: element e lives at b[e], b[cap + e], b[2 cap + e] (uint4 units).  A wave reading or
// writing elements e0 .. e0 + 63 then moves 1 KiB of whole 128-byte lines per instruction.  With the elements stored whole
// (12 words after each other) every one of the three 16-byte instructions of an access touched all 24 lines of the wave's
// 3 KiB, a third of each: with ~30 such streams in flight per CU the lines did not survive in L2 between the three (round 3:
// FETCH_SIZE / WRITE_SIZE of the passes ~1.6x the bytes they need).  An Fp2 (one lane per point: Fp2Ops) is two consecutive
// elements, as in the lane-pair form.
// At level 0 pass 1 leaves, per pair, (numerator of the slope) x (product of the denominators before the pair) instead of the running
// product alone: pass 2 — the kernel that saturates VALU issue — gets the slope with ONE product instead of two, and pass 1 does the
// other one (it then reads the y coordinates too: at level 0 they lie in the 128-byte row it gathers anyway).  Level 0 only: there it
// takes 0.45 ms per call off the two passes together; on the deeper levels the two kernels merely swap 0.21 ms and pass 1 reads 96
// bytes per pair more (measured in round 4: DESIGN.md §6).
template <bool L0>
struct TREE_QNUM_AT {
    static constexpr bool value = L0;
};
__device__ __forceinline__ Fp plane_ld_fp(const uint4* __restrict__ b, size_t cap, size_t u) {
    const uint4 a = b[u], c = b[cap + u], d = b[2 * cap + u];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = c.x; r.v[5] = c.y; r.v[6] = c.z; r.v[7] = c.w;
    r.v[8] = d.x; r.v[9] = d.y; r.v[10] = d.z; r.v[11] = d.w;
    return r;
}
__device__ __forceinline__ void plane_st_fp(uint4* __restrict__ b, size_t cap, size_t u, const Fp& v) {
    b[u] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    b[cap + u] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
    b[2 * cap + u] = make_uint4(v.v[8], v.v[9], v.v[10], v.v[11]);
}
// element e of a plane of `cap` elements of type F (Fp, or Fp2 = two Fp)
template <class F>
__device__ __forceinline__ F plane_ld(const F* __restrict__ base, size_t cap, size_t e) {
    const uint4* b = reinterpret_cast<const uint4*>(base);
    if constexpr (sizeof(F) == sizeof(Fp)) {
        return plane_ld_fp(b, cap, e);
    } else {
        F r;
        r.c0 = plane_ld_fp(b, 2 * cap, 2 * e);
        r.c1 = plane_ld_fp(b, 2 * cap, 2 * e + 1);
        return r;
    }
}
template <class F>
__device__ __forceinline__ void plane_st(F* __restrict__ base, size_t cap, size_t e, const F& v) {
    uint4* b = reinterpret_cast<uint4*>(base);
    if constexpr (sizeof(F) == sizeof(Fp)) {
        plane_st_fp(b, cap, e, v);
    } else {
        plane_st_fp(b, 2 * cap, 2 * e, v.c0);
        plane_st_fp(b, 2 * cap, 2 * e + 1, v.c1);
    }
}

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

// Where point i of a proof lives in a plane of cap = np * pt_stride * LN elements (LN lanes per point, part h of it): points
// with even and odd index in separate halves of the plane — the two operands of pair q (points 2q, 2q + 1 where the runs are
// padded, a record's r.x, r.x + 1 elsewhere) are then each contiguous across the lanes of a wave, and so are the results
// (point t + j NT from lane t).  pt_stride is even.
template <uint32_t LN>
__device__ __forceinline__ size_t tree_pt_base(uint32_t p, size_t pt_stride) {
    return (size_t)p * (pt_stride >> 1) * LN;
}
template <uint32_t LN>
__device__ __forceinline__ size_t tree_pt_slot(size_t i, uint32_t h, size_t cap) {
    return (i >> 1) * LN + h + (i & 1) * (cap >> 1);
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

// `a` where m, else `b`, word by word (v_cndmask on the lane mask: no branch, no exec-masked block)
template <class F>
__device__ __forceinline__ F tree_select(bool m, const F& a, const F& b) {
    F r;
    const uint32_t *pa = reinterpret_cast<const uint32_t*>(&a), *pb = reinterpret_cast<const uint32_t*>(&b);
    uint32_t* pr = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(F) / 4; ++i) pr[i] = m ? pa[i] : pb[i];
    return r;
}
// THE exceptional pair that is not rare: the second operand is the point at infinity.  The sort pads every run to a multiple of four
// entries, so an h + l MSM (32 768 buckets of ~70 entries) has one pair in 36 with the padding entry as its second operand at level
// 0, and as many with the point at infinity (the sum of two padding entries) at level 1: a wave of 64 lanes met one in five iterations
// of six and ran the whole classification — ~400 instructions of exec-masked blocks and the copies that merge their results into the
// common path's registers — for it.  Such a pair is now handled by the common path's own data flow: its denominator is replaced by 1
// (what tree_classify's TREE_FIRST / TREE_INF meant for the shared inversion) and its result by the first operand, with word-wise
// selects on the lane mask; the classification branch is left to the pairs that are really rare (P + P, P - P, infinity as the FIRST
// operand only, a point with x = 0).  Same values in `pre`, `tp` and the next level as before, pair by pair.

// the operands of a pair from its record — level 0: the two digit-list words (table row | sign << 31, or the padding entry =
// the point at infinity), gathered from the table and negated if the digit is negative; deeper levels: (first input point,
// output point), read from the previous level's points
// The loads of the software pipelines are unconditional (the last iteration requests its own pair once more; the padding entry
// reads a row of zeros): a conditional load keeps the old value of its 12 destination registers alive — the compiler copied every
// operand twice per pair (old value into the destination before the load, destination into the working set after it).
static __device__ uint4 g_tree_zero_row[16];  // 256 bytes of zeros: the row a padding entry gathers (TabRow<Fp2Ops> is the larger)
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
    const F *xs, *ys;  // the level's point planes (deeper levels): `cap` elements each
    size_t cap, off;   // (set() below)
    uint32_t h;
    static __device__ __forceinline__ uint32_t lane_part() {
        if constexpr (O::LANES > 1)
            return O::half();
        else
            return 0u;
    }
    // the padding entry reads a row of zeros (= the point at infinity) that lives next to the code: the address is selected, the
    // loads are unconditional — no exec-masked blocks, no registers cleared for the lanes that skip them
    __device__ __forceinline__ const TabRow<typename O::Base>* row(uint32_t w) const {
        const TabRow<typename O::Base>* z = reinterpret_cast<const TabRow<typename O::Base>*>(g_tree_zero_row);
        return w == MSM_PAD_ENTRY ? z : tab + (w & 0x7fffffffu);
    }
    __device__ __forceinline__ F row_x(uint32_t w) const { return reinterpret_cast<const F*>(&row(w)->p.x)[h]; }
    __device__ __forceinline__ F row_y(uint32_t w) const { return reinterpret_cast<const F*>(&row(w)->p.y)[h]; }
    __device__ __forceinline__ size_t at(size_t i) const { return i * O::LANES + h; }
    // proof p of np, pt_stride points per proof (even)
    __device__ __forceinline__ void set(const F* xs_, const F* ys_, uint32_t p, uint32_t np, size_t pt_stride) {
        xs = xs_;
        ys = ys_;
        h = lane_part();
        cap = (size_t)np * pt_stride * O::LANES;
        off = tree_pt_base<O::LANES>(p, pt_stride);
    }
    __device__ __forceinline__ size_t slot(size_t i) const { return off + tree_pt_slot<O::LANES>(i, h, cap); }
    __device__ __forceinline__ void load_x(const Rec& r, F& x1, F& x2) const {
        if constexpr (L0) {
            x1 = row_x(r.x);
            x2 = row_x(r.y);
        } else {
            x1 = plane_ld(xs, cap, slot(r.x));
            x2 = plane_ld(xs, cap, slot(r.x + 1));
        }
    }
    // the y coordinates as stored (no arithmetic on them here: a load that is consumed at once cannot be overlapped with the
    // previous pair's products) ...
    __device__ __forceinline__ void load_y_raw(const Rec& r, F& y1, F& y2) const {
        if constexpr (L0) {
            y1 = row_y(r.x);
            y2 = row_y(r.y);
        } else {
            y1 = plane_ld(ys, cap, slot(r.x));
            y2 = plane_ld(ys, cap, slot(r.x + 1));
        }
    }
    __device__ __forceinline__ void load_y1_raw(const Rec& r, F& y1) const {
        if constexpr (L0)
            y1 = row_y(r.x);
        else
            y1 = plane_ld(ys, cap, slot(r.x));
    }
    __device__ __forceinline__ void load_y2_raw(const Rec& r, F& y2) const {
        if constexpr (L0)
            y2 = row_y(r.y);
        else
            y2 = plane_ld(ys, cap, slot(r.x + 1));
    }
    // ... and the signs of the digits applied (level 0; the padding entry stays (0, 0): -0 = 0)
    // (word-wise selects of an unconditional negation: as `if (sign) y = -y` the compiler built two exec-masked blocks of ~190
    // instructions each around the 36 of the negation — 8 % of the instructions of level 0's additions pass)
    static __device__ __forceinline__ F neg_if(const F& y, uint32_t sign_word) {
        const F n = O::neg(y);
        const uint32_t m = (uint32_t)((int32_t)sign_word >> 31);
        F r;
        const uint32_t *py = reinterpret_cast<const uint32_t*>(&y), *pn = reinterpret_cast<const uint32_t*>(&n);
        uint32_t* pr = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
        for (uint32_t i = 0; i < sizeof(F) / 4; ++i) pr[i] = (pn[i] & m) | (py[i] & ~m);
        return r;
    }
    static __device__ __forceinline__ void fix_y(const Rec& r, F& y1, F& y2) {
        if constexpr (L0) {
            y1 = neg_if(y1, r.x);
            y2 = neg_if(y2, r.y);
        }
    }
    static __device__ __forceinline__ void fix_y1(const Rec& r, F& y1) {
        if constexpr (L0) y1 = neg_if(y1, r.x);
    }
    static __device__ __forceinline__ void fix_y2(const Rec& r, F& y2) {
        if constexpr (L0) y2 = neg_if(y2, r.y);
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
             typename O::T* __restrict__ pre, size_t pre_cap, typename O::T* __restrict__ tp) {
    typedef typename O::T F;
    typedef typename TreeRec<L0>::type Rec;
    constexpr uint32_t LN = O::LANES;  // lanes per element (see k_tree_pass2)
    const uint32_t p = MSM_P, np = gridDim.y, t = (blockIdx.x * blockDim.x + threadIdx.x) / LN;
    if (t >= NT) return;
    const uint32_t P = Ql[(size_t)p * (nb + 1) + nb];
    // rec_ == nullptr: a level >= 1 whose runs all have even lengths (the sort padded to a multiple of 2^(level + 1)): pair q is
    // points 2q, 2q + 1 and lands at point q, like level 0 over the digit list
    const Rec* recs = reinterpret_cast<const Rec*>(rec_) + (size_t)p * rec_stride;
    // (level 0 always has its records — the digit list; on the deeper levels the load is unconditional too, from a harmless
    // address when there are no records: a load inside a conditional block is waited for at the block's end, with vmcnt(0))
    const bool synth = !L0 && rec_ == nullptr;
    const Rec* recs_ld = synth ? reinterpret_cast<const Rec*>(Ql) : recs;
    auto rec_at = [&](uint32_t q) -> Rec {
        const Rec r = recs_ld[synth ? 0u : q];
        return synth ? make_uint2(2u * q, q) : r;
    };
    TreeSrc<O, L0> src;
    src.tab = tab;
    src.set(xs, ys, p, np, pt_stride);
    F chain = O::one();
    if constexpr (TREE_QNUM_AT<L0>::value) {
    // two-stage software pipeline: the record of pair j + 2 and the operands of pair j + 1 are requested before pair j is
    // multiplied in — an operand is two dependent loads away (record, then row / point) and nothing else hides that
    struct Ops {
        F x1, y1, x2, y2;
    };
    auto fetch = [&](const Rec& r, Ops& o) {
        src.load_x(r, o.x1, o.x2);
        src.load_y_raw(r, o.y1, o.y2);
    };
    Rec ra{}, rb{};
    Ops nxt{O::zero(), O::zero(), O::zero(), O::zero()};
    if (t < P) {
        ra = rec_at(t);
        if (t + NT < P) rb = rec_at(t + NT);
        fetch(ra, nxt);
    }
    // (gfx9 counts loads and stores in ONE counter and stores may complete out of order, so a wait for loaded data drains every
    // store issued before it: what pair j leaves is therefore stored at the top of iteration j + 1, right after that
    // iteration's wait and before its loads — by the next wait it has had a whole iteration to complete)
    F hq = O::zero();
    uint32_t j = 0;
    for (uint32_t q = t; q < P; q += NT, ++j) {
        const Rec cr = ra;
        Ops c = nxt;
        if (j) plane_st(pre, pre_cap, src.at(((size_t)(j - 1) * np + p) * NT + t), hq);
        ra = rb;
        if (q + NT >= P) ra = cr;  // the last iteration asks for its own pair again: nothing reads it
        fetch(ra, nxt);
        rb = rec_at(q + 2 * (uint64_t)NT < P ? q + 2 * NT : q);
        TreeSrc<O, L0>::fix_y(cr, c.y1, c.y2);
        const bool pad2 = cr.y == MSM_PAD_ENTRY;  // (this form is level 0's: the second operand is the padding entry)
        F d = O::sub(c.x2, c.x1), n = O::sub(c.y2, c.y1);
        if (!pad2 && (O::is_zero(c.x1) || O::is_zero(c.x2) || O::is_zero(d))) {  // rare
            const int kind = tree_classify<O>(c.x1, c.y1, c.x2, c.y2, d);
            if (kind == TREE_DBL) {
                const F s2 = O::sqr(c.x1);
                n = O::add(O::dbl(s2), s2);  // the slope of the tangent: 3 x^2 over 2 y (tree_classify left d = 2 y)
            } else if (kind > TREE_DBL) {
                d = O::one();  // nothing is divided here, and pass 2 does not read this pair's numerator
                n = O::one();
            }
        }
        d = tree_select(pad2, O::one(), d);  // nothing is divided for a pair with the padding entry; pass 2 does not use its numerator
        // (both left in [0, 2p) — O::mul_lazy —: they are only ever multiplied again, by pass 2 and the shared inversion)
        hq = O::mul_lazy(n, chain);    // numerator x (denominators before this pair): pass 2 multiplies by 1 / (denominators up to this pair)
        chain = O::mul_lazy(chain, d);
    }
    if (j) plane_st(pre, pre_cap, src.at(((size_t)(j - 1) * np + p) * NT + t), hq);
    tp[src.at((size_t)p * NT + t)] = chain;
    } else {
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
        if (j) plane_st(pre, pre_cap, src.at(((size_t)(j - 1) * np + p) * NT + t), chain);
        ra = rb;
        if (q + NT >= P) ra = cr;
        src.load_x(ra, x1, x2);
        rb = rec_at(q + 2 * (uint64_t)NT < P ? q + 2 * NT : q);
        F d = O::sub(cx2, cx1);
        if (O::is_zero(cx1) || O::is_zero(cx2) || O::is_zero(d)) {  // rare: needs the y coordinates to decide
            F y1, y2;
            src.load_y(cr, y1, y2);
            if (tree_classify<O>(cx1, y1, cx2, y2, d) > TREE_DBL) d = O::one();
        }
        chain = O::mul_lazy(chain, d);  // in [0, 2p): only ever multiplied again
    }
    if (j) plane_st(pre, pre_cap, src.at(((size_t)(j - 1) * np + p) * NT + t), chain);
    tp[src.at((size_t)p * NT + t)] = chain;
    }
}

// ---- pass 2: the additions --------------------------------------------------------------------------------------------
// tinv[p NT + t] = 1 / tp[p NT + t].  The lane walks its pairs backwards: 1 / d_j = (1 / (d_0 .. d_j)) (d_0 .. d_{j-1}).
template <class O, bool L0>
__global__ void __launch_bounds__(256, (sizeof(typename O::T) > 48 ? 1 : 2))   // two waves per SIMD (<= 256 VGPRs) where an element is 12 registers
k_tree_pass2(const TabRow<typename O::Base>* __restrict__ tab, const typename O::T* __restrict__ xs, const typename O::T* __restrict__ ys,
             size_t pt_stride, const void* __restrict__ rec_, size_t rec_stride, const uint32_t* __restrict__ Ql, uint32_t nb, uint32_t NT,
             const typename O::T* __restrict__ pre, size_t pre_cap, const typename O::T* __restrict__ tinv, typename O::T* __restrict__ ox,
             typename O::T* __restrict__ oy, size_t out_stride, uint32_t out_whole) {
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
    // (level 0 always has its records — the digit list; on the deeper levels the load is unconditional too, from a harmless
    // address when there are no records: a load inside a conditional block is waited for at the block's end, with vmcnt(0))
    const bool synth = !L0 && rec_ == nullptr;
    const Rec* recs_ld = synth ? reinterpret_cast<const Rec*>(Ql) : recs;
    auto rec_at = [&](uint32_t q) -> Rec {
        const Rec r = recs_ld[synth ? 0u : q];
        return synth ? make_uint2(2u * q, q) : r;
    };
    TreeSrc<O, L0> src;
    src.tab = tab;
    src.set(xs, ys, p, np, pt_stride);
    // the results: point `out` of the next level's planes — or, from the last level (out_whole), whole elements proof after
    // proof, point after point: k_msm_accumulate_pts walks them one lane per chunk
    const size_t out_cap = (size_t)np * out_stride * LN, out_off = tree_pt_base<LN>(p, out_stride), whole_off = (size_t)p * out_stride * LN;
    auto put = [&](uint32_t out, const F& x, const F& y) {
        if (out_whole) {
            ox[whole_off + src.at(out)] = x;
            oy[whole_off + src.at(out)] = y;
        } else {
            const size_t e = out_off + tree_pt_slot<LN>(out, src.h, out_cap);
            plane_st(ox, out_cap, e, x);
            plane_st(oy, out_cap, e, y);
        }
    };
    F I = tinv[src.at((size_t)p * NT + t)];
    if constexpr (TREE_QNUM_AT<L0>::value) {
    // two-stage software pipeline, backwards: the record of pair j - 2 and the operands of pair j - 1 are requested before pair
    // j is computed (see pass 1).  y2 is only read where a pair is exceptional: its slope's numerator came with `pre`.
    struct Ops {
        F x1, y1, x2;
    };
    auto fetch = [&](const Rec& r, Ops& o) {
        src.load_x(r, o.x1, o.x2);
        src.load_y1_raw(r, o.y1);
    };
    uint32_t j = (P - 1 - t) / NT;
    Rec ra = rec_at(t + j * NT), rb{};
    if (j) rb = rec_at(t + (j - 1) * NT);
    Ops nxt;
    fetch(ra, nxt);
    // (the result of a pair is stored at the top of the NEXT iteration, after that iteration's wait for its operands and before
    // its loads: see pass 1 — a store in flight would otherwise be drained by the wait)
    // (`c = nxt` is 36 register copies per pair.  The loop unrolled by two over two operand sets that swap roles — round 6 — keeps as
    // many: the compiler then copies the values that merge behind the rare path instead; pass 2 25.6 -> 25.9 ms per MSM for twice the
    // code: profiles/r06_same_box_ab_pass2_unrolled_by_two_rejected.txt)
    F hx = O::zero(), hy = O::zero();
    uint32_t hout = 0;
    bool held = false;
    for (;; --j) {
        Ops c = nxt;
        const Rec cr = ra;
        ra = rb;
        if (held) put(hout, hx, hy);
        const F qn = plane_ld(pre, pre_cap, src.at(((size_t)j * np + p) * NT + t));
        if (!j) ra = cr;  // the last iteration asks for its own pair again: nothing reads it
        fetch(ra, nxt);
        rb = rec_at(t + (j > 1 ? j - 2 : 0u) * NT);
        const uint32_t out = TreeSrc<O, L0>::out_index(cr, t + j * NT);
        TreeSrc<O, L0>::fix_y1(cr, c.y1);
        const bool pad2 = cr.y == MSM_PAD_ENTRY;  // the second operand is the padding entry: the result is the first operand (see tree_select)
        F d = O::sub(c.x2, c.x1);
        int kind = TREE_ADD;
        F y2 = O::zero();
        if (!pad2 && (O::is_zero(c.x1) || O::is_zero(c.x2) || O::is_zero(d))) {  // rare
            src.load_y2_raw(cr, y2);
            TreeSrc<O, L0>::fix_y2(cr, y2);
            kind = tree_classify<O>(c.x1, c.y1, c.x2, y2, d);
        }
        d = tree_select(pad2, O::one(), d);
        F x3, y3;
        if (kind <= TREE_DBL) {
            // (I, qn and lam stay in [0, 2p): the square and the product below make x3 and y3 canonical)
            const F Inext = O::mul_lazy(I, d);
            const F lam = O::mul_lazy(I, qn);  // (numerator x denominators before this pair) / (denominators up to this pair)
            I = Inext;
            const F xx = kind == TREE_ADD ? c.x2 : c.x1;
            x3 = O::sub(O::sub(O::sqr(lam), c.x1), xx);
            y3 = O::sub(O::mul(lam, O::sub(c.x1, x3)), c.y1);
        } else if (kind == TREE_FIRST) {
            x3 = c.x1;
            y3 = c.y1;
        } else if (kind == TREE_SECOND) {
            x3 = c.x2;
            y3 = y2;
        } else {
            x3 = O::zero();
            y3 = O::zero();
        }
        hx = tree_select(pad2, c.x1, x3);
        hy = tree_select(pad2, c.y1, y3);
        hout = out;
        held = true;
        if (!j) break;
    }
    put(hout, hx, hy);
    } else {
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
        if (held) put(hout, hx, hy);
        F pp = O::one();
        if (j) pp = plane_ld(pre, pre_cap, src.at(((size_t)(j - 1) * np + p) * NT + t));
        if (!j) ra = cr;  // the last iteration asks for its own pair again: nothing reads it
        fetch(ra, nxt);
        rb = rec_at(t + (j > 1 ? j - 2 : 0u) * NT);
        const uint32_t out = TreeSrc<O, L0>::out_index(cr, t + j * NT);
        TreeSrc<O, L0>::fix_y(cr, c.y1, c.y2);
        // the second operand is the point at infinity (level 1: what two padding entries of level 0 summed to): the result is the first
        // operand, by selects (see tree_select)
        const bool zx2 = O::is_zero(c.x2), inf2 = zx2 && O::is_zero(c.y2);
        F d = O::sub(c.x2, c.x1);
        int kind = TREE_ADD;
        if (!inf2 && (O::is_zero(c.x1) || zx2 || O::is_zero(d))) kind = tree_classify<O>(c.x1, c.y1, c.x2, c.y2, d);
        d = tree_select(inf2, O::one(), d);
        F x3, y3;
        if (kind <= TREE_DBL) {
            const F Inext = O::mul_lazy(I, d);   // I, pp and inv in [0, 2p): lam below is a reducing product of canonical x inv
            const F inv = j ? O::mul_lazy(I, pp) : I;
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
        hx = tree_select(inf2, c.x1, x3);
        hy = tree_select(inf2, c.y1, y3);
        hout = out;
        held = true;
        if (!j) break;
    }
    put(hout, hx, hy);
    }
}

// the last point of a bucket with an odd number of points goes to the next level as it is.  grid (nb / 256, np)
template <class O, bool L0>
__global__ void __launch_bounds__(256)
k_tree_copy(const TabRow<O>* __restrict__ tab, const uint32_t* __restrict__ sorted, size_t ent_stride, const typename O::T* __restrict__ xs,
            const typename O::T* __restrict__ ys, size_t pt_stride, const uint32_t* __restrict__ Dl, const uint32_t* __restrict__ Dn, uint32_t nb,
            typename O::T* __restrict__ ox, typename O::T* __restrict__ oy, size_t out_stride, uint32_t out_whole) {
    const uint32_t p = MSM_P, np = gridDim.y, b = blockIdx.x * blockDim.x + threadIdx.x;
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
        const size_t cap = (size_t)np * pt_stride, e = tree_pt_base<1>(p, pt_stride) + tree_pt_slot<1>(in, 0, cap);
        x = plane_ld(xs, cap, e);
        y = plane_ld(ys, cap, e);
    }
    if (out_whole) {
        ox[(size_t)p * out_stride + out] = x;
        oy[(size_t)p * out_stride + out] = y;
    } else {
        const size_t cap = (size_t)np * out_stride, e = tree_pt_base<1>(p, out_stride) + tree_pt_slot<1>(out, 0, cap);
        plane_st(ox, cap, e, x);
        plane_st(oy, cap, e, y);
    }
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
