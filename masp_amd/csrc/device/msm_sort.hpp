// Counting sort of the signed window digits of an MSM by bucket (stages 1-3 of device/msm.hpp's plan): curve-independent,
// compiled once (k_msm_sort.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "msm_geom.h"

namespace masp {

#define MSM_P (blockIdx.y)

// ---- (1)-(3) counting sort of the signed digits by bucket, without global atomics -------------------------
// The scalars of one proof are cut into `ng` contiguous ranges, one workgroup each.  A workgroup keeps the whole bucket
// histogram (2^(c-1) counters, 128 KiB for c = 16) in LDS:
//   k_msm_hist     counts the digits of its range into LDS and stores the histogram            hist_wg[p][wg][b]
//   k_msm_offsets_*  turn them into  rel[p][wg][b] = entries of bucket b in earlier ranges  and  start[p][b]
//   k_msm_scatter  reloads  start[b] + rel[wg][b]  into LDS, recomputes the digits of the same range and places every
//                  entry with one LDS atomic.
// Scalars equal to 1 (a third of a MASP witness: booleans) all land in bucket 0 of window 0; a wave counts / places them
// with one ballot instead of 64 colliding atomics.  Zero scalars (38 %) produce nothing.
// scalars: n x 8 canonical little-endian limbs.  sorted entry = table row (j*n + i) | sign << 31, j = the window of the digit.
// (Round 5 also had width-w NAF digits over a table per bit position here, as a template parameter of the iterator and its three kernels;
// removed in round 6 — the tables it needs are beyond an XCD's TLB reach: EXPERIMENTS.md, profiles/r05_naf_digits_*.txt.)
struct MsmDigitIter {
    const uint32_t* sw;
    uint32_t carry, mask, half, pos, j;
    int c;
    __device__ __forceinline__ MsmDigitIter(const uint32_t* sw_, const MsmGeom& g)
        : sw(sw_), carry(0), mask((1u << g.c) - 1u), half(1u << (g.c - 1)), pos(0), j(0), c(g.c) {}
    // 32 bits of the scalar from bit `bit` on, zeros beyond bit 255 (re-read from L1/L2 instead of indexing a register array dynamically;
    // the last window starts below bit 256)
    __device__ __forceinline__ uint32_t bits(uint32_t bit) const {
        const uint32_t w = bit >> 5, off = bit & 31u;
        const uint64_t two = ((uint64_t)(w + 1 < 8 ? sw[w + 1] : 0u) << 32) | sw[w];
        return (uint32_t)(two >> off);
    }
    // The next digit: call W times, window after window; false = this window's digit is zero.  table: the window.
    __device__ __forceinline__ bool next(uint32_t& table, uint32_t& bucket, uint32_t& neg) {
        uint32_t v = (bits(pos) & mask) + carry;
        table = j++;
        pos += (uint32_t)c;
        neg = 0;
        carry = 0;
        if (v > half) {
            v = (1u << c) - v;
            neg = 1;
            carry = 1;
        }
        bucket = v - 1;
        return v != 0;
    }
};
// 0: zero, 1: one, 2: anything else
__device__ __forceinline__ int msm_scalar_class(const uint32_t* sw) {
    const uint4* sp = reinterpret_cast<const uint4*>(sw);
    uint4 lo = sp[0], hi = sp[1];
    uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
    if (rest == 0 && lo.x <= 1) return (int)lo.x;
    return 2;
}
__global__ void __launch_bounds__(1024)
k_msm_hist(const uint32_t* __restrict__ scalars, size_t scalar_stride, uint32_t n, MsmGeom g, uint32_t ng, uint32_t* __restrict__ hist_wg) {
    extern __shared__ uint32_t msm_lds[];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, nb = (uint32_t)g.nb;
    scalars += MSM_P * scalar_stride;
    hist_wg += ((size_t)MSM_P * ng + wg) * nb;
    for (uint32_t b = tid; b < nb; b += MSM_SORT_THREADS) msm_lds[b] = 0;
    __syncthreads();
    const uint32_t per = (n + ng - 1) / ng, lo = wg * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = lo; base < hi; base += MSM_SORT_THREADS) {
        const uint32_t i = base + tid;
        const uint32_t* sw = scalars + (size_t)i * 8;
        const int cls = i < hi ? msm_scalar_class(sw) : 0;
        const uint64_t ones = __ballot(cls == 1);
        if (cls == 1) {
            if ((uint32_t)__ffsll((unsigned long long)ones) - 1u == (tid & 63u)) atomicAdd(&msm_lds[0], (uint32_t)__popcll(ones));
        } else if (cls == 2) {
            MsmDigitIter it(sw, g);
            for (int j = 0; j < g.W; ++j) {
                uint32_t table, bucket, neg;
                if (it.next(table, bucket, neg)) atomicAdd(&msm_lds[bucket], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < nb; b += MSM_SORT_THREADS) hist_wg[b] = msm_lds[b];
}
// hist_wg[wg][b] -> rel[wg][b] (in place); dense[0..nb] = offsets of the runs packed (dense[nb] = number of entries),
// start[0..nb] = offsets with every run starting at a multiple of 2^pad_log (start[nb] likewise rounded up).  Two kernels (round 5:
// as ONE workgroup per proof — which read and rewrote the 64 ranges x 32 768 buckets of a lone proof's h query, 8 MB, by itself —
// the step took 0.22 ms of the 2.2 ms a lone proof waits for that MSM):
//   k_msm_offsets_cols  grid (nb / 256, np): one lane per bucket walks the ranges (sixteen loads in flight), leaves the bucket's
//                       total in dense[b]
//   k_msm_offsets_scan  one workgroup per proof: exclusive scans of the totals and of the padded totals, 1 024 buckets at a time
__global__ void __launch_bounds__(256)
k_msm_offsets_cols(uint32_t* __restrict__ hist_wg, uint32_t ng, uint32_t nb, uint32_t* __restrict__ dense) {
    hist_wg += (size_t)MSM_P * ng * nb;
    dense += (size_t)MSM_P * (nb + 1);
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t v = 0;
    for (uint32_t w0 = 0; w0 < ng; w0 += 16) {
        uint32_t h[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) h[k] = w0 + k < ng ? hist_wg[(size_t)(w0 + k) * nb + b] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k)
            if (w0 + k < ng) {
                hist_wg[(size_t)(w0 + k) * nb + b] = v;
                v += h[k];
            }
    }
    dense[b] = v;
}
// Exclusive scans of the buckets' totals (packed: dense[]) and of the totals rounded up to 2^pad_log (aligned: start[]), in two kernels of
// SMALL workgroups: grid (ceil(nb / 1024), np) x 256 lanes, four buckets per lane.
//   k_msm_offsets_scan_a  the scan inside a block of 1 024 buckets, the block's two totals to btot[p][blk]
//   k_msm_offsets_scan_b  every lane adds the totals of the blocks before its own; the last block leaves dense[nb], start[nb]
// As ONE workgroup of 1 024 lanes per proof the step took a lone proof's h MSM 49 us — whatever the kernel did inside (two rewrites changed
// nothing): a workgroup of sixteen waves waits until one CU has four free wave slots on each SIMD, and the chip is full of the other chains'
// accumulation waves, which run for 200 - 500 us each.  Workgroups of four waves find room at once.
static constexpr uint32_t MSM_SCAN_BLOCK = 1024;  // buckets per workgroup (256 lanes x 4)
__global__ void __launch_bounds__(256)
k_msm_offsets_scan_a(uint32_t nb, uint32_t* __restrict__ start, uint32_t* __restrict__ dense, uint32_t pad_log, uint2* __restrict__ btot) {
    __shared__ uint32_t wsum[2][4];
    start += (size_t)MSM_P * (nb + 1);
    dense += (size_t)MSM_P * (nb + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, pad = (1u << pad_log) - 1u, b0 = blockIdx.x * MSM_SCAN_BLOCK + tid * 4u;
    uint32_t v[4], pv[4], sv = 0, spv = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        v[k] = b0 + k < nb ? dense[b0 + k] : 0u;
        pv[k] = (v[k] + pad) & ~pad;
        sv += v[k];
        spv += pv[k];
    }
    uint32_t x = sv, px = spv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64), py = __shfl_up(px, d, 64);
        if ((int)lane >= d) {
            x += y;
            px += py;
        }
    }
    if (lane == 63) {
        wsum[0][wid] = x;
        wsum[1][wid] = px;
    }
    __syncthreads();
    uint32_t woff = 0, pwoff = 0;
    for (uint32_t k = 0; k < wid; ++k) {
        woff += wsum[0][k];
        pwoff += wsum[1][k];
    }
    uint32_t run = woff + x - sv, prun = pwoff + px - spv;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (b0 + k < nb) {
            dense[b0 + k] = run;
            start[b0 + k] = prun;
        }
        run += v[k];
        prun += pv[k];
    }
    if (tid == 255) btot[(size_t)MSM_P * gridDim.x + blockIdx.x] = make_uint2(run, prun);
}
__global__ void __launch_bounds__(256)
k_msm_offsets_scan_b(uint32_t nb, uint32_t* __restrict__ start, uint32_t* __restrict__ dense, const uint2* __restrict__ btot) {
    start += (size_t)MSM_P * (nb + 1);
    dense += (size_t)MSM_P * (nb + 1);
    btot += (size_t)MSM_P * gridDim.x;
    // (lane k of every wave fetches block k's totals — at most 32 blocks: 2^15 buckets —, one shuffle reduction: a loop over the blocks
    // before was as many dependent round trips to L2)
    const uint32_t lane = threadIdx.x & 63u;
    uint2 t = make_uint2(0u, 0u);
    if (lane < blockIdx.x) t = btot[lane];
    uint32_t base = t.x, pbase = t.y;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        base += __shfl_xor(base, d, 64);
        pbase += __shfl_xor(pbase, d, 64);
    }
    const uint32_t b0 = blockIdx.x * MSM_SCAN_BLOCK + threadIdx.x * 4u;
    if (blockIdx.x) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (b0 + k < nb) {
                dense[b0 + k] += base;
                start[b0 + k] += pbase;
            }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const uint2 last = btot[blockIdx.x];
        dense[nb] = base + last.x;
        start[nb] = pbase + last.y;
    }
}
__global__ void __launch_bounds__(1024)
k_msm_scatter(const uint32_t* __restrict__ scalars, size_t scalar_stride, uint32_t n, MsmGeom g, uint32_t ng, const uint32_t* __restrict__ rel,
              const uint32_t* __restrict__ start, uint32_t* __restrict__ sorted, size_t sorted_stride) {
    extern __shared__ uint32_t msm_lds[];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, nb = (uint32_t)g.nb;
    scalars += MSM_P * scalar_stride;
    rel += ((size_t)MSM_P * ng + wg) * nb;
    start += (size_t)MSM_P * (nb + 1);
    sorted += (size_t)MSM_P * sorted_stride;
    for (uint32_t b = tid; b < nb; b += MSM_SORT_THREADS) msm_lds[b] = start[b] + rel[b];
    __syncthreads();
    const uint32_t per = (n + ng - 1) / ng, lo = wg * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = lo; base < hi; base += MSM_SORT_THREADS) {
        const uint32_t i = base + tid;
        const uint32_t* sw = scalars + (size_t)i * 8;
        const int cls = i < hi ? msm_scalar_class(sw) : 0;
        const uint64_t ones = __ballot(cls == 1);
        if (ones) {
            const int leader = __ffsll((unsigned long long)ones) - 1;
            uint32_t first = 0;
            if ((int)(tid & 63u) == leader) first = atomicAdd(&msm_lds[0], (uint32_t)__popcll(ones));
            first = __shfl(first, leader, 64);
            if (cls == 1) sorted[first + (uint32_t)__popcll(ones & ((1ull << (tid & 63u)) - 1ull))] = i;  // window 0: row i, positive
        }
        if (cls == 2) {
            MsmDigitIter it(sw, g);
            for (int j = 0; j < g.W; ++j) {
                uint32_t table, bucket, neg;
                if (it.next(table, bucket, neg)) sorted[atomicAdd(&msm_lds[bucket], 1u)] = (table * n + i) | (neg << 31);
            }
        }
    }
}


// ---- two-pass placement (replaces k_msm_scatter when the entries fit 24 bits) ------------------------------------------
// k_msm_scatter writes every entry on its own: a wave's 64 entries go to 64 unrelated buckets, i.e. 64 four-byte writes
// into 64 different lines — the kernel is bound by DRAM read-modify-write of partial lines (10.4 ms per 128-proof batch for
// 1.76 GB of payload).  Here the same entries reach their places in two steps that only ever write runs:
//   k_msm_partition  (one workgroup per scalar range)   entries -> COARSE bins of 128 buckets.  A tile of 1024 scalars is
//                    sorted by bin inside LDS first (count, scan, place), then copied out bin by bin: runs of ~64 entries.
//   k_msm_bucketize  (one workgroup per proof and bin)  the bin's entries -> their buckets, again through an LDS-sorted
//                    tile of 4096 entries: runs of ~32 entries.
// Between the two, an entry carries its bucket's low 7 bits:  row (24 bits) | fine << 24 | sign << 31 — or, where the table has more than
// 2^24 rows (`wide`: more than a million points on 16 windows), the 7 bits travel in a byte array of their own
// (`tmpf`, next to `tmp`) and the word keeps 31 bits for the row.
static constexpr uint32_t MSM_FINE_LOG = 7, MSM_FINE = 1u << MSM_FINE_LOG;
static constexpr uint32_t MSM_PART_TILE = 1024;   // scalars per tile of k_msm_partition (= threads)
static constexpr uint32_t MSM_BKT_TILE = 4096;    // entries per tile of k_msm_bucketize

// crel[p][wg][B] = entries of bin B that come from earlier scalar ranges = sum over the bin's buckets of rel[p][wg][b].  One WAVE per
// (bin, range): grid (bins, np, ceil(ng / waves per workgroup)) — as one workgroup per bin walking the ranges one after the other (two
// barriers each) the kernel was 110 us of a lone proof's h MSM, whose 64 ranges it took in turn (round 5).
__global__ void __launch_bounds__(256) k_msm_coarse(const uint32_t* __restrict__ rel, uint32_t ng, uint32_t nb, uint32_t* __restrict__ crel) {
    const uint32_t B = blockIdx.x, nbins = nb >> MSM_FINE_LOG, lane = threadIdx.x & 63u, w = blockIdx.z * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= ng) return;
    rel += (size_t)MSM_P * ng * nb + (size_t)w * nb + (B << MSM_FINE_LOG);
    uint32_t v = rel[lane] + rel[64 + lane];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) crel[((size_t)MSM_P * ng + w) * nbins + B] = v;
}
// position of k in the exclusive offsets off[0..n): largest i with off[i] <= k
__device__ __forceinline__ uint32_t msm_find_run(const uint32_t* off, uint32_t n, uint32_t k) {
    uint32_t i = 0, span = n;
    while (span > 1) {
        uint32_t half = span >> 1;
        if (off[i + half] <= k) i += half;
        span -= half;
    }
    return i;
}
// exclusive scan of cnt[0..n) (n <= 256) into off[], by the first four waves of the workgroup (shuffle scan inside each wave,
// the waves' totals through wsum[4]); also clears fill[].  Two barriers inside: every thread of the workgroup must call it.
__device__ __forceinline__ void msm_small_scan(const uint32_t* cnt, uint32_t* off, uint32_t* fill, uint32_t n, uint32_t* total, uint32_t* wsum) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t v = 0, x = 0;
    if (tid < 256) {
        v = tid < n ? cnt[tid] : 0u;
        x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) wsum[tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t base = 0;
        for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
        if (tid < n) {
            off[tid] = base + x - v;
            fill[tid] = 0;
        }
        if (tid == 255) *total = base + x;
    }
}
__global__ void __launch_bounds__(1024)
k_msm_partition(const uint32_t* __restrict__ scalars, size_t scalar_stride, uint32_t n, MsmGeom g, uint32_t ng, const uint32_t* __restrict__ crel,
                const uint32_t* __restrict__ start /* packed offsets: MsmSortBuf::dense */, uint32_t* __restrict__ tmp, uint8_t* __restrict__ tmpf,
                uint32_t wide) {
    extern __shared__ uint32_t msm_lds[];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, nb = (uint32_t)g.nb, nbins = nb >> MSM_FINE_LOG;
    uint32_t* cursor = msm_lds;            // [256] global position of the next entry of each bin from this workgroup
    uint32_t* cnt = cursor + 256;          // [256] entries of the tile per bin
    uint32_t* off = cnt + 256;             // [256] their exclusive scan
    uint32_t* fill = off + 256;            // [256]
    uint32_t* total = fill + 256;          // [1] (+3 pad)
    uint32_t* wsum = total + 4;            // [4]
    uint32_t* stage = wsum + 4;            // [MSM_PART_TILE * W]
    uint8_t* stageb = reinterpret_cast<uint8_t*>(stage + MSM_PART_TILE * (uint32_t)g.W);  // [MSM_PART_TILE * W] the bin of a staged entry
    uint8_t* stagef = stageb + MSM_PART_TILE * (uint32_t)g.W;                              // [MSM_PART_TILE * W] its low bucket bits (wide only)
    scalars += MSM_P * scalar_stride;
    crel += ((size_t)MSM_P * ng + wg) * nbins;
    start += (size_t)MSM_P * (nb + 1);
    tmp += (size_t)MSM_P * n * g.W;
    tmpf += (size_t)MSM_P * n * g.W;
    if (tid < nbins) cursor[tid] = start[tid << MSM_FINE_LOG] + crel[tid];
    const uint32_t per = (n + ng - 1) / ng, lo = wg * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = lo; base < hi; base += MSM_PART_TILE) {
        if (tid < nbins) cnt[tid] = 0;
        __syncthreads();
        const uint32_t i = base + tid;
        const uint32_t* sw = scalars + (size_t)i * 8;
        const int cls = i < hi ? msm_scalar_class(sw) : 0;
        const uint64_t ones = __ballot(cls == 1);
        const int leader = ones ? __ffsll((unsigned long long)ones) - 1 : -1;
        // ONE pass over the digits of a tile: the counting atomic already hands out the entry's rank inside its bin, the entry
        // and (rank, bin, low bucket bits) wait in registers for the scan of the counts (the digits were extracted and counted twice before)
        uint32_t ent[32], key[32];  // key: rank (16 bits: at most 1024 x 32 entries per tile) | bin << 16 | low bucket bits << 24
        uint32_t unit_rank = 0;
        if (ones) {  // unit scalars: table 0, bucket 0, positive
            if ((int)(tid & 63u) == leader) unit_rank = atomicAdd(&cnt[0], (uint32_t)__popcll(ones));
            unit_rank = __shfl(unit_rank, leader, 64) + (uint32_t)__popcll(ones & ((1ull << (tid & 63u)) - 1ull));
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) key[j] = 0xffffffffu;
        if (cls == 2) {
            MsmDigitIter it(sw, g);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j < g.W) {
                    uint32_t table, bucket, neg;
                    if (it.next(table, bucket, neg)) {
                        const uint32_t B = bucket >> MSM_FINE_LOG;
                        key[j] = atomicAdd(&cnt[B], 1u) | (B << 16) | ((bucket & (MSM_FINE - 1u)) << 24);
                        ent[j] = (table * n + i) | (neg << 31) | (wide ? 0u : (bucket & (MSM_FINE - 1u)) << 24);
                    }
                }
            }
        }
        __syncthreads();
        msm_small_scan(cnt, off, fill, nbins, total, wsum);
        __syncthreads();
        if (cls == 1) {
            stage[off[0] + unit_rank] = i;
            stageb[off[0] + unit_rank] = 0;
            if (wide) stagef[off[0] + unit_rank] = 0;
        }
        if (cls == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (key[j] != 0xffffffffu) {
                    const uint32_t at = off[(key[j] >> 16) & 0xffu] + (key[j] & 0xffffu);
                    stage[at] = ent[j];
                    stageb[at] = (uint8_t)(key[j] >> 16);
                    if (wide) stagef[at] = (uint8_t)(key[j] >> 24);
                }
        }
        __syncthreads();
        const uint32_t tot = *total;
        for (uint32_t k = tid; k < tot; k += MSM_PART_TILE) {
            const uint32_t B = stageb[k];  // (a binary search in off[] — eight dependent LDS loads per entry — found it before round 5)
            const uint32_t at = cursor[B] + (k - off[B]);
            tmp[at] = stage[k];
            if (wide) tmpf[at] = stagef[k];
        }
        __syncthreads();
        if (tid < nbins) cursor[tid] += cnt[tid];
    }
}
__global__ void __launch_bounds__(1024)
k_msm_bucketize(const uint32_t* __restrict__ tmp, const uint8_t* __restrict__ tmpf, size_t tmp_stride, const uint32_t* __restrict__ dense,
                const uint32_t* __restrict__ start, uint32_t nb, uint32_t* __restrict__ sorted, size_t sorted_stride, uint32_t wide) {
    __shared__ uint32_t cur[MSM_FINE], cnt[MSM_FINE], off[MSM_FINE], fill[MSM_FINE], total[4], wsum[4], stage[MSM_BKT_TILE];
    __shared__ uint8_t stagef[MSM_BKT_TILE];  // the low bucket bits of a staged entry (wide: the entry word does not carry them)
    const uint32_t tid = threadIdx.x, B = blockIdx.x;
    tmp += MSM_P * tmp_stride;
    tmpf += MSM_P * tmp_stride;
    sorted += MSM_P * sorted_stride;
    start += (size_t)MSM_P * (nb + 1);
    dense += (size_t)MSM_P * (nb + 1);
    const uint32_t b0 = B << MSM_FINE_LOG, lo = dense[b0], hi = dense[b0 + MSM_FINE];  // the bin in `tmp`: packed
    if (tid < MSM_FINE) cur[tid] = start[b0 + tid];                                      // its buckets in `sorted`: aligned runs
    for (uint32_t base = lo; base < hi; base += MSM_BKT_TILE) {
        if (tid < MSM_FINE) cnt[tid] = 0;
        __syncthreads();
        uint32_t e[MSM_BKT_TILE / 1024], fr[MSM_BKT_TILE / 1024];  // fr: low bucket bits | rank << 8 (the counting atomic hands out the rank inside the bucket)
#pragma unroll
        for (uint32_t q = 0; q < MSM_BKT_TILE / 1024; ++q) {
            const uint32_t k = base + q * 1024 + tid;
            const bool valid = k < hi;
            e[q] = valid ? tmp[k] : 0xffffffffu;
            const uint32_t f = !valid ? 0u : wide ? (uint32_t)tmpf[k] : (e[q] >> 24) & (MSM_FINE - 1u);
            // (a bucket that holds most of a tile — the unit scalars' bucket 0 — is counted once per wave, not 64 times)
            const uint64_t same = __ballot(valid && f == 0);
            uint32_t rank = 0;
            if (same) {
                const int leader = __ffsll((unsigned long long)same) - 1;
                uint32_t first = 0;
                if ((int)(tid & 63u) == leader) first = atomicAdd(&cnt[0], (uint32_t)__popcll(same));
                first = __shfl(first, leader, 64);
                if (valid && f == 0) rank = first + (uint32_t)__popcll(same & ((1ull << (tid & 63u)) - 1ull));
            }
            if (valid && f != 0) rank = atomicAdd(&cnt[f], 1u);
            fr[q] = f | (rank << 8);
        }
        __syncthreads();
        msm_small_scan(cnt, off, fill, MSM_FINE, total, wsum);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < MSM_BKT_TILE / 1024; ++q) {
            const uint32_t k = base + q * 1024 + tid;
            if (k < hi) {
                const uint32_t at = off[fr[q] & 0xffu] + (fr[q] >> 8);
                stage[at] = e[q];
                if (wide) stagef[at] = (uint8_t)fr[q];
            }
        }
        __syncthreads();
        const uint32_t tot = *total;
        for (uint32_t k = tid; k < tot; k += 1024) {
            const uint32_t v = stage[k], f = wide ? (uint32_t)stagef[k] : (v >> 24) & (MSM_FINE - 1u);  // (no search in off[]: the entry knows its bucket)
            sorted[cur[f] + (k - off[f])] = wide ? v : v & 0x80ffffffu;
        }
        __syncthreads();
        if (tid < MSM_FINE) cur[tid] += cnt[tid];
    }
    // aligned runs: the gap between the end of a run and the start of the next holds the point at infinity
    __syncthreads();
    if (tid < MSM_FINE)
        for (uint32_t k = cur[tid], end = start[b0 + tid + 1]; k < end; ++k) sorted[k] = MSM_PAD_ENTRY;
}

}  // namespace masp
