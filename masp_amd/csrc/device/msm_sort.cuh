// Counting sort of the signed window digits of an MSM by bucket (stages 1-3 of device/msm.cuh's plan): curve-independent,
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
//   k_msm_offsets  turns them into  rel[p][wg][b] = entries of bucket b in earlier ranges  and  start[p][b]
//   k_msm_scatter  reloads  start[b] + rel[wg][b]  into LDS, recomputes the digits of the same range and places every
//                  entry with one LDS atomic.
// Scalars equal to 1 (a third of a MASP witness: booleans) all land in bucket 0 of window 0; a wave counts / places them
// with one ballot instead of 64 colliding atomics.  Zero scalars (38 %) produce nothing.
// scalars: n x 8 canonical little-endian limbs.  sorted entry = table row (j*n + i) | sign << 31.
struct MsmDigitIter {
    const uint32_t* sw;
    uint32_t carry, mask, half;
    int c;
    __device__ __forceinline__ MsmDigitIter(const uint32_t* sw_, int c_) : sw(sw_), carry(0), mask((1u << c_) - 1u), half(1u << (c_ - 1)), c(c_) {}
    // digit of window j (call with j = 0, 1, 2, ... in order); false if it is zero
    __device__ __forceinline__ bool next(int j, uint32_t& bucket, uint32_t& neg) {
        int bit = j * c;
        int w = bit >> 5, off = bit & 31;
        // (re-read from L1/L2 instead of indexing a register array dynamically)
        uint64_t two = ((uint64_t)(w + 1 < 8 ? sw[w + 1] : 0u) << 32) | sw[w];
        uint32_t v = ((uint32_t)(two >> off) & mask) + carry;
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
            MsmDigitIter it(sw, g.c);
            for (int j = 0; j < g.W; ++j) {
                uint32_t bucket, neg;
                if (it.next(j, bucket, neg)) atomicAdd(&msm_lds[bucket], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < nb; b += MSM_SORT_THREADS) hist_wg[b] = msm_lds[b];
}
// one workgroup per proof: hist_wg[wg][b] -> rel[wg][b] (in place), start[0..nb] (start[nb] = number of entries)
__global__ void __launch_bounds__(1024) k_msm_offsets(uint32_t* __restrict__ hist_wg, uint32_t ng, uint32_t nb, uint32_t* __restrict__ start) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t base;
    hist_wg += (size_t)MSM_P * ng * nb;
    start += (size_t)MSM_P * (nb + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += blockDim.x) {
        const uint32_t b = b0 + tid;
        uint32_t v = 0;
        if (b < nb)
            for (uint32_t w = 0; w < ng; ++w) {
                uint32_t h = hist_wg[(size_t)w * nb + b];
                hist_wg[(size_t)w * nb + b] = v;
                v += h;
            }
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t k = 0; k < wid; ++k) woff += wsum[k];
        const uint32_t bs = base;
        if (b < nb) start[b] = bs + woff + x - v;
        __syncthreads();
        if (tid == blockDim.x - 1) base = bs + woff + x;
        __syncthreads();
    }
    if (tid == 0) start[nb] = base;
}
__global__ void __launch_bounds__(1024)
k_msm_scatter(const uint32_t* __restrict__ scalars, size_t scalar_stride, uint32_t n, MsmGeom g, uint32_t ng, const uint32_t* __restrict__ rel,
              const uint32_t* __restrict__ start, uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t msm_lds[];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, nb = (uint32_t)g.nb;
    scalars += MSM_P * scalar_stride;
    rel += ((size_t)MSM_P * ng + wg) * nb;
    start += (size_t)MSM_P * (nb + 1);
    sorted += (size_t)MSM_P * n * g.W;
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
            MsmDigitIter it(sw, g.c);
            for (int j = 0; j < g.W; ++j) {
                uint32_t bucket, neg;
                if (it.next(j, bucket, neg)) sorted[atomicAdd(&msm_lds[bucket], 1u)] = ((uint32_t)j * n + i) | (neg << 31);
            }
        }
    }
}

}  // namespace masp
