// Stage (4) of the MSM (see device/msm.hpp for the plan): bucket accumulation, the dominant kernel of the whole prover.
#pragma once
#include <hip/hip_runtime.h>

#include "curve.hpp"
#include "msm_geom.h"

namespace masp {

#ifndef MSM_P
#define MSM_P (blockIdx.y)
#endif

// ---- (4) accumulate: equal chunks of the sorted list ------------------------------------------------
// start[0..nb] from the scan (start[nb] = number of entries).  Lane `ch` owns entries [ch*K, ch*K + K) with
// K = ceil(total / nchunks); the partial sum of its run inside bucket b goes to part[ch + b] — a slot no other
// (chunk, bucket) pair can hit, because chunk and bucket indices both only grow along the list.
#ifndef MASP_ACC_MIN_WAVES
#define MASP_ACC_MIN_WAVES 1  // (2 was measured: the G2 kernel then spills 1 590 registers and the batch runs 11 % slower)
#endif
template <class O>
__global__ void __launch_bounds__(64, MASP_ACC_MIN_WAVES)
k_msm_accumulate(const TabRow<O>* __restrict__ tab, const uint32_t* __restrict__ sorted, size_t ent_stride,
                 const uint32_t* __restrict__ start, uint32_t nb, uint32_t nchunks, Xyzz<O>* __restrict__ part) {
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nchunks) return;
    sorted += MSM_P * ent_stride;
    start += (size_t)MSM_P * (nb + 1);
    part += (size_t)MSM_P * ((size_t)nchunks + nb);
    const uint32_t total = start[nb];
    const uint32_t K = msm_chunk_len(total, nchunks);
    const uint32_t lo = ch * K;
    if (lo >= total) return;
    const uint32_t hi = lo + K < total ? lo + K : total;
    // bucket of the first entry: largest b with start[b] <= lo (and a non-empty run there)
    uint32_t b = 0, span = nb;
    while (span > 1) {
        uint32_t half = span >> 1;
        if (start[b + half] <= lo) b += half;
        span -= half;
    }
    uint32_t next = start[b + 1];
    Xyzz<O> acc = xyzz_inf<O>();
    for (uint32_t pos = lo; pos < hi; ++pos) {
        if (pos >= next) {
            part[ch + b] = acc;
            acc = xyzz_inf<O>();
            do {
                ++b;
                next = start[b + 1];
            } while (pos >= next);
        }
        uint32_t e = sorted[pos];
        if (e == MSM_PAD_ENTRY) continue;  // aligned runs (MsmSortBuf::pad_log): the padding behind a run
        xyzz_madd(acc, tab[e & 0x7fffffffu].p, (e >> 31) != 0);
    }
    part[ch + b] = acc;
}

}  // namespace masp
