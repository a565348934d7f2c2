// The launch of the dominant kernel, k_msm_accumulate<O>, on its own: one translation unit per curve (k_msm_g1_acc.hip,
// k_msm_g2_acc.hip) so that the hot kernel can be rebuilt in seconds.
#pragma once
#include "device/msm_acc.hpp"
#include "msm_host.h"

namespace masp {

template <class O>
void msm_launch_accumulate(hipStream_t s, const TabRow<O>* tab, const uint32_t* sorted, size_t ent_stride, const uint32_t* start, uint32_t nb,
                           uint32_t nchunks, Xyzz<O>* part, uint32_t np) {
    MASP_LAUNCH((k_msm_accumulate<O>), dim3((nchunks + 63) / 64, np), dim3(64), 0, s, tab, sorted, ent_stride, start, nb, nchunks, part);
}

}  // namespace masp
