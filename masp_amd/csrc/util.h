// small host-side helpers shared by the product's translation units
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "../../include/masp_hip.h"

namespace masp {
// thread-local text of the last HIP error, surfaced through masp_hip_last_error()
inline std::string& last_hip_error() {
    static thread_local std::string s;
    return s;
}
// hipFuncSetAttribute applies to the CURRENT device: a process that proves on several GPUs (masp_hip_ctx_create_multi, one
// host thread per device) has to raise a kernel's dynamic-LDS limit on each of them.  One instance per call site; `f` runs
// once per device and its verdict is remembered.
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t tried[2] = {0, 0}, ok[2] = {0, 0};  // bit per device index (< 128)
    template <class F>
    bool operator()(F f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 128) return false;
        std::lock_guard<std::mutex> g(mu);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(tried[dev >> 6] & bit)) {
            tried[dev >> 6] |= bit;
            if (f()) ok[dev >> 6] |= bit;
        }
        return (ok[dev >> 6] & bit) != 0;
    }
};
}  // namespace masp

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            char _b[512];                                                                                 \
            snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            masp::last_hip_error() = _b;                                                                  \
            return MASP_HIP_E_HIP;                                                                        \
        }                                                                                                 \
    } while (0)
