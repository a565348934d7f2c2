// small host-side helpers shared by the product's translation units
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/masp_hip.h"

namespace masp {
// thread-local text of the last HIP error, surfaced through masp_hip_last_error()
inline std::string& last_hip_error() {
    static thread_local std::string s;
    return s;
}
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
