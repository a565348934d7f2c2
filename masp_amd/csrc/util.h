// small host-side helpers shared by the product's translation units
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/masp_hip.h"

namespace masp {
// thread-local text of the last HIP error, surfaced through masp_hip_last_error()
inline std::string& last_hip_error() {
    static thread_local std::string s;
    return s;
}
// hipLaunchKernelGGL reports nothing itself: a launch that the runtime refuses (dynamic LDS above the device's limit, a block
// too large, an invalid grid) would only show at the next synchronisation, or as wrong bytes.  MASP_LAUNCH asks hipGetLastError()
// after every launch and keeps the FIRST failure of this host thread here (kernel, file:line, HIP's text); launch_status() hands
// it to the caller as MASP_HIP_E_HIP + masp_hip_last_error() and clears it.
inline std::string& launch_error() {
    static thread_local std::string s;
    return s;
}
inline void note_launch_error(hipError_t e, const char* kernel, const char* file, int line) {
    if (!launch_error().empty()) return;
    char b[640];
    snprintf(b, sizeof(b), "%s:%d: launch of %s -> %s", file, line, kernel, hipGetErrorString(e));
    launch_error() = b;
}
inline int launch_status() {
    if (launch_error().empty()) return MASP_HIP_OK;
    last_hip_error() = launch_error();
    launch_error().clear();
    return MASP_HIP_E_HIP;
}
// First statement of every extern "C" entry point that launches kernels: whatever an earlier call of this thread left behind —
// a latched launch failure that no launch_status() picked up, or a HIP error the embedding application (torch, RCCL ...) left on the
// thread — must not be reported by this call as its own.  Each such entry point asks launch_status() before it reports success.
struct ApiLaunchScope {
    ApiLaunchScope() {
        (void)hipGetLastError();
        launch_error().clear();
    }
};
// Every device allocation and release of the library goes through these two: they count.  A captured launch graph (the lone
// proof's, Slot::graphs) holds raw device pointers of workspaces that grow on demand, so it is valid only as long as nothing
// has been allocated or freed since it was captured — the count is that test (conservative: any allocation anywhere drops
// the graphs; in steady state there are none).
inline std::atomic<uint64_t>& device_alloc_epoch() {
    static std::atomic<uint64_t> e{0};
    return e;
}
// hipFree waits for EVERY stream of the device to drain — a workspace that grows on one slot (a slot meets its first Spend batch
// after Outputs: 34 GB of tree scratch instead of 3) stalled that slot's host thread for as long as the other slots kept the chip
// busy, 1.1 - 1.8 s in a mixed job list (`profiles/r04e_mixed_calls_timing.txt`).  So a released buffer only goes on a list; the
// list is emptied where waiting costs nothing — when the last busy slot of a context is released, when a context is destroyed —
// and when an allocation fails for lack of memory.  What lingers in between is bounded by the buffers' earlier, smaller sizes.
// One list per DEVICE (round 5): with one process-global list an idle release on one device context called hipFree — and its
// device-wide wait — on buffers of OTHER devices that were in the middle of a batch.  A buffer goes on the list of the device that OWNS
// it (asked of the runtime when it is released); dev_free_drain() empties the current device's list.
struct DevGraveyard {
    std::mutex mu;
    std::vector<std::pair<int, void*>> v;  // (device, buffer)
};
inline DevGraveyard& dev_graveyard() {
    static DevGraveyard* g = new DevGraveyard;  // (never destroyed: buffers are released from static destructors too)
    return *g;
}
inline int dev_current() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        d = 0;
    }
    return d;
}
inline void dev_free_drain() {
    const int dev = dev_current();
    std::vector<void*> mine;
    {
        std::lock_guard<std::mutex> g(dev_graveyard().mu);
        auto& v = dev_graveyard().v;
        size_t k = 0;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].first == dev)
                mine.push_back(v[i].second);
            else
                v[k++] = v[i];
        v.resize(k);
    }
    for (void* p : mine) (void)hipFree(p);
}
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) {
    device_alloc_epoch().fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {  // what was released but not yet returned to the device may be what is missing
        const int dev = dev_current();
        bool any = false;
        {
            std::lock_guard<std::mutex> g(dev_graveyard().mu);
            for (auto& x : dev_graveyard().v) any = any || x.first == dev;
        }
        if (any) {
            (void)hipGetLastError();
            dev_free_drain();
            e = hipMalloc(p, bytes);
        }
    }
    return e;
}
inline hipError_t dev_free(void* p) {
    device_alloc_epoch().fetch_add(1, std::memory_order_relaxed);
    // the device that OWNS the buffer, not the one that happens to be current on the releasing thread (a destructor run from static
    // teardown, a Python GC thread, a multi-device front's other thread: ADVICE r05) — the runtime knows; if it no longer answers
    // (process teardown), the current device is as good a guess as any
    int dev = -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice)
        dev = attr.device;
    else
        (void)hipGetLastError();
    if (dev < 0) dev = dev_current();
    std::lock_guard<std::mutex> g(dev_graveyard().mu);
    dev_graveyard().v.push_back({dev, p});
    return hipSuccess;
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

#define MASP_LAUNCH(kernel, grid, block, shmem, stream, ...)                                \
    do {                                                                                    \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                \
        const hipError_t _le = hipGetLastError();                                           \
        if (_le != hipSuccess) masp::note_launch_error(_le, #kernel, __FILE__, __LINE__);   \
    } while (0)

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
