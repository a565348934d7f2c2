// How many hardware queues do N streams of this process get, and when does the HIP runtime read GPU_MAX_HW_QUEUES?
//   hwq_probe <streams> [setenv-value]     — with a second argument the variable is set INSIDE main(), before the first HIP call
// Launches one spinning kernel per stream at once and reports how many ran concurrently (the maximum overlap of their
// [start, end] device timestamps): streams that share a hardware queue run one after the other.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(unsigned long long* out, int idx, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    out[2 * idx] = t0;
    out[2 * idx + 1] = wall_clock64();
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    if (argc > 2) setenv("GPU_MAX_HW_QUEUES", argv[2], 1);
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    std::vector<hipStream_t> s(n);
    for (auto& x : s)
        if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) return 1;
    unsigned long long* d;
    hipMalloc(&d, 16 * n);
    for (int r = 0; r < 2; ++r) {  // (first round: warm-up)
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], d, i, 20000000LL);  // 100 MHz clock: 0.2 s
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * n);
    hipMemcpy(h.data(), d, 16 * n, hipMemcpyDeviceToHost);
    int best = 0;
    for (int i = 0; i < n; ++i) {
        int c = 0;
        for (int j = 0; j < n; ++j) c += h[2 * j] <= h[2 * i] && h[2 * i] < h[2 * j + 1];
        best = std::max(best, c);
    }
    printf("streams %d  GPU_MAX_HW_QUEUES=%s%s  concurrently running kernels: %d\n", n, e ? e : "(unset)", argc > 2 ? " (set inside main)" : "", best);
    return 0;
}
