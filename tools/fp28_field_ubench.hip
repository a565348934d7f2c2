// The 28-bit-limb field as built (masp_amd/csrc/device/fp28.hpp) against field.hpp's 12 x 32-bit one: chains of products,
// squares, and the additions pass's arithmetic per pair (3 products + 1 square + differences + two canonicalisations), whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "fp28.hpp"
using namespace masp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_mul28(F28* d, int it) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F28 a = d[t], b = d[t ^ 1];
    for (int i = 0; i < it; ++i) { a = fp28_mul(a, b); b = fp28_mul(b, a); }
    d[t] = fp28_add_lazy(a, b);
}
__global__ void __launch_bounds__(256) k_sqr28(F28* d, int it) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F28 a = d[t];
    for (int i = 0; i < it; ++i) { a = fp28_sqr(a); a = fp28_sqr(a); }
    d[t] = a;
}
__global__ void __launch_bounds__(256) k_mul32(Fp* d, int it) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = d[t], b = d[t ^ 1];
    for (int i = 0; i < it; ++i) { a = fe_mul_lazy(a, b); b = fe_mul_lazy(b, a); }
    d[t] = fe_add(a, b);
}
__global__ void __launch_bounds__(256) k_sqr32(Fp* d, int it) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a = d[t];
    for (int i = 0; i < it; ++i) { a = fe_sqr(a); a = fe_sqr(a); }
    d[t] = a;
}
// one affine addition's arithmetic of the tree's pass 2 per iteration (operands recycled)
// FpOps with the hooks Fp28Ops has (every FpOps value is canonical: nothing to do)
struct FpOpsH : FpOps {
    static __device__ __forceinline__ T sub_lazy(const T& a, const T& b) { return sub(a, b); }
    static __device__ __forceinline__ T canon(const T& a) { return a; }
};
template <class O>
__global__ void __launch_bounds__(256) k_pair(typename O::T* d, int it) {
    typedef typename O::T F;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F x1 = d[t], x2 = d[t ^ 1], y1 = d[t ^ 2], I = d[t ^ 3], qn = d[t ^ 4];
    for (int i = 0; i < it; ++i) {
        const F dd = O::sub_lazy(x2, x1);
        const F In = O::mul_lazy(I, dd), lam = O::mul_lazy(I, qn);
        I = In;
        const F x3 = O::canon(O::sub_lazy(O::sub_lazy(O::sqr(lam), x1), x2));
        const F y3 = O::canon(O::sub_lazy(O::mul(lam, O::sub_lazy(x1, x3)), y1));
        x2 = x1; x1 = x3; y1 = y3;
    }
    d[t] = O::canon(O::sub_lazy(I, x1));
}
template <class F, class K>
static void run(const char* name, K kern, F* d, int blocks, int it, double per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, it); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * it * per_iter;
    printf("%-58s %8.3f ms  %8.2f G/s  (%.0f SIMD-cycles@2.4GHz per wave op)\n", name, ms, n / ms / 1e6, ms * 1e-3 * 2.4e9 / (n / 64 / 1024));
}
int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8, n = blocks * 256, it = 200;
    F28* d28; CHECK(hipMalloc(&d28, (size_t)n * sizeof(F28))); CHECK(hipMemset(d28, 3, (size_t)n * sizeof(F28)));   // limbs 0x03030303 < 2^28
    Fp* d32; CHECK(hipMalloc(&d32, (size_t)n * sizeof(Fp))); CHECK(hipMemset(d32, 1, (size_t)n * sizeof(Fp)));
    run("fp28_mul (14 x 28 bits, no carry word)", k_mul28, d28, blocks, it, 2);
    run("fe_mul_lazy (12 x 32 bits)", k_mul32, d32, blocks, it, 2);
    run("fp28_sqr", k_sqr28, d28, blocks, it, 2);
    run("fe_sqr", k_sqr32, d32, blocks, it, 2);
    CHECK(hipMemset(d28, 0, (size_t)n * sizeof(F28)));
    run("pass-2 arithmetic per pair, Fp28Ops", k_pair<Fp28Ops>, d28, blocks, it / 2, 1);
    run("pass-2 arithmetic per pair, FpOps", k_pair<FpOpsH>, d32, blocks, it / 2, 1);
    return 0;
}
