// k_ntt_pass in isolation: the two LDS passes of a 2^17 transform over a sub-batch of proofs, and diagnostic variants
// (twiddles from one address / no global traffic) that show where the time goes.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../masp_amd/csrc/device/ntt.hpp"
using namespace masp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// DIAG 0 = the product kernel's body; 1 = every twiddle from tw[1]; 2 = no global loads / stores of the data
template <int DIAG>
__global__ void __launch_bounds__(256) k_pass_diag(Fr* __restrict__ data, const Fr* __restrict__ tw, uint32_t logm, uint32_t s0, uint32_t nst) {
    __shared__ uint4 tile[2 << NTT_LT];
    const uint32_t lt = nst + (NTT_LT - nst < logm - nst ? NTT_LT - nst : logm - nst);
    const uint32_t cols_log = lt - nst, cols = 1u << cols_log, tsize = 1u << lt, col0 = blockIdx.x << cols_log, lomask = (1u << s0) - 1u;
    data += (size_t)NTT_P << logm;
    for (uint32_t L = threadIdx.x; L < tsize; L += blockDim.x) {
        uint32_t h = L >> cols_log, col = col0 + (L & (cols - 1));
        uint32_t idx = ((col >> s0) << (s0 + nst)) | (h << s0) | (col & lomask);
        if (DIAG == 2) idx = threadIdx.x;
        const uint4* q = reinterpret_cast<const uint4*>(data + idx);
        tile[L] = q[0];
        tile[tsize + L] = q[1];
    }
    __syncthreads();
    auto ld = [&](uint32_t L) {
        Fr r;
        uint4 a0 = tile[L], a1 = tile[tsize + L];
        r.v[0] = a0.x; r.v[1] = a0.y; r.v[2] = a0.z; r.v[3] = a0.w; r.v[4] = a1.x; r.v[5] = a1.y; r.v[6] = a1.z; r.v[7] = a1.w;
        return r;
    };
    auto st = [&](uint32_t L, const Fr& x) {
        tile[L] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        tile[tsize + L] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    };
    uint32_t q = 0;
    for (; q + 1 < nst; q += 2) {
        const uint32_t s = s0 + q;
        for (uint32_t gidx = threadIdx.x; gidx < (tsize >> 2); gidx += blockDim.x) {
            const uint32_t cl = gidx & (cols - 1), r = gidx >> cols_log;
            const uint32_t hj = r & ((1u << q) - 1u), hg = r >> q;
            const uint32_t h00 = (hg << (q + 2)) | hj;
            const uint32_t L0 = (h00 << cols_log) | cl, d = cols << q;
            const uint32_t lo = (col0 + cl) & lomask;
            const uint32_t j1 = (hj << s0) | lo;
            const uint32_t j2a = j1, j2b = ((hj | (1u << q)) << s0) | lo;
            Fr w1, w2a, w2b;
            if (DIAG == 1) { w1 = fr_load(tw + 1); w2a = fr_load(tw + 2); w2b = fr_load(tw + 3); }
            else {
                w1 = fr_load(tw + ((size_t)j1 << (logm - s - 1)));
                w2a = fr_load(tw + ((size_t)j2a << (logm - s - 2)));
                w2b = fr_load(tw + ((size_t)j2b << (logm - s - 2)));
            }
            Fr x0 = ld(L0), x1 = ld(L0 + d), x2 = ld(L0 + 2 * d), x3 = ld(L0 + 3 * d);
            Fr t = fe_mul(x1, w1);
            Fr y0 = fe_add(x0, t), y1 = fe_sub(x0, t);
            t = fe_mul(x3, w1);
            Fr y2 = fe_add(x2, t), y3 = fe_sub(x2, t);
            t = fe_mul(y2, w2a);
            st(L0, fe_add(y0, t));
            st(L0 + 2 * d, fe_sub(y0, t));
            t = fe_mul(y3, w2b);
            st(L0 + d, fe_add(y1, t));
            st(L0 + 3 * d, fe_sub(y1, t));
        }
        __syncthreads();
    }
    for (; q < nst; ++q) {
        const uint32_t s = s0 + q;
        for (uint32_t b = threadIdx.x; b < (tsize >> 1); b += blockDim.x) {
            uint32_t cl = b & (cols - 1), hb = b >> cols_log;
            uint32_t hj = hb & ((1u << q) - 1u), hg = hb >> q;
            uint32_t L0 = (((hg << (q + 1)) | hj) << cols_log) | cl;
            uint32_t L1 = L0 + (cols << q);
            uint32_t col = col0 + cl;
            uint32_t jglob = (hj << s0) | (col & lomask);
            Fr w = fr_load(tw + (DIAG == 1 ? 1 : ((size_t)jglob << (logm - s - 1))));
            Fr u = ld(L0), v = fe_mul(ld(L1), w);
            st(L0, fe_add(u, v));
            st(L1, fe_sub(u, v));
        }
        __syncthreads();
    }
    for (uint32_t L = threadIdx.x; L < tsize; L += blockDim.x) {
        uint32_t h = L >> cols_log, col = col0 + (L & (cols - 1));
        uint32_t idx = ((col >> s0) << (s0 + nst)) | (h << s0) | (col & lomask);
        if (DIAG == 2) { if (tile[L].x == 0x12345u) data[threadIdx.x].v[0] = 1; continue; }
        uint4* q4 = reinterpret_cast<uint4*>(data + idx);
        q4[0] = tile[L];
        q4[1] = tile[tsize + L];
    }
}

#ifdef VARIANT_HEADER
#include VARIANT_HEADER
#endif

template <class F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const uint32_t logm = 17, m = 1u << logm;
    const uint32_t np = argc > 1 ? atoi(argv[1]) : 8;
    const int nbuf = 6;  // rotate over the six work buffers of a sub-batch like the quotient does
    std::vector<Fr> h((size_t)np * m);
    uint64_t x = 88172645463325252ull;
    for (auto& e : h) for (int k = 0; k < 8; ++k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; e.v[k] = (uint32_t)x & (k == 7 ? 0x3fffffffu : 0xffffffffu); }
    Fr* d[nbuf]; Fr *tw, *ref, *out;
    for (int b = 0; b < nbuf; ++b) { CHECK(hipMalloc(&d[b], sizeof(Fr) * np * m)); CHECK(hipMemcpy(d[b], h.data(), sizeof(Fr) * np * m, hipMemcpyHostToDevice)); }
    CHECK(hipMalloc(&tw, sizeof(Fr) * m / 2)); CHECK(hipMemcpy(tw, h.data(), sizeof(Fr) * m / 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ref, sizeof(Fr) * np * m)); CHECK(hipMalloc(&out, sizeof(Fr) * np * m));
    const dim3 grid(m >> NTT_LT, np);
    const double mul1 = (double)np * (m / 2) * 10, mul2 = (double)np * (m / 2) * 7;
    int k = 0;
    auto report = [&](const char* name, float ms, double muls) { printf("%-44s %8.2f us   %6.1f G butterflies/s\n", name, ms * 1e3, muls / ms * 1e-6); };
    report("k_ntt_pass s0=0 nst=10", time_ms([&] { hipLaunchKernelGGL(k_ntt_pass, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 0u, 10u); }, 60), mul1);
    report("k_ntt_pass s0=10 nst=7", time_ms([&] { hipLaunchKernelGGL(k_ntt_pass, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 10u, 7u); }, 60), mul2);
    report("  twiddles from one address, pass 1", time_ms([&] { hipLaunchKernelGGL(k_pass_diag<1>, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 0u, 10u); }, 60), mul1);
    report("  twiddles from one address, pass 2", time_ms([&] { hipLaunchKernelGGL(k_pass_diag<1>, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 10u, 7u); }, 60), mul2);
    report("  no global data traffic, pass 1", time_ms([&] { hipLaunchKernelGGL(k_pass_diag<2>, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 0u, 10u); }, 60), mul1);
    report("  no global data traffic, pass 2", time_ms([&] { hipLaunchKernelGGL(k_pass_diag<2>, grid, dim3(256), 0, 0, d[k++ % nbuf], tw, logm, 10u, 7u); }, 60), mul2);
    report("k_ntt_copy_bitrev", time_ms([&] { hipLaunchKernelGGL(k_ntt_copy_bitrev, dim3(m / 256, np), dim3(256), 0, 0, d[k % nbuf], (size_t)m, m, d[(k + 1) % nbuf], logm); ++k; }, 60), 0);
    report("k_ntt_scale_bitrev", time_ms([&] { hipLaunchKernelGGL(k_ntt_scale_bitrev, dim3(m / 256, np), dim3(256), 0, 0, d[k % nbuf], d[(k + 2) % nbuf], d[(k + 1) % nbuf], logm); ++k; }, 60), 0);
#ifdef VARIANT_HEADER
    variant_main(d, nbuf, tw, ref, out, logm, np, h);
#endif
    return 0;
}
