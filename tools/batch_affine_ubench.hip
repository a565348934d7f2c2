// Go / no-go measurement for shared-inversion ("batch-affine") bucket accumulation on gfx950.
//
// A bucket's run of gathered window-table rows is summed as a pairwise tree; every tree level is two passes over the data
// with ONE grid-wide Montgomery batch inversion in between:
//   pass 1 (K1)  d = x2 - x1 of every pair, running product along a lane's pairs, prefix stored (48 B per pair)
//   (K2)         the lanes' products -> their inverses (3 products per lane + a handful of inversions: negligible, not timed here:
//                the second pass is fed a stand-in value, which costs the same arithmetic)
//   pass 2 (K3)  back-substitution (2 products), lambda = dy / d (1), x3 = lambda^2 - x1 - x2 (1 square), y3 (1): 5M + 1S per
//                addition in total against 8M + 2S for the XYZZ mixed addition of k_msm_accumulate.
// The price is memory traffic: level 0 gathers every table row twice (the x coordinates in pass 1, the whole row in pass 2),
// deeper levels stream the previous level's affine points.  This tool times, on one MSM-shaped problem (nb buckets, mean run
// length `mean`, np proofs, a table far larger than the Infinity Cache):
//   T0  raw gather bandwidth of the access patterns involved
//   T1  the product kernel k_msm_accumulate<G1> (ns per addition)
//   T2  level 0 of the tree (K1 + K3, lanes walking the sorted digit list)
//   T3  a deeper level (K1 + K3 over dense linear arrays)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/batch_affine_ubench.hip -o tools/_build/batch_affine_ubench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../masp_amd/csrc/device/msm_acc.hpp"
using namespace masp;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef TabRow<FpOps> Row;

// ---- data generation -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_fill_rows(Row* tab, Fp* xtab, uint32_t xstride_words, uint32_t nrows) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    Fp x, y;
    for (int k = 0; k < 12; ++k) { x.v[k] = mix(i * 24u + k); y.v[k] = mix(i * 24u + 12 + k); }
    x.v[11] &= 0x0fffffffu; y.v[11] &= 0x0fffffffu;  // < p
    tab[i].p.x = x; tab[i].p.y = y;
    *reinterpret_cast<Fp*>(reinterpret_cast<uint32_t*>(xtab) + (size_t)i * xstride_words) = x;
}
__global__ void k_fill_entries(uint32_t* sorted, size_t ent_stride, uint32_t total, uint32_t nrows) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    uint32_t h = mix(k * 2654435761u + blockIdx.y * 0x9e3779b9u + 17u);
    sorted[blockIdx.y * ent_stride + k] = (h % nrows) | ((h >> 7) << 31);
}

// ---- T0: gather bandwidth --------------------------------------------------------------------------------------
template <int STRIDE, int NLOAD>   // NLOAD x 16 bytes from the start of a row
__global__ void __launch_bounds__(256) k_gather_bw(const uint8_t* __restrict__ tab, uint32_t nrows, int iters, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint4* row = reinterpret_cast<const uint4*>(tab + (size_t)(mix(x) % nrows) * STRIDE);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            uint4 v = row[k];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[t] = acc;
}

// ---- level geometry ------------------------------------------------------------------------------------------
// Level 0 is the sorted digit list itself.  Its pairs are numbered in a padded virtual layout: bucket b owns pair slots
// [f0(b), f0(b+1)), f0(b) = (start[b] >> 1) + b, of which the first ceil(len / 2) are real.  Level-1 points of bucket b live
// at [off1(b), ...), off1(b) = 2 * ((start[b] >> 2) + b)  (even: pairs of level 1 never straddle buckets).
__device__ __forceinline__ uint32_t f0(const uint32_t* start, uint32_t b) { return (start[b] >> 1) + b; }
__device__ __forceinline__ uint32_t off1(const uint32_t* start, uint32_t b) { return 2u * ((start[b] >> 2) + b); }

// K1, level 0: lane `ln` walks pair slots [ln K, ln K + K) forwards; the prefix products go to pre[c * nlanes + ln] (c = count
// of real pairs so far: transposed, so a wave's stores are contiguous); the lane's product and pair count to tp / cnt.
template <bool XTAB>
__global__ void __launch_bounds__(64)
k_aff0_pass1(const Row* __restrict__ tab, const uint32_t* __restrict__ xtab, uint32_t xstride_words, const uint32_t* __restrict__ sorted, size_t ent_stride,
             const uint32_t* __restrict__ start, uint32_t nb, uint32_t nlanes, uint32_t K, Fp* __restrict__ pre, Fp* __restrict__ tp,
             uint32_t* __restrict__ cnt) {
    const uint32_t ln = blockIdx.x * blockDim.x + threadIdx.x;
    if (ln >= nlanes) return;
    sorted += blockIdx.y * ent_stride;
    const size_t lane_g = (size_t)blockIdx.y * nlanes + ln, nl_all = (size_t)gridDim.y * nlanes;
    const uint32_t V = f0(start, nb);
    const uint32_t g0 = ln * K;
    uint32_t c = 0;
    Fp chain = FpOps::one();
    if (g0 < V) {
        const uint32_t g1 = g0 + K < V ? g0 + K : V;
        uint32_t b = 0, span = nb;
        while (span > 1) {
            uint32_t half = span >> 1;
            if (f0(start, b + half) <= g0) b += half;
            span -= half;
        }
        uint32_t fb = f0(start, b), fnext = f0(start, b + 1), sb = start[b], se = start[b + 1];
        for (uint32_t g = g0; g < g1; ++g) {
            while (g >= fnext) {
                ++b;
                fb = fnext;
                fnext = f0(start, b + 1);
                sb = se;
                se = start[b + 1];
            }
            const uint32_t e0 = sb + 2 * (g - fb);
            if (e0 + 1 >= se) continue;  // empty slot or a lone last entry (passes through in pass 2)
            const uint32_t r1 = sorted[e0] & 0x7fffffffu, r2 = sorted[e0 + 1] & 0x7fffffffu;
            Fp x1, x2;
            if (XTAB) {
                x1 = *reinterpret_cast<const Fp*>(xtab + (size_t)r1 * xstride_words);
                x2 = *reinterpret_cast<const Fp*>(xtab + (size_t)r2 * xstride_words);
            } else {
                x1 = tab[r1].p.x;
                x2 = tab[r2].p.x;
            }
            Fp d = FpOps::sub(x2, x1);
            if (FpOps::is_zero(d)) d = FpOps::one();  // (exceptional pair: resolved in pass 2; placeholder here)
            chain = FpOps::mul(chain, d);
            pre[(size_t)c * nl_all + lane_g] = chain;
            ++c;
        }
    }
    tp[lane_g] = chain;
    cnt[lane_g] = c;
}
// K3, level 0: the same lanes walk BACKWARDS (back-substitution runs from the last pair to the first).
__global__ void __launch_bounds__(64)
k_aff0_pass2(const Row* __restrict__ tab, const uint32_t* __restrict__ sorted, size_t ent_stride, const uint32_t* __restrict__ start, uint32_t nb,
             uint32_t nlanes, uint32_t K, const Fp* __restrict__ pre, const Fp* __restrict__ tinv, const uint32_t* __restrict__ cnt,
             Fp* __restrict__ ox, Fp* __restrict__ oy, size_t out_stride) {
    const uint32_t ln = blockIdx.x * blockDim.x + threadIdx.x;
    if (ln >= nlanes) return;
    sorted += blockIdx.y * ent_stride;
    ox += blockIdx.y * out_stride;
    oy += blockIdx.y * out_stride;
    const size_t lane_g = (size_t)blockIdx.y * nlanes + ln, nl_all = (size_t)gridDim.y * nlanes;
    const uint32_t V = f0(start, nb);
    const uint32_t g0 = ln * K;
    if (g0 >= V) return;
    const uint32_t g1 = g0 + K < V ? g0 + K : V;
    uint32_t c = cnt[lane_g];
    Fp I = tinv[lane_g];
    uint32_t b = 0, span = nb;
    while (span > 1) {
        uint32_t half = span >> 1;
        if (f0(start, b + half) <= g1 - 1) b += half;
        span -= half;
    }
    uint32_t fb = f0(start, b), sb = start[b], se = start[b + 1], o1 = off1(start, b);
    for (uint32_t g = g1; g-- > g0;) {
        while (g < fb) {
            --b;
            se = sb;
            sb = start[b];
            fb = f0(start, b);
            o1 = off1(start, b);
        }
        const uint32_t j = g - fb, e0 = sb + 2 * j;
        if (e0 >= se) continue;  // empty slot
        const uint32_t w1 = sorted[e0];
        const Affine<FpOps> p1 = tab[w1 & 0x7fffffffu].p;
        const Fp y1 = (w1 >> 31) ? FpOps::neg(p1.y) : p1.y;
        if (e0 + 1 >= se) {  // lone last entry of its bucket: passes through
            ox[o1 + j] = p1.x;
            oy[o1 + j] = y1;
            continue;
        }
        const uint32_t w2 = sorted[e0 + 1];
        const Affine<FpOps> p2 = tab[w2 & 0x7fffffffu].p;
        const Fp y2 = (w2 >> 31) ? FpOps::neg(p2.y) : p2.y;
        --c;
        Fp d = FpOps::sub(p2.x, p1.x);
        Fp inv;
        if (c) {
            const Fp pp = pre[(size_t)(c - 1) * nl_all + lane_g];
            inv = FpOps::mul(I, pp);
        } else {
            inv = I;
        }
        if (FpOps::is_zero(d)) {  // (exceptional pair: placeholder output)
            ox[o1 + j] = FpOps::zero();
            oy[o1 + j] = FpOps::zero();
            continue;
        }
        I = FpOps::mul(I, d);
        const Fp lam = FpOps::mul(FpOps::sub(y2, y1), inv);
        const Fp x3 = FpOps::sub(FpOps::sub(FpOps::sqr(lam), p1.x), p2.x);
        const Fp y3 = FpOps::sub(FpOps::mul(lam, FpOps::sub(p1.x, x3)), y1);
        ox[o1 + j] = x3;
        oy[o1 + j] = y3;
    }
}

// Deeper level, dense variant: pair g = (points 2g, 2g + 1) of linear SoA arrays, KP pairs per lane at a stride of the whole
// grid (lane t: pairs t, t + NT, t + 2 NT, ...): every load and store of a wave is contiguous.
__global__ void __launch_bounds__(256)
k_affL_pass1(const Fp* __restrict__ x, uint32_t npairs, uint32_t KP, Fp* __restrict__ pre, Fp* __restrict__ tp) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, NT = gridDim.x * blockDim.x;
    Fp chain = FpOps::one();
    for (uint32_t j = 0; j < KP; ++j) {
        const uint32_t g = j * NT + t;
        if (g >= npairs) break;
        const Fp x1 = x[2 * (size_t)g], x2 = x[2 * (size_t)g + 1];
        Fp d = FpOps::sub(x2, x1);
        if (FpOps::is_zero(d)) d = FpOps::one();
        chain = FpOps::mul(chain, d);
        pre[(size_t)j * NT + t] = chain;
    }
    tp[t] = chain;
}
__global__ void __launch_bounds__(256)
k_affL_pass2(const Fp* __restrict__ x, const Fp* __restrict__ y, uint32_t npairs, uint32_t KP, const Fp* __restrict__ pre, const Fp* __restrict__ tinv,
             Fp* __restrict__ ox, Fp* __restrict__ oy) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, NT = gridDim.x * blockDim.x;
    if (t >= npairs) return;
    uint32_t jmax = (npairs - 1 - t) / NT;  // last j with j * NT + t < npairs
    if (jmax >= KP) jmax = KP - 1;
    Fp I = tinv[t];
    for (uint32_t j = jmax + 1; j-- > 0;) {
        const size_t g = (size_t)j * NT + t;
        const Fp x1 = x[2 * g], x2 = x[2 * g + 1], y1 = y[2 * g], y2 = y[2 * g + 1];
        const Fp d = FpOps::sub(x2, x1);
        Fp inv;
        if (j) {
            const Fp pp = pre[(size_t)(j - 1) * NT + t];
            inv = FpOps::mul(I, pp);
        } else {
            inv = I;
        }
        if (FpOps::is_zero(d)) {
            ox[g] = FpOps::zero();
            oy[g] = FpOps::zero();
            continue;
        }
        I = FpOps::mul(I, d);
        const Fp lam = FpOps::mul(FpOps::sub(y2, y1), inv);
        const Fp x3 = FpOps::sub(FpOps::sub(FpOps::sqr(lam), x1), x2);
        const Fp y3 = FpOps::sub(FpOps::mul(lam, FpOps::sub(x1, x3)), y1);
        ox[g] = x3;
        oy[g] = y3;
    }
}

template <class F> static float time_ms(F f, int reps = 3) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();  // warm
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CHECK(hipGetLastError());
    return best;
}

int main(int argc, char** argv) {
    const uint32_t c = argc > 1 ? atoi(argv[1]) : 16;          // window bits -> nb = 2^(c-1) buckets
    const uint32_t mean = argc > 2 ? atoi(argv[2]) : 80;       // mean run length of a bucket
    const uint32_t np = argc > 3 ? atoi(argv[3]) : 64;         // proofs per launch
    const uint32_t nrows = argc > 4 ? atoi(argv[4]) : 3705088; // table rows (h + l of Spend: 231 568 x 16)
    const uint32_t lanes = argc > 5 ? atoi(argv[5]) : 4096;    // lanes per proof of the walking kernels
    const uint32_t nb = 1u << (c - 1);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; nb %u, mean run %u, np %u, table %u rows (%.0f MB), %u lanes per proof\n", prop.name, prop.multiProcessorCount, nb, mean, np,
           nrows, nrows * 128.0 / 1e6, lanes);
    // bucket run lengths: mean +- sqrt(mean) (what a uniform scalar's digits give), bucket 0 heavy like the unit scalars' bucket
    std::mt19937 rng(7);
    std::vector<uint32_t> start(nb + 1);
    std::poisson_distribution<uint32_t> pois(mean);
    uint32_t total = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        start[b] = total;
        total += b == 0 ? 20000u : pois(rng);
    }
    start[nb] = total;
    const uint32_t V = (total >> 1) + nb, S1 = 2 * ((total >> 2) + nb) + 2;
    uint64_t adds0 = 0;  // real pairs of level 0 per proof
    for (uint32_t b = 0; b < nb; ++b) adds0 += (start[b + 1] - start[b]) / 2;
    printf("entries per proof %u, level-0 pair slots %u (real pairs %llu), level-1 slots %u\n", total, V, (unsigned long long)adds0, S1);

    Row* tab; Fp* xtab48; Fp* xtab64; uint32_t *d_sorted, *d_start, *d_cnt, *d_out;
    CHECK(hipMalloc(&tab, (size_t)nrows * sizeof(Row)));
    CHECK(hipMalloc(&xtab48, (size_t)nrows * 48));
    CHECK(hipMalloc(&xtab64, (size_t)nrows * 64));
    CHECK(hipMalloc(&d_sorted, (size_t)np * total * 4));
    CHECK(hipMalloc(&d_start, (nb + 1) * 4));
    CHECK(hipMemcpy(d_start, start.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill_rows, dim3((nrows + 255) / 256), dim3(256), 0, 0, tab, xtab48, 12u, nrows);
    hipLaunchKernelGGL(k_fill_rows, dim3((nrows + 255) / 256), dim3(256), 0, 0, tab, xtab64, 16u, nrows);
    hipLaunchKernelGGL(k_fill_entries, dim3((total + 255) / 256, np), dim3(256), 0, 0, d_sorted, (size_t)total, total, nrows);
    CHECK(hipDeviceSynchronize());

    // ---- T0 gather bandwidth
    {
        const uint32_t nl = 1u << 21; const int iters = 32;
        CHECK(hipMalloc(&d_out, nl * 4));
        const double rows = (double)nl * iters;
        float a = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<128, 6>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)tab, nrows, iters, d_out); });
        float b = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<128, 3>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)tab, nrows, iters, d_out); });
        float c48 = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<48, 3>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)xtab48, nrows, iters, d_out); });
        float c64 = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<64, 3>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)xtab64, nrows, iters, d_out); });
        printf("T0 gather: 96 B of a 128-B row: %.2f G rows/s (%.2f TB/s of lines) | x only (48 B) of a 128-B row: %.2f G rows/s | 48-B packed x table "
               "(%.0f MB): %.2f G rows/s | 64-B-stride x table (%.0f MB): %.2f G rows/s\n",
               rows / a / 1e6, rows * 128 / a / 1e9, rows / b / 1e6, nrows * 48.0 / 1e6, rows / c48 / 1e6, nrows * 64.0 / 1e6, rows / c64 / 1e6);
        // a table far beyond the Infinity Cache, for reference
        const uint32_t big_rows = 24u << 20;  // 3 GiB of 128-byte rows
        uint8_t* big; CHECK(hipMalloc(&big, (size_t)big_rows * 128)); CHECK(hipMemset(big, 1, (size_t)big_rows * 128));
        float d = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<128, 6>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)big, big_rows, iters, d_out); });
        float e = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<128, 3>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)big, big_rows, iters, d_out); });
        float f = time_ms([&] { hipLaunchKernelGGL((k_gather_bw<64, 3>), dim3(nl / 256), dim3(256), 0, 0, (const uint8_t*)big, big_rows * 2, iters, d_out); });
        printf("T0 gather, 3 GiB table: 96 B of a 128-B row %.2f G rows/s (%.2f TB/s of lines) | 48 B of a 128-B row %.2f G rows/s | 48 B at a 64-B stride %.2f G rows/s\n",
               rows / d / 1e6, rows * 128 / d / 1e9, rows / e / 1e6, rows / f / 1e6);
        CHECK(hipFree(big));
    }

    // ---- T1 the product kernel
    const uint64_t adds_total = (uint64_t)np * total;
    {
        const uint32_t nchunks = std::max(3072u, std::min(8192u, (3u << 18) / np));
        G1Xyzz* part; CHECK(hipMalloc(&part, (size_t)np * (nchunks + nb) * sizeof(G1Xyzz)));
        uint32_t* d_start_np;  // the product kernel takes one start[] per proof
        CHECK(hipMalloc(&d_start_np, (size_t)np * (nb + 1) * 4));
        for (uint32_t p = 0; p < np; ++p) CHECK(hipMemcpy(d_start_np + (size_t)p * (nb + 1), start.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
        float ms = time_ms([&] {
            hipLaunchKernelGGL((k_msm_accumulate<FpOps>), dim3((nchunks + 63) / 64, np), dim3(64), 0, 0, tab, d_sorted, (size_t)total, d_start_np, nb, nchunks, part);
        });
        CHECK(hipFree(d_start_np));
        printf("T1 k_msm_accumulate<G1>: %u chunks per proof, %.3f ms, %.4f ns per addition (%.1f G products/s at 9.6 per addition)\n", nchunks, ms,
               ms * 1e6 / adds_total, adds_total * 9.6 / ms / 1e6);
        CHECK(hipFree(part));
    }

    // ---- T2 level 0 of the tree
    const uint32_t K = (V + lanes - 1) / lanes;
    const size_t nl_all = (size_t)np * lanes;
    Fp *pre, *tp, *ox, *oy;
    CHECK(hipMalloc(&pre, (size_t)K * nl_all * sizeof(Fp)));
    CHECK(hipMalloc(&tp, nl_all * sizeof(Fp)));
    CHECK(hipMalloc(&d_cnt, nl_all * 4));
    CHECK(hipMalloc(&ox, (size_t)np * S1 * sizeof(Fp)));
    CHECK(hipMalloc(&oy, (size_t)np * S1 * sizeof(Fp)));
    CHECK(hipMemset(ox, 0, (size_t)np * S1 * sizeof(Fp)));
    CHECK(hipMemset(oy, 0, (size_t)np * S1 * sizeof(Fp)));
    const uint64_t pairs0 = adds0 * np;
    {
        const dim3 grid((lanes + 63) / 64, np), block(64);
        float a_full = time_ms([&] {
            hipLaunchKernelGGL((k_aff0_pass1<false>), grid, block, 0, 0, tab, (const uint32_t*)xtab48, 12u, d_sorted, (size_t)total, d_start, nb, lanes, K, pre, tp, d_cnt);
        });
        float a_x48 = time_ms([&] {
            hipLaunchKernelGGL((k_aff0_pass1<true>), grid, block, 0, 0, tab, (const uint32_t*)xtab48, 12u, d_sorted, (size_t)total, d_start, nb, lanes, K, pre, tp, d_cnt);
        });
        float a_x64 = time_ms([&] {
            hipLaunchKernelGGL((k_aff0_pass1<true>), grid, block, 0, 0, tab, (const uint32_t*)xtab64, 16u, d_sorted, (size_t)total, d_start, nb, lanes, K, pre, tp, d_cnt);
        });
        float b = time_ms([&] {
            hipLaunchKernelGGL(k_aff0_pass2, grid, block, 0, 0, tab, d_sorted, (size_t)total, d_start, nb, lanes, K, pre, tp, d_cnt, ox, oy, (size_t)S1);
        });
        const float a = std::min(a_full, std::min(a_x48, a_x64));
        printf("T2 level 0 (%llu additions, K = %u pair slots per lane): pass 1 %.3f ms with x from the table rows, %.3f ms from a packed 48-B x table, "
               "%.3f ms from a 64-B-stride x table; pass 2 %.3f ms  ->  %.4f ns per addition (best pass 1 + pass 2)\n",
               (unsigned long long)pairs0, K, a_full, a_x48, a_x64, b, (a + b) * 1e6 / pairs0);
    }
    // ---- T3 a deeper level on dense arrays: the level-1 arrays of T2 taken as they lie (all slots as pairs)
    {
        const uint32_t npairs = (uint32_t)std::min<uint64_t>((uint64_t)np * S1 / 2, 0xfffffff0u);
        for (uint32_t KP : {4u, 8u, 16u}) {
            const uint32_t nt = (npairs + KP - 1) / KP, blocks = (nt + 255) / 256;
            Fp *pre2, *tp2, *x2, *y2;
            CHECK(hipMalloc(&pre2, (size_t)KP * blocks * 256 * sizeof(Fp)));
            CHECK(hipMalloc(&tp2, (size_t)blocks * 256 * sizeof(Fp)));
            CHECK(hipMalloc(&x2, (size_t)npairs * sizeof(Fp)));
            CHECK(hipMalloc(&y2, (size_t)npairs * sizeof(Fp)));
            float a = time_ms([&] { hipLaunchKernelGGL(k_affL_pass1, dim3(blocks), dim3(256), 0, 0, ox, npairs, KP, pre2, tp2); });
            float b = time_ms([&] { hipLaunchKernelGGL(k_affL_pass2, dim3(blocks), dim3(256), 0, 0, ox, oy, npairs, KP, pre2, tp2, x2, y2); });
            printf("T3 deeper level, dense, %u pairs, %u pairs per lane: pass 1 %.3f ms, pass 2 %.3f ms  ->  %.4f ns per addition\n", npairs, KP, a, b,
                   (a + b) * 1e6 / npairs);
            CHECK(hipFree(pre2)); CHECK(hipFree(tp2)); CHECK(hipFree(x2)); CHECK(hipFree(y2));
        }
    }
    printf("model: a tree does half of its additions at level 0 and half at deeper levels; XYZZ reference = T1\n");
    return 0;
}
