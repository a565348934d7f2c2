// Issue cost of the VALU instructions around a Montgomery product's carry handling on gfx950, in SIMD cycles per wave64
// instruction (nominal 2.4 GHz), at 8 waves per SIMD: four independent chains per lane, 16 instructions per asm statement.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_ubench.hip -o tools/_build/valu_rate_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define R4(S) S S S S
#define BODY(I0, I1, I2, I3) R4(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t")

template <int OP>
__global__ void __launch_bounds__(256) k_op(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x + 1, c = seed ^ 0x9e3779b9u, d = a * 7 + 1;
    uint64_t x0 = a, x1 = b, x2 = c, x3 = d;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (OP == 0) asm volatile(BODY("v_add_u32 %0, %0, %1", "v_add_u32 %1, %1, %2", "v_add_u32 %2, %2, %3", "v_add_u32 %3, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 1) asm volatile(BODY("v_add_co_u32 %0, vcc, %0, %1", "v_add_co_u32 %1, vcc, %1, %2", "v_add_co_u32 %2, vcc, %2, %3", "v_add_co_u32 %3, vcc, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");
            if (OP == 2) asm volatile(BODY("v_addc_co_u32 %0, vcc, 0, %0, vcc", "v_addc_co_u32 %1, vcc, 0, %1, vcc", "v_addc_co_u32 %2, vcc, 0, %2, vcc", "v_addc_co_u32 %3, vcc, 0, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");
            if (OP == 3) asm volatile(BODY("v_addc_co_u32_e64 %0, s[20:21], 0, %0, s[22:23]", "v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[22:23]", "v_addc_co_u32_e64 %2, s[20:21], 0, %2, s[22:23]", "v_addc_co_u32_e64 %3, s[20:21], 0, %3, s[22:23]") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21");
            if (OP == 4) asm volatile(BODY("v_cndmask_b32_e64 %0, %1, %0, s[22:23]", "v_cndmask_b32_e64 %1, %2, %1, s[22:23]", "v_cndmask_b32_e64 %2, %3, %2, s[22:23]", "v_cndmask_b32_e64 %3, %0, %3, s[22:23]") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 5) asm volatile(BODY("v_cndmask_b32_e32 %0, %1, %0, vcc", "v_cndmask_b32_e32 %1, %2, %1, vcc", "v_cndmask_b32_e32 %2, %3, %2, vcc", "v_cndmask_b32_e32 %3, %0, %3, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 6) asm volatile(BODY("v_add3_u32 %0, %0, %1, %2", "v_add3_u32 %1, %1, %2, %3", "v_add3_u32 %2, %2, %3, %0", "v_add3_u32 %3, %3, %0, %1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 7) asm volatile(BODY("v_lshl_add_u64 %0, %0, 0, %1", "v_lshl_add_u64 %1, %1, 0, %2", "v_lshl_add_u64 %2, %2, 0, %3", "v_lshl_add_u64 %3, %3, 0, %0") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            if (OP == 8) asm volatile(BODY("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 9) asm volatile(BODY("v_mad_u64_u32 %0, vcc, %4, %5, %0", "v_mad_u64_u32 %1, vcc, %5, %6, %1", "v_mad_u64_u32 %2, vcc, %6, %7, %2", "v_mad_u64_u32 %3, vcc, %7, %4, %3") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            if (OP == 10) asm volatile(BODY("v_mad_u64_u32 %0, vcc, %4, s20, %0", "v_mad_u64_u32 %1, vcc, %5, s21, %1", "v_mad_u64_u32 %2, vcc, %6, s22, %2", "v_mad_u64_u32 %3, vcc, %7, s23, %3") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            if (OP == 11) asm volatile(BODY("v_subb_co_u32 %0, vcc, %0, %1, vcc", "v_subb_co_u32 %1, vcc, %1, %2, vcc", "v_subb_co_u32 %2, vcc, %2, %3, vcc", "v_subb_co_u32 %3, vcc, %3, %0, vcc") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");
            if (OP == 12) asm volatile(BODY("v_mul_u32_u24 %0, %0, %1", "v_mul_u32_u24 %1, %1, %2", "v_mul_u32_u24 %2, %2, %3", "v_mul_u32_u24 %3, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 13) asm volatile(BODY("v_mad_u32_u24 %0, %0, %1, %2", "v_mad_u32_u24 %1, %1, %2, %3", "v_mad_u32_u24 %2, %2, %3, %0", "v_mad_u32_u24 %3, %3, %0, %1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 14) asm volatile(BODY("v_mul_hi_u32_u24 %0, %0, %1", "v_mul_hi_u32_u24 %1, %1, %2", "v_mul_hi_u32_u24 %2, %2, %3", "v_mul_hi_u32_u24 %3, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 15) asm volatile(BODY("v_alignbit_b32 %0, %0, %1, 7", "v_alignbit_b32 %1, %1, %2, 7", "v_alignbit_b32 %2, %2, %3, 7", "v_alignbit_b32 %3, %3, %0, 7") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 16) asm volatile(BODY("v_and_b32 %0, %0, %1", "v_and_b32 %1, %1, %2", "v_and_b32 %2, %2, %3", "v_and_b32 %3, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 17) asm volatile(BODY("v_pk_add_u16 %0, %0, %1", "v_pk_add_u16 %1, %1, %2", "v_pk_add_u16 %2, %2, %3", "v_pk_add_u16 %3, %3, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 18) asm volatile(BODY("v_mad_u64_u32 %0, s[20:21], %4, %5, %0", "v_mad_u64_u32 %1, s[22:23], %5, %6, %1", "v_mad_u64_u32 %2, s[24:25], %6, %7, %2", "v_mad_u64_u32 %3, s[26:27], %7, %4, %3") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == 19) asm volatile(BODY("v_mad_i64_i32 %0, vcc, %4, %5, %0", "v_mad_i64_i32 %1, vcc, %5, %6, %1", "v_mad_i64_i32 %2, vcc, %6, %7, %2", "v_mad_i64_i32 %3, vcc, %7, %4, %3") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(x0 ^ x1 ^ x2 ^ x3) ^ (uint32_t)((x0 ^ x1 ^ x2 ^ x3) >> 32);
}

template <int OP>
static void run(const char* name, uint32_t* d) {
    const int blocks = 2048, threads = 256, iters = 500;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * threads / 64 * iters * 64.0;  // wave-level instructions
    printf("%-40s %8.3f ms  %8.1f G lane-ops/s  %5.2f SIMD-cycles@2.4GHz per wave instruction\n", name, ms, winstr * 64 / ms * 1e-6, ms * 1e-3 * 2.4e9 / (winstr / 1024));
}

int main() {
    uint32_t* d;
    CHECK(hipMalloc(&d, 64u << 20));
    run<0>("v_add_u32", d);
    run<1>("v_add_co_u32 (vcc out)", d);
    run<2>("v_addc_co_u32 e32 (vcc in, vcc out)", d);
    run<3>("v_addc_co_u32 e64 (sgpr in, sgpr out)", d);
    run<4>("v_cndmask_b32 e64 (sgpr mask)", d);
    run<5>("v_cndmask_b32 e32 (vcc)", d);
    run<6>("v_add3_u32", d);
    run<7>("v_lshl_add_u64", d);
    run<8>("v_mov_b32", d);
    run<9>("v_mad_u64_u32 (vgpr x vgpr)", d);
    run<10>("v_mad_u64_u32 (vgpr x sgpr)", d);
    run<18>("v_mad_u64_u32 (carry to sgpr pairs)", d);
    run<19>("v_mad_i64_i32", d);
    run<11>("v_subb_co_u32", d);
    run<12>("v_mul_u32_u24", d);
    run<13>("v_mad_u32_u24", d);
    run<14>("v_mul_hi_u32_u24", d);
    run<15>("v_alignbit_b32", d);
    run<16>("v_and_b32", d);
    run<17>("v_pk_add_u16", d);
    return 0;
}
