// Which XCD does workgroup (x, y) of a 2-D grid run on?  (The tree's level-0 passes over table regions rely on: linear id x + y * gridDim.x,
// round-robin over the 8 XCDs — so with gridDim.x a multiple of 8, XCD = x mod 8 whatever y.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t* out) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = id & 0xf;
}
int main() {
    for (auto g : {dim3(16, 6), dim3(24, 5), dim3(8, 86), dim3(1040, 86)}) {
        uint32_t* d; hipMalloc(&d, 4 * g.x * g.y);
        hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
        std::vector<uint32_t> h(g.x * g.y);
        hipMemcpy(h.data(), d, 4 * h.size(), hipMemcpyDeviceToHost);
        size_t match = 0;
        for (uint32_t y = 0; y < g.y; ++y) for (uint32_t x = 0; x < g.x; ++x) match += h[y * g.x + x] == x % 8;
        printf("grid (%u, %u): %zu of %zu workgroups on XCD x mod 8;  row 0:", g.x, g.y, match, h.size());
        for (uint32_t x = 0; x < 16 && x < g.x; ++x) printf(" %u", h[x]);
        printf("  row 1:");
        for (uint32_t x = 0; x < 16 && x < g.x; ++x) printf(" %u", h[g.x + x]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
