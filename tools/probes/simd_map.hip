// Probe: which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits [5:4] = SIMD id on gfx9-family)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d);
    unsigned h[1024 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, total = 0;
    for (int b = 0; b < 1024; ++b) {
        for (int w = 0; w < 4; ++w) { total++; if (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3)) same++; }
    }
    printf("waves w and w+4 on the same SIMD: %d of %d pairs\n", same, total);
    for (int b = 0; b < 4; ++b) { printf("block %d simd ids:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u\n", (h[b*8] >> 8) & 15); }
    return 0;
}
