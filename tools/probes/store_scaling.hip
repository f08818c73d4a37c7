// Probe (round 5): is the ~6.7 us a GEMM epilogue takes to store a 128 KiB tile a per-CU limit or a chip-wide one?
// N workgroups of 512 threads (one per CU) each store `tiles` 128 KiB tiles (256 rows x 512 B, 64-byte row segments per wave instruction like the
// fast16 epilogue) with `gap` us of idle time between tiles (an s_sleep loop standing in for the K loop), and time each tile's stores from the first
// instruction to `s_waitcnt vmcnt(0)` with s_memrealtime (100 MHz).  If the time per tile does not depend on N it is the CU's own store path.
//   hipcc --offload-arch=gfx950 -O3 -o bin/store_scaling store_scaling.hip && bin/store_scaling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(char* out, long ld, int tiles, int tiles_per_row, int gap_ticks, unsigned long long* times) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const v4i val = {lane, wave, 3, 4};
    const int wm = wave >> 2, wn = wave & 3;
    unsigned long long acc = 0;
    for (int t = 0; t < tiles; ++t) {
        const long tile = (long)blockIdx.x + (long)t * gridDim.x;
        const long tm = tile / tiles_per_row, tn = tile % tiles_per_row;
        char* base = out + tm * 256 * ld + tn * 512;
        const unsigned long long g0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - g0 < (unsigned long long)gap_ticks) __builtin_amdgcn_s_sleep(16);
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int r0 = 0; r0 < 128; r0 += 16)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 128 + r0 + lane / 4;
                *(v4i*)(base + row * ld + wn * 128 + i * 64 + (lane % 4) * 16) = val;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += __builtin_amdgcn_s_memrealtime() - t0;
    }
    if (threadIdx.x == 0) times[blockIdx.x] = acc;
}

int main() {
    const long ld = 8448;
    const int tiles_per_row = 16, tiles = 12;
    const long rows = (256L * tiles / 16 + 2) * 256;
    char* buf;
    unsigned long long* d;
    (void)hipMalloc(&buf, rows * ld);
    (void)hipMalloc(&d, 256 * 8);
    for (int gap_us : {0, 40})
        for (int n : {8, 32, 64, 128, 256}) {
            for (int rep = 0; rep < 2; ++rep) {
                k<<<n, 512>>>(buf, ld, tiles, tiles_per_row, gap_us * 100, d);
                (void)hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(n);
            (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0;
            for (auto v : h) { s += (double)v; mx = std::max(mx, (double)v); }
            printf("gap %2d us, %3d workgroups: %6.2f us per 128 KiB tile (mean), %6.2f (slowest workgroup)   -> %5.1f B/clk per CU at 1.9 GHz\n", gap_us, n,
                   s / n / tiles / 100.0, mx / tiles / 100.0, 131072.0 / (s / n / tiles / 100.0 * 1900.0));
        }
    return 0;
}
