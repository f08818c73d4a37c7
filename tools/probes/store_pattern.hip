// Probe: per-CU global store throughput of a GEMM epilogue as a function of the store shape.
// 256 workgroups x 512 threads; each writes `tiles` 256-row x 512-byte tiles of a row-major matrix with row stride ld bytes
// (8448 = ViT-g qkv output in bf16), 16 bytes per lane per instruction; SEG = contiguous bytes per row per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int SEG>
__global__ __launch_bounds__(512) void k(char* out, long ld, int tiles, int tiles_per_row, int spread) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;          // lanes per row segment
    constexpr int RPI = 64 / LPR;          // rows per instruction
    constexpr int IPR = 512 / SEG;         // instructions to cover a 512-byte tile row
    const v4i val = {lane, wave, 3, 4};
    for (int t = 0; t < tiles; ++t) {
        const long tile = (long)blockIdx.x * spread + (long)t * 256 * spread;   // distinct tiles per block / iteration
        const long tm = tile / tiles_per_row, tn = tile % tiles_per_row;
        char* base = out + tm * 256 * ld + tn * 512;
        // the wave owns rows [wave*32, +32) here (SEG >= 512 shapes) or a 128-byte column strip (SEG <= 128), like the epilogue
        if (SEG <= 128) {
            // wave (wm = wave>>2, wn = wave&3): 128 rows x 128 bytes
            const int wm = wave >> 2, wn = wave & 3;
            for (int r0 = 0; r0 < 128; r0 += RPI)
#pragma unroll
                for (int i = 0; i < 128 / SEG; ++i) {
                    const int row = wm * 128 + r0 + lane / LPR;
                    *(v4i*)(base + row * ld + wn * 128 + i * SEG + (lane % LPR) * 16) = val;
                }
        } else {
            for (int r0 = 0; r0 < 32; r0 += RPI)
#pragma unroll
                for (int i = 0; i < IPR; ++i) {
                    const int row = wave * 32 + r0 + lane / LPR;
                    *(v4i*)(base + row * ld + i * SEG + (lane % LPR) * 16) = val;
                }
        }
    }
}

template <int SEG> void run(char* buf, long ld, int tiles_per_row) {
    const int tiles = 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SEG><<<256, 512>>>(buf, ld, 2, tiles_per_row, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<SEG><<<256, 512>>>(buf, ld, tiles, tiles_per_row, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * tiles * 256 * 512;
    printf("segment %4d B per row per instruction: %6.2f us per 128 KiB tile, %6.1f GB/s per CU, %5.2f TB/s chip (%s)\n", SEG,
           ms * 1e3 / tiles, bytes / ms / 1e6 / 256, bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const long ld = 8448;
    const int tiles_per_row = 16;                 // 16 x 512 B = 8192 <= ld
    const long rows = (256L * 20 / 16 + 2) * 256;
    char* buf;
    (void)hipMalloc(&buf, rows * ld);
    run<64>(buf, ld, tiles_per_row);
    run<128>(buf, ld, tiles_per_row);
    run<512>(buf, ld, tiles_per_row);
    run<64>(buf, ld, tiles_per_row);
    return 0;
}
