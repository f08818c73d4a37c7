// Probe: operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands on gfx950.
// Hypothesis: lane l holds row (l % 16) of A (resp. column of B), k = 32 * (l / 16) + j for byte j of its 8 operand registers, and the
// scale operand's selected byte is the E8M0 scale of exactly that 32-element block.  Prints the max error against a host reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* C, int lay, int smode) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    for (int j = 0; j < 8; ++j) {
        // lay 0: k = 32 g + 4 j .. ; lay 1: registers 0-3 hold k = 16 g + .., registers 4-7 hold k = 64 + 16 g + ..
        const int kk = lay == 0 ? g * 32 + j * 4 : (j < 4 ? g * 16 + j * 4 : 64 + g * 16 + (j - 4) * 4);
        a[j] = *(const int*)(A + r * 128 + kk);
        b[j] = *(const int*)(B + r * 128 + kk);
    }
    // smode 0: the lane's scale byte = block g of its row; 1: block index by the k range its FIRST registers hold (lay 1: 16 g / 32)
    int sa, sb;
    if (smode == 0) { sa = SA[r * 4 + g]; sb = SB[r * 4 + g]; }
    else if (smode == 1) { sa = SA[r * 4 + (g >> 1)] | (SA[r * 4 + 2 + (g >> 1)] << 8); sb = SB[r * 4 + (g >> 1)] | (SB[r * 4 + 2 + (g >> 1)] << 8); }
    else { sa = SA[r * 4] | (SA[r * 4 + 1] << 8) | (SA[r * 4 + 2] << 16) | (SA[r * 4 + 3] << 24); sb = SB[r * 4] | (SB[r * 4 + 1] << 8) | (SB[r * 4 + 2] << 16) | (SB[r * 4 + 3] << 24); }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + i
    for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) x = NAN;
    return s ? -x : x;
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128], hSA[64], hSB[64];
    uint8_t *A, *B, *SA, *SB; float* C;
    hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&SA, 64); hipMalloc(&SB, 64); hipMalloc(&C, 1024);
    for (int unit = 1; unit >= 0; --unit)
        for (int lay = 0; lay < 2; ++lay)
            for (int smode = 0; smode < 3; ++smode) {
                srand(1);
                for (int i = 0; i < 2048; ++i) {
                    do { hA[i] = rand() & 0xFF; } while ((hA[i] & 0x7F) == 0x7F);
                    do { hB[i] = rand() & 0xFF; } while ((hB[i] & 0x7F) == 0x7F);
                }
                for (int i = 0; i < 64; ++i) { hSA[i] = unit ? 127 : 120 + rand() % 14; hSB[i] = unit ? 127 : 120 + rand() % 14; }
                hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
                hipMemcpy(SA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(SB, hSB, 64, hipMemcpyHostToDevice);
                k<<<1, 64>>>(A, B, SA, SB, C, lay, smode);
                float hC[256];
                hipMemcpy(hC, C, 1024, hipMemcpyDeviceToHost);
                double worst = 0, ref_max = 0;
                for (int m = 0; m < 16; ++m)
                    for (int n = 0; n < 16; ++n) {
                        double acc = 0;
                        for (int kk = 0; kk < 128; ++kk)
                            acc += (double)e4m3(hA[m * 128 + kk]) * ldexp(1.0, hSA[m * 4 + kk / 32] - 127) * (double)e4m3(hB[n * 128 + kk]) * ldexp(1.0, hSB[n * 4 + kk / 32] - 127);
                        worst = fmax(worst, fabs(acc - hC[m * 16 + n]));
                        ref_max = fmax(ref_max, fabs(acc));
                    }
                printf("unit-scales %d  data layout %d  scale mode %d: max |err| %.3e (max |ref| %.3e) %s\n", unit, lay, smode, worst, ref_max,
                       worst < 1e-4 * ref_max ? "<== MATCH" : "");
            }
    return 0;
}
