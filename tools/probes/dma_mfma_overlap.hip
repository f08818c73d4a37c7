// Probe: do LDS-DMA fills (buffer_load ... lds) overlap with MFMA bursts and with ds_read traffic on one CU?
// Per tile a 512-thread workgroup issues 64 KiB of DMA (8 instr/wave, 128-byte row segments) into a 2-stage LDS ring, then each
// wave runs NM MFMAs (16x16x32 bf16, register operands) and NR ds_read_b128 - the 256x256x64 GEMM tile's budget is NM=64, NR=24.
// Modes: DMA only, MFMA only, reads only, DMA+MFMA, DMA+reads, all.   time(all) ~ max(...) => overlap; ~ sum => serialised.
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDS_AS __attribute__((address_space(3)))
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool DMA, int NM, int NR, int LOOK = 0>
__global__ __launch_bounds__(512) void k(const char* base, int panels, int row_bytes, int tiles, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pa = (blockIdx.x * 7) % panels;
    const int64_t panel_bytes = (int64_t)512 * row_bytes;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(base + pa * panel_bytes), 0, (int)panel_bytes, 0x00020000);
    const int r_in = lane >> 3, c_in = (lane & 7) * 16;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = (s16x8){(short)lane, 1, 2, 3, 4, 5, 6, 7}; fb[i] = (s16x8){1, (short)i, 2, 3, 4, 5, 6, 7}; }
    const int slabs = row_bytes / 128;
    for (int t = 0; t < tiles; ++t) {
        const int s = t % slabs;
        LDS_AS char* st = lds + (LOOK ? 0 : (t & 1) * 65536);
        if (DMA) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int row = c * 64 + wave * 8 + r_in;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)(st + c * 8192 + wave * 1024), 16, row * row_bytes + s * 128 + c_in, 0, 0, 0);
            }
        }
        if (NR) {
            LDS_AS const char* rd = lds + ((t + 1) & 1) * 65536 + lane * 16;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                s16x8 v = *(LDS_AS const s16x8*)(rd + ((i * 8 + wave) & 63) * 1024);
                if (i < 4) fa[i] = v; else if (i < 8) fb[i - 4] = v; else asm volatile("" ::"v"(v));
            }
        }
#pragma unroll
        for (int i = 0; i < NM; ++i)
            acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[(i >> 2) & 3]), acc[i & 15], 0, 0, 0);
        if (DMA) wait_vmcnt<LOOK * 8>();   // LOOK tiles of DMA may stay in flight across the barrier
        __builtin_amdgcn_s_barrier();
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink) sink[blockIdx.x * 512 + threadIdx.x] = r;
}

template <bool DMA, int NM, int NR, int LOOK = 0> void run(const char* name, const char* buf, int panels, float* sink) {
    const int tiles = 2000, row_bytes = 2816;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<DMA, NM, NR, LOOK><<<256, 512>>>(buf, panels, row_bytes, 50, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<DMA, NM, NR, LOOK><<<256, 512>>>(buf, panels, row_bytes, tiles, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s panels %4d : %7.3f us per tile   (%s)\n", name, panels, ms * 1e3 / tiles, hipGetErrorString(hipGetLastError()));
}

int main() {
    char* buf; float* sink;
    const int maxp = 1024;
    (void)hipMalloc(&buf, (size_t)maxp * 512 * 2816);
    (void)hipMemset(buf, 1, (size_t)maxp * 512 * 2816);
    (void)hipMalloc(&sink, 256 * 512 * 4);
    for (int panels : {2, 32}) {
        run<true, 0, 0>("DMA only", buf, panels, sink);
        run<false, 64, 0>("MFMA only (64)", buf, panels, sink);
        run<false, 0, 24>("reads only (24)", buf, panels, sink);
        run<true, 64, 0>("DMA + MFMA", buf, panels, sink);
        run<true, 0, 24>("DMA + reads", buf, panels, sink);
        run<false, 64, 24>("MFMA + reads", buf, panels, sink);
        run<true, 64, 24>("DMA + MFMA + reads", buf, panels, sink);
        run<true, 0, 0, 1>("DMA only, look 1", buf, panels, sink);
        run<true, 0, 0, 3>("DMA only, look 3", buf, panels, sink);
        run<true, 64, 0, 1>("DMA + MFMA, look 1", buf, panels, sink);
        run<true, 64, 0, 3>("DMA + MFMA, look 3", buf, panels, sink);
        run<true, 64, 24, 3>("DMA+MFMA+reads, look 3", buf, panels, sink);
    }
    return 0;
}
