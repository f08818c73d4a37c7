// Probe: does moving the LDS-DMA issue to dedicated producer waves take it off the consumers' critical path?
// Workgroup = 8 consumer waves (64 MFMA + 24 ds_read_b128 per 256x256x64 tile each, as dma_mfma_overlap.hip) + NP producer waves
// that issue the tile's 64 KiB of DMA (64 instructions of 1 KiB, split among the producers) two tiles ahead.
// Compare with the same work when every consumer wave issues its own 8 DMA instructions (NP = 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#define LDS_AS __attribute__((address_space(3)))
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NP>
__global__ __launch_bounds__(512 + NP * 64) void k(const char* base, int panels, int row_bytes, int tiles, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pa = (blockIdx.x * 7) % panels;
    const int64_t panel_bytes = (int64_t)512 * row_bytes;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(base + pa * panel_bytes), 0, (int)panel_bytes, 0x00020000);
    const int r_in = lane >> 3, c_in = (lane & 7) * 16;
    const int slabs = row_bytes / 128;
    if (NP > 0 && wave >= 8) {   // ---- producer ----
        const int pw = wave - 8;
        constexpr int PER = 64 / (NP > 0 ? NP : 1);   // DMA instructions per producer per tile
        for (int t = 0; t < tiles; ++t) {
            const int s = t % slabs;
            LDS_AS char* st = lds + (t & 1) * 65536;
#pragma unroll
            for (int c = 0; c < PER; ++c) {
                const int chunk = pw * PER + c;       // 1 KiB chunk index within the tile (64 of them)
                const int row = chunk * 8 + r_in;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)(st + chunk * 1024), 16, row * row_bytes + s * 128 + c_in, 0, 0, 0);
            }
            wait_vmcnt<(PER > 32 ? 32 : PER)>();       // (most of) the previous tile landed
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = (s16x8){(short)lane, 1, 2, 3, 4, 5, 6, 7}; fb[i] = (s16x8){1, (short)i, 2, 3, 4, 5, 6, 7}; }
    for (int t = 0; t < tiles; ++t) {
        const int s = t % slabs;
        LDS_AS char* st = lds + (t & 1) * 65536;
        if (NP == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int row = c * 64 + wave * 8 + r_in;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)(st + c * 8192 + wave * 1024), 16, row * row_bytes + s * 128 + c_in, 0, 0, 0);
            }
        }
        LDS_AS const char* rd = lds + ((t + 1) & 1) * 65536 + lane * 16;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            s16x8 v = *(LDS_AS const s16x8*)(rd + ((i * 8 + wave) & 63) * 1024);
            if (i < 4) fa[i] = v; else if (i < 8) fb[i - 4] = v; else asm volatile("" ::"v"(v));
        }
#pragma unroll
        for (int i = 0; i < 64; ++i)
            acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 3]), __builtin_bit_cast(bf16x8, fb[(i >> 2) & 3]), acc[i & 15], 0, 0, 0);
        if (NP == 0) wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sink) sink[blockIdx.x * 512 + (threadIdx.x & 511)] = r;
}

template <int NP> void run(const char* buf, int panels, float* sink) {
    const int tiles = 2000, row_bytes = 2816;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NP><<<256, 512 + NP * 64>>>(buf, panels, row_bytes, 50, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NP><<<256, 512 + NP * 64>>>(buf, panels, row_bytes, tiles, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("producers %d  panels %4d : %7.3f us per tile   (%s)\n", NP, panels, ms * 1e3 / tiles, hipGetErrorString(hipGetLastError()));
}

int main() {
    char* buf; float* sink;
    const int maxp = 64;
    (void)hipMalloc(&buf, (size_t)maxp * 512 * 2816);
    (void)hipMemset(buf, 1, (size_t)maxp * 512 * 2816);
    (void)hipMalloc(&sink, 256 * 512 * 4);
    for (int panels : {2, 32}) {
        run<0>(buf, panels, sink);
        run<1>(buf, panels, sink);
        run<2>(buf, panels, sink);
        run<4>(buf, panels, sink);
    }
    return 0;
}
