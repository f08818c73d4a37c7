// Probe: per-CU global->LDS fill rate of the LDS-DMA path (buffer_load ... lds, 16 B/lane) as a function of
//   RB    - contiguous bytes per matrix row touched by one pass (64 = BK 32 bf16, 128 = BK 64, 256 = BK 128)
//   F     - DMA instructions in flight per wave (each 1 KiB)
//   panels- number of distinct 256-row panels the blocks share (small = L2 resident, medium = MALL, large = HBM)
// One 512-thread workgroup per CU streams two 256-row panels (the GEMM's A and B tiles), k-slab after k-slab, into a 128 KiB
// LDS ring.  No MFMA: this is the ceiling the GEMM main loop's fill can reach.
//   hipcc --offload-arch=gfx950 -O3 -o dma_fill dma_fill.hip && ./dma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LDS_AS __attribute__((address_space(3)))
typedef int v4i __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int RB, int F>
__global__ __launch_bounds__(512) void fill_kernel(const char* base, int panels, int row_bytes, int iters, int* sink) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pa = (blockIdx.x * 7) % panels, pb = (blockIdx.x * 13 + 1) % panels;
    const int64_t panel_bytes = (int64_t)256 * row_bytes;
    constexpr int LPR = RB / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;         // rows per wave-instruction
    constexpr int RPC = RPI * 8;          // rows per chunk (8 waves)
    constexpr int CPS = 512 / RPC;        // chunks per k-slab (512 rows: A panel then B panel)
    const int slabs = row_bytes / RB;
    const int r_in = lane / LPR, c_in = (lane % LPR) * 16;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(base + pa * panel_bytes), 0, (int)panel_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(base + pb * panel_bytes), 0, (int)panel_bytes, 0x00020000);
    int chunk = 0;
    for (int it = 0; it < iters; ++it)
        for (int s = 0; s < slabs; ++s) {
#pragma unroll
            for (int c = 0; c < CPS; ++c) {
                const int row = (c * RPC + wave * RPI + r_in) & 255;
                const int voff = row * row_bytes + s * RB + c_in;
                LDS_AS char* dst = lds + ((chunk & 15) * 8192 + wave * 1024);
                if (c * RPC < 256) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_AS void*)dst, 16, voff, 0, 0, 0);
                wait_vmcnt<F>();
                ++chunk;
            }
        }
    wait_vmcnt<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(LDS_AS int*)lds;
}

template <int RB, int F> void run(const char* buf, int panels, int row_bytes, int* sink) {
    const int iters = 40;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fill_kernel<RB, F><<<256, 512>>>(buf, panels, row_bytes, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fill_kernel<RB, F><<<256, 512>>>(buf, panels, row_bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * iters * 512.0 * row_bytes;
    printf("RB %3d  in-flight %2d KiB/wave  panels %5d : %7.1f GB/s per CU   %6.2f TB/s total   (%s)\n", RB, F + 1, panels,
           bytes / ms / 1e6 / 256, bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int row_bytes = 2816;   // K = 1408 bf16
    const int max_panels = 2048;
    char* buf; int* sink;
    hipMalloc(&buf, (size_t)max_panels * 256 * row_bytes);
    hipMemset(buf, 1, (size_t)max_panels * 256 * row_bytes);
    hipMalloc(&sink, 4096);
    for (int panels : {4, 64, 2048}) {
        run<64, 3>(buf, panels, row_bytes, sink);
        run<64, 7>(buf, panels, row_bytes, sink);
        run<64, 11>(buf, panels, row_bytes, sink);
        run<64, 15>(buf, panels, row_bytes, sink);
        run<128, 3>(buf, panels, row_bytes, sink);
        run<128, 7>(buf, panels, row_bytes, sink);
        run<128, 11>(buf, panels, row_bytes, sink);
        run<128, 15>(buf, panels, row_bytes, sink);
        run<256, 7>(buf, panels, row_bytes, sink);
        run<256, 15>(buf, panels, row_bytes, sink);
    }
    return 0;
}
