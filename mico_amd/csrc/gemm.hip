// MFMA GEMM for gfx950 with fused epilogues (see include/mico_hip.h: mico_gemm).
//
// Measured on MI355X these GEMMs are bound by the per-CU HBM/L2 -> LDS fill rate (~45 GB/s per CU, ~11 TB/s chip) long
// before the matrix pipe: 128x128 tiles (64 flop per staged byte) plateau near 750 TFLOP/s, 256x128 near 930.  Hence two
// configurations:
//   * BIG   256x256x32, 8 waves (2x4, 128x64 per wave), 4-stage LDS ring of 32 KiB K-tiles (128 KiB, one workgroup per CU):
//           128 flop per staged byte, prefetch distance 3 tiles.  Schedule = 2-phase PING-PONG: waves w and w+4 share a
//           SIMD (measured: wave->SIMD order 0,2,1,3,0,2,1,3) and form two groups; every K-tile (one 32-deep MFMA k-step) is
//           two barrier-separated phases and in each phase one group issues its 12 LDS fragment reads AND its 4 DMA
//           instructions for tile t+3, while the other group runs its 32 MFMAs on fragments fetched one phase earlier -
//               group0:  R(t)   | M(t)          group1:  M(t-1) | R(t)
//           so a SIMD's matrix pipe always has one wave feeding it, the partner's LDS latency and the (expensive: ~100-200
//           issue cycles each) DMA instructions hide behind it, fragments are single-buffered (48 VGPRs next to 128
//           accumulator VGPRs), and the DMA queue is never drained: counted `s_waitcnt vmcnt(8)` + raw `s_barrier`.
//   * SMALL 128x128x64, 4 waves (2x2), 2-stage ring, two workgroups per CU: short-M / narrow problems (BERT text, heads).
// Common: operand tiles go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B/lane, no VGPR round trip); the buffer
// descriptor's bounds check zero-fills rows past the end of the matrix, so M/N/K tails need no masking in the main loop.
// LDS images are lane-linear (DMA constraint); bank conflicts are removed by XOR-swizzling the 16-byte chunk index on the
// *source* address and applying the same involution on the read side (measured: SQ_LDS_BANK_CONFLICT = 0).  K-contiguous
// operands are read with ds_read_b128, reduction-major operands (dX / dW GEMMs) with the gfx950 transposing read
// ds_read_b64_tr_b16, so the backward GEMMs need no transposed copies of weights or activations in HBM.  MFMA operands are
// swapped (D^T = B A^T) so a lane owns 4 consecutive output columns; the epilogue then transposes through LDS so that global
// accesses are whole cache lines.  Workgroup ids are remapped XCD-contiguously (8 private L2s) and walk the tile grid in
// groups of row-panels so concurrently resident tiles share A and B panels in L2.  Long-reduction weight-gradient GEMMs are
// split along K in whole waves of resident workgroups and combined with fp32 atomics.
#include "common.h"
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>

#ifndef MICO_W4_DBG
#define MICO_W4_DBG 0
#endif
#ifndef MICO_W4_STAGGER
#define MICO_W4_STAGGER 1
#endif
#ifndef MICO_BIG_MIDBAR   // 8-wave kernel: the barrier between the two phases of a K-tile (aligns the two wave groups; no data hazard needs
                          // it).  1 = always, 0 = never, 2 = only with a k-contiguous B operand (forward orientation): measured in one run,
                          // forward 968 with / 909 without, dX (reduction-major W, transposing reads) 978 with / 1003 without
#define MICO_BIG_MIDBAR 2
#endif
#ifndef MICO_SLAB_FIXED   // split-K slab path: fixed cost of a wave of workgroups in 32-deep K-tiles (see mico_gemm)
#define MICO_SLAB_FIXED 48
#endif
#ifndef MICO_GEMM_ABLATE   // benchmark-only ablation builds (tools/): 1 = no steady-state DMA, 2 = no LDS reads, 3 = no MFMA,
                           // 4 = DMA issued but out of bounds (no memory traffic; zero operands), 5 = DMA re-reads two K-tiles,
                           // 6 = no epilogue
#define MICO_GEMM_ABLATE 0
#endif

#if MICO_GEMM_ABLATE == 7   // timing build: per-workgroup phase timestamps (s_memrealtime, 100 MHz) of the 8-wave kernel
__device__ unsigned long long g_mico_phase_times[8192 * 8];
extern "C" int mico_debug_phase_times(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mico_phase_times), sizeof(unsigned long long) * n);
}
#define PHASE_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_mico_phase_times[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PHASE_STAMP(slot) do {} while (0)
#endif

#if MICO_GEMM_ABLATE == 8   // timing build of the one-wave-per-SIMD kernel: cycles at the top-of-iteration wait + barrier vs the whole K loop
__device__ unsigned long long g_mico_w4_prof[4];
extern "C" int mico_debug_w4_prof(unsigned long long* out) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mico_w4_prof), sizeof(unsigned long long) * 4);
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mico_w4_prof), z, sizeof(z));
    return rc;
}
#endif

namespace {

#ifndef MICO_MMA_PRIO   // s_setprio(1) around the MFMA bursts of the two ping-pong GEMM kernels
#define MICO_MMA_PRIO 0
#endif
#ifndef MICO_GROUP_M
#define MICO_GROUP_M 4   // row-tiles per group of the tile order: 4 x 8 blocks per XCD round measured +1-2 % over 8 x 4 on the forward / dX GEMMs (in situ 891 -> 901, 937 -> 958), 16 and 3 worse
#endif
constexpr int GROUP_M = MICO_GROUP_M;
#ifndef MICO_MID_DB   // MID kernel: both k-steps' fragments of a K-tile read up front (48 more registers): layer forward 951 -> 985, dX 989 -> 1040 TFLOP/s
#define MICO_MID_DB 1
#endif
#ifndef MICO_MID_M32   // MID kernel, forward orientation: 32x32x16 MFMAs instead of 16x16x32 (VERDICT r2 lever (a))
#define MICO_MID_M32 0
#endif
#ifndef MICO_MID_PRIO   // MID kernel: 1 = static priority for alternate rounds of workgroups, 2 = s_setprio(1) around the MFMA bursts
#define MICO_MID_PRIO 0
#endif
#ifndef MICO_MID_IL   // MID kernel: the refill DMA dealt out between the second k-step's MFMAs
#define MICO_MID_IL 0
#endif
#ifndef MICO_MID_GROUP_M
#define MICO_MID_GROUP_M 4   // row-tiles per group of the MID kernel's tile order (64 tiles of 256x128 per XCD at a time): 4 -> 8 -> 16 = layer forward 1020 / 986 / 910 TFLOP/s
#endif

template <int BM_, int BN_, int WM_, int WN_, int BK_, int STAGES_> struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, STAGES = STAGES_;
    static constexpr int WAVES = WM * WN, THREADS = WAVES * 64;
    static constexpr int MT = BM / WM / 16, NT = BN / WN / 16;   // 16x16 MFMA tiles per wave
    static constexpr int KSTEPS = BK / 32;                       // MFMA k-steps per K-tile
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_DMA = A_BYTES / 16 / THREADS, B_DMA = B_BYTES / 16 / THREADS;   // DMA instructions per thread per tile
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static_assert(NT == 4, "wave tiles are 64 columns wide");
};
using Big = TileCfg<256, 256, 2, 4, 32, 4>;
using Small = TileCfg<128, 128, 2, 2, 64, 2>;
// MID 256x128x32, 4 waves (2x2, 128x64 per wave), 3-stage ring of 24 KiB K-tiles (72 KiB): TWO workgroups per CU.  The 8-wave kernel
// spends ~20 us of prologue + epilogue + dispatch gap per 256x256 tile with the matrix pipes idle (a CU stores at ~10 B/clk whatever
// the store shape) around ~26 us of K loop at K = 1408; with two independent workgroups per CU one's epilogue / prologue runs under
// the other's K loop, and 128-wide tiles fit N = 1408 / 4224 / 6144 exactly (no half-empty sixth tile column).
using Mid = TileCfg<256, 128, 2, 2, 32, 3>;

struct GemmArgs {
    const char* A;
    const char* B;
    char* C;
    int64_t M, N, K, lda, ldb, ldc;
    int ntm, ntn, ntiles, split_k, ktiles, ktiles_per_split;
    int c_dtype;
    int group_m;                // MID kernel: row-tiles per group of the tile order
    int fast16;                 // P8: the launch qualifies for p8_epilogue_fast16 (16-bit C, no row scale / map / residual / accumulate)
    int tm0;                    // first row tile of this launch (a problem split into a P8 launch over full rounds + a MID launch over the rest)
    int a_wrap;                 // P8: K-tiles after which the A stream starts over (x W_hi + x W_lo as ONE product over the weights' [hi | lo] rows); 0: never
    int64_t ka_rows, kb_rows;   // physical reduction extents of A / B (differ from K in k-segment mode)
    mico_gemm_epilogue e;
};

// swizzle keys (16-byte chunk index XOR) - see file header
__device__ __forceinline__ int key_kc(int row) { return (row >> 1) & 7; }                               // [rows][64] k-contiguous
__device__ __forceinline__ int key_k32(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }      // [rows][32]: {0,3,2,1}[(row>>2)&3]
__device__ __forceinline__ int key_tr(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }   // [64][cols] reduction-major

// One LDS-DMA piece: 16 bytes per lane from the buffer `rs` at per-lane byte offset voff into the lane-linear LDS image starting at lds_dst.
// MICO_ASM_DMA (default on): issued by inline assembly.  With the builtin, hipcc knows an LDS write is pending and puts `s_waitcnt vmcnt(0)` in
// front of the first __builtin_amdgcn_ds_read_tr16_b64 that follows (the transposing read carries no address information it could tell from
// the DMA destination; tools/probes/README.md) - in the dX / dW kernels that drained the three K-tiles in flight once per K-tile.  The asm form
// is invisible to the compiler's counters: completion is the kernels' own counted `s_waitcnt vmcnt(N)` + barrier (they never relied on the
// compiler for that, except in front of the __syncthreads() of the two-stage loops, which now carry an explicit vmcnt(0)).  M0 (the LDS
// destination register of the instruction) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
#ifndef MICO_ASM_DMA
#define MICO_ASM_DMA 1
#endif
// Where it is used (in situ A/B of the timed step, asm everywhere vs builtin everywhere): the 8-wave kernel's dX orientation 951 -> 996 TFLOP/s and
// BERT's small-tile weight gradients 364 -> 417, but the producer / consumer dW kernel 1004 -> 865 (its producer waves interleave the DMA with the
// bias column sums' own inline-asm LDS reads) and the forward orientation (no transposing reads) 975 -> 962: so ASM = the launch reads an
// operand with transposing reads AND is not the producer / consumer kernel.
template <bool ASM>
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, LDS_AS void* lds_dst, unsigned voff) {
    if constexpr (ASM && MICO_ASM_DMA) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs) : "memory");
    } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_dst, 16, voff, 0, 0, 0);
    }
}

// The same with a scalar byte offset in the instruction's SOFFSET operand (address = base + voff + soff; the descriptor's bounds check
// looks at voff alone - callers keep everything that can leave the buffer, i.e. the row part, in voff): no VALU add per DMA.
__device__ __forceinline__ void lds_dma16_soff(__amdgpu_buffer_rsrc_t rs, LDS_AS void* lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs), "s"(soff) : "memory");
}

// generic (masked) staging of one operand tile HBM -> LDS.  ROWS = tile extent along the non-reduction dim.
template <bool TR, int ROWS, int THREADS, int BK, bool ASM = false>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, LDS_AS char* lds_tile, int wave, int lane,
                                           int64_t ld_bytes, int k0, int64_t kdim, int64_t cdim_rem) {
    constexpr int NDMA = ROWS * BK * 2 / 16 / THREADS;
#pragma unroll
    for (int it = 0; it < NDMA; ++it) {
        const int c = it * THREADS + wave * 64 + lane;
        unsigned voff;
        if (!TR) {
            constexpr int CPRK = BK / 8;        // 16-byte chunks per k-contiguous row (8 or 4)
            const int row = c / CPRK, cpos = c % CPRK;
            const int cg = cpos ^ (BK == 64 ? key_kc(row) : key_k32(row));
            const int k = k0 + cg * 8;
            voff = (unsigned)(row * ld_bytes + (int64_t)k * 2);
            if (k >= kdim) voff = 0xFFFFFFF0u;   // K tail: force out-of-bounds -> zero fill
        } else {
            constexpr int CPR = ROWS / 8;       // 16-byte chunks per tile row
            const int row = c / CPR, cpos = c % CPR;
            const int cg = cpos ^ key_tr(row);
            voff = (unsigned)((int64_t)(k0 + row) * ld_bytes + cg * 16);
            if (cg * 8 >= cdim_rem || (k0 + row) >= kdim) voff = 0xFFFFFFF0u;
        }
        lds_dma16<ASM>(rs, (LDS_AS void*)(lds_tile + (it * THREADS + wave * 64) * 16), voff);
    }
}

// ---- register-lean fragment addressing ----------------------------------------------------------------------------------
// Per lane and operand only two LDS byte offsets (k-step 0 / 1) are kept; the 16-row tiles of a wave are reached by an
// immediate (+i*2048, k-contiguous image) or an XOR (^(i<<5), reduction-major image) and the second transposing read by a
// constant (+4 rows).  These identities follow from the swizzle keys: key_kc depends on (row>>1)&7 only, hence not on the
// 16-row tile index; key_tr is identical for rows r and r+4 inside an 8-row group and only touches chunk bits 1-3, which the
// tile index occupies exclusively because wave column bases are multiples of 64 (and of 128 when a wave owns 8 tiles).
struct FragBase { int b0, b1; };

template <bool TR, int ROWS, int BK>
__device__ __forceinline__ FragBase frag_base(int wbase, int lane) {
    FragBase f;
    const int g = lane >> 4, p = lane & 15;
    if (!TR && BK == 64) {
        const int keyl = (p >> 1) & 7;
        const int base = (wbase + p) * 128;
        f.b0 = base + ((g ^ keyl) << 4);
        f.b1 = base + (((4 + g) ^ keyl) << 4);
    } else if (!TR) {   // 64-byte rows, one k-step per tile
        f.b0 = (wbase + p) * 64 + ((g ^ key_k32(p)) << 4);
        f.b1 = f.b0;
    } else {
        constexpr int RB = ROWS * 2;
        const int key0 = ((p >> 2) | ((g & 1) << 2)) << 1;
        const int col = ((((wbase >> 3) + ((lane >> 1) & 1)) ^ key0) << 4) + (lane & 1) * 8;
        f.b0 = (g * 8 + (p >> 2)) * RB + col;
        f.b1 = (32 + g * 8 + (p >> 2)) * RB + col;
    }
    return f;
}

template <bool TR, int ROWS, int BK>
__device__ __forceinline__ s16x8 read_frag_b(LDS_AS const char* tile, int base, int i) {
    if (!TR) {
        return *(LDS_AS const s16x8*)(tile + base + i * (BK * 2 * 16));
    } else {
        constexpr int RB = ROWS * 2;
        LDS_AS const char* a = tile + (base ^ (i << 5));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)a);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(a + 4 * RB));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// The same, for kernels that keep LDS-DMA in flight ACROSS their fragment reads (the one-wave-per-SIMD kernel): the transposing read
// as inline assembly.  hipcc 7.2 puts `s_waitcnt vmcnt(0)` in front of the first __builtin_amdgcn_ds_read_tr16_b64 after every LDS-DMA
// instruction (it cannot tell the intrinsic's LDS address from the DMA's destination; plain ds_read_b128 loads carry alias information
// and are left alone) - with DMA interleaved between the reads that drains the queue six times per K-tile (dX 600, dW 360 TFLOP/s).
// The compiler does not count an asm read either: the CALLER must execute `s_waitcnt lgkmcnt(0)` + sched_barrier between these reads
// and the first use of the fragments (the kernel's top-of-iteration wait), and the loop's ISA must show no copy of the fragment
// registers in between (tools/isa_audit.sh).
template <bool TR, int ROWS, int BK>
__device__ __forceinline__ s16x8 read_frag_dma(LDS_AS const char* tile, int base, int i) {
    if constexpr (!TR) {
        return *(LDS_AS const s16x8*)(tile + base + i * (BK * 2 * 16));
    } else {
        constexpr int RB = ROWS * 2;
        const unsigned a = (unsigned)(uintptr_t)(tile + (base ^ (i << 5)));
        s16x4 lo, hi;
#if MICO_W4_DBG == 3
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + (base ^ (i << 5))));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + (base ^ (i << 5)) + 4 * RB));
#elif MICO_W4_DBG == 2
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(hi) : "v"(a), "n"(4 * RB));
#else
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * RB));
#endif
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

// loop-invariant per-lane DMA offsets of one operand tile (k0 = 0); 0xFFFFFFF0 marks a chunk past the matrix edge
template <bool TR, int ROWS, int THREADS, int BK, int NDMA>
__device__ __forceinline__ void dma_offsets(unsigned (&vo)[NDMA], int wave, int lane, int64_t ld_bytes, int64_t cdim_rem) {
#pragma unroll
    for (int it = 0; it < NDMA; ++it) {
        const int c = it * THREADS + wave * 64 + lane;
        if (!TR) {
            constexpr int CPRK = BK / 8;
            const int row = c / CPRK, cpos = c % CPRK;
            vo[it] = (unsigned)(row * ld_bytes + ((cpos ^ (BK == 64 ? key_kc(row) : key_k32(row))) << 4));
        } else {
            constexpr int CPR = ROWS / 8;
            const int row = c / CPR, cpos = c % CPR;
            const int cg = cpos ^ key_tr(row);
            vo[it] = (cg * 8 >= cdim_rem) ? 0xFFFFFFF0u : (unsigned)(row * ld_bytes + cg * 16);
        }
    }
}

template <int THREADS, int NDMA, int NISSUE = NDMA, bool ASM = false>
__device__ __forceinline__ void dma_issue(__amdgpu_buffer_rsrc_t rs, LDS_AS char* lds_tile, int wave, const unsigned (&vo)[NDMA],
                                          unsigned koff) {
#pragma unroll
    for (int it = 0; it < NISSUE; ++it) {
        unsigned v = (vo[it] == 0xFFFFFFF0u) ? 0xFFFFFFF0u : vo[it] + koff;
        if (MICO_GEMM_ABLATE == 4) v |= 0xFFFFFFF0u;   // ablation: every DMA out of bounds (issue + LDS zero-fill, no memory traffic)
        lds_dma16<ASM>(rs, (LDS_AS void*)(lds_tile + (it * THREADS + wave * 64) * 16), v);
    }
}

// ---- epilogues ------------------------------------------------------------------------------------------------------------
// Through LDS: the MFMA layout gives a lane 4 consecutive columns of 16 different rows (32-byte row segments per store
// instruction - measured as ~14 us of fixed cost per 256x128 tile, a write-bound tail at ~1.2 TB/s).  Each wave therefore
// parks a 64x64 fp32 block of its accumulators in LDS (16 KiB, XOR-swizzled 16-byte chunks, conflict free both ways) and
// re-reads it so that a lane owns 16 consecutive columns of one row: bias / residual / auxiliary loads and the stores are
// then 32-64 contiguous bytes per lane, 128-256 per row - whole cache lines.  acc[0..3] is the block for 64 rows at mrow0.
// LDS chunk swizzle of the epilogue staging image (16 chunks of 16 B per 64-column row): a 4-bit permutation found by search
// that makes the MFMA-layout writes (8-lane groups) and both row-major read-back ownerships below conflict free.
__device__ __forceinline__ int epi_key(int row) { return (int)((0xF615B0AC843297DEull >> ((row & 15) * 4)) & 15); }

// ACT: 0 = the activation is a run-time field (every launch but the two below); MICO_ACT_GELU_SAVE_DERIV / MICO_ACT_MUL_AUX = the MLP
// pair compiled into its own kernel instantiation - as two more run-time branches of the shared epilogue they pushed the 8-wave
// kernel from 6 to 51 spilled registers and slowed EVERY launch by 10-15 % (tools/probes/README.md).
constexpr int ACT_LEAN = 5;
// ACT_RESID: the towers' output-projection / fc2 forward epilogue as its own instantiation - fp32 out[map(m)] = resid[map(m)] + row_scale * (acc
// + bias) through a frame map - with the row bookkeeping of all passes hoisted in front of the pass loop (row_map -> row_scale -> resid were
// three dependent loads inside every pass) and pass p + 1's residual rows requested before pass p is finished.  Measured in round 2 inside
// the SHARED lean epilogue: +3-8 % on these launches, but 30-70 spilled registers on every other launch (tools/probes/README.md) - as a
// separate instantiation the lean kernels keep their 208 registers.
constexpr int ACT_RESID = 6;
// M32: the accumulators come from 32x32x16 MFMAs (operands swapped like the 16x16 kernels: lane l holds output row l & 31 and the four
// 4-column groups 8 rg + 4 (l >> 5) of a 32x32 tile): `acc` then points at the block's [2 row-tiles][2 column-tiles][4 groups] f32x4 values.
// SHIFT: the block's 64 columns are TWO strips of 32 - local columns 0-31 at ncol0, 32-63 at ncol0 + 32 + SHIFT (the 8-phase kernel's
// waves own columns [32 wn, +32) and [128 + 32 wn, +32) of a 256-wide tile: SHIFT = 96); 0 = one contiguous strip.  Every 4- / 8-column
// group a lane owns lies inside one strip.
template <typename T, int MB = 4, int ACT = 0, bool M32 = false, int SHIFT = 0>   // MB = 16-row MFMA tiles per block (4: 64 rows, 16 KiB of LDS; 2: 32 rows, 8 KiB)
__device__ __forceinline__ void gemm_epilogue_block(const GemmArgs& g, const f32x4 (*acc)[4], LDS_AS char* wbuf, int64_t mrow0,
                                                    int64_t ncol0, int lane) {
    // ACT == ACT_LEAN: launches that use none of {aux copy, activation, dropout, positional table, patch->token remap} - the qkv / fc2 /
    // projection forwards and every dX of the towers - get an instantiation with those features compiled out
    // (the MLP pair's instantiations carry their own activation code and none of the other optional features either)
    constexpr bool LEAN = ACT != 0;   // (ACT_RESID leaves through its own path below)
    // every argument field the epilogue needs, read ONCE into scalars: left as g.e.<field> references the compiler re-loaded
    // them from the kernel-argument segment inside every pass (66 s_load_dwordx8 + waits in the unrolled code)
    const struct {
        const float* bias; void* aux_out; const void* aux_in; int64_t ldaux; int act; const float* row_scale; int rows_per_scale;
        const float* resid; const float* pos; int pos_rows; int remap_group, remap_skip, remap_offset; float alpha; int accumulate;
        const int* row_map; int rows_per_map; float drop_p; unsigned drop_seed; int drop_site;
    } e = {g.e.bias, g.e.aux_out, g.e.aux_in, g.e.ldaux, g.e.act, g.e.row_scale, g.e.rows_per_scale, g.e.resid, g.e.pos, g.e.pos_rows,
           g.e.remap_group, g.e.remap_skip, g.e.remap_offset, g.e.alpha, g.e.accumulate, g.e.row_map, g.e.rows_per_map,
           g.e.drop_p, g.e.drop_seed, g.e.drop_site};
    const int64_t gM = g.M, gN = g.N, gldc = g.ldc;
    char* const gC = g.C;
    if constexpr (M32) {
        static_assert(MB == 4, "32x32 accumulator blocks are 64 rows");
        const int r = lane & 31, hh = lane >> 5, kp = epi_key(r);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int row = mi * 32 + r, cc = nj * 8 + 2 * rg + hh;
                    *(LDS_AS f32x4*)(wbuf + row * 256 + ((cc ^ kp) << 4)) = acc[mi * 2 + nj][rg] * e.alpha;
                }
    } else {
        const int p = lane & 15, gq = lane >> 4, kp = epi_key(p);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = i * 16 + p, cc = j * 4 + gq;
                *(LDS_AS f32x4*)(wbuf + row * 256 + ((cc ^ kp) << 4)) = acc[i][j] * e.alpha;
            }
    }
    // Read-back ownership: a lane owns four 4-column groups of one row, chosen so that ONE store instruction covers 64
    // contiguous bytes per row (4 lanes x 16 B) - with the former "16 consecutive columns per lane" ownership every store
    // instruction wrote 8- or 16-byte pieces at a 32- / 64-byte stride, i.e. each output line was written by four partial
    // stores, and the epilogue took 18 us per 256x256 tile (26 % of the forward GEMM, tools/probes/gemm_phases.py).
    //   fp32 output : group v = columns v*16 + q*4 .. +3          (store v: 16 B per lane)
    //   16-bit      : groups (2u, 2u+1) = columns u*32 + q*8 .. +7 (store u: 16 B per lane)
    const int q = lane & 3;
    const bool wide = g.c_dtype == MICO_F32;
    int col[4];    // column inside the block's LDS image
    int gcol[4];   // the same group's column relative to ncol0 in the output (== col[] unless the block is two strips)
    bool ok[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        col[v] = wide ? (v * 16 + q * 4) : ((v >> 1) * 32 + q * 8 + (v & 1) * 4);
        gcol[v] = col[v] + (col[v] >= 32 ? SHIFT : 0);
        ok[v] = ncol0 + gcol[v] < gN;    // N % 4 == 0: a 4-column group is either fully inside or fully outside
    }
    if (!ok[0]) return;   // group 0 is the lane's leftmost
    f32x4 bias4[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bias4[v] = (e.bias && ok[v]) ? *(const f32x4*)(e.bias + ncol0 + gcol[v]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (ACT == ACT_RESID) {
        // (launch contract, checked on the host: fp32 C, resid, no aux / activation / dropout / pos / remap / accumulate)
        int64_t mo_[MB];
        float rs_[MB];
        bool live_[MB];
#pragma unroll
        for (int pass = 0; pass < MB; ++pass) {
            const int64_t m = mrow0 + pass * 16 + (lane >> 2);
            live_[pass] = m < gM;
            int64_t mo = m;
            if (e.row_map && live_[pass]) { const int64_t f = m / e.rows_per_map; mo = (int64_t)e.row_map[f] * e.rows_per_map + (m - f * e.rows_per_map); }
            mo_[pass] = mo;
        }
#pragma unroll
        for (int pass = 0; pass < MB; ++pass) rs_[pass] = (e.row_scale && live_[pass]) ? e.row_scale[mo_[pass] / e.rows_per_scale] : 1.f;
        f32x4 rcur[4], rnxt[4];
        auto load_resid = [&](int pass, f32x4 (&r)[4]) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                r[v] = (live_[pass] && ok[v]) ? *(const f32x4*)(e.resid + mo_[pass] * gldc + ncol0 + gcol[v]) : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        load_resid(0, rcur);
#pragma unroll
        for (int pass = 0; pass < MB; ++pass) {
            if (pass + 1 < MB) load_resid(pass + 1, rnxt);
            const int row = pass * 16 + (lane >> 2);
            const int kr = epi_key(row);
            if (live_[pass]) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (!ok[v]) continue;
                    const f32x4 a4 = *(LDS_AS const f32x4*)(wbuf + row * 256 + ((((col[v] >> 2)) ^ kr) << 4)) + bias4[v];
                    *(f32x4*)((float*)gC + mo_[pass] * gldc + ncol0 + gcol[v]) = a4 * rs_[pass] + rcur[v];
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) rcur[v] = rnxt[v];
        }
        return;
    }
    // NOT unrolled: the pass body is ~1.5k instructions with every epilogue feature inlined; unrolled 4x (and twice per tile) the
    // epilogue was ~90 KiB of straight-line code executed once per tile - far beyond the 64 KiB instruction cache shared by two
    // CUs - and took 15 us per 256x256 tile against a 5.5 us store-bandwidth floor (tools/probes/gemm_phases.py, store_pattern.hip)
#pragma unroll 1
    for (int pass = 0; pass < MB; ++pass) {
        const int row = pass * 16 + (lane >> 2);
        const int64_t m = mrow0 + row;
        if (m >= gM) continue;
        int64_t mo = m;
        if (!LEAN && e.remap_group) mo = m + (m / e.remap_group) * e.remap_skip + e.remap_offset;
        else if (e.row_map) { const int64_t f = m / e.rows_per_map; mo = (int64_t)e.row_map[f] * e.rows_per_map + (m - f * e.rows_per_map); }
        float rscale = 1.f;
        if (e.row_scale) rscale = e.row_scale[(e.row_map ? mo : m) / e.rows_per_scale];
        const int kr = epi_key(row);
        f32x4 v4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) v4[v] = *(LDS_AS const f32x4*)(wbuf + row * 256 + ((((col[v] >> 2)) ^ kr) << 4)) + bias4[v];
        if ((ACT == 0 || ACT == MICO_ACT_GELU_SAVE_DERIV) && e.aux_out && !(ACT == 0 && e.act == MICO_ACT_GELU_GRAD)) {   // 16-bit copy of the pre-activation (ACT 3: of gelu'); present only with 16-bit outputs, i.e. the paired ownership
            T* ap = (T*)e.aux_out + m * e.ldaux + ncol0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 a0 = v4[2 * u], a1 = v4[2 * u + 1];
                if constexpr (ACT == MICO_ACT_GELU_SAVE_DERIV) {   // gelu and gelu' share the erf and the Gaussian; v becomes gelu(v) here
                    f32x4 d0, d1;
                    v4[2 * u] = gelu_pair4(a0, d0);
                    v4[2 * u + 1] = gelu_pair4(a1, d1);
                    a0 = d0;
                    a1 = d1;
                }
                const s16x4 lo = pack4<T>(a0[0], a0[1], a0[2], a0[3]);
                const s16x4 hi = pack4<T>(a1[0], a1[1], a1[2], a1[3]);
                if (!wide && ok[2 * u + 1]) *(s16x8*)(ap + gcol[2 * u]) = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                else {
                    if (ok[2 * u]) *(s16x4*)(ap + gcol[2 * u]) = lo;
                    if (ok[2 * u + 1]) *(s16x4*)(ap + gcol[2 * u + 1]) = hi;
                }
            }
        }
        if constexpr (ACT == MICO_ACT_MUL_AUX) {
            const T* hp = (const T*)e.aux_in + m * e.ldaux + ncol0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (ok[v]) v4[v] *= unpack4<T>(*(const s16x4*)(hp + gcol[v]));
        } else if constexpr (ACT == 0) {
            if (e.act == MICO_ACT_GELU) {
#pragma unroll
                for (int v = 0; v < 4; ++v) v4[v] = gelu4(v4[v]);
            } else if (e.act == MICO_ACT_GELU_GRAD) {
                const T* hp = (const T*)e.aux_in + m * e.ldaux + ncol0;
                if (e.aux_out) {      // ... and gelu(aux_in) goes to aux_out (the pre-activation-keeping MLP pair: the operand of fc2's weight gradient);
                    T* gp = (T*)e.aux_out + m * e.ldaux + ncol0;      // gelu_pair4: bit for bit the persistent kernel's ACT_GRAD_TILED epilogue
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (ok[v]) {
                            f32x4 d;
                            const f32x4 gl = gelu_pair4(unpack4<T>(*(const s16x4*)(hp + gcol[v])), d);
                            v4[v] *= d;
                            *(s16x4*)(gp + gcol[v]) = pack4<T>(gl[0], gl[1], gl[2], gl[3]);
                        }
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (ok[v]) {
                            const f32x4 h = unpack4<T>(*(const s16x4*)(hp + gcol[v]));
                            v4[v][0] *= gelu_grad_f(h[0]); v4[v][1] *= gelu_grad_f(h[1]); v4[v][2] *= gelu_grad_f(h[2]); v4[v][3] *= gelu_grad_f(h[3]);
                        }
                }
            }
        }
        if (!LEAN && e.drop_p > 0.f) {
            const unsigned thr = drop_threshold(e.drop_p);
            const float ik = 1.f / (1.f - e.drop_p);
            const unsigned long long i0 = (unsigned long long)m * (unsigned long long)gN + (unsigned long long)ncol0;
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int k = 0; k < 4; ++k) v4[v][k] *= drop_mult(e.drop_seed, e.drop_site, i0 + gcol[v] + k, thr, ik);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            v4[v] *= rscale;
            if (!ok[v]) continue;
            if (!LEAN && e.pos) v4[v] += *(const f32x4*)(e.pos + (mo % e.pos_rows) * gN + ncol0 + gcol[v]);
            if (e.resid) v4[v] += *(const f32x4*)(e.resid + mo * gldc + ncol0 + gcol[v]);
        }
        if (wide) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (!ok[v]) continue;
                float* cp = (float*)gC + mo * gldc + ncol0 + gcol[v];
                if (e.accumulate) *(f32x4*)cp += v4[v];
                else *(f32x4*)cp = v4[v];
            }
        } else {
            T* cp = (T*)gC + mo * gldc + ncol0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const s16x4 lo = pack4<T>(v4[2 * u][0], v4[2 * u][1], v4[2 * u][2], v4[2 * u][3]);
                const s16x4 hi = pack4<T>(v4[2 * u + 1][0], v4[2 * u + 1][1], v4[2 * u + 1][2], v4[2 * u + 1][3]);
                if (ok[2 * u + 1]) *(s16x8*)(cp + gcol[2 * u]) = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                else if (ok[2 * u]) *(s16x4*)(cp + gcol[2 * u]) = lo;
            }
        }
    }
}

// Direct (register-layout) epilogue for split-K partial tiles: alpha-scaled fp32 atomics straight from the MFMA layout
// (4 consecutive columns per lane); measured 2x faster for atomics than the LDS-transposed 16-column form.
template <int MB = 4>
__device__ __forceinline__ void gemm_epilogue_atomic(const GemmArgs& g, const f32x4 (*acc)[4], int64_t mrow0, int64_t ncol0,
                                                     int lane) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int64_t m = mrow0 + i * 16 + (lane & 15);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = ncol0 + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            const f32x4 v = acc[i][j] * g.e.alpha;
            float* cp = (float*)g.C + m * g.ldc + n;
            unsafeAtomicAdd(cp + 0, v[0]); unsafeAtomicAdd(cp + 1, v[1]);
            unsafeAtomicAdd(cp + 2, v[2]); unsafeAtomicAdd(cp + 3, v[3]);
        }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, bool TA, bool TB, typename CFG, int ACT = 0>
// (two waves per SIMD in every configuration: the 4-wave small-tile kernel otherwise takes 260 registers and ONE workgroup per CU)
__global__ __launch_bounds__(CFG::THREADS, 2) void gemm_kernel(const GemmArgs g) {
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, THREADS = CFG::THREADS, MT = CFG::MT;
    constexpr bool PINGPONG = CFG::WAVES == 8;
    __shared__ __attribute__((aligned(16))) char smem[CFG::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    PHASE_STAMP(0);
    // ---- workgroup -> (k-split, tile) : XCD-contiguous remap (bijective), then grouped row-panel order ----
    int bid = blockIdx.x;
    const int ks = bid / g.ntiles;
    bid -= ks * g.ntiles;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    // Half-width edge tiles (N = 1408 = 5.5 x 256: every sixth tile of the towers' N = 1408 GEMMs): the 8 waves regroup as 4 x 2 over the
    // valid 256 x 128 half, 64 x 64 outputs each - half the MFMAs and fragment reads per wave on all four SIMDs - instead of four waves
    // multiplying zero-filled columns.  Same LDS images and DMA (the out-of-bounds half of the B image costs no memory traffic).
    const bool halfn = PINGPONG && (g.N - n0) * 2 <= BN;
    const int wm = halfn ? (wave >> 1) : wave / CFG::WN, wn = halfn ? (wave & 1) : wave % CFG::WN;
    const int wrow = halfn ? wm * 64 : wm * (BM / CFG::WM), wcol = wn * 64;
    const int kt0 = ks * g.ktiles_per_split;
    const int kt1 = min(g.ktiles, kt0 + g.ktiles_per_split);
    const int T_ = kt1 - kt0;

    // ---- buffer descriptors (block-relative base so 32-bit offsets never overflow) ----
    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
    const char* a_base = TA ? g.A + m0 * 2 : g.A + m0 * lda_b;
    const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
    int64_t a_bytes = TA ? g.ka_rows * lda_b - m0 * 2 : (g.M - m0) * lda_b;
    int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
    if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
    if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
    const int64_t a_crem = g.M - m0, b_crem = g.N - n0;

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging: fast path = precomputed per-lane offsets + one scalar k offset; masked generic path for a ragged last
    // K-tile.  k-segments (split-precision GEMMs) map the logical k-tile to per-operand physical column offsets.
    const int nseg = g.e.nseg, kseg = g.e.kseg;
    const FragBase ab = frag_base<TA, BM, BK>(wrow, lane), bb = frag_base<TB, BN, BK>(wcol, lane);
    unsigned voa[CFG::A_DMA], vob[CFG::B_DMA];
    dma_offsets<TA, BM, THREADS, BK, CFG::A_DMA>(voa, wave, lane, lda_b, a_crem);
    dma_offsets<TB, BN, THREADS, BK, CFG::B_DMA>(vob, wave, lane, ldb_b, b_crem);
    const bool ktail = (g.K % BK) != 0;
    auto stage = [&](int kt, int bo) {
        if (MICO_GEMM_ABLATE == 5) kt = kt0 + (kt & 1);   // ablation: re-read the first two K-tiles (cache-resident operands)
        const int k0 = kt * BK;
        int ka = k0, kb = k0;
        int64_t kda = g.K, kdb = g.K;
        if (nseg > 0) {
            const int sg = k0 / kseg, kin = k0 - sg * kseg;
            ka = g.e.a_seg_off[sg] + kin;
            kb = g.e.b_seg_off[sg] + kin;
            kda = g.e.a_seg_off[sg] + kseg;
            kdb = g.e.b_seg_off[sg] + kseg;
        }
        if (MICO_GEMM_ABLATE == 1 && kt >= kt0 + 3) return;   // ablation: no DMA in the steady state
        constexpr bool ADMA = TA || TB;   // a transposing-read orientation: DMA by inline assembly (see lds_dma16)
        if (ktail && kt == g.ktiles - 1) {
            stage_tile<TA, BM, THREADS, BK, ADMA>(rsa, lds + bo, wave, lane, lda_b, ka, kda, a_crem);
            stage_tile<TB, BN, THREADS, BK, ADMA>(rsb, lds + bo + CFG::A_BYTES, wave, lane, ldb_b, kb, kdb, b_crem);
            return;
        }
        const unsigned koa = TA ? (unsigned)((int64_t)ka * lda_b) : (unsigned)(ka * 2);
        const unsigned kob = TB ? (unsigned)((int64_t)kb * ldb_b) : (unsigned)(kb * 2);
        dma_issue<THREADS, CFG::A_DMA, CFG::A_DMA, ADMA>(rsa, lds + bo, wave, voa, koa);
        dma_issue<THREADS, CFG::B_DMA, CFG::B_DMA, ADMA>(rsb, lds + bo + CFG::A_BYTES, wave, vob, kob);
    };

    s16x8 fa[MT], fb[4];   // fragments of ONE 32-deep k-step
    auto read_k = [&](int bo, int kk, auto mtv) {
        constexpr int MV = decltype(mtv)::value;   // 16-row tiles this wave owns (MT, or MT / 2 on a half-width edge tile)
        if (MICO_GEMM_ABLATE == 2 && g.K > 0) {   // ablation: no LDS reads (keep fragments opaque)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[j]));
            return;
        }
        LDS_AS const char* ta = lds + bo;
        LDS_AS const char* tb = ta + CFG::A_BYTES;
        const int abase = kk ? ab.b1 : ab.b0, bbase = kk ? bb.b1 : bb.b0;
#pragma unroll
        for (int i = 0; i < MV; ++i) fa[i] = read_frag_b<TA, BM, BK>(ta, abase, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag_b<TB, BN, BK>(tb, bbase, j);
    };
    auto mma_k = [&](auto mtv) {
        constexpr int MV = decltype(mtv)::value;
        if (MICO_GEMM_ABLATE == 3 && g.K > 0) {   // ablation: no MFMA (keep operands live)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fb[j]));
            return;
        }
        if (MICO_MMA_PRIO) __builtin_amdgcn_s_setprio(1);   // the wave entering its MFMA burst outranks the one issuing reads / DMA on the same SIMD
#pragma unroll
        for (int i = 0; i < MV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = T16<T>::mfma(fb[j], fa[i], acc[i][j]);
        if (MICO_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    using FullT = std::integral_constant<int, MT>;
    using HalfT = std::integral_constant<int, MT / 2>;

    if constexpr (!PINGPONG && CFG::STAGES == 3) {
        // one barrier per 32-deep K-tile; tiles t+1 (and, from its issue on, t+2) stay in flight across it: counted vmcnt + raw s_barrier.
        // Order inside an iteration: fragment reads of tile t, THEN the DMA of tile t+2 into the buffer of tile t-1 (every wave has passed
        // this iteration's barrier, i.e. retired its reads of t-1) - hipcc waits vmcnt(0) in front of the first transposing read after an
        // LDS-DMA (tools/probes/README.md), which with this order costs the dX orientation one tile of lookahead, not two.
        static_assert(CFG::KSTEPS == 1, "Mid path: one k-step per tile");
        constexpr int PT = CFG::A_DMA + CFG::B_DMA;
        for (int i = 0; i < 2 && i < T_; ++i) stage(kt0 + i, i * CFG::STAGE_BYTES);
        int bo = 0, bn = 2 * CFG::STAGE_BYTES;
        for (int t = 0; t < T_; ++t) {
            asm volatile("" : "+s"(bo), "+s"(bn));
            if (t + 1 < T_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_k(bo, 0, FullT{});
            if (t + 2 < T_) stage(kt0 + t + 2, bn);
            __builtin_amdgcn_sched_barrier(0);
            mma_k(FullT{});
            __builtin_amdgcn_sched_barrier(0);
            bo = bo == 2 * CFG::STAGE_BYTES ? 0 : bo + CFG::STAGE_BYTES;
            bn = bn == 2 * CFG::STAGE_BYTES ? 0 : bn + CFG::STAGE_BYTES;
        }
    } else if constexpr (!PINGPONG) {
        int bo = 0;
        if (T_ > 0) stage(kt0, 0);
        for (int t = 0; t < T_; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage `bo` landed (explicit: the DMA may be invisible to the compiler) ...
            __syncthreads();   // ... and is published; the other stage is no longer being read
            if (t + 1 < T_) stage(kt0 + t + 1, bo ^ CFG::STAGE_BYTES);
#pragma unroll
            for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                read_k(bo, kk, FullT{});
                mma_k(FullT{});
            }
            bo ^= CFG::STAGE_BYTES;
        }
    } else {
        // see the file header for the schedule.  The two groups run separate straight-line loops with the same number of
        // barriers per K-tile, so the fragment registers have one unambiguous live range in each.
        static_assert(CFG::KSTEPS == 1 && CFG::STAGES == 4, "ping-pong path: one k-step per tile, 4-stage ring");
        constexpr int PT = CFG::A_DMA + CFG::B_DMA;           // DMA instructions per thread per tile
        constexpr int RING = CFG::STAGES * CFG::STAGE_BYTES;   // power of two
        const int grp = wave >> 2;
        auto head = [&](int t) {
            // this wave's share of tile t has landed (tiles t+1, t+2 may stay in flight) and its own LDS reads of the buffer
            // that is about to be refilled have returned; the barrier then publishes tile t to every wave
            const int ahead = T_ - 1 - t;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PT) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if constexpr (MICO_BIG_MIDBAR == 1 || (MICO_BIG_MIDBAR == 2 && !TB)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int i = 0; i < 3 && i < T_; ++i) stage(kt0 + i, i * CFG::STAGE_BYTES);
        PHASE_STAMP(1);
        auto k_loop = [&](auto mtv) {
            int bo = 0;   // ring offset of tile t (kept opaque so LDS addresses are not hoisted per buffer)
            if (grp == 0) {
                for (int t = 0; t < T_; ++t) {
                    asm volatile("" : "+s"(bo));
                    head(t);
                    read_k(bo, 0, mtv);
                    if (t + 3 < T_) stage(kt0 + t + 3, (bo + 3 * CFG::STAGE_BYTES) & (RING - 1));   // buffer of tile t-1: free
                    bar();
                    mma_k(mtv);
                    __builtin_amdgcn_sched_barrier(0);
                    bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
                }
            } else {
                for (int t = 0; t < T_; ++t) {
                    asm volatile("" : "+s"(bo));
                    head(t);
                    if (t > 0) mma_k(mtv);      // tile t-1 (fragments read in the second phase of that tile)
                    bar();
                    read_k(bo, 0, mtv);
                    if (t + 3 < T_) stage(kt0 + t + 3, (bo + 3 * CFG::STAGE_BYTES) & (RING - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
                }
                if (T_ > 0) mma_k(mtv);
            }
        };
        if (halfn) k_loop(HalfT{});
        else k_loop(FullT{});
    }

    PHASE_STAMP(2);
    // ---- epilogue ----
    if (MICO_GEMM_ABLATE == 6 && g.K > 0) {   // ablation: no epilogue at all (keep the accumulators live)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (!PINGPONG && g.split_k > 1 && g.e.splitk_ws != nullptr) {
        // small-tile kernel, split-K slab path (see gemm_pc_kernel): this split's partial tile -> its fp32 [M, N] slab, plain stores
        __syncthreads();
        GemmArgs gp = g;
        gp.C = (char*)g.e.splitk_ws + (int64_t)ks * g.M * g.N * 4;
        gp.ldc = g.N;
        gp.c_dtype = MICO_F32;
        gp.e.alpha = 1.f;
        gp.e.accumulate = 0;
#pragma unroll
        for (int h = 0; h < MT / 4; ++h) gemm_epilogue_block<T, 4, ACT>(gp, &acc[h * 4], lds + wave * 16384, m0 + wrow + h * 64, n0 + wcol, lane);
    } else if (g.split_k > 1) {   // split-K partials carry no bias / activation / residual (checked on the host side)
#pragma unroll
        for (int h = 0; h < MT / 4; ++h)
            if (h == 0 || !halfn) gemm_epilogue_atomic(g, &acc[h * 4], m0 + wrow + h * 64, n0 + wcol, lane);
    } else {
        __syncthreads();   // every wave is done with the operand tiles (and the DMA queue is empty) before LDS is reused
        PHASE_STAMP(4);
#pragma unroll
        for (int h = 0; h < MT / 4; ++h) {
            if (h > 0 && halfn) break;
            gemm_epilogue_block<T, 4, ACT>(g, &acc[h * 4], lds + wave * 16384, m0 + wrow + h * 64, n0 + wcol, lane);
            if (h == 0) PHASE_STAMP(5);
        }
    }
    PHASE_STAMP(3);
#if MICO_GEMM_ABLATE == 7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_STAMP(6);
#endif
}

// ======================================================================================================================
// EXPERIMENT (round 2), compiled only with -DMICO_GEMM_PERSIST: persistent form of the 8-wave ping-pong kernel (forward / dX GEMMs without
// split-K).  Result (tools/probes/README.md): microbench forward +1 %, dX +5 %; in situ 91.6 vs 92.5 samples/s - no gain, not routed.
// Per 256x256 tile the kernel above pays ~11 us around its K loop - 2.9 us until the first K-tile has landed, a 12.7 us epilogue that
// nothing overlaps, and ~4 us before the hardware has a new workgroup running on the CU - against 36 us of K loop at K = 1408
// (tools/probes/gemm_phases.py).  Here 256 workgroups stay resident and walk the tiles their XCD would have been dealt (same tile
// order, so the same panel sharing in L2); the first two K-tiles of the NEXT tile are requested before the epilogue of the current
// one starts - the epilogue stages through the lower half of the LDS (8 KiB per wave, 32-row blocks), the prefetch lands in the
// upper half (ring slots 2, 3; every tile's ring starts at slot 2) - so that the loads of tile i + 1 travel while the stores of tile i
// drain, and there is no dispatch gap.  One `vmcnt(0)` + barrier between epilogue and K loop (stores and prefetch have both had the
// whole epilogue to complete) keeps the K loop's counted waits free of stores.
// ======================================================================================================================
#ifdef MICO_GEMM_PERSIST
template <typename T, bool TA, bool TB, int ACT>
__global__ __launch_bounds__(Big::THREADS, 2) void gemm_persist_kernel(const GemmArgs g) {
    using CFG = Big;
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, THREADS = CFG::THREADS, MT = CFG::MT;
    constexpr int PT = CFG::A_DMA + CFG::B_DMA, RING = CFG::STAGES * CFG::STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[CFG::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / CFG::WN, wn = wave % CFG::WN;
    const int wrow = wm * (BM / CFG::WM), wcol = wn * 64;
    const int grp = wave >> 2;
    // this workgroup's tiles: positions j, j + per_xcd, ... of XCD x's contiguous chunk of the remapped tile order
    const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx;
    const int x = blockIdx.x % nx, j0 = blockIdx.x / nx, per_xcd = gridDim.x / nx;
    const int chunk0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, chunk_n = q + (x < r ? 1 : 0);
    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
    const int T_ = g.ktiles;
    // (K % 32 == 0 only - mico_gemm routes ragged K to the one-tile-per-workgroup kernel and its masked staging path)
    const FragBase ab = frag_base<TA, BM, BK>(wrow, lane), bb = frag_base<TB, BN, BK>(wcol, lane);

    struct Tile { int tm, tn; __amdgpu_buffer_rsrc_t rsa, rsb; unsigned voa[CFG::A_DMA], vob[CFG::B_DMA]; };
    auto setup = [&](int pos, Tile& t) {
        const int bid = chunk0 + pos;
        const int gsz = GROUP_M * g.ntn;
        const int grp_ = bid / gsz;
        const int first = grp_ * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp_ * gsz;
        t.tm = first + in % gm;
        t.tn = in / gm;
        const int64_t m0 = (int64_t)t.tm * BM, n0 = (int64_t)t.tn * BN;
        const char* a_base = TA ? g.A + m0 * 2 : g.A + m0 * lda_b;
        const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
        int64_t a_bytes = TA ? g.ka_rows * lda_b - m0 * 2 : (g.M - m0) * lda_b;
        int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
        if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
        if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
        t.rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
        t.rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
        dma_offsets<TA, BM, THREADS, BK, CFG::A_DMA>(t.voa, wave, lane, lda_b, g.M - m0);
        dma_offsets<TB, BN, THREADS, BK, CFG::B_DMA>(t.vob, wave, lane, ldb_b, g.N - n0);
    };
    auto stage = [&](const Tile& t, int kt, int bo) {
        const int k0 = kt * BK;
        const unsigned koa = TA ? (unsigned)((int64_t)k0 * lda_b) : (unsigned)(k0 * 2);
        const unsigned kob = TB ? (unsigned)((int64_t)k0 * ldb_b) : (unsigned)(k0 * 2);
        dma_issue<THREADS, CFG::A_DMA>(t.rsa, lds + bo, wave, t.voa, koa);
        dma_issue<THREADS, CFG::B_DMA>(t.rsb, lds + bo + CFG::A_BYTES, wave, t.vob, kob);
    };
    constexpr int SLOT2 = 2 * CFG::STAGE_BYTES;
    Tile cur;
    if (j0 >= chunk_n) return;
    setup(j0, cur);
    for (int i = 0; i < 2 && i < T_; ++i) stage(cur, i, (SLOT2 + i * CFG::STAGE_BYTES) & (RING - 1));

    for (int pos = j0; pos < chunk_n; pos += per_xcd) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (T_ > 2) stage(cur, 2, (SLOT2 + 2 * CFG::STAGE_BYTES) & (RING - 1));
        s16x8 fa[MT], fb[4];
        auto read_k = [&](int bo) {
            LDS_AS const char* ta = lds + bo;
            LDS_AS const char* tb = ta + CFG::A_BYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = read_frag_b<TA, BM, BK>(ta, ab.b0, i);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fb[jj] = read_frag_b<TB, BN, BK>(tb, bb.b0, jj);
        };
        auto mma_k = [&]() {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = T16<T>::mfma(fb[jj], fa[i], acc[i][jj]);
        };
        auto head = [&](int t) {
            const int ahead = T_ - 1 - t;
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PT) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        int bo = SLOT2;
        if (grp == 0) {
            for (int t = 0; t < T_; ++t) {
                asm volatile("" : "+s"(bo));
                head(t);
                read_k(bo);
                if (t + 3 < T_) stage(cur, t + 3, (bo + 3 * CFG::STAGE_BYTES) & (RING - 1));
                bar();
                mma_k();
                __builtin_amdgcn_sched_barrier(0);
                bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
            }
        } else {
            for (int t = 0; t < T_; ++t) {
                asm volatile("" : "+s"(bo));
                head(t);
                if (t > 0) mma_k();
                bar();
                read_k(bo);
                if (t + 3 < T_) stage(cur, t + 3, (bo + 3 * CFG::STAGE_BYTES) & (RING - 1));
                __builtin_amdgcn_sched_barrier(0);
                bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
            }
            if (T_ > 0) mma_k();
        }
        __syncthreads();   // every wave is done with the operand tiles; the DMA queue is empty (last head waited vmcnt(0))
        // ---- the next tile's first two K-tiles travel while this tile's epilogue runs ----
        const int64_t em0 = (int64_t)cur.tm * BM, en0 = (int64_t)cur.tn * BN;
        const bool more = pos + per_xcd < chunk_n;
        if (more) {
            setup(pos + per_xcd, cur);
            for (int i = 0; i < 2 && i < T_; ++i) stage(cur, i, (SLOT2 + i * CFG::STAGE_BYTES) & (RING - 1));
        }
        // (the lane index is laundered: everything the epilogue derives from it alone - column groups, swizzle keys, bias loads - is
        // invariant across this loop and would otherwise be hoisted above it and held in registers through every K loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int h = 0; h < MT / 2; ++h)
            gemm_epilogue_block<T, 2, ACT>(g, &acc[h * 2], lds + wave * 8192, em0 + wrow + h * 32, en0 + wcol, lane_e);
        // stores and prefetch have landed; nobody reads the epilogue staging area any more
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
#endif   // MICO_GEMM_PERSIST

// ======================================================================================================================
// "MID" kernel (round 3): 256x128 output tile, 64-deep K-tiles, FOUR waves (2x2, 128x64 each), TWO workgroups per CU.
// Why: (1) around its K loop the 8-wave kernel spends ~11 us per 256x256 tile (prologue, a 12 us epilogue - a CU stores at ~10 B/clk
// whatever the store shape - and the dispatch gap) with the matrix pipes idle, against ~37 us of K loop at K = 1408; with two independent
// workgroups per CU one's epilogue / prologue runs under the other's K loop.  (2) 128-wide tiles fit N = 1408 / 4224 / 6144 exactly.
// (3) k-contiguous operands arrive in whole 128-byte lines (the fill path moves 61 B/clk/CU in 128-byte row segments, 37 in the 64-byte
// segments of 32-deep stages - tools/probes/dma_fill.hip).  A 64-deep K-tile of this tile is 48 KiB and a workgroup owns 80 KiB, so the
// LDS is a ring of FIVE 16 KiB units - A rows 0-127 | A rows 128-255 | B - filled unit by unit: tile t occupies three consecutive ring
// slots and the DMA of {B(t+1), A-top(t+2), A-bottom(t+2)} goes into them as soon as every wave has read its fragments of tile t
// (second barrier of the iteration), i.e. 1 2/3 K-tiles are resident or in flight at any time.  Counted vmcnt, raw barriers.
// Forward (B k-contiguous) and dX (B reduction-major, transposing reads) orientations; K % 64 == 0; no split-K.
// ======================================================================================================================
struct Mid64 {
    static constexpr int BM = 256, BN = 128, BK = 64, THREADS = 256, MT = 8, UNIT = 128 * 64 * 2, NUNITS = 5, LDS_BYTES = NUNITS * UNIT;
    static constexpr int UD = UNIT / 16 / THREADS;   // DMA instructions per thread per unit (4)
};

template <typename T, bool TB, int ACT>
__global__ __launch_bounds__(Mid64::THREADS, 2) void gemm_mid_kernel(const GemmArgs g) {
    constexpr int BM = Mid64::BM, BN = Mid64::BN, BK = Mid64::BK, THREADS = Mid64::THREADS, MT = Mid64::MT, UNIT = Mid64::UNIT, UD = Mid64::UD;
    __shared__ __attribute__((aligned(16))) char smem[Mid64::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup -> tile: XCD-contiguous remap (bijective), then grouped row-panel order (as gemm_kernel; 64 tiles per XCD at a time)
    int bid = blockIdx.x;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = g.group_m * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * g.group_m;
        const int gm = min(g.ntm - first, g.group_m);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm + g.tm0;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int wm = wave >> 1, wn = wave & 1;
    const int T_ = g.ktiles;

    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
    const char* a_base = g.A + m0 * lda_b;
    const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
    int64_t a_bytes = (g.M - m0) * lda_b;
    int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
    if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
    if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);

    constexpr bool M32 = MICO_MID_M32 && !TB;   // 32x32x16 MFMAs (forward orientation: both operands k-contiguous)
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment addressing inside a unit: a wave reads all 128 rows of ITS A unit (wm picks the unit) and 64 of the B unit's 128 rows / columns
    const FragBase ab = frag_base<false, 128, BK>(0, lane), bb = frag_base<TB, 128, BK>(wn * 64, lane);
    // 32x32x16 fragments: row l & 31 of a 32-row tile, 16-byte chunk 2 ks + (l >> 5) of the row's 64 k (k-step ks of 16); the swizzle key
    // (row >> 1) & 7 does not depend on the tile, so one base per k-step serves both operands (+ tile * 4096 bytes)
    int b32[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b32[ks] = (lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
    unsigned voa[UD], vob[UD];
    dma_offsets<false, 128, THREADS, BK, UD>(voa, wave, lane, lda_b, 128);
    dma_offsets<TB, 128, THREADS, BK, UD>(vob, wave, lane, ldb_b, g.N - n0);
    const unsigned a_half = (unsigned)(128 * lda_b);
    const int nseg = g.e.nseg, kseg = g.e.kseg;
    // unit u = 3 t + w (w: 0 A-top, 1 A-bottom, 2 B) -> ring slot u % 5; cs = slot of tile t's A-top = 3 t mod 5
    auto k_of = [&](int t, int& ka, int& kb) {
        const int k0 = t * BK;
        ka = k0; kb = k0;
        if (nseg > 0) {
            const int sg = k0 / kseg, kin = k0 - sg * kseg;
            ka = g.e.a_seg_off[sg] + kin;
            kb = g.e.b_seg_off[sg] + kin;
        }
    };
    auto issue = [&](int t, int w, int slot) {
        int ka, kb;
        k_of(t, ka, kb);
        LDS_AS char* dst = lds + slot * UNIT;
        if (w < 2) dma_issue<THREADS, UD, UD, TB>(rsa, dst, wave, voa, (unsigned)(ka * 2) + (w ? a_half : 0u));
        else dma_issue<THREADS, UD, UD, TB>(rsb, dst, wave, vob, TB ? (unsigned)((int64_t)kb * ldb_b) : (unsigned)(kb * 2));
    };
    // prologue: A-top / A-bottom / B of tile 0, A-top / A-bottom of tile 1 -> slots 0..4
    if (T_ > 0) { issue(0, 0, 0); issue(0, 1, 1); issue(0, 2, 2); }
    if (T_ > 1) { issue(1, 0, 3); issue(1, 1, 4); }

#if MICO_MID_DB
    s16x8 fa[2][MT], fb[2][4];   // both k-steps' fragments of a K-tile (all 24 reads issued up front; k-step 1's land under k-step 0's MFMAs)
#else
    s16x8 fa[1][MT], fb[1][4];
#endif
    auto read_k = [&](int sa, int sb, int kk) {
        LDS_AS const char* ta = lds + sa * UNIT;
        LDS_AS const char* tb = lds + sb * UNIT;
        const int s = MICO_MID_DB ? kk : 0;
        if constexpr (M32) {   // half kk of the K-tile = k-steps 2 kk, 2 kk + 1 of 16: fa[s][ks2 * 4 + mi], fb[s][ks2 * 2 + nj]
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const int bs = (kk * 2 + ks2 == 0) ? b32[0] : (kk * 2 + ks2 == 1) ? b32[1] : (kk * 2 + ks2 == 2) ? b32[2] : b32[3];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) fa[s][ks2 * 4 + mi] = *(LDS_AS const s16x8*)(ta + bs + mi * 4096);
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) fb[s][ks2 * 2 + nj] = *(LDS_AS const s16x8*)(tb + bs + (wn * 64 + nj * 32) * 128);
            }
            return;
        }
        const int abase = kk ? ab.b1 : ab.b0, bbase = kk ? bb.b1 : bb.b0;
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[s][i] = read_frag_b<false, 128, BK>(ta, abase, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[s][j] = read_frag_b<TB, 128, BK>(tb, bbase, j);
    };
    f32x16 acc32[4][2];
    if constexpr (M32) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[mi][nj][r] = 0.f;
    }
    auto mma_k = [&](int kk) {
        const int s = MICO_MID_DB ? kk : 0;
        if (MICO_GEMM_ABLATE == 3 && g.K > 0) {   // ablation: no MFMA (keep operands live)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(fa[s][i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fb[s][j]));
            return;
        }
        if constexpr (M32) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) acc32[mi][nj] = mfma32<T>(fb[s][ks2 * 2 + nj], fa[s][ks2 * 4 + mi], acc32[mi][nj]);
            return;
        }
        if (MICO_MID_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = T16<T>::mfma(fb[s][j], fa[s][i], acc[i][j]);
        if (MICO_MID_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    };
#if MICO_MID_PRIO == 1   // static priority for every other round of workgroups (block b and b + 256 are the likely co-residents of a CU)
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
#endif
    int cs = 0;   // slot of tile t's first unit
    for (int t = 0; t < T_; ++t) {
        asm volatile("" : "+s"(cs));
        // tile t has landed (this wave's pieces; the barrier publishes everyone's): at most the two A units of tile t+1 stay in flight
        if (t + 1 < T_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * UD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int s1 = cs + 1, s2 = cs + 2;
        if (s1 >= 5) s1 -= 5;
        if (s2 >= 5) s2 -= 5;
        const int sa = wm ? s1 : cs;
        read_k(sa, s2, 0);
#if MICO_MID_DB
        read_k(sa, s2, 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        mma_k(0);
        __builtin_amdgcn_sched_barrier(0);
#if !MICO_MID_DB
        read_k(sa, s2, 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile t have returned ...
        __builtin_amdgcn_s_barrier();                          // ... and so have everyone's: the three slots are free
        __builtin_amdgcn_sched_barrier(0);
        // B(t+1) -> slot of A-top(t), A-top(t+2) -> slot of A-bottom(t), A-bottom(t+2) -> slot of B(t)
        if (MICO_GEMM_ABLATE != 1) {   // (ablation 1: no DMA in the steady state)
            if (t + 1 < T_) issue(t + 1, 2, cs);
            if (t + 2 < T_) { issue(t + 2, 0, s1); issue(t + 2, 1, s2); }
        }
#if MICO_MID_IL
        mma_k(1);
        // the 12 DMA instructions dealt out between the 32 MFMAs (each data-moving LDS-DMA instruction stalls its wave for tens of cycles:
        // behind an MFMA it has just issued that stall costs this wave's matrix-pipe slot nothing)
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 VMEM
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#else
        __builtin_amdgcn_sched_barrier(0);
        mma_k(1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        cs += 3;
        if (cs >= 5) cs -= 5;
    }
    __syncthreads();   // every wave is done with the operand units (and the DMA queue is empty) before LDS is reused
    if constexpr (M32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 blk[4][4];   // [row-tile mi * 2 + column-tile nj][4-column group rg] of the 64x64 block
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        blk[mi * 2 + nj][rg] = (f32x4){acc32[h * 2 + mi][nj][rg * 4], acc32[h * 2 + mi][nj][rg * 4 + 1], acc32[h * 2 + mi][nj][rg * 4 + 2],
                                                       acc32[h * 2 + mi][nj][rg * 4 + 3]};
            gemm_epilogue_block<T, 4, ACT, true>(g, blk, lds + wave * 16384, m0 + wm * 128 + h * 64, n0 + wn * 64, lane);
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) gemm_epilogue_block<T, 4, ACT>(g, &acc[h * 4], lds + wave * 16384, m0 + wm * 128 + h * 64, n0 + wn * 64, lane);
}

// ======================================================================================================================
// "P8" kernel (round 4): 256x256 output tile, 64-deep K-tiles, 8 waves, EIGHT barrier-separated intervals per K-tile.
// What it changes against the 4-stage BK = 32 ping-pong kernel above:
//   * 128-byte DMA rows (a 64-deep row of a k-contiguous operand is one cache line; the fill path moves 128 GB/s per CU in 128-byte
//     segments against 77 in the 64-byte segments of 32-deep stages, tools/probes/dma_fill.hip) in 16 KiB HALF-TILES of 128 rows;
//   * every half-tile is READ EXACTLY ONCE, one phase after the counted wait that retires it, and its LDS slot is refilled four phases
//     later: the DMA stream, the fragment reads and the MFMAs each advance by one quarter tile per phase, so three half-tiles (6 DMA
//     instructions per wave) are always in flight across the barriers - `s_waitcnt vmcnt(6)` in every phase, never 0;
//   * phases of 16 MFMAs with 8 / 4 / 4 / 8 fragment reads and 2 DMA instructions per wave: the two waves of a SIMD (w, w + 4: the two
//     row groups) run the same program half a phase apart - one multiplies while the other reads and issues - and no load section holds
//     more than 8 ds_read_b128 + 2 LDS-DMA against the partner's 256 MFMA cycles (the 32-deep kernel: 12 + 4 against 512).
// Tile decomposition (chosen so that every half-tile is a contiguous 128-row block of its operand):
//   wave (wm = w / 4, wn = w % 4) owns rows   [64 wm, +64) (quadrant row block 0)  and [128 + 64 wm, +64) (block 1)
//                                  and columns [32 wn, +32) (quadrant col block 0) and [128 + 32 wn, +32) (block 1)
//   half-tiles of K-tile t:  A-lo = A rows 0-127 (row block 0 of both wm), A-hi = rows 128-255, B-lo / B-hi likewise in columns.
// Phase p of tile t (cur = tile t's 64 KiB buffer, nxt = the other one); every fragment set is loaded once per tile:
//   p0: read a0 <- A-lo(cur)   | DMA A-lo(t+1) | wait | bar | MFMA a0 x b0 | bar       (b0, a1 were read in p2, p3 of tile t-1)
//   p1: read b1 <- B-hi(cur)   | DMA B-hi(t+1) | wait | bar | MFMA a1 x b0 | bar
//   p2: read b0 <- B-lo(nxt)   | DMA B-lo(t+2) | wait | bar | MFMA a1 x b1 | bar
//   p3: read a1 <- A-hi(nxt)   | DMA A-hi(t+2) | wait | bar | MFMA a0 x b1 | bar
// Stream order of the half-tiles: h = 4 t + {0: B-lo, 1: A-hi, 2: A-lo, 3: B-hi}; h is read in phase h - 2, waited for in phase h - 3
// (before that phase's first barrier: the reading phase is two barriers later for the group that waited, one for the staggered
// group - the DMA of every wave has landed by then), and issued in phase h - 6 into the slot half-tile h - 8 left in phase h - 10.
// Row group 1 executes one extra barrier before the loop (the stagger) and row group 0 one after it.
// Forward (B k-contiguous) and dX (B reduction-major, transposing reads, two [64][128] images) orientations; K % 64 == 0; no split-K.
// ======================================================================================================================
// Specialised epilogue of the 8-phase kernel for its hot launches: 16-bit output (and the MLP pair's 16-bit auxiliary tensor), optional bias,
// alpha - no row scale / frame scatter / residual / accumulate (host-checked, wave-uniform GemmArgs::fast16).  Same LDS staging image and
// read-back ownership as gemm_epilogue_block (a lane owns 8 consecutive columns of each of the wave's two 32-column strips in one row per
// pass), but straight-line: the bias is loaded once per tile (both 64-row blocks share the columns) and its latency hides behind the staging
// writes, the four passes of a block are unrolled (16 ds_read_b128 in flight), addresses are 32-bit offsets from a per-block scalar base,
// and none of the generic epilogue's per-pass bookkeeping (64-bit row arithmetic, row_map / row_scale lookups, feature branches) exists.
// The generic epilogue takes ~3.3 us per 64-row block and wave (tools/probes/gemm_phases.py), most of it exposed latency.
#ifndef MICO_P8_NTSTORE
#define MICO_P8_NTSTORE 0   // 1: the fast epilogue's C / aux stores carry the non-temporal hint (A/B: tools/probes/README.md)
#endif
template <typename V> __device__ __forceinline__ void p8_store(char* p, V v) {
    if constexpr (MICO_P8_NTSTORE != 0) __builtin_nontemporal_store(v, (V*)p); else *(V*)p = v;
}
template <typename T, int ACT>
__device__ __forceinline__ void p8_epilogue_fast16(const GemmArgs& g, const f32x4 (&acc)[8][4], LDS_AS char* wbuf, int64_t m0, int64_t n0, int wm, int wn,
                                                   int lane) {
    const int p = lane & 15, gq = lane >> 4, kp = epi_key(p);
    const int q = lane & 3, rr = lane >> 2;
    const float alpha = g.e.alpha;
    const int64_t ncol0 = n0 + wn * 32;
    // the lane's two 8-column groups (u = 0: strip 0, u = 1: strip 1) relative to ncol0, and the validity of their 4-column halves
    const int gcol[2] = {q * 8, 128 + q * 8};
    bool ok[2][2];
    f32x4 bias[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ok[u][h] = ncol0 + gcol[u] + h * 4 < g.N;
            bias[u][h] = (g.e.bias && ok[u][h]) ? *(const f32x4*)(g.e.bias + ncol0 + gcol[u] + h * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const unsigned ldc2 = (unsigned)(g.ldc * 2), ldaux2 = (unsigned)(g.e.ldaux * 2);
    // (The activation epilogues are VALU work - exact-erf GELU: ~18 instruction slots per element, two of them quarter rate, ~10 us per tile -
    // in front of stores a CU retires at ~10 B/clk (6.7 us per tile), and the two ADD UP: 47 -> 56 us per tile for GELU, 57.6 for the pair.
    // Delaying the second wave of every SIMD by 8 ... 64 s_sleep units at this point, so that one wave computes while the other's stores drain,
    // changed nothing - pair 904 -> 893-901, GELU 1031 -> 1026-1038 TFLOP/s, tools/probes/README.md round 5: the waves are not in lockstep.)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int64_t mrow0 = m0 + hb * 128 + wm * 64;
        char* const cbase = g.C + (mrow0 * g.ldc + ncol0) * 2;
        char* const abase = (ACT == MICO_ACT_GELU_SAVE_DERIV) ? (char*)g.e.aux_out + (mrow0 * g.e.ldaux + ncol0) * 2
                          : (ACT == MICO_ACT_MUL_AUX) ? (char*)g.e.aux_in + (mrow0 * g.e.ldaux + ncol0) * 2 : nullptr;
        const int rows_left = (int)(g.M - mrow0 < 64 ? g.M - mrow0 : 64);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(LDS_AS f32x4*)(wbuf + (i * 16 + p) * 256 + (((j * 4 + gq) ^ kp) << 4)) = acc[hb * 4 + i][j] * alpha;
        s16x8 aux[4][2];
        if constexpr (ACT == MICO_ACT_MUL_AUX) {   // the stored derivative: requested before the staged block is read back
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = ps * 16 + rr;
                    aux[ps][u] = (row < rows_left && ok[u][1]) ? *(const s16x8*)(abase + row * ldaux2 + gcol[u] * 2) : (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (row < rows_left && ok[u][0] && !ok[u][1]) {
                        const s16x4 a4 = *(const s16x4*)(abase + row * ldaux2 + gcol[u] * 2);
                        aux[ps][u] = (s16x8){a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
                    }
                }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 16 + rr;
            const int kr = epi_key(row);
            f32x4 v[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    v[u][h] = *(LDS_AS const f32x4*)(wbuf + row * 256 + (((u * 8 + q * 2 + h) ^ kr) << 4)) + bias[u][h];
            if (row >= rows_left) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u][0]) continue;
                if constexpr (ACT == MICO_ACT_GELU_SAVE_DERIV) {
                    f32x4 d[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[u][h] = gelu_pair4(v[u][h], d[h]);
                    const s16x4 lo = pack4<T>(d[0][0], d[0][1], d[0][2], d[0][3]), hi = pack4<T>(d[1][0], d[1][1], d[1][2], d[1][3]);
                    char* ap = abase + row * ldaux2 + gcol[u] * 2;
                    if (ok[u][1]) p8_store(ap, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                    else p8_store(ap, lo);
                } else if constexpr (ACT == MICO_ACT_MUL_AUX) {
                    const s16x8 a8 = aux[ps][u];
                    v[u][0] *= unpack4<T>((s16x4){a8[0], a8[1], a8[2], a8[3]});
                    v[u][1] *= unpack4<T>((s16x4){a8[4], a8[5], a8[6], a8[7]});
                } else if constexpr (ACT == MICO_ACT_GELU) {   // GELU alone (fc1 of a forward that keeps no derivative: no grad / activation diet)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[u][h] = gelu4(v[u][h]);
                }
                const s16x4 lo = pack4<T>(v[u][0][0], v[u][0][1], v[u][0][2], v[u][0][3]), hi = pack4<T>(v[u][1][0], v[u][1][1], v[u][1][2], v[u][1][3]);
                char* cp = cbase + row * ldc2 + gcol[u] * 2;
                if (ok[u][1]) p8_store(cp, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                else p8_store(cp, lo);
            }
        }
    }
}

// Round 5: the same epilogue with the tile staged in 16 BITS.  tools/probes/store_scaling.hip showed that the stores are not what the 6.7 us of
// p8_epilogue_fast16 are: a CU writes a 128 KiB tile in 1.1 us alone and in 1.9 us when all 256 CUs burst together after 40 us of silence (the GEMM's
// regime) - the time is the two LDS round trips of 16 KiB fp32 blocks per wave (write, wait, read, wait, twice) in front of them.  Here bias / alpha /
// activation are applied in the MFMA layout (elementwise: the layout does not matter), the values are rounded to the output type and staged as 8-byte
// pieces - BOTH 64-row blocks at once in the wave's 16 KiB (128 rows x 128 B), half the LDS bytes, one round trip - and read back as 16-byte pieces
// (8 columns of one row per lane: 64 contiguous bytes per lane quad, as before).  Same fp32 arithmetic per element in the same order (the product
// with alpha is kept from contracting with the bias add: the fp32 staging separated the two), so the outputs are bit-identical
// (tools/gemm_bench.py --variants on the probe build; MICO_P8_EPI16=0 rebuilds the fp32 staging).  The GELU pair stages its two outputs per
// 64-row block (8 + 8 KiB).  MUL_AUX keeps the fp32 staging: its second factor arrives in the read-back layout and multiplies before the rounding.
// Chunk swizzle of the 128-byte rows: key(row) = bit 0 of the row -> bit 2, bits 1-2 -> bits 0-1: the 16 rows x 8 bytes of a write instruction use
// every bank exactly twice per 32 lanes (the minimum for 256 bytes), and the two rows x four chunks of 8 consecutive reading lanes are 8 distinct chunks.
#ifndef MICO_P8_EPI16
#define MICO_P8_EPI16 1
#endif
__device__ __forceinline__ int epi16_key(int row) { return ((row & 1) << 2) | ((row >> 1) & 3); }
template <typename T, int ACT>
__device__ __forceinline__ void p8_epilogue_fast16_h(const GemmArgs& g, const f32x4 (&acc)[8][4], LDS_AS char* wbuf, int64_t m0, int64_t n0, int wm, int wn,
                                                     int lane) {
    static_assert(ACT == ACT_LEAN || ACT == MICO_ACT_GELU || ACT == MICO_ACT_GELU_SAVE_DERIV, "MUL_AUX keeps the fp32 staging");
    const int p = lane & 15, gq = lane >> 4;
    const int q = lane & 3, rr = lane >> 2;
    const float alpha = g.e.alpha;
    const int64_t ncol0 = n0 + wn * 32;
    f32x4 bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t col = ncol0 + (j >> 1) * 128 + (j & 1) * 16 + gq * 4;
        bias[j] = (g.e.bias && col < g.N) ? *(const f32x4*)(g.e.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int wkey = epi16_key(p), rkey = epi16_key(rr);
    const int woff = p * 128 + (gq & 1) * 8, wch = gq >> 1;
    auto value = [&](int i, int j) {
        f32x4 t = acc[i][j] * alpha;
        asm volatile("" : "+v"(t));      // (no fused multiply-add with the bias: the fp32 staging rounded the product first)
        return t + bias[j];
    };
    auto store_row = [&](char* base, int64_t ld2, int64_t grow, int u, s16x8 o) {
        const int64_t col = ncol0 + u * 128 + q * 8;
        if (grow >= g.M || col >= g.N) return;
        char* cp = base + grow * ld2 + col * 2;
        if (col + 4 < g.N) p8_store(cp, o);
        else p8_store(cp, (s16x4){o[0], o[1], o[2], o[3]});
    };
    const int64_t ldc2 = g.ldc * 2;
    if constexpr (ACT != MICO_ACT_GELU_SAVE_DERIV) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = value(i, j);
                if constexpr (ACT == MICO_ACT_GELU) v = gelu4(v);
                *(LDS_AS s16x4*)(wbuf + i * 2048 + woff + (((j * 2 + wch) ^ wkey) << 4)) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        s16x8 o[8][2];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps)
#pragma unroll
            for (int u = 0; u < 2; ++u) o[ps][u] = *(LDS_AS const s16x8*)(wbuf + ps * 2048 + rr * 128 + (((u * 4 + q) ^ rkey) << 4));
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int64_t grow = m0 + (ps >> 2) * 128 + wm * 64 + (ps & 3) * 16 + rr;
#pragma unroll
            for (int u = 0; u < 2; ++u) store_row(g.C, ldc2, grow, u, o[ps][u]);
        }
    } else {
        const int64_t lda2 = g.e.ldaux * 2;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 d;
                    const f32x4 v = gelu_pair4(value(hb * 4 + i, j), d);
                    const int off = i * 2048 + woff + (((j * 2 + wch) ^ wkey) << 4);
                    *(LDS_AS s16x4*)(wbuf + off) = pack4<T>(v[0], v[1], v[2], v[3]);
                    *(LDS_AS s16x4*)(wbuf + 8192 + off) = pack4<T>(d[0], d[1], d[2], d[3]);
                }
            s16x8 o[4][2], od[4][2];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int off = ps * 2048 + rr * 128 + (((u * 4 + q) ^ rkey) << 4);
                    o[ps][u] = *(LDS_AS const s16x8*)(wbuf + off);
                    od[ps][u] = *(LDS_AS const s16x8*)(wbuf + 8192 + off);
                }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int64_t grow = m0 + hb * 128 + wm * 64 + ps * 16 + rr;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    store_row(g.C, ldc2, grow, u, o[ps][u]);
                    store_row((char*)g.e.aux_out, lda2, grow, u, od[ps][u]);
                }
            }
        }
    }
}

#ifndef MICO_P8_PRIO
#define MICO_P8_PRIO 0     // s_setprio(1) around the 16-MFMA bursts
#endif
#ifndef MICO_P8_DEFAULT
#define MICO_P8_DEFAULT 1  // 1: default routing (variant 0) sends the eligible forward / dX problems here
#endif
#ifndef MICO_P8_STAGGER
#define MICO_P8_STAGGER 0
#endif
#ifndef MICO_P8_FAST16
#define MICO_P8_FAST16 1   // specialised epilogue for plain 16-bit outputs (variant 15 = P8 with the generic epilogue, for A/B runs)
#endif
#ifndef MICO_P8_SPLIT
#define MICO_P8_SPLIT 0    // 1: rows beyond the last full round of 256 tiles go to the 256x128 kernel (see mico_gemm); variant 14 forces it on
#endif
#ifndef MICO_P8_WALK
#define MICO_P8_WALK 1     // quadrant walk: 0 = (a0,b0) (a1,b0) (a1,b1) (a0,b1) - a0's 8 reads are the ones needed in the phase that issues them;
#endif                     //                1 = (a0,b0) (a0,b1) (a1,b1) (a1,b0) - b0's 4 reads are
#ifndef MICO_P8_L
#define MICO_P8_L 4        // phases between the issue of a half-tile and the wait that retires it, + 1 (3 / 4 / 5: 2 / 3 / 4 half-tiles in flight)
#endif
#ifndef MICO_P8_LGKM0
#define MICO_P8_LGKM0 0    // 1: s_waitcnt lgkmcnt(0) right after each phase's first barrier (all reads issued in the load section land first)
#endif
struct P8C {
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512, HALF = 16384, TILE = 65536, LDS_BYTES = 2 * TILE;
    // LDS-DMA instructions one wave issues per half-tile (16 bytes per lane): every hand-counted `s_waitcnt vmcnt(N)` of the 8-phase kernels
    // is a multiple of it (+ the scale-word loads of the fp8 form) - the asm DMA is invisible to the compiler's counters, so a change of the
    // stage size or the workgroup shape has to fail HERE, not as an LDS race
    static constexpr int DMA_PER_HALF = HALF / (THREADS * 16);
    static_assert(DMA_PER_HALF * THREADS * 16 == HALF && DMA_PER_HALF == 2, "the vmcnt immediates of gemm_p8_kernel / gemm_p8mx_kernel assume two DMA instructions per wave and half-tile");
};

template <typename T, bool TB, int ACT>
__global__ __launch_bounds__(P8C::THREADS, 2) void gemm_p8_kernel(const GemmArgs g) {
    constexpr int BM = P8C::BM, BN = P8C::BN, BK = P8C::BK, HALF = P8C::HALF, TILE = P8C::TILE;
    __shared__ __attribute__((aligned(16))) char smem[P8C::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PHASE_STAMP(0);
#if MICO_P8_STAGGER > 0
    // experiment (negative result, tools/probes/README.md): the first round's workgroups start MICO_P8_STAGGER * 0.32 us apart in four
    // groups per XCD so that the rounds' epilogue store bursts do not coincide - layer forward 1121 -> 1114 / 1108 / 1090 TFLOP/s at 8 / 16 / 32
    if (blockIdx.x < 256) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long dl = (unsigned long long)(((blockIdx.x >> 3) & 3) * MICO_P8_STAGGER * 32);
        while (__builtin_amdgcn_s_memrealtime() - t0 < dl) __builtin_amdgcn_s_sleep(8);
    }
#endif
    // workgroup -> tile: XCD-contiguous remap (bijective), then grouped row-panel order (as gemm_kernel)
    int bid = blockIdx.x;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm + g.tm0;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int wm = wave >> 2, wn = wave & 3;
    const int T_ = g.ktiles;

    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
    const char* a_base = g.A + m0 * lda_b;
    const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
    int64_t a_bytes = (g.M - m0) * lda_b;
    int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
    if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
    if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);

    f32x4 acc[8][4];   // [row tile: 0-3 row block 0, 4-7 row block 1][column tile: 0-1 column block 0, 2-3 column block 1]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- per-lane DMA source offsets of a half-tile (two instructions per wave: it = 0, 1), relative to (half, K-tile) ----
    // k-contiguous operand: half-tile = 128 rows x 128 B; instruction `it` of wave w covers rows it * 64 + 8 w .. + 7 (8 lanes per row)
    // reduction-major B   : half-tile = 64 k-rows x 256 B (128 columns); instruction `it` covers k-rows it * 32 + 4 w .. + 3 (16 lanes per row)
    // The row part lives in 4 + 4 loop-invariant VGPRs (it can leave the buffer: rows / columns past the matrix edge must zero-fill),
    // the K-tile part in the instruction's scalar offset.
    const int rl = wave * 8 + (lane >> 3);
    unsigned vra[2][2], vrb[2][2];   // [half][it]
    {
        const unsigned va = (unsigned)(rl * lda_b) + (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < 2; ++it) vra[h][it] = va + (unsigned)((h * 128 + it * 64) * lda_b);
        if constexpr (!TB) {
            const unsigned vb = (unsigned)(rl * ldb_b) + (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int it = 0; it < 2; ++it) vrb[h][it] = vb + (unsigned)((h * 128 + it * 64) * ldb_b);
        } else {
            const int kr = wave * 4 + (lane >> 4), cg = (lane & 15) ^ key_tr(kr);
            const int64_t crem = g.N - n0;      // columns past the matrix edge: out of bounds -> zero fill (the descriptor only bounds the last row)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    vrb[h][it] = (h * 128 + cg * 8 >= crem) ? 0xFFFFFFF0u : (unsigned)((kr + it * 32) * ldb_b) + (unsigned)(cg << 4) + (unsigned)(h * 256);
        }
    }
    const unsigned ld_dst = (unsigned)(wave * 1024);
    // half-tile h of the stream (h = 4 t + w; w: 0 B-lo, 1 A-hi, 2 A-lo, 3 B-hi) -> the buffer of tile t; past the last tile: out-of-bounds
    // DMA (zero fill into a dead slot, no memory traffic) so that the counted waits see the same number of instructions in every phase
    auto issue = [&](int t, auto wv) {
        constexpr int W = decltype(wv)::value;
        // stream position W of a tile -> (operand, half):  walk 0: B-lo, A-hi, A-lo, B-hi    walk 1: A-lo, B-hi, B-lo, A-hi
        constexpr bool isA = MICO_P8_WALK ? (W == 0 || W == 3) : (W == 1 || W == 2);
        constexpr int half = (W == 1 || W == 3) ? 1 : 0;
        const bool valid = t < T_;
        // (k-segment launches stay on the 32-deep kernels, except the one pattern that is a plain product here: the SAME activation against
        // the weight's hi and lo halves, adjacent along K - the A stream simply starts over after a_wrap K-tiles)
        const int ta_ = (!TB && g.a_wrap > 0 && t >= g.a_wrap) ? t - g.a_wrap : t;
        const unsigned soff = isA ? (unsigned)(ta_ * BK * 2) : (TB ? (unsigned)((int64_t)t * BK * ldb_b) : (unsigned)(t * BK * 2));
        LDS_AS char* dst = lds + (t & 1) * TILE + (isA ? 0 : 2 * HALF) + half * HALF + ld_dst;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned v = isA ? vra[half][it] : vrb[half][it];
            if (!valid) v = 0xFFFFFFF0u;
            if (MICO_GEMM_ABLATE == 4) v |= 0xFFFFFFF0u;
            lds_dma16_soff(isA ? rsa : rsb, (LDS_AS void*)(dst + it * 8192), v, soff);
        }
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;

    // ---- fragments ----
    const FragBase ab = frag_base<false, 256, BK>(wm * 64, lane);
    const FragBase bb = TB ? frag_base<true, 128, BK>(wn * 32, lane) : frag_base<false, 256, BK>(wn * 32, lane);
    s16x8 a0[4][2] = {}, a1[4][2] = {}, b0[2][2] = {}, b1[2][2] = {};   // [16-row / 16-column tile][k-step]
    auto rdA = [&](s16x8 (&d)[4][2], int boff, int blk) {
        if (MICO_GEMM_ABLATE == 2 && boff >= 0) {   // ablation: no LDS reads in the loop (fragments kept opaque)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(d[i][kk]));
            return;
        }
        LDS_AS const char* t = lds + boff + blk * HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i][kk] = read_frag_b<false, 256, BK>(t, kk ? ab.b1 : ab.b0, i);
    };
    auto rdB = [&](s16x8 (&d)[2][2], int boff, int blk) {
        if (MICO_GEMM_ABLATE == 2 && boff >= 0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(d[j][kk]));
            return;
        }
        LDS_AS const char* t = lds + boff + 2 * HALF + blk * HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) d[j][kk] = read_frag_b<TB, TB ? 128 : 256, BK>(t, kk ? bb.b1 : bb.b0, j);
    };
    auto mma = [&](const s16x8 (&a)[4][2], const s16x8 (&b)[2][2], auto iqv, auto jqv) {
        constexpr int IQ = decltype(iqv)::value, JQ = decltype(jqv)::value;
        if (MICO_GEMM_ABLATE == 3) {   // ablation: no MFMA (operands kept live)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[i][kk]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[j][kk]));
            }
            return;
        }
        if (MICO_P8_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[IQ * 4 + i][JQ * 2 + j] = T16<T>::mfma(b[j][kk], a[i][kk], acc[IQ * 4 + i][JQ * 2 + j]);
        if (MICO_P8_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    constexpr int L = MICO_P8_L, VMW = P8C::DMA_PER_HALF * (L - 1);   // L - 1 half-tiles stay in flight across every barrier
    static_assert(L >= 3 && L <= 5, "lookahead of the half-tile stream");
    auto wait_bar = [&]() {   // end of a load section: this wave's share of the half-tile read in the NEXT phase has landed; rendezvous
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (MICO_P8_LGKM0) __builtin_amdgcn_s_waitcnt(0xC07F);
        fence();
    };
    auto bar = [&]() {
        fence();
        __builtin_amdgcn_s_barrier();
        fence();
    };

    // ---- prologue: half-tiles 0 .. L + 1 of the stream; the two fragment sets phase 0 finds loaded ----
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto issue_h = [&](int t, auto pv) {   // the half-tile issued in phase p of tile t: h = 4 t + p + L + 2
        constexpr int IDX = decltype(pv)::value + L + 2;
        if (MICO_GEMM_ABLATE == 1) return;   // ablation: no DMA in the steady state (the counted waits pass at once)
        issue(t + IDX / 4, std::integral_constant<int, IDX % 4>{});
    };
    issue(0, W0{}); issue(0, W1{}); issue(0, W2{}); issue(0, W3{});
    if (L >= 3) issue(1, W0{});
    if (L >= 4) issue(1, W1{});
    if (L >= 5) issue(1, W2{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");   // stream positions 0, 1, 2 of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    fence();
    if (MICO_P8_WALK == 0) { rdB(b0, 0, 0); rdA(a1, 0, 1); }
    else { rdA(a0, 0, 0); rdB(b1, 0, 1); }
    fence();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // the stagger: row group 1 runs half a phase behind row group 0
    fence();
    PHASE_STAMP(1);
    // (a half-width edge tile - N = 1408 = 5.5 x 256 - runs all four quadrants: a run-time branch around the column-block-1 MFMAs leaves
    // the b1 reads pending in hipcc's scoreboard on the skipping path and it then waits lgkmcnt(0) in front of the first DMA of EVERY
    // iteration; an else-branch with a visible s_waitcnt, or two copies of the loop, cost 40-80 spilled registers)
    int cur = 0;
    for (int t = 0; t < T_; ++t) {
        asm volatile("" : "+s"(cur));
        const int nxt = cur ^ TILE;
        if (MICO_P8_WALK == 0) {
            rdA(a0, cur, 0);  issue_h(t, I0{}); wait_bar(); mma(a0, b0, I0{}, I0{}); bar();
            rdB(b1, cur, 1);  issue_h(t, I1{}); wait_bar(); mma(a1, b0, I1{}, I0{}); bar();
            rdB(b0, nxt, 0);  issue_h(t, std::integral_constant<int, 2>{}); wait_bar(); mma(a1, b1, I1{}, I1{}); bar();
            rdA(a1, nxt, 1);  issue_h(t, std::integral_constant<int, 3>{}); wait_bar(); mma(a0, b1, I0{}, I1{}); bar();
        } else {
            rdB(b0, cur, 0);  issue_h(t, I0{}); wait_bar(); mma(a0, b0, I0{}, I0{}); bar();
            rdA(a1, cur, 1);  issue_h(t, I1{}); wait_bar(); mma(a0, b1, I0{}, I1{}); bar();
            rdA(a0, nxt, 0);  issue_h(t, std::integral_constant<int, 2>{}); wait_bar(); mma(a1, b1, I1{}, I1{}); bar();
            rdB(b1, nxt, 1);  issue_h(t, std::integral_constant<int, 3>{}); wait_bar(); mma(a1, b0, I1{}, I0{}); bar();
        }
        cur = nxt;
    }
    PHASE_STAMP(2);
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs issued past the last tile must not land in the epilogue's staging
    __syncthreads();
    PHASE_STAMP(4);
    if (MICO_GEMM_ABLATE == 6 && g.K > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if constexpr (ACT == ACT_LEAN || ACT == MICO_ACT_GELU_SAVE_DERIV || ACT == MICO_ACT_MUL_AUX || ACT == MICO_ACT_GELU) {
        // (256 rows of ldc / ldaux 16-bit elements fit 32-bit byte offsets: checked on the host - GemmArgs::fast16)
        if (MICO_P8_FAST16 && g.fast16) {
            if constexpr (MICO_P8_EPI16 != 0 && ACT != MICO_ACT_MUL_AUX) p8_epilogue_fast16_h<T, ACT>(g, acc, lds + wave * 16384, m0, n0, wm, wn, lane);
            else p8_epilogue_fast16<T, ACT>(g, acc, lds + wave * 16384, m0, n0, wm, wn, lane);
            PHASE_STAMP(5);
            PHASE_STAMP(3);
            return;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gemm_epilogue_block<T, 4, ACT, false, 96>(g, &acc[h * 4], lds + wave * 16384, m0 + h * 128 + wm * 64, n0 + wn * 32, lane);
        if (h == 0) PHASE_STAMP(5);
    }
    PHASE_STAMP(3);
#if MICO_GEMM_ABLATE == 7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_STAMP(6);
#endif
}

// ======================================================================================================================
// PERSISTENT form of the 8-phase kernel (round 5) for the launches whose epilogue is the 16-bit straight-line one (lean / GELU / GELU pair).
// tools/probes/gemm_phases.py on a K = 1408 tile: prologue 2.3 us (workgroup start, descriptors, ~1.5 us of DMA latency before the first MFMA)
// + K loop 41.4 + epilogue 4.35 + 1.2 us between a workgroup's end and its successor's first instruction = 49.2 us, of which 3.5 us are spent
// with NOTHING in flight.  Here min(256, tiles) workgroups walk the same XCD-contiguous tile list (virtual block id += gridDim.x: a workgroup's
// tiles stay on its XCD, the order inside an XCD's chunk is the dispatcher's own) and the stream positions 0 .. 5 of tile i + 1 - the six
// half-tiles the prologue issues - are requested BEFORE the epilogue of tile i, into the six ring slots the epilogue does not use (its 16-bit
// staging needs 4 KiB per wave: 32 rows at a time, the two free slots of the second tile buffer), so they land while the tile is stored.
// The epilogue's stores go through a buffer descriptor with per-lane out-of-bounds offsets (rows beyond M by the descriptor's extent, columns
// beyond N by the offset).  The K loop's counted waits are the one-tile kernel's vmcnt(6): they never rely on a store still being outstanding
// (see VM0 below).  N % 8 == 0 (no 8-byte tail stores), reduction-major B: N % 128 == 0, no k-segments except the wrapped A stream, fast16
// launches only (mico_gemm routes; MICO_P8_PERSIST=0 or, in the probe build, variant 16 = off).
// ======================================================================================================================
#ifndef MICO_P8_PERSIST
#define MICO_P8_PERSIST 1
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
// Epilogue of the persistent 8-phase kernels (16-bit and MX-fp8): the tile's accumulators through 4 KiB of LDS per wave (32 rows of 128 bytes per
// round; the GELU pair: 16 rows x two outputs) and out through buffer descriptors whose base is the tile's (m0, n0) - rows beyond M end the buffer,
// columns beyond N get an out-of-range offset.  Arithmetic per element as p8_epilogue_fast16_h.
// The TILED layout of the GELU' tensor (mico_gemm_epilogue::aux_tiled; include/mico_hip.h): written by the pair epilogue and read by the MUL_AUX
// epilogue of THESE kernels only, so it is stored the way their accumulators hold it - tile (tm, tn) of 256 x 256 is the 128 KiB at
// ((tm * N / 256) + tn) * 128 KiB, wave w its 16 KiB at w * 16 KiB, and unit u (0..15) of a wave is one KiB: lane l's 16 bytes = the eight values
// acc[u >> 1][2 (u & 1)] and acc[u >> 1][2 (u & 1) + 1] of that lane.  Both sides move it with whole-KiB 16-byte-per-lane buffer accesses straight
// from / to registers (no LDS staging; the accumulator layout of a ROW-major aux is 32-byte row segments: measured 6 % of the launch).
// MUL_AUX: all 16 loads of a wave are issued BEFORE the K loop's closing wait (the fragment registers are dead by then), so their latency is behind
// the next tile's requests and the first epilogue rounds; loads return in order, so they must be older than those 96 KiB of requests.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p8p_aux_tile_desc(const void* aux, const GemmArgs& g, int64_t m0, int64_t n0) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)aux + ((m0 >> 8) * (g.N >> 8) + (n0 >> 8)) * 131072), 0, 131072, 0x00020000);
}
__device__ __forceinline__ void p8p_aux_load(i32x4 (&xa)[16], __amdgpu_buffer_rsrc_t rsx, int wave, int lane) {
    int le = lane;
    asm volatile("" : "+v"(le));
    const unsigned base = (unsigned)(wave * 16384 + le * 16);
#pragma unroll
    for (int u = 0; u < 16; ++u) xa[u] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(base + u * 1024), 0, 0));
}
// ACT_PAIR_TILED / ACT_MUL_TILED: the GELU pair / the GELU' multiply with the tiled aux layout (rsx: p8p_aux_tile_desc).
// `xa` / `more`: ACT_MUL_TILED only (the multipliers of p8p_aux_load, 16 loads older than the next tile's twelve requests when there is a next tile).
// ACT_PRE_TILED / ACT_GRAD_TILED (round 6): the MLP pair that keeps the PRE-ACTIVATION h = x W1^T + b1 (one 16-bit tensor per block, in the same tiled
// layout) instead of gelu(h) AND gelu'(h): the forward is GELU with h leaving from the registers (no GELU' arithmetic), and fc2's dX launch reads h
// back, multiplies its result by gelu'(h) and writes gelu(h) - the operand of fc2's weight gradient, which therefore runs AFTER it - row-major through
// `rso` next to its own output (MICO_ACT_GELU + aux_out / MICO_ACT_GELU_GRAD + aux_in + aux_out with aux_tiled; include/mico_hip.h).
constexpr int ACT_PAIR_TILED = 7, ACT_MUL_TILED = 8, ACT_PRE_TILED = 9, ACT_GRAD_TILED = 10;
template <typename T, int ACT>
__device__ __forceinline__ void p8p_epilogue(const GemmArgs& g, const f32x4 (&acc)[8][4], LDS_AS char* wbuf, __amdgpu_buffer_rsrc_t rsc, __amdgpu_buffer_rsrc_t rsx,
                                             __amdgpu_buffer_rsrc_t rso, int64_t n0e, int wm, int wn, int lane, const i32x4* xa = nullptr, bool more = false) {
        int le = lane;
        asm volatile("" : "+v"(le));      // (lane arithmetic of the epilogue recomputed per tile: hoisted, it would live through every K loop)
        const int p = le & 15, gq = le >> 4, q = le & 3, rr = le >> 2;
                const float alpha = g.e.alpha;
        const int64_t ncol0 = n0e + wn * 32;
        f32x4 bias[ACT == ACT_GRAD_TILED ? 1 : 4];      // (GRAD_TILED: an input-gradient launch, routed here without a bias only - 16 registers its epilogue needs)
        if constexpr (ACT != ACT_GRAD_TILED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t col = ncol0 + (j >> 1) * 128 + (j & 1) * 16 + gq * 4;
                bias[j] = (g.e.bias && col < g.N) ? *(const f32x4*)(g.e.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const int wkey = epi16_key(p), rkey = epi16_key(rr);
        const int woff = p * 128 + (gq & 1) * 8, wch = gq >> 1;
        const unsigned ldc2 = (unsigned)(g.ldc * 2), ldx2 = (unsigned)(g.e.ldaux * 2);
        unsigned coff[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) coff[u] = (ncol0 + u * 128 + q * 8 < g.N) ? (unsigned)((wn * 32 + u * 128 + q * 8) * 2) : 0xFFFFFFF0u;
        auto value = [&](int i, int j) {
            f32x4 t = acc[i][j] * alpha;
            if constexpr (ACT == ACT_GRAD_TILED) return t;
            asm volatile("" : "+v"(t));
            return t + bias[j];
        };
        auto bstore = [&](__amdgpu_buffer_rsrc_t rs, unsigned ld2, int row_in_tile, int u, s16x8 o) {
            const unsigned off = coff[u] == 0xFFFFFFF0u ? 0xFFFFFFF0u : (unsigned)row_in_tile * ld2 + coff[u];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rs, (int)off, 0, 0);
        };
        if constexpr (ACT == ACT_GRAD_TILED) {
            // eight rounds of 16 rows x two outputs (the SAVE_DERIV staging): round i reads the pre-activation units 2 i, 2 i + 1 of p8p_aux_load -
            // 2 (7 - i) younger loads of it + the next tile's twelve requests may stay out (never a store counted on: see gemm_p8p_kernel)
            auto round = [&](auto iv) {
                constexpr int i = decltype(iv)::value;
                __builtin_amdgcn_sched_barrier(0);
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + 2 * (7 - i)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (7 - i)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4 x4 = xa[i * 2 + (j >> 1)];
                    const f32x4 h = unpack4<T>(__builtin_bit_cast(s16x4, (j & 1) ? (i32x2){x4[2], x4[3]} : (i32x2){x4[0], x4[1]}));
                    f32x4 d;
                    const f32x4 gl = gelu_pair4(h, d);
                    const f32x4 v = value(i, j) * d;
                    const int off = woff + (((j * 2 + wch) ^ wkey) << 4);
                    *(LDS_AS s16x4*)(wbuf + off) = pack4<T>(v[0], v[1], v[2], v[3]);
                    *(LDS_AS s16x4*)(wbuf + 2048 + off) = pack4<T>(gl[0], gl[1], gl[2], gl[3]);
                }
                const int row = (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + rr;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int off = rr * 128 + (((u * 4 + q) ^ rkey) << 4);
                    bstore(rsc, ldc2, row, u, *(LDS_AS const s16x8*)(wbuf + off));
                    bstore(rso, ldx2, row, u, *(LDS_AS const s16x8*)(wbuf + 2048 + off));
                }
            };
            round(std::integral_constant<int, 0>{}); round(std::integral_constant<int, 1>{}); round(std::integral_constant<int, 2>{}); round(std::integral_constant<int, 3>{});
            round(std::integral_constant<int, 4>{}); round(std::integral_constant<int, 5>{}); round(std::integral_constant<int, 6>{}); round(std::integral_constant<int, 7>{});
        } else if constexpr (ACT != MICO_ACT_GELU_SAVE_DERIV) {
            s16x4 dlo = {0, 0, 0, 0};
            const unsigned xbase = (unsigned)((wm * 4 + wn) * 16384 + le * 16);
            (void)dlo; (void)xbase;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (ACT == ACT_MUL_TILED) {
                    // round r reads units 4 r .. 4 r + 3: 4 (3 - r) younger multiplier loads + the next tile's twelve requests may stay out (never a
                    // store counted on: see gemm_p8p_kernel)
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        if (r == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                        else if (r == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                        else if (r == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    } else {
                        if (r == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                        else if (r == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (r == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = value(r * 2 + ii, j);
                        if constexpr (ACT == MICO_ACT_GELU) v = gelu4(v);
                        if constexpr (ACT == ACT_PAIR_TILED || ACT == ACT_PRE_TILED) {      // GELU' (PRE: the pre-activation) leaves from the registers: unit (i, j >> 1) of the wave's 16 KiB (p8p_aux_load)
                            f32x4 d;
                            if constexpr (ACT == ACT_PAIR_TILED) v = gelu_pair4(v, d);
                            else { d = v; v = gelu4(v); }
                            const s16x4 dp = pack4<T>(d[0], d[1], d[2], d[3]);
                            if ((j & 1) == 0) dlo = dp;
                            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, (s16x8){dlo[0], dlo[1], dlo[2], dlo[3], dp[0], dp[1], dp[2], dp[3]}), rsx,
                                                                        (int)(xbase + (unsigned)(((r * 2 + ii) * 2 + (j >> 1)) * 1024)), 0, 0);
                        }
                        if constexpr (ACT == ACT_MUL_TILED) {
                            const i32x4 x4 = xa[(r * 2 + ii) * 2 + (j >> 1)];
                            v *= unpack4<T>(__builtin_bit_cast(s16x4, (j & 1) ? (i32x2){x4[2], x4[3]} : (i32x2){x4[0], x4[1]}));
                        }
                        *(LDS_AS s16x4*)(wbuf + ii * 2048 + woff + (((j * 2 + wch) ^ wkey) << 4)) = pack4<T>(v[0], v[1], v[2], v[3]);
                    }
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int i = r * 2 + ps;
                    const int row = (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + rr;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        bstore(rsc, ldc2, row, u, *(LDS_AS const s16x8*)(wbuf + ps * 2048 + rr * 128 + (((u * 4 + q) ^ rkey) << 4)));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 d;
                    const f32x4 v = gelu_pair4(value(i, j), d);
                    const int off = woff + (((j * 2 + wch) ^ wkey) << 4);
                    *(LDS_AS s16x4*)(wbuf + off) = pack4<T>(v[0], v[1], v[2], v[3]);
                    *(LDS_AS s16x4*)(wbuf + 2048 + off) = pack4<T>(d[0], d[1], d[2], d[3]);
                }
                const int row = (i >> 2) * 128 + wm * 64 + (i & 3) * 16 + rr;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int off = rr * 128 + (((u * 4 + q) ^ rkey) << 4);
                    bstore(rsc, ldc2, row, u, *(LDS_AS const s16x8*)(wbuf + off));
                    bstore(rsx, ldx2, row, u, *(LDS_AS const s16x8*)(wbuf + 2048 + off));
                }
            }
        }
}

template <typename T, bool TB, int ACT>
__global__ __launch_bounds__(P8C::THREADS, 2) void gemm_p8p_kernel(const GemmArgs g) {
    // (Measured and not kept: ACT_RESID - the towers' fp32 residual-scatter forward - and MUL_AUX - fc2's dX - through the shared fp32-staged
    // epilogue in eight 16-row rounds of 4 KiB per wave: fc2 forward 1177 -> 1063, projection forward 830 -> 784, the GELU' multiply 1007 -> 880
    // TFLOP/s in situ - eight dependent load -> LDS -> store round trips cost more than the hidden prologue saves.  Those launches stay on the
    // one-tile kernel.)
    static_assert(ACT == ACT_LEAN || ACT == MICO_ACT_GELU || ACT == MICO_ACT_GELU_SAVE_DERIV || ACT == ACT_PAIR_TILED || ACT == ACT_MUL_TILED || ACT == ACT_PRE_TILED ||
                  ACT == ACT_GRAD_TILED, "the 16-bit staged epilogues");
    constexpr bool AUX_READ = ACT == ACT_MUL_TILED || ACT == ACT_GRAD_TILED;      // the tiled aux tensor is an INPUT of the epilogue (p8p_aux_load)
    constexpr int BM = P8C::BM, BN = P8C::BN, BK = P8C::BK, HALF = P8C::HALF, TILE = P8C::TILE;
    constexpr int NS = (ACT == MICO_ACT_GELU_SAVE_DERIV || ACT == ACT_PAIR_TILED || ACT == ACT_PRE_TILED || ACT == ACT_GRAD_TILED) ? 32 : 16;      // buffer stores per wave and tile
    // Counted waits never count on a STORE being outstanding: loads return in order among themselves, stores among themselves, but a store can
    // retire before an older load (the first version waited vmcnt(NS + 6) while the wanted half-tile was older than the previous tile's NS stores -
    // and read half-tiles that had not landed: 41 of 48 outputs wrong in tools/probes/epi16_check.py).  vmcnt(6) = "at most the three youngest
    // half-tiles outstanding" is right whatever the stores do; where they are still in flight it simply waits for them too.
    constexpr int VMW = 6, VM0 = 6;
    (void)NS;
    __shared__ __attribute__((aligned(16))) char smem[P8C::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int T_ = g.ktiles;
    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;

    // ---- tile state (changes per tile) ----
    int64_t m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t rsa, rsb, rsc, rsx;
    unsigned vra[2][2], vrb[2][2];
    const int rl = wave * 8 + (lane >> 3);
    {
        const unsigned va = (unsigned)(rl * lda_b) + (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < 2; ++it) vra[h][it] = va + (unsigned)((h * 128 + it * 64) * lda_b);
        if constexpr (!TB) {
            const unsigned vb = (unsigned)(rl * ldb_b) + (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int it = 0; it < 2; ++it) vrb[h][it] = vb + (unsigned)((h * 128 + it * 64) * ldb_b);
        } else {
            // reduction-major B: N % 128 == 0 (launcher), so a half-tile of 128 columns is inside the matrix or outside it as a whole - the
            // per-lane column test of the one-tile kernel becomes the wave-uniform `bhi` below and the offsets do not depend on the tile
            const int kr = wave * 4 + (lane >> 4), cg = (lane & 15) ^ key_tr(kr);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int it = 0; it < 2; ++it) vrb[h][it] = (unsigned)((kr + it * 32) * ldb_b) + (unsigned)(cg << 4) + (unsigned)(h * 256);
        }
    }
    bool bhi = true;      // TB: the tile's upper 128 columns exist
    auto locate = [&](int vb) {
        int bid = vb;
        {
            const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
        }
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        m0 = (int64_t)(first + in % gm) * BM;
        n0 = (int64_t)(in / gm) * BN;
        const char* a_base = g.A + m0 * lda_b;
        const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
        int64_t a_bytes = (g.M - m0) * lda_b;
        int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
        if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
        if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
        rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
        if constexpr (TB) bhi = g.N - n0 > 128;
    };
    // descriptors of the output tile (base at (m0, n0)): rows beyond M end the buffer
    auto out_desc = [&](char* base, int64_t ld2) {
        int64_t bytes = (g.M - m0) * ld2 - n0 * 2;
        if (bytes > 0xFFFFFF00ll) bytes = 0xFFFFFF00ll;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + m0 * ld2 + n0 * 2), 0, (int)bytes, 0x00020000);
    };
    const unsigned ld_dst = (unsigned)(wave * 1024);
    // CHECKED = false: K-tile t exists for sure (the prologue's K-tiles 0 and 1: the launcher requires >= 2) - no select on the offsets, i.e.
    // no per-tile-invariant copies of them for the register allocator to keep (it spilled twelve, reloaded behind vmcnt(0) in every tile)
    auto issue = [&](int t, auto wv, auto checked, unsigned ldd) {      // ldd: this wave's offset inside a half-tile image (ld_dst, or a per-tile copy of it)
        constexpr int W = decltype(wv)::value;
        constexpr bool isA = (W == 0 || W == 3);               // stream order of a tile: A-lo, B-hi, B-lo, A-hi (quadrant walk 1)
        constexpr int half = (W == 1 || W == 3) ? 1 : 0;
        const bool valid = (!decltype(checked)::value || t < T_) && (!TB || isA || half == 0 || bhi);
        const int ta_ = (!TB && g.a_wrap > 0 && t >= g.a_wrap) ? t - g.a_wrap : t;
        const unsigned soff = isA ? (unsigned)(ta_ * BK * 2) : (TB ? (unsigned)((int64_t)t * BK * ldb_b) : (unsigned)(t * BK * 2));
        LDS_AS char* dst = lds + (t & 1) * TILE + (isA ? 0 : 2 * HALF) + half * HALF + ldd;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned v = isA ? vra[half][it] : vrb[half][it];
            if (!valid) v = 0xFFFFFFF0u;
            lds_dma16_soff(isA ? rsa : rsb, (LDS_AS void*)(dst + it * 8192), v, soff);
        }
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using NC = std::false_type;

    const FragBase ab = frag_base<false, 256, BK>(wm * 64, lane);
    const FragBase bb = TB ? frag_base<true, 128, BK>(wn * 32, lane) : frag_base<false, 256, BK>(wn * 32, lane);
    s16x8 a0[4][2] = {}, a1[4][2] = {}, b0[2][2] = {}, b1[2][2] = {};
    auto rdA = [&](s16x8 (&d)[4][2], int boff, int blk) {
        LDS_AS const char* t = lds + boff + blk * HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i][kk] = read_frag_b<false, 256, BK>(t, kk ? ab.b1 : ab.b0, i);
    };
    auto rdB = [&](s16x8 (&d)[2][2], int boff, int blk) {
        LDS_AS const char* t = lds + boff + 2 * HALF + blk * HALF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) d[j][kk] = read_frag_b<TB, TB ? 128 : 256, BK>(t, kk ? bb.b1 : bb.b0, j);
    };
    f32x4 acc[8][4];
    auto mma = [&](const s16x8 (&a)[4][2], const s16x8 (&b)[2][2], auto iqv, auto jqv) {
        constexpr int IQ = decltype(iqv)::value, JQ = decltype(jqv)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[IQ * 4 + i][JQ * 2 + j] = T16<T>::mfma(b[j][kk], a[i][kk], acc[IQ * 4 + i][JQ * 2 + j]);
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto wait_bar = [&](auto vmv) {
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(vmv)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        fence();
    };
    auto bar = [&]() {
        fence();
        __builtin_amdgcn_s_barrier();
        fence();
    };
    using VMF = std::integral_constant<int, VM0>;
    using VMS = std::integral_constant<int, VMW>;
    auto issue_h = [&](int t, auto pv) {   // the half-tile issued in phase p of tile t: stream position 4 t + p + 6
        constexpr int IDX = decltype(pv)::value + 6;
        issue(t + IDX / 4, std::integral_constant<int, IDX % 4>{}, std::true_type{}, ld_dst);
    };
    auto ktile = [&](int t, int cur, auto vm) {
        const int nxt = cur ^ TILE;
        rdB(b0, cur, 0);  issue_h(t, I0{}); wait_bar(vm); mma(a0, b0, I0{}, I0{}); bar();
        rdA(a1, cur, 1);  issue_h(t, I1{}); wait_bar(vm); mma(a0, b1, I0{}, I1{}); bar();
        rdA(a0, nxt, 0);  issue_h(t, std::integral_constant<int, 2>{}); wait_bar(vm); mma(a1, b1, I1{}, I1{}); bar();
        rdB(b1, nxt, 1);  issue_h(t, std::integral_constant<int, 3>{}); wait_bar(VMS{}); mma(a1, b0, I1{}, I0{}); bar();
    };

    int vb = blockIdx.x;
    locate(vb);
    // the six requests of a tile's prologue: their twelve LDS destinations are constants of the kernel - as such hipcc keeps them in (vector) registers
    // across the whole tile loop, and where the epilogue is register-hungry (ACT_GRAD_TILED) spills them: a scratch reload + vmcnt(0) in front of EVERY
    // request, i.e. the next tile's requests issued one memory round trip apart.  Recomputed per tile from a laundered copy of the wave's offset instead.
    auto first6 = [&]() {
        unsigned ldd = ld_dst;
        asm volatile("" : "+s"(ldd));
        issue(0, W0{}, NC{}, ldd); issue(0, W1{}, NC{}, ldd); issue(0, W2{}, NC{}, ldd); issue(0, W3{}, NC{}, ldd); issue(1, W0{}, NC{}, ldd); issue(1, W1{}, NC{}, ldd);
    };
    first6();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // stream positions 0, 1, 2 of this tile have landed (at most the six requests of positions 3, 4, 5 are still out)
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM0) : "memory");
        __builtin_amdgcn_s_barrier();
        fence();
        rdA(a0, 0, 0); rdB(b1, 0, 1);
        fence();
        if (wm == 1) __builtin_amdgcn_s_barrier();   // the stagger: row group 1 runs half a phase behind row group 0
        fence();
        ktile(0, 0, VMF{});
        int cur = TILE;
        for (int t = 1; t < T_; ++t) {
            asm volatile("" : "+s"(cur));
            ktile(t, cur, VMS{});
            cur ^= TILE;
        }
        i32x4 xa[AUX_READ ? 16 : 1];
        if constexpr (AUX_READ) {   // the multiplier (GRAD: pre-activation) tile, requested before the closing wait (which then leaves these 16 loads out)
            fence();
            rsx = p8p_aux_tile_desc(g.e.aux_in, g, m0, n0);
            p8p_aux_load(xa, rsx, wave, lane);
            fence();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        // the zero-fill DMAs issued past the last tile must not land in what follows
        if constexpr (AUX_READ) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- the next tile's first six half-tiles, requested before this tile is stored ----
        rsc = out_desc(g.C, g.ldc * 2);
        __amdgpu_buffer_rsrc_t rso = rsc;
        if constexpr (ACT == MICO_ACT_GELU_SAVE_DERIV) rsx = out_desc((char*)g.e.aux_out, g.e.ldaux * 2);
        else if constexpr (ACT == ACT_PAIR_TILED || ACT == ACT_PRE_TILED) rsx = p8p_aux_tile_desc(g.e.aux_out, g, m0, n0);
        else if constexpr (ACT == ACT_GRAD_TILED) rso = out_desc((char*)g.e.aux_out, g.e.ldaux * 2);
        else if constexpr (!AUX_READ) rsx = rsc;
        const int64_t n0e = n0;
        vb += gridDim.x;
        const bool more = vb < g.ntiles;
        if (more) {
            locate(vb);
            first6();
        }
        // ---- epilogue: staging in the two ring slots the requests above do not use (second tile buffer: A-hi, B-lo), 4 KiB per wave ----
        p8p_epilogue<T, ACT>(g, acc, lds + TILE + HALF + wave * 4096, rsc, rsx, rso, n0e, wm, wn, lane, xa, more);
        if (!more) return;
    }
}

// ======================================================================================================================
// MX-fp8 GEMM (BASELINE.json configs[4]: "fp8 MFMA"): C = epilogue(A B^T) with A [M, K] and B [N, K] in OCP e4m3 and one E8M0 scale per
// 32 consecutive k (the OCP microscaling format), on v_mfma_scale_f32_16x16x128_f8f6f4 - the only fp8 MFMA on gfx950 that runs above
// the bf16 rate (2x; the unscaled fp8 forms run AT the bf16 rate).
// Operand layout of that instruction, measured (tools/probes/mx_layout.hip): lane (r = l % 16, g = l / 16) supplies row r; its eight
// operand registers hold k = 16 g .. 16 g + 15 (registers 0-3) and k = 64 + 16 g .. + 15 (registers 4-7) - two 16-byte chunks, chunk g and
// chunk 4 + g of the row's 128 bytes, i.e. exactly the two k-step reads of the 16-bit kernels' 128-byte-row LDS image - and the scale
// operand of lane (r, b) is the scale of row r's block b (k = 32 b .. 32 b + 31).  So the tile images, the DMA staging, the swizzle, the
// fragment addressing and the epilogues are those of the 16-bit kernels with "64 16-bit elements" read as "128 bytes"; only the MFMA
// differs (one scaled instruction where the 16-bit kernels issue two) and a 1 KiB scale image per operand rides along with each tile.
// Scales in memory: uint32 [K / 128][rows] - the four E8M0 bytes of one row's 128-k tile in one word, K-tile-major so that a tile's
// scales are one contiguous KiB.  Schedule: 256x256x128 tile, 8 waves, 2 stages, one barrier pair per K-tile (first version).
// ======================================================================================================================
struct Mx8 {
    static constexpr int BM = 256, BN = 256, BKB = 128, THREADS = 512, MT = 8;
    static constexpr int A_BYTES = 256 * BKB, S_BYTES = 256 * 4, STAGE_BYTES = 2 * A_BYTES + 2 * S_BYTES, LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int NDMA = A_BYTES / 16 / THREADS;   // 4 DMA instructions per thread per operand tile
};
typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Mx8Args {
    GemmArgs g;               // M, N, K (K in elements = bytes), A / B / C, lda / ldb (bytes), tiles, epilogue
    const unsigned* sa;       // [K / 128][M]
    const unsigned* sb;       // [K / 128][N]
};

template <typename T, int ACT>
__global__ __launch_bounds__(Mx8::THREADS, 2) void gemm_mx8_kernel(const Mx8Args a) {
    const GemmArgs& g = a.g;
    constexpr int BM = Mx8::BM, BN = Mx8::BN, THREADS = Mx8::THREADS, ND = Mx8::NDMA;
    __shared__ __attribute__((aligned(16))) char smem[Mx8::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wrow = wm * 128, wcol = wn * 64;
    int bid = blockIdx.x;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int T_ = g.ktiles;
    // the tile images are those of a 16-bit [rows][64] operand: leading dimensions and k offsets in BYTES
    const char* a_base = g.A + m0 * g.lda;
    const char* b_base = g.B + n0 * g.ldb;
    int64_t a_bytes = (g.M - m0) * g.lda, b_bytes = (g.N - n0) * g.ldb;
    if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
    if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
    // scale words: one descriptor per operand over the whole [K / 128][rows] array, bounds = its end (rows past M / N of the last
    // tile row read the next K-tile's words or zero-fill - they scale operand rows that are zero-filled themselves)
    __amdgpu_buffer_rsrc_t rssa = __builtin_amdgcn_make_buffer_rsrc((void*)a.sa, 0, (int)min((int64_t)T_ * g.M * 4, (int64_t)0x7FFFFF00), 0x00020000);
    __amdgpu_buffer_rsrc_t rssb = __builtin_amdgcn_make_buffer_rsrc((void*)a.sb, 0, (int)min((int64_t)T_ * g.N * 4, (int64_t)0x7FFFFF00), 0x00020000);
    unsigned voa[ND], vob[ND];
    dma_offsets<false, BM, THREADS, 64, ND>(voa, wave, lane, g.lda, g.M - m0);
    dma_offsets<false, BN, THREADS, 64, ND>(vob, wave, lane, g.ldb, g.N - n0);
    const FragBase ab = frag_base<false, BM, 64>(wrow, lane), bb = frag_base<false, BN, 64>(wcol, lane);
    auto stage = [&](int kt, int bo) {
        dma_issue<THREADS, ND>(rsa, lds + bo, wave, voa, (unsigned)kt * 128u);
        dma_issue<THREADS, ND>(rsb, lds + bo + Mx8::A_BYTES, wave, vob, (unsigned)kt * 128u);
        if (wave == 0)        // 256 scale words of the A tile rows: 64 lanes x 16 bytes
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rssa, (LDS_AS void*)(lds + bo + 2 * Mx8::A_BYTES), 16,
                                                     (unsigned)(((int64_t)kt * g.M + m0) * 4 + lane * 16), 0, 0, 0);
        else if (wave == 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rssb, (LDS_AS void*)(lds + bo + 2 * Mx8::A_BYTES + Mx8::S_BYTES), 16,
                                                     (unsigned)(((int64_t)kt * g.N + n0) * 4 + lane * 16), 0, 0, 0);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int p = lane & 15, gq = lane >> 4;
    int bo = 0;
    stage(0, 0);
    for (int t = 0; t < T_; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage `bo` landed (explicit: the operand DMA may be invisible to the compiler) ...
        __syncthreads();   // ... and is published; the other stage is no longer being read
        if (t + 1 < T_) stage(t + 1, bo ^ Mx8::STAGE_BYTES);
        LDS_AS const char* ta = lds + bo;
        LDS_AS const char* tb = ta + Mx8::A_BYTES;
        LDS_AS const unsigned* sca = (LDS_AS const unsigned*)(ta + 2 * Mx8::A_BYTES);
        LDS_AS const unsigned* scb = sca + 256;
        i32x8 fb[4];
        int sb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const s16x8 lo = *(LDS_AS const s16x8*)(tb + bb.b0 + j * 2048), hi = *(LDS_AS const s16x8*)(tb + bb.b1 + j * 2048);
            const u32x4 l4 = __builtin_bit_cast(u32x4, lo), h4 = __builtin_bit_cast(u32x4, hi);
            fb[j] = (i32x8){(int)l4[0], (int)l4[1], (int)l4[2], (int)l4[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
            sb[j] = (int)((scb[wcol + j * 16 + p] >> (8 * gq)) & 0xFFu);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // the A row tiles in two halves: 32 fragment registers live instead of 64
            i32x8 fa[4];
            int sa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = h * 4 + i;
                const s16x8 lo = *(LDS_AS const s16x8*)(ta + ab.b0 + ii * 2048), hi = *(LDS_AS const s16x8*)(ta + ab.b1 + ii * 2048);
                const u32x4 l4 = __builtin_bit_cast(u32x4, lo), h4 = __builtin_bit_cast(u32x4, hi);
                fa[i] = (i32x8){(int)l4[0], (int)l4[1], (int)l4[2], (int)l4[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
                sa[i] = (int)((sca[wrow + ii * 16 + p] >> (8 * gq)) & 0xFFu);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)   // operands swapped (D^T = B A^T) like the 16-bit kernels: a lane owns 4 consecutive columns
                    acc[h * 4 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[h * 4 + i][j], 0, 0, 0, sb[j], 0, sa[i]);
        }
        bo ^= Mx8::STAGE_BYTES;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
        gemm_epilogue_block<T, 4, ACT>(g, &acc[h * 4], lds + wave * 16384, m0 + wrow + h * 64, n0 + wcol, lane);
}

// ======================================================================================================================
// MX-fp8 GEMM on the 8-phase schedule (round 4).  A 128-deep K-tile of e4m3 bytes has exactly the LDS image of a 64-deep K-tile of
// 16-bit elements, so the half-tile DMA stream, the slots, the waits, the quadrant walk and the fragment reads are those of gemm_p8_kernel
// (walk 1, three half-tiles in flight); a phase issues 8 scaled MFMAs (16x16x128: 32 cycles each) where the 16-bit kernel issues 16 of 16
// cycles - the same 256 matrix-pipe cycles per phase for twice the flops.  The E8M0 block scales (one uint32 per row and K-tile: four
// 32-blocks) travel as 2 KiB per K-tile through a 4-slot LDS ring behind the operand buffers: every wave issues ONE 4-byte-per-lane
// LDS-DMA per tile (waves 0-3: 64 A rows each, 4-7: B) in phase 0, two tiles ahead - that instruction is the 7th in flight at the waits
// of phases 0-2 and has left the window of three issuing phases at phase 3's: vmcnt 7 / 7 / 7 / 6.  A lane reads its fragment row's word
// next to the fragment and extracts the byte of its k-group (l / 16).  Replaces the 2-stage first version (one vmcnt(0) + __syncthreads
// per K-tile, 1507 TFLOP/s in situ) for the lean / residual / MLP-pair launches.
// ======================================================================================================================
__device__ __forceinline__ void lds_dma4_soff(__amdgpu_buffer_rsrc_t rs, LDS_AS void* lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs), "s"(soff) : "memory");
}

template <typename T, int ACT>
__global__ __launch_bounds__(P8C::THREADS, 2) void gemm_p8mx_kernel(const Mx8Args a) {
    const GemmArgs& g = a.g;
    constexpr int BM = P8C::BM, BN = P8C::BN, HALF = P8C::HALF, TILE = P8C::TILE, SCB = 2 * P8C::TILE;   // SCB: the scale ring (4 x 2 KiB)
    __shared__ __attribute__((aligned(16))) char smem[P8C::LDS_BYTES + 4 * 2048];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int wm = wave >> 2, wn = wave & 3;
    const int T_ = g.ktiles;
    // leading dimensions and k offsets in BYTES (one byte per element)
    const char* a_base = g.A + m0 * g.lda;
    const char* b_base = g.B + n0 * g.ldb;
    int64_t a_bytes = (g.M - m0) * g.lda, b_bytes = (g.N - n0) * g.ldb;
    if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
    if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
    // scale words: one descriptor per operand over the whole [K / 128][rows] array (rows past M / N of the last tile row read the next
    // K-tile's words or zero-fill - they scale operand rows that are zero-filled themselves)
    __amdgpu_buffer_rsrc_t rss = wave < 4
        ? __builtin_amdgcn_make_buffer_rsrc((void*)a.sa, 0, (int)min((int64_t)T_ * g.M * 4, (int64_t)0x7FFFFF00), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)a.sb, 0, (int)min((int64_t)T_ * g.N * 4, (int64_t)0x7FFFFF00), 0x00020000);
    const unsigned vsc = (unsigned)(((wave < 4 ? m0 : n0) + (wave & 3) * 64 + lane) * 4);
    const unsigned sc_stride = (unsigned)((wave < 4 ? g.M : g.N) * 4);
    const unsigned sc_dst = (unsigned)(SCB + (wave < 4 ? 0 : 1024) + (wave & 3) * 256);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int rl = wave * 8 + (lane >> 3);
    unsigned vra[2][2], vrb[2][2];   // [half][it]
    {
        const unsigned sw = (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                vra[h][it] = (unsigned)((rl + h * 128 + it * 64) * g.lda) + sw;
                vrb[h][it] = (unsigned)((rl + h * 128 + it * 64) * g.ldb) + sw;
            }
    }
    const unsigned ld_dst = (unsigned)(wave * 1024);
    auto issue = [&](int t, auto wv) {   // stream position W of tile t: 0 A-lo, 1 B-hi, 2 B-lo, 3 A-hi
        constexpr int W = decltype(wv)::value;
        constexpr bool isA = (W == 0 || W == 3);
        constexpr int half = (W == 1 || W == 3) ? 1 : 0;
        const bool valid = t < T_;
        const unsigned soff = (unsigned)(t * 128);
        LDS_AS char* dst = lds + (t & 1) * TILE + (isA ? 0 : 2 * HALF) + half * HALF + ld_dst;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned v = isA ? vra[half][it] : vrb[half][it];
            if (!valid) v = 0xFFFFFFF0u;
            lds_dma16_soff(isA ? rsa : rsb, (LDS_AS void*)(dst + it * 8192), v, soff);
        }
    };
    auto issue_scales = [&](int t) {
        lds_dma4_soff(rss, (LDS_AS void*)(lds + sc_dst + (t & 3) * 2048), t < T_ ? vsc : 0xFFFFFFF0u, t < T_ ? (unsigned)t * sc_stride : 0u);
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;

    const FragBase ab = frag_base<false, 256, 64>(wm * 64, lane);
    const FragBase bb = frag_base<false, 256, 64>(wn * 32, lane);
    const int p = lane & 15, sh = (lane >> 4) * 8;
    i32x8 a0[4], a1[4], b0[2], b1[2];
    int sa0 = 0, sa1 = 0, sb0 = 0, sb1 = 0;   // the scale bytes of a fragment set packed into one register: byte i = tile i (the MFMA's op_sel picks it)
    auto frag = [&](LDS_AS const char* t, const FragBase& fbs, int i) {
        const u32x4 l4 = __builtin_bit_cast(u32x4, *(LDS_AS const s16x8*)(t + fbs.b0 + i * 2048));
        const u32x4 h4 = __builtin_bit_cast(u32x4, *(LDS_AS const s16x8*)(t + fbs.b1 + i * 2048));
        return (i32x8){(int)l4[0], (int)l4[1], (int)l4[2], (int)l4[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
    };
    auto rdA = [&](i32x8 (&d)[4], int& s, int boff, int blk, int sslot) {
        LDS_AS const char* t = lds + boff + blk * HALF;
        LDS_AS const unsigned* sc = (LDS_AS const unsigned*)(lds + SCB + sslot) + blk * 128 + wm * 64 + p;
        unsigned pk = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = frag(t, ab, i);
            pk |= ((sc[i * 16] >> sh) & 0xFFu) << (8 * i);
        }
        s = (int)pk;
    };
    auto rdB = [&](i32x8 (&d)[2], int& s, int boff, int blk, int sslot) {
        LDS_AS const char* t = lds + boff + 2 * HALF + blk * HALF;
        LDS_AS const unsigned* sc = (LDS_AS const unsigned*)(lds + SCB + sslot + 1024) + blk * 128 + wn * 32 + p;
        unsigned pk = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            d[j] = frag(t, bb, j);
            pk |= ((sc[j * 16] >> sh) & 0xFFu) << (8 * j);
        }
        s = (int)pk;
    };
    auto mma = [&](const i32x8 (&fa)[4], int sa, const i32x8 (&fb)[2], int sb, auto iqv, auto jqv) {
        constexpr int IQ = decltype(iqv)::value, JQ = decltype(jqv)::value;
        // operands swapped (D^T = B A^T) like the 16-bit kernels: a lane owns 4 consecutive columns; op_sel = the scale byte of the tile
#define P8MX_MMA(I, J) acc[IQ * 4 + I][JQ * 2 + J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[J], fa[I], acc[IQ * 4 + I][JQ * 2 + J], 0, 0, J, sb, I, sa)
        P8MX_MMA(0, 0); P8MX_MMA(0, 1); P8MX_MMA(1, 0); P8MX_MMA(1, 1); P8MX_MMA(2, 0); P8MX_MMA(2, 1); P8MX_MMA(3, 0); P8MX_MMA(3, 1);
#undef P8MX_MMA
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto wait_bar = [&](auto nv) {
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nv)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        fence();
    };
    auto bar = [&]() {
        fence();
        __builtin_amdgcn_s_barrier();
        fence();
    };
    using N6 = std::integral_constant<int, 3 * P8C::DMA_PER_HALF>;       // three half-tiles in flight
    using N7 = std::integral_constant<int, 3 * P8C::DMA_PER_HALF + 1>;   // ... and one scale-word load behind them
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // prologue: the scale words of tiles 0 and 1, half-tiles 0..5 of the stream; a0, b1 of tile 0
    issue_scales(0); issue_scales(1);
    issue(0, W0{}); issue(0, W1{}); issue(0, W2{}); issue(0, W3{}); issue(1, W0{}); issue(1, W1{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N6::value) : "memory");
    __builtin_amdgcn_s_barrier();
    fence();
    rdA(a0, sa0, 0, 0, 0);
    rdB(b1, sb1, 0, 1, 0);
    fence();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // the stagger (see gemm_p8_kernel)
    fence();
    int cur = 0;
    for (int t = 0; t < T_; ++t) {
        asm volatile("" : "+s"(cur));
        const int nxt = cur ^ TILE;
        const int sc_cur = (t & 3) * 2048, sc_nxt = ((t + 1) & 3) * 2048;
        // p0: half-tile h = 4 t + 6 (B-lo of t + 1) and the scales of tile t + 2
        rdB(b0, sb0, cur, 0, sc_cur);  issue(t + 1, W2{}); issue_scales(t + 2); wait_bar(N7{}); mma(a0, sa0, b0, sb0, I0{}, I0{}); bar();
        rdA(a1, sa1, cur, 1, sc_cur);  issue(t + 1, W3{}); wait_bar(N7{}); mma(a0, sa0, b1, sb1, I0{}, I1{}); bar();
        rdA(a0, sa0, nxt, 0, sc_nxt);  issue(t + 2, W0{}); wait_bar(N7{}); mma(a1, sa1, b1, sb1, I1{}, I1{}); bar();
        rdB(b1, sb1, nxt, 1, sc_nxt);  issue(t + 2, W1{}); wait_bar(N6{}); mma(a1, sa1, b0, sb0, I1{}, I0{}); bar();
        cur = nxt;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (zero-fill DMAs past the last tile: see gemm_p8_kernel)
    __syncthreads();
    if constexpr (ACT == ACT_LEAN || ACT == MICO_ACT_GELU_SAVE_DERIV || ACT == MICO_ACT_MUL_AUX || ACT == MICO_ACT_GELU) {
        if (MICO_P8_FAST16 && g.fast16) {
            // (round 5: the 16-bit staged epilogue of the 16-bit kernels - same accumulator layout; at K = 1408 an fp8 tile is 11 K-tiles next to the
            // same fixed cost, so the 2.5 us it saves weigh twice as much here)
            if constexpr (MICO_P8_EPI16 != 0 && ACT != MICO_ACT_MUL_AUX) p8_epilogue_fast16_h<T, ACT>(g, acc, lds + wave * 16384, m0, n0, wm, wn, lane);
            else p8_epilogue_fast16<T, ACT>(g, acc, lds + wave * 16384, m0, n0, wm, wn, lane);
            return;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        gemm_epilogue_block<T, 4, ACT, false, 96>(g, &acc[h * 4], lds + wave * 16384, m0 + h * 128 + wm * 64, n0 + wn * 32, lane);
}

// The persistent form of gemm_p8mx_kernel (round 5; see gemm_p8p_kernel): 256 workgroups walk the XCD-contiguous tile list, the scale words of
// K-tiles 0 / 1 and the first six half-tiles of tile i + 1 are requested before tile i's epilogue (16-bit staging in the two free ring slots,
// buffer stores).  An fp8 tile at K = 1408 is 11 K-tiles (~15 us) next to the same ~8 us of prologue + epilogue + dispatch gap the 16-bit tiles
// have next to 41 us: the fixed cost is what the format's 2x runs into.  Lean and GELU-pair launches with 16-bit outputs, N % 8 == 0.
template <typename T, int ACT>
__global__ __launch_bounds__(P8C::THREADS, 2) void gemm_p8pmx_kernel(const Mx8Args a) {
    static_assert(ACT == ACT_LEAN || ACT == MICO_ACT_GELU_SAVE_DERIV, "the 16-bit staged epilogues of the fp8 launches");
    const GemmArgs& g = a.g;
    constexpr int BM = P8C::BM, BN = P8C::BN, HALF = P8C::HALF, TILE = P8C::TILE, SCB = 2 * P8C::TILE;
    __shared__ __attribute__((aligned(16))) char smem[P8C::LDS_BYTES + 4 * 2048];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int T_ = g.ktiles;
    int64_t m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t rsa, rsb, rsc, rsx;
    __amdgpu_buffer_rsrc_t rss = wave < 4
        ? __builtin_amdgcn_make_buffer_rsrc((void*)a.sa, 0, (int)min((int64_t)T_ * g.M * 4, (int64_t)0x7FFFFF00), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)a.sb, 0, (int)min((int64_t)T_ * g.N * 4, (int64_t)0x7FFFFF00), 0x00020000);
    unsigned vsc = 0;
    const unsigned sc_stride = (unsigned)((wave < 4 ? g.M : g.N) * 4);
    const unsigned sc_dst = (unsigned)(SCB + (wave < 4 ? 0 : 1024) + (wave & 3) * 256);
    const int rl = wave * 8 + (lane >> 3);
    unsigned vra[2][2], vrb[2][2];
    {
        const unsigned sw = (unsigned)(((lane & 7) ^ key_kc(rl)) << 4);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                vra[h][it] = (unsigned)((rl + h * 128 + it * 64) * g.lda) + sw;
                vrb[h][it] = (unsigned)((rl + h * 128 + it * 64) * g.ldb) + sw;
            }
    }
    auto locate = [&](int vb) {
        int bid = vb;
        {
            const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
        }
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        m0 = (int64_t)(first + in % gm) * BM;
        n0 = (int64_t)(in / gm) * BN;
        int64_t a_bytes = (g.M - m0) * g.lda, b_bytes = (g.N - n0) * g.ldb;
        if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
        if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
        rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + m0 * g.lda), 0, (int)a_bytes, 0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + n0 * g.ldb), 0, (int)b_bytes, 0x00020000);
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        vsc = (unsigned)(((wave < 4 ? m0 : n0) + (wave & 3) * 64 + l2) * 4);
    };
    auto out_desc = [&](char* base, int64_t ld2) {
        int64_t bytes = (g.M - m0) * ld2 - n0 * 2;
        if (bytes > 0xFFFFFF00ll) bytes = 0xFFFFFF00ll;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + m0 * ld2 + n0 * 2), 0, (int)bytes, 0x00020000);
    };
    const unsigned ld_dst = (unsigned)(wave * 1024);
    auto issue = [&](int t, auto wv, auto checked) {   // stream position W of tile t: 0 A-lo, 1 B-hi, 2 B-lo, 3 A-hi
        constexpr int W = decltype(wv)::value;
        constexpr bool isA = (W == 0 || W == 3);
        constexpr int half = (W == 1 || W == 3) ? 1 : 0;
        const bool valid = !decltype(checked)::value || t < T_;
        const unsigned soff = (unsigned)(t * 128);
        LDS_AS char* dst = lds + (t & 1) * TILE + (isA ? 0 : 2 * HALF) + half * HALF + ld_dst;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            unsigned v = isA ? vra[half][it] : vrb[half][it];
            if (!valid) v = 0xFFFFFFF0u;
            lds_dma16_soff(isA ? rsa : rsb, (LDS_AS void*)(dst + it * 8192), v, soff);
        }
    };
    auto issue_scales = [&](int t) {
        lds_dma4_soff(rss, (LDS_AS void*)(lds + sc_dst + (t & 3) * 2048), t < T_ ? vsc : 0xFFFFFFF0u, t < T_ ? (unsigned)t * sc_stride : 0u);
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;
    using W3 = std::integral_constant<int, 3>;
    using NC = std::false_type;
    using CK = std::true_type;
    const FragBase ab = frag_base<false, 256, 64>(wm * 64, lane);
    const FragBase bb = frag_base<false, 256, 64>(wn * 32, lane);
    const int p = lane & 15, sh = (lane >> 4) * 8;
    i32x8 a0[4], a1[4], b0[2], b1[2];
    int sa0 = 0, sa1 = 0, sb0 = 0, sb1 = 0;
    auto frag = [&](LDS_AS const char* t, const FragBase& fbs, int i) {
        const u32x4 l4 = __builtin_bit_cast(u32x4, *(LDS_AS const s16x8*)(t + fbs.b0 + i * 2048));
        const u32x4 h4 = __builtin_bit_cast(u32x4, *(LDS_AS const s16x8*)(t + fbs.b1 + i * 2048));
        return (i32x8){(int)l4[0], (int)l4[1], (int)l4[2], (int)l4[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
    };
    auto rdA = [&](i32x8 (&d)[4], int& s, int boff, int blk, int sslot) {
        LDS_AS const char* t = lds + boff + blk * HALF;
        LDS_AS const unsigned* sc = (LDS_AS const unsigned*)(lds + SCB + sslot) + blk * 128 + wm * 64 + p;
        unsigned pk = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[i] = frag(t, ab, i);
            pk |= ((sc[i * 16] >> sh) & 0xFFu) << (8 * i);
        }
        s = (int)pk;
    };
    auto rdB = [&](i32x8 (&d)[2], int& s, int boff, int blk, int sslot) {
        LDS_AS const char* t = lds + boff + 2 * HALF + blk * HALF;
        LDS_AS const unsigned* sc = (LDS_AS const unsigned*)(lds + SCB + sslot + 1024) + blk * 128 + wn * 32 + p;
        unsigned pk = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            d[j] = frag(t, bb, j);
            pk |= ((sc[j * 16] >> sh) & 0xFFu) << (8 * j);
        }
        s = (int)pk;
    };
    f32x4 acc[8][4];
    auto mma = [&](const i32x8 (&fa)[4], int sa, const i32x8 (&fb)[2], int sb, auto iqv, auto jqv) {
        constexpr int IQ = decltype(iqv)::value, JQ = decltype(jqv)::value;
#define P8MX_MMA(I, J) acc[IQ * 4 + I][JQ * 2 + J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[J], fa[I], acc[IQ * 4 + I][JQ * 2 + J], 0, 0, J, sb, I, sa)
        P8MX_MMA(0, 0); P8MX_MMA(0, 1); P8MX_MMA(1, 0); P8MX_MMA(1, 1); P8MX_MMA(2, 0); P8MX_MMA(2, 1); P8MX_MMA(3, 0); P8MX_MMA(3, 1);
#undef P8MX_MMA
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto wait_bar = [&](auto nv) {
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nv)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        fence();
    };
    auto bar = [&]() {
        fence();
        __builtin_amdgcn_s_barrier();
        fence();
    };
    using N6 = std::integral_constant<int, 3 * P8C::DMA_PER_HALF>;
    using N7 = std::integral_constant<int, 3 * P8C::DMA_PER_HALF + 1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto request_first = [&]() {      // what a tile finds in flight when it starts: the scale words of K-tiles 0 and 1, stream positions 0 .. 5
        issue_scales(0); issue_scales(1);
        issue(0, W0{}, NC{}); issue(0, W1{}, NC{}); issue(0, W2{}, NC{}); issue(0, W3{}, NC{}); issue(1, W0{}, NC{}); issue(1, W1{}, NC{});
    };
    int vb = blockIdx.x;
    locate(vb);
    request_first();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fence();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N6::value) : "memory");      // (never counts on a store being outstanding: see gemm_p8p_kernel)
        __builtin_amdgcn_s_barrier();
        fence();
        rdA(a0, sa0, 0, 0, 0);
        rdB(b1, sb1, 0, 1, 0);
        fence();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        fence();
        int cur = 0;
        for (int t = 0; t < T_; ++t) {
            asm volatile("" : "+s"(cur));
            const int nxt = cur ^ TILE;
            const int sc_cur = (t & 3) * 2048, sc_nxt = ((t + 1) & 3) * 2048;
            rdB(b0, sb0, cur, 0, sc_cur);  issue(t + 1, W2{}, CK{}); issue_scales(t + 2); wait_bar(N7{}); mma(a0, sa0, b0, sb0, I0{}, I0{}); bar();
            rdA(a1, sa1, cur, 1, sc_cur);  issue(t + 1, W3{}, CK{}); wait_bar(N7{}); mma(a0, sa0, b1, sb1, I0{}, I1{}); bar();
            rdA(a0, sa0, nxt, 0, sc_nxt);  issue(t + 2, W0{}, CK{}); wait_bar(N7{}); mma(a1, sa1, b1, sb1, I1{}, I1{}); bar();
            rdB(b1, sb1, nxt, 1, sc_nxt);  issue(t + 2, W1{}, CK{}); wait_bar(N6{}); mma(a1, sa1, b0, sb0, I1{}, I0{}); bar();
            cur = nxt;
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        rsc = out_desc(g.C, g.ldc * 2);
        if constexpr (ACT == MICO_ACT_GELU_SAVE_DERIV) rsx = out_desc((char*)g.e.aux_out, g.e.ldaux * 2);
        else rsx = rsc;
        const int64_t n0e = n0;
        vb += gridDim.x;
        const bool more = vb < g.ntiles;
        if (more) {
            locate(vb);
            request_first();
        }
        p8p_epilogue<T, ACT>(g, acc, lds + TILE + HALF + wave * 4096, rsc, rsx, rsc, n0e, wm, wn, lane);
        if (!more) return;
    }
}

// 16-bit [rows, cols] -> e4m3 [rows, cols] + E8M0 block scales (one per 32 consecutive columns, packed 4 per uint32, K-tile-major).
// scale = 2^e with the smallest e such that amax / 2^e <= 448 (the largest e4m3 magnitude): nothing saturates, at most one binade of the
// element format's range is given up.  A wave covers 512 columns of one row per pass (8 per lane: one 16-byte load), a 32-block is 4 lanes.
template <typename T>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const T* __restrict__ x, int64_t ld, int64_t rows, int cols, unsigned char* __restrict__ q,
                                                        int64_t ldq, unsigned* __restrict__ sc, float pre_scale) {
    const int lane = threadIdx.x & 63;
    const int chunks = (cols + 511) / 512;
    const int64_t units = rows * chunks;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < units; u += (int64_t)gridDim.x * 4) {
        const int64_t row = u / chunks;
        const int c0 = (int)(u - row * chunks) * 512 + lane * 8;
        const bool live = c0 < cols;     // cols % 128 == 0: a lane's 8 columns and its 16-lane scale word are all in or all out
        float v[8];
        if (live) unpack8<T>(*(const s16x8*)(x + row * ld + c0), v);
        else
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] *= pre_scale; amax = fmaxf(amax, fabsf(v[k])); }
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        // e = ceil(log2(amax / 448)) from the float's bits (amax / 448 = m * 2^ex, 1 <= m < 2: e = ex + (m > 1))
        const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
        int e = (int)((bits >> 23) & 0xFF) - 127 + ((bits & 0x7FFFFF) ? 1 : 0);
        e = amax > 0.f ? max(-127, min(127, e)) : -127;
        const float inv = __uint_as_float((unsigned)(127 - e) << 23);      // 2^-e (e = -127: 2^254 is not a float; such blocks are all-zero)
        unsigned w0 = 0, w1 = 0;
        if (amax > 0.f && e > -127) {
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, w1, true);
        }
        if (live) *(u32x2*)(q + row * ldq + c0) = (u32x2){w0, w1};
        // the four block scales of a 128-column tile sit in lanes 0, 4, 8, 12 of each 16-lane group
        const unsigned sbyte = (unsigned)(e + 127);
        const int base = lane & ~15;
        const unsigned word = (__shfl(sbyte, base, 64) & 0xFF) | ((__shfl(sbyte, base + 4, 64) & 0xFF) << 8) |
                              ((__shfl(sbyte, base + 8, 64) & 0xFF) << 16) | ((__shfl(sbyte, base + 12, 64) & 0xFF) << 24);
        if (live && (lane & 15) == 0) sc[(int64_t)(c0 >> 7) * rows + row] = word;
    }
}

// ======================================================================================================================
// Producer / consumer kernel for the large GEMMs.
// Probes (tools/probes/producer_consumer.hip, dma_mfma_overlap.hip) showed what bounds the 8-wave kernel above: a wave that
// issues LDS-DMA instructions stalls on VMEM issue back-pressure (the memory side returns a 256x256x64 tile's 64 KiB in
// 0.5-0.65 us) and cannot issue MFMAs meanwhile, so fill time adds to compute time instead of hiding behind it; the same DMA
// issued by waves that do nothing else costs the MFMA waves ~5 %.  Hence 12 waves per workgroup: waves 0-7 are CONSUMERS
// (2x4, ds_read + MFMA only, the same two-group ping-pong as above), waves 8-11 are PRODUCERS (one per SIMD; all DMA, counted
// vmcnt, 3 K-tiles in flight).  Three waves per SIMD leave 168 VGPRs per wave, so the block tile is 192x256: a consumer owns
// 96x64 (24 accumulator tiles = 96 VGPRs, 10 fragments).  The LDS images keep the 256-row layout and swizzles of the kernel
// above - only rows 0-191 of the A image are filled - and a consumer's 96 rows are rows wm*64..+63 and 128+wm*32..+31, so both
// operand orientations reuse the same fragment addressing.
// ======================================================================================================================
// K-tile depth: 32 (4-stage ring) when both operands are reduction-major (their DMA rows are 512 bytes anyway); 64 (2 stages)
// when an operand is k-contiguous, so that its DMA rows are whole 128-byte lines instead of 64-byte halves (fill ceiling 95-128
// instead of 58-77 GB/s per CU, tools/probes/dma_fill.hip) - with producers the shallower ring costs the consumers nothing.
// Column sums of one reduction-major A tile image ([32 k-rows][256 columns], key_tr swizzle) for the producer waves of the weight-gradient
// kernel: lane = column.  The bias gradient db = sum_rows dy is the column sum of the very dy panel the dW GEMM stages for its A operand,
// so the workgroups of tile column 0 take it along (mico_gemm_epilogue::colsum_out) instead of a separate pass over dy (2.5 % of the
// step as `colsum_kernel`).  Inline-asm reads with their own wait: the compiler would put `vmcnt(0)` in front of an LDS read it cannot
// tell from the DMA destinations, draining the producers' queue (tools/probes/README.md).
template <typename T>
__device__ __forceinline__ void tile_colsum32(float (&cs)[8], unsigned a_even, unsigned a_odd, int rg) {
    // the 256 producer lanes = 32 chunk columns (8 matrix columns each) x 8 row groups: lane (lc, rg) reads its chunk of rows rg, rg + 8,
    // rg + 16, rg + 24 - four 16-byte reads per K-tile instead of 32 two-byte ones (those took the dW kernel from 792 to 726 TFLOP/s) -
    // and keeps 8 running sums; the row groups are folded when the kernel ends.  key_tr(r) depends on r & 3 (= rg & 3, fixed per lane)
    // and on bit 3 of r, i.e. on the parity of j in r = rg + 8 j: two addresses per lane.
    u32x4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:12288\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a_even), "v"(a_odd) : "memory");
    (void)rg;
    const u32x4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float f[8];
        unpack8<T>(__builtin_bit_cast(s16x8, vv[j]), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] += f[k];
    }
}

template <int BK_> struct Wide {
    static constexpr int BM = 192, BN = 256, BK = BK_, STAGES = BK_ == 32 ? 4 : 2, MT = 6, KSTEPS = BK_ / 32;
    static constexpr int CWAVES = 8, PWAVES = 4, THREADS = (CWAVES + PWAVES) * 64, PTHREADS = PWAVES * 64;
    static constexpr int A_BYTES = 256 * BK * 2, STAGE_BYTES = 2 * A_BYTES, LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int NDMA = A_BYTES / 16 / PTHREADS;   // DMA instructions per producer thread per 256-row operand image
};

#if MICO_GEMM_ABLATE == 10   // ablation: no DMA in the steady state AND no workgroup barriers (wrong results; what the barriers cost)
#define PC_BARRIER() do {} while (0)
#else
#define PC_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef MICO_PC_MIDBAR
#define MICO_PC_MIDBAR 0
#endif
#if MICO_PC_MIDBAR   // the mid-tile barrier only aligns the two consumer groups' phases (no data hazard depends on it)
#define PC_MIDBAR() PC_BARRIER()
#else
#define PC_MIDBAR() do {} while (0)
#endif
template <typename T, bool TA, bool TB, int BKW>
__global__ __launch_bounds__(Wide<BKW>::THREADS) void gemm_pc_kernel(const GemmArgs g) {
    using CFG = Wide<BKW>;
    constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK, MT = CFG::MT, PTH = CFG::PTHREADS, ND = CFG::NDMA;
    constexpr int RING = CFG::STAGES * CFG::STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[CFG::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // workgroup -> (k-split, tile).  The hardware deals workgroups to the 8 XCDs round robin; each XCD gets a CONTIGUOUS chunk of the
    // split-major (k-split, tile) list, so the 32 workgroups an XCD runs at a time are one K-range of a compact block of tiles (8 rows x 4
    // columns in the grouped order below: 80 operand columns fetched per tile and k through that XCD's private L2, where the per-split
    // remap of round 1 mixed two K-ranges per XCD at 128).  This kernel is power-limited (GRBM clock 1.85 GHz with the operand traffic,
    // 2.12 without - tools/probes/README.md), so L2-miss bytes are clock.
    int bid;
    int ks;
    {
        const int W = g.ntiles * g.split_k, b0 = blockIdx.x;
        const int nx = 8, q = W / nx, r = W % nx, x = b0 % nx, o = b0 / nx;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
        ks = lin / g.ntiles;
        bid = lin - ks * g.ntiles;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int kt0 = ks * g.ktiles_per_split;
    const int kt1 = min(g.ktiles, kt0 + g.ktiles_per_split);
    const int T_ = kt1 - kt0;

    if (wave >= CFG::CWAVES) {
        // ------------------------------------------------ producers ------------------------------------------------
        const int pw = wave - CFG::CWAVES;
        const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
        const char* a_base = TA ? g.A + m0 * 2 : g.A + m0 * lda_b;
        const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
        int64_t a_bytes = TA ? g.ka_rows * lda_b - m0 * 2 : (g.M - m0) * lda_b;
        int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
        if (a_bytes > 0xFFFFFF00ll) a_bytes = 0xFFFFFF00ll;
        if (b_bytes > 0xFFFFFF00ll) b_bytes = 0xFFFFFF00ll;
        __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
        const int64_t a_crem = min(g.M - m0, (int64_t)BM), b_crem = g.N - n0;
        unsigned voa[ND], vob[ND];
        dma_offsets<TA, 256, PTH, BK, ND>(voa, pw, lane, lda_b, a_crem);
        dma_offsets<TB, 256, PTH, BK, ND>(vob, pw, lane, ldb_b, b_crem);
        // k-contiguous A image: rows are chunk-major, the last quarter of the instructions is exactly rows 192..255 -> never
        // issued; reduction-major A image: columns >= 192 are scattered over all instructions -> issued, marked out of bounds
        constexpr int NA = TA ? ND : ND - ND / 4;
        constexpr int PTI = NA + ND;
        const int nseg = g.e.nseg, kseg = g.e.kseg;
        const bool ktail = (g.K % BK) != 0;
        auto stage = [&](int kt, int bo, int part = 3) {   // part: bit 0 = the A image, bit 1 = the B image
            const int k0 = kt * BK;
            int ka = k0, kb = k0;
            int64_t kda = g.K, kdb = g.K;
            if (nseg > 0) {
                const int sg = k0 / kseg, kin = k0 - sg * kseg;
                ka = g.e.a_seg_off[sg] + kin;
                kb = g.e.b_seg_off[sg] + kin;
                kda = g.e.a_seg_off[sg] + kseg;
                kdb = g.e.b_seg_off[sg] + kseg;
            }
            if ((MICO_GEMM_ABLATE == 1 || MICO_GEMM_ABLATE == 10 || MICO_GEMM_ABLATE == 11) && kt >= kt0 + 3) return;
            if (ktail && kt == g.ktiles - 1) {   // ragged last K-tile: masked path (issues ND + ND instructions; it is waited with vmcnt(0))
                if (part & 1) stage_tile<TA, 256, PTH, BK>(rsa, lds + bo, pw, lane, lda_b, ka, kda, a_crem);
                if (part & 2) stage_tile<TB, 256, PTH, BK>(rsb, lds + bo + CFG::A_BYTES, pw, lane, ldb_b, kb, kdb, b_crem);
                return;
            }
            const unsigned koa = TA ? (unsigned)((int64_t)ka * lda_b) : (unsigned)(ka * 2);
            const unsigned kob = TB ? (unsigned)((int64_t)kb * ldb_b) : (unsigned)(kb * 2);
            if (part & 1) dma_issue<PTH, ND, NA>(rsa, lds + bo, pw, voa, koa);
            if (part & 2) dma_issue<PTH, ND, ND>(rsb, lds + bo + CFG::A_BYTES, pw, vob, kob);
        };
        constexpr int AHEAD = CFG::STAGES - 1;   // K-tiles in flight
        for (int i = 0; i < AHEAD && i < T_; ++i) stage(kt0 + i, i * CFG::STAGE_BYTES);
        // bias gradient riding along (tile column 0 only): this lane's column of the A image, its 8 swizzled chunk offsets
        // (round 3: EVERY tile column takes a share - the workgroups of one tile row stage the same A panel over the same K range, so column
        // n sums the K-tiles with index % ntn == n.  With tile column 0 alone carrying it those workgroups ran ~2x longer than the rest and
        // the launch waited for them: +8 / 26 / 11 / 9 % on the qkv / proj / fc1 / fc2 weight gradients, tools/probes/dw_colsum_cost.py.)
        const bool do_colsum = TA && TB && g.e.colsum_out != nullptr;
        const int cs_ntn = g.ntn;
        int cs_phase = do_colsum ? (tile_n + cs_ntn - kt0 % cs_ntn) % cs_ntn : 0;   // iterations until this column's next K-tile
        const int pid = pw * 64 + lane, lc = pid & 31, rg = pid >> 5;
        // row r = rg + 8 j of the image: byte r * 512 + ((lc ^ key_tr(r)) << 4); key_tr(r) = ((r & 3) | (bit 3 of r) << 2) << 1
        const unsigned cbase = (unsigned)(uintptr_t)lds + (unsigned)rg * 512u;
        const unsigned cx_even = cbase + (unsigned)((lc ^ ((rg & 3) << 1)) << 4), cx_odd = cbase + (unsigned)((lc ^ (((rg & 3) | 4) << 1)) << 4);
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int bo = 0;
        for (int t = 0; t < T_; ++t) {
            asm volatile("" : "+s"(bo));
            const int ahead = T_ - 1 - t;
            // this wave's share of tile t has landed; the barrier publishes it.  (A masked last tile carries more
            // instructions than PTI: the counts below then over-wait, never under-wait.)
            if (AHEAD >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PTI) : "memory");
            else if (AHEAD >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PTI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PC_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            // the buffer of tile t-1: every consumer retired its reads of it before the barrier above.  The refill is issued in two
            // halves, one per consumer phase: every data-moving LDS-DMA instruction stalls its wave ~50 cycles, and all 8 of a K-tile in
            // the first phase (~400 cycles) made the producers the last arrivals at the mid-tile barrier of a 384-cycle MFMA phase.
            const bool refill = t + AHEAD < T_;
            const int rbo = (bo + AHEAD * CFG::STAGE_BYTES) & (RING - 1);
            if (refill) stage(kt0 + t + AHEAD, rbo, CFG::KSTEPS == 1 ? 1 : 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TA && TB) {
                if (do_colsum) {   // tile t is published and stays until the next iteration's barrier
                    if (cs_phase == 0) {
                        tile_colsum32<T>(cs, cx_even + (unsigned)bo, cx_odd + (unsigned)bo, rg);
                        if constexpr (BK == 64) tile_colsum32<T>(cs, cx_even + (unsigned)bo + 32u * 512u, cx_odd + (unsigned)bo + 32u * 512u, rg);   // rows 32..63: same keys
                        cs_phase = cs_ntn;
                    }
                    --cs_phase;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2 * CFG::KSTEPS - 1; ++r) {   // the consumers' remaining barriers of this K-tile
                PC_MIDBAR();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (CFG::KSTEPS == 1 && refill) stage(kt0 + t + AHEAD, rbo, 2);
            __builtin_amdgcn_sched_barrier(0);
            bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
        }
        PC_BARRIER();   // the consumers' end-of-loop barrier
        if (do_colsum) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = cs[k] + __shfl_xor(cs[k], 32, 64);     // the wave's two row groups (lanes l and l + 32 share lc)
                const int col = lc * 8 + k;
                if (lane < 32 && col < BM && m0 + col < g.M) unsafeAtomicAdd(g.e.colsum_out + m0 + col, v * g.e.alpha);
            }
        }
        return;
    }

    // -------------------------------------------------- consumers --------------------------------------------------
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const FragBase ab = frag_base<TA, 256, BK>(wm * 64, lane), ab2 = frag_base<TA, 256, BK>(128 + wm * 32, lane);
    const FragBase bb = frag_base<TB, 256, BK>(wn * 64, lane);
    s16x8 fa[MT], fb[4];
    auto read_k = [&](int bo, int kk) {
        if (MICO_GEMM_ABLATE == 2 && g.K > 0) {   // ablation: no LDS reads (keep fragments opaque)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[j]));
            return;
        }
        LDS_AS const char* ta = lds + bo;
        LDS_AS const char* tb = ta + CFG::A_BYTES;
        const int a1 = kk ? ab.b1 : ab.b0, a2 = kk ? ab2.b1 : ab2.b0, b1 = kk ? bb.b1 : bb.b0;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag_b<TA, 256, BK>(ta, a1, i);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[4 + i] = read_frag_b<TA, 256, BK>(ta, a2, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = read_frag_b<TB, 256, BK>(tb, b1, j);
    };
    auto mma_k = [&]() {
        if (MICO_MMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = T16<T>::mfma(fb[j], fa[i], acc[i][j]);
        if (MICO_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto head = [&]() {   // own LDS reads of the buffer about to be refilled have returned; then the producers' barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PC_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        PC_MIDBAR();
        __builtin_amdgcn_sched_barrier(0);
    };
    int bo = 0;
    if (wm == 0) {
        for (int t = 0; t < T_; ++t) {
            asm volatile("" : "+s"(bo));
            head();
            read_k(bo, 0);
            bar();
            mma_k();
            if constexpr (CFG::KSTEPS == 2) {
                bar();
                read_k(bo, 1);
                bar();
                mma_k();
            }
            __builtin_amdgcn_sched_barrier(0);
            bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
        }
    } else {
        for (int t = 0; t < T_; ++t) {
            asm volatile("" : "+s"(bo));
            head();
            if (t > 0) mma_k();
            bar();
            read_k(bo, 0);
            if constexpr (CFG::KSTEPS == 2) {
                bar();
                mma_k();
                bar();
                read_k(bo, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            bo = (bo + CFG::STAGE_BYTES) & (RING - 1);
        }
        if (T_ > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mma_k();
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PC_BARRIER();   // every wave is done with the operand tiles: LDS may be reused by the epilogue
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: three 32-row blocks per consumer ----
#pragma unroll
    for (int hb = 0; hb < 3; ++hb) {
        const int64_t mrow = m0 + (hb < 2 ? wm * 64 + hb * 32 : 128 + wm * 32);
        if (g.split_k > 1 && g.e.splitk_ws == nullptr) gemm_epilogue_atomic<2>(g, &acc[hb * 2], mrow, n0 + wn * 64, lane);
        else if (g.split_k > 1) {
            // this K-split's partial tile -> its own fp32 [M, N] slab of the scratch, whole lines through the LDS-transposing epilogue;
            // splitk_reduce_kernel adds the slabs into C afterwards
            GemmArgs gp = g;
            gp.C = (char*)g.e.splitk_ws + (int64_t)ks * g.M * g.N * 4;
            gp.ldc = g.N;
            gp.c_dtype = MICO_F32;
            gp.e.alpha = 1.f;
            gp.e.accumulate = 0;
            gemm_epilogue_block<T, 2, ACT_LEAN>(gp, &acc[hb * 2], lds + wave * 8192, mrow, n0 + wn * 64, lane);
        } else gemm_epilogue_block<T, 2, ACT_LEAN>(g, &acc[hb * 2], lds + wave * 8192, mrow, n0 + wn * 64, lane);   // see `pc` in mico_gemm
    }
}


// ======================================================================================================================
// EXPERIMENT (round 2), compiled only with -DMICO_GEMM_W4: a one-wave-per-SIMD kernel.  Findings (tools/probes/README.md, "W4"):
// its MFMA + fragment-read structure runs at 1530-1830 TFLOP/s with the steady-state DMA ablated (the shipped 8-wave kernel: 1160-1480),
// but every LDS-DMA instruction that fetches real data stalls the issuing wave ~40-60 cycles and an in-order wave that is alone on its
// SIMD has no partner to keep the matrix pipe busy meanwhile: with DMA it lands where the shipped kernels are (fwd 880-1110, dX 1010-1100,
// dW 720-900).  The 4-stage variant + transposing reads also showed an unexplained data race on power-of-two row strides.  Not routed.
// ======================================================================================================================
#ifdef MICO_GEMM_W4
// ======================================================================================================================
// W4 kernel: ONE wave per SIMD.
// The two kernels above keep two waves per SIMD (256 registers each), so a wave owns 128x64 outputs and every K-tile costs the CU
// 192 KiB of fragment reads next to the 64 KiB DMA fill - the LDS is as busy as the matrix pipe - and fragments are single-buffered,
// so every phase exposes its ds_read latency behind a barrier.  At one wave per SIMD the unified register file gives a wave 512
// registers: 256x256x64 tile, FOUR waves (2x2), each owning 128x128 = 64 accumulator tiles (256 registers) next to TWO fragment
// sets (2 x 16 fragments = 128 registers).  Per 64-deep K-tile: 128 KiB of fragment reads (-33 %), ONE barrier, and no exposed
// LDS latency - the schedule is a software pipeline inside each wave:
//     first half : 64 MFMAs on set 0 (k-step 0 of tile t)   || ds_read set 1 = k-step 1 of tile t
//     mid        : lgkmcnt(0), vmcnt(0) [tile t+1 landed], s_barrier -> tile t fully read (its stage is free), tile t+1 published
//     second half: 64 MFMAs on set 1                        || ds_read set 0 = k-step 0 of tile t+1, DMA of tile t+2 -> stage of tile t
// with the side operations spread one pair per 8 MFMAs (sched_barrier-pinned) so that the in-order wave never leaves the matrix
// pipe idle behind a burst of VMEM / LDS issues.  Two 64 KiB stages; a tile's DMA is issued one full iteration (~2k cycles) before
// the barrier that publishes it.  P1 = DMA pieces (of 16 per thread and tile) issued in the second half; the other 16 - P1 follow
// in the first half of the next iteration (spreads the TA load; they then have half an iteration to land).
// LDS images, swizzles, fragment addressing, operand orientations and epilogues are those of the kernels above.
// ======================================================================================================================
template <int DEEP> struct W4C {   // DEEP 0: 2 stages of 64-deep K-tiles; 1: 4 stages of 32-deep K-tiles (see the schedule notes in the kernel)
    static constexpr int BM = 256, BN = 256, BK = DEEP ? 32 : 64, STAGES = DEEP ? 4 : 2, THREADS = 256, MT = 8;
    static constexpr int A_BYTES = 256 * BK * 2, STAGE_BYTES = 2 * A_BYTES, LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int NDMA = A_BYTES / 16 / THREADS;   // DMA instructions per thread per operand tile (8 / 4)
};
using W4 = W4C<0>;

template <typename T, bool TA, bool TB, int ACT, int P1, int DEEP>
__global__ __launch_bounds__(W4::THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(const GemmArgs g) {
    using W4 = W4C<DEEP>;
    constexpr int BM = W4::BM, BN = W4::BN, BK = W4::BK, THREADS = W4::THREADS, ND = W4::NDMA;
    constexpr int STAGE = W4::STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[W4::LDS_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int wrow = wm * 128, wcol = wn * 128;

    int bid = blockIdx.x;
    const int ks = bid / g.ntiles;
    bid -= ks * g.ntiles;
    {
        const int nx = 8, q = g.ntiles / nx, r = g.ntiles % nx, x = bid % nx, o = bid / nx;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    }
    int tile_m, tile_n;
    {
        const int gsz = GROUP_M * g.ntn;
        const int grp = bid / gsz;
        const int first = grp * GROUP_M;
        const int gm = min(g.ntm - first, GROUP_M);
        const int in = bid - grp * gsz;
        tile_m = first + in % gm;
        tile_n = in / gm;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int kt0 = ks * g.ktiles_per_split;
    const int kt1 = min(g.ktiles, kt0 + g.ktiles_per_split);
    const int T_ = kt1 - kt0;

    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;
    const char* a_base = TA ? g.A + m0 * 2 : g.A + m0 * lda_b;
    const char* b_base = TB ? g.B + n0 * 2 : g.B + n0 * ldb_b;
    int64_t a_bytes = TA ? g.ka_rows * lda_b - m0 * 2 : (g.M - m0) * lda_b;
    int64_t b_bytes = TB ? g.kb_rows * ldb_b - n0 * 2 : (g.N - n0) * ldb_b;
    if (a_bytes > 0x7FFFFF00ll) a_bytes = 0x7FFFFF00ll;   // (mico_gemm routes here only when every offset in use stays below 2^31)
    if (b_bytes > 0x7FFFFF00ll) b_bytes = 0x7FFFFF00ll;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
    const int64_t a_crem = g.M - m0, b_crem = g.N - n0;

    f32x4 acc[2][8][4];   // [64-column half][16-row tile][16-column tile]: &acc[c][4 h] is one 64x64 epilogue block
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // k-segment s (split-precision GEMMs) maps logical k to physical column k - s * kseg + seg_off[s]
    // (readfirstlane: pins the six offsets into SGPRs here - left to itself hipcc re-selects and re-loads them inside the loop)
    const bool segs = g.e.nseg > 0;
    const int kseg_ = __builtin_amdgcn_readfirstlane(segs ? g.e.kseg : 0x3FFFFFFF);
    const int aso0 = __builtin_amdgcn_readfirstlane(segs ? g.e.a_seg_off[0] : 0);
    const int aso1 = __builtin_amdgcn_readfirstlane(segs ? g.e.a_seg_off[1] - kseg_ - g.e.a_seg_off[0] : 0);          // increments
    const int aso2 = __builtin_amdgcn_readfirstlane(segs ? g.e.a_seg_off[2] - kseg_ - g.e.a_seg_off[1] : 0);
    const int bso0 = __builtin_amdgcn_readfirstlane(segs ? g.e.b_seg_off[0] : 0);
    const int bso1 = __builtin_amdgcn_readfirstlane(segs ? g.e.b_seg_off[1] - kseg_ - g.e.b_seg_off[0] : 0);
    const int bso2 = __builtin_amdgcn_readfirstlane(segs ? g.e.b_seg_off[2] - kseg_ - g.e.b_seg_off[1] : 0);
    const FragBase ab = frag_base<TA, BM, BK>(wrow, lane), bb = frag_base<TB, BN, BK>(wcol, lane);
    unsigned voa[ND], vob[ND];
    dma_offsets<TA, BM, THREADS, BK, ND>(voa, wave, lane, lda_b, a_crem);
    dma_offsets<TB, BN, THREADS, BK, ND>(vob, wave, lane, ldb_b, b_crem);
#pragma unroll
    for (int p = 0; p < ND; ++p) {   // w4_edge
        if (voa[p] == 0xFFFFFFF0u) voa[p] = 0x80000000u;
        if (vob[p] == 0xFFFFFFF0u) vob[p] = 0x80000000u;
    }

    // source of one K-tile: scalar byte offsets of both operands.  The loop body is ONE basic block (256 accumulator registers live
    // across control flow made the register allocator rotate them through copies): a tile past the end of this workgroup's K range
    // is "loaded" with every lane out of bounds (no memory traffic, the LDS image is zero-filled and never read).  A ragged last
    // K-tile needs no masking here: reduction-major operands end at the descriptor's bound (rows >= K zero-fill) and problems with
    // a k-contiguous operand and K % 64 != 0 are not routed to this kernel (mico_gemm).
    struct Src { unsigned koa, kob; bool valid; };
    auto src_of = [&](int kt) {
        Src s;
        s.valid = kt < kt1;
        const int k0 = (MICO_GEMM_ABLATE == 5 ? kt0 + ((kt - kt0) & 1) : kt) * BK;   // ablation 5: re-read the first two K-tiles (cache-resident)
        // k-segments without control flow: segment index by comparison, offsets from scalars read once
        // (arithmetic, not ?: - a select between by-reference captures becomes a select of ADDRESSES that keeps them in memory)
        const int sg1 = k0 >= kseg_, sg2 = k0 >= 2 * kseg_;
        const int ka = k0 + aso0 + sg1 * aso1 + sg2 * aso2, kb = k0 + bso0 + sg1 * bso1 + sg2 * bso2;
        s.koa = TA ? (unsigned)((int64_t)ka * lda_b) : (unsigned)(ka * 2);
        s.kob = TB ? (unsigned)((int64_t)kb * ldb_b) : (unsigned)(kb * 2);
        return s;
    };
    // one DMA piece of the 16 per thread and tile: 0-7 operand A, 8-15 operand B.  A piece past the matrix edge carries offset 2^31
    // (w4_edge above): adding a tile's k offset (< 2^31, checked in mico_gemm) keeps it beyond the descriptor's bound (< 2^31) without
    // a per-piece condition (16 SGPR pairs the kernel does not have).
    auto piece = [&](const Src& s, int bo, int p) {
        if (MICO_GEMM_ABLATE == 1) return;
        if (p < ND) {
            unsigned v = s.valid ? voa[p] + s.koa : 0xFFFFFFF0u;
            if (MICO_GEMM_ABLATE == 4) v = 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (LDS_AS void*)(lds + bo + (p * THREADS + wave * 64) * 16), 16, v, 0, 0, 0);
        } else {
            unsigned v = s.valid ? vob[p - ND] + s.kob : 0xFFFFFFF0u;
            if (MICO_GEMM_ABLATE == 4) v = 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (LDS_AS void*)(lds + bo + W4::A_BYTES + ((p - ND) * THREADS + wave * 64) * 16), 16, v, 0, 0, 0);
        }
    };
    // two fragment sets, named so that every index is a compile-time constant (k-step 0 / k-step 1)
    s16x8 fa0[8], fb0[8], fa1[8], fb1[8];
    auto frag = [&](s16x8 (&fax)[8], s16x8 (&fbx)[8], LDS_AS const char* pa, LDS_AS const char* pb, int abase, int bbase, int idx) {
        if (MICO_GEMM_ABLATE == 2) {
            if (idx < 8) asm volatile("" : "+v"(fax[idx])); else asm volatile("" : "+v"(fbx[idx - 8]));
            return;
        }
        if (idx < 8) fax[idx] = read_frag_dma<TA, BM, BK>(pa, abase, idx);
        else fbx[idx - 8] = read_frag_dma<TB, BN, BK>(pb, bbase, idx - 8);
    };
    // One half = 64 MFMAs (8 row tiles x 8 column tiles) on set (fax, fbx).  The wave is alone on its SIMD and issues in order, so
    // whatever sits between two MFMAs must fit the 16 cycles the first one occupies the matrix pipe: the half's side work - the 16
    // fragment requests of the NEXT set and NP DMA pieces - is dealt out one item per MFMA slot (fragments in the even slots of
    // the first half of the slots, pieces evenly over the odd ones) and every slot is pinned with a sched_barrier.
    // WOFF (0-3, the wave's index): the four waves issue their pieces in DIFFERENT slots.  A piece that fetches real data occupies the
    // CU's one texture-address unit for ~16 cycles = one MFMA slot; issued in the same slot by all four waves (they run in lockstep
    // between barriers) the last one waits for three others every time - measured as ~1000 lost cycles per 2048-cycle K-tile.
    auto half = [&](const s16x8 (&fax)[8], const s16x8 (&fbx)[8], s16x8 (&nax)[8], s16x8 (&nbx)[8], LDS_AS const char* pa,
                    LDS_AS const char* pb, int abase, int bbase, const Src& s, int bo, auto p0_c, auto np_c, auto woff_c, auto late_c) {
        constexpr int P0 = decltype(p0_c)::value, NP = decltype(np_c)::value, WOFF = decltype(woff_c)::value;
        constexpr bool LATE = decltype(late_c)::value != 0;
        constexpr int STEP = LATE ? 4 : 64 / (NP > 0 ? NP : 1);          // late pieces (first half, needed at the middle barrier): early slots
        constexpr int SLOT0 = MICO_W4_STAGGER ? (STEP >= 4 ? WOFF * (STEP / 4) : 0) : (STEP > 1 ? 1 : 0);
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int i = m >> 3, cj = m & 7;
            mfma_acc<T>(fbx[cj], fax[i], acc[cj >> 2][i][cj & 3]);
#if MICO_W4_DBG == 6
            if (m >= 30 && m < 62 && (m & 1) == 0) frag(nax, nbx, pa, pb, abase, bbase, (m - 30) >> 1);
#else
            if (m < 32 && (m & 1) == 0) frag(nax, nbx, pa, pb, abase, bbase, m >> 1);
#endif
            if (NP > 0 && (m % STEP) == SLOT0 && m / STEP < NP) piece(s, bo, P0 + m / STEP);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#define IC(N) std::integral_constant<int, (N)>{}

    if constexpr (DEEP) {
        // ---- 4 stages of 32-deep K-tiles (one MFMA k-step each).  Iteration t: 64 MFMAs on the fragments of tile t (in registers since
        // iteration t-1) || fragment requests of tile t+1 || DMA of tile t+4 into the stage tile t occupied (already in registers).
        // Top of the iteration: vmcnt(16) = this wave's share of tile t+1 has landed (t+2, t+3 stay in flight), lgkmcnt(0), ONE
        // barrier: tile t+1 is visible and nobody reads tile t's stage any more.  A tile's DMA is issued three iterations (~3k cycles)
        // before the barrier that publishes it and 96 KiB are in flight per CU.  Tiles past the end of the K range are "loaded"
        // out of bounds, i.e. zero-filled: the loop runs in pairs (static fragment-set names) and an odd tail multiplies zeros.
        constexpr int PT = 2 * ND;   // 8 pieces per thread and tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const Src sq = src_of(kt0 + q);
#pragma unroll
            for (int p = 0; p < PT; ++p) piece(sq, q * STAGE, p);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MICO_GEMM_ABLATE == 1 ? 0 : 3 * PT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) frag(fa0, fb0, lds, lds + W4::A_BYTES, ab.b0, bb.b0, idx);
        int bo = 0;
#if MICO_GEMM_ABLATE == 8
        unsigned long long prof_wait = 0, prof_bar = 0;
        const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
#endif
        auto top = [&]() {
#if MICO_GEMM_ABLATE == 8
            const unsigned long long a0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PT) : "memory");
            const unsigned long long a1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long a2 = __builtin_amdgcn_s_memtime();
            prof_wait += a1 - a0; prof_bar += a2 - a1;
#else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((MICO_GEMM_ABLATE == 1 || MICO_W4_DBG == 1) ? 0 : MICO_W4_DBG == 4 ? PT : 2 * PT) : "memory");
            __builtin_amdgcn_s_barrier();
#if MICO_W4_DBG == 5
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int t = 0; t < T_; t += 2) {
            asm volatile("" : "+s"(bo));
            {
                top();
                const Src s4 = src_of(kt0 + t + 4);
                const int nbo = (bo + STAGE) & (4 * STAGE - 1);
                half(fa0, fb0, fa1, fb1, lds + nbo, lds + nbo + W4::A_BYTES, ab.b0, bb.b0, s4, bo, IC(0), IC(PT), IC(0), IC(0));
                bo = nbo;
            }
            {
                top();
                const Src s4 = src_of(kt0 + t + 5);
                const int nbo = (bo + STAGE) & (4 * STAGE - 1);
                half(fa1, fb1, fa0, fb0, lds + nbo, lds + nbo + W4::A_BYTES, ab.b0, bb.b0, s4, bo, IC(0), IC(PT), IC(0), IC(0));
                bo = nbo;
            }
        }
#if MICO_GEMM_ABLATE == 8
        if (threadIdx.x == 0) {
            atomicAdd(&g_mico_w4_prof[0], prof_wait);
            atomicAdd(&g_mico_w4_prof[1], prof_bar);
            atomicAdd(&g_mico_w4_prof[2], __builtin_amdgcn_s_memtime() - prof_t0);
            atomicAdd(&g_mico_w4_prof[3], (unsigned long long)((T_ + 1) / 2 * 2));
        }
#endif
    } else {
    // ---- prologue: tile 0 and the first P1 pieces of tile 1 in flight, tile 0 published, its k-step 0 fragments requested ----
    {
        const Src s0 = src_of(kt0), s1 = src_of(kt0 + 1);
#pragma unroll
        for (int p = 0; p < 16; ++p) piece(s0, 0, p);
#pragma unroll
        for (int p = 0; p < P1; ++p) piece(s1, STAGE, p);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MICO_GEMM_ABLATE == 1 ? 0 : P1) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) frag(fa0, fb0, lds, lds + W4::A_BYTES, ab.b0, bb.b0, idx);
    }
    auto kloop = [&](auto woff_c) {
    int bo = 0;
        for (int t = 0; t < T_; ++t) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bo) : : "memory");   // the fragment requests of the previous half (asm reads are not counted by the compiler)
            __builtin_amdgcn_sched_barrier(0);
            LDS_AS const char* ta = lds + bo;
            LDS_AS const char* tb = ta + W4::A_BYTES;
            const Src sn = src_of(kt0 + t + 1);   // late pieces of the next tile (other stage)
            // ---- first half: MFMAs of k-step 0; requests k-step 1; the late pieces of tile t+1 ----
            half(fa0, fb0, fa1, fb1, ta, tb, ab.b1, bb.b1, sn, bo ^ STAGE, IC(P1), IC(16 - P1), woff_c, IC(1));
            // ---- mid: tile t+1 landed (own pieces), barrier: stage `bo` is no longer read by anyone, tile t+1 is visible ----
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const Src s2 = src_of(kt0 + t + 2);
            LDS_AS const char* na = lds + (bo ^ STAGE);
            LDS_AS const char* nb = na + W4::A_BYTES;
            // ---- second half: MFMAs of k-step 1; requests tile t+1's k-step 0 (past the last tile: stale bytes, never used); the first P1
            // pieces of tile t+2 into this stage ----
            half(fa1, fb1, fa0, fb0, na, nb, ab.b0, bb.b0, s2, bo, IC(0), IC(P1), woff_c, IC(0));
            bo ^= STAGE;
        }
    };
    // four copies of the loop, one per wave, differing only in the slots their DMA pieces are issued in
    if (!MICO_W4_STAGGER || wave == 0) kloop(IC(0));
    else if (wave == 1) kloop(IC(1));
    else if (wave == 2) kloop(IC(2));
    else kloop(IC(3));
    }
#undef IC
    mfma_acc_fence();
    // ---- epilogue: four 64x64 blocks per wave through its 16 KiB of LDS ----
    if (g.split_k > 1) {
        // the trailing pieces (tiles past the end, zero-fill) must have landed before this workgroup ends: LDS-DMA still in flight at
        // s_endpgm lands in the LDS of the NEXT workgroup on this CU (seen as sporadic wrong split-K tiles)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) gemm_epilogue_atomic(g, &acc[c][h * 4], m0 + wrow + h * 64, n0 + wcol + c * 64, lane);
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                gemm_epilogue_block<T, 4, ACT>(g, &acc[c][h * 4], lds + wave * 16384, m0 + wrow + h * 64, n0 + wcol + c * 64, lane);
    }
}

#endif   // MICO_GEMM_W4 (kernel)

#ifndef MICO_PC_BK
#define MICO_PC_BK 32
#endif
constexpr int pc_bk(int, int) { return MICO_PC_BK; }

// C[m, n] += alpha * sum_s ws[s][m][n]   (the split-K slabs of the weight-gradient kernel; N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t M, int64_t N, float* __restrict__ C,
                                                            int64_t ldc, float alpha) {
    const int64_t n4 = N / 4, total = M * n4, slab = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / n4, n = (i - m * n4) * 4;
        f32x4 acc = *(const f32x4*)(ws + m * N + n);
        for (int s = 1; s < nsplit; ++s) acc += *(const f32x4*)(ws + s * slab + m * N + n);
        f32x4* cp = (f32x4*)(C + m * ldc + n);
        *cp = *cp + acc * alpha;
    }
}

// the launches ACT_RESID serves: lean + fp32 residual output, no split, no accumulate (MICO_NO_RESID_EPI: A/B switch, shared epilogue)
bool resid_epilogue(const GemmArgs& g) {
#ifdef MICO_NO_RESID_EPI
    return false;
#else
    return g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group && g.e.resid != nullptr &&
           g.c_dtype == MICO_F32 && !g.e.accumulate && g.split_k == 1 && g.e.alpha == 1.f;
#endif
}

template <typename T>
void launch_pc(int ta, int tb, const GemmArgs& g, hipStream_t st) {
    // only the weight-gradient orientation is routed here (see mico_gemm); the kernel template also covers k-contiguous operands
    // with 64-deep K-tiles (measured slower than the 8-wave kernel on the forward / dX shapes, hence not instantiated)
    const dim3 grid(g.ntiles * g.split_k), block(Wide<32>::THREADS);
    if (ta && tb) MICO_LAUNCH((gemm_pc_kernel<T, true, true, MICO_PC_BK>), grid, block, 0, st, g);
#ifdef MICO_GEMM_PC_ALL   // experiment build: every large problem through the producer/consumer kernel (32-deep K-tiles)
    else if (!ta && !tb) MICO_LAUNCH((gemm_pc_kernel<T, false, false, 32>), grid, block, 0, st, g);
    else if (!ta && tb) MICO_LAUNCH((gemm_pc_kernel<T, false, true, 32>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_pc_kernel<T, true, false, 32>), grid, block, 0, st, g);
#endif
}

// persistent launch of the 8-wave kernel: the lean and the MLP-pair instantiations (the launches that carry the towers' forward / dX work)
template <typename T>
bool launch_persist(int ta, int tb, const GemmArgs& g, hipStream_t st) {
#ifndef MICO_GEMM_PERSIST
    return false;
#else
    const dim3 grid(256), block(Big::THREADS);
    if (ta) return false;
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { MICO_LAUNCH((gemm_persist_kernel<T, false, false, MICO_ACT_GELU_SAVE_DERIV>), grid, block, 0, st, g); return true; }
    if (g.e.act == MICO_ACT_MUL_AUX) { MICO_LAUNCH((gemm_persist_kernel<T, false, true, MICO_ACT_MUL_AUX>), grid, block, 0, st, g); return true; }
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    if (!lean) return false;
    if (!tb) MICO_LAUNCH((gemm_persist_kernel<T, false, false, ACT_LEAN>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_persist_kernel<T, false, true, ACT_LEAN>), grid, block, 0, st, g);
    return true;
#endif
}

template <typename T, typename CFG>
void launch(int ta, int tb, const GemmArgs& g, hipStream_t st) {
    const dim3 grid(g.ntiles * g.split_k), block(CFG::THREADS);
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { MICO_LAUNCH((gemm_kernel<T, false, false, CFG, MICO_ACT_GELU_SAVE_DERIV>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_MUL_AUX) { MICO_LAUNCH((gemm_kernel<T, false, true, CFG, MICO_ACT_MUL_AUX>), grid, block, 0, st, g); return; }
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    if (resid_epilogue(g) && !ta && !tb && CFG::BM == 256) { MICO_LAUNCH((gemm_kernel<T, false, false, CFG, ACT_RESID>), grid, block, 0, st, g); return; }
    if (lean && !ta && !tb) { MICO_LAUNCH((gemm_kernel<T, false, false, CFG, ACT_LEAN>), grid, block, 0, st, g); return; }
    if (lean && !ta && tb) { MICO_LAUNCH((gemm_kernel<T, false, true, CFG, ACT_LEAN>), grid, block, 0, st, g); return; }
    if (lean && ta && tb && CFG::BM == 128) { MICO_LAUNCH((gemm_kernel<T, true, true, CFG, ACT_LEAN>), grid, block, 0, st, g); return; }   // BERT weight gradients
    if (!ta && !tb) MICO_LAUNCH((gemm_kernel<T, false, false, CFG>), grid, block, 0, st, g);
    else if (!ta && tb) MICO_LAUNCH((gemm_kernel<T, false, true, CFG>), grid, block, 0, st, g);
    else if (ta && tb) MICO_LAUNCH((gemm_kernel<T, true, true, CFG>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_kernel<T, true, false, CFG>), grid, block, 0, st, g);
}

template <typename T>
void launch_mid(int tb, const GemmArgs& g, hipStream_t st) {
    const dim3 grid(g.ntiles), block(Mid64::THREADS);
    if (resid_epilogue(g) && !tb) { MICO_LAUNCH((gemm_mid_kernel<T, false, ACT_RESID>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { MICO_LAUNCH((gemm_mid_kernel<T, false, MICO_ACT_GELU_SAVE_DERIV>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_MUL_AUX) { MICO_LAUNCH((gemm_mid_kernel<T, true, MICO_ACT_MUL_AUX>), grid, block, 0, st, g); return; }
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    if (lean && !tb) MICO_LAUNCH((gemm_mid_kernel<T, false, ACT_LEAN>), grid, block, 0, st, g);
    else if (lean) MICO_LAUNCH((gemm_mid_kernel<T, true, ACT_LEAN>), grid, block, 0, st, g);
    else if (!tb) MICO_LAUNCH((gemm_mid_kernel<T, false, 0>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_mid_kernel<T, true, 0>), grid, block, 0, st, g);
}

// the persistent form: fast16 launches with one of the 16-bit staged epilogues, enough tiles for several per CU
template <typename T>
bool launch_p8p(int tb, const GemmArgs& g, hipStream_t st) {
    if (!MICO_P8_PERSIST || !g.fast16 || g.N % 8 != 0 || g.tm0 != 0 || g.ntiles <= 256 || g.ktiles < 2) return false;
    const dim3 grid(256), block(P8C::THREADS);
    if (g.e.aux_tiled) {      // the MLP pair with the tiled GELU' tensor (p8p_aux_tile_desc): the only kernels that know the layout
        if (g.N % 256 != 0) return false;
        if (g.e.act == MICO_ACT_MUL_AUX && tb && g.e.aux_in && !g.e.aux_out) { MICO_LAUNCH((gemm_p8p_kernel<T, true, ACT_MUL_TILED>), grid, block, 0, st, g); return true; }
        if (g.e.act == MICO_ACT_GELU_SAVE_DERIV && !tb && g.e.aux_out) { MICO_LAUNCH((gemm_p8p_kernel<T, false, ACT_PAIR_TILED>), grid, block, 0, st, g); return true; }
        // the pre-activation-keeping pair (round 6): GELU with the tiled pre-activation copy / GELU' multiply from it + gelu(pre-activation) row-major
        if (g.e.act == MICO_ACT_GELU && !tb && g.e.aux_out && !g.e.aux_in) { MICO_LAUNCH((gemm_p8p_kernel<T, false, ACT_PRE_TILED>), grid, block, 0, st, g); return true; }
        if (g.e.act == MICO_ACT_GELU_GRAD && tb && g.e.aux_in && g.e.aux_out && !g.e.bias) { MICO_LAUNCH((gemm_p8p_kernel<T, true, ACT_GRAD_TILED>), grid, block, 0, st, g); return true; }
        return false;
    }
    if (g.e.aux_in) return false;
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { if (tb) return false; MICO_LAUNCH((gemm_p8p_kernel<T, false, MICO_ACT_GELU_SAVE_DERIV>), grid, block, 0, st, g); return true; }
    if (g.e.act == MICO_ACT_GELU) { if (tb || g.e.aux_out) return false; MICO_LAUNCH((gemm_p8p_kernel<T, false, MICO_ACT_GELU>), grid, block, 0, st, g); return true; }
    if (g.e.act != MICO_ACT_NONE || g.e.aux_out) return false;
    if (!tb) MICO_LAUNCH((gemm_p8p_kernel<T, false, ACT_LEAN>), grid, block, 0, st, g);
    else if (g.N % 128 == 0) MICO_LAUNCH((gemm_p8p_kernel<T, true, ACT_LEAN>), grid, block, 0, st, g);
    else return false;
    return true;
}

template <typename T>
void launch_p8(int tb, const GemmArgs& g, hipStream_t st, bool persist = true) {
    if (persist && launch_p8p<T>(tb, g, st)) return;
    const dim3 grid(g.ntiles), block(P8C::THREADS);
    if (resid_epilogue(g) && !tb) { MICO_LAUNCH((gemm_p8_kernel<T, false, ACT_RESID>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { MICO_LAUNCH((gemm_p8_kernel<T, false, MICO_ACT_GELU_SAVE_DERIV>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_MUL_AUX) { MICO_LAUNCH((gemm_p8_kernel<T, true, MICO_ACT_MUL_AUX>), grid, block, 0, st, g); return; }
    // plain GELU with the straight-line 16-bit epilogue (the towers' fc1 when the forward keeps no derivative)
    if (g.e.act == MICO_ACT_GELU && g.fast16 && !tb && !g.e.aux_out && !g.e.aux_in) { MICO_LAUNCH((gemm_p8_kernel<T, false, MICO_ACT_GELU>), grid, block, 0, st, g); return; }
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    if (lean && !tb) MICO_LAUNCH((gemm_p8_kernel<T, false, ACT_LEAN>), grid, block, 0, st, g);
    else if (lean) MICO_LAUNCH((gemm_p8_kernel<T, true, ACT_LEAN>), grid, block, 0, st, g);
    else if (!tb) MICO_LAUNCH((gemm_p8_kernel<T, false, 0>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_p8_kernel<T, true, 0>), grid, block, 0, st, g);
}

#ifdef MICO_GEMM_W4
#ifndef MICO_W4_P1
#define MICO_W4_P1 16
#endif
template <typename T, int DEEP>
void launch_w4(int ta, int tb, const GemmArgs& g, hipStream_t st) {
    constexpr int P1 = MICO_W4_P1;
    const dim3 grid(g.ntiles * g.split_k), block(W4::THREADS);
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) { MICO_LAUNCH((gemm_w4_kernel<T, false, false, MICO_ACT_GELU_SAVE_DERIV, P1, DEEP>), grid, block, 0, st, g); return; }
    if (g.e.act == MICO_ACT_MUL_AUX) { MICO_LAUNCH((gemm_w4_kernel<T, false, true, MICO_ACT_MUL_AUX, P1, DEEP>), grid, block, 0, st, g); return; }
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    if (lean && !ta && !tb) { MICO_LAUNCH((gemm_w4_kernel<T, false, false, ACT_LEAN, P1, DEEP>), grid, block, 0, st, g); return; }
    if (lean && !ta && tb) { MICO_LAUNCH((gemm_w4_kernel<T, false, true, ACT_LEAN, P1, DEEP>), grid, block, 0, st, g); return; }
    if (lean && ta && tb) { MICO_LAUNCH((gemm_w4_kernel<T, true, true, ACT_LEAN, P1, DEEP>), grid, block, 0, st, g); return; }
    if (!ta && !tb) MICO_LAUNCH((gemm_w4_kernel<T, false, false, 0, P1, DEEP>), grid, block, 0, st, g);
    else if (!ta && tb) MICO_LAUNCH((gemm_w4_kernel<T, false, true, 0, P1, DEEP>), grid, block, 0, st, g);
    else if (ta && tb) MICO_LAUNCH((gemm_w4_kernel<T, true, true, 0, P1, DEEP>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_w4_kernel<T, true, false, 0, P1, DEEP>), grid, block, 0, st, g);
}

#endif   // MICO_GEMM_W4 (launcher)

// split factor for fp32-accumulating (weight-gradient) GEMMs: fill `slots` resident workgroups in whole waves.
// cost(s) = waves(s) * (k-tiles per split + fixed prologue / atomic-epilogue cost in k-tile units)
// per_split: cost of one more split that does not scale with the number of waves (the slab reduction pass reads one [M, N] slab per split)
int auto_split(int tiles, int ktiles, int slots, int min_tiles, int fixed, int max_split = 32, int per_split = 0) {
    int best = 1;
    long best_cost = -1;
    for (int s = 1; s <= 32 && s <= max_split; ++s) {
        if (s > 1 && ktiles / s < min_tiles) break;
        const long waves = ((long)tiles * s + slots - 1) / slots;
        const long cost = waves * ((ktiles + s - 1) / s + fixed) + (long)(s > 1 ? s * per_split : 0);
        if (best_cost < 0 || cost < best_cost) { best = s; best_cost = cost; }
    }
    return best;
}

// Split-K of the small-tile kernel for fp32-accumulate (weight-gradient) problems of one round of workgroups or less - BERT's 768 /
// 2304 / 3072-wide layers over a few thousand token rows.  There the atomic epilogues of all splits land together at the end and are
// not hidden behind other workgroups' K loops: measured ~4 us per MB of fp32 atomics (tools/probes/README.md), against ~1 us per
// 64-deep K-tile of a workgroup that has its CU to itself.  Cost in microseconds; the old rule (as many splits as fill the chip) paid
// 134 us where one pass takes 85 (3072 x 768 x 4928; in situ 143 -> 107) and 97 where three splits take 81 (768 x 768 x 4928, in situ).
// The same with the split-K slab path (mico_gemm_epilogue::splitk_ws): plain-store epilogues (~0.03 us per 128x128 tile at a few TB/s),
// then one reduction pass over s slabs of mn fp32 elements (~4 TB/s) + its launch.
int small_slab_split(int tiles, int ktiles, double mn, int max_split) {
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 32 && s <= ktiles && s <= max_split; ++s) {
        const double wgs = (double)tiles * s;
        const double compute = ((ktiles + s - 1) / s) * 1.0 * (wgs > 512.0 ? wgs / 512.0 : 1.0) + 3.0;   // 2 workgroups per CU; pipeline fill
        const double slabs = s > 1 ? wgs * 0.03 + s * mn * 1e-6 + 4.0 : 0.0;
        const double cost = compute + slabs;
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

int small_acc_split(int tiles, int ktiles) {
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 16 && s <= ktiles; ++s) {
        const double wgs = (double)tiles * s;
        const double compute = ((ktiles + s - 1) / s) * 1.0 * (wgs > 256.0 ? wgs / 256.0 : 1.0);
        const double atomics = s > 1 ? wgs * (128.0 * 128.0 * 4.0 / 1e6) * 4.0 : 0.0;
        const double cost = compute + atomics;
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

}  // namespace

thread_local char g_mico_err[256] = "";

int mico_set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mico_err, sizeof(g_mico_err), fmt, ap);
    va_end(ap);
    return code;
}

thread_local int g_mico_last_gemm_kernel = 0;   // per calling thread: mico_gemm_last_kernel() is safe next to other threads' launches
// Kernel routing of mico_gemm.  The product library routes by the problem alone (variant 0, a compile-time constant: mico_gemm keeps no
// process-global routing state and the kernels only other variants reach are not instantiated).  The experiment switch - force the 32-deep
// 8-wave kernel / the 256x128 kernels / the 8-phase kernel with its generic epilogue ... for same-process A/B runs and for the test that
// runs EVERY large-tile kernel on a chip-filling problem - exists in the probe build only: `make -C mico_amd/csrc variants`
// (-DMICO_GEMM_VARIANTS=1 -> tools/probes/bin/libmico_variants.so, which also exports mico_gemm_set_variant).
#ifdef MICO_GEMM_VARIANTS
static int g_mico_gemm_variant = 0;   // 0 = default routing; see the routing block of mico_gemm for the values
static int g_mico_mid_group = 0;       // sweeps: variant / 100 overrides the MID kernel's tile-order group height
extern "C" int mico_gemm_set_variant(int v) { const int old = g_mico_gemm_variant + 100 * g_mico_mid_group; g_mico_gemm_variant = v % 100; g_mico_mid_group = v / 100; return old; }
#else
static constexpr int g_mico_gemm_variant = 0;
static constexpr int g_mico_mid_group = 0;
#endif
extern "C" int mico_gemm_last_kernel(void) { return g_mico_last_gemm_kernel; }
extern "C" int mico_version(void) { return 115; }
extern "C" const char* mico_last_error_string(void) { return g_mico_err; }

extern "C" int mico_struct_layout(int* out, int n) {
#define OFF(S, F) (int)offsetof(S, F)
    const int t[] = {
        (int)sizeof(mico_gemm_epilogue),
        OFF(mico_gemm_epilogue, bias), OFF(mico_gemm_epilogue, aux_out), OFF(mico_gemm_epilogue, aux_in), OFF(mico_gemm_epilogue, ldaux),
        OFF(mico_gemm_epilogue, act), OFF(mico_gemm_epilogue, row_scale), OFF(mico_gemm_epilogue, rows_per_scale), OFF(mico_gemm_epilogue, resid),
        OFF(mico_gemm_epilogue, pos), OFF(mico_gemm_epilogue, pos_rows), OFF(mico_gemm_epilogue, remap_group), OFF(mico_gemm_epilogue, remap_skip),
        OFF(mico_gemm_epilogue, remap_offset), OFF(mico_gemm_epilogue, alpha), OFF(mico_gemm_epilogue, accumulate), OFF(mico_gemm_epilogue, nseg),
        OFF(mico_gemm_epilogue, kseg), OFF(mico_gemm_epilogue, a_seg_off), OFF(mico_gemm_epilogue, b_seg_off), OFF(mico_gemm_epilogue, row_map),
        OFF(mico_gemm_epilogue, rows_per_map), OFF(mico_gemm_epilogue, drop_p), OFF(mico_gemm_epilogue, drop_seed), OFF(mico_gemm_epilogue, drop_site),
        OFF(mico_gemm_epilogue, colsum_out), OFF(mico_gemm_epilogue, splitk_ws), OFF(mico_gemm_epilogue, splitk_ws_bytes),
        OFF(mico_gemm_epilogue, aux_tiled),
        -1,
        (int)sizeof(mico_attn_params),
        OFF(mico_attn_params, B), OFF(mico_attn_params, H), OFF(mico_attn_params, Sq), OFF(mico_attn_params, Sk), OFF(mico_attn_params, hd),
        OFF(mico_attn_params, q_bs), OFF(mico_attn_params, q_rs), OFF(mico_attn_params, k_bs), OFF(mico_attn_params, k_rs), OFF(mico_attn_params, v_bs),
        OFF(mico_attn_params, v_rs), OFF(mico_attn_params, o_bs), OFF(mico_attn_params, o_rs), OFF(mico_attn_params, scale), OFF(mico_attn_params, mask),
        OFF(mico_attn_params, mask_mode), OFF(mico_attn_params, drop_p), OFF(mico_attn_params, drop_seed), OFF(mico_attn_params, drop_site),
        OFF(mico_attn_params, kv_batch_mod), OFF(mico_attn_params, batch0), OFF(mico_attn_params, dkv_accumulate),
        -1,
        (int)sizeof(mico_ln_fwd_params),
        OFF(mico_ln_fwd_params, x), OFF(mico_ln_fwd_params, x_dtype), OFF(mico_ln_fwd_params, x_normalized), OFF(mico_ln_fwd_params, gamma),
        OFF(mico_ln_fwd_params, beta), OFF(mico_ln_fwd_params, y16), OFF(mico_ln_fwd_params, y32), OFF(mico_ln_fwd_params, mean),
        OFF(mico_ln_fwd_params, rstd), OFF(mico_ln_fwd_params, rows), OFF(mico_ln_fwd_params, cols), OFF(mico_ln_fwd_params, eps),
        OFF(mico_ln_fwd_params, post_add), OFF(mico_ln_fwd_params, post_rows_per_group), OFF(mico_ln_fwd_params, post_groups),
        OFF(mico_ln_fwd_params, y16_split), OFF(mico_ln_fwd_params, frame_map), OFF(mico_ln_fwd_params, rows_per_frame),
        OFF(mico_ln_fwd_params, x_copy), OFF(mico_ln_fwd_params, xhat16), OFF(mico_ln_fwd_params, drop_p), OFF(mico_ln_fwd_params, drop_seed),
        OFF(mico_ln_fwd_params, drop_site), OFF(mico_ln_fwd_params, valid_cols), OFF(mico_ln_fwd_params, q8), OFF(mico_ln_fwd_params, ldq),
        OFF(mico_ln_fwd_params, scales),
        -1,
        (int)sizeof(mico_ln_bwd_params),
        OFF(mico_ln_bwd_params, dy), OFF(mico_ln_bwd_params, dy_dtype), OFF(mico_ln_bwd_params, dy_scale), OFF(mico_ln_bwd_params, x),
        OFF(mico_ln_bwd_params, x_dtype), OFF(mico_ln_bwd_params, x_normalized), OFF(mico_ln_bwd_params, gamma), OFF(mico_ln_bwd_params, mean),
        OFF(mico_ln_bwd_params, rstd), OFF(mico_ln_bwd_params, dx_add), OFF(mico_ln_bwd_params, dx32), OFF(mico_ln_bwd_params, dx16),
        OFF(mico_ln_bwd_params, scale16), OFF(mico_ln_bwd_params, dgamma), OFF(mico_ln_bwd_params, dbeta), OFF(mico_ln_bwd_params, grad_scale),
        OFF(mico_ln_bwd_params, ws), OFF(mico_ln_bwd_params, rows), OFF(mico_ln_bwd_params, cols), OFF(mico_ln_bwd_params, frame_map),
        OFF(mico_ln_bwd_params, rows_per_frame), OFF(mico_ln_bwd_params, valid_cols), OFF(mico_ln_bwd_params, dx16_dst),
        OFF(mico_ln_bwd_params, dx16_frame_scale), OFF(mico_ln_bwd_params, dx16_drop_p), OFF(mico_ln_bwd_params, dx16_drop_seed),
        OFF(mico_ln_bwd_params, dx16_drop_site),
        -1,
    };
#undef OFF
    const int total = (int)(sizeof(t) / sizeof(t[0]));
    for (int i = 0; i < n && i < total; ++i) out[i] = t[i];
    return total;
}

extern "C" int mico_quant_mx8(const void* x, int64_t ld, int64_t rows, int cols, void* q, int64_t ldq, void* scales, float pre_scale,
                              int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype), "mico_quant_mx8: dtype must be MICO_F16 or MICO_BF16");
    MICO_CHECK(x && q && scales && rows > 0 && cols > 0, "mico_quant_mx8: bad args");
    MICO_CHECK(cols % 128 == 0 && ld % 8 == 0 && ldq % 8 == 0 && ld >= cols && ldq >= cols, "mico_quant_mx8: cols must be a multiple of 128, ld / ldq multiples of 8 and >= cols");
    const int chunks = (cols + 511) / 512;
    const int64_t units = rows * chunks;
    const unsigned grid = (unsigned)std::min<int64_t>((units + 3) / 4, 256 * 16);
    DISPATCH_T16(dtype, MICO_LAUNCH((quant_mx8_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, rows, cols,
                                    (unsigned char*)q, ldq, (unsigned*)scales, pre_scale));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_gemm_mx8(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* a_scales, const void* B, int64_t ldb,
                             const void* b_scales, void* C, int64_t ldc, int c_dtype, const mico_gemm_epilogue* epi, int out_dtype, void* stream) {
    MICO_CHECK(dtype_ok(out_dtype), "mico_gemm_mx8: out_dtype (the 16-bit type of C / aux tensors) must be MICO_F16 or MICO_BF16");
    MICO_CHECK(A && B && C && a_scales && b_scales, "mico_gemm_mx8: null operand");
    MICO_CHECK(M > 0 && N > 0 && K > 0 && K % 128 == 0, "mico_gemm_mx8: K must be a positive multiple of 128 (got %lld)", (long long)K);
    MICO_CHECK(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K, "mico_gemm_mx8: lda / ldb must be multiples of 16 bytes and >= K");
    MICO_CHECK(N % 4 == 0 && ldc % 4 == 0, "mico_gemm_mx8: N and ldc must be multiples of 4");
    MICO_CHECK(c_dtype == MICO_F32 || c_dtype == out_dtype, "mico_gemm_mx8: c_dtype must be MICO_F32 or out_dtype");
    if (c_dtype != MICO_F32) MICO_CHECK(ldc % 8 == 0 && ((uintptr_t)C & 15) == 0, "mico_gemm_mx8: a 16-bit C needs ldc %% 8 == 0 and a 16-byte aligned base");
    MICO_CHECK(256 * lda < 0x7FFFFFFFll && 256 * ldb < 0x7FFFFFFFll, "mico_gemm_mx8: leading dimension too large");
    Mx8Args a;
    GemmArgs& g = a.g;
    g.A = (const char*)A; g.B = (const char*)B; g.C = (char*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.c_dtype = c_dtype;
    g.tm0 = 0;
    g.a_wrap = 0;
    g.fast16 = 0;
    if (epi) g.e = *epi;
    else { g.e = mico_gemm_epilogue{}; g.e.alpha = 1.f; }
    MICO_CHECK(g.e.nseg == 0, "mico_gemm_mx8: k-segments are a 16-bit feature");
    MICO_CHECK(!g.e.aux_tiled, "mico_gemm_mx8: the tiled aux layout belongs to the 16-bit MLP pair (its fp8 dX launch reads a row-major aux)");
    MICO_CHECK(g.e.act >= MICO_ACT_NONE && g.e.act <= MICO_ACT_MUL_AUX, "mico_gemm_mx8: unknown act %d", g.e.act);
    if (g.e.act == MICO_ACT_GELU_GRAD || g.e.act == MICO_ACT_MUL_AUX) MICO_CHECK(g.e.aux_in != nullptr, "mico_gemm_mx8: GELU_GRAD / MUL_AUX need aux_in");
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) MICO_CHECK(g.e.aux_out && c_dtype != MICO_F32, "mico_gemm_mx8: GELU_SAVE_DERIV needs aux_out and a 16-bit C");
    if (epi && (epi->aux_out || epi->aux_in)) MICO_CHECK(epi->ldaux % 8 == 0, "mico_gemm_mx8: ldaux must be a multiple of 8");
    if (g.e.row_scale) MICO_CHECK(g.e.rows_per_scale > 0, "mico_gemm_mx8: rows_per_scale must be > 0");
    if (g.e.row_map) MICO_CHECK(g.e.rows_per_map > 0 && !g.e.remap_group, "mico_gemm_mx8: row_map needs rows_per_map > 0 and no remap_group");
    g.ntm = (int)((M + 255) / 256); g.ntn = (int)((N + 255) / 256);
    g.ntiles = g.ntm * g.ntn;
    g.ktiles = (int)(K / 128);
    g.split_k = 1; g.ktiles_per_split = g.ktiles;
    g.ka_rows = g.kb_rows = K;
    a.sa = (const unsigned*)a_scales; a.sb = (const unsigned*)b_scales;
    const dim3 grid(g.ntiles), block(Mx8::THREADS);
    hipStream_t st = (hipStream_t)stream;
    const bool lean = g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
    // the 8-phase form (gemm_p8mx_kernel) for the launches its epilogues cover; variant 16 = the 2-stage first version everywhere (A/B runs)
    const bool p8mx = g_mico_gemm_variant != 16 && g.lda * 256 + K < 0x7FFFFF00ll && g.ldb * 256 + K < 0x7FFFFF00ll &&
                      (lean || g.e.act == MICO_ACT_GELU_SAVE_DERIV || g.e.act == MICO_ACT_MUL_AUX);
    g.tm0 = 0;
    g.a_wrap = 0;
    g.fast16 = p8mx && c_dtype != MICO_F32 && !g.e.row_scale && !g.e.row_map && !g.e.resid && !g.e.accumulate && 256 * ldc * 2 < 0x7FFFFFFFll &&
               (g.e.aux_out || g.e.aux_in ? 256 * g.e.ldaux * 2 < 0x7FFFFFFFll : true);
#define MX8(ACTV) do { if (p8mx) DISPATCH_T16(out_dtype, MICO_LAUNCH((gemm_p8mx_kernel<T, ACTV>), grid, block, 0, st, a)); \
                       else DISPATCH_T16(out_dtype, MICO_LAUNCH((gemm_mx8_kernel<T, ACTV>), grid, block, 0, st, a)); } while (0)
    // the persistent form (gemm_p8pmx_kernel): 16-bit outputs through the 16-bit staged epilogue, several tiles per CU (variant 17, probe build: off)
    const bool persist = MICO_P8_PERSIST && p8mx && g.fast16 && N % 8 == 0 && g.ntiles > 256 && g.ktiles >= 2 && !g.e.aux_in && g_mico_gemm_variant != 17;
    if (persist && g.e.act == MICO_ACT_GELU_SAVE_DERIV) DISPATCH_T16(out_dtype, MICO_LAUNCH((gemm_p8pmx_kernel<T, MICO_ACT_GELU_SAVE_DERIV>), dim3(256), block, 0, st, a));
    else if (persist && lean) DISPATCH_T16(out_dtype, MICO_LAUNCH((gemm_p8pmx_kernel<T, ACT_LEAN>), dim3(256), block, 0, st, a));
    else if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) MX8(MICO_ACT_GELU_SAVE_DERIV);
    else if (g.e.act == MICO_ACT_MUL_AUX) MX8(MICO_ACT_MUL_AUX);
    else if (lean) MX8(ACT_LEAN);
    else DISPATCH_T16(out_dtype, MICO_LAUNCH((gemm_mx8_kernel<T, 0>), grid, block, 0, st, a));
#undef MX8
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

// The element count of a TILED aux tensor for an [M, N] MLP hidden activation (mico_gemm_epilogue::aux_tiled), or 0 when the pair of launches that
// would write (forward orientation, GELU_SAVE_DERIV) and read it (dX orientation, MUL_AUX) - both M x N over a reduction of K (the forward's may
// be 2 K: x W_hi + x W_lo on the wrapped A stream), contiguous operands, 16-bit MFMA - does not run on the persistent 8-phase kernel in this build
// (then the aux tensor is row-major as documented).  Mirrors mico_gemm's routing.
extern "C" int64_t mico_gemm_aux_tiled_elems(int64_t M, int64_t N, int64_t K) {
    static const bool no_wrap = getenv("MICO_P8_NOWRAP") != nullptr;
    if (!MICO_P8_PERSIST || !MICO_P8_DEFAULT || g_mico_gemm_variant != 0 || no_wrap) return 0;
    if (M <= 0 || N <= 0 || K <= 0 || N % 256 != 0 || K % 64 != 0 || K < 128) return 0;
    const int64_t tiles = ((M + 255) / 256) * (N / 256);
    if (tiles <= 256 || tiles >= 0x7FFFFFFFll) return 0;                       // (> 256: several tiles per CU; also mico_gemm's `big`: >= 128 tiles, N >= 192)
    if (256 * N * 2 >= 0x7FFFFFFFll || 256 * (2 * K) * 2 + 2 * K * 2 >= 0x7FFFFF00ll || (K + 64) * N * 2 >= 0x7FFFFF00ll) return 0;   // the 31-bit offsets of fast16 / p8_ok
    return ((M + 255) / 256) * 256 * N;
}

extern "C" int mico_gemm(int ta, int tb, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                         int64_t ldb, void* C, int64_t ldc, int c_dtype, const mico_gemm_epilogue* epi, int split_k,
                         int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype), "mico_gemm: dtype must be MICO_F16 or MICO_BF16");
    MICO_CHECK(A && B && C, "mico_gemm: null operand");
    MICO_CHECK(M > 0 && N > 0 && K > 0, "mico_gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    MICO_CHECK(lda % 8 == 0 && ldb % 8 == 0, "mico_gemm: lda/ldb must be multiples of 8 elements (got %lld, %lld)", (long long)lda, (long long)ldb);
    MICO_CHECK(N % 4 == 0 && ldc % 4 == 0, "mico_gemm: N and ldc must be multiples of 4 (got %lld, %lld)", (long long)N, (long long)ldc);
    if (c_dtype != MICO_F32) MICO_CHECK(ldc % 8 == 0 && ((uintptr_t)C & 15) == 0, "mico_gemm: a 16-bit C needs ldc %% 8 == 0 and a 16-byte aligned base (16-byte stores)");
    if (epi && (epi->aux_out || epi->aux_in)) MICO_CHECK(epi->ldaux % 8 == 0, "mico_gemm: ldaux must be a multiple of 8");
    const bool segs = epi && epi->nseg > 0;
    if (!ta) MICO_CHECK(K % 8 == 0 && (segs || lda >= K), "mico_gemm: A[M,K] needs K %% 8 == 0 and lda >= K");
    else MICO_CHECK(lda >= M, "mico_gemm: A^T[K,M] needs lda >= M");
    if (!tb) MICO_CHECK(K % 8 == 0 && (segs || ldb >= K), "mico_gemm: B[N,K] needs K %% 8 == 0 and ldb >= K");
    else MICO_CHECK(ldb >= N, "mico_gemm: B^T[K,N] needs ldb >= N");
    MICO_CHECK(c_dtype == MICO_F32 || c_dtype == dtype, "mico_gemm: c_dtype must be MICO_F32 or dtype");
    MICO_CHECK(256 * lda * 2 < 0x7FFFFFFFll && 256 * ldb * 2 < 0x7FFFFFFFll, "mico_gemm: leading dimension too large");
    // a reduction-major operand is addressed by 32-bit byte offsets from the first row of its tile column (buffer descriptors): K rows of it must
    // stay below 4 GiB - larger reductions are issued as accumulating launches over row chunks (functional.CrossKVFn.backward does)
    if (ta) MICO_CHECK(K * lda * 2 <= 0xFFFFFF00ll, "mico_gemm: A^T[K,M] spans %lld bytes (K * lda * 2); reduction-major operands are limited to 4 GiB per launch", (long long)(K * lda * 2));
    if (tb) MICO_CHECK(K * ldb * 2 <= 0xFFFFFF00ll, "mico_gemm: B^T[K,N] spans %lld bytes (K * ldb * 2); reduction-major operands are limited to 4 GiB per launch", (long long)(K * ldb * 2));
    GemmArgs g;
    g.A = (const char*)A; g.B = (const char*)B; g.C = (char*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.c_dtype = c_dtype;
    g.tm0 = 0;
    g.a_wrap = 0;
    g.fast16 = 0;
    if (epi) g.e = *epi;
    else {
        g.e = mico_gemm_epilogue{};
        g.e.alpha = 1.f;
    }
    // tile configuration: the 256x256 ping-pong kernel whenever it yields enough workgroups to occupy most CUs
    const int64_t big_tiles = ((M + 255) / 256) * ((N + 255) / 256);
    // weight-gradient GEMMs (fp32 accumulate, auto split) have few output tiles but a very long reduction: split-K supplies
    // the parallelism, so the big tile pays as soon as K is long
    // (at least 16 big tiles: BERT's 768 x 768 layers over the step's ~15 k text rows do no better on the 192x256 tiles, whose fp32
    // atomics then outweigh their K loops, than on the small-tile kernel with small_acc_split(); in situ 156 vs 160 us)
    const bool long_k_acc = c_dtype == MICO_F32 && g.e.accumulate && K >= 8192 && big_tiles >= 16;
    const bool big = (big_tiles >= 128 || long_k_acc) && N >= 192;
    // the producer/consumer kernel (192x256) takes every large problem; MICO_GEMM_NO_PC (ablation builds) keeps the 8-wave kernel
#ifdef MICO_GEMM_NO_PC
    bool pc = false;
#else
    // measured (tools/gemm_bench.py, ViT-g/14 shapes): the producer/consumer kernel wins for the weight-gradient orientation
    // (both operands reduction-major, long K per workgroup: 911 vs 730 TFLOP/s) and loses for the short-K forward / dX GEMMs
    // (742-866 vs 946-960), whose per-tile prologue + epilogue cost weighs more on the smaller 192x256 tile
#ifdef MICO_GEMM_PC_ALL
    bool pc = big;
#else
    // (its epilogue is the lean instantiation: launches with an aux copy / activation / dropout / pos / remap stay on the 8-wave kernel)
    bool pc = big && ta && tb && g.e.act == MICO_ACT_NONE && !g.e.aux_out && !g.e.aux_in && g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group;
#endif
#endif
    // the one-wave-per-SIMD kernel has no masked K-tail path: a k-contiguous operand needs K % 64 == 0 (reduction-major ones end at
    // their descriptor bound); its DMA offsets are 31-bit (operand extent reachable from a tile's base < 2 GiB)
    const int64_t a_ext = ta ? (K + 64) * lda * 2 : 256 * lda * 2 + K * 2, b_ext = tb ? (K + 64) * ldb * 2 : 256 * ldb * 2 + K * 2;
#ifdef MICO_GEMM_W4
    const bool w4_built = true;
#else
    const bool w4_built = false;
#endif
    const bool w4 = w4_built && big && g_mico_gemm_variant >= 2 && (ta || K % 64 == 0) && (tb || K % 64 == 0) && a_ext < 0x7FFFFF00ll && b_ext < 0x7FFFFF00ll;
    if (w4) pc = false;
    // the 256x128 two-workgroups-per-CU kernel: forward / dX orientation, no split (variant 5: every such problem, 6: K <= 2048 only, 7: off)
    // (an fp32-accumulate launch with an explicit split_k = 1 - the condition-token gradients summed over BERT's layers - is one pass too: its
    // epilogue adds to C; only the automatic choice (split_k <= 0) of an accumulating launch may split)
    const bool no_split = split_k == 1 || (split_k <= 0 && !(c_dtype == MICO_F32 && g.e.accumulate));
    const bool mid = big && !pc && !w4 && !ta && no_split &&
                     (g_mico_gemm_variant == 5 || (g_mico_gemm_variant == 6 && K <= 2048));
    // the 64-deep unit-ring form of it (variant 8: every such problem, 9: K <= 2048 only)
    // Default routing (variant 0), from in-situ A/B runs of the timed step (bench.py --gemm-detail, MICO_GEMM_VARIANT=8 against 0): the launches
    // whose epilogue is heavy next to a short K loop gain from the second workgroup on the CU - fc1 forward with the GELU pair (two 16-bit
    // outputs: 845 -> 870 TFLOP/s) and the output projection's fp32 residual scatter at K = 1408 (727 -> 752 in the microbench); everywhere else
    // the two kernels are within +-2 % of each other and the 8-wave kernel keeps the launch.
    const bool mid_default = (g_mico_gemm_variant == 0 || g_mico_gemm_variant == 11 || g_mico_gemm_variant == 16) && M >= 8192 &&
                             (g.e.act == MICO_ACT_GELU_SAVE_DERIV || (g.e.resid != nullptr && c_dtype == MICO_F32 && K <= 2048));
    // the 8-phase 256x256x64 kernel: forward / dX orientation, no split, K % 64 == 0 (variant 10: every such problem, 11: those the
    // 256x128 kernel does not take by default, 12: off)
    // x W_hi + x W_lo (the head-split blocks' weights-split forward: two k-segments, the activation repeated, the weight halves adjacent) is
    // one product of K = 2 kseg for the 8-phase kernel whose A stream wraps (GemmArgs::a_wrap); MICO_P8_NOWRAP=1: the 32-deep k-segment kernels
    static const bool no_wrap = getenv("MICO_P8_NOWRAP") != nullptr;
    const bool wrap_ok = !no_wrap && g.e.nseg == 2 && !ta && !tb && g.e.kseg % 64 == 0 && g.e.a_seg_off[0] == 0 && g.e.a_seg_off[1] == 0 &&
                         g.e.b_seg_off[0] == 0 && g.e.b_seg_off[1] == g.e.kseg;
    const bool p8_ok = big && !pc && !w4 && !ta && no_split && N % 4 == 0 && K % 64 == 0 && (g.e.nseg == 0 || wrap_ok) &&
                       256 * lda * 2 + K * 2 < 0x7FFFFF00ll && (tb ? (K + 64) * ldb * 2 : 256 * ldb * 2 + K * 2) < 0x7FFFFF00ll;
    const bool p8 = p8_ok && (g_mico_gemm_variant == 10 || g_mico_gemm_variant == 13 || g_mico_gemm_variant == 14 || g_mico_gemm_variant == 15 || (g_mico_gemm_variant == 11 && !mid_default) ||
                              ((g_mico_gemm_variant == 0 || g_mico_gemm_variant == 16) && MICO_P8_DEFAULT));      // (16: default routing with the persistent form off)
    const bool mid64 = big && !pc && !w4 && !p8 && !ta && no_split && N % 4 == 0 && K % 64 == 0 && (g.e.nseg == 0 || g.e.kseg % 64 == 0) &&
                       (g_mico_gemm_variant == 8 || (g_mico_gemm_variant == 9 && K <= 2048) || mid_default);
    const int BM = pc ? Wide<32>::BM : (big ? 256 : 128), BN = (mid || mid64) ? 128 : (big ? 256 : 128);
    const int slots = big && !mid && !mid64 ? 256 : 512;
    g.ntm = (int)((M + BM - 1) / BM); g.ntn = (int)((N + BN - 1) / BN);
    g.ntiles = g.ntm * g.ntn;
    const bool deep = g_mico_gemm_variant == 3;
    const int BKc = w4 ? (deep ? 32 : 64) : pc ? pc_bk(ta, tb) : p8 ? P8C::BK : mid64 ? Mid64::BK : mid ? Mid::BK : (big ? Big::BK : Small::BK);
    g.ktiles = (int)((K + BKc - 1) / BKc);
    if (split_k <= 0) {
        if (!(c_dtype == MICO_F32 && g.e.accumulate)) split_k = 1;
        else if (!big && g.ntiles <= 512 && g.e.splitk_ws && g.e.splitk_ws_bytes >= 2 * M * N * 4 && N % 4 == 0)
            split_k = small_slab_split(g.ntiles, g.ktiles, (double)M * N, (int)std::min<int64_t>(32, g.e.splitk_ws_bytes / (M * N * 4)));
        else if (!big && g.ntiles <= 512) split_k = small_acc_split(g.ntiles, g.ktiles);
        // fixed cost of one more wave of workgroups, in K-tiles: fitted on split sweeps of the ViT-g/14 weight-gradient shapes
        // (tools/gemm_bench.py --split-k; 0.59 us per 32-deep K-tile, ~100 us per wave of 256 atomic epilogues = 170 K-tiles - the old
        // 24 took the qkv gradient to 9 splits at 664 TFLOP/s where 3 splits run at 815)
        else if (pc && g.e.splitk_ws && g.e.splitk_ws_bytes >= 2 * M * N * 4) {
            // slab path (mico_gemm_epilogue::splitk_ws): a wave of plain-store epilogues + pipeline fill ~ MICO_SLAB_FIXED K-tiles, the
            // reduction pass ~ M * N * 4 bytes per split at ~4 TB/s (0.59 us per K-tile)
            const int max_s = (int)std::min<int64_t>(32, g.e.splitk_ws_bytes / (M * N * 4));
            split_k = auto_split(g.ntiles, g.ktiles, slots, 1024 / BKc, MICO_SLAB_FIXED, max_s, (int)(M * N / 590000) + 8);
        }
        else split_k = auto_split(g.ntiles, g.ktiles, slots, big ? 1024 / BKc : 16, big ? 5440 / BKc : 12);
    }
    if (split_k > g.ktiles) split_k = g.ktiles;
    g.ktiles_per_split = (g.ktiles + split_k - 1) / split_k;
    split_k = (g.ktiles + g.ktiles_per_split - 1) / g.ktiles_per_split;
    g.split_k = split_k;
    // the slab path needs the producer/consumer kernel, a real split and room for every split's slab; otherwise atomics
    // (splitk_reduce_kernel reads and writes C with 16-byte accesses: a C that is only 4-byte aligned stays on the atomics path)
    if (!((pc || !big) && split_k > 1 && g.e.splitk_ws && g.e.splitk_ws_bytes >= (int64_t)split_k * M * N * 4 && N % 4 == 0 &&
          ((uintptr_t)C & 15) == 0 && ((uintptr_t)g.e.splitk_ws & 15) == 0))
        g.e.splitk_ws = nullptr;
    g.ka_rows = g.kb_rows = K;
    g.group_m = g_mico_mid_group > 0 ? g_mico_mid_group : MICO_MID_GROUP_M;
    if (g.e.nseg > 0) {
        MICO_CHECK(g.e.nseg <= 3 && g.e.kseg > 0 && g.e.kseg % 64 == 0 && (int64_t)g.e.nseg * g.e.kseg == K,
                   "mico_gemm: k-segments need nseg <= 3, kseg %% 64 == 0 and nseg * kseg == K");
        g.ka_rows = g.kb_rows = 0;
        for (int i = 0; i < g.e.nseg; ++i) {
            MICO_CHECK(g.e.a_seg_off[i] % 8 == 0 && g.e.b_seg_off[i] % 8 == 0 && g.e.a_seg_off[i] >= 0 && g.e.b_seg_off[i] >= 0, "mico_gemm: bad segment offset");
            if (g.e.a_seg_off[i] + g.e.kseg > g.ka_rows) g.ka_rows = g.e.a_seg_off[i] + g.e.kseg;
            if (g.e.b_seg_off[i] + g.e.kseg > g.kb_rows) g.kb_rows = g.e.b_seg_off[i] + g.e.kseg;
        }
    }
    if (g.split_k > 1) {
        MICO_CHECK(c_dtype == MICO_F32 && g.e.accumulate, "mico_gemm: split_k > 1 needs fp32 accumulate output");
        MICO_CHECK(!g.e.bias && !g.e.aux_out && g.e.act == MICO_ACT_NONE && !g.e.row_scale && !g.e.resid && !g.e.pos && !g.e.remap_group && !g.e.row_map && g.e.drop_p == 0.f,
                   "mico_gemm: split_k > 1 supports only the alpha-scaled accumulate epilogue");
    }
    if (g.e.act == MICO_ACT_GELU_GRAD || g.e.act == MICO_ACT_MUL_AUX) MICO_CHECK(g.e.aux_in != nullptr, "mico_gemm: GELU_GRAD / MUL_AUX need aux_in");
    if (g.e.act == MICO_ACT_GELU_GRAD && g.e.aux_out) MICO_CHECK(c_dtype != MICO_F32 && g.e.aux_out != g.e.aux_in, "mico_gemm: GELU_GRAD with aux_out (= gelu(aux_in)) needs a 16-bit C and a buffer of its own");
    MICO_CHECK(g.e.act >= MICO_ACT_NONE && g.e.act <= MICO_ACT_MUL_AUX, "mico_gemm: unknown act %d", g.e.act);
    // the MLP pair exists as dedicated instantiations only: forward orientation / dX orientation, no split-K, 16-bit output
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV) MICO_CHECK(!ta && !tb && g.e.aux_out && c_dtype != MICO_F32, "mico_gemm: GELU_SAVE_DERIV is the forward epilogue (ta = tb = 0, aux_out, 16-bit C)");
    if (g.e.act == MICO_ACT_MUL_AUX) MICO_CHECK(!ta && tb, "mico_gemm: MUL_AUX is the dX epilogue (ta = 0, tb = 1)");
    if (g.e.act == MICO_ACT_GELU_SAVE_DERIV || g.e.act == MICO_ACT_MUL_AUX)
        MICO_CHECK(g.e.drop_p == 0.f && !g.e.pos && !g.e.remap_group, "mico_gemm: GELU_SAVE_DERIV / MUL_AUX do not combine with dropout, pos or remap");
    if (g.e.row_scale) MICO_CHECK(g.e.rows_per_scale > 0, "mico_gemm: rows_per_scale must be > 0");
    if (g.e.row_map) MICO_CHECK(g.e.rows_per_map > 0 && !g.e.remap_group, "mico_gemm: row_map needs rows_per_map > 0 and no remap_group");
    if (g.e.pos) MICO_CHECK(g.e.pos_rows > 0, "mico_gemm: pos_rows must be > 0");
    MICO_CHECK(g.e.drop_p >= 0.f && g.e.drop_p < 1.f, "mico_gemm: drop_p must be in [0, 1)");
    hipStream_t st = (hipStream_t)stream;
    if (g.e.colsum_out) {
        MICO_CHECK(ta && tb && c_dtype == MICO_F32, "mico_gemm: colsum_out belongs to the weight-gradient orientation (ta = tb = 1, fp32 C)");
        if (!pc) {   // not the kernel that takes it along: the stand-alone column-sum pass, same result
            const int rc = mico_colsum(A, dtype, lda, K, (int)M, g.e.colsum_out, g.e.alpha, 1, stream);
            if (rc != MICO_OK) return rc;
            g.e.colsum_out = nullptr;
        }
    }
    g_mico_last_gemm_kernel = w4 ? 3 : pc ? 2 : (big ? 1 : 0);
    // (checked in front of the routing chain: between the W4 build's dangling `else` and `if (pc)` it would become that else's body - ADVICE r5)
    if (g.e.aux_tiled) MICO_CHECK(p8 && !pc && !w4, "mico_gemm: aux_tiled is the layout of the persistent 8-phase kernel's MLP pair (ask mico_gemm_aux_tiled_elems first)");
#ifdef MICO_GEMM_W4
    if (w4 && deep) DISPATCH_T16(dtype, (launch_w4<T, 1>(ta, tb, g, st)));
    else if (w4) DISPATCH_T16(dtype, (launch_w4<T, 0>(ta, tb, g, st)));
    else
#endif
    if (pc) DISPATCH_T16(dtype, (launch_pc<T>(ta, tb, g, st)));
    else if (p8) {
        g_mico_last_gemm_kernel = 8;
        g.a_wrap = wrap_ok ? g.e.kseg / 64 : 0;
        g.fast16 = c_dtype != MICO_F32 && !g.e.row_scale && !g.e.row_map && !g.e.resid && !g.e.accumulate && !g.e.pos && !g.e.remap_group && g.e.drop_p == 0.f &&
                   (g.e.act == MICO_ACT_NONE ? (!g.e.aux_out && !g.e.aux_in) : true) && 256 * ldc * 2 < 0x7FFFFFFFll &&
                   (g.e.aux_out || g.e.aux_in ? 256 * g.e.ldaux * 2 < 0x7FFFFFFFll : true) && g_mico_gemm_variant != 15;
        if (g.e.aux_tiled)      // exactly launch_p8p's conditions for the two tiled instantiations
            MICO_CHECK(MICO_P8_PERSIST && g_mico_gemm_variant != 16 && g.fast16 && N % 256 == 0 && g.ntiles > 256 && g.ktiles >= 2 &&
                           ((g.e.act == MICO_ACT_GELU_SAVE_DERIV && !tb && g.e.aux_out) || (g.e.act == MICO_ACT_MUL_AUX && tb && g.e.aux_in && !g.e.aux_out) ||
                            (g.e.act == MICO_ACT_GELU && !tb && g.e.aux_out && !g.e.aux_in) || (g.e.act == MICO_ACT_GELU_GRAD && tb && g.e.aux_in && g.e.aux_out && !g.e.bias)),
                       "mico_gemm: aux_tiled is the layout of the persistent 8-phase kernel's MLP pair (ask mico_gemm_aux_tiled_elems first)");
        // Round quantisation: T tiles on 256 CUs take ceil(T / 256) rounds and the towers' N = 1408 launches have only ~6 (M = kept frames x 257
        // rows: 257 row tiles x 6 = 6.02 rounds is SEVEN).  The row tiles beyond the last full round can go to the 256x128 two-workgroups-per-CU
        // kernel instead: the same rows as <= 512 half-size tiles in one pass (cost model below; MICO_P8_SPLIT in the build or variant 14).
        int ntm1 = g.ntm;
        if (((MICO_P8_SPLIT && g_mico_gemm_variant != 13) || g_mico_gemm_variant == 14) && !g.e.aux_tiled) {      // (a tiled aux tensor belongs to ONE persistent launch)
            const int ntn128 = (int)((N + 127) / 128);
            auto cost = [&](int rows_big) {
                const long big_tiles = (long)rows_big * g.ntn, small_tiles = (long)(g.ntm - rows_big) * ntn128;
                return (double)((big_tiles + 255) / 256) + 0.58 * (double)((small_tiles + 511) / 512);
            };
            double best = cost(g.ntm);
            const int full = g.ntiles / 256;
            for (int k = full; k >= 1 && k >= full - 1; --k) {
                const int rows_big = (int)((long)k * 256 / g.ntn);
                if (rows_big <= 0 || rows_big >= g.ntm) continue;
                const double c = cost(rows_big);
                if (c < best - 0.05) { best = c; ntm1 = rows_big; }
            }
        }
        if (ntm1 < g.ntm) {
            GemmArgs g2 = g;
            g2.tm0 = ntm1;
            g2.ntm = g.ntm - ntm1;
            g2.ntn = (int)((N + 127) / 128);
            g2.ntiles = g2.ntm * g2.ntn;
            g.ntm = ntm1;
            g.ntiles = g.ntm * g.ntn;
            DISPATCH_T16(dtype, (launch_p8<T>(tb, g, st, g_mico_gemm_variant != 16)));
            DISPATCH_T16(dtype, (launch_mid<T>(tb, g2, st)));
        } else DISPATCH_T16(dtype, (launch_p8<T>(tb, g, st, g_mico_gemm_variant != 16)));
    }
    else if (mid64) { g_mico_last_gemm_kernel = 7; DISPATCH_T16(dtype, (launch_mid<T>(tb, g, st))); }
#ifdef MICO_GEMM_VARIANTS
    else if (mid) { g_mico_last_gemm_kernel = 6; DISPATCH_T16(dtype, (launch<T, Mid>(ta, tb, g, st))); }
#endif
    else if (big) {
        // persistent form when every CU gets several tiles and nothing is split (variant 4 forces it off, for A/B runs)
        bool done = false;
        if (g.split_k == 1 && g.ntiles >= 512 && g.e.nseg == 0 && K % 32 == 0 && g_mico_gemm_variant != 4) DISPATCH_T16(dtype, (done = launch_persist<T>(ta, tb, g, st)));
        if (done) g_mico_last_gemm_kernel = 5;
        else DISPATCH_T16(dtype, (launch<T, Big>(ta, tb, g, st)));
    }
    else DISPATCH_T16(dtype, (launch<T, Small>(ta, tb, g, st)));
    if (g.e.splitk_ws) {   // split-K slab path: add the slabs into C
        MICO_LAUNCH_CHECK();
        const int64_t total = M * (N / 4);
        MICO_LAUNCH(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, st, (const float*)g.e.splitk_ws,
                    g.split_k, M, N, (float*)C, ldc, g.e.alpha);
    }
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
