// MFMA GEMM for gfx950 with fused epilogues (see include/mico_hip.h: mico_gemm).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16x32 tiles),
// BK = 64.  Operand tiles are brought HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B/lane, no VGPR round trip)
// into a 2-stage ring; the buffer descriptor's bounds check zero-fills rows past the end of the matrix, so M/N/K
// tails need no masking code in the main loop.  LDS images are lane-linear (DMA constraint); bank conflicts are
// removed by XOR-swizzling the 16-byte chunk index on the *source* address and applying the same involution on the
// read side.  K-contiguous operands are read with ds_read_b128, reduction-major operands (dX / dW GEMMs) with the
// gfx950 transposing read ds_read_b64_tr_b16, so the backward GEMMs need no transposed copies of weights or
// activations in HBM.  MFMA operands are swapped (D^T = B A^T) so every lane owns 4 consecutive columns of one
// output row: 8/16-byte epilogue accesses.  Workgroup ids are remapped XCD-contiguously (8 private L2s) and walk the
// tile grid in groups of 8 row-panels so concurrently resident tiles share A and B panels in L2.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;   // 16 KiB per operand tile, either orientation
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int GROUP_M = 8;

struct GemmArgs {
    const char* A;
    const char* B;
    char* C;
    int64_t M, N, K, lda, ldb, ldc;
    int ntm, ntn, ntiles, split_k, ktiles, ktiles_per_split;
    int c_dtype;
    int64_t ka_rows, kb_rows;   // physical reduction extents of A / B (differ from K in k-segment mode)
    mico_gemm_epilogue e;
};

// swizzle keys (16-byte chunk index XOR) - see file header
__device__ __forceinline__ int key_kc(int row) { return (row >> 1) & 7; }                               // [128][64] k-contiguous
__device__ __forceinline__ int key_tr(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }   // [64][128] reduction-major

template <bool TR>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, LDS_AS char* lds_tile, int wave, int lane,
                                           int64_t ld_bytes, int k0, int64_t kdim, int64_t cdim_rem) {
    // one tile = 1024 chunks of 16 B; 4 DMA instructions per thread
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 256 + wave * 64 + lane;
        unsigned voff;
        if (!TR) {
            const int row = c >> 3, cpos = c & 7;
            const int cg = cpos ^ key_kc(row);
            const int k = k0 + cg * 8;
            voff = (unsigned)(row * ld_bytes + (int64_t)k * 2);
            if (k >= kdim) voff = 0xFFFFFFF0u;   // K tail: force out-of-bounds -> zero fill
        } else {
            const int row = c >> 4, cpos = c & 15;
            const int cg = cpos ^ key_tr(row);
            voff = (unsigned)((int64_t)(k0 + row) * ld_bytes + cg * 16);
            if (cg * 8 >= cdim_rem || (k0 + row) >= kdim) voff = 0xFFFFFFF0u;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds_tile + (it * 256 + wave * 64) * 16), 16, voff, 0, 0, 0);
    }
}

// MFMA fragment (8 x 16-bit along the reduction dim) for 16 tile rows starting at `r0`, k-step kk (0/1)
template <bool TR>
__device__ __forceinline__ s16x8 read_frag(LDS_AS const char* tile, int r0, int kk, int lane) {
    if (!TR) {
        const int row = r0 + (lane & 15);
        const int cg = kk * 4 + (lane >> 4);
        return *(LDS_AS const s16x8*)(tile + row * 128 + ((cg ^ key_kc(row)) << 4));
    } else {
        const int kb = kk * 32 + (lane >> 4) * 8 + ((lane & 15) >> 2);
        const int chunk = (r0 >> 3) + ((lane >> 1) & 1);
        const int half = (lane & 1) * 8;
        const int k0r = kb, k1r = kb + 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + k0r * 256 + ((chunk ^ key_tr(k0r)) << 4) + half));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + k1r * 256 + ((chunk ^ key_tr(k1r)) << 4) + half));
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
}

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    LDS_AS char* lds = (LDS_AS char*)smem;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

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
    const int kt0 = ks * g.ktiles_per_split;
    const int kt1 = min(g.ktiles, kt0 + g.ktiles_per_split);

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

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // logical k-tile -> physical k offsets of A and B.  With k-segments (split-precision GEMMs) the logical reduction is
    // the concatenation of nseg segments of kseg elements, each mapped to its own physical column offset per operand.
    const int nseg = g.e.nseg, kseg = g.e.kseg;
    auto kmap = [&](int kt, int& ka, int& kb, int64_t& kda, int64_t& kdb) {
        const int k0 = kt * BK;
        if (nseg > 0) {
            const int sg = k0 / kseg, kin = k0 - sg * kseg;
            ka = g.e.a_seg_off[sg] + kin;
            kb = g.e.b_seg_off[sg] + kin;
            kda = g.e.a_seg_off[sg] + kseg;
            kdb = g.e.b_seg_off[sg] + kseg;
        } else {
            ka = kb = k0;
            kda = kdb = g.K;
        }
    };
    int cur = 0;
    if (kt0 < kt1) {
        int ka, kb;
        int64_t kda, kdb;
        kmap(kt0, ka, kb, kda, kdb);
        stage_tile<TA>(rsa, lds, wave, lane, lda_b, ka, kda, a_crem);
        stage_tile<TB>(rsb, lds + TILE_BYTES, wave, lane, ldb_b, kb, kdb, b_crem);
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();   // stage `cur` landed (vmcnt(0) precedes the barrier); stage cur^1 no longer being read
        if (kt + 1 < kt1) {
            LDS_AS char* nxt = lds + (cur ^ 1) * STAGE_BYTES;
            int ka, kb;
            int64_t kda, kdb;
            kmap(kt + 1, ka, kb, kda, kdb);
            stage_tile<TA>(rsa, nxt, wave, lane, lda_b, ka, kda, a_crem);
            stage_tile<TB>(rsb, nxt + TILE_BYTES, wave, lane, ldb_b, kb, kdb, b_crem);
        }
        LDS_AS const char* ta = lds + cur * STAGE_BYTES;
        LDS_AS const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            s16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<TA>(ta, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<TB>(tb, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T16<T>::mfma(fb[j], fa[i], acc[i][j]);
        }
        cur ^= 1;
    }

    // ---- epilogue: lane owns C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + 0..3] ----
    const mico_gemm_epilogue& e = g.e;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= g.M) continue;
        int64_t mo = m;
        if (e.remap_group) mo = m + (m / e.remap_group) * e.remap_skip + e.remap_offset;
        float rscale = 1.f;
        if (e.row_scale) rscale = e.row_scale[m / e.rows_per_scale];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[i][j] * e.alpha;
            if (e.bias) v += *(const f32x4*)(e.bias + n);
            if (e.aux_out) *(s16x4*)((T*)e.aux_out + m * e.ldaux + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            if (e.act == MICO_ACT_GELU) {
                v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]);
            } else if (e.act == MICO_ACT_GELU_GRAD) {
                f32x4 h = unpack4<T>(*(const s16x4*)((const T*)e.aux_in + m * e.ldaux + n));
                v[0] *= gelu_grad_f(h[0]); v[1] *= gelu_grad_f(h[1]); v[2] *= gelu_grad_f(h[2]); v[3] *= gelu_grad_f(h[3]);
            }
            v *= rscale;
            if (e.pos) v += *(const f32x4*)(e.pos + (mo % e.pos_rows) * g.N + n);
            if (e.resid) v += *(const f32x4*)(e.resid + mo * g.ldc + n);
            if (g.c_dtype == MICO_F32) {
                float* cp = (float*)g.C + mo * g.ldc + n;
                if (e.accumulate) {
                    if (g.split_k > 1) {
                        unsafeAtomicAdd(cp + 0, v[0]); unsafeAtomicAdd(cp + 1, v[1]);
                        unsafeAtomicAdd(cp + 2, v[2]); unsafeAtomicAdd(cp + 3, v[3]);
                    } else {
                        *(f32x4*)cp += v;
                    }
                } else {
                    *(f32x4*)cp = v;
                }
            } else {
                *(s16x4*)((T*)g.C + mo * g.ldc + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <typename T>
int launch(int ta, int tb, const GemmArgs& g, hipStream_t st) {
    const dim3 grid(g.ntiles * g.split_k), block(256);
    if (!ta && !tb) MICO_LAUNCH((gemm_kernel<T, false, false>), grid, block, 0, st, g);
    else if (!ta && tb) MICO_LAUNCH((gemm_kernel<T, false, true>), grid, block, 0, st, g);
    else if (ta && tb) MICO_LAUNCH((gemm_kernel<T, true, true>), grid, block, 0, st, g);
    else MICO_LAUNCH((gemm_kernel<T, true, false>), grid, block, 0, st, g);
    return 0;
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

extern "C" int mico_version(void) { return 100; }
extern "C" const char* mico_last_error_string(void) { return g_mico_err; }

extern "C" int mico_gemm(int ta, int tb, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                         int64_t ldb, void* C, int64_t ldc, int c_dtype, const mico_gemm_epilogue* epi, int split_k,
                         int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype), "mico_gemm: dtype must be MICO_F16 or MICO_BF16");
    MICO_CHECK(A && B && C, "mico_gemm: null operand");
    MICO_CHECK(M > 0 && N > 0 && K > 0, "mico_gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    MICO_CHECK(lda % 8 == 0 && ldb % 8 == 0, "mico_gemm: lda/ldb must be multiples of 8 elements (got %lld, %lld)", (long long)lda, (long long)ldb);
    MICO_CHECK(N % 4 == 0 && ldc % 4 == 0, "mico_gemm: N and ldc must be multiples of 4 (got %lld, %lld)", (long long)N, (long long)ldc);
    const bool segs = epi && epi->nseg > 0;
    if (!ta) MICO_CHECK(K % 8 == 0 && (segs || lda >= K), "mico_gemm: A[M,K] needs K %% 8 == 0 and lda >= K");
    else MICO_CHECK(lda >= M, "mico_gemm: A^T[K,M] needs lda >= M");
    if (!tb) MICO_CHECK(K % 8 == 0 && (segs || ldb >= K), "mico_gemm: B[N,K] needs K %% 8 == 0 and ldb >= K");
    else MICO_CHECK(ldb >= N, "mico_gemm: B^T[K,N] needs ldb >= N");
    MICO_CHECK(c_dtype == MICO_F32 || c_dtype == dtype, "mico_gemm: c_dtype must be MICO_F32 or dtype");
    MICO_CHECK(128 * lda * 2 < 0x7FFFFFFFll && 128 * ldb * 2 < 0x7FFFFFFFll, "mico_gemm: leading dimension too large");
    if (split_k < 1) split_k = 1;
    GemmArgs g;
    g.A = (const char*)A; g.B = (const char*)B; g.C = (char*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.ntm = (int)((M + BM - 1) / BM); g.ntn = (int)((N + BN - 1) / BN);
    g.ntiles = g.ntm * g.ntn;
    g.ktiles = (int)((K + BK - 1) / BK);
    if (split_k > g.ktiles) split_k = g.ktiles;
    g.ktiles_per_split = (g.ktiles + split_k - 1) / split_k;
    split_k = (g.ktiles + g.ktiles_per_split - 1) / g.ktiles_per_split;
    g.split_k = split_k;
    g.c_dtype = c_dtype;
    if (epi) g.e = *epi;
    else {
        g.e = mico_gemm_epilogue{};
        g.e.alpha = 1.f;
    }
    g.ka_rows = g.kb_rows = K;
    if (g.e.nseg > 0) {
        MICO_CHECK(g.e.nseg <= 3 && g.e.kseg > 0 && g.e.kseg % BK == 0 && (int64_t)g.e.nseg * g.e.kseg == K,
                   "mico_gemm: k-segments need nseg <= 3, kseg %% 64 == 0 and nseg * kseg == K");
        g.ka_rows = g.kb_rows = 0;
        for (int i = 0; i < g.e.nseg; ++i) {
            MICO_CHECK(g.e.a_seg_off[i] % 8 == 0 && g.e.b_seg_off[i] % 8 == 0 && g.e.a_seg_off[i] >= 0 && g.e.b_seg_off[i] >= 0, "mico_gemm: bad segment offset");
            if (g.e.a_seg_off[i] + g.e.kseg > g.ka_rows) g.ka_rows = g.e.a_seg_off[i] + g.e.kseg;
            if (g.e.b_seg_off[i] + g.e.kseg > g.kb_rows) g.kb_rows = g.e.b_seg_off[i] + g.e.kseg;
        }
    }
    if (g.split_k > 1) MICO_CHECK(c_dtype == MICO_F32 && g.e.accumulate, "mico_gemm: split_k > 1 needs fp32 accumulate output");
    if (g.e.act == MICO_ACT_GELU_GRAD) MICO_CHECK(g.e.aux_in != nullptr, "mico_gemm: GELU_GRAD needs aux_in");
    if (g.e.row_scale) MICO_CHECK(g.e.rows_per_scale > 0, "mico_gemm: rows_per_scale must be > 0");
    if (g.e.pos) MICO_CHECK(g.e.pos_rows > 0, "mico_gemm: pos_rows must be > 0");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T16(dtype, launch<T>(ta, tb, g, st));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
