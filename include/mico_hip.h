/* libmico_hip.so - C-ABI of the MI355X (gfx950) kernels behind the MiCo omni-modal forward/backward hot path.
 *
 * The reference (invictus717/MiCo) has no FFI/operator layer: its hot path is PyTorch eager op sequences
 * (SURVEY.md section 2.2).  Each entry point below replaces one such sequence; the reference site it replaces is
 * cited as file:line relative to the reference root.  Conventions:
 *   - raw device pointers, caller owns all memory (incl. workspaces); plain ints/floats; no torch types;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates, never syncs;
 *   - returns 0 on success, a negative MICO_E* code otherwise; mico_last_error_string() describes the failure;
 *   - `dtype` selects the 16-bit MFMA element type for activations / weights (MICO_F16 | MICO_BF16); the
 *     residual stream, LayerNorm statistics, softmax, losses, parameters' gradients are fp32.
 *   - matrices are row-major with explicit leading dimensions given in ELEMENTS.
 */
#ifndef MICO_HIP_H
#define MICO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MICO_F16 0
#define MICO_BF16 1
#define MICO_F32 2

#define MICO_OK 0
#define MICO_EINVAL (-22)
#define MICO_ELAUNCH (-5)

/* library identity / errors */
int mico_version(void);
const char* mico_last_error_string(void);
/* Layout of the parameter structs as THIS library was compiled, for bindings to verify theirs (tests/test_host_cpu.py compares the
 * ctypes mirror field by field): writes up to n ints - sizeof(mico_gemm_epilogue), then the byte offset of each of its fields in
 * declaration order, then -1, then the same for mico_attn_params, mico_ln_fwd_params and mico_ln_bwd_params (each: sizeof, field offsets, -1).
 * Returns the number of ints the full table has. */
int mico_struct_layout(int* out, int n);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM on MFMA (v_mfma_f32_16x16x32_{f16,bf16}), fp32 accumulate, fused epilogues.
 *   C[M,N] = epilogue( sum_k opA(A)[m,k] * opB(B)[k,n] )
 *   ta == 0: A is [M,K] (lda >= K)      ta == 1: A is stored [K,M] (lda >= M)   (reduction-major, read transposed)
 *   tb == 0: B is [N,K] (ldb >= K)      tb == 1: B is stored [K,N] (ldb >= N)
 * so nn.Linear forward  y = x W^T        is (ta=0,tb=0, A=x, B=W);                  eva_vit_model.py:191,197,310,363
 *    its input gradient  dx = dy W        is (ta=0,tb=1, A=dy, B=W);
 *    its weight gradient dW = dy^T x      is (ta=1,tb=1, A=dy, B=x) with M=out_features, N=in_features.
 * All leading dimensions and K-contiguous extents must be multiples of 8 elements (16-byte rows).
 * split_k: 1 = none; > 1 = explicit (fp32 accumulate output only, partial tiles combined with atomics); <= 0 = automatic
 * (sized so the resident workgroups are filled in whole waves).
 *
 * Epilogue (applied per element v = alpha * acc, in this order):
 *   v += bias[n]                              (bias  != NULL, fp32 [N])
 *   aux_out[m,n] = T(v)                       (aux_out != NULL: pre-activation copy, 16-bit, ld = ldaux)
 *   act: MICO_ACT_GELU -> v = gelu_erf(v);  MICO_ACT_GELU_GRAD -> v *= gelu'(aux_in[m,n]);
 *        MICO_ACT_GELU_SAVE_DERIV -> aux_out holds gelu'(v) instead of v, v = gelu_erf(v);  MICO_ACT_MUL_AUX -> v *= aux_in[m,n]
 *        MICO_ACT_SILU_MUL_GRAD etc. see enum
 *   v *= row_scale[m / rows_per_scale]        (row_scale != NULL: DropPath per-sample factor, eva_vit_model.py:121-138)
 *   v += resid[m',n]                          (resid != NULL, fp32, ld = ldc; may alias C)
 *   out row index: m' = m + (m / remap_group) * remap_skip + remap_offset   (patch rows -> token rows, :616-619)
 *   v += pos[(m' % pos_rows) , n]             (pos != NULL, fp32 [pos_rows,N]: positional table, eva_vit_model.py:619)
 *   C[m',n] = v  (fp32 if c_dtype == MICO_F32 (beta=1 accumulates: C += v, uses atomics when split_k > 1) else T)
 * ------------------------------------------------------------------------------------------------------------- */
#define MICO_ACT_NONE 0
#define MICO_ACT_GELU 1      /* v = gelu(v) */
#define MICO_ACT_GELU_GRAD 2 /* v = v * gelu'(aux_in); with aux_out != NULL also aux_out = T(gelu(aux_in)) (ABI 115, see below) */
/* The pair the MLPs use: the forward GEMM keeps gelu'(pre-activation) instead of the pre-activation itself (aux_out, same bytes), so
 * the backward epilogue is one multiply - the erf polynomial is ~10 % of a 256x256 tile's time and was paid in both directions.
 * Dedicated kernel instantiations: SAVE_DERIV needs ta = tb = 0, aux_out and a 16-bit C; MUL_AUX needs ta = 0, tb = 1 and aux_in. */
#define MICO_ACT_GELU_SAVE_DERIV 3 /* aux_out = T(gelu'(v)); v = gelu(v) */
#define MICO_ACT_MUL_AUX 4         /* v = v * aux_in */
/* The pair that keeps ONE 16-bit tensor per MLP (round 6, ABI 115; the towers' saved activations: half the bytes of gelu + gelu'): the forward is
 * MICO_ACT_GELU with aux_out = the pre-activation copy h, and fc2's input-gradient launch is MICO_ACT_GELU_GRAD with aux_in = h and
 * aux_out = a second 16-bit [M, N] buffer (ldaux) that receives gelu(h) - the operand of fc2's weight gradient, re-created by the launch that
 * reads h anyway (~0.3 ms per block against the 3.7 ms fc1 GEMM a block without kept intermediates re-runs).  Both take aux_tiled (h in the
 * persistent kernel's accumulator layout; gelu(h) is always row-major). */

typedef struct mico_gemm_epilogue {
    const float* bias;      /* [N] or NULL */
    void* aux_out;          /* 16-bit [M,N] pre-activation copy (GELU_SAVE_DERIV: gelu'; GELU_GRAD: gelu(aux_in)) or NULL */
    const void* aux_in;     /* 16-bit [M,N] (GELU_GRAD) or NULL */
    int64_t ldaux;
    int act;
    const float* row_scale; /* [ceil(M / rows_per_scale)] or NULL */
    int rows_per_scale;
    const float* resid;     /* fp32 [M,N] (ld = ldc, indexed with the remapped row) or NULL */
    const float* pos;       /* fp32 [pos_rows, N] or NULL */
    int pos_rows;
    int remap_group;        /* 0 = no row remap */
    int remap_skip;
    int remap_offset;
    float alpha;
    int accumulate;         /* fp32 output only: C += v */
    /* k-segments (split-precision GEMM, fp16 parity configuration): the logical reduction of length K = nseg * kseg is the
     * concatenation of nseg segments; segment s reads A columns a_seg_off[s].. and B columns b_seg_off[s]..  With
     * A = [x_hi | x_lo], B = [W_hi | W_lo] and offsets a = {0, D, 0}, b = {0, 0, D} one launch computes
     * x_hi W_hi + x_lo W_hi + x_hi W_lo  (~22 mantissa bits from fp16 MFMAs).  nseg == 0 disables. */
    int nseg, kseg;
    int a_seg_off[3], b_seg_off[3];
    /* frame scatter (stochastic-depth frame skipping): the M rows are a compacted list of whole frames; row m is written to
     * (and resid / pos / row_scale are indexed with) row  row_map[m / rows_per_map] * rows_per_map + m % rows_per_map.
     * NULL = identity.  Not combinable with remap_group. */
    const int* row_map;
    int rows_per_map;
    /* dropout on the GEMM result (after bias / activation, before row_scale / pos / resid): element (m, n) is kept with
     * probability 1 - drop_p and scaled by 1/(1 - drop_p), decided by mico_dropout's counter hash of
     * (drop_seed, drop_site, m * N + n).  drop_p == 0 disables.  BertSelfOutput / BertOutput, bert.py:291-297, 369-375. */
    float drop_p;
    unsigned drop_seed;
    int drop_site;
    /* weight-gradient launches (ta = tb = 1, fp32 C) only: colsum_out[m] += alpha * sum_k A[k, m] - the bias gradient db = sum_rows dy
     * of the layer whose dW = dy^T x this launch computes, taken from the dy panel the kernel stages anyway (the 192x256 kernel's
     * producer waves sum the columns of tile column 0; other kernels run the stand-alone column-sum pass).  fp32 [M], accumulated. */
    float* colsum_out;
    /* split-K scratch (optional, caller-owned, fp32, 16-byte aligned): with at least split_k * M * N * 4 bytes the large weight-gradient
     * kernel writes each K-split's partial tile with plain full-line stores into its own [M, N] slab and a reduction pass adds the slabs
     * into C (C += alpha * sum) - a wave of 256 atomic epilogues costs ~100 us, the same tiles as plain stores ~20 us + one streaming
     * pass.  NULL / too small / C not 16-byte aligned: fp32 atomics into C.  When the library sizes the split itself (split_k = 0) it stays
     * within the scratch.  SINGLE-STREAM: the slabs are written by the GEMM and read back by the reduction pass launched right behind it on
     * `stream`; two launches that may run concurrently (different streams) must be given different scratch buffers. */
    void* splitk_ws;
    int64_t splitk_ws_bytes;
    /* The MLP pair's private layout for gelu'(pre-activation) (round 5) - and for the pre-activation itself (round 6: GELU + aux_out writes it,
     * GELU_GRAD + aux_in reads it; the same rules).  The tensor GELU_SAVE_DERIV writes (aux_out) is read by exactly one other
     * launch, MUL_AUX (aux_in), of the same [M, N] - so with aux_tiled != 0 both keep it the way the persistent 8-phase kernel's accumulators
     * hold it instead of row-major: tile (tm, tn) of 256 x 256 is the 128 KiB at ((tm * N / 256) + tn) * 128 KiB, wave w its 16 KiB at w * 16 KiB,
     * unit u (0..15) of a wave is one KiB with lane l's eight values at l * 16 bytes.  Written and read with whole-KiB accesses straight from /
     * into registers (a row-major aux costs the dX launch 6 %: 32-byte row segments in the accumulator layout).  ldaux is ignored; the buffer holds
     * mico_gemm_aux_tiled_elems(M, N, K) elements (M rounded up to 256 rows), which is 0 when these launches would not take that kernel
     * - then the aux tensor stays row-major.  16-bit launches only (mico_gemm_mx8 refuses it).  A launch given aux_tiled that cannot honour it fails with MICO_EINVAL. */
    int aux_tiled;
} mico_gemm_epilogue;

/* which kernel the calling thread's last mico_gemm launched: 0 = 128x128 tile, 1 = 256x256 8-wave ping-pong, 2 = 192x256
 * producer/consumer, 3 = 256x256 one wave per SIMD (experiment builds), 4 = MX-fp8, 5 = 256x256 8-wave persistent, 6 / 7 = 256x128 four-wave kernels, two
 * workgroups per CU (32-deep stages / 64-deep unit ring) (profiling aid: lets a caller attribute its per-launch timings to the kernel rocprofv3 reports) */
int mico_gemm_last_kernel(void);
/* (The product library routes by the problem alone and keeps no process-global routing state.  The experiment switch of earlier rounds,
 * mico_gemm_set_variant(v) - force the 32-deep 8-wave kernel (12), the 256x128 two-workgroups-per-CU kernels (5 / 6, 8 / 9), the 8-phase kernel
 * (10; 15 = with its generic epilogue) ... onto every large problem they support, for same-process A/B measurements - is exported by the probe
 * build only: `make -C mico_amd/csrc variants` -> tools/probes/bin/libmico_variants.so, loaded through MICO_HIP_LIB.) */
int64_t mico_gemm_aux_tiled_elems(int64_t M, int64_t N, int64_t K);   /* see mico_gemm_epilogue::aux_tiled */
int mico_gemm(int ta, int tb, int64_t M, int64_t N, int64_t K,
              const void* A, int64_t lda, const void* B, int64_t ldb,
              void* C, int64_t ldc, int c_dtype,
              const mico_gemm_epilogue* epi, int split_k, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * MX-fp8 GEMM (BASELINE.json configs[4], "fp8 MFMA"): block-scaled OCP microscaling operands on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 MFMA rate; gfx950's unscaled fp8 MFMAs run at the bf16 rate).
 *   mico_quant_mx8: x [rows, cols] 16-bit (ld) * pre_scale -> q [rows, cols] e4m3 (ldq bytes) + scales uint32 [cols / 128][rows]: the
 *     four E8M0 bytes of one row's 128-column tile per word (byte b = columns 32 b .. 32 b + 31 of the tile), scale 2^e with the
 *     smallest e such that the block's amax / 2^e <= 448.  cols % 128 == 0.
 *   mico_gemm_mx8:  C[M,N] = epilogue( sum_k A[m,k] sa[m,k/32] * B[n,k] sb[n,k/32] ), A [M,K] / B [N,K] e4m3 with leading dimensions in
 *     bytes, scales as written by mico_quant_mx8, fp32 accumulation; the epilogue struct, c_dtype and out_dtype (the 16-bit type of C and
 *     of the aux tensors) mean what they mean for mico_gemm.  K % 128 == 0; only this orientation (y = x W^T; an input gradient
 *     dx = dy W is the same call on a transposed fp8 copy of W, quantised along ITS reduction dimension).  No split-K.
 * The reference has no fp8 path (its only reduced precision is fp16 autocast, data/utils/pipeline.py:30,43): configs[4] is this
 * build's to honour; tolerance and use are documented in DESIGN.md section 4.
 * ------------------------------------------------------------------------------------------------------------- */
int mico_quant_mx8(const void* x, int64_t ld, int64_t rows, int cols, void* q, int64_t ldq, void* scales, float pre_scale,
                   int dtype, void* stream);
int mico_gemm_mx8(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* a_scales, const void* B, int64_t ldb,
                  const void* b_scales, void* C, int64_t ldc, int c_dtype, const mico_gemm_epilogue* epi, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm (row statistics in fp32).  Replaces model/evaclip/transformer.py:121-127 (eps 1e-6) and
 * torch.nn.LayerNorm in model/bert.py:92,147,288,296 / model/mico.py:49,400-403 (eps 1e-12).
 *   x: [rows, cols] fp32 (x_dtype == MICO_F32) or 16-bit;  y16 / y32 optional outputs;  mean/rstd [rows] saved.
 *   post_add (optional, fp32 [post_groups, cols]): y += post_add[(row / post_rows_per_group) % post_groups]
 *   (frame + type embeddings of model/mico.py:201,209).
 *   y16_split != 0: y16 is [rows, 2*cols] = [hi | lo] with lo = T(y - hi) (A operand of a split-precision mico_gemm).
 *   frame_map (optional, int32 [rows / rows_per_frame]): input row r is read from row
 *   frame_map[r / rows_per_frame] * rows_per_frame + r % rows_per_frame of x (compacting gather of whole frames; all outputs
 *   are compact);  x_copy (optional, fp32 [rows, cols]) receives the gathered input rows (saved for the backward).
 *   drop_p > 0: dropout on the outputs (BertEmbeddings, bert.py:147-148), element index row * cols + col, see mico_dropout.
 *   valid_cols (0 = cols): the rows are zero-padded [valid_cols | 0 ...] vectors - statistics (and, in the backward, the two row means)
 *   are taken over the valid columns only; gamma / beta must be zero-padded so that the padded outputs are 0 (EVA02-CLIP-L's 2730-wide
 *   SwiGLU hidden, eva_vit_model.py:203-224, lives in 2752-wide buffers: GEMM operands need 16-byte rows).
 *   xhat16 (optional, fp16 [rows, cols] whatever `dtype` is) receives the NORMALISED gathered rows (x - mean) * rstd rounded to fp16 - half the
 *   bytes of x_copy, and all a backward needs next to rstd: mico_layernorm_bwd takes it as x with x_normalized = 1, and a later forward with
 *   x = that buffer, x_dtype = MICO_F16, x_normalized = 1 re-creates the output (y = x gamma + beta: the activation diet's LayerNorm recompute;
 *   no statistics, no copies).  fp16 for both compute types: |xhat| <= sqrt(cols), and bf16 would keep 8 bits of it.
 *   q8 / ldq / scales (fp8 mode; the LayerNorm-fed GEMMs qkv, fc1): the 16-bit output is ALSO written as the block-scaled fp8 A operand of
 *   mico_gemm_mx8 - q8 [rows, cols] e4m3 (row stride ldq), scales as mico_quant_mx8 lays them out ([cols / 128][rows] words of four E8M0
 *   bytes) - bit-identical to mico_quant_mx8(y16), without its pass over y16.  cols % 128 == 0, <= 2048; the towers' subset of the features
 *   (no y32 / post_add / dropout / split / valid_cols).
 * Both entry points take ONE parameter struct (ABI 111; they took 24 and 28 positional arguments before - one swapped int was silent).  Zero-
 * initialise it and set what the call uses; mico_struct_layout() reports the compiled layout of both structs.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mico_ln_fwd_params {
    const void* x;              /* [rows(, gathered through frame_map), cols] */
    int x_dtype;                /* MICO_F32 | the call's dtype | MICO_F16 with x_normalized */
    int x_normalized;           /* != 0: x holds (x - mean) * rstd already (an earlier call's xhat16) */
    const float* gamma;
    const float* beta;
    void* y16;                  /* optional 16-bit output ([rows, 2 cols] = [hi | lo] with y16_split) */
    float* y32;                 /* optional fp32 output */
    float* mean;                /* optional fp32 [rows] */
    float* rstd;                /* optional fp32 [rows] */
    int64_t rows;
    int cols;
    float eps;
    const float* post_add;      /* optional fp32 [post_groups, cols] */
    int post_rows_per_group, post_groups;
    int y16_split;
    const int* frame_map;       /* optional int32 [rows / rows_per_frame] */
    int rows_per_frame;
    float* x_copy;              /* optional fp32 [rows, cols]: the gathered input rows */
    void* xhat16;               /* optional fp16 [rows, cols]: the gathered rows, normalised */
    float drop_p;
    unsigned drop_seed;
    int drop_site;
    int valid_cols;             /* 0 = cols */
    void* q8;                   /* optional (fp8 mode): e4m3 [rows, ldq] */
    int64_t ldq;
    void* scales;               /* with q8: uint32 [cols / 128][rows] */
} mico_ln_fwd_params;
int mico_layernorm_fwd(const mico_ln_fwd_params* p, int dtype, void* stream);

/* dx = LN'(dy_scale * dy) [+ dx_add]; dy fp32 or 16-bit (dy_dtype); outputs dx32 (may alias dx_add) and/or dx16
 * (dx16 = T(dx * scale16)).
 * x: the forward's input rows (fp32 or 16-bit, with mean / rstd), or - x_normalized != 0, x_dtype MICO_F16 - the forward's xhat16 copy (mean is
 * then not read and may be NULL).
 * dgamma/dbeta: partial sums are written to ws [2, nblk, cols] (nblk = mico_layernorm_bwd_nblk(rows)), then reduced
 * and ACCUMULATED (+=) into dgamma/dbeta (fp32 [cols]) scaled by grad_scale.
 * frame_map (optional): dx_add and dx32 are indexed with the scattered row
 * frame_map[r / rows_per_frame] * rows_per_frame + r % rows_per_frame (dy, x, mean, rstd, dx16 stay compact).
 * dx16_dst (optional, int32 per compact frame; needs rows_per_frame): dx16 is the 16-bit operand of the NEXT consumer, which keeps another
 * frame set - compact frame j goes to frame slot dx16_dst[j] of dx16 (< 0: not kept there, not written) and is multiplied by
 * scale16 * dx16_frame_scale[scattered frame] (dx16_frame_scale optional, fp32 per frame of the full stream).  Replaces the
 * mico_gather_rows_cast pass over the frames both sets share (the stochastic-depth backward: the residual-stream gradient a branch's
 * LayerNorm backward just produced is what the next branch's GEMMs read).
 * dx16_drop_p > 0: dx16 is additionally multiplied by mico_dropout's keep / (1 - p) mask of (dx16_drop_seed, dx16_drop_site, element index
 * in the [rows, cols] dx16) - the gradient side of a hidden-state dropout that sat between this LayerNorm's input and the dense layer
 * (bert.py:295,373), without the separate mico_dropout pass. */
typedef struct mico_ln_bwd_params {
    const void* dy;
    int dy_dtype;
    float dy_scale;
    const void* x;
    int x_dtype;
    int x_normalized;
    const float* gamma;
    const float* mean;
    const float* rstd;
    const float* dx_add;
    float* dx32;
    void* dx16;
    float scale16;
    float* dgamma;
    float* dbeta;
    float grad_scale;
    float* ws;
    int64_t rows;
    int cols;
    const int* frame_map;
    int rows_per_frame;
    int valid_cols;
    const int* dx16_dst;
    const float* dx16_frame_scale;
    float dx16_drop_p;
    unsigned dx16_drop_seed;
    int dx16_drop_site;
} mico_ln_bwd_params;
int mico_layernorm_bwd_nblk(int64_t rows);
int mico_layernorm_bwd(const mico_ln_bwd_params* p, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention (flash style: scores never materialised), forward and backward.
 * Replaces eva_vit_model.py:340-361 and bert.py:246-277 (self and cross attention).
 *   q: rows (b, i) at q + b*q_bs + i*q_rs + h*hd ;  k,v likewise with Sk rows ;  o: (b,i) at o + b*o_bs + i*o_rs + h*hd
 *   (strides in elements, multiples of 8) - lets q/k/v alias one fused [M, 3*H*hd] projection buffer.
 *   hd in {64, 88, 96, 128}(padded to a multiple of 32 in LDS); scale multiplies q.k (EVA scales q first, BERT divides
 *   the scores - equal up to rounding).
 *   mask_mode 0: none; 1: additive key mask fp32 [B, Sk]; 2: additive fp32 [B, Sq, Sk]   (bert.py:764-780, -10000 based)
 *   lse: fp32 [B, H, Sq] log-sum-exp saved for backward.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mico_attn_params {
    int B, H, Sq, Sk, hd;
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    float scale;
    const float* mask;
    int mask_mode;
    /* dropout on the attention probabilities (bert.py:267): P[b,h,i,j] kept with probability 1 - drop_p and scaled by
     * 1/(1 - drop_p); the decision is mico_dropout's counter hash of (drop_seed, drop_site, ((b*H + h)*Sq + i)*Sk + j), so the
     * backward kernels regenerate it instead of storing a mask.  drop_p == 0 disables. */
    float drop_p;
    unsigned drop_seed;
    int drop_site;
    /* > 0: batch entry b reads the K/V of batch entry b % kv_batch_mod (k_bs / v_bs strides) - shared cross-attention memory: the
     * ITM triplet [own | hard-negative | own] of vast.py:438-447 is one [own | hard-negative] K/V buffer with kv_batch_mod = 2 b.
     * dK / dV are still written per batch entry b (the caller adds the aliased parts).  0: every batch entry has its own K/V. */
    int kv_batch_mod;
    /* A launch over a SLICE of a larger batch: batch0 = index of the slice's first entry in the whole batch - the dropout counters use
     * b + batch0, so that a backward issued in several launches regenerates the masks of a forward issued in one.  0 otherwise. */
    int batch0;
    /* != 0 (mico_attn_bwd, one-pass kernel for Sq <= 80 at hd 64 only): dK / dV are ADDED to what the buffers hold (16-bit
     * read-modify-write) instead of overwriting it - the second launch over the ITM triplet's third third, whose K/V set is the first
     * third's, leaves the sum in place (no per-entry dK/dV buffer, no add pass). */
    int dkv_accumulate;
} mico_attn_params;

int mico_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                  const mico_attn_params* p, int dtype, void* stream);
/* dq/dk/dv use the q/k/v strides; d_o uses the o strides; delta: fp32 workspace [B,H,Sq] (scratch: the one-pass kernel for
 * Sq <= 80 at hd 64 - BERT's text rows - keeps rowsum(O * dO) in LDS and leaves it untouched). */
int mico_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  void* dq, void* dk, void* dv, float* delta,
                  const mico_attn_params* p, int dtype, void* stream);

/* RoPE on tokens 1.. of a [B, N, H, hd] (row stride rs, batch stride bs) buffer, in place; inverse = transposed
 * rotation for the backward.  cos/sin fp32 [N-1, hd].  rope.py:121-137, eva_vit_model.py:314-322. */
int mico_rope(void* x, int64_t bs, int64_t rs, int B, int N, int H, int hd, const float* cos_t, const float* sin_t,
              int inverse, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Patch embedding front-end: im2row of [B,C,H,W] fp32 pixels into 16-bit rows [B*(H/P)*(W/P), kpad]
 * (k = c*P*P + i*P + j, zero padded to kpad); the projection itself is mico_gemm with pos/remap epilogue.
 * eva_vit_model.py:440-448.  C may be 1 (audio spectrogram with channel-summed weights, mico.py:140).
 * ------------------------------------------------------------------------------------------------------------- */
int mico_im2row(const float* pixels, void* rows16, int B, int C, int H, int W, int P, int kpad, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Elementwise / data-movement helpers (all HBM-bound, 16-byte vectorised).
 * ------------------------------------------------------------------------------------------------------------- */
/* dst16[r, 0:cols_pad] = T(scale * src32[r, 0:cols]) zero-padded; */
int mico_cast_f32_to_16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, int cols_pad,
                        float scale, int dtype, void* stream);
int mico_cast_16_to_f32(const void* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int cols,
                        float scale, int accumulate, int dtype, void* stream);
/* dst16[m', :] = T(scale * row_scale[m'/rows_per_scale] * src32[m, :]) with the same row remap as the GEMM epilogue
 * (gathers token rows out of the residual-gradient stream, skipping CLS rows).  rows = number of output rows.
 * frame_map (optional, int32): source row = frame_map[m' / rows_per_frame] * rows_per_frame + m' % rows_per_frame
 * (compacting gather of whole frames); row_scale is indexed with the SOURCE row in both remap modes.
 * dst_map (optional, with frame_map): frame j of the list is written to frame slot dst_map[j] of dst instead of slot j (rows then counts the
 * listed frames' rows) - the frames a fused producer (mico_layernorm_bwd's dx16_dst) did not cover. */
int mico_gather_rows_cast(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                          int remap_group, int remap_skip, int remap_offset,
                          const float* row_scale, int rows_per_scale, float scale,
                          const int* frame_map, int rows_per_frame, const int* dst_map, int dtype, void* stream);
/* In-place dropout x[r, c] *= keep(r * cols + c) / (1 - p) on an fp32 or 16-bit [rows, cols] tensor (leading dim ld): the
 * backward of every hidden-state dropout (the same mask multiplies the gradient).  keep() is a stateless counter hash of
 * (seed, site, index) - no mask tensor exists anywhere:
 *   h = seed ^ site * 0x9E3779B9;  h ^= lo32(idx) * 0x85EBCA6B;  h = rotl(h, 13) * 5 + 0xE6546B64;
 *   h ^= hi32(idx) * 0xC2B2AE35;   h = rotl(h, 13) * 5 + 0xE6546B64;
 *   h ^= h >> 16;  h *= 0x85EBCA6B;  h ^= h >> 13;  h *= 0xC2B2AE35;  h ^= h >> 16;     keep = (h >> 8) >= (uint32)(p * 2^24) */
int mico_dropout(void* x, int x_dtype, int64_t rows, int cols, int64_t ld, float p, unsigned seed, int site, void* stream);
/* out[c] (+)= scale * sum_r x[r, c]   (bias gradients, positional-table gradients).  x fp32 or 16-bit. */
int mico_colsum(const void* x, int x_dtype, int64_t ld, int64_t rows, int cols, float* out, float scale, int accumulate,
                void* stream);
/* x[b*group_rows + 0, :] = cls[:] + pos[0, :]  for every frame b (fp32).  eva_vit_model.py:616-619. */
int mico_cls_rows(float* x, int64_t ld, int B, int group_rows, const float* cls, const float* pos0, int cols, void* stream);
/* y[r,:] = a[r,:] + b[r,:] (fp32), optional 16-bit copy */
int mico_add_f32(const float* a, const float* b, float* y, void* y16, int64_t n, float scale16, int dtype, void* stream);

/* Weight gradient of a Linear whose input is a LayerNorm output y = xhat * gamma + beta (eva_vit_model.py:409-416: norm1 -> attn.qkv,
 * norm2 -> mlp.fc1; torch autograd forms dy^T y from the saved y).  With only the fp16 normalised rows xhat kept (mico_ln_fwd_params::xhat16)
 * the backward runs mico_gemm(ta = tb = 1) against xhat into the zeroed scratch pair dwt[M, N] / dbt[M] (colsum_out) and this call adds
 *     dw[m, n] += dwt[m, n] * gamma[n] + dbt[m] * beta[n],    db[m] += dbt[m]  (db may be NULL)
 * - the same sums, without re-creating y over all rows.  fp32 throughout (16-byte accesses when N, ld_dw are multiples of 4 and the bases aligned). */
int mico_dw_colfold(const float* dwt, const float* dbt, const float* gamma, const float* beta, float* dw, int64_t ld_dw, float* db, int M, int N,
                    void* stream);

/* exact-erf GELU (nn.GELU; mico.py:22-28) forward / backward, fp32 and 16-bit flat arrays. */
int mico_gelu_f32(const float* x, float* y, int64_t n, void* stream);
int mico_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int mico_gelu_16(const void* x, void* y, int64_t n, int dtype, void* stream);
int mico_gelu_bwd_16(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);
/* CLS pooling (mico.py:157-182): pooled[b,:] = mean_f tokens[(b*n+f)*frame_stride + 0..D); backward adds into dtokens. */
int mico_cls_pool_fwd(const float* tokens, float* pooled, int b, int n, int64_t frame_stride, int D, void* stream);
int mico_cls_pool_bwd(const float* dpooled, float* dtokens, int b, int n, int64_t frame_stride, int D, void* stream);
/* pool_video (mico.py:190-191, 217-218, 233-234: torch.cat([x[:, :, 0:1], x[:, :, 1:].mean(2, keepdim=True)], dim=2)): tokens [frames, N, D] fp32
   -> pooled [frames, 2, D] = the CLS row and the mean of the N - 1 patch rows; the backward writes every element of dtokens (no accumulate).
   N >= 2, D % 4 == 0; frames == 0 is a no-op. */
int mico_pool_video_fwd(const float* tokens, float* pooled, int64_t frames, int N, int D, void* stream);
int mico_pool_video_bwd(const float* dpooled, float* dtokens, int64_t frames, int N, int D, void* stream);

/* SwiGLU gate (eva_vit_model.py:217-220): h = silu(x1) * x2, 16-bit in/out, and its backward. */
int mico_swiglu_fwd(const void* x1, const void* x2, void* h, int64_t n, int dtype, void* stream);
int mico_swiglu_fwd_f32(const float* x1, const float* x2, float* h, int64_t n, void* stream);   /* fp32 (parity configuration) */
int mico_swiglu_bwd(const void* x1, const void* x2, const void* dh, void* dx1, void* dx2, int64_t n, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * BERT embeddings: out = LN(word[ids] + type[0] + pos[s]) (bert.py:139-148) and the word-gradient scatter-add.
 * ------------------------------------------------------------------------------------------------------------- */
int mico_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                        float* sum32, int64_t rows, int S, int cols, int vocab, void* stream);
int mico_embed_scatter_add(const int64_t* ids, const float* dsum, float* dword, float* dpos, float* dtype0,
                           int64_t rows, int S, int cols, int vocab, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Losses.
 *  mico_ce_fwd_bwd: row-wise cross-entropy with label smoothing over fp32 or 16-bit logits [rows, cols] (ld):
 *     loss_sum += sum_r CE(logits[r], target[r]),  n_valid += #(target != ignore_index)
 *     dlogits (16-bit or fp32, may be NULL) = dscale * (softmax - smoothed_onehot)   (0 for ignored rows)
 *  Replaces F.cross_entropy in vast.py:411-414,455 and bert.py:1088-1090 (logits_scale = 1/temp for ITC).
 * ------------------------------------------------------------------------------------------------------------- */
/* ITM hard-negative sampling (data/model/vast.py:423-440: weights = softmax(sim, 1) + 1e-4, own-rank diagonal zeroed, one multinomial
 * draw per row - there a Python loop with a .item() host sync per row).  sim: fp32 [rows, cols] (ld) similarity logits AFTER the
 * temperature; row r never draws column diag_offset + r (= rank * b + r); u: fp32 [rows] uniform numbers in [0, 1) supplied by the
 * caller (injected for parity, torch.rand otherwise); out[r] = #{j : cdf_r[j] <= u[r] * total_r} (inverse-CDF draw: the first column whose CDF
 * exceeds the target, so a zero-weight column is never drawn), int64. */
int mico_itm_sample(const float* sim, int64_t ld, int rows, int cols, int diag_offset, const float* u, int64_t* out, void* stream);
/* Caption-loss token masking (TokenMasker.perform_mask, data/model/general_module.py:64-97 - there two Python loops over b x S on the host behind
 * a .cpu() copy, i.e. a stream sync per step).  tokens: int64 [rows, S].  A token at position j >= 1 with id != 0 is selected when
 * u_mask[r][row][j] < mask_prob; round r = 0 stands unless it selects nothing in the row, then round 1 is drawn, ... (the reference's
 * "while all(indicator == 0)" retry, which guarantees >= 1 masked token per row; after `rounds` empty rounds ONE position is forced instead:
 * the floor(u_tok[row][0] * n)-th of the row's n maskable positions - u_tok of position 0 is otherwise unused; a row with n = 0 stays unmasked).  A selected token becomes mask_token when u_kind < 0.8, the id
 * range_start + floor(u_tok * (range_end - range_start)) when 0.8 <= u_kind < 0.9, and is kept otherwise; labels = the source id at selected
 * positions, -100 elsewhere.  u_mask: fp32 [rounds, rows, S]; u_kind, u_tok: fp32 [rows, S] - uniform numbers in [0, 1) supplied by the caller
 * (injected for parity, torch.rand otherwise). */
int mico_token_mask(const int64_t* tokens, int rows, int S, float mask_prob, const float* u_mask, int rounds, const float* u_kind,
                    const float* u_tok, int mask_token, int range_start, int range_end, int64_t* out_tokens, int64_t* labels, void* stream);

int mico_ce_fwd_bwd(const void* logits, int logits_dtype, int64_t ld, int64_t rows, int cols,
                    const int64_t* target, int ignore_index, float label_smoothing, float logits_scale,
                    float* row_loss, float* row_lse,
                    void* dlogits, int dlogits_dtype, int64_t ld_d, const float* dscale_ptr, float dscale,
                    int dtype, void* stream);
/* Small exact-fp32 GEMM for the tiny heads and similarity matrices (contra heads, itm head, ITC logits; vast.py:405-408,
 * mico.py:36-52): C = alpha * opA(A) opB(B) + beta * C, same ta/tb convention as mico_gemm, any sizes, fp32 everywhere. */
int mico_sgemm_small(int ta, int tb, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                     float* C, int64_t ldc, float alpha, float beta, const float* bias, void* stream);
/* L2 normalise rows (F.normalize, eps 1e-12) forward / backward, fp32. */
int mico_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int cols, void* stream);
int mico_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int cols, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Device-side input preprocessing.
 *  mico_image_preprocess: n decoded RGB frames, uint8 [n, H, W, 3]  ->  fp32 [n, 3, out_h, out_w]:
 *     ToTensor (/255), Resize((out_h, out_w)) bilinear without antialias (source index (d + 0.5) * in/out - 0.5 clamped at 0,
 *     as torch's upsample_bilinear2d with align_corners = False), Normalize(mean, std)   - model/imageprocessor.py:26-38,51-55,
 *     model/videoprocessor.py:35-50.
 *  mico_fbank_windows: log-mel filterbank [T, mel] fp32 -> n windows [n, target_len, mel]:
 *     out[i, t, :] = (fbank[win[i] * target_len + t, :] - mean) * inv_scale, zero past T  (normalise, zero-pad, slice:
 *     model/audioprocessor.py:45-70; inv_scale = 1 / (2 std)).
 * ------------------------------------------------------------------------------------------------------------- */
int mico_image_preprocess(const unsigned char* src, int n, int H, int W, float* dst, int out_h, int out_w,
                          float mean0, float mean1, float mean2, float istd0, float istd1, float istd2, void* stream);
int mico_fbank_windows(const float* fbank, int T, int mel, const int* win, int n, int target_len, float mean, float inv_scale,
                       float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Optimizer step: the decoupled-weight-decay Adam of data/utils/build_optimizer.py:105-197,
 * one launch for a whole parameter group (multi-tensor).  Per element, in this order (fp32):
 *     m = beta1 m + (1 - beta1) g;   v = beta2 v + (1 - beta2) g g;   p -= step_size * m / (sqrt(v) + eps);
 *     if (weight_decay > 0) p -= lr * weight_decay * p;
 * step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) when correct_bias, else lr (computed by the caller).
 * tensors: device array of n_tensors descriptors; chunk_tensor / chunk_start (device int32 / int64, nchunks each) split the
 * tensors into pieces of at most chunk_elems elements, one workgroup each.  A descriptor may name a 16-bit mirror of the
 * parameter (the GEMM-operand copy the engine keeps): w16[(i / cols) * ld16 + i % cols] = T(p[i]) is refreshed in the same
 * pass (and, for the split-precision layout, the low half T(p - hi) at + lo_off), so no re-cast pass follows a step.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mico_adamw_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t numel;
    void* w16;        /* optional 16-bit mirror (NULL: none) */
    int64_t ld16;     /* row stride of the mirror in elements */
    int64_t lo_off;   /* > 0: also write the low half at w16 + lo_off (split-precision layout) */
    int cols;         /* row length of the parameter viewed as [rows, cols] */
    int w16_dtype;    /* MICO_F16 / MICO_BF16 */
} mico_adamw_tensor;

/* grad_mult: every gradient is multiplied by it before use (1 / loss scale: the unscale_ of torch's GradScaler folded into the update,
 * data/utils/pipeline.py:88,106; 1.0 otherwise). */
int mico_adamw_step(const mico_adamw_tensor* tensors, int n_tensors, const int* chunk_tensor, const int64_t* chunk_start,
                    int nchunks, int chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float step_size, float grad_mult, void* stream);
/* Overflow check of a scaled backward (GradScaler.step, data/utils/pipeline.py:106): *flag = 1.0f if any gradient of the table (only
 * .g / .numel of each entry are read) is inf or NaN; *flag is left untouched otherwise (the caller zeroes it). */
int mico_grads_finite(const mico_adamw_tensor* tensors, int n_tensors, const int* chunk_tensor, const int64_t* chunk_start,
                      int nchunks, int chunk_elems, float* flag, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Swin tower pieces (model/swin.py).  Everything else of the tower runs on the entry points above (patch
 * embedding = mico_im2row + mico_gemm, LayerNorm, qkv / proj / MLP GEMMs with their epilogues).
 *  mico_win_attn_fwd / _bwd: (shifted-)window multi-head self-attention, 7x7 windows, head dim 32 (Swin-T/S/B/L), replacing the chain
 *     roll -> window_partition -> WindowAttention core -> window_reverse -> roll of swin.py:258-289 + :134-152 without any permuted copy:
 *     qkv [batch*res*res, 3*heads*32] is the qkv Linear's output in token order (q | k | v, head-major), out / dout [batch*res*res, heads*32]
 *     in the same token order; scores = scale * q k^T + bias_table[rel_index, head] (+ the -100 shifted-window mask of :232-253, computed
 *     from the geometry when shift > 0); lse [batch*res*res, heads] fp32 is the softmax log-sum-exp the backward reuses.
 *     Backward writes dqkv completely (same layout and gradient scale as dout) and ADDS dbias_scale * dS, binned by relative position,
 *     to dbias_table [169, heads] (fp32, caller zero-fills).  res must be a multiple of 7; shift = 0 when res == 7 (swin.py:206-209).
 *  mico_patch_merge: the 2x2 neighbourhood gather of PatchMerging (swin.py:340-346) on the fp32 stream:
 *     out[b, (y, x), q*C + c] = in[b, (2y + (q & 1), 2x + (q >> 1)), c]; backward = 1 runs the map in reverse (in = merged gradient,
 *     out = token-grid gradient; the map is a bijection, so no accumulation).  channels % 4 == 0.
 * ------------------------------------------------------------------------------------------------------------- */
int mico_win_attn_fwd(const void* qkv, void* out, float* lse, const float* bias_table, int batch, int res, int heads, int shift,
                      float scale, int dtype, void* stream);
int mico_win_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* bias_table, void* dqkv, float* dbias_table,
                      int batch, int res, int heads, int shift, float scale, float dbias_scale, int dtype, void* stream);
int mico_patch_merge(const float* in, float* out, int batch, int res, int channels, int backward, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Collectives of the data-parallel alignment step over RCCL / xGMI (ABI 112; SURVEY section 8b: mico_comm_*).  For a binding that does not go
 * through torch.distributed; mico_amd/distributed.py keeps torch.distributed as its default transport and takes these with MICO_COMM=1
 * (mico_amd/comm.py).  One communicator per process and GPU.  RCCL is resolved at run time (the librccl.so.1 already resident in the
 * process, else the ROCm one): libmico_hip.so has no link-time dependency on it and a single-GPU user never loads it.
 * Every call is asynchronous on `stream`; buffers are device memory owned by the caller; counts are host integers.
 *   mico_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId); the caller distributes them (a broadcast of its own, a file, MPI ...).
 *   mico_comm_init: collective over all ranks, on the calling thread's current HIP device.  mico_comm_destroy(NULL) is a no-op.
 *   mico_comm_allgather: recv [nranks][bytes_per_rank] <- every rank's send [bytes_per_rank].
 *   mico_comm_allgather_packed: replaces the 3 + #subtasks all-gathers of data/utils/distributed.py:50-66 as called at vast.py:395-404 with ONE:
 *     nparts (<= 8) per-rank tensors of `rows` rows and row_bytes[i] bytes per row are packed row by row into pack_scratch
 *     [rows, sum(row_bytes)] (one small kernel) and gathered into recv [nranks * rows, sum(row_bytes)]; part i of row r of rank q is at
 *     recv + ((q * rows + r) * sum + off_i).  The parts are autograd constants exactly like concat_all_gather's outputs.
 *   mico_comm_alltoallv: the row exchange of the index-then-fetch that replaces all_gather_with_grad(condition_feats)[neg_idx]
 *     (data/utils/distributed.py:12-47, vast.py:421-433): send holds send_bytes[p] bytes for every peer p back to back, recv receives
 *     recv_bytes[p] from every peer p back to back (counts known on the host after the index all-gather); the gradient route is the same
 *     call with the two count arrays swapped.
 *   mico_comm_allreduce_f32 (in place; average != 0: divided by nranks) / mico_comm_reduce_scatter_f32: gradient averaging - the towers'
 *     arena slices in place from inside the backward, flat buckets as reduce-scatter (+ mico_comm_allgather).
 * ------------------------------------------------------------------------------------------------------------- */
#define MICO_COMM_ID_BYTES 128
int mico_comm_unique_id(void* id_out);
int mico_comm_init(void** comm_out, int rank, int nranks, const void* id);
int mico_comm_destroy(void* comm);
int mico_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int mico_comm_allgather_packed(void* comm, const void* const* parts, const int64_t* row_bytes, int nparts, int64_t rows,
                               void* pack_scratch, void* recv, void* stream);
int mico_comm_alltoallv(void* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes, void* stream);
int mico_comm_allreduce_f32(void* comm, float* buf, int64_t count, int average, void* stream);
int mico_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, int64_t count_per_rank, int average, void* stream);

#ifdef __cplusplus
}
#endif
#endif
