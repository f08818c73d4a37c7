// Row-wise cross-entropy with label smoothing, forward and gradient in one kernel (see include/mico_hip.h).
// One 256-thread workgroup per row; online max/sum-exp in a single pass over the row (fp32 math), block reduction
// through LDS; the optional gradient pass re-reads the row (L2-resident for the small ITC/ITM rows; one extra HBM
// read for the 30522-wide LM-head rows) and writes dlogits in the GEMM's 16-bit dtype, so fp32 logits of the LM head
// never exist in HBM.
#include "common.h"

namespace {

template <typename LT> __device__ __forceinline__ float ld_logit(const LT* p, int64_t i) { return (float)p[i]; }

template <typename LT, typename DT>
__global__ __launch_bounds__(256) void ce_kernel(const LT* __restrict__ logits, int64_t ld, int cols,
                                                 const int64_t* __restrict__ target, int ignore_index, float ls,
                                                 float lscale, float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                 DT* __restrict__ dlogits, int64_t ld_d, const float* __restrict__ dscale_ptr,
                                                 float dscale) {
    __shared__ float red[3][4];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LT* x = logits + row * ld;
    const int64_t t = target[row];
    const bool ignored = (t == ignore_index) || t < 0 || t >= cols;
    float m = -1.0e30f, s = 0.f, sx = 0.f;
    for (int c = tid; c < cols; c += 256) {
        const float v = ld_logit(x, c) * lscale;
        sx += v;
        if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
        else s += __expf(v - m);
    }
    // wave reduce (m, s), then block
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
        m = mn;
        sx += __shfl_xor(sx, o, 64);
    }
    if (lane == 0) { red[0][wave] = m; red[1][wave] = s; red[2][wave] = sx; }
    __syncthreads();
    float M = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float S = 0.f, SX = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += red[1][w] * __expf(red[0][w] - M); SX += red[2][w]; }
    const float lse = M + __logf(S);
    if (tid == 0) {
        float loss = 0.f;
        if (!ignored) {
            const float xt = ld_logit(x, t) * lscale;
            loss = (1.f - ls) * (lse - xt) + ls * (lse - SX / (float)cols);
        }
        if (row_loss) row_loss[row] = loss;
        if (row_lse) row_lse[row] = lse;
    }
    if (!dlogits) return;
    float ds = dscale * lscale;
    if (dscale_ptr) ds *= dscale_ptr[0];
    if (ignored) ds = 0.f;
    const float smooth = ls / (float)cols;
    DT* d = dlogits + row * ld_d;
    for (int c = tid; c < cols; c += 256) {
        const float v = ld_logit(x, c) * lscale;
        float gsm = __expf(v - lse) - smooth;
        if (c == t) gsm -= (1.f - ls);
        d[c] = (DT)(gsm * ds);
    }
}

// ITM hard-negative draw (vast.py:423-440): one wave per row, every lane a contiguous chunk of the row (the row is read three times
// from L1: max, sum of exponentials, weights).  idx = #{j : cdf_j < u * total} is a pure counting problem once every lane knows the
// exclusive prefix of its chunk, so no lane has to be singled out.
__global__ __launch_bounds__(64) void itm_sample_kernel(const float* __restrict__ sim, int64_t ld, int cols, int diag_offset,
                                                        const float* __restrict__ u, int64_t* __restrict__ out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float* x = sim + (int64_t)row * ld;
    const int chunk = (cols + 63) / 64;
    const int j0 = min(cols, lane * chunk), j1 = min(cols, j0 + chunk);
    float m = -3.0e38f;
    for (int j = j0; j < j1; ++j) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float se = 0.f;
    for (int j = j0; j < j1; ++j) se += __expf(x[j] - m);
    se = wave_sum(se);
    const float inv = 1.f / se;
    const int dcol = diag_offset + row;
    float part = 0.f;
    for (int j = j0; j < j1; ++j) part += (j == dcol) ? 0.f : fmaf(__expf(x[j] - m), inv, 1e-4f);
    float incl = part;   // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    const float total = __shfl(incl, 63, 64);
    const float tgt = u[row] * total;
    float run = incl - part;
    int cnt = 0;
    for (int j = j0; j < j1; ++j) {
        run += (j == dcol) ? 0.f : fmaf(__expf(x[j] - m), inv, 1e-4f);
        cnt += run <= tgt;     // first column whose CDF exceeds the target: a zero-weight column (the diagonal) is never chosen
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0) {
        if (cnt >= cols) cnt = (dcol == cols - 1) ? cols - 2 : cols - 1;   // u * total rounded up to the total: the last column with weight
        out[row] = max(cnt, 0);
    }
}

}  // namespace

// TokenMasker.perform_mask (data/model/general_module.py:64-97) with the random numbers supplied by the caller: one wave per row.
__global__ __launch_bounds__(64) void token_mask_kernel(const int64_t* __restrict__ tokens, int S, float mask_prob, const float* __restrict__ u_mask,
                                                        int rounds, int64_t round_stride, const float* __restrict__ u_kind,
                                                        const float* __restrict__ u_tok, int mask_token, int range_start, int range_end,
                                                        int64_t* __restrict__ out_tokens, int64_t* __restrict__ labels) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int64_t* t = tokens + (int64_t)row * S;
    // indicator: position 0 and pads (id 0) are never masked; a row is re-drawn (next round of uniforms) until it has >= 1 masked token
    unsigned long long any = 0;
    int r = 0;
    for (; r < rounds && !any; ++r) {
        const float* u = u_mask + r * round_stride + (int64_t)row * S;
        for (int j0 = 0; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            const bool hit = j >= 1 && j < S && t[j] != 0 && u[j] < mask_prob;
            any |= __ballot(hit);
        }
    }
    const int used = r - 1;   // the round whose draw stands
    // `rounds` exhausted with nothing selected: the reference would keep drawing (general_module.py:71 guarantees >= 1 masked token per
    // row), so ONE maskable position is forced - the floor(u_tok[row][0] * n)-th of the row's n maskable positions (u_tok of position 0 is
    // otherwise unused: position 0 is never masked).  A row without any maskable token keeps nothing masked.
    int forced = -1;
    if (!any) {
        int n = 0;
        for (int j0 = 0; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            n += __popcll(__ballot(j >= 1 && j < S && t[j] != 0));
        }
        if (n > 0) {
            int k = (int)(u_tok[(int64_t)row * S] * (float)n);
            k = k < n ? k : n - 1;
            for (int j0 = 0; j0 < S && forced < 0; j0 += 64) {
                const int j = j0 + lane;
                unsigned long long m = __ballot(j >= 1 && j < S && t[j] != 0);
                const int c = __popcll(m);
                if (k < c) {
                    for (int q = 0; q < k; ++q) m &= m - 1;     // drop the k lowest set bits
                    forced = j0 + __ffsll((long long)m) - 1;
                } else k -= c;
            }
        }
    }
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        if (j >= S) break;
        const int64_t src = t[j];
        const bool hit = any ? (j >= 1 && src != 0 && u_mask[used * round_stride + (int64_t)row * S + j] < mask_prob) : j == forced;
        int64_t tok = src, lab = -100;
        if (hit) {
            lab = src;
            const float pk = u_kind[(int64_t)row * S + j];
            if (pk < 0.8f) tok = mask_token;
            else if (pk < 0.9f) {
                int c = range_start + (int)(u_tok[(int64_t)row * S + j] * (float)(range_end - range_start));
                tok = c < range_end ? c : range_end - 1;
            }
        }
        out_tokens[(int64_t)row * S + j] = tok;
        labels[(int64_t)row * S + j] = lab;
    }
}

extern "C" int mico_token_mask(const int64_t* tokens, int rows, int S, float mask_prob, const float* u_mask, int rounds, const float* u_kind,
                               const float* u_tok, int mask_token, int range_start, int range_end, int64_t* out_tokens, int64_t* labels,
                               void* stream) {
    MICO_CHECK(tokens && u_mask && u_kind && u_tok && out_tokens && labels && rows > 0 && S > 0 && rounds > 0, "mico_token_mask: bad args");
    MICO_CHECK(range_end > range_start && mask_prob >= 0.f && mask_prob <= 1.f, "mico_token_mask: bad range / probability");
    MICO_LAUNCH(token_mask_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, tokens, S, mask_prob, u_mask, rounds, (int64_t)rows * S, u_kind, u_tok,
                mask_token, range_start, range_end, out_tokens, labels);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_itm_sample(const float* sim, int64_t ld, int rows, int cols, int diag_offset, const float* u, int64_t* out,
                               void* stream) {
    MICO_CHECK(sim && u && out && cols > 0 && ld >= cols, "mico_itm_sample: bad args");
    if (rows <= 0) return MICO_OK;
    MICO_LAUNCH(itm_sample_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, sim, ld, cols, diag_offset, u, out);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_ce_fwd_bwd(const void* logits, int logits_dtype, int64_t ld, int64_t rows, int cols,
                               const int64_t* target, int ignore_index, float label_smoothing, float logits_scale,
                               float* row_loss, float* row_lse, void* dlogits, int dlogits_dtype, int64_t ld_d,
                               const float* dscale_ptr, float dscale, int dtype, void* stream) {
    MICO_CHECK(logits && target && cols > 0, "mico_ce_fwd_bwd: bad args");
    MICO_CHECK(logits_dtype == MICO_F32 || logits_dtype == MICO_F16 || logits_dtype == MICO_BF16, "mico_ce_fwd_bwd: logits dtype");
    MICO_CHECK(!dlogits || dlogits_dtype == logits_dtype, "mico_ce_fwd_bwd: dlogits dtype must equal the logits dtype");
    (void)dtype;
    if (rows <= 0) return MICO_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)rows), block(256);
#define CE(LT) MICO_LAUNCH((ce_kernel<LT, LT>), grid, block, 0, st, (const LT*)logits, ld, cols, target, ignore_index, label_smoothing, logits_scale, row_loss, row_lse, (LT*)dlogits, ld_d, dscale_ptr, dscale)
    if (logits_dtype == MICO_F32) CE(float);
    else if (logits_dtype == MICO_F16) CE(f16);
    else CE(bf16);
#undef CE
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
