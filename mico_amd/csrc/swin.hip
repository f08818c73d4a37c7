// Swin window attention and patch merging for gfx950 (reference: model/swin.py:45-74 window_partition / window_reverse, :77-156
// WindowAttention, :232-253 the shifted-window mask, :258-289 cyclic shift, :315-352 PatchMerging).
//
// The reference rolls the token grid, cuts it into 7x7 windows, runs attention per window and undoes both permutations.  Here no
// permuted copy exists: the kernels address the tokens of a window directly in the [B, H*W, 3C] output of the qkv GEMM -
//     window (wy, wx), in-window position t = (iy, ix)  ->  rolled-grid cell (Y, X) = (7 wy + iy, 7 wx + ix)
//                                                        ->  token ((Y + s) mod H) * W + (X + s) mod W            (roll by -s)
// - and write the result to the same token, which is what window_reverse + the roll back produce.  The shift mask is arithmetic as well:
// the reference paints the rolled grid with 3 x 3 region ids (rows / columns [0, H-7), [H-7, H-s), [H-s, H)) and adds -100 where two
// cells of a window differ; region(Y) = (Y >= H - 7) + (Y >= H - s).  The relative-position bias is read from the parameter table itself
// ([169, heads], index (iy_q - iy_k + 6) * 13 + (ix_q - ix_k + 6)); its gradient is binned in LDS and flushed once per workgroup.
//
// Work shape: 49 x 49 scores at head dim 32 - two orders of magnitude below an MFMA tile's worth of work per (window, head), and the
// tower is not on the timed path (unreachable from model/mico.py): plain fp32 VALU, one wave per (window, head),
// lane = query row (forward, dQ) or key row (dK / dV), K / V / Q / dO of the window broadcast from LDS.  Exact softmax in fp32.
#include "common.h"

namespace {
constexpr int WS = 7, WT = 49, HD = 32, NB = 169;

struct WinGeom {
    int res, nw, shift;   // grid side, windows per side, cyclic shift
    __device__ __forceinline__ int token(int w, int t) const {
        const int wy = w / nw, wx = w - wy * nw, iy = t / WS, ix = t - iy * WS;
        int y = wy * WS + iy + shift, x = wx * WS + ix + shift;
        if (y >= res) y -= res;
        if (x >= res) x -= res;
        return y * res + x;
    }
    __device__ __forceinline__ int region(int w, int t) const {   // 0 when shift == 0
        if (shift == 0) return 0;
        const int wy = w / nw, wx = w - wy * nw, iy = t / WS, ix = t - iy * WS;
        const int Y = wy * WS + iy, X = wx * WS + ix;
        return 3 * ((Y >= res - WS) + (Y >= res - shift)) + (X >= res - WS) + (X >= res - shift);
    }
};
__device__ __forceinline__ int rel_index(int tq, int tk) {
    const int yq = tq / WS, xq = tq - yq * WS, yk = tk / WS, xk = tk - yk * WS;
    return (yq - yk + WS - 1) * (2 * WS - 1) + (xq - xk + WS - 1);
}

template <typename T>
__device__ __forceinline__ void load_row32(const T* p, float* o) {
#pragma unroll
    for (int c = 0; c < 4; ++c) unpack8<T>(*(const s16x8*)(p + c * 8), o + c * 8);
}
template <typename T>
__device__ __forceinline__ void store_row32(T* p, const float* o) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *(s16x8*)(p + c * 8) = pack8<T>(o + c * 8);
}

// grid (window chunks, heads), block 64.  qkv [B*L, 3C] (q | k | v, head-major inside each), out [B*L, C], lse [B*L, heads]
template <typename T>
__global__ __launch_bounds__(64) void win_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                           const float* __restrict__ table, int n_win_total, int win_per_img, WinGeom g,
                                                           int heads, float scale, int wpb) {
    __shared__ float sk[WT][HD + 1], sv[WT][HD + 1], stab[NB];   // +1: lane-per-row writes without bank conflicts
    __shared__ int sreg[WT];
    const int lane = threadIdx.x, h = blockIdx.y, C = heads * HD, L = g.res * g.res;
    for (int i = lane; i < NB; i += 64) stab[i] = table[i * heads + h];
    for (int wi = 0; wi < wpb; ++wi) {
        const int gw = blockIdx.x * wpb + wi;
        if (gw >= n_win_total) break;
        const int img = gw / win_per_img, w = gw - img * win_per_img;
        __syncthreads();   // previous window's LDS reads are done (and the bias table is visible)
        int tok = 0;
        float q[HD];
        if (lane < WT) {
            tok = g.token(w, lane);
            const T* row = qkv + ((int64_t)img * L + tok) * 3 * C + h * HD;
            float kk[HD], vv[HD];
            load_row32<T>(row, q);
            load_row32<T>(row + C, kk);
            load_row32<T>(row + 2 * C, vv);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                q[d] *= scale;
                sk[lane][d] = kk[d];
                sv[lane][d] = vv[d];
            }
            sreg[lane] = g.region(w, lane);
        }
        __syncthreads();
        if (lane < WT) {
            const int myreg = sreg[lane];
            float m = -3.0e38f, l = 0.f, o[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = 0.f;
            for (int j = 0; j < WT; ++j) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s = fmaf(q[d], sk[j][d], s);
                s += stab[rel_index(lane, j)];
                if (sreg[j] != myreg) s -= 100.0f;
                const float mn = fmaxf(m, s), corr = __expf(m - mn), p = __expf(s - mn);
                l = l * corr + p;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] = fmaf(p, sv[j][d], o[d] * corr);
                m = mn;
            }
            const float inv = 1.0f / l;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= inv;
            store_row32<T>(out + ((int64_t)img * L + tok) * C + h * HD, o);
            lse[((int64_t)img * L + tok) * heads + h] = m + __logf(l);
        }
    }
}

// dout [B*L, C] (16-bit, carries the caller's gradient scale), dqkv [B*L, 3C] written completely (every token is in exactly one window),
// dtable [169, heads] += dtab_scale * sum of dS over the workgroup's windows.
template <typename T>
__global__ __launch_bounds__(64) void win_attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ table, T* __restrict__ dqkv, float* __restrict__ dtable,
                                                           int n_win_total, int win_per_img, WinGeom g, int heads, float scale,
                                                           float dtab_scale, int wpb) {
    __shared__ float sq[WT][HD + 1], sk[WT][HD + 1], sv[WT][HD + 1], sdo[WT][HD + 1], stab[NB], sdtab[NB], slse[WT], sD[WT];
    __shared__ int sreg[WT];
    const int lane = threadIdx.x, h = blockIdx.y, C = heads * HD, L = g.res * g.res;
    for (int i = lane; i < NB; i += 64) {
        stab[i] = table[i * heads + h];
        sdtab[i] = 0.f;
    }
    for (int wi = 0; wi < wpb; ++wi) {
        const int gw = blockIdx.x * wpb + wi;
        if (gw >= n_win_total) break;
        const int img = gw / win_per_img, w = gw - img * win_per_img;
        __syncthreads();
        int tok = 0;
        if (lane < WT) {
            tok = g.token(w, lane);
            const T* row = qkv + ((int64_t)img * L + tok) * 3 * C + h * HD;
            float a[HD];
            load_row32<T>(row, a);
#pragma unroll
            for (int d = 0; d < HD; ++d) sq[lane][d] = a[d] * scale;
            load_row32<T>(row + C, a);
#pragma unroll
            for (int d = 0; d < HD; ++d) sk[lane][d] = a[d];
            load_row32<T>(row + 2 * C, a);
#pragma unroll
            for (int d = 0; d < HD; ++d) sv[lane][d] = a[d];
            load_row32<T>(dout + ((int64_t)img * L + tok) * C + h * HD, a);
#pragma unroll
            for (int d = 0; d < HD; ++d) sdo[lane][d] = a[d];
            slse[lane] = lse[((int64_t)img * L + tok) * heads + h];
            sreg[lane] = g.region(w, lane);
        }
        __syncthreads();
        // ---- lane = query row i: D_i = sum_j P_ij dP_ij, then dQ_i = scale * sum_j dS_ij K_j and the bias-table bins ----
        if (lane < WT) {
            float q[HD], dO[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                q[d] = sq[lane][d];
                dO[d] = sdo[lane][d];
            }
            const int myreg = sreg[lane];
            const float mylse = slse[lane];
            float D = 0.f;
            for (int j = 0; j < WT; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s = fmaf(q[d], sk[j][d], s);
                    dp = fmaf(dO[d], sv[j][d], dp);
                }
                s += stab[rel_index(lane, j)];
                if (sreg[j] != myreg) s -= 100.0f;
                D = fmaf(__expf(s - mylse), dp, D);
            }
            sD[lane] = D;
            float dq[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = 0.f;
            for (int j = 0; j < WT; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s = fmaf(q[d], sk[j][d], s);
                    dp = fmaf(dO[d], sv[j][d], dp);
                }
                const int ri = rel_index(lane, j);
                s += stab[ri];
                if (sreg[j] != myreg) s -= 100.0f;
                const float ds = __expf(s - mylse) * (dp - D);
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, sk[j][d], dq[d]);
                atomicAdd(&sdtab[ri], ds);
            }
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] *= scale;
            store_row32<T>(dqkv + ((int64_t)img * L + tok) * 3 * C + h * HD, dq);
        }
        __syncthreads();
        // ---- lane = key row j: dV_j = sum_i P_ij dO_i, dK_j = sum_i dS_ij (scale q_i) ----
        if (lane < WT) {
            float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                k[d] = sk[lane][d];
                v[d] = sv[lane][d];
                dk[d] = 0.f;
                dv[d] = 0.f;
            }
            const int myreg = sreg[lane];
            for (int i = 0; i < WT; ++i) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s = fmaf(sq[i][d], k[d], s);
                    dp = fmaf(sdo[i][d], v[d], dp);
                }
                s += stab[rel_index(i, lane)];
                if (sreg[i] != myreg) s -= 100.0f;
                const float p = __expf(s - slse[i]), ds = p * (dp - sD[i]);
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dv[d] = fmaf(p, sdo[i][d], dv[d]);
                    dk[d] = fmaf(ds, sq[i][d], dk[d]);
                }
            }
            T* row = dqkv + ((int64_t)img * L + tok) * 3 * C + h * HD;
            store_row32<T>(row + C, dk);
            store_row32<T>(row + 2 * C, dv);
        }
    }
    __syncthreads();
    for (int i = lane; i < NB; i += 64) unsafeAtomicAdd(dtable + i * heads + h, sdtab[i] * dtab_scale);
}

// PatchMerging's 2x2 gather (swin.py:340-346): out[b, (y2, x2), q*C + c] = in[b, (2 y2 + (q & 1), 2 x2 + (q >> 1)), c] - the reference
// concatenates x0 = [0::2, 0::2], x1 = [1::2, 0::2], x2 = [0::2, 1::2], x3 = [1::2, 1::2].  A bijection: backward = the same map reversed.
__global__ void patch_merge_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int res, int C, int backward) {
    const int r2 = res / 2, c4 = C / 4;
    const int64_t total = (int64_t)B * res * res * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        int64_t r = i / c4;
        const int q = (int)(r % 4);
        r /= 4;
        const int x2 = (int)(r % r2);
        r /= r2;
        const int y2 = (int)(r % r2), b = (int)(r / r2);
        const int64_t merged = (((int64_t)b * r2 + y2) * r2 + x2) * 4 * C + (int64_t)q * C + c * 4;
        const int64_t plain = (((int64_t)b * res + 2 * y2 + (q & 1)) * res + 2 * x2 + (q >> 1)) * C + c * 4;
        if (backward) *(f32x4*)(out + plain) = *(const f32x4*)(in + merged);
        else *(f32x4*)(out + merged) = *(const f32x4*)(in + plain);
    }
}

inline int win_chunk(int n_win_total) { return n_win_total >= 8192 ? 8 : (n_win_total >= 1024 ? 4 : 1); }
}  // namespace

#define ST ((hipStream_t)stream)
extern "C" int mico_win_attn_fwd(const void* qkv, void* out, float* lse, const float* bias_table, int batch, int res, int heads, int shift,
                                 float scale, int dtype, void* stream) {
    MICO_CHECK(qkv && out && lse && bias_table && dtype_ok(dtype), "mico_win_attn_fwd: bad args");
    MICO_CHECK(batch > 0 && heads > 0 && res >= 7 && res % 7 == 0 && shift >= 0 && shift < 7 && (res > 7 || shift == 0),
               "mico_win_attn_fwd: 7x7 windows over a res x res grid (res a multiple of 7), shift < 7 and 0 when the grid is one window");
    const int nw = res / 7, n_win = batch * nw * nw, wpb = win_chunk(n_win);
    const WinGeom g{res, nw, shift};
    const dim3 grid((n_win + wpb - 1) / wpb, heads);
    DISPATCH_T16(dtype, MICO_LAUNCH((win_attn_fwd_kernel<T>), grid, dim3(64), 0, ST, (const T*)qkv, (T*)out, lse, bias_table, n_win, nw * nw, g,
                                     heads, scale, wpb));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_win_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* bias_table, void* dqkv, float* dbias_table,
                                 int batch, int res, int heads, int shift, float scale, float dbias_scale, int dtype, void* stream) {
    MICO_CHECK(qkv && dout && lse && bias_table && dqkv && dbias_table && dtype_ok(dtype), "mico_win_attn_bwd: bad args");
    MICO_CHECK(batch > 0 && heads > 0 && res >= 7 && res % 7 == 0 && shift >= 0 && shift < 7 && (res > 7 || shift == 0),
               "mico_win_attn_bwd: 7x7 windows over a res x res grid (res a multiple of 7), shift < 7 and 0 when the grid is one window");
    const int nw = res / 7, n_win = batch * nw * nw, wpb = win_chunk(n_win);
    const WinGeom g{res, nw, shift};
    const dim3 grid((n_win + wpb - 1) / wpb, heads);
    DISPATCH_T16(dtype, MICO_LAUNCH((win_attn_bwd_kernel<T>), grid, dim3(64), 0, ST, (const T*)qkv, (const T*)dout, lse, bias_table, (T*)dqkv,
                                     dbias_table, n_win, nw * nw, g, heads, scale, dbias_scale, wpb));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_patch_merge(const float* in, float* out, int batch, int res, int channels, int backward, void* stream) {
    MICO_CHECK(in && out && batch > 0 && res > 0 && res % 2 == 0 && channels % 4 == 0, "mico_patch_merge: bad args");
    const int64_t total = (int64_t)batch * res * res * (channels / 4);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    MICO_LAUNCH(patch_merge_kernel, dim3(blocks), dim3(256), 0, ST, in, out, batch, res, channels, backward);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
