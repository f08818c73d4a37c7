// LayerNorm forward / backward for gfx950 (HBM-bound row kernels; see include/mico_hip.h).
// One wave64 owns one row at a time and keeps it in registers (cols <= 4096): a single HBM read per element,
// 16-byte loads/stores, fp32 statistics via two in-register passes (mean, then centred variance - the same
// arithmetic order as torch.nn.functional.layer_norm).  The backward accumulates dgamma/dbeta per lane across the
// rows a wave walks, reduces the 4 waves of a block through LDS and leaves one partial row per block; a second tiny
// kernel folds the partials into the fp32 parameter gradients.
#include "common.h"

namespace {

// Non-temporal hints on the row streams (every element is touched once per launch): bit 0 = stores, bit 1 = loads.  Measured with
// tools/ln_bench.py at 717 frames x 257 x 1408 (tools/probes/README.md "Round 4: LayerNorm"): backward in situ 802 -> 770 us, forward
// 312 -> 289 us with both; -DMICO_LN_NT=0 rebuilds the plain form.
#ifndef MICO_LN_NT
#define MICO_LN_NT 3
#endif
template <typename V> __device__ __forceinline__ void st_stream(V* p, V v) {
    if constexpr ((MICO_LN_NT & 1) != 0) __builtin_nontemporal_store(v, p); else *p = v;
}
template <typename V> __device__ __forceinline__ V ld_stream(const V* p) {
    if constexpr ((MICO_LN_NT & 2) != 0) return __builtin_nontemporal_load(p); else return *p;
}
constexpr int MAXV = 16;   // float4 per lane -> cols <= 4096 (forward); backward supports cols <= 2048
constexpr int LN_BLOCK = 256;

template <typename XT> struct RowIO;
template <> struct RowIO<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return ld_stream((const f32x4*)p); }
};
template <> struct RowIO<f16> {
    static __device__ __forceinline__ f32x4 load(const f16* p) { return unpack4<f16>(ld_stream((const s16x4*)p)); }
};
template <> struct RowIO<bf16> {
    static __device__ __forceinline__ f32x4 load(const bf16* p) { return unpack4<bf16>(ld_stream((const s16x4*)p)); }
};

// LEAN: the towers' block LayerNorms use neither the post-add table, nor dropout, nor an fp32 output, nor the hi|lo split output
// (bf16 configuration): compiled without them
// MX: the 16-bit output is ALSO written as the block-scaled fp8 operand of mico_gemm_mx8 (e4m3 + one E8M0 scale per 32 columns, the layout and
// the arithmetic of quant_mx8_kernel - the rounded 16-bit values are what gets quantised, so the result is bit-identical to quantising y16 in
// a pass of its own): mico_layernorm_fwd_mx8, the fp8 mode's LayerNorm-fed GEMMs (qkv, fc1)
template <typename T, typename XT, int NV, bool LEAN = false, bool MX = false>
__global__ __launch_bounds__(LN_BLOCK) void ln_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y16,
                                                           float* __restrict__ y32, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int64_t rows, int cols, float eps,
                                                           const float* __restrict__ post_add, int post_rpg,
                                                           int post_groups, int split16,
                                                           const int* __restrict__ frame_map, int rpf,
                                                           float* __restrict__ x_copy, float drop_p, unsigned drop_seed,
                                                           int drop_site, int valid_cols, f16* __restrict__ xhat16, int x_norm,
                                                           unsigned char* __restrict__ q8 = nullptr,
                                                           int64_t ldq = 0, unsigned* __restrict__ sc8 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform for the compiler: row bases, statistics and frame lookups go to SGPRs)
    const int nv = cols >> 2;
    // valid_cols < cols: the row is a zero-padded [valid_cols | 0 ...] vector (EVA02-L's 2730-wide SwiGLU hidden in a 2752-wide buffer):
    // statistics over the valid columns only - the zeros add nothing to the sum and exactly (cols - valid) * mean^2 to the centred sum of
    // squares, which is taken out in closed form; gamma / beta are zero-padded by the caller, so the padded outputs are 0
    const float inv = 1.0f / (float)valid_cols;
    const float npad = (float)(cols - valid_cols);
    // the next row of this wave is requested before the current one is reduced and stored (one row at a time the wave has no load in
    // flight during its two reductions and its stores)
    auto fetch = [&](int64_t row, f32x4 (&dst)[NV]) {
        if (row >= rows) return;
        int64_t srow = row;
        if (frame_map) { const int64_t f = row / rpf; srow = (int64_t)frame_map[f] * rpf + (row - f * rpf); }
        const XT* xr = x + srow * cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) dst[i] = RowIO<XT>::load(xr + c * 4);
        }
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    f32x4 vn[NV];
    fetch((int64_t)blockIdx.x * 4 + wave, vn);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += stride) {
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vn[i];
        fetch(row + stride, vn);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                if (x_copy) st_stream((f32x4*)(x_copy + row * cols + c * 4), v[i]);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        // x_norm: the input rows ARE (x - mean) * rstd already (the 16-bit xhat16 copy an earlier forward left for the backward - the
        // activation diet's LayerNorm recompute): no statistics, y = x gamma + beta
        float mean = 0.f, rstd = 1.f;
        if (!x_norm) {
            mean = wave_sum(s) * inv;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 64 + lane;
                if (c < nv) {
                    f32x4 d = v[i] - mean;
                    q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                }
            }
            rstd = rsqrtf(fmaxf(wave_sum(q) - npad * mean * mean, 0.f) * inv + eps);
            if (lane == 0) {
                if (mean_o) mean_o[row] = mean;
                if (rstd_o) rstd_o[row] = rstd;
            }
        }
        const float* pa = (!LEAN && post_add) ? post_add + (int64_t)((row / post_rpg) % post_groups) * cols : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                // (gamma / beta are re-read per row from L1: held in registers across rows they push the kernel past 128 VGPRs and
                // it loses more occupancy than the saved L1 traffic buys - measured 207 vs 181 us at the tower shape)
                f32x4 g = *(const f32x4*)(gamma + c * 4), b = *(const f32x4*)(beta + c * 4);
                const f32x4 nh = (v[i] - mean) * rstd;
                if (xhat16) st_stream((s16x4*)(xhat16 + row * cols + c * 4), pack4<f16>(nh[0], nh[1], nh[2], nh[3]));
                f32x4 o = nh * g + b;
                if (pa) o += *(const f32x4*)(pa + c * 4);
                if (!LEAN && drop_p > 0.f) {
                    const unsigned thr = drop_threshold(drop_p);
                    const float ik = 1.f / (1.f - drop_p);
                    const unsigned long long i0 = (unsigned long long)row * cols + c * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] *= drop_mult(drop_seed, drop_site, i0 + k, thr, ik);
                }
                if (!LEAN && y32) *(f32x4*)(y32 + row * cols + c * 4) = o;
                if (MX) {
                    // cols % 128 == 0: the 8 lanes of a 32-column block and the 32 lanes of a 128-column scale word are all inside or all
                    // outside the row (c < nv), so the lane exchanges below stay among active lanes
                    const f32x4 r4 = unpack4<T>(pack4<T>(o[0], o[1], o[2], o[3]));
                    float amax = fmaxf(fmaxf(fabsf(r4[0]), fabsf(r4[1])), fmaxf(fabsf(r4[2]), fabsf(r4[3])));
                    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                    const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
                    int e = (int)((bits >> 23) & 0xFF) - 127 + ((bits & 0x7FFFFF) ? 1 : 0);
                    e = amax > 0.f ? max(-127, min(127, e)) : -127;
                    const float sinv = __uint_as_float((unsigned)(127 - e) << 23);
                    unsigned w0 = 0;
                    if (amax > 0.f && e > -127) {
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(r4[0] * sinv, r4[1] * sinv, w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(r4[2] * sinv, r4[3] * sinv, w0, true);
                    }
                    *(unsigned*)(q8 + row * ldq + c * 4) = w0;
                    const unsigned sbyte = (unsigned)(e + 127);
                    const int base = lane & ~31;
                    const unsigned word = (__shfl(sbyte, base, 64) & 0xFF) | ((__shfl(sbyte, base + 8, 64) & 0xFF) << 8) |
                                          ((__shfl(sbyte, base + 16, 64) & 0xFF) << 16) | ((__shfl(sbyte, base + 24, 64) & 0xFF) << 24);
                    if ((lane & 31) == 0) sc8[(int64_t)(i * 2 + (lane >> 5)) * rows + row] = word;
                }
                if (y16) {
                    if (LEAN || !split16) {
                        st_stream((s16x4*)(y16 + row * cols + c * 4), pack4<T>(o[0], o[1], o[2], o[3]));
                    } else {   // [rows, 2*cols]: hi | lo halves (split-precision GEMM operand)
                        const s16x4 hi = pack4<T>(o[0], o[1], o[2], o[3]);
                        const f32x4 hf = unpack4<T>(hi);
                        *(s16x4*)(y16 + row * 2 * cols + c * 4) = hi;
                        *(s16x4*)(y16 + row * 2 * cols + cols + c * 4) = pack4<T>(o[0] - hf[0], o[1] - hf[1], o[2] - hf[2], o[3] - hf[3]);
                    }
                }
            }
        }
    }
}

// y = xhat * gamma + beta over fp16 normalised rows (mico_ln_fwd_params::x_normalized, the activation diet's LayerNorm recompute): no row
// statistics, so no row ownership either - a thread keeps the gamma / beta of its 8 columns in registers and walks down the rows (the row
// kernel re-reads both vectors for every row: 11 KB through L1 next to a row's 2.8 + 2.8 KB, and ran this case at 2.9 TB/s - 355 us at 717
// frames; tools/ln_bench.py).  16-byte loads and stores, four rows in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void ln_affine16_kernel(const f16* __restrict__ xh, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ y, int64_t rows, int cols) {
    const int chunk = blockIdx.x * blockDim.x + threadIdx.x;      // 8 columns
    if (chunk * 8 >= cols) return;
    const f32x4 g0 = *(const f32x4*)(gamma + chunk * 8), g1 = *(const f32x4*)(gamma + chunk * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(beta + chunk * 8), b1 = *(const f32x4*)(beta + chunk * 8 + 4);
    const int64_t stride = gridDim.y;
    constexpr int U = 4;
    for (int64_t r0 = blockIdx.y; r0 < rows; r0 += stride * U) {
        s16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            if (r < rows) v[u] = ld_stream((const s16x8*)(xh + r * cols + chunk * 8));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            if (r < rows) {
                const f32x4 lo = unpack4<f16>((s16x4){v[u][0], v[u][1], v[u][2], v[u][3]}) * g0 + b0;
                const f32x4 hi = unpack4<f16>((s16x4){v[u][4], v[u][5], v[u][6], v[u][7]}) * g1 + b1;
                const s16x4 pl = pack4<T>(lo[0], lo[1], lo[2], lo[3]), ph = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
                st_stream((s16x8*)(y + r * cols + chunk * 8), (s16x8){pl[0], pl[1], pl[2], pl[3], ph[0], ph[1], ph[2], ph[3]});
            }
        }
    }
}

template <typename T, typename DT, typename XT, int NV>
__global__ __launch_bounds__(LN_BLOCK) void ln_bwd_kernel(const DT* __restrict__ dy, const XT* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ dx_add,
                                                           float* __restrict__ dx32, T* __restrict__ dx16, float scale16,
                                                           float* __restrict__ ws, int64_t rows, int cols, float dy_scale,
                                                           const int* __restrict__ frame_map, int rpf, int valid_cols,
                                                           const int* __restrict__ dx16_dst, const float* __restrict__ dx16_fscale,
                                                           float d16_drop_p, unsigned d16_drop_seed, int d16_drop_site, int x_norm) {
    __shared__ f32x4 red[2][4][64];   // per (gamma/beta, wave, lane) scratch, reused per column slab
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform for the compiler: row bases, statistics and frame lookups go to SGPRs)
    const int nv = cols >> 2;
    const float inv = 1.0f / (float)valid_cols;   // (zero-padded rows: gamma is zero-padded, so the padded columns add nothing to either mean)
    f32x4 dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        // x_norm: x holds the normalised rows (x - mean) * rstd themselves (the forward's 16-bit xhat16 copy): no mean needed
        const float rs = rstd[row];
        const float mu = x_norm ? 0.f : mean[row], xs = x_norm ? 1.f : rs;
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4 d = RowIO<DT>::load(dy + row * cols + c * 4) * dy_scale;
                xh[i] = (RowIO<XT>::load(x + row * cols + c * 4) - mu) * xs;
                g[i] = d * *(const f32x4*)(gamma + c * 4);
                dg[i] += d * xh[i];
                db[i] += d;
                s1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
                f32x4 t = g[i] * xh[i];
                s2 += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        const float c1 = wave_sum(s1) * inv, c2 = wave_sum(s2) * inv;
        int64_t orow = row;   // scattered row of the residual-gradient stream (dx_add / dx32)
        int64_t fidx = 0, fpos = 0;
        if (frame_map || dx16_dst) { fpos = row / rpf; fidx = frame_map ? frame_map[fpos] : fpos; }
        if (frame_map) orow = fidx * rpf + (row - fpos * rpf);
        // dx16 for the NEXT consumer's frame set (dx16_dst): frame slot dx16_dst[compact frame] of a compact 16-bit buffer, or none (< 0);
        // its per-frame multiplier dx16_fscale[scattered frame] rides along
        int64_t drow = row;
        float s16 = scale16;
        bool w16 = dx16 != nullptr;
        if (dx16_dst) {
            const int d = dx16_dst[fpos];
            w16 = d >= 0;
            drow = (int64_t)d * rpf + (row - fpos * rpf);
            if (dx16_fscale) s16 *= dx16_fscale[fidx];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4 o = (g[i] - c1 - xh[i] * c2) * rs;
                if (dx_add) o += ld_stream((const f32x4*)(dx_add + orow * cols + c * 4));
                if (dx32) st_stream((f32x4*)(dx32 + orow * cols + c * 4), o);
                if (w16) {
                    o *= s16;
                    if (d16_drop_p > 0.f) {   // the dense branch sat behind a dropout in the forward: the same mask multiplies its gradient
                        const unsigned thr = drop_threshold(d16_drop_p);
                        const float ik = 1.f / (1.f - d16_drop_p);
                        const unsigned long long i0 = (unsigned long long)drow * cols + c * 4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] *= drop_mult(d16_drop_seed, d16_drop_site, i0 + k, thr, ik);
                    }
                    st_stream((s16x4*)(dx16 + drow * cols + c * 4), pack4<T>(o[0], o[1], o[2], o[3]));
                }
            }
        }
    }
    if (!ws) return;
    // block reduce of dgamma/dbeta partials -> ws[0][blk][cols], ws[1][blk][cols]
    float* wg = ws + (int64_t)blockIdx.x * cols;
    float* wb = ws + ((int64_t)gridDim.x + blockIdx.x) * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * 64 >= nv) break;
        __syncthreads();
        red[0][wave][lane] = dg[i];
        red[1][wave][lane] = db[i];
        __syncthreads();
        if (wave == 0) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4 a = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
                f32x4 b = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
                *(f32x4*)(wg + c * 4) = a;
                *(f32x4*)(wb + c * 4) = b;
            }
        }
    }
}

// folds the per-block partials: 32 columns x 8 partial-row lanes per workgroup, grid.y splits the partial rows further and
// the few resulting partial sums are combined with atomics (dgamma/dbeta are accumulated into anyway)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblk, int cols,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float scale) {
    __shared__ float ra[8][33], rb[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int k0 = blockIdx.y * per, k1 = min(nblk, k0 + per);
    float a = 0.f, b = 0.f;
    if (c < cols) {
        for (int k = k0 + ry; k < k1; k += 8) {
            a += ws[(int64_t)k * cols + c];
            b += ws[((int64_t)nblk + k) * cols + c];
        }
    }
    ra[ry][cx] = a;
    rb[ry][cx] = b;
    __syncthreads();
    if (ry == 0 && c < cols) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { a += ra[k][cx]; b += rb[k][cx]; }
        if (dgamma) unsafeAtomicAdd(dgamma + c, a * scale);
        if (dbeta) unsafeAtomicAdd(dbeta + c, b * scale);
    }
}

// workgroups of 4 waves walking the rows grid-stride.  The backward runs 18 % faster with 2048 than with 1024 workgroups at the
// tower's 82k rows (tools/ln_bench.py); the forward does not care.
int ln_grid(int64_t rows, int cap) {
    int64_t nb = (rows + 3) / 4;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

extern "C" int mico_layernorm_bwd_nblk(int64_t rows);

template <typename T, typename XT>
void ln_fwd_launch(dim3 grid, hipStream_t st, const mico_ln_fwd_params& p, int valid) {
    const int cols = p.cols;
    unsigned char* q8 = (unsigned char*)p.q8;
    unsigned* sc8 = (unsigned*)p.scales;
#define LN_ARGS (const XT*)p.x, p.gamma, p.beta, (T*)p.y16, p.y32, p.mean, p.rstd, p.rows, cols, p.eps, p.post_add, p.post_rows_per_group, \
                p.post_groups, p.y16_split, p.frame_map, p.rows_per_frame, p.x_copy, p.drop_p, p.drop_seed, p.drop_site, valid, (f16*)p.xhat16, \
                p.x_normalized
    if (q8) {   // fp8 mode: the 16-bit output also as the block-scaled e4m3 operand (cols % 128 == 0, <= 2048: checked by the caller)
#define LNMX(NV) MICO_LAUNCH((ln_fwd_kernel<T, XT, NV, false, true>), grid, dim3(LN_BLOCK), 0, st, LN_ARGS, q8, p.ldq, sc8)
        if (cols <= 1024) LNMX(4); else if (cols <= 1536) LNMX(6); else LNMX(8);
#undef LNMX
        return;
    }
#define LNF(NV) MICO_LAUNCH((ln_fwd_kernel<T, XT, NV>), grid, dim3(LN_BLOCK), 0, st, LN_ARGS)
    if (!p.post_add && p.drop_p == 0.f && !p.y32 && !p.y16_split && cols > 1024 && cols <= 1536) {
        MICO_LAUNCH((ln_fwd_kernel<T, XT, 6, true>), grid, dim3(LN_BLOCK), 0, st, LN_ARGS);
        return;
    }
    if (cols <= 1024) LNF(4);
    else if (cols <= 1536) LNF(6);
    else if (cols <= 2048) LNF(8);
    else if (cols <= 3072) LNF(12);
    else LNF(16);
#undef LNF
#undef LN_ARGS
}

template <typename T, typename DT, typename XT>
void ln_bwd_launch(dim3 grid, hipStream_t st, const mico_ln_bwd_params& p, float* wsp, int valid) {
    const int cols = p.cols;
#define LNB(NV) MICO_LAUNCH((ln_bwd_kernel<T, DT, XT, NV>), grid, dim3(LN_BLOCK), 0, st, (const DT*)p.dy, (const XT*)p.x, p.gamma, p.mean, p.rstd, \
                            p.dx_add, p.dx32, (T*)p.dx16, p.scale16, wsp, p.rows, cols, p.dy_scale, p.frame_map, p.rows_per_frame, valid, p.dx16_dst, \
                            p.dx16_frame_scale, p.dx16_drop_p, p.dx16_drop_seed, p.dx16_drop_site, p.x_normalized)
    if (cols <= 1024) LNB(4);
    else if (cols <= 1536) LNB(6);
    else if (cols <= 2048) LNB(8);
    else if (cols <= 2816) LNB(11);
    else LNB(12);   // 3072 = the 4C of Swin-L's last PatchMerging norm (only user)
#undef LNB
}

}  // namespace

extern "C" int mico_layernorm_bwd_nblk(int64_t rows) { return ln_grid(rows, 2048); }

extern "C" int mico_layernorm_fwd(const mico_ln_fwd_params* pp, int dtype, void* stream) {
    MICO_CHECK(pp != nullptr, "mico_layernorm_fwd: null parameter struct");
    const mico_ln_fwd_params& p = *pp;
    MICO_CHECK(dtype_ok(dtype), "mico_layernorm_fwd: bad dtype");
    const int cols = p.cols;
    const int valid_cols = p.valid_cols <= 0 ? cols : p.valid_cols;
    MICO_CHECK(valid_cols <= cols, "mico_layernorm_fwd: valid_cols > cols");
    if (p.rows <= 0) return MICO_OK;   // an empty batch is a no-op (its tensors have no storage to point to)
    MICO_CHECK(p.x && p.gamma && p.beta && (p.y16 || p.y32), "mico_layernorm_fwd: null pointer");
    MICO_CHECK(cols % 4 == 0 && cols > 0 && cols <= MAXV * 256, "mico_layernorm_fwd: cols must be a multiple of 4 and <= %d (got %d)", MAXV * 256, cols);
    if (p.x_normalized) {
        MICO_CHECK(p.x_dtype == MICO_F16, "mico_layernorm_fwd: a normalised input (x_normalized) is the fp16 xhat16 copy of an earlier forward");
        MICO_CHECK(!p.mean && !p.rstd && !p.x_copy && !p.xhat16, "mico_layernorm_fwd: x_normalized produces no statistics and no input copies");
    } else {
        MICO_CHECK(p.x_dtype == MICO_F32 || p.x_dtype == dtype, "mico_layernorm_fwd: x_dtype must be fp32 or dtype");
    }
    if (p.post_add) MICO_CHECK(p.post_rows_per_group > 0 && p.post_groups > 0, "mico_layernorm_fwd: bad post_add grouping");
    if (p.frame_map) MICO_CHECK(p.rows_per_frame > 0, "mico_layernorm_fwd: frame_map needs rows_per_frame > 0");
    if (p.q8) {
        MICO_CHECK(p.y16 && p.scales, "mico_layernorm_fwd: the fp8 operand (q8) needs y16 and scales");
        MICO_CHECK(cols % 128 == 0 && cols <= 2048, "mico_layernorm_fwd: the fp8 operand needs cols %% 128 == 0 and <= 2048 (got %d)", cols);
        MICO_CHECK(p.ldq >= cols && p.ldq % 8 == 0, "mico_layernorm_fwd: ldq must be >= cols and a multiple of 8");
        MICO_CHECK(!p.y32 && !p.post_add && p.drop_p == 0.f && !p.y16_split && valid_cols == cols, "mico_layernorm_fwd: the fp8 operand form takes the towers' subset of the features");
    }
    hipStream_t st = (hipStream_t)stream;
    if (p.x_normalized && p.y16 && !p.y32 && !p.q8 && !p.post_add && !p.y16_split && !p.frame_map && p.drop_p == 0.f && cols % 8 == 0 && valid_cols == cols &&
        (((uintptr_t)p.x | (uintptr_t)p.y16) & 15) == 0) {
        // the recompute from normalised rows: a streaming affine map, no row statistics (ln_affine16_kernel)
        const int chunks = cols / 8;
        const int bx = chunks >= 192 ? 192 : 64;
        const dim3 grid2((chunks + bx - 1) / bx, (unsigned)std::min<int64_t>(p.rows, 2048));
        DISPATCH_T16(dtype, MICO_LAUNCH((ln_affine16_kernel<T>), grid2, dim3(bx), 0, st, (const f16*)p.x, p.gamma, p.beta, (T*)p.y16, p.rows, cols));
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    const dim3 grid(ln_grid(p.rows, 1024));
    DISPATCH_T16(dtype, {
        if (p.x_dtype == MICO_F32) ln_fwd_launch<T, float>(grid, st, p, valid_cols);
        else if (p.x_dtype == dtype) ln_fwd_launch<T, T>(grid, st, p, valid_cols);
        else ln_fwd_launch<T, f16>(grid, st, p, valid_cols);     // (the fp16 xhat16 rows under bf16 compute)
    });
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_layernorm_bwd(const mico_ln_bwd_params* pp, int dtype, void* stream) {
    MICO_CHECK(pp != nullptr, "mico_layernorm_bwd: null parameter struct");
    const mico_ln_bwd_params& p = *pp;
    MICO_CHECK(dtype_ok(dtype), "mico_layernorm_bwd: bad dtype");
    MICO_CHECK(p.dx16_drop_p >= 0.f && p.dx16_drop_p < 1.f && (p.dx16_drop_p == 0.f || p.dx16), "mico_layernorm_bwd: dx16_drop_p must be in [0, 1) and needs dx16");
    if (p.dx16_dst) MICO_CHECK(p.dx16 && p.rows_per_frame > 0, "mico_layernorm_bwd: dx16_dst needs dx16 and rows_per_frame > 0");
    const int cols = p.cols;
    const int valid_cols = p.valid_cols <= 0 ? cols : p.valid_cols;
    MICO_CHECK(valid_cols <= cols, "mico_layernorm_bwd: valid_cols > cols");
    if (p.rows <= 0) return MICO_OK;
    if (p.frame_map) MICO_CHECK(p.rows_per_frame > 0, "mico_layernorm_bwd: frame_map needs rows_per_frame > 0");
    MICO_CHECK(p.dy && p.x && p.gamma && p.rstd && (p.mean || p.x_normalized), "mico_layernorm_bwd: null pointer");
    MICO_CHECK(cols % 4 == 0 && cols > 0 && cols <= 3072, "mico_layernorm_bwd: cols must be a multiple of 4 and <= 3072 (got %d)", cols);
    MICO_CHECK(p.dy_dtype == MICO_F32 || p.dy_dtype == dtype, "mico_layernorm_bwd: bad dy dtype");
    if (p.x_normalized) MICO_CHECK(p.x_dtype == MICO_F16, "mico_layernorm_bwd: a normalised x (x_normalized) is the fp16 xhat16 copy of the forward");
    else MICO_CHECK(p.x_dtype == MICO_F32 || p.x_dtype == dtype, "mico_layernorm_bwd: bad x dtype");
    MICO_CHECK(!(p.dgamma || p.dbeta) || p.ws, "mico_layernorm_bwd: dgamma/dbeta need a workspace");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ln_grid(p.rows, 2048);
    const dim3 grid(nblk);
    float* wsp = (p.dgamma || p.dbeta) ? p.ws : nullptr;
    DISPATCH_T16(dtype, {
        if (p.dy_dtype == MICO_F32) {
            if (p.x_dtype == MICO_F32) ln_bwd_launch<T, float, float>(grid, st, p, wsp, valid_cols);
            else if (p.x_dtype == dtype) ln_bwd_launch<T, float, T>(grid, st, p, wsp, valid_cols);
            else ln_bwd_launch<T, float, f16>(grid, st, p, wsp, valid_cols);
        } else {
            if (p.x_dtype == MICO_F32) ln_bwd_launch<T, T, float>(grid, st, p, wsp, valid_cols);
            else if (p.x_dtype == dtype) ln_bwd_launch<T, T, T>(grid, st, p, wsp, valid_cols);
            else ln_bwd_launch<T, T, f16>(grid, st, p, wsp, valid_cols);     // (the fp16 xhat16 rows under bf16 compute)
        }
    });
    MICO_LAUNCH_CHECK();
    if (wsp) {
        MICO_LAUNCH(ln_bwd_reduce_kernel, dim3((cols + 31) / 32, nblk >= 64 ? 16 : 1), dim3(256), 0, st, wsp, nblk, cols, p.dgamma, p.dbeta, p.grad_scale);
        MICO_LAUNCH_CHECK();
    }
    return MICO_OK;
}
