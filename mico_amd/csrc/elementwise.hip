// HBM-bound helpers: casts, row gathers, column sums, im2row, RoPE, SwiGLU gate, BERT embedding gather/scatter,
// L2-normalise.  All of them move 16 bytes per lane where the layout allows and are grid-stride over at most
// 2048 workgroups (cdna_hip_programming.md Guideline 11/13).  See include/mico_hip.h for the contracts.
#include "common.h"

namespace {

constexpr int EB = 256;
inline int egrid(int64_t work_items) {
    int64_t nb = (work_items + EB - 1) / EB;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// ---- casts --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_f32_to_16_kernel(const float* __restrict__ src, int64_t ld_src, T* __restrict__ dst, int64_t ld_dst,
                                      int64_t rows, int cols, int cols_pad, float scale) {
    const int vpr = cols_pad >> 2;   // 4-element groups per output row
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 4;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (c + k < cols) ? src[r * ld_src + c + k] * scale : 0.f;
        *(s16x4*)(dst + r * ld_dst + c) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}

template <typename T>
__global__ void cast_16_to_f32_kernel(const T* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
                                      int64_t rows, int cols, float scale, int accumulate) {
    const int vpr = cols >> 2;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 4;
        f32x4 v = unpack4<T>(*(const s16x4*)(src + r * ld_src + c)) * scale;
        f32x4* d = (f32x4*)(dst + r * ld_dst + c);
        if (accumulate) *d += v;
        else *d = v;
    }
}

template <typename T>
__global__ void gather_rows_cast_kernel(const float* __restrict__ src, int64_t ld_src, T* __restrict__ dst, int64_t ld_dst,
                                        int64_t rows, int cols, int remap_group, int remap_skip, int remap_offset,
                                        const float* __restrict__ row_scale, int rows_per_scale, float scale,
                                        const int* __restrict__ frame_map, int rpf, const int* __restrict__ dst_map) {
    const int vpr = cols >> 2;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 4;
        int64_t rs = r, rd = r;
        if (remap_group) rs = r + (r / remap_group) * remap_skip + remap_offset;
        else if (frame_map) {
            const int64_t f = r / rpf;
            rs = (int64_t)frame_map[f] * rpf + (r - f * rpf);
            if (dst_map) rd = (int64_t)dst_map[f] * rpf + (r - f * rpf);   // frame f of the list lands in frame slot dst_map[f]
        }
        float sc = scale;
        if (row_scale) sc *= row_scale[rs / rows_per_scale];
        f32x4 v = *(const f32x4*)(src + rs * ld_src + c) * sc;
        *(s16x4*)(dst + rd * ld_dst + c) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}

// ---- in-place dropout (gradient side of the hidden-state dropouts) ------------------------------------------------------
template <typename XT>
__global__ void dropout_kernel(XT* __restrict__ x, int64_t rows, int cols, int64_t ld, unsigned thr, float inv_keep, unsigned seed,
                               int site) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        XT* p = x + r * ld + c;
        *p = (XT)((float)*p * drop_mult(seed, site, (unsigned long long)i, thr, inv_keep));
    }
}

// ---- device-side input preprocessing ----------------------------------------------------------------------------------
__global__ void image_preprocess_kernel(const unsigned char* __restrict__ src, int n, int H, int W, float* __restrict__ dst, int oh,
                                        int ow, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t total = (int64_t)n * oh * ow;
    const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int x = (int)(i % ow), y = (int)((i / ow) % oh), f = (int)(i / ((int64_t)ow * oh));
        float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const unsigned char* b = src + (int64_t)f * H * W * 3;
        const unsigned char *p00 = b + ((int64_t)y0 * W + x0) * 3, *p01 = b + ((int64_t)y0 * W + x1) * 3;
        const unsigned char *p10 = b + ((int64_t)y1 * W + x0) * 3, *p11 = b + ((int64_t)y1 * W + x1) * 3;
        const float mean[3] = {m0, m1, m2}, istd[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float k = 1.f / 255.f;
            const float v = hy * (hx * (p00[c] * k) + lx * (p01[c] * k)) + ly * (hx * (p10[c] * k) + lx * (p11[c] * k));
            dst[(((int64_t)f * 3 + c) * oh + y) * ow + x] = (v - mean[c]) * istd[c];
        }
    }
}

__global__ void fbank_windows_kernel(const float* __restrict__ fbank, int T, int mel, const int* __restrict__ win, int n, int tl,
                                     float mean, float inv, float* __restrict__ out) {
    const int64_t total = (int64_t)n * tl * mel;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int m = (int)(i % mel), t = (int)((i / mel) % tl), w = (int)(i / ((int64_t)mel * tl));
        const int64_t row = (int64_t)win[w] * tl + t;
        out[i] = row < T ? (fbank[row * mel + m] - mean) * inv : 0.f;
    }
}

// ---- multi-tensor AdamW (data/utils/build_optimizer.py:105-197) ---------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(const mico_adamw_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                    const int64_t* __restrict__ chunk_start, int chunk_elems, float lr, float beta1,
                                                    float beta2, float eps, float wd, float step_size, float grad_mult) {
    const mico_adamw_tensor t = tensors[chunk_tensor[blockIdx.x]];
    const int64_t s0 = chunk_start[blockIdx.x];
    const int64_t s1 = min(t.numel, s0 + (int64_t)chunk_elems);
    const float ob1 = 1.f - beta1, ob2 = 1.f - beta2;
    auto upd = [&](float p, float g, float& m, float& v) {
        g *= grad_mult;     // 1 / loss scale (GradScaler.unscale_ folded into the update: no pass of its own over the gradients)
        m = m * beta1 + ob1 * g;
        v = v * beta2 + ob2 * g * g;
        p = p - step_size * (m / (sqrtf(v) + eps));
        if (wd > 0.f) p = p - lr * wd * p;
        return p;
    };
    auto mirror = [&](int64_t i, float p) {
        const int64_t r = i / t.cols, c = i - r * t.cols;
        if (t.w16_dtype == MICO_BF16) {
            bf16* d = (bf16*)t.w16 + r * t.ld16 + c;
            const bf16 hi = (bf16)p;
            *d = hi;
            if (t.lo_off > 0) d[t.lo_off] = (bf16)(p - (float)hi);
        } else {
            f16* d = (f16*)t.w16 + r * t.ld16 + c;
            const f16 hi = (f16)p;
            *d = hi;
            if (t.lo_off > 0) d[t.lo_off] = (f16)(p - (float)hi);
        }
    };
    const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0 && (s0 & 3) == 0;
    int64_t i = s0 + (vec ? threadIdx.x * 4 : threadIdx.x);
    if (vec) {
        for (; i + 3 < s1; i += 256 * 4) {
            f32x4 p = *(const f32x4*)(t.p + i), g = *(const f32x4*)(t.g + i), m = *(const f32x4*)(t.m + i), v = *(const f32x4*)(t.v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mk = m[k], vk = v[k];
                p[k] = upd(p[k], g[k], mk, vk);
                m[k] = mk; v[k] = vk;
            }
            *(f32x4*)(t.p + i) = p; *(f32x4*)(t.m + i) = m; *(f32x4*)(t.v + i) = v;
            if (t.w16) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mirror(i + k, p[k]);
            }
        }
        // ragged tail of the tensor (numel % 4): handled by the first threads one element each
        const int64_t tail0 = s1 - ((s1 - s0) & 3);
        i = tail0 + threadIdx.x;
        if (i < s1 && i >= tail0) {
            float mk = t.m[i], vk = t.v[i];
            const float p = upd(t.p[i], t.g[i], mk, vk);
            t.p[i] = p; t.m[i] = mk; t.v[i] = vk;
            if (t.w16) mirror(i, p);
        }
    } else {
        for (; i < s1; i += 256) {
            float mk = t.m[i], vk = t.v[i];
            const float p = upd(t.p[i], t.g[i], mk, vk);
            t.p[i] = p; t.m[i] = mk; t.v[i] = vk;
            if (t.w16) mirror(i, p);
        }
    }
}

// ---- column sums: out[c] (+)= scale * sum_r x[r,c] ----------------------------------------------------------------
// grid (col slabs of 256, row chunks); each thread owns one column for a chunk of rows, partial sums via atomics.
template <typename XT>
__global__ void colsum_kernel(const XT* __restrict__ x, int64_t ld, int64_t rows, int cols, float* __restrict__ out,
                              float scale, int64_t rows_per_chunk) {
    const int c = blockIdx.x * EB + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = min(rows, r0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += (float)x[r * ld + c];
        s1 += (float)x[(r + 1) * ld + c];
        s2 += (float)x[(r + 2) * ld + c];
        s3 += (float)x[(r + 3) * ld + c];
    }
    for (; r < r1; ++r) s0 += (float)x[r * ld + c];
    unsafeAtomicAdd(out + c, ((s0 + s1) + (s2 + s3)) * scale);
}

__global__ void zero_kernel(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) p[i] = 0.f;
}

__global__ void cls_rows_kernel(float* __restrict__ x, int64_t ld, int B, int group_rows, const float* __restrict__ cls,
                                const float* __restrict__ pos0, int cols) {
    const int64_t total = (int64_t)B * cols;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t b = i / cols;
        const int c = (int)(i - b * cols);
        x[b * group_rows * ld + c] = cls[c] + pos0[c];
    }
}

template <typename T>
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                               T* __restrict__ y16, int64_t n4, float scale16) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
        f32x4 v = ((const f32x4*)a)[i];
        if (b) v += ((const f32x4*)b)[i];
        if (y) ((f32x4*)y)[i] = v;
        if (y16) {
            v *= scale16;
            ((s16x4*)y16)[i] = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    }
}

// ---- LayerNorm affine folded into a weight gradient ----------------------------------------------------------------------
// The weight gradient of a Linear fed by y = xhat * gamma + beta (xhat: the fp16 normalised rows the activation diet keeps) is
//     dy^T y = (dy^T xhat) . gamma[n]  +  colsum(dy)[m] beta[n]
// so the backward multiplies against the normalised rows themselves (into the scratch pair dwt / dbt) and this pass adds the result, column-scaled
// and with the rank-one term, to the parameter gradients - instead of re-creating y with a pass over all rows (ln_affine16_kernel: 4 bytes per row
// element against 12 bytes per WEIGHT element here).
template <int V>   // V = 4: 16-byte accesses (every base 16-byte aligned, N and ld_dw multiples of 4);  1: element by element
__global__ void dw_colfold_kernel(const float* __restrict__ dwt, const float* __restrict__ dbt, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  float* __restrict__ dw, int64_t ld_dw, float* __restrict__ db, int M, int nv) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const int64_t total = (int64_t)M * nv;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int m = (int)(i / nv), c = (int)(i - (int64_t)m * nv);
        const vec t = ((const vec*)dwt)[i];
        const vec gm = ((const vec*)gamma)[c], bt = ((const vec*)beta)[c];
        const float s = dbt[m];
        vec* o = (vec*)(dw + (int64_t)m * ld_dw) + c;
        vec v = *o;
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] += __builtin_fmaf(t[k], gm[k], s * bt[k]);
        *o = v;
        if (db && c == 0) db[m] += s;
    }
}

// ---- SwiGLU gate ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ void swiglu_fwd_kernel(const T* __restrict__ x1, const T* __restrict__ x2, T* __restrict__ h, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EB) {
        float a[8], b[8], o[8];
        unpack8<T>(((const s16x8*)x1)[i], a);
        unpack8<T>(((const s16x8*)x2)[i], b);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = a[k] * sigmoid_f(a[k]) * b[k];
        ((s16x8*)h)[i] = pack8<T>(o);
    }
}

__global__ void swiglu_fwd_f32_kernel(const float* __restrict__ x1, const float* __restrict__ x2, float* __restrict__ h, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * EB) {
        const f32x4 a = ((const f32x4*)x1)[i], b = ((const f32x4*)x2)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k] * sigmoid_f(a[k]) * b[k];
        ((f32x4*)h)[i] = o;
    }
}

template <typename T>
__global__ void swiglu_bwd_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ dh,
                                  T* __restrict__ dx1, T* __restrict__ dx2, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EB) {
        float a[8], b[8], d[8], o1[8], o2[8];
        unpack8<T>(((const s16x8*)x1)[i], a);
        unpack8<T>(((const s16x8*)x2)[i], b);
        unpack8<T>(((const s16x8*)dh)[i], d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float s = sigmoid_f(a[k]);
            const float silu = a[k] * s;
            o1[k] = d[k] * b[k] * (s + silu * (1.f - s));
            o2[k] = d[k] * silu;
        }
        ((s16x8*)dx1)[i] = pack8<T>(o1);
        ((s16x8*)dx2)[i] = pack8<T>(o2);
    }
}

// ---- im2row ---------------------------------------------------------------------------------------------------------
// One wave writes one output row (a patch) at a time: kpad 16-bit elements.  Reads are P-element runs of a pixel row.
template <typename T>
__global__ void im2row_kernel(const float* __restrict__ px, T* __restrict__ rows16, int B, int C, int H, int W, int P,
                              int kpad) {
    const int gw = W / P, gh = H / P;
    const int64_t npatch = (int64_t)B * gh * gw;
    const int k_real = C * P * P;
    const int vpr = kpad >> 2;
    const int64_t total = npatch * vpr;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t p = i / vpr;
        const int k0 = (int)(i - p * vpr) * 4;
        const int b = (int)(p / (gh * gw));
        const int pr = (int)(p - (int64_t)b * gh * gw);
        const int py = pr / gw, pxx = pr - py * gw;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            if (k < k_real) {
                const int c = k / (P * P);
                const int rem = k - c * P * P;
                const int ii = rem / P, jj = rem - ii * P;
                v[j] = px[(((int64_t)b * C + c) * H + py * P + ii) * W + pxx * P + jj];
            } else {
                v[j] = 0.f;
            }
        }
        *(s16x4*)(rows16 + p * kpad + k0) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}

// ---- RoPE (pairs (2k, 2k+1) rotated; tokens 1.. only) ---------------------------------------------------------------
template <typename T>
__global__ void rope_kernel(T* __restrict__ x, int64_t bs, int64_t rs, int B, int N, int H, int hd,
                            const float* __restrict__ cos_t, const float* __restrict__ sin_t, int inverse) {
    const int hv = hd >> 2;   // 4 elements (2 pairs) per work item
    const int64_t total = (int64_t)B * (N - 1) * H * hv;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        int64_t t = i;
        const int d0 = (int)(t % hv) * 4; t /= hv;
        const int h = (int)(t % H); t /= H;
        const int n = (int)(t % (N - 1)); t /= (N - 1);
        const int64_t b = t;
        T* p = x + b * bs + (int64_t)(n + 1) * rs + h * hd + d0;
        f32x4 v = unpack4<T>(*(const s16x4*)p);
        const f32x4 c = *(const f32x4*)(cos_t + (int64_t)n * hd + d0);
        f32x4 s = *(const f32x4*)(sin_t + (int64_t)n * hd + d0);
        f32x4 o;
        if (!inverse) {   // y = x*cos + rotate_half(x)*sin ; rotate_half: (x0,x1) -> (-x1, x0)
            o[0] = v[0] * c[0] - v[1] * s[0];
            o[1] = v[1] * c[1] + v[0] * s[1];
            o[2] = v[2] * c[2] - v[3] * s[2];
            o[3] = v[3] * c[3] + v[2] * s[3];
        } else {          // transpose of the above (gradient)
            o[0] = v[0] * c[0] + v[1] * s[1];
            o[1] = v[1] * c[1] - v[0] * s[0];
            o[2] = v[2] * c[2] + v[3] * s[3];
            o[3] = v[3] * c[3] - v[2] * s[2];
        }
        *(s16x4*)p = pack4<T>(o[0], o[1], o[2], o[3]);
    }
}

// ---- BERT embeddings ------------------------------------------------------------------------------------------------
__global__ void bert_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                  const float* __restrict__ pos, const float* __restrict__ type0, float* __restrict__ out,
                                  int64_t rows, int S, int cols, int vocab) {
    const int vpr = cols >> 2;
    const int64_t total = rows * vpr;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / vpr;
        const int c = (int)(i - r * vpr) * 4;
        int64_t id = ids[r];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        const int s = (int)(r % S);
        f32x4 v = *(const f32x4*)(word + id * cols + c) + *(const f32x4*)(type0 + c) + *(const f32x4*)(pos + (int64_t)s * cols + c);
        *(f32x4*)(out + r * cols + c) = v;
    }
}

__global__ void embed_scatter_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dsum, float* __restrict__ dword,
                                     float* __restrict__ dpos, float* __restrict__ dtype0, int64_t rows, int S, int cols,
                                     int vocab, float scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float g = dsum[i] * scale;
        int64_t id = ids[r];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        if (dword) unsafeAtomicAdd(dword + id * cols + c, g);
        if (dpos) unsafeAtomicAdd(dpos + (int64_t)(r % S) * cols + c, g);
        if (dtype0) unsafeAtomicAdd(dtype0 + c, g);
    }
}

// ---- L2 normalise (F.normalize, eps = 1e-12): one wave per row ------------------------------------------------------
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm,
                                  int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[row * cols + c];
        s += v * v;
    }
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < cols; c += 64) y[row * cols + c] = x[row * cols + c] * inv;
    if (lane == 0 && inv_norm) inv_norm[row] = inv;
}

__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv_norm,
                                  float* __restrict__ dx, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += dy[row * cols + c] * y[row * cols + c];
    s = wave_sum(s);
    const float inv = inv_norm[row];
    for (int c = lane; c < cols; c += 64) dx[row * cols + c] = (dy[row * cols + c] - y[row * cols + c] * s) * inv;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int mico_cast_f32_to_16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                                   int cols_pad, float scale, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && src && dst, "mico_cast_f32_to_16: bad args");
    MICO_CHECK(cols_pad % 4 == 0 && cols_pad >= cols && ld_dst % 4 == 0 && ld_dst >= cols_pad, "mico_cast_f32_to_16: cols_pad/ld_dst must be multiples of 4");
    if (rows <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(cast_f32_to_16_kernel<T>, dim3(egrid(rows * (cols_pad / 4))), dim3(EB), 0, ST, src, ld_src, (T*)dst, ld_dst, rows, cols, cols_pad, scale));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_cast_16_to_f32(const void* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int cols,
                                   float scale, int accumulate, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && src && dst, "mico_cast_16_to_f32: bad args");
    MICO_CHECK(cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "mico_cast_16_to_f32: cols/ld must be multiples of 4");
    if (rows <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(cast_16_to_f32_kernel<T>, dim3(egrid(rows * (cols / 4))), dim3(EB), 0, ST, (const T*)src, ld_src, dst, ld_dst, rows, cols, scale, accumulate));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_gather_rows_cast(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                                     int remap_group, int remap_skip, int remap_offset, const float* row_scale,
                                     int rows_per_scale, float scale, const int* frame_map, int rows_per_frame, const int* dst_map,
                                     int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && src && dst, "mico_gather_rows_cast: bad args");
    if (dst_map) MICO_CHECK(frame_map != nullptr, "mico_gather_rows_cast: dst_map goes with frame_map");
    MICO_CHECK(cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "mico_gather_rows_cast: cols/ld must be multiples of 4");
    if (row_scale) MICO_CHECK(rows_per_scale > 0, "mico_gather_rows_cast: rows_per_scale");
    if (frame_map) MICO_CHECK(rows_per_frame > 0 && !remap_group, "mico_gather_rows_cast: frame_map needs rows_per_frame > 0 and no remap_group");
    if (rows <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(gather_rows_cast_kernel<T>, dim3(egrid(rows * (cols / 4))), dim3(EB), 0, ST, src, ld_src, (T*)dst, ld_dst, rows, cols, remap_group, remap_skip, remap_offset, row_scale, rows_per_scale, scale, frame_map, rows_per_frame, dst_map));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_dropout(void* x, int x_dtype, int64_t rows, int cols, int64_t ld, float p, unsigned seed, int site, void* stream) {
    MICO_CHECK(x && cols > 0 && ld >= cols, "mico_dropout: bad args");
    MICO_CHECK(p >= 0.f && p < 1.f, "mico_dropout: p must be in [0, 1)");
    MICO_CHECK(x_dtype == MICO_F32 || x_dtype == MICO_F16 || x_dtype == MICO_BF16, "mico_dropout: bad dtype");
    if (rows <= 0 || p == 0.f) return MICO_OK;
    const unsigned thr = drop_threshold(p);
    const float ik = 1.f / (1.f - p);
    const dim3 grid(egrid(rows * cols));
    if (x_dtype == MICO_F32) MICO_LAUNCH(dropout_kernel<float>, grid, dim3(EB), 0, ST, (float*)x, rows, cols, ld, thr, ik, seed, site);
    else if (x_dtype == MICO_F16) MICO_LAUNCH(dropout_kernel<f16>, grid, dim3(EB), 0, ST, (f16*)x, rows, cols, ld, thr, ik, seed, site);
    else MICO_LAUNCH(dropout_kernel<bf16>, grid, dim3(EB), 0, ST, (bf16*)x, rows, cols, ld, thr, ik, seed, site);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_image_preprocess(const unsigned char* src, int n, int H, int W, float* dst, int out_h, int out_w, float mean0,
                                     float mean1, float mean2, float istd0, float istd1, float istd2, void* stream) {
    MICO_CHECK(src && dst && n > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0, "mico_image_preprocess: bad args");
    MICO_LAUNCH(image_preprocess_kernel, dim3(egrid((int64_t)n * out_h * out_w)), dim3(EB), 0, ST, src, n, H, W, dst, out_h, out_w,
                mean0, mean1, mean2, istd0, istd1, istd2);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_fbank_windows(const float* fbank, int T, int mel, const int* win, int n, int target_len, float mean,
                                  float inv_scale, float* out, void* stream) {
    MICO_CHECK(fbank && win && out && T > 0 && mel > 0 && n > 0 && target_len > 0, "mico_fbank_windows: bad args");
    MICO_LAUNCH(fbank_windows_kernel, dim3(egrid((int64_t)n * target_len * mel)), dim3(EB), 0, ST, fbank, T, mel, win, n, target_len, mean,
                inv_scale, out);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

// any non-finite gradient among the tensors -> *flag = 1 (GradScaler's overflow check, pipeline.py:106 -> torch GradScaler.step)
__global__ __launch_bounds__(256) void grads_finite_kernel(const mico_adamw_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                           const int64_t* __restrict__ chunk_start, int chunk_elems, float* __restrict__ flag) {
    const mico_adamw_tensor t = tensors[chunk_tensor[blockIdx.x]];
    const int64_t s0 = chunk_start[blockIdx.x], s1 = min(t.numel, s0 + (int64_t)chunk_elems);
    bool bad = false;
    const bool vec = (((uintptr_t)t.g) & 15) == 0 && (s0 & 3) == 0;
    if (vec) {
        int64_t i = s0 + threadIdx.x * 4;
        for (; i + 3 < s1; i += 256 * 4) {
            const f32x4 g = *(const f32x4*)(t.g + i);
            bad |= !(fabsf(g[0]) <= 3.4e38f) | !(fabsf(g[1]) <= 3.4e38f) | !(fabsf(g[2]) <= 3.4e38f) | !(fabsf(g[3]) <= 3.4e38f);
        }
        const int64_t tail0 = s1 - ((s1 - s0) & 3);
        i = tail0 + threadIdx.x;
        if (i < s1) bad |= !(fabsf(t.g[i]) <= 3.4e38f);
    } else {
        for (int64_t i = s0 + threadIdx.x; i < s1; i += 256) bad |= !(fabsf(t.g[i]) <= 3.4e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;
}

extern "C" int mico_grads_finite(const mico_adamw_tensor* tensors, int n_tensors, const int* chunk_tensor, const int64_t* chunk_start,
                                 int nchunks, int chunk_elems, float* flag, void* stream) {
    MICO_CHECK(tensors && chunk_tensor && chunk_start && flag && n_tensors > 0, "mico_grads_finite: null table");
    MICO_CHECK(chunk_elems > 0 && chunk_elems % 4 == 0, "mico_grads_finite: chunk_elems must be a positive multiple of 4");
    if (nchunks <= 0) return MICO_OK;
    MICO_LAUNCH(grads_finite_kernel, dim3(nchunks), dim3(256), 0, ST, tensors, chunk_tensor, chunk_start, chunk_elems, flag);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_adamw_step(const mico_adamw_tensor* tensors, int n_tensors, const int* chunk_tensor, const int64_t* chunk_start,
                               int nchunks, int chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float step_size, float grad_mult, void* stream) {
    MICO_CHECK(tensors && chunk_tensor && chunk_start && n_tensors > 0, "mico_adamw_step: null table");
    MICO_CHECK(chunk_elems > 0 && chunk_elems % 4 == 0, "mico_adamw_step: chunk_elems must be a positive multiple of 4");
    MICO_CHECK(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "mico_adamw_step: bad hyper-parameters");
    if (nchunks <= 0) return MICO_OK;
    MICO_LAUNCH(adamw_kernel, dim3(nchunks), dim3(256), 0, ST, tensors, chunk_tensor, chunk_start, chunk_elems, lr, beta1, beta2, eps,
                weight_decay, step_size, grad_mult);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_colsum(const void* x, int x_dtype, int64_t ld, int64_t rows, int cols, float* out, float scale,
                           int accumulate, void* stream) {
    MICO_CHECK(x && out && cols > 0, "mico_colsum: bad args");
    MICO_CHECK(x_dtype == MICO_F32 || x_dtype == MICO_F16 || x_dtype == MICO_BF16, "mico_colsum: bad dtype");
    if (!accumulate) {
        MICO_LAUNCH(zero_kernel, dim3(egrid(cols)), dim3(EB), 0, ST, out, (int64_t)cols);
        MICO_LAUNCH_CHECK();
    }
    if (rows <= 0) return MICO_OK;
    const int ncb = (cols + EB - 1) / EB;
    int64_t chunks = 2048 / ncb;
    if (chunks < 1) chunks = 1;
    int64_t rpc = (rows + chunks - 1) / chunks;
    if (rpc < 16) rpc = 16;
    chunks = (rows + rpc - 1) / rpc;
    const dim3 grid(ncb, (unsigned)chunks);
    if (x_dtype == MICO_F32) MICO_LAUNCH(colsum_kernel<float>, grid, dim3(EB), 0, ST, (const float*)x, ld, rows, cols, out, scale, rpc);
    else if (x_dtype == MICO_F16) MICO_LAUNCH(colsum_kernel<f16>, grid, dim3(EB), 0, ST, (const f16*)x, ld, rows, cols, out, scale, rpc);
    else MICO_LAUNCH(colsum_kernel<bf16>, grid, dim3(EB), 0, ST, (const bf16*)x, ld, rows, cols, out, scale, rpc);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_cls_rows(float* x, int64_t ld, int B, int group_rows, const float* cls, const float* pos0, int cols,
                             void* stream) {
    MICO_CHECK(x && cls && pos0 && B > 0 && cols > 0, "mico_cls_rows: bad args");
    MICO_LAUNCH(cls_rows_kernel, dim3(egrid((int64_t)B * cols)), dim3(EB), 0, ST, x, ld, B, group_rows, cls, pos0, cols);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_add_f32(const float* a, const float* b, float* y, void* y16, int64_t n, float scale16, int dtype,
                            void* stream) {
    MICO_CHECK(dtype_ok(dtype) && a && (y || y16) && n % 4 == 0, "mico_add_f32: bad args (n must be a multiple of 4)");
    if (n <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(add_f32_kernel<T>, dim3(egrid(n / 4)), dim3(EB), 0, ST, a, b, y, (T*)y16, n / 4, scale16));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_dw_colfold(const float* dwt, const float* dbt, const float* gamma, const float* beta, float* dw, int64_t ld_dw, float* db, int M, int N,
                               void* stream) {
    MICO_CHECK(dwt && dbt && gamma && beta && dw && M > 0 && N > 0 && ld_dw >= N, "mico_dw_colfold: bad args");
    const bool vec = N % 4 == 0 && ld_dw % 4 == 0 && (((uintptr_t)dwt | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)dw) & 15) == 0;
    if (vec) MICO_LAUNCH(dw_colfold_kernel<4>, dim3(egrid((int64_t)M * (N / 4))), dim3(EB), 0, ST, dwt, dbt, gamma, beta, dw, ld_dw, db, M, N / 4);
    else MICO_LAUNCH(dw_colfold_kernel<1>, dim3(egrid((int64_t)M * N)), dim3(EB), 0, ST, dwt, dbt, gamma, beta, dw, ld_dw, db, M, N);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_swiglu_fwd(const void* x1, const void* x2, void* h, int64_t n, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && x1 && x2 && h && n % 8 == 0, "mico_swiglu_fwd: bad args (n must be a multiple of 8)");
    if (n <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(swiglu_fwd_kernel<T>, dim3(egrid(n / 8)), dim3(EB), 0, ST, (const T*)x1, (const T*)x2, (T*)h, n / 8));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_swiglu_fwd_f32(const float* x1, const float* x2, float* h, int64_t n, void* stream) {
    MICO_CHECK(x1 && x2 && h && n % 4 == 0, "mico_swiglu_fwd_f32: bad args (n must be a multiple of 4)");
    if (n <= 0) return MICO_OK;
    MICO_LAUNCH(swiglu_fwd_f32_kernel, dim3(egrid(n / 4)), dim3(EB), 0, ST, x1, x2, h, n / 4);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_swiglu_bwd(const void* x1, const void* x2, const void* dh, void* dx1, void* dx2, int64_t n, int dtype,
                               void* stream) {
    MICO_CHECK(dtype_ok(dtype) && x1 && x2 && dh && dx1 && dx2 && n % 8 == 0, "mico_swiglu_bwd: bad args");
    if (n <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(swiglu_bwd_kernel<T>, dim3(egrid(n / 8)), dim3(EB), 0, ST, (const T*)x1, (const T*)x2, (const T*)dh, (T*)dx1, (T*)dx2, n / 8));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_im2row(const float* pixels, void* rows16, int B, int C, int H, int W, int P, int kpad, int dtype,
                           void* stream) {
    MICO_CHECK(dtype_ok(dtype) && pixels && rows16, "mico_im2row: bad args");
    MICO_CHECK(P > 0 && H % P == 0 && W % P == 0, "mico_im2row: image %dx%d is not a multiple of the patch size %d", H, W, P);
    MICO_CHECK(kpad % 8 == 0 && kpad >= C * P * P, "mico_im2row: kpad must be a multiple of 8 and >= C*P*P");
    if (B <= 0) return MICO_OK;
    const int64_t total = (int64_t)B * (H / P) * (W / P) * (kpad / 4);
    DISPATCH_T16(dtype, MICO_LAUNCH(im2row_kernel<T>, dim3(egrid(total)), dim3(EB), 0, ST, pixels, (T*)rows16, B, C, H, W, P, kpad));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_rope(void* x, int64_t bs, int64_t rs, int B, int N, int H, int hd, const float* cos_t,
                         const float* sin_t, int inverse, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && x && cos_t && sin_t, "mico_rope: bad args");
    MICO_CHECK(hd % 4 == 0 && rs % 4 == 0 && bs % 4 == 0, "mico_rope: hd and strides must be multiples of 4");
    if (B <= 0 || N <= 1) return MICO_OK;
    const int64_t total = (int64_t)B * (N - 1) * H * (hd / 4);
    DISPATCH_T16(dtype, MICO_LAUNCH(rope_kernel<T>, dim3(egrid(total)), dim3(EB), 0, ST, (T*)x, bs, rs, B, N, H, hd, cos_t, sin_t, inverse));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                                   float* sum32, int64_t rows, int S, int cols, int vocab, void* stream) {
    MICO_CHECK(ids && word && pos && type0 && sum32 && cols % 4 == 0 && S > 0, "mico_bert_embed_fwd: bad args");
    if (rows <= 0) return MICO_OK;
    MICO_LAUNCH(bert_embed_kernel, dim3(egrid(rows * (cols / 4))), dim3(EB), 0, ST, ids, word, pos, type0, sum32, rows, S, cols, vocab);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_embed_scatter_add(const int64_t* ids, const float* dsum, float* dword, float* dpos, float* dtype0,
                                      int64_t rows, int S, int cols, int vocab, float scale, void* stream) {
    MICO_CHECK(ids && dsum && S > 0, "mico_embed_scatter_add: bad args");
    if (rows <= 0) return MICO_OK;
    MICO_LAUNCH(embed_scatter_kernel, dim3(egrid(rows * cols)), dim3(EB), 0, ST, ids, dsum, dword, dpos, dtype0, rows, S, cols, vocab, scale);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int cols, void* stream) {
    MICO_CHECK(x && y, "mico_l2norm_fwd: bad args");
    if (rows <= 0) return MICO_OK;
    MICO_LAUNCH(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, x, y, inv_norm, rows, cols);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int cols,
                               void* stream) {
    MICO_CHECK(dy && y && inv_norm && dx, "mico_l2norm_bwd: bad args");
    if (rows <= 0) return MICO_OK;
    MICO_LAUNCH(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, dy, y, inv_norm, dx, rows, cols);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

// ---- small exact-fp32 GEMM (heads, similarity matrices): 16x16 LDS-tiled, one output per thread ---------------------
namespace {
__global__ __launch_bounds__(256) void sgemm_small_kernel(int ta, int tb, int M, int N, int K, const float* __restrict__ A,
                                                          int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ Cm, int64_t ldc, float alpha, float beta,
                                                          const float* __restrict__ bias) {
    __shared__ float as[16][17], bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        {   // A tile [16 m][16 k]
            const int mm = blockIdx.y * 16 + ty, kk = k0 + tx;
            float v = 0.f;
            if (mm < M && kk < K) v = ta ? A[(int64_t)kk * lda + mm] : A[(int64_t)mm * lda + kk];
            as[ty][tx] = v;
            const int nn = blockIdx.x * 16 + ty;
            float w = 0.f;
            if (nn < N && kk < K) w = tb ? B[(int64_t)kk * ldb + nn] : B[(int64_t)nn * ldb + kk];
            bs[ty][tx] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(as[ty][k], bs[tx][k], acc);
        __syncthreads();
    }
    if (m < M && n < N) {
        float v = alpha * acc;
        if (bias) v += bias[n];
        float* c = Cm + (int64_t)m * ldc + n;
        *c = (beta != 0.f) ? v + beta * *c : v;
    }
}
}  // namespace

extern "C" int mico_sgemm_small(int ta, int tb, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                float* Cm, int64_t ldc, float alpha, float beta, const float* bias, void* stream) {
    MICO_CHECK(A && B && Cm && M > 0 && N > 0 && K > 0, "mico_sgemm_small: bad args");
    MICO_LAUNCH(sgemm_small_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, ST, ta, tb, M, N, K, A, lda, B, ldb, Cm, ldc, alpha, beta, bias);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

// ---- GELU helpers and CLS pooling ---------------------------------------------------------------------------------------
namespace {
__global__ void gelu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) y[i] = gelu_f(x[i]);
}
__global__ void gelu_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) dx[i] = dy[i] * gelu_grad_f(x[i]);
}
template <typename T>
__global__ void gelu_16_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EB) {
        float a[8];
        unpack8<T>(((const s16x8*)x)[i], a);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = gelu_f(a[k]);
        ((s16x8*)y)[i] = pack8<T>(a);
    }
}
template <typename T>
__global__ void gelu_bwd_16_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * EB) {
        float a[8], d[8];
        unpack8<T>(((const s16x8*)x)[i], a);
        unpack8<T>(((const s16x8*)dy)[i], d);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] *= gelu_grad_f(a[k]);
        ((s16x8*)dx)[i] = pack8<T>(d);
    }
}
// tokens [b, n, N, D] fp32 -> pooled[b, D] = mean_n tokens[b, n, 0, :]   (model/mico.py:157-164)
__global__ void cls_pool_fwd_kernel(const float* __restrict__ tok, float* __restrict__ out, int b, int n, int64_t frame_stride, int D) {
    const int64_t total = (int64_t)b * D;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t bi = i / D;
        const int c = (int)(i - bi * D);
        float s = 0.f;
        for (int f = 0; f < n; ++f) s += tok[(bi * n + f) * frame_stride + c];
        out[i] = s / (float)n;
    }
}
// dtok[b, n, 0, :] += dpooled[b, :] / n
__global__ void cls_pool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtok, int b, int n, int64_t frame_stride, int D) {
    const int64_t total = (int64_t)b * n * D;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t bf = i / D;
        const int c = (int)(i - bf * D);
        dtok[bf * frame_stride + c] += dout[(bf / n) * D + c] / (float)n;
    }
}
// pool_video (model/mico.py:190-191, 217-218, 233-234): tokens [F, N, D] fp32 -> out [F, 2, D]: row 0 = the frame's CLS token, row 1 = the
// mean of its N - 1 patch tokens (summed in token order, divided once).  One thread per (frame, 4 columns): the N rows of a frame are read with
// 16-byte loads, consecutive threads on consecutive columns.
__global__ void pool_video_fwd_kernel(const float* __restrict__ tok, float* __restrict__ out, int64_t F, int N, int D) {
    const int dv = D >> 2;
    const int64_t total = F * dv;
    const float inv = 1.f / (float)(N - 1);
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t f = i / dv;
        const int c = (int)(i - f * dv) * 4;
        const float* src = tok + f * N * D + c;
        *(f32x4*)(out + (f * 2) * D + c) = *(const f32x4*)src;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = 1; t < N; ++t) s += *(const f32x4*)(src + (int64_t)t * D);
        *(f32x4*)(out + (f * 2 + 1) * D + c) = s * inv;
    }
}
// dtok[f, 0, :] = dout[f, 0, :];  dtok[f, t >= 1, :] = dout[f, 1, :] / (N - 1)   (every element of dtok is written: no zero fill needed)
__global__ void pool_video_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtok, int64_t F, int N, int D) {
    const int dv = D >> 2;
    const int64_t total = F * N * dv;
    const float inv = 1.f / (float)(N - 1);
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int64_t row = i / dv;
        const int c = (int)(i - row * dv) * 4;
        const int64_t f = row / N;
        const int t = (int)(row - f * N);
        const f32x4 g = *(const f32x4*)(dout + (f * 2 + (t ? 1 : 0)) * D + c);
        *(f32x4*)(dtok + row * D + c) = t ? g * inv : g;
    }
}
}  // namespace

extern "C" int mico_gelu_f32(const float* x, float* y, int64_t n, void* stream) {
    MICO_CHECK(x && y, "mico_gelu_f32: bad args");
    if (n <= 0) return MICO_OK;
    MICO_LAUNCH(gelu_f32_kernel, dim3(egrid(n)), dim3(EB), 0, ST, x, y, n);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_gelu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    MICO_CHECK(x && dy && dx, "mico_gelu_bwd_f32: bad args");
    if (n <= 0) return MICO_OK;
    MICO_LAUNCH(gelu_bwd_f32_kernel, dim3(egrid(n)), dim3(EB), 0, ST, x, dy, dx, n);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_gelu_16(const void* x, void* y, int64_t n, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && x && y && n % 8 == 0, "mico_gelu_16: bad args");
    if (n <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(gelu_16_kernel<T>, dim3(egrid(n / 8)), dim3(EB), 0, ST, (const T*)x, (T*)y, n / 8));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_gelu_bwd_16(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && x && dy && dx && n % 8 == 0, "mico_gelu_bwd_16: bad args");
    if (n <= 0) return MICO_OK;
    DISPATCH_T16(dtype, MICO_LAUNCH(gelu_bwd_16_kernel<T>, dim3(egrid(n / 8)), dim3(EB), 0, ST, (const T*)x, (const T*)dy, (T*)dx, n / 8));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_cls_pool_fwd(const float* tokens, float* pooled, int b, int n, int64_t frame_stride, int D, void* stream) {
    MICO_CHECK(tokens && pooled && b > 0 && n > 0, "mico_cls_pool_fwd: bad args");
    MICO_LAUNCH(cls_pool_fwd_kernel, dim3(egrid((int64_t)b * D)), dim3(EB), 0, ST, tokens, pooled, b, n, frame_stride, D);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_cls_pool_bwd(const float* dpooled, float* dtokens, int b, int n, int64_t frame_stride, int D, void* stream) {
    MICO_CHECK(dpooled && dtokens && b > 0 && n > 0, "mico_cls_pool_bwd: bad args");
    MICO_LAUNCH(cls_pool_bwd_kernel, dim3(egrid((int64_t)b * n * D)), dim3(EB), 0, ST, dpooled, dtokens, b, n, frame_stride, D);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_pool_video_fwd(const float* tokens, float* pooled, int64_t frames, int N, int D, void* stream) {
    MICO_CHECK(N >= 2 && D > 0 && D % 4 == 0, "mico_pool_video_fwd: needs N >= 2 tokens per frame and D %% 4 == 0 (got N = %d, D = %d)", N, D);
    if (frames <= 0) return MICO_OK;
    MICO_CHECK(tokens && pooled, "mico_pool_video_fwd: null pointer");
    MICO_LAUNCH(pool_video_fwd_kernel, dim3(egrid(frames * (D / 4))), dim3(EB), 0, ST, tokens, pooled, frames, N, D);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
extern "C" int mico_pool_video_bwd(const float* dpooled, float* dtokens, int64_t frames, int N, int D, void* stream) {
    MICO_CHECK(N >= 2 && D > 0 && D % 4 == 0, "mico_pool_video_bwd: needs N >= 2 tokens per frame and D %% 4 == 0 (got N = %d, D = %d)", N, D);
    if (frames <= 0) return MICO_OK;
    MICO_CHECK(dpooled && dtokens, "mico_pool_video_bwd: null pointer");
    MICO_LAUNCH(pool_video_bwd_kernel, dim3(egrid(frames * N * (D / 4))), dim3(EB), 0, ST, dpooled, dtokens, frames, N, D);
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
