// Shared device helpers for the MiCo gfx950 kernels (CDNA4 only: wave64, MFMA 16x16x32, LDS-DMA, tr-reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mico_hip.h"

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LDS_AS __attribute__((address_space(3)))

extern thread_local char g_mico_err[256];
int mico_set_err(int code, const char* fmt, ...);

#define MICO_CHECK(cond, ...)                         \
    do {                                              \
        if (!(cond)) return mico_set_err(MICO_EINVAL, __VA_ARGS__); \
    } while (0)

// hipGetLastError() is sticky per host thread: clear whatever an unrelated earlier runtime call left behind before the
// launch so MICO_LAUNCH_CHECK reports this launch only.
#define MICO_LAUNCH(...)               \
    do {                               \
        (void)hipGetLastError();       \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define MICO_LAUNCH_CHECK()                                                       \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) return mico_set_err(MICO_ELAUNCH, "%s: %s", __func__, hipGetErrorString(e__)); \
    } while (0)

// ---- 16-bit element traits -------------------------------------------------------------------------------------
template <typename T> struct T16;
template <> struct T16<f16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
// In-place MFMA with the accumulator pinned to the AGPR half of the register file (one-wave-per-SIMD kernels keep 256 accumulator
// registers there).  Inline assembly because hipcc 7.2 does not keep 64 loop-carried accumulator tiles in the tied (dst == srcC) form:
// it emits the three-address form and rotates the tiles through v_accvgpr copies (~300 extra instructions per 128 MFMAs).  Rules for
// the caller: the compiler does not know this is an MFMA - it still waits for the ds_read that produced a / b (they are plain operands),
// back-to-back accumulation into the same tile needs no wait states, but the FIRST read of c by other code must be preceded by
// mfma_acc_fence() (the XDL write -> VALU read wait states).
template <typename T> __device__ __forceinline__ void mfma_acc(s16x8 a, s16x8 b, f32x4& c);
template <> __device__ __forceinline__ void mfma_acc<f16>(s16x8 a, s16x8 b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_acc<bf16>(s16x8 a, s16x8 b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_fence() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

template <> struct T16<bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 32x32x16: lane l supplies row l & 31 of each operand, k = 8 (l >> 5) .. + 7; D[i][j]: j = l & 31, i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
template <typename T> __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<f16>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<bf16>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

template <typename T> __device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
    typename T16<T>::v4 v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    return __builtin_bit_cast(s16x4, v);
}
// (T)(a s) | (T)(b s) << 16.  fp16: v_fma_mix{lo,hi}_f16 forms the fp32 product exactly and rounds it to fp16 ONCE (a v_mul_f32 +
// v_cvt_pk_f16_f32 pair rounds twice); written out because the compiler picks one form or the other per element as it schedules, which
// makes the bits of a kernel's output depend on unrelated edits around the conversion.  bf16: one multiply, one conversion.
template <typename T> __device__ __forceinline__ uint32_t pack2_scaled(float a, float b, float s);
template <> __device__ __forceinline__ uint32_t pack2_scaled<f16>(float a, float b, float s) {
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r) : "v"(b), "v"(s));
    return r;
}
template <> __device__ __forceinline__ uint32_t pack2_scaled<bf16>(float a, float b, float s) {
    typedef bf16 v2 __attribute__((ext_vector_type(2)));
    v2 v;
    v[0] = (bf16)(a * s); v[1] = (bf16)(b * s);
    return __builtin_bit_cast(uint32_t, v);
}
template <typename T> __device__ __forceinline__ f32x4 unpack4(s16x4 s) {
    typename T16<T>::v4 v = __builtin_bit_cast(typename T16<T>::v4, s);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <typename T> __device__ __forceinline__ void unpack8(s16x8 s, float* o) {
    typename T16<T>::v8 v = __builtin_bit_cast(typename T16<T>::v8, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ s16x8 pack8(const float* o) {
    typename T16<T>::v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)o[i];
    return __builtin_bit_cast(s16x8, v);
}

// exact (erf) GELU and its derivative - matches nn.GELU / mico.py:22-28
// erfc(|z|) * via Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7): y = poly(t) * exp(-z^2), t = 1 / (1 + p |z|).  libm's erff costs
// ~3x more VALU and sits un-overlapped in the GEMM epilogues (measured +40% on the fc1 GEMM).  Returning 1 + erf(z) as 2 - y
// (z >= 0) or y (z < 0) avoids the cancellation of 1 + erf for negative arguments; e2 = exp(-z^2) is shared with the Gaussian
// density of the GELU derivative.
__device__ __forceinline__ float one_plus_erf(float z, float& e2) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    e2 = __expf(-az * az);
    const float y = poly * t * e2;
    return z >= 0.f ? 2.0f - y : y;
}
__device__ __forceinline__ float gelu_f(float x) {
    float e2;
    return 0.5f * x * one_plus_erf(x * 0.70710678118654752f, e2);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float e2;   // = exp(-x^2 / 2)
    const float cdf = 0.5f * one_plus_erf(x * 0.70710678118654752f, e2);
    return fmaf(x * 0.39894228040143268f, e2, cdf);
}

// The same arithmetic, operation for operation, on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two elements per
// instruction; the reciprocal, the exponential and the sign select stay per element): the GEMM epilogues' GELU and GELU / GELU' pair over
// the four accumulator values a lane holds.  One definition for every kernel, so a value computed by the pair epilogue and by the GELU-only
// epilogue is the same bit pattern (the activation diet recomputes what the forward produced).
__device__ __forceinline__ void gelu_terms2(const f32x2 x, f32x2& cdf, f32x2& e2) {
    const f32x2 z = x * 0.70710678118654752f;
    const f32x2 az = {fabsf(z[0]), fabsf(z[1])};
    const f32x2 den = __builtin_elementwise_fma(az, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 poly = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 w = (az * az) * -1.4426950408889634f;
    e2 = (f32x2){__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    const f32x2 y = poly * t * e2;
    const f32x2 y2 = (f32x2){2.0f, 2.0f} - y;
    const f32x2 ope = {z[0] >= 0.f ? y2[0] : y[0], z[1] >= 0.f ? y2[1] : y[1]};   // 1 + erf(z)
    cdf = ope * 0.5f;
}
__device__ __forceinline__ f32x4 gelu4(const f32x4 x) {
    f32x2 c0, c1, e0, e1;
    gelu_terms2((f32x2){x[0], x[1]}, c0, e0);
    gelu_terms2((f32x2){x[2], x[3]}, c1, e1);
    const f32x2 g0 = (f32x2){x[0], x[1]} * c0, g1 = (f32x2){x[2], x[3]} * c1;
    return (f32x4){g0[0], g0[1], g1[0], g1[1]};
}
// x -> gelu(x) (returned), deriv = gelu'(x) = cdf + x pdf
__device__ __forceinline__ f32x4 gelu_pair4(const f32x4 x, f32x4& deriv) {
    f32x2 c0, c1, e0, e1;
    const f32x2 x0 = {x[0], x[1]}, x1 = {x[2], x[3]};
    gelu_terms2(x0, c0, e0);
    gelu_terms2(x1, c1, e1);
    const f32x2 d0 = __builtin_elementwise_fma(x0 * 0.39894228040143268f, e0, c0), d1 = __builtin_elementwise_fma(x1 * 0.39894228040143268f, e1, c1);
    const f32x2 g0 = x0 * c0, g1 = x1 * c1;
    deriv = (f32x4){d0[0], d0[1], d1[0], d1[1]};
    return (f32x4){g0[0], g0[1], g1[0], g1[1]};
}

// wave64 reductions
// ---- stateless dropout decision (documented at mico_dropout in include/mico_hip.h; oracle/mico_oracle.py restates it) ----
__device__ __forceinline__ unsigned drop_hash(unsigned seed, int site, unsigned long long idx) {
    unsigned h = seed ^ ((unsigned)site * 0x9E3779B9u);
    h ^= (unsigned)idx * 0x85EBCA6Bu;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    h ^= (unsigned)(idx >> 32) * 0xC2B2AE35u;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h >> 8;
}
// drop_hash() of an index below 2^32 (its high-word term vanishes) whose low-word product t = (unsigned)idx * 0x85EBCA6B the caller
// carries incrementally (idx + n -> t + n * 0x85EBCA6B): the attention kernels' per-score decision without the 64-bit index
// arithmetic and one of the three 32-bit multiplies.  h0 = seed ^ site * 0x9E3779B9.
__device__ __forceinline__ unsigned drop_hash_t(unsigned h0, unsigned t) {
    unsigned h = h0 ^ t;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h >> 8;
}
// multiplier of element idx: 0 or 1/(1-p).  thr = (unsigned)(p * 2^24), inv_keep = 1/(1-p)
__device__ __forceinline__ float drop_mult(unsigned seed, int site, unsigned long long idx, unsigned thr, float inv_keep) {
    return drop_hash(seed, site, idx) >= thr ? inv_keep : 0.f;
}
__host__ __device__ __forceinline__ unsigned drop_threshold(float p) { return (unsigned)(p * 16777216.f); }

// Cross-lane steps without the LDS crossbar (ds_bpermute costs ~100 cycles of latency per step on the critical path of every
// row reduction): DPP adds inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows - called with the
// same value in both registers a swap leaves {own half, partner half}, whose combination is the xor-16 / xor-32 exchange.
// (inline asm: hipcc 7.2 lowers the second result of __builtin_amdgcn_permlane{16,32}_swap to the first one's register; the
// s_nop covers the VALU-write -> swap-read hazard the compiler would otherwise pad)
#define MICO_XLANE(NAME, INSN, OP)                                                  \
    __device__ __forceinline__ float NAME(float v) {                                 \
        float a = v, b = v;                                                          \
        asm volatile("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));               \
        return OP;                                                                   \
    }
MICO_XLANE(xor16_sum, "v_permlane16_swap_b32", a + b)
MICO_XLANE(xor32_sum, "v_permlane32_swap_b32", a + b)
MICO_XLANE(xor16_max, "v_permlane16_swap_b32", fmaxf(a, b))
MICO_XLANE(xor32_max, "v_permlane32_swap_b32", fmaxf(a, b))
#undef MICO_XLANE
#define MICO_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float row16_sum(float v) {   // every lane ends with the sum over its row of 16 lanes
    v += MICO_DPP_F32(v, 0xb1);    // quad_perm [1,0,3,2]
    v += MICO_DPP_F32(v, 0x4e);    // quad_perm [2,3,0,1]
    v += MICO_DPP_F32(v, 0x124);   // row_ror:4
    v += MICO_DPP_F32(v, 0x128);   // row_ror:8
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, MICO_DPP_F32(v, 0xb1));
    v = fmaxf(v, MICO_DPP_F32(v, 0x4e));
    v = fmaxf(v, MICO_DPP_F32(v, 0x124));
    v = fmaxf(v, MICO_DPP_F32(v, 0x128));
    return v;
}
#undef MICO_DPP_F32
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// Transposed LDS read (ds_read_b64_tr_b16): within each 16-lane group, lane p supplies the address of 4
// contiguous 16-bit elements [row p>>2][cols (p&3)*4..+3] of a 4x16 block; lane i receives column i (4 rows).
__device__ __forceinline__ s16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(uintptr_t)p);
}

template <typename F> __global__ void generic_kernel(F f) { f(); }

static inline int dtype_ok(int dt) { return dt == MICO_F16 || dt == MICO_BF16; }

#define DISPATCH_T16(dt, ...)                 \
    do {                                      \
        if ((dt) == MICO_F16) { typedef f16 T; __VA_ARGS__; } \
        else { typedef bf16 T; __VA_ARGS__; } \
    } while (0)
