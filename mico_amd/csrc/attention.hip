// Fused attention forward / backward for gfx950 (flash style, fp32 online softmax, MFMA 16x16x32).
// See include/mico_hip.h (mico_attn_fwd / mico_attn_bwd) for the contract.
//
// Layout idea ("everything transposed"): a wave owns 16 query rows and computes S^T = K Q^T, so lane l holds, for
// query row i = l & 15, sixteen scores of the 64-key tile (keys tn*16 + (l>>4)*4 + r).  Row max / row sum are then 15
// in-lane ops + two cross-lane steps (xor 16, 32), and the probabilities a lane holds are *already* the MFMA B operand
// of O^T = V^T P^T once the reduction index is renumbered (slot (g,e) <-> key (2s + (e>>2))*16 + g*4 + (e&3)); V^T
// comes out of LDS with the gfx950 transposing read (ds_read_b64_tr_b16), so P never touches LDS.  The same trick
// gives the backward: dQ^T = K^T dS^T in the per-query-block kernel and dK^T = Q^T dS, dV^T = dO^T P in the
// per-key-block kernel.  hd = 88 (EVA01-g) is zero-padded to 96 in LDS/registers only.
#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;

template <int HDP> struct Cfg {
    // LDS row stride: 256 B = 16 chunk slots (<= 12 used) with the 16-byte chunk index XOR-ed by 2*(row & 7).  With that key
    // both access modes of a tile are bank-conflict free: ds_read_b128 row fragments (16 rows x 2 adjacent chunk columns per
    // lane group) and ds_read_b64_tr_b16 (8 consecutive rows x one chunk pair per half-wave) - the +16-byte padding used
    // before measured 0.4 conflict cycles per LDS-active cycle.
    static constexpr int RS = 256;
    static constexpr int TILE = 64 * RS;           // one [64][HDP] tile
    static constexpr int KS = HDP / 32;            // k-steps over the head dim
    static constexpr int TD = HDP / 16;            // 16-wide tiles over the head dim
    static constexpr int NCH = 64 * (HDP / 8) / 256;   // 16-B chunks per thread per tile
};

// Per-thread, loop-invariant addressing of a [64][HDP] tile (16-byte chunk c = it*256 + tid -> tile row, head-dim chunk):
// global element offset, tile row (1<<20 when the chunk lies beyond the real head dim -> always fetched as zero) and swizzled
// LDS byte offset.  Computed once per kernel: the divisions by HDP/8 = 12 and the swizzle would otherwise be redone for every
// tile (the forward kernel was VALU-bound at ~13 VALU instructions per MFMA).
template <int HDP> struct TileMap {
    int goff[Cfg<HDP>::NCH], row[Cfg<HDP>::NCH], loff[Cfg<HDP>::NCH];
};
template <int HDP>
__device__ __forceinline__ TileMap<HDP> tile_map(int64_t rs, int hd, int tid) {
    TileMap<HDP> m;
#pragma unroll
    for (int it = 0; it < Cfg<HDP>::NCH; ++it) {
        const int c = it * 256 + tid;
        const int row = c / (HDP / 8), ch = c % (HDP / 8);
        m.goff[it] = (int)(row * rs + ch * 8);
        m.row[it] = (ch * 8 < hd) ? row : (1 << 20);
        m.loff[it] = row * Cfg<HDP>::RS + ((ch ^ ((row & 7) << 1)) << 4);
    }
    return m;
}
// global -> registers for a [64][HDP] tile (zero beyond nrows / hd)
template <typename T, int HDP>
__device__ __forceinline__ void tile_fetch(s16x8 (&reg)[Cfg<HDP>::NCH], const T* base, int64_t rs, int row0, int nrows,
                                           const TileMap<HDP>& m) {
    const T* b0 = base + (int64_t)row0 * rs;
#pragma unroll
    for (int it = 0; it < Cfg<HDP>::NCH; ++it) {
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row0 + m.row[it] < nrows) v = *(const s16x8*)(b0 + m.goff[it]);
        reg[it] = v;
    }
}
template <int HDP>
__device__ __forceinline__ void tile_commit(const s16x8 (&reg)[Cfg<HDP>::NCH], LDS_AS char* tile, const TileMap<HDP>& m) {
#pragma unroll
    for (int it = 0; it < Cfg<HDP>::NCH; ++it) *(LDS_AS s16x8*)(tile + m.loff[it]) = reg[it];
}

// row-operand fragment straight from global: X[row][ks*32 + (lane>>4)*8 .. +8]
template <typename T, int HDP>
__device__ __forceinline__ void row_frags(s16x8 (&f)[Cfg<HDP>::KS], const T* base, int64_t rs, int row, int nrows, int hd,
                                          int lane) {
#pragma unroll
    for (int ks = 0; ks < Cfg<HDP>::KS; ++ks) {
        const int d = ks * 32 + (lane >> 4) * 8;
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < nrows && d < hd) v = *(const s16x8*)(base + (int64_t)row * rs + d);
        f[ks] = v;
    }
}

// A-operand fragment of a row-major LDS tile: rows r0 + (lane&15), head-dim chunk ks
template <int HDP>
__device__ __forceinline__ s16x8 lds_row_frag(LDS_AS const char* tile, int r0, int ks, int lane) {
    const int row = r0 + (lane & 15), ch = ks * 4 + (lane >> 4);
    return *(LDS_AS const s16x8*)(tile + row * Cfg<HDP>::RS + ((ch ^ ((row & 7) << 1)) << 4));
}
// A-operand fragment of the TRANSPOSED tile: output rows d = td*16 + (lane&15), reduction slots over tile rows
// (2*s2 + r2)*16 + (lane>>4)*4 + {0..3}
template <int HDP>
__device__ __forceinline__ s16x8 lds_tr_frag(LDS_AS const char* tile, int td, int s2, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int ch = td * 2 + ((p >> 1) & 1), half = (p & 1) * 8;
    const int r_lo = (2 * s2) * 16 + g * 4 + (p >> 2);   // rows r_lo and r_lo + 16 share (row & 7), hence the swizzle key
    const int col_b = ((ch ^ ((r_lo & 7) << 1)) << 4) + half;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + r_lo * Cfg<HDP>::RS + col_b));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + (r_lo + 16) * Cfg<HDP>::RS + col_b));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <typename T>
__device__ __forceinline__ s16x8 pack_pair(const f32x4& a, const f32x4& b) {
    typename T16<T>::v8 v;
    v[0] = (T)a[0]; v[1] = (T)a[1]; v[2] = (T)a[2]; v[3] = (T)a[3];
    v[4] = (T)b[0]; v[5] = (T)b[1]; v[6] = (T)b[2]; v[7] = (T)b[3];
    return __builtin_bit_cast(s16x8, v);
}

__device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups that share lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ float mask_val(const float* mask, int mode, int b, int i, int j, int Sq, int Sk) {
    if (mode == 1) return mask[(int64_t)b * Sk + j];
    if (mode == 2) return mask[((int64_t)b * Sq + i) * Sk + j];
    return 0.f;
}

// ======================================================================================================================
// forward
// ======================================================================================================================
template <typename T, int HDP, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, T* __restrict__ o,
                                                          float* __restrict__ lse, const mico_attn_params p) {
    using C = Cfg<HDP>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE];
    LDS_AS char* kt = (LDS_AS char*)smem;
    LDS_AS char* vt = kt + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
    const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
    const int i = q0 + wave * 16 + (lane & 15);   // this lane's query row
    const int g = lane >> 4;
    const bool wave_live = q0 + wave * 16 < p.Sq;  // wave-uniform: this wave owns at least one real query row

    s16x8 qf[C::KS];
    row_frags<T, HDP>(qf, qb, p.q_rs, i, p.Sq, p.hd, lane);

    f32x4 oacc[C::TD];
#pragma unroll
    for (int t = 0; t < C::TD; ++t) oacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_BIG, l_run = 0.f;   // running max (log2 domain) and sum
    const float sc2 = p.scale * 1.4426950408889634f;

    const int nt = (p.Sk + 63) / 64;
    const TileMap<HDP> tm_a = tile_map<HDP>(p.k_rs, p.hd, tid), tm_b = tile_map<HDP>(p.v_rs, p.hd, tid);
    s16x8 kr[C::NCH], vr[C::NCH];
    tile_fetch<T, HDP>(kr, kb, p.k_rs, 0, p.Sk, tm_a);
    tile_fetch<T, HDP>(vr, vb, p.v_rs, 0, p.Sk, tm_b);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        tile_commit<HDP>(kr, kt, tm_a);
        tile_commit<HDP>(vr, vt, tm_b);
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<T, HDP>(kr, kb, p.k_rs, (t + 1) * 64, p.Sk, tm_a);
            tile_fetch<T, HDP>(vr, vb, p.v_rs, (t + 1) * 64, p.Sk, tm_b);
        }
        // S^T = K Q^T.  Only the 16-key sub-tiles that hold real keys are computed: N = 257 (ViT-g/14) ends in a tile with one
        // valid sub-tile, and a full fifth tile cost 15 % of the kernel (tools/attn_bench.py, 257 vs 256 tokens).
        const bool edge = (t == nt - 1) && (p.Sk & 63);   // workgroup-uniform: only the ragged last key tile needs bounds
        const int ntn = edge ? ((p.Sk - t * 64 + 15) >> 4) : 4;
        f32x4 s[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            s[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < ntn && wave_live) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) s[tn] = T16<T>::mfma(lds_row_frag<HDP>(kt, tn * 16, ks, lane), qf[ks], s[tn]);
            }
        }
        // online softmax in the exp2 domain (v_exp_f32 is 2^x): scores are pre-multiplied by scale * log2(e)
        float mloc = NEG_BIG;
        if (!p.mask_mode && !edge) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                s[tn] *= sc2;
                mloc = fmaxf(fmaxf(mloc, fmaxf(s[tn][0], s[tn][1])), fmaxf(s[tn][2], s[tn][3]));
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = t * 64 + tn * 16 + g * 4 + r;
                    float x = s[tn][r] * sc2;
                    if (j < p.Sk) {
                        if (p.mask_mode && i < p.Sq) x += mask_val(p.mask, p.mask_mode, b, i, j, p.Sq, p.Sk) * 1.4426950408889634f;
                    } else {
                        x = NEG_BIG;
                    }
                    s[tn][r] = x;
                    mloc = fmaxf(mloc, x);
                }
        }
        const float m_new = fmaxf(m_run, group_max(mloc));
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lloc = 0.f;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[tn][r] - m_new);
                s[tn][r] = e;
                lloc += e;
            }
        if (DROP) {   // dropout on the probabilities (compile-time variant: the ViT towers never pay its registers): the row sum above stays the undropped one
            const unsigned thr = drop_threshold(p.drop_p);
            const float ik = 1.f / (1.f - p.drop_p);
            const unsigned long long rowbase = (((unsigned long long)b * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[tn][r] *= drop_mult(p.drop_seed, p.drop_site, rowbase + (t * 64 + tn * 16 + g * 4 + r), thr, ik);
        }
        l_run = l_run * alpha + group_sum(lloc);
        m_run = m_new;
#pragma unroll
        for (int t2 = 0; t2 < C::TD; ++t2) oacc[t2] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (2 * s2 >= ntn || !wave_live) continue;    // no real key in this 32-key half / no real query in this wave
            const s16x8 pf = pack_pair<T>(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int td = 0; td < C::TD; ++td) oacc[td] = T16<T>::mfma(lds_tr_frag<HDP>(vt, td, s2, lane), pf, oacc[td]);
        }
    }
    if (i < p.Sq) {
        const float inv = 1.f / l_run;
        T* ob = o + (int64_t)b * p.o_bs + (int64_t)i * p.o_rs + h * p.hd;
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            const int d = td * 16 + g * 4;
            if (d < p.hd) *(s16x4*)(ob + d) = pack4<T>(oacc[td][0] * inv, oacc[td][1] * inv, oacc[td][2] * inv, oacc[td][3] * inv);
        }
        if (g == 0) lse[((int64_t)b * p.H + h) * p.Sq + i] = (m_run + __log2f(l_run)) * 0.6931471805599453f;
    }
}

// ======================================================================================================================
// backward, kernel 1: dQ (and delta = rowsum(dO * O)) - one workgroup per 64-query block, loops over key tiles
// ======================================================================================================================
template <typename T, int HDP, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ v, const T* __restrict__ o,
                                                             const T* __restrict__ d_o, const float* __restrict__ lse,
                                                             T* __restrict__ dq, float* __restrict__ delta,
                                                             const mico_attn_params p) {
    using C = Cfg<HDP>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE];
    LDS_AS char* kt = (LDS_AS char*)smem;
    LDS_AS char* vt = kt + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
    const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
    const T* ob = o + (int64_t)b * p.o_bs + h * p.hd;
    const T* dob = d_o + (int64_t)b * p.o_bs + h * p.hd;
    const int i = q0 + wave * 16 + (lane & 15);
    const int g = lane >> 4;
    const bool wave_live = q0 + wave * 16 < p.Sq;

    s16x8 qf[C::KS], dof[C::KS];
    row_frags<T, HDP>(qf, qb, p.q_rs, i, p.Sq, p.hd, lane);
    row_frags<T, HDP>(dof, dob, p.o_rs, i, p.Sq, p.hd, lane);
    float dl = 0.f;
    {
        s16x8 of[C::KS];
        row_frags<T, HDP>(of, ob, p.o_rs, i, p.Sq, p.hd, lane);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            float a[8], c[8];
            unpack8<T>(of[ks], a);
            unpack8<T>(dof[ks], c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += a[e] * c[e];
        }
        dl = group_sum(dl);
    }
    const int64_t stat_idx = ((int64_t)b * p.H + h) * p.Sq + i;
    float lse_i = 0.f;
    if (i < p.Sq) {
        lse_i = lse[stat_idx];
        if (g == 0) delta[stat_idx] = dl;
    }

    f32x4 dqacc[C::TD];
#pragma unroll
    for (int t = 0; t < C::TD; ++t) dqacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (p.Sk + 63) / 64;
    const TileMap<HDP> tm_a = tile_map<HDP>(p.k_rs, p.hd, tid), tm_b = tile_map<HDP>(p.v_rs, p.hd, tid);
    s16x8 kr[C::NCH], vr[C::NCH];
    tile_fetch<T, HDP>(kr, kb, p.k_rs, 0, p.Sk, tm_a);
    tile_fetch<T, HDP>(vr, vb, p.v_rs, 0, p.Sk, tm_b);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        tile_commit<HDP>(kr, kt, tm_a);
        tile_commit<HDP>(vr, vt, tm_b);
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<T, HDP>(kr, kb, p.k_rs, (t + 1) * 64, p.Sk, tm_a);
            tile_fetch<T, HDP>(vr, vb, p.v_rs, (t + 1) * 64, p.Sk, tm_b);
        }
        const int ntn = (t == nt - 1 && (p.Sk & 63)) ? ((p.Sk - t * 64 + 15) >> 4) : 4;   // 16-key sub-tiles holding real keys
        f32x4 s[4], dp[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            s[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < ntn && wave_live) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    s[tn] = T16<T>::mfma(lds_row_frag<HDP>(kt, tn * 16, ks, lane), qf[ks], s[tn]);
                    dp[tn] = T16<T>::mfma(lds_row_frag<HDP>(vt, tn * 16, ks, lane), dof[ks], dp[tn]);
                }
            }
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 64 + tn * 16 + g * 4 + r;
                float ds = 0.f;
                if (j < p.Sk && i < p.Sq) {
                    float x = s[tn][r] * p.scale;
                    if (p.mask_mode) x += mask_val(p.mask, p.mask_mode, b, i, j, p.Sq, p.Sk);
                    const float pr = __expf(x - lse_i);
                    float dpe = dp[tn][r];
                    if (DROP)
                        dpe *= drop_mult(p.drop_seed, p.drop_site, (((unsigned long long)b * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk + j,
                                         drop_threshold(p.drop_p), 1.f / (1.f - p.drop_p));
                    ds = pr * (dpe - dl) * p.scale;
                }
                s[tn][r] = ds;
            }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (2 * s2 >= ntn || !wave_live) continue;
            const s16x8 df = pack_pair<T>(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int td = 0; td < C::TD; ++td) dqacc[td] = T16<T>::mfma(lds_tr_frag<HDP>(kt, td, s2, lane), df, dqacc[td]);
        }
    }
    if (i < p.Sq) {
        T* dqb = dq + (int64_t)b * p.q_bs + (int64_t)i * p.q_rs + h * p.hd;
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            const int d = td * 16 + g * 4;
            if (d < p.hd) *(s16x4*)(dqb + d) = pack4<T>(dqacc[td][0], dqacc[td][1], dqacc[td][2], dqacc[td][3]);
        }
    }
}

// ======================================================================================================================
// backward, kernel 2: dK, dV - one workgroup per 64-key block, loops over query tiles
// ======================================================================================================================
template <typename T, int HDP, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, const T* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              T* __restrict__ dk, T* __restrict__ dv,
                                                              const mico_attn_params p) {
    using C = Cfg<HDP>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE + 512];
    LDS_AS char* qt = (LDS_AS char*)smem;
    LDS_AS char* dot = qt + C::TILE;
    LDS_AS float* lse_t = (LDS_AS float*)(dot + C::TILE);   // [64] lse, [64] delta
    LDS_AS float* del_t = lse_t + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
    const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
    const T* dob = d_o + (int64_t)b * p.o_bs + h * p.hd;
    const int j = k0 + wave * 16 + (lane & 15);   // this lane's key row
    const int g = lane >> 4;
    const bool wave_live = k0 + wave * 16 < p.Sk;   // wave-uniform: this wave owns at least one real key
    const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;

    s16x8 kf[C::KS], vf[C::KS];
    row_frags<T, HDP>(kf, kb, p.k_rs, j, p.Sk, p.hd, lane);
    row_frags<T, HDP>(vf, vb, p.v_rs, j, p.Sk, p.hd, lane);

    f32x4 dkacc[C::TD], dvacc[C::TD];
#pragma unroll
    for (int t = 0; t < C::TD; ++t) {
        dkacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dvacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int nt = (p.Sq + 63) / 64;
    const TileMap<HDP> tm_a = tile_map<HDP>(p.q_rs, p.hd, tid), tm_b = tile_map<HDP>(p.o_rs, p.hd, tid);
    s16x8 qr[C::NCH], dor[C::NCH];
    float st_l = 0.f, st_d = 0.f;
    tile_fetch<T, HDP>(qr, qb, p.q_rs, 0, p.Sq, tm_a);
    tile_fetch<T, HDP>(dor, dob, p.o_rs, 0, p.Sq, tm_b);
    if (tid < 64 && tid < p.Sq) { st_l = lse[stat_base + tid]; st_d = delta[stat_base + tid]; }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        tile_commit<HDP>(qr, qt, tm_a);
        tile_commit<HDP>(dor, dot, tm_b);
        if (tid < 64) { lse_t[tid] = st_l; del_t[tid] = st_d; }
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<T, HDP>(qr, qb, p.q_rs, (t + 1) * 64, p.Sq, tm_a);
            tile_fetch<T, HDP>(dor, dob, p.o_rs, (t + 1) * 64, p.Sq, tm_b);
            st_l = 0.f; st_d = 0.f;
            const int ii = (t + 1) * 64 + tid;
            if (tid < 64 && ii < p.Sq) { st_l = lse[stat_base + ii]; st_d = delta[stat_base + ii]; }
        }
        // S = Q K^T, dP = dO V^T  (lane: key j = lane&15, query rows ti*16 + g*4 + r); only 16-query sub-tiles with real queries
        const int nti = (t == nt - 1 && (p.Sq & 63)) ? ((p.Sq - t * 64 + 15) >> 4) : 4;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            s[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ti < nti && wave_live) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    s[ti] = T16<T>::mfma(lds_row_frag<HDP>(qt, ti * 16, ks, lane), kf[ks], s[ti]);
                    dp[ti] = T16<T>::mfma(lds_row_frag<HDP>(dot, ti * 16, ks, lane), vf[ks], dp[ti]);
                }
            }
        }
        f32x4 pr[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + ti * 16 + g * 4);
            const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + ti * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = t * 64 + ti * 16 + g * 4 + r;
                float pv = 0.f, ds = 0.f;
                if (i < p.Sq && j < p.Sk) {
                    float x = s[ti][r] * p.scale;
                    if (p.mask_mode) x += mask_val(p.mask, p.mask_mode, b, i, j, p.Sq, p.Sk);
                    pv = __expf(x - lv[r]);
                    float dm = 1.f;
                    if (DROP)
                        dm = drop_mult(p.drop_seed, p.drop_site, (((unsigned long long)b * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk + j,
                                       drop_threshold(p.drop_p), 1.f / (1.f - p.drop_p));
                    ds = pv * (dp[ti][r] * dm - dv4[r]) * p.scale;
                    pv *= dm;      // dV sees the dropped probabilities
                }
                pr[ti][r] = pv;
                s[ti][r] = ds;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (2 * s2 >= nti || !wave_live) continue;
            const s16x8 pf = pack_pair<T>(pr[2 * s2], pr[2 * s2 + 1]);
            const s16x8 df = pack_pair<T>(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                dvacc[td] = T16<T>::mfma(lds_tr_frag<HDP>(dot, td, s2, lane), pf, dvacc[td]);
                dkacc[td] = T16<T>::mfma(lds_tr_frag<HDP>(qt, td, s2, lane), df, dkacc[td]);
            }
        }
    }
    if (j < p.Sk) {
        T* dkb = dk + (int64_t)b * p.k_bs + (int64_t)j * p.k_rs + h * p.hd;
        T* dvb = dv + (int64_t)b * p.v_bs + (int64_t)j * p.v_rs + h * p.hd;
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            const int d = td * 16 + g * 4;
            if (d < p.hd) {
                *(s16x4*)(dkb + d) = pack4<T>(dkacc[td][0], dkacc[td][1], dkacc[td][2], dkacc[td][3]);
                *(s16x4*)(dvb + d) = pack4<T>(dvacc[td][0], dvacc[td][1], dvacc[td][2], dvacc[td][3]);
            }
        }
    }
}

int check_params(const mico_attn_params* p, const char* who) {
    MICO_CHECK(p, "%s: null params", who);
    MICO_CHECK(p->B > 0 && p->H > 0 && p->Sq > 0 && p->Sk > 0, "%s: empty problem", who);
    MICO_CHECK(p->hd % 8 == 0 && p->hd > 0 && p->hd <= 128, "%s: head dim must be a multiple of 8 and <= 128 (got %d)", who, p->hd);
    MICO_CHECK(p->q_rs % 8 == 0 && p->k_rs % 8 == 0 && p->v_rs % 8 == 0 && p->o_rs % 8 == 0 && p->q_bs % 8 == 0 &&
                   p->k_bs % 8 == 0 && p->v_bs % 8 == 0 && p->o_bs % 8 == 0,
               "%s: strides must be multiples of 8 elements", who);
    MICO_CHECK(p->mask_mode >= 0 && p->mask_mode <= 2 && (p->mask_mode == 0 || p->mask), "%s: bad mask", who);
    MICO_CHECK(p->drop_p >= 0.f && p->drop_p < 1.f, "%s: drop_p must be in [0, 1)", who);
    return MICO_OK;
}

}  // namespace

#define ATTN_DISPATCH_HD(hd, ...)                        \
    do {                                                 \
        if ((hd) <= 64) { constexpr int HDP = 64; __VA_ARGS__; } \
        else if ((hd) <= 96) { constexpr int HDP = 96; __VA_ARGS__; } \
        else { constexpr int HDP = 128; __VA_ARGS__; }   \
    } while (0)

extern "C" int mico_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const mico_attn_params* p,
                             int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && q && k && v && o && lse, "mico_attn_fwd: bad args");
    int rc = check_params(p, "mico_attn_fwd");
    if (rc) return rc;
    const dim3 grid((p->Sq + 63) / 64, p->H, p->B), block(256);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, { if (p->drop_p > 0.f) MICO_LAUNCH((attn_fwd_kernel<T, HDP, true>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p); else MICO_LAUNCH((attn_fwd_kernel<T, HDP, false>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p); }));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}

extern "C" int mico_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                             void* dq, void* dk, void* dv, float* delta, const mico_attn_params* p, int dtype,
                             void* stream) {
    MICO_CHECK(dtype_ok(dtype) && q && k && v && o && d_o && lse && dq && dk && dv && delta, "mico_attn_bwd: bad args");
    int rc = check_params(p, "mico_attn_bwd");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(256);
    const dim3 gq((p->Sq + 63) / 64, p->H, p->B), gk((p->Sk + 63) / 64, p->H, p->B);
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, { if (p->drop_p > 0.f) MICO_LAUNCH((attn_bwd_dq_kernel<T, HDP, true>), gq, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p); else MICO_LAUNCH((attn_bwd_dq_kernel<T, HDP, false>), gq, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p); }));
    MICO_LAUNCH_CHECK();
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, { if (p->drop_p > 0.f) MICO_LAUNCH((attn_bwd_dkv_kernel<T, HDP, true>), gk, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, *p); else MICO_LAUNCH((attn_bwd_dkv_kernel<T, HDP, false>), gk, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, *p); }));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
