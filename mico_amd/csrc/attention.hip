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
#include <type_traits>

namespace {

constexpr float NEG_BIG = -1.0e30f;

#ifndef MICO_ATTN_PRIO
#define MICO_ATTN_PRIO 0   // experiment: alternating s_setprio between the two waves of a SIMD in the persistent tower kernels
#endif

#ifdef MICO_ATTN_PHASES   // timing build (tools/probes/attn_phases.py): per-wave cycle counts of the forward kernel's phases
__device__ unsigned long long g_attn_phase[4096 * 8];
#define PH_DECL unsigned long long ph_t = __builtin_readcyclecounter(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PH(slot) do { const unsigned long long n_ = __builtin_readcyclecounter(); ph_acc[slot] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_STORE do { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
        if (lane == 0 && wave == 0 && wg_ < 4096) for (int e_ = 0; e_ < 8; ++e_) g_attn_phase[wg_ * 8 + e_] = ph_acc[e_]; } while (0)
#else
#define PH_DECL
#define PH(slot)
#define PH_STORE
#endif

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

// chunk-swizzle key of a row of the 256-byte-row tile images.  KM 0: 2 (row & 7) - the tiled kernels and the 16x16-MFMA resident kernels
// (conflict free for their ds_read_b128 / transposing access patterns, see Cfg).  KM 1: the bit pairs of row & 15 swapped - the 32x32-MFMA
// kernels: a ds_read_b128 lane group then holds 16 rows with ONE chunk index (16 distinct keys), and the four consecutive rows of a
// transposing read land in four different 64-byte windows.
template <int KM> __device__ __forceinline__ int swz_key(int row) {
    return KM == 0 ? ((row & 7) << 1) : (((row & 3) << 2) | ((row >> 2) & 3));
}
// A-operand fragment of a row-major LDS tile: rows r0 + (lane&15), head-dim chunk ks
template <int HDP, int KM = 0>
__device__ __forceinline__ s16x8 lds_row_frag(LDS_AS const char* tile, int r0, int ks, int lane) {
    const int row = r0 + (lane & 15), ch = ks * 4 + (lane >> 4);
    return *(LDS_AS const s16x8*)(tile + row * Cfg<HDP>::RS + ((ch ^ swz_key<KM>(row)) << 4));
}
// A-operand fragment of the TRANSPOSED tile: output rows d = td*16 + (lane&15), reduction slots over tile rows
// (2*s2 + r2)*16 + (lane>>4)*4 + {0..3}
template <int HDP, int KM = 0>
__device__ __forceinline__ s16x8 lds_tr_frag(LDS_AS const char* tile, int td, int s2, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int ch = td * 2 + ((p >> 1) & 1), half = (p & 1) * 8;
    const int r_lo = (2 * s2) * 16 + g * 4 + (p >> 2);   // rows r_lo and r_lo + 16 share (row & 15), hence the swizzle key
    const int col_b = ((ch ^ swz_key<KM>(r_lo)) << 4) + half;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + r_lo * Cfg<HDP>::RS + col_b));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + (r_lo + 16) * Cfg<HDP>::RS + col_b));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// one LDS-DMA piece (16 bytes per lane, lane-linear destination starting at the wave-uniform LDS byte address lds_dst) issued by inline assembly
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, unsigned voffset) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffset), "s"(dst), "s"(rsrc) : "memory");
}

// a pointer the compiler can see is wave-uniform (buffer descriptors must live in SGPRs)
__device__ __forceinline__ void* uniform_ptr(const void* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ s16x8 pack_pair(const f32x4& a, const f32x4& b) {
    typename T16<T>::v8 v;
    v[0] = (T)a[0]; v[1] = (T)a[1]; v[2] = (T)a[2]; v[3] = (T)a[3];
    v[4] = (T)b[0]; v[5] = (T)b[1]; v[6] = (T)b[2]; v[7] = (T)b[3];
    return __builtin_bit_cast(s16x8, v);
}

__device__ __forceinline__ float group_max(float v) { return xor32_max(xor16_max(v)); }   // across the 4 lane groups that share lane&15
__device__ __forceinline__ float group_sum(float v) { return xor32_sum(xor16_sum(v)); }

__device__ __forceinline__ float mask_val(const float* mask, int mode, int b, int i, int j, int Sq, int Sk) {
    if (mode == 1) return mask[(int64_t)b * Sk + j];
    if (mode == 2) return mask[((int64_t)b * Sq + i) * Sk + j];
    return 0.f;
}

// ======================================================================================================================
// forward
// ======================================================================================================================
// RB = 16-row query blocks per wave.  With RB = 1 every K / V^T fragment read from LDS feeds one MFMA, and the kernel is bound by
// LDS bandwidth at twice the MFMA time (24.5 KB of LDS reads per 24 MFMAs per wave and key tile); RB = 2 (long self-attention:
// the ViT towers) reuses each fragment for two query blocks and stages every K/V tile for 128 instead of 64 queries.
// NW = waves per workgroup (round 6).  4: 64 RB query rows per workgroup.  5: 80 - BERT's 77 text rows in training (dropout: RB = 1) as ONE
// workgroup per (b, h) instead of two of 64 + 13 rows: the second one staged every key tile again for one live wave, and since the two sit on
// different XCDs (consecutive workgroup ids) the K / V rows of every head were fetched twice - the cross-attention forward ran at 2.1 TB/s of
// algorithmic K / V bytes on an HBM-bound kernel.  The fifth wave takes no part in the staging (the tile map covers 256 threads); every wave's
// arithmetic is unchanged, so the results are bit for bit the four-wave kernel's.
template <typename T, int HDP, bool DROP, int RB, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, T* __restrict__ o,
                                                              float* __restrict__ lse, const mico_attn_params p) {
    using C = Cfg<HDP>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::TILE];
    LDS_AS char* kt = (LDS_AS char*)smem;
    LDS_AS char* vt = kt + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool stager = NW == 4 || tid < 256;      // (wave-uniform)
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (NW * 16 * RB);
    const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;   // shared K/V memory (see mico_attn_params)
    const T* kb = k + (int64_t)kvb * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)kvb * p.v_bs + h * p.hd;
    const int i0 = q0 + wave * (16 * RB) + (lane & 15);   // this lane's query rows: i0 + rb * 16
    const int g = lane >> 4;
    const bool wave_live = q0 + wave * (16 * RB) < p.Sq;  // wave-uniform: this wave owns at least one real query row

    s16x8 qf[RB][C::KS];
    f32x4 oacc[RB][C::TD];
    float m_run[RB], l_run[RB];   // running max (log2 domain) and sum
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        row_frags<T, HDP>(qf[rb], qb, p.q_rs, i0 + rb * 16, p.Sq, p.hd, lane);
#pragma unroll
        for (int t = 0; t < C::TD; ++t) oacc[rb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_run[rb] = NEG_BIG;
        l_run[rb] = 0.f;
    }
    const float sc2 = p.scale * 1.4426950408889634f;

    const int nt = (p.Sk + 63) / 64;
    const TileMap<HDP> tm_a = tile_map<HDP>(p.k_rs, p.hd, tid), tm_b = tile_map<HDP>(p.v_rs, p.hd, tid);
    s16x8 kr[C::NCH], vr[C::NCH];
    if (stager) {
        tile_fetch<T, HDP>(kr, kb, p.k_rs, 0, p.Sk, tm_a);
        tile_fetch<T, HDP>(vr, vb, p.v_rs, 0, p.Sk, tm_b);
    }
    PH_DECL;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        PH(0);
        if (stager) {
            tile_commit<HDP>(kr, kt, tm_a);
            tile_commit<HDP>(vr, vt, tm_b);
        }
        PH(1);
        __syncthreads();
        PH(2);
        if (stager && t + 1 < nt) {
            tile_fetch<T, HDP>(kr, kb, p.k_rs, (t + 1) * 64, p.Sk, tm_a);
            tile_fetch<T, HDP>(vr, vb, p.v_rs, (t + 1) * 64, p.Sk, tm_b);
        }
        PH(3);
        // S^T = K Q^T.  Only the 16-key sub-tiles that hold real keys are computed: N = 257 (ViT-g/14) ends in a tile with one
        // valid sub-tile, and a full fifth tile cost 15 % of the kernel (tools/attn_bench.py, 257 vs 256 tokens).
        const bool edge = (t == nt - 1) && (p.Sk & 63);   // workgroup-uniform: only the ragged last key tile needs bounds
        const int ntn = edge ? ((p.Sk - t * 64 + 15) >> 4) : 4;
        f32x4 s[RB][4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < ntn && wave_live) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    const s16x8 a = lds_row_frag<HDP>(kt, tn * 16, ks, lane);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) s[rb][tn] = T16<T>::mfma(a, qf[rb][ks], s[rb][tn]);
                }
            }
        }
        s16x8 pf[RB][2];
        PH(4);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int i = i0 + rb * 16;
            // online softmax in the exp2 domain (v_exp_f32 is 2^x): scores are pre-multiplied by scale * log2(e)
            float mloc = NEG_BIG;
            if (!p.mask_mode && !edge) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    s[rb][tn] *= sc2;
                    mloc = fmaxf(fmaxf(mloc, fmaxf(s[rb][tn][0], s[rb][tn][1])), fmaxf(s[rb][tn][2], s[rb][tn][3]));
                }
            } else {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = t * 64 + tn * 16 + g * 4 + r;
                        float x = s[rb][tn][r] * sc2;
                        if (j < p.Sk) {
                            if (p.mask_mode && i < p.Sq) x += mask_val(p.mask, p.mask_mode, b, i, j, p.Sq, p.Sk) * 1.4426950408889634f;
                        } else {
                            x = NEG_BIG;
                        }
                        s[rb][tn][r] = x;
                        mloc = fmaxf(mloc, x);
                    }
            }
            const float m_new = fmaxf(m_run[rb], group_max(mloc));
            const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
            float lloc = 0.f;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[rb][tn][r] - m_new);
                    s[rb][tn][r] = e;
                    lloc += e;
                }
            if (DROP) {   // dropout on the probabilities (compile-time variant: the ViT towers never pay its registers): the row sum above stays the undropped one
                const unsigned thr = drop_threshold(p.drop_p);
                const float ik = 1.f / (1.f - p.drop_p);
                const unsigned long long rowbase = (((unsigned long long)(b + p.batch0) * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk;
                if ((unsigned long long)(p.B + p.batch0) * p.H * p.Sq * p.Sk <= 0xFFFFFFFFull) {   // (launch-uniform) every index below 2^32: the incremental hash
                    const unsigned h0 = p.drop_seed ^ ((unsigned)p.drop_site * 0x9E3779B9u);
                    const unsigned tb = ((unsigned)rowbase + (unsigned)(t * 64 + g * 4)) * 0x85EBCA6Bu;
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s[rb][tn][r] *= drop_hash_t(h0, tb + (unsigned)(tn * 16 + r) * 0x85EBCA6Bu) >= thr ? ik : 0.f;
                } else {
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s[rb][tn][r] *= drop_mult(p.drop_seed, p.drop_site, rowbase + (t * 64 + tn * 16 + g * 4 + r), thr, ik);
                }
            }
            l_run[rb] = l_run[rb] * alpha + group_sum(lloc);
            m_run[rb] = m_new;
#pragma unroll
            for (int t2 = 0; t2 < C::TD; ++t2) oacc[rb][t2] *= alpha;
            pf[rb][0] = pack_pair<T>(s[rb][0], s[rb][1]);
            pf[rb][1] = pack_pair<T>(s[rb][2], s[rb][3]);
        }
        PH(5);
        // O^T += V^T P^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (2 * s2 >= ntn || !wave_live) continue;    // no real key in this 32-key half / no real query in this wave
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                const s16x8 a = lds_tr_frag<HDP>(vt, td, s2, lane);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) oacc[rb][td] = T16<T>::mfma(a, pf[rb][s2], oacc[rb][td]);
            }
        }
        PH(6);
    }
    PH_STORE;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int i = i0 + rb * 16;
        if (i < p.Sq) {
            const float inv = 1.f / l_run[rb];
            T* ob = o + (int64_t)b * p.o_bs + (int64_t)i * p.o_rs + h * p.hd;
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                const int d = td * 16 + g * 4;
                if (d < p.hd)
                    *(s16x4*)(ob + d) = pack4<T>(oacc[rb][td][0] * inv, oacc[rb][td][1] * inv, oacc[rb][td][2] * inv, oacc[rb][td][3] * inv);
            }
            if (g == 0) lse[((int64_t)b * p.H + h) * p.Sq + i] = (m_run[rb] + __log2f(l_run[rb])) * 0.6931471805599453f;
        }
    }
}

// ======================================================================================================================
// forward, K/V-resident variant for unmasked self-attention with Sq, Sk <= 272 (the ViT towers: 257 tokens at patch 14, 197 at
// patch 16).  Persistent 8-wave workgroups, one per CU, each walking over (b, h) items: all of K and V of the head are staged into
// LDS once (2 x 288 rows x 256 B), then the key loop runs without a single barrier, every wave owning two 16-query blocks (rows
// wave*32 ..), while the NEXT item's K/V are already on their way into registers.  The tiled kernel above stages every key tile
// once per 64 queries behind two barriers and is bound by exactly that (tools/probes/attn_phases.py: of 5700 cycles per key tile
// 2150 are barriers, global-load issue and the register -> LDS commit; a fifth tile holding the 257th key and another workgroup
// holding the 257th query cost as much as full ones; a non-persistent resident kernel lost a third of its time to the dispatch
// gap and the exposed staging).  The ragged ends of N = 257:
//  * keys: steps cover 64 keys, the last one 80 when 1..16 keys are left over (5 sub-tiles instead of a fifth step);
//  * queries 256..271 are a 17th block that no wave has room for: it is split over the KEYS instead - wave w adds sub-tiles
//    2w, 2w+1 (wave 7 also the 17th) as a third, small query block to the step that holds them - and the eight partial
//    (max, sum, O) triples are merged through the then idle K region of LDS.
// ======================================================================================================================
constexpr int RES_KR = 288;   // resident key rows (272 rounded up to the 32 keys one transposed V fragment spans)

// One step over NSUB 16-key sub-tiles starting at LDS row pointers ktile / vtile (global key index key0), restricted to sub-tiles
// [lo, hi).  mode 0: all NSUB sub-tiles in range and inside Sk; 1: as 0 but the last sub-tile is ragged; 2: general.
// ALL: every sub-tile is in range (lo = 0, hi = NSUB) - no per-sub-tile branches, so each of the two MFMA phases is one basic block and
// the compiler can run its LDS fragment reads many MFMAs ahead (with the branches it waited on nearly every read: ~30 exposed LDS
// latencies per step, tools/probes/attn_phases.py: 4400 cycles per step against ~1500 of issue work).
template <typename T, int HDP, int RB, int NSUB, bool ALL>
__device__ __forceinline__ void res_tile_step(LDS_AS const char* ktile, LDS_AS const char* vtile, const int key0, const int lo,
                                              const int hi, const int mode, const s16x8 (&qf)[RB][Cfg<HDP>::KS],
                                              f32x4 (&oacc)[RB][Cfg<HDP>::TD], float (&m_run)[RB], float (&l_run)[RB],
                                              const int Sk, const float sc2, const int lane,
                                              const int (&rf)[Cfg<HDP>::KS], const int (&tf)[Cfg<HDP>::TD]) {
    using C = Cfg<HDP>;
    constexpr int NH = (NSUB + 1) / 2;
    const int g = lane >> 4;
    f32x4 s[RB][NH * 2];
    if (ALL) {   // fragments of sub-tile tn + 1 are requested before the MFMAs of sub-tile tn are issued; the order is pinned
        s16x8 an[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) an[ks] = *(LDS_AS const s16x8*)(ktile + rf[ks]);
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn >= NSUB) continue;
            s16x8 ac[C::KS];
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) ac[ks] = an[ks];
            if (tn + 1 < NSUB) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) an[ks] = *(LDS_AS const s16x8*)(ktile + rf[ks] + (tn + 1) * 16 * C::RS);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) s[rb][tn] = T16<T>::mfma(ac[ks], qf[rb][ks], s[rb][tn]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (tn < NSUB && tn >= lo && tn < hi) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    const s16x8 a = *(LDS_AS const s16x8*)(ktile + rf[ks] + tn * 16 * C::RS);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) s[rb][tn] = T16<T>::mfma(a, qf[rb][ks], s[rb][tn]);
                }
            }
        }
    }
    s16x8 pf[RB][NH];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float mloc = NEG_BIG;
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
            if (tn >= NSUB) {
                s[rb][tn] = (f32x4){NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
            } else if (mode == 0 || (mode == 1 && tn < NSUB - 1)) {
                s[rb][tn] *= sc2;
                mloc = fmaxf(fmaxf(mloc, fmaxf(s[rb][tn][0], s[rb][tn][1])), fmaxf(s[rb][tn][2], s[rb][tn][3]));
            } else {
                // one uniform limit per sub-tile and a lane compare per element (per-element uniform conditions end up as spilled
                // SGPR pairs read back lane by lane)
                const int lim = (ALL || (tn >= lo && tn < hi)) ? Sk - key0 - tn * 16 - g * 4 : 0;   // this lane's keys r < lim are real
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = r < lim ? s[rb][tn][r] * sc2 : NEG_BIG;
                    s[rb][tn][r] = x;
                    mloc = fmaxf(mloc, x);
                }
            }
        }
        const float m_new = fmaxf(m_run[rb], group_max(mloc));
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        float lloc = 0.f;
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
            if (tn >= NSUB) {
                s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[rb][tn][r] - m_new);
                    s[rb][tn][r] = e;
                    lloc += e;
                }
            }
        }
        l_run[rb] = l_run[rb] * alpha + group_sum(lloc);
        m_run[rb] = m_new;
#pragma unroll
        for (int t2 = 0; t2 < C::TD; ++t2) oacc[rb][t2] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < NH; ++s2) pf[rb][s2] = pack_pair<T>(s[rb][2 * s2], s[rb][2 * s2 + 1]);
    }
    auto vfrag = [&](int idx) -> s16x8 {   // idx = s2 * TD + td
        const int s2 = idx / C::TD, td = idx - s2 * C::TD;
        const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vtile + tf[td] + s2 * 32 * C::RS));
        const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(vtile + tf[td] + s2 * 32 * C::RS + 16 * C::RS));
        return (s16x8){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
    };
    if (ALL) {   // three V^T fragments in flight ahead of the MFMAs
        constexpr int NF = NH * C::TD, AH = 3;
        s16x8 ring[AH];
#pragma unroll
        for (int i = 0; i < AH; ++i) ring[i] = vfrag(i);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const s16x8 a = ring[i % AH];
            if (i + AH < NF) ring[i % AH] = vfrag(i + AH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) oacc[rb][i % C::TD] = T16<T>::mfma(a, pf[rb][i / C::TD], oacc[rb][i % C::TD]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int s2 = 0; s2 < NH; ++s2) {
            if (2 * s2 + 1 < lo || 2 * s2 >= hi) continue;   // no sub-tile of this 32-key half in range
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                const s16x8 a = vfrag(s2 * C::TD + td);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) oacc[rb][td] = T16<T>::mfma(a, pf[rb][s2], oacc[rb][td]);
            }
        }
    }
}

// EXPERIMENT (round 5, off): the last key of Sk = 64 k + 1 as the initial state of the running softmax instead of a fifth key step (see `r1` in
// attn_fwd_res_kernel).  Measured (tools/attn_bench.py, 320 frames x 16 heads x 257 x hd 88): forward 0.386 -> 0.362 ms (-6 %; ~0.13 % of the omni
// step), all attention tests green - but the un-rounded last probability moves the max-norm statistic of the timed precision over 8 images
// (tests/test_precision_stats_gpu.py) from 9.2e-4 / 9.8e-4 to 1.03e-3 / 1.12e-3 (tokens / feat_v): the gate's margin is worth more than 2 ms.
#ifndef MICO_ATTN_R1
#define MICO_ATTN_R1 0
#endif
template <int HDP> struct ResCfg {
    static constexpr int NXMAX = HDP <= 96 ? 4 : 3;    // query rows beyond 256 (their partials live in the LDS left over by K/V)
    static constexpr int PST = HDP + 2;                // floats per partial row: O[HDP], max, sum
};

template <typename T, int HDP>
__global__ __launch_bounds__(512, 1) void attn_fwd_res_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, T* __restrict__ o,
                                                              float* __restrict__ lse, const mico_attn_params p) {
    using C = Cfg<HDP>;
    using R = ResCfg<HDP>;
    constexpr int CPR = HDP / 8;                  // 16-byte chunks per row
    constexpr int NLD = 256 * CPR / 512;          // chunks per thread and matrix for rows 0..255
    __shared__ __attribute__((aligned(16))) char smem[2 * RES_KR * C::RS + 8 * R::NXMAX * R::PST * 4];
    LDS_AS char* kt = (LDS_AS char*)smem;
    LDS_AS char* vt = kt + RES_KR * C::RS;
    LDS_AS float* part = (LDS_AS float*)(vt + RES_KR * C::RS);   // [wave][NXMAX][PST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const float sc2 = p.scale * 1.4426950408889634f;
    const int nitems = p.B * p.H;

    // rows 256..287 start as zeros; rows Sk..287 stay zero for the whole kernel (the commits below only write rows < max(Sk, 256))
    for (int c = tid; c < 32 * 16; c += 512) {
        *(LDS_AS s16x8*)(kt + 256 * C::RS + c * 16) = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        *(LDS_AS s16x8*)(vt + 256 * C::RS + c * 16) = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    // this thread's chunks: rows 0..255 -> NLD per matrix; rows 256.. -> at most one
    const int xrow = 256 + tid / CPR, xch = tid - (tid / CPR) * CPR;
    const bool xlive = xrow < p.Sk && xch * 8 < p.hd;
    const int xoff = xrow * C::RS + ((xch ^ ((xrow & 7) << 1)) << 4);

    // lane-dependent, tile-relative LDS byte offsets of the row fragments (per k-step) and transposed fragments (per 16 head dims);
    // sub-tile and 32-key-half displacements are immediates on top (the swizzle key only depends on row & 7)
    int rf[C::KS], tf[C::TD];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) rf[ks] = (lane & 15) * C::RS + (((ks * 4 + g) ^ ((lane & 7) << 1)) << 4);
    {
        const int pp = lane & 15, r8 = g * 4 + (pp >> 2);
#pragma unroll
        for (int td = 0; td < C::TD; ++td) tf[td] = r8 * C::RS + (((td * 2 + ((pp >> 1) & 1)) ^ ((r8 & 7) << 1)) << 4) + (pp & 1) * 8;
    }
    // key steps: 64 keys each; the last one takes 80 when that saves a step
    const int nst = (p.Sk + 15) >> 4;
    constexpr bool WIDE = HDP <= 64;   // the 80-key step needs 24 more registers than hd 96 / 128 leave next to the prefetch
    const bool wide_last = WIDE && nst > 4 && (nst & 3) == 1;
    // Sk = 64 k + 1 (the towers' 257 tokens) without the 80-key step: the LAST key is not a fifth step over one live sub-tile column (a 16-key
    // MFMA pass, a softmax round and a rescale of the output tiles for ONE key) but the INITIAL state of the running softmax - m = s(q, k_last),
    // l = 1, O = v_last - computed with a dot product per query row (MICO_ATTN_R1=0: the fifth step, for A/B runs)
    const bool r1 = MICO_ATTN_R1 && !WIDE && p.Sk > 64 && (p.Sk & 63) == 1;
    const int nstm = r1 ? nst - 1 : nst;      // 16-key sub-tiles of the main pass
    const int nsteps = wide_last ? (nst >> 2) : ((nstm + 3) >> 2);
    // the 17th query block: this wave's key sub-tiles [4*xt + xlo, 4*xt + xhi)
    const int nx = p.Sq - 256;
    const int xt = wave >> 1, xlo = (wave & 1) * 2;
    const int xhi = wave == 7 ? nst - 12 : min(xlo + 2, nst - 4 * xt);   // wave 7: sub-tiles 14.. incl. the 17th (5-wide step template)

    // buffer loads: 32-bit per-thread byte offsets (item invariant) against a per-item descriptor; dead chunks (row >= Sk, head-dim
    // padding) point out of bounds and come back as zeros
    // (K and V share the offsets: the launcher requires k_rs == v_rs; recomputed per use - a handful of integer operations against 16 registers)
    auto kv_off = [&](int it) -> unsigned {
        if (it == NLD) return xlive ? (unsigned)(xrow * p.k_rs * 2 + xch * 16) : 0xFFFFFFF0u;
        const int c = it * 512 + tid;
        const int row = c / CPR, ch = c - row * CPR;
        return (row < p.Sk && ch * 8 < p.hd) ? (unsigned)(row * p.k_rs * 2 + ch * 16) : 0xFFFFFFF0u;
    };
    const int kbytes = (int)(((int64_t)(p.Sk - 1) * p.k_rs + p.hd) * 2), vbytes = (int)(((int64_t)(p.Sk - 1) * p.v_rs + p.hd) * 2);
    // Q row fragments the same way (no divergent branches around the loads: the wait counts stay static)
    const int qbytes = (int)(((int64_t)(p.Sq - 1) * p.q_rs + p.hd) * 2);
    auto q_off = [&](int rb, int ks) -> unsigned {
        const int row = (rb < 2 ? wave * 32 + rb * 16 : 256) + (lane & 15), d = ks * 32 + g * 8;
        return (row < p.Sq && d < p.hd) ? (unsigned)(row * p.q_rs * 2 + d * 2) : 0xFFFFFFF0u;
    };
    s16x8 kr[NLD + 1], vr[NLD + 1];
    // The next item's K chunks are requested in four parts spread over the key steps of the current item (back to back they stall every
    // wave for ~3000 cycles on the address / data path of the CU and leave the waves staggered by that much); its V chunks and query
    // fragments follow right after the last step, into the registers the score tiles no longer need - they have the output stores, the
    // barrier and the merge to arrive.
    auto prefetch_part = [&](int item, int part_id) {
        const int b = item / p.H, h = item - b * p.H;
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * p.k_bs + h * p.hd), 0, kbytes, 0x00020000);
#pragma unroll
        for (int it = 0; it <= NLD; ++it) {
            if (it * 4 / (NLD + 1) != part_id) continue;
            kr[it] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, kv_off(it), 0, 0));
        }
    };
    auto load_v = [&](int item) {
        const int b = item / p.H, h = item - b * p.H;
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * p.v_bs + h * p.hd), 0, vbytes, 0x00020000);
#pragma unroll
        for (int it = 0; it <= NLD; ++it) vr[it] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, kv_off(it), 0, 0));
    };
    s16x8 qf[2][C::KS], qx[1][C::KS];
    auto load_q = [&](int item) {
        const int b = item / p.H, h = item - b * p.H;
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(q + (int64_t)b * p.q_bs + h * p.hd), 0, qbytes, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            qf[0][ks] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off(0, ks), 0, 0));
            qf[1][ks] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off(1, ks), 0, 0));
            qx[0][ks] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, q_off(2, ks), 0, 0));
        }
    };
    // merge of the eight key-slices of query rows 256.. of item `it` (partials in LDS) -> O rows and lse
    auto merge = [&](int it) {
        const int b = it / p.H, h = it - b * p.H;
        for (int e = tid; e < nx * (HDP / 4); e += 512) {
            const int row = e / (HDP / 4), d = (e - row * (HDP / 4)) * 4;
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < 8; ++w) M = fmaxf(M, part[(w * R::NXMAX + row) * R::PST + HDP]);
            float L = 0.f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                LDS_AS const float* pr = part + (w * R::NXMAX + row) * R::PST;
                const float f = __builtin_amdgcn_exp2f(pr[HDP] - M);
                L += pr[HDP + 1] * f;
                acc[0] += pr[d] * f; acc[1] += pr[d + 1] * f; acc[2] += pr[d + 2] * f; acc[3] += pr[d + 3] * f;
            }
            const float inv = 1.f / L;
            const int i = 256 + row;
            if (d < p.hd) *(s16x4*)(o + (int64_t)b * p.o_bs + (int64_t)i * p.o_rs + h * p.hd + d) = pack4<T>(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
            if (d == 0) lse[((int64_t)b * p.H + h) * p.Sq + i] = (M + __log2f(L)) * 0.6931471805599453f;
        }
    };
    // consecutive workgroups sit on different XCDs: give each XCD a contiguous range of items, so that the heads of one frame
    // (interleaved pieces of the same QKV rows) meet in one L2
    const int nwg = gridDim.x;
    int item = (int)blockIdx.x;
    int item_step = nwg;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {
        const int per_wg = nitems / nwg;
        item = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * per_wg;
        item_step = 1;
    }
    const int item_end = item_step == 1 ? item + nitems / nwg : nitems;
    if (item < item_end) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) prefetch_part(item, pt);
        load_v(item);
        load_q(item);
    }

    int prev = -1;
    PH_DECL;
    for (; item < item_end; item += item_step) {
        PH(7);
        const int b = item / p.H, h = item - b * p.H;
#if MICO_ATTN_PRIO
        if (((item ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (see attn_bwd_onepass_kernel)
#endif
        const int next = item + item_step < item_end ? item + item_step : item;   // the last item re-fetches itself (static wait counts)
        __syncthreads();   // every wave is done with the previous item's K/V and has written its partials
        PH(0);
        if (nx > 0 && prev >= 0) merge(prev);
        PH(1);
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int c = it * 512 + tid;
            const int row = c / CPR, ch = c - row * CPR;
            const int off = row * C::RS + ((ch ^ ((row & 7) << 1)) << 4);
            *(LDS_AS s16x8*)(kt + off) = kr[it];
            *(LDS_AS s16x8*)(vt + off) = vr[it];
        }
        if (xlive) {
            *(LDS_AS s16x8*)(kt + xoff) = kr[NLD];
            *(LDS_AS s16x8*)(vt + xoff) = vr[NLD];
        }
        PH(2);
        __syncthreads();
        PH(3);
        prefetch_part(next, 0);
        PH(4);
        if (nx > 0) {   // this wave's key slice of query rows 256..: partial (max, sum, O) -> LDS, merged at the next item's top
            f32x4 ox[1][C::TD];
            float mx[1] = {NEG_BIG}, lx[1] = {0.f};
#pragma unroll
            for (int t = 0; t < C::TD; ++t) ox[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xhi > xlo)
                res_tile_step<T, HDP, 1, 5, false>(kt + xt * C::TILE, vt + xt * C::TILE, xt * 64, xlo, xhi, 2, qx, ox, mx, lx, p.Sk, sc2, lane, rf, tf);
            if ((lane & 15) < nx) {
                LDS_AS float* pr = part + (wave * R::NXMAX + (lane & 15)) * R::PST;
                if (g == 0) { pr[HDP] = mx[0]; pr[HDP + 1] = lx[0]; }
#pragma unroll
                for (int td = 0; td < C::TD; ++td)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[td * 16 + g * 4 + r] = ox[0][td][r];
            }
        }
        prev = item;
        PH(5);
        const bool main_live = wave * 32 < p.Sq;   // wave-uniform: this wave owns query rows of the main pass
        const int i0 = wave * 32 + (lane & 15);
        f32x4 oacc[2][C::TD];
        float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < C::TD; ++t) {
            oacc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            oacc[1][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (r1 && main_live) {
            LDS_AS const char* krow = kt + (p.Sk - 1) * C::RS;      // (row & 7 == 0: chunk c of the row sits in chunk slot c)
            LDS_AS const char* vrow = vt + (p.Sk - 1) * C::RS;
            float dot[2] = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                float k8[8];
                unpack8<T>(*(LDS_AS const s16x8*)(krow + ((ks * 4 + g) << 4)), k8);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    float q8[8];
                    unpack8<T>(qf[rb][ks], q8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot[rb] = __builtin_fmaf(q8[e], k8[e], dot[rb]);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                m_run[rb] = group_sum(dot[rb]) * sc2;
                l_run[rb] = 1.f;
            }
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {      // this lane's head dims of output tile td: td * 16 + g * 4 .. + 3
                const f32x4 v4 = unpack4<T>(*(LDS_AS const s16x4*)(vrow + ((td * 2 + (g >> 1)) << 4) + (g & 1) * 8));
                oacc[0][td] = v4;
                oacc[1][td] = v4;
            }
        }
        // The step loop is unrolled by hand (at most 5 steps) with the prefetch parts as unconditional straight-line code in between:
        // loads issued under control flow come back through copies at the loop back-edge, i.e. behind an s_waitcnt vmcnt(0).
        auto key_step = [&](int t) {
            if (!main_live || t >= nsteps) return;
            LDS_AS const char* ktile = kt + t * C::TILE;
            LDS_AS const char* vtile = vt + t * C::TILE;
            if (WIDE && t == nsteps - 1 && wide_last) {
                res_tile_step<T, HDP, 2, 5, true>(ktile, vtile, t * 64, 0, 5, (p.Sk & 15) ? 1 : 0, qf, oacc, m_run, l_run, p.Sk, sc2, lane, rf, tf);
            } else {
                const int ntn = min(4, nstm - 4 * t);
                if (ntn == 4) res_tile_step<T, HDP, 2, 4, true>(ktile, vtile, t * 64, 0, 4, t * 64 + 64 <= p.Sk ? 0 : 1, qf, oacc, m_run, l_run, p.Sk, sc2, lane, rf, tf);
                else res_tile_step<T, HDP, 2, 4, false>(ktile, vtile, t * 64, 0, ntn, 2, qf, oacc, m_run, l_run, p.Sk, sc2, lane, rf, tf);
            }
        };
        key_step(0);
        prefetch_part(next, 1);
        key_step(1);
        prefetch_part(next, 2);
        key_step(2);
        prefetch_part(next, 3);
        key_step(3);
        key_step(4);
        load_v(next);
        load_q(next);   // qf / qx are dead from here on: the next item's query fragments take their registers
        PH(6);
        if (main_live) {
            // 16-byte stores (as attn_bwd_onepass_kernel's dK / dV rows): v_permlane16_swap trades one head-dim tile's quartet with the
            // neighbouring lane group for the next tile's, so even groups hold 8 consecutive head dims of tile td and odd groups of tile td + 1 -
            // half the store instructions (each touches 16 token rows: the address path, not the bytes, is what they cost)
            static_assert(C::TD % 2 == 0, "tile pairs");
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int i = i0 + rb * 16;
                const float inv = 1.f / l_run[rb];
                T* ob = o + (int64_t)b * p.o_bs + (int64_t)i * p.o_rs + h * p.hd;
#pragma unroll
                for (int tp = 0; tp < C::TD; tp += 2) {
                    const f32x4 ta = oacc[rb][tp], tb = oacc[rb][tp + 1];
                    u32x2 a2 = {pack2_scaled<T>(ta[0], ta[1], inv), pack2_scaled<T>(ta[2], ta[3], inv)};
                    u32x2 c2 = {pack2_scaled<T>(tb[0], tb[1], inv), pack2_scaled<T>(tb[2], tb[3], inv)};
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a2[0]), "+v"(c2[0]));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a2[1]), "+v"(c2[1]));
                    const int d = (tp + (g & 1)) * 16 + (g >> 1) * 8;
                    if (i < p.Sq && d < p.hd) *(u32x4*)(ob + d) = (u32x4){a2[0], a2[1], c2[0], c2[1]};
                }
                if (i < p.Sq && g == 0) lse[((int64_t)b * p.H + h) * p.Sq + i] = (m_run[rb] + __log2f(l_run[rb])) * 0.6931471805599453f;
            }
        }
    }
    if (nx > 0 && prev >= 0) {
        __syncthreads();
        merge(prev);
    }
#ifdef MICO_ATTN_PHASES
    if (lane == 0 && blockIdx.x < 512) for (int e_ = 0; e_ < 8; ++e_) g_attn_phase[(blockIdx.x * 8 + wave) * 8 + e_] = ph_acc[e_];   // per wave
#endif
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
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;   // shared K/V memory (see mico_attn_params)
    const T* kb = k + (int64_t)kvb * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)kvb * p.v_bs + h * p.hd;
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
                        dpe *= drop_mult(p.drop_seed, p.drop_site, (((unsigned long long)(b + p.batch0) * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk + j,
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
// backward dQ, K/V-resident variant (same eligibility, work split and ragged-end handling as attn_fwd_res_kernel): persistent 8-wave
// workgroups, K and V of one (b, h) in LDS, two 16-query blocks per wave, no barrier inside the key loop; the 17th query block is
// split over the keys and its eight partial dQ tiles are summed through LDS.  No register prefetch of the next item here: the two
// score tiles (S and dP) take the registers the forward kernel spends on it.
// ======================================================================================================================
template <typename T, int HDP, int RB, int NSUB, bool ALL>
__device__ __forceinline__ void res_dq_step(LDS_AS const char* ktile, LDS_AS const char* vtile, const int key0, const int lo,
                                            const int hi, const bool full, const s16x8 (&qf)[RB][Cfg<HDP>::KS],
                                            const s16x8 (&dof)[RB][Cfg<HDP>::KS], f32x4 (&dqacc)[RB][Cfg<HDP>::TD],
                                            const float (&lse2)[RB], const float (&dl)[RB], const int Sk, const float scale,
                                            const int lane, const int (&rf)[Cfg<HDP>::KS], const int (&tf)[Cfg<HDP>::TD]) {
    using C = Cfg<HDP>;
    constexpr int NH = (NSUB + 1) / 2;
    const int g = lane >> 4;
    const float sc2 = scale * 1.4426950408889634f;
    f32x4 s[RB][NH * 2], dp[RB][NH * 2];
    if (ALL) {   // K / V fragments of sub-tile tn + 1 are requested before the MFMAs of sub-tile tn are issued; the order is pinned
        s16x8 an[C::KS], cn[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            an[ks] = *(LDS_AS const s16x8*)(ktile + rf[ks]);
            cn[ks] = *(LDS_AS const s16x8*)(vtile + rf[ks]);
        }
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (tn >= NSUB) continue;
            s16x8 ac[C::KS], cc[C::KS];
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) { ac[ks] = an[ks]; cc[ks] = cn[ks]; }
            if (tn + 1 < NSUB) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    an[ks] = *(LDS_AS const s16x8*)(ktile + rf[ks] + (tn + 1) * 16 * C::RS);
                    cn[ks] = *(LDS_AS const s16x8*)(vtile + rf[ks] + (tn + 1) * 16 * C::RS);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    s[rb][tn] = T16<T>::mfma(ac[ks], qf[rb][ks], s[rb][tn]);
                    dp[rb][tn] = T16<T>::mfma(cc[ks], dof[rb][ks], dp[rb][tn]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                s[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[rb][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (tn < NSUB && tn >= lo && tn < hi) {
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    const s16x8 a = *(LDS_AS const s16x8*)(ktile + rf[ks] + tn * 16 * C::RS);
                    const s16x8 c = *(LDS_AS const s16x8*)(vtile + rf[ks] + tn * 16 * C::RS);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        s[rb][tn] = T16<T>::mfma(a, qf[rb][ks], s[rb][tn]);
                        dp[rb][tn] = T16<T>::mfma(c, dof[rb][ks], dp[rb][tn]);
                    }
                }
            }
        }
    }
    s16x8 df[RB][NH];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int tn = 0; tn < NH * 2; ++tn) {
            const int lim = full ? 4 : ((tn < NSUB && (ALL || (tn >= lo && tn < hi))) ? Sk - key0 - tn * 16 - g * 4 : 0);   // lane's keys r < lim are real
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(s[rb][tn][r] * sc2 - lse2[rb]);
                const float d = pr * (dp[rb][tn][r] - dl[rb]) * scale;
                s[rb][tn][r] = r < lim ? d : 0.f;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < NH; ++s2) df[rb][s2] = pack_pair<T>(s[rb][2 * s2], s[rb][2 * s2 + 1]);
    }
    auto kfrag = [&](int idx) -> s16x8 {   // idx = s2 * TD + td
        const int s2 = idx / C::TD, td = idx - s2 * C::TD;
        const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ktile + tf[td] + s2 * 32 * C::RS));
        const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ktile + tf[td] + s2 * 32 * C::RS + 16 * C::RS));
        return (s16x8){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
    };
    if (ALL) {
        constexpr int NF = NH * C::TD, AH = 3;
        s16x8 ring[AH];
#pragma unroll
        for (int i = 0; i < AH; ++i) ring[i] = kfrag(i);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const s16x8 a = ring[i % AH];
            if (i + AH < NF) ring[i % AH] = kfrag(i + AH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dqacc[rb][i % C::TD] = T16<T>::mfma(a, df[rb][i / C::TD], dqacc[rb][i % C::TD]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int s2 = 0; s2 < NH; ++s2) {
            if (2 * s2 + 1 < lo || 2 * s2 >= hi) continue;
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                const s16x8 a = kfrag(s2 * C::TD + td);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) dqacc[rb][td] = T16<T>::mfma(a, df[rb][s2], dqacc[rb][td]);
            }
        }
    }
}

template <typename T, int HDP>
__global__ __launch_bounds__(512, 1) void attn_bwd_dq_res_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                 const T* __restrict__ v, const T* __restrict__ o,
                                                                 const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                 T* __restrict__ dq, float* __restrict__ delta,
                                                                 const mico_attn_params p) {
    using C = Cfg<HDP>;
    using R = ResCfg<HDP>;
    constexpr int CPR = HDP / 8;
    constexpr int NLD = 256 * CPR / 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * RES_KR * C::RS + 8 * R::NXMAX * R::PST * 4];
    LDS_AS char* kt = (LDS_AS char*)smem;
    LDS_AS char* vt = kt + RES_KR * C::RS;
    LDS_AS float* part = (LDS_AS float*)(vt + RES_KR * C::RS);   // [wave][NXMAX][PST] partial dQ rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int nitems = p.B * p.H;
    constexpr float LOG2E = 1.4426950408889634f;

    for (int c = tid; c < 32 * 16; c += 512) {
        *(LDS_AS s16x8*)(kt + 256 * C::RS + c * 16) = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        *(LDS_AS s16x8*)(vt + 256 * C::RS + c * 16) = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const int xrow = 256 + tid / CPR, xch = tid - (tid / CPR) * CPR;
    const bool xlive = xrow < p.Sk && xch * 8 < p.hd;
    const int xoff = xrow * C::RS + ((xch ^ ((xrow & 7) << 1)) << 4);
    int rf[C::KS], tf[C::TD];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) rf[ks] = (lane & 15) * C::RS + (((ks * 4 + g) ^ ((lane & 7) << 1)) << 4);
    {
        const int pp = lane & 15, r8 = g * 4 + (pp >> 2);
#pragma unroll
        for (int td = 0; td < C::TD; ++td) tf[td] = r8 * C::RS + (((td * 2 + ((pp >> 1) & 1)) ^ ((r8 & 7) << 1)) << 4) + (pp & 1) * 8;
    }
    const int nst = (p.Sk + 15) >> 4;
    const int nsteps = (nst + 3) >> 2;
    const int nx = p.Sq - 256;
    const int xt = wave >> 1, xlo = (wave & 1) * 2;
    const int xhi = wave == 7 ? nst - 12 : min(xlo + 2, nst - 4 * xt);
    const int kbytes = (int)(((int64_t)(p.Sk - 1) * p.k_rs + p.hd) * 2), vbytes = (int)(((int64_t)(p.Sk - 1) * p.v_rs + p.hd) * 2);
    const int qbytes = (int)(((int64_t)(p.Sq - 1) * p.q_rs + p.hd) * 2), obytes = (int)(((int64_t)(p.Sq - 1) * p.o_rs + p.hd) * 2);

    // sum of the eight key-slices of dQ rows 256.. of item `it`
    auto merge = [&](int it) {
        const int b = it / p.H, h = it - b * p.H;
        for (int e = tid; e < nx * (HDP / 4); e += 512) {
            const int row = e / (HDP / 4), d = (e - row * (HDP / 4)) * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                LDS_AS const float* pr = part + (w * R::NXMAX + row) * R::PST;
                acc[0] += pr[d]; acc[1] += pr[d + 1]; acc[2] += pr[d + 2]; acc[3] += pr[d + 3];
            }
            if (d < p.hd) *(s16x4*)(dq + (int64_t)b * p.q_bs + (int64_t)(256 + row) * p.q_rs + h * p.hd + d) = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
        }
    };
    const int nwg = gridDim.x;
    int item = (int)blockIdx.x;
    int item_step = nwg;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {
        const int per_wg = nitems / nwg;
        item = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * per_wg;
        item_step = 1;
    }
    const int item_end = item_step == 1 ? item + nitems / nwg : nitems;

    int prev = -1;
    for (; item < item_end; item += item_step) {
        const int b = item / p.H, h = item - b * p.H;
        const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)b * p.k_bs + h * p.hd), 0, kbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)b * p.v_bs + h * p.hd), 0, vbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(q + (int64_t)b * p.q_bs + h * p.hd), 0, qbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(o + (int64_t)b * p.o_bs + h * p.hd), 0, obytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(d_o + (int64_t)b * p.o_bs + h * p.hd), 0, obytes, 0x00020000);
        // ---- K, V -> registers -> LDS; this wave's query-side operands: rows wave*32 + {0, 16} + (lane & 15), and rows 256.. ----------
        // (O is only needed for delta = rowsum(dO * O): it is fetched after the K/V staging registers are free again)
        s16x8 qf[2][C::KS], dof[2][C::KS], qx[1][C::KS], dox[1][C::KS];
        float dl[2], lse2[2], dlx[1], lsx[1];
        auto row_of = [&](int rb) { return (rb < 2 ? wave * 32 + rb * 16 : 256) + (lane & 15); };
        {
            s16x8 kr[NLD + 1], vr[NLD + 1];
#pragma unroll
            for (int it = 0; it <= NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = it < NLD ? c / CPR : xrow, ch = it < NLD ? c - (c / CPR) * CPR : xch;
                const bool live = it < NLD ? (row < p.Sk && ch * 8 < p.hd) : xlive;
                const unsigned off = live ? (unsigned)(row * p.k_rs * 2 + ch * 16) : 0xFFFFFFF0u;
                kr[it] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0));
                vr[it] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0));
            }
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) {
                    const int row = row_of(rb), d = ks * 32 + g * 8;
                    const bool live = row < p.Sq && d < p.hd;
                    const s16x8 qv = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rq, live ? (unsigned)(row * p.q_rs * 2 + d * 2) : 0xFFFFFFF0u, 0, 0));
                    const s16x8 dv = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rd, live ? (unsigned)(row * p.o_rs * 2 + d * 2) : 0xFFFFFFF0u, 0, 0));
                    if (rb < 2) { qf[rb][ks] = qv; dof[rb][ks] = dv; } else { qx[0][ks] = qv; dox[0][ks] = dv; }
                }
            __syncthreads();   // every wave is done with the previous item's K/V and has written its partials
            if (nx > 0 && prev >= 0) merge(prev);
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = c / CPR, ch = c - row * CPR;
                const int off = row * C::RS + ((ch ^ ((row & 7) << 1)) << 4);
                *(LDS_AS s16x8*)(kt + off) = kr[it];
                *(LDS_AS s16x8*)(vt + off) = vr[it];
            }
            if (xlive) {
                *(LDS_AS s16x8*)(kt + xoff) = kr[NLD];
                *(LDS_AS s16x8*)(vt + xoff) = vr[NLD];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int row = row_of(rb);
            float acc = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const int d = ks * 32 + g * 8;
                const unsigned oo = (row < p.Sq && d < p.hd) ? (unsigned)(row * p.o_rs * 2 + d * 2) : 0xFFFFFFF0u;
                const s16x8 ov = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(ro, oo, 0, 0));
                float a8[8], c8[8];
                unpack8<T>(ov, a8);
                unpack8<T>(rb < 2 ? dof[rb][ks] : dox[0][ks], c8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += a8[e] * c8[e];
            }
            acc = group_sum(acc);
            float l = 0.f;
            if (row < p.Sq) {
                l = lse[stat_base + row];
                if (g == 0 && (rb < 2 || wave == 0)) delta[stat_base + row] = acc;
            }
            if (rb < 2) { dl[rb] = acc; lse2[rb] = l * LOG2E; } else { dlx[0] = acc; lsx[0] = l * LOG2E; }
        }
        __syncthreads();
        // ---- dQ rows 256..: this wave's key slice -> partial tile in LDS, summed at the next item's top ---------------------------
        if (nx > 0) {
            f32x4 ox[1][C::TD];
#pragma unroll
            for (int t = 0; t < C::TD; ++t) ox[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xhi > xlo)
                res_dq_step<T, HDP, 1, 5, false>(kt + xt * C::TILE, vt + xt * C::TILE, xt * 64, xlo, xhi, false, qx, dox, ox, lsx, dlx, p.Sk, p.scale, lane, rf, tf);
            if ((lane & 15) < nx) {
                LDS_AS float* pr = part + (wave * R::NXMAX + (lane & 15)) * R::PST;
#pragma unroll
                for (int td = 0; td < C::TD; ++td)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[td * 16 + g * 4 + r] = ox[0][td][r];
            }
        }
        prev = item;
        if (wave * 32 >= p.Sq) continue;   // wave-uniform
        f32x4 dqacc[2][C::TD];
#pragma unroll
        for (int t = 0; t < C::TD; ++t) {
            dqacc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dqacc[1][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int t = 0; t < nsteps; ++t) {
            const int ntn = min(4, nst - 4 * t);
            if (ntn == 4) res_dq_step<T, HDP, 2, 4, true>(kt + t * C::TILE, vt + t * C::TILE, t * 64, 0, 4, t * 64 + 64 <= p.Sk, qf, dof, dqacc, lse2, dl, p.Sk, p.scale, lane, rf, tf);
            else res_dq_step<T, HDP, 2, 4, false>(kt + t * C::TILE, vt + t * C::TILE, t * 64, 0, ntn, false, qf, dof, dqacc, lse2, dl, p.Sk, p.scale, lane, rf, tf);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int i = wave * 32 + rb * 16 + (lane & 15);
            if (i < p.Sq) {
                T* dqb = dq + (int64_t)b * p.q_bs + (int64_t)i * p.q_rs + h * p.hd;
#pragma unroll
                for (int td = 0; td < C::TD; ++td) {
                    const int d = td * 16 + g * 4;
                    if (d < p.hd) *(s16x4*)(dqb + d) = pack4<T>(dqacc[rb][td][0], dqacc[rb][td][1], dqacc[rb][td][2], dqacc[rb][td][3]);
                }
            }
        }
    }
    if (nx > 0 && prev >= 0) {
        __syncthreads();
        merge(prev);
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
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;   // shared K/V memory (see mico_attn_params)
    const T* kb = k + (int64_t)kvb * p.k_bs + h * p.hd;
    const T* vb = v + (int64_t)kvb * p.v_bs + h * p.hd;
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
    // LDS carries lse * log2(e) and delta * scale: the element loop below is then fma - exp2 - fma - mul
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    if (tid < 64 && tid < p.Sq) { st_l = lse[stat_base + tid] * LOG2E; st_d = delta[stat_base + tid] * p.scale; }
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
            if (tid < 64 && ii < p.Sq) { st_l = lse[stat_base + ii] * LOG2E; st_d = delta[stat_base + ii] * p.scale; }
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
        // workgroup-uniform fast path: unmasked, no dropout, all 64 queries of the tile and all 64 keys of the block real (16 of the 25
        // tile pairs at N = 257): 4 VALU operations per score instead of ~14 - these kernels are bound by VALU + MFMA issue
        const bool fast = !DROP && !p.mask_mode && t * 64 + 64 <= p.Sq && k0 + 64 <= p.Sk;
        if (fast) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + ti * 16 + g * 4);
                const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + ti * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[ti][r], sc2, -lv[r]));
                    pr[ti][r] = pv;
                    s[ti][r] = pv * fmaf(dp[ti][r], p.scale, -dv4[r]);
                }
            }
        } else {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + ti * 16 + g * 4);
                const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + ti * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = t * 64 + ti * 16 + g * 4 + r;
                    float pv = 0.f, ds = 0.f;
                    if (i < p.Sq && j < p.Sk) {
                        float x = s[ti][r] * sc2;
                        if (p.mask_mode) x += mask_val(p.mask, p.mask_mode, b, i, j, p.Sq, p.Sk) * LOG2E;
                        pv = __builtin_amdgcn_exp2f(x - lv[r]);
                        float dm = 1.f;
                        if (DROP)
                            dm = drop_mult(p.drop_seed, p.drop_site, (((unsigned long long)(b + p.batch0) * p.H + h) * p.Sq + i) * (unsigned long long)p.Sk + j,
                                           drop_threshold(p.drop_p), 1.f / (1.f - p.drop_p));
                        ds = pv * fmaf(dp[ti][r] * dm, p.scale, -dv4[r]);
                        pv *= dm;      // dV sees the dropped probabilities
                    }
                    pr[ti][r] = pv;
                    s[ti][r] = ds;
                }
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

// ======================================================================================================================
// backward dK / dV, Q/dO-resident variant (the towers' unmasked self-attention; same eligibility as the K/V-resident kernels above).
// The tiled kernel gives every 64-key block its own workgroup, which stages all of Q and dO through LDS again (two barriers per 64
// queries) - five workgroups per head at 257 keys, the fifth for ONE key.  Here persistent 8-wave workgroups walk over (b, h) items with
// all of Q and dO of the head in LDS (2 x 288 rows x 256 B, staged once per item), wave w owns keys 32 w .. 32 w + 31 (two 16-key blocks,
// K / V fragments in registers), and nothing synchronises inside the query loop.  Keys 256.. (the 257th token) form one more 16-key block
// that is split over the QUERIES: wave t takes query tile t, the partial dK / dV rows are summed through LDS.
// ======================================================================================================================
template <typename T, int HDP, int KM = 0>
__device__ __forceinline__ void dkv_tile(LDS_AS const char* qt, LDS_AS const char* dot, LDS_AS const float* lse_t, LDS_AS const float* del_t,
                                         const int nti, const bool fast, const int i0, const int j, const int Sq, const int Sk, const float sc2,
                                         const float scale, const s16x8 (&kf)[Cfg<HDP>::KS], const s16x8 (&vf)[Cfg<HDP>::KS],
                                         f32x4 (&dkacc)[Cfg<HDP>::TD], f32x4 (&dvacc)[Cfg<HDP>::TD], const int lane) {
    using C = Cfg<HDP>;
    const int g = lane >> 4;
    f32x4 s[4], dp[4], pr[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        s[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ti < nti) {
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s[ti] = T16<T>::mfma(lds_row_frag<HDP, KM>(qt, ti * 16, ks, lane), kf[ks], s[ti]);
                dp[ti] = T16<T>::mfma(lds_row_frag<HDP, KM>(dot, ti * 16, ks, lane), vf[ks], dp[ti]);
            }
        }
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + ti * 16 + g * 4);
        const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + ti * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(fmaf(s[ti][r], sc2, -lv[r]));
            float ds = pv * fmaf(dp[ti][r], scale, -dv4[r]);
            if (!fast && !(i0 + ti * 16 + g * 4 + r < Sq && j < Sk)) { pv = 0.f; ds = 0.f; }
            pr[ti][r] = pv;
            s[ti][r] = ds;
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (2 * s2 >= nti) continue;
        const s16x8 pf = pack_pair<T>(pr[2 * s2], pr[2 * s2 + 1]);
        const s16x8 df = pack_pair<T>(s[2 * s2], s[2 * s2 + 1]);
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            dvacc[td] = T16<T>::mfma(lds_tr_frag<HDP, KM>(dot, td, s2, lane), pf, dvacc[td]);
            dkacc[td] = T16<T>::mfma(lds_tr_frag<HDP, KM>(qt, td, s2, lane), df, dkacc[td]);
        }
    }
}

template <typename T, int HDP>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv_res_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                  const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                  const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                                  const mico_attn_params p) {
    using C = Cfg<HDP>;
    constexpr int CPR = HDP / 8, QR = RES_KR;
    constexpr int NLD = (QR * CPR + 511) / 512;
    __shared__ __attribute__((aligned(16))) char smem[2 * QR * C::RS + 2 * QR * 4 + 8 * 2 * HDP * 4];
    LDS_AS char* qt = (LDS_AS char*)smem;
    LDS_AS char* dot = qt + QR * C::RS;
    LDS_AS float* lse_t = (LDS_AS float*)(dot + QR * C::RS);
    LDS_AS float* del_t = lse_t + QR;
    LDS_AS float* part = del_t + QR;   // [wave][dK | dV][HDP]: partial rows of key 256
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const int nitems = p.B * p.H;
    const int nt = (p.Sq + 63) / 64;
    const bool ragged = p.Sk > 256;
    const int nwg = gridDim.x;
    int item = (int)blockIdx.x, item_step = nwg;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {   // XCD-contiguous runs of items (see attn_fwd_res_kernel)
        const int per_wg = nitems / nwg;
        item = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * per_wg;
        item_step = 1;
    }
    const int item_end = item_step == 1 ? item + nitems / nwg : nitems;
    for (; item < item_end; item += item_step) {
        const int b = item / p.H, h = item - b * p.H;
        const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
        const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
        const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
        const T* dob = d_o + (int64_t)b * p.o_bs + h * p.hd;
        const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;
        // ---- Q, dO of the head -> registers -> LDS (zero beyond Sq / hd); lse * log2(e) and delta * scale next to them ----
        {
            s16x8 qr[NLD], dor[NLD];
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = c / CPR, ch = c - row * CPR;
                s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, d = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row < p.Sq && ch * 8 < p.hd) {
                    a = *(const s16x8*)(qb + (int64_t)row * p.q_rs + ch * 8);
                    d = *(const s16x8*)(dob + (int64_t)row * p.o_rs + ch * 8);
                }
                qr[it] = a;
                dor[it] = d;
            }
            float sl = 0.f, sd = 0.f;
            if (tid < QR && tid < p.Sq) { sl = lse[stat_base + tid] * LOG2E; sd = delta[stat_base + tid] * p.scale; }
            __syncthreads();   // the previous item's readers are done with the LDS images (and its key-256 partials are consumed)
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = c / CPR, ch = c - row * CPR;
                if (row < QR) {
                    const int off = row * C::RS + ((ch ^ ((row & 7) << 1)) << 4);
                    *(LDS_AS s16x8*)(qt + off) = qr[it];
                    *(LDS_AS s16x8*)(dot + off) = dor[it];
                }
            }
            if (tid < QR) { lse_t[tid] = sl; del_t[tid] = sd; }
        }
        __syncthreads();
        // this wave's keys: two 16-key blocks, one after the other (one block's accumulators and K / V fragments live at a time: with both
        // blocks in flight the kernel spilled 129 registers)
#pragma unroll 1
        for (int rb = 0; rb < 2; ++rb) {
            const int kb0 = wave * 32 + rb * 16;
            if (kb0 >= p.Sk) break;   // wave-uniform
            const int j = kb0 + (lane & 15);
            s16x8 kf[C::KS], vf[C::KS];
            f32x4 dkacc[C::TD], dvacc[C::TD];
            row_frags<T, HDP>(kf, kb, p.k_rs, j, p.Sk, p.hd, lane);
            row_frags<T, HDP>(vf, vb, p.v_rs, j, p.Sk, p.hd, lane);
#pragma unroll
            for (int t = 0; t < C::TD; ++t) {
                dkacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dvacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int t = 0; t < nt; ++t) {
                const int nti = (t == nt - 1 && (p.Sq & 63)) ? ((p.Sq - t * 64 + 15) >> 4) : 4;
                dkv_tile<T, HDP>(qt + t * C::TILE, dot + t * C::TILE, lse_t + t * 64, del_t + t * 64, nti, t * 64 + 64 <= p.Sq && kb0 + 16 <= p.Sk, t * 64, j,
                                 p.Sq, p.Sk, sc2, p.scale, kf, vf, dkacc, dvacc, lane);
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
        if (ragged) {
            // ---- keys 256..: wave t takes query tile t, partial rows through LDS ----
            f32x4 dkx[C::TD], dvx[C::TD];
#pragma unroll
            for (int t = 0; t < C::TD; ++t) {
                dkx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dvx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (wave < nt) {
                s16x8 kx[C::KS], vx[C::KS];
                const int j = 256 + (lane & 15);
                row_frags<T, HDP>(kx, kb, p.k_rs, j, p.Sk, p.hd, lane);
                row_frags<T, HDP>(vx, vb, p.v_rs, j, p.Sk, p.hd, lane);
                const int t = wave;
                const int nti = (t == nt - 1 && (p.Sq & 63)) ? ((p.Sq - t * 64 + 15) >> 4) : 4;
                dkv_tile<T, HDP>(qt + t * C::TILE, dot + t * C::TILE, lse_t + t * 64, del_t + t * 64, nti, false, t * 64, j, p.Sq, p.Sk, sc2, p.scale, kx,
                                 vx, dkx, dvx, lane);
            }
            // rows of keys 256 + (lane & 15): up to 16 keys beyond 256 (Sk <= 272); each lane parks its 4 head-dim values per tile
            LDS_AS float* pw = part + wave * 2 * HDP;
            if ((lane & 15) == 0) {
#pragma unroll
                for (int td = 0; td < C::TD; ++td)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pw[td * 16 + g * 4 + r] = dkx[td][r];
                        pw[HDP + td * 16 + g * 4 + r] = dvx[td][r];
                    }
            }
            __syncthreads();
            if (tid < 2 * HDP / 4) {
                const int which = tid / (HDP / 4), d = (tid - which * (HDP / 4)) * 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < 8; ++w) {
                    LDS_AS const float* pr = part + w * 2 * HDP + which * HDP + d;
                    acc[0] += pr[0]; acc[1] += pr[1]; acc[2] += pr[2]; acc[3] += pr[3];
                }
                if (d < p.hd) {
                    T* out = (which ? dv + (int64_t)b * p.v_bs + (int64_t)256 * p.v_rs : dk + (int64_t)b * p.k_bs + (int64_t)256 * p.k_rs) + h * p.hd + d;
                    *(s16x4*)out = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
                }
            }
        }
    }
}

// (The two round-3 experiment kernels below are compiled into the probe build only: `make -C mico_amd/csrc attnexp` ->
// tools/probes/bin/libmico_attnexp.so, -DMICO_ATTN_EXPERIMENTS=1, where MICO_ATTN_DKV=res32|stream selects them; the product library
// has neither them nor the switch.)
#ifdef MICO_ATTN_EXPERIMENTS
// ======================================================================================================================
// backward dK / dV on 32x32x16 MFMAs (round 3; hd 65..96, i.e. the g/14 towers).  Same residency as attn_bwd_dkv_res_kernel - persistent 8-wave
// workgroups, Q and dO of one (b, h) in LDS - but a wave owns 32 keys as ONE block: K / V fragments of the 32x32x16 operand layout in
// registers (2 x 6 k-steps of 16 head dims), query tiles of 32.  Per (32 queries x 32 keys): 6 + 6 MFMAs for S and dP, 6 + 6 for dV^T / dK^T
// = 24 MFMAs of 32 cycles on 12 ds_read_b128 + 24 transposing reads - half the LDS bytes per MFMA cycle of the 16x16 kernel (where every
// fragment fed one MFMA), a quarter of the MFMA instructions, and the softmax statistics are ONE value per lane and tile (lane = key column of
// the score tile would need a value per register; here S is computed as Q K^T with lane = key, registers = queries, so P / dS are already the
// B operands of dV^T = dO^T P and dK^T = Q^T dS, and lse / delta are read per register row ... see the layout notes below).
// Layout: S[q][key] = mfma32(A = Q rows (LDS, row = query), B = K fragment (lane l: key l & 31, dims 16 ks + 8 (l >> 5) ..)): lane l holds key
// l & 31 and queries i(r) = (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15.  Registers 0-7 / 8-15 are the B operands of two MFMAs over queries
// 0-15 / 16-31 of the tile: slot (h, e) <-> query 16 m + 4 h + (e & 3) + 8 (e >> 2); the A operands dO^T / Q^T follow the same numbering with
// two transposing reads per fragment (4 consecutive queries each, 8 rows apart).
// Key 256 (the 257th token) keeps the 16x16 path of the kernel above, split over the query tiles.
// ======================================================================================================================
template <typename T>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv_res32_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                    const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                    const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                                    const mico_attn_params p) {
    constexpr int HDP = 96, KM = 1;
    using C = Cfg<HDP>;
    constexpr int CPR = HDP / 8, QR = RES_KR, RS = C::RS;
    constexpr int NLD = (QR * CPR + 511) / 512;
    constexpr int NKS = HDP / 16;   // k-steps of 16 head dims (6)
    constexpr int NTD = HDP / 32;   // 32-wide head-dim tiles (3)
    __shared__ __attribute__((aligned(16))) char smem[2 * QR * RS + 2 * QR * 4 + 8 * 2 * HDP * 4];
    LDS_AS char* qt = (LDS_AS char*)smem;
    LDS_AS char* dot = qt + QR * RS;
    LDS_AS float* lse_t = (LDS_AS float*)(dot + QR * RS);
    LDS_AS float* del_t = lse_t + QR;
    LDS_AS float* part = del_t + QR;   // [wave][dK | dV][HDP]: partial rows of key 256
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const int nitems = p.B * p.H;
#ifndef MICO_DKV_ABL   // ablation builds (tools/probes): 1 = no tile loop (the load / stage / store skeleton), 2 = no global loads (compute only)
#define MICO_DKV_ABL 0
#endif
    const int nt64 = (p.Sq + 63) / 64, nt32 = MICO_DKV_ABL == 1 ? 0 : (p.Sq + 31) / 32;
    const bool ragged = p.Sk > 256 && MICO_DKV_ABL == 0;
    // per-lane LDS offsets.  Row fragments: row l & 31 of a 32-query tile, chunk 2 ks + (l >> 5).
    const int r31 = lane & 31, hh = lane >> 5, key_r = swz_key<KM>(r31);
    int rfo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) rfo[ks] = r31 * RS + (((ks * 2 + hh) ^ key_r) << 4);
    // Transposed fragments: 16-lane group u = l >> 4 reads the block [4 queries][16 head dims]: head dims 32 td + 16 (u & 1) + .., queries
    // 16 m + 4 (u >> 1) + (p >> 2) (first read) and 8 rows further (second read); lane p points at [row p >> 2][dims 4 (p & 3) ..].
    int tfo[NTD][2];
    {
        const int pp = lane & 15, u = lane >> 4;
        const int row0 = 4 * (u >> 1) + (pp >> 2);
#pragma unroll
        for (int td = 0; td < NTD; ++td) {
            const int ch = td * 4 + (u & 1) * 2 + ((pp & 3) >> 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = row0 + 8 * s;
                tfo[td][s] = row * RS + ((ch ^ swz_key<KM>(row)) << 4) + (pp & 1) * 8;
            }
        }
    }
    const int nwg = gridDim.x;
    int item = (int)blockIdx.x, item_step = nwg;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {   // XCD-contiguous runs of items (see attn_fwd_res_kernel)
        const int per_wg = nitems / nwg;
        item = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * per_wg;
        item_step = 1;
    }
    const int item_end = item_step == 1 ? item + nitems / nwg : nitems;
    for (; item < item_end; item += item_step) {
        const int b = item / p.H, h = item - b * p.H;
        const T* qb = q + (int64_t)b * p.q_bs + h * p.hd;
        const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
        const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
        const T* dob = d_o + (int64_t)b * p.o_bs + h * p.hd;
        const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;
        // ---- this wave's K / V fragments (keys 32 wave + (l & 31)), requested before the staging so that they travel with it ----
        const int kb0 = wave * 32;
        const bool own = kb0 < p.Sk && kb0 < 256;   // wave-uniform
        const int j = kb0 + r31;
        s16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 16 + hh * 8;
            s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
            if (MICO_DKV_ABL != 2 && own && j < p.Sk && d < p.hd) {
                a = *(const s16x8*)(kb + (int64_t)j * p.k_rs + d);
                c = *(const s16x8*)(vb + (int64_t)j * p.v_rs + d);
            }
            kf[ks] = a;
            vf[ks] = c;
        }
        // ---- Q, dO of the head -> registers -> LDS (zero beyond Sq / hd); lse * log2(e) and delta * scale next to them ----
        {
            s16x8 qr[NLD], dor[NLD];
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = c / CPR, ch = c - row * CPR;
                s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, d = {0, 0, 0, 0, 0, 0, 0, 0};
                if (MICO_DKV_ABL != 2 && row < p.Sq && ch * 8 < p.hd) {
                    a = *(const s16x8*)(qb + (int64_t)row * p.q_rs + ch * 8);
                    d = *(const s16x8*)(dob + (int64_t)row * p.o_rs + ch * 8);
                }
                qr[it] = a;
                dor[it] = d;
            }
            // rows beyond Sq: lse = +big makes their probabilities (and dS) exactly zero - no masking in the tile loop
            float sl = 1.0e30f, sd = 0.f;
            if (tid < QR && tid < p.Sq) { sl = lse[stat_base + tid] * LOG2E; sd = delta[stat_base + tid] * p.scale; }
            __syncthreads();   // the previous item's readers are done with the LDS images (and its key-256 partials are consumed)
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int c = it * 512 + tid;
                const int row = c / CPR, ch = c - row * CPR;
                if (row < QR) {
                    const int off = row * RS + ((ch ^ swz_key<KM>(row)) << 4);
                    *(LDS_AS s16x8*)(qt + off) = qr[it];
                    *(LDS_AS s16x8*)(dot + off) = dor[it];
                }
            }
            if (tid < QR) { lse_t[tid] = sl; del_t[tid] = sd; }
        }
        __syncthreads();
        if (own) {
            f32x16 dkacc[NTD], dvacc[NTD];
#pragma unroll
            for (int td = 0; td < NTD; ++td)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dkacc[td][r] = 0.f; dvacc[td][r] = 0.f; }
            for (int t = 0; t < nt32; ++t) {
                LDS_AS const char* qtile = qt + t * 32 * RS;
                LDS_AS const char* dtile = dot + t * 32 * RS;
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    s = mfma32<T>(*(LDS_AS const s16x8*)(qtile + rfo[ks]), kf[ks], s);
                    dp = mfma32<T>(*(LDS_AS const s16x8*)(dtile + rfo[ks]), vf[ks], dp);
                }
                // statistics of this lane's queries i(r) = (r & 3) + 8 (r >> 2) + 4 hh: two float4 reads each per half m (registers 8 m .. 8 m + 7
                // = the B operands of the MFMAs over queries 16 m .. 16 m + 15 of the tile)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float pr[8], ds[8];
#pragma unroll
                    for (int rq = 0; rq < 2; ++rq) {
                        const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + t * 32 + 16 * m + 8 * rq + 4 * hh);
                        const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + t * 32 + 16 * m + 8 * rq + 4 * hh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = m * 8 + rq * 4 + e;
                            const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lv[e]));
                            pr[rq * 4 + e] = pv;
                            ds[rq * 4 + e] = pv * fmaf(dp[r], p.scale, -dv4[e]);
                        }
                    }
                    const s16x8 pf = pack8<T>(pr);
                    const s16x8 df = pack8<T>(ds);
#pragma unroll
                    for (int td = 0; td < NTD; ++td) {
                        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dtile + m * 16 * RS + tfo[td][0]));
                        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dtile + m * 16 * RS + tfo[td][1]));
                        dvacc[td] = mfma32<T>((s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, pf, dvacc[td]);
                        const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(qtile + m * 16 * RS + tfo[td][0]));
                        const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(qtile + m * 16 * RS + tfo[td][1]));
                        dkacc[td] = mfma32<T>((s16x8){c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, df, dkacc[td]);
                    }
                }
            }
            if (j < p.Sk) {   // dK^T / dV^T tile td: this lane's key, head dims 32 td + 8 (r >> 2) + 4 hh + (r & 3)
                T* dkb = dk + (int64_t)b * p.k_bs + (int64_t)j * p.k_rs + h * p.hd;
                T* dvb = dv + (int64_t)b * p.v_bs + (int64_t)j * p.v_rs + h * p.hd;
#pragma unroll
                for (int td = 0; td < NTD; ++td)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int d = td * 32 + 8 * rq + 4 * hh;
                        if (d < p.hd) {
                            *(s16x4*)(dkb + d) = pack4<T>(dkacc[td][rq * 4], dkacc[td][rq * 4 + 1], dkacc[td][rq * 4 + 2], dkacc[td][rq * 4 + 3]);
                            *(s16x4*)(dvb + d) = pack4<T>(dvacc[td][rq * 4], dvacc[td][rq * 4 + 1], dvacc[td][rq * 4 + 2], dvacc[td][rq * 4 + 3]);
                        }
                    }
            }
        }
        if (ragged) {
            // ---- keys 256..: wave t takes query tile t (64 queries, 16x16 MFMAs), partial rows through LDS ----
            f32x4 dkx[C::TD], dvx[C::TD];
#pragma unroll
            for (int t = 0; t < C::TD; ++t) {
                dkx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dvx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (wave < nt64) {
                s16x8 kx[C::KS], vx[C::KS];
                const int jx = 256 + (lane & 15);
                row_frags<T, HDP>(kx, kb, p.k_rs, jx, p.Sk, p.hd, lane);
                row_frags<T, HDP>(vx, vb, p.v_rs, jx, p.Sk, p.hd, lane);
                const int t = wave;
                const int nti = (t == nt64 - 1 && (p.Sq & 63)) ? ((p.Sq - t * 64 + 15) >> 4) : 4;
                dkv_tile<T, HDP, KM>(qt + t * C::TILE, dot + t * C::TILE, lse_t + t * 64, del_t + t * 64, nti, false, t * 64, jx, p.Sq, p.Sk, sc2, p.scale, kx,
                                     vx, dkx, dvx, lane);
            }
            LDS_AS float* pw = part + wave * 2 * HDP;
            if ((lane & 15) == 0) {
#pragma unroll
                for (int td = 0; td < C::TD; ++td)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pw[td * 16 + g * 4 + r] = dkx[td][r];
                        pw[HDP + td * 16 + g * 4 + r] = dvx[td][r];
                    }
            }
            __syncthreads();
            if (tid < 2 * HDP / 4) {
                const int which = tid / (HDP / 4), d = (tid - which * (HDP / 4)) * 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < 8; ++w) {
                    LDS_AS const float* prow = part + w * 2 * HDP + which * HDP + d;
                    acc[0] += prow[0]; acc[1] += prow[1]; acc[2] += prow[2]; acc[3] += prow[3];
                }
                if (d < p.hd) {
                    T* out = (which ? dv + (int64_t)b * p.v_bs + (int64_t)256 * p.v_rs : dk + (int64_t)b * p.k_bs + (int64_t)256 * p.k_rs) + h * p.hd + d;
                    *(s16x4*)out = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
                }
            }
        }
    }
}

// ======================================================================================================================
// backward dK / dV, STREAMING variant (round 3; Sk == 257, hd 65..96: the g/14 towers).  Measured on the resident 32x32 kernel above
// (tools/probes/README.md, "attention: what bounds the resident kernels"): its load / stage / store skeleton alone takes 0.24 ms per launch
// (the HBM floor of the q, k, v, dO reads), its tile loop alone 0.36 ms, together 0.82 ms - the two add up because an item's operands are
// fetched, committed to LDS behind two barriers and only then multiplied, one item after the other on a CU whose LDS (147 KB of resident Q / dO)
// admits a single workgroup.  Here only K / V live per item (register fragments, 32 keys per wave); Q and dO arrive by LDS-DMA (16-byte pieces,
// lane-linear [rows][16 chunks] images, the chunk swizzle applied to the source address) in two HALVES per item - query tiles 0..3 into buffer A
// (128 rows), tiles 4..8 into buffer B (160 rows; together with the statistics 156 of the 160 KiB) - and while one half is multiplied the
// next one (of this item or of the next) is on its way: two barriers per item, none inside a half, so the waves of a SIMD drift apart and
// overlap their MFMA and VALU / LDS phases as in the resident kernels.  (First version: a 4-slot ring of single tiles with a barrier per tile -
// the eight waves then run every phase in lockstep and the tile arithmetic alone took 0.65 ms against 0.36 without barriers.)
// Key 256 (the 257th token) is a rank-one update - s = Q k256, dV256 = sum_q P dO, dK256 = sum_q dS Q - done on the VALU: per tile a wave
// takes 4 of the 32 queries (16 lanes x 8 head dims each), partial rows are folded through LDS once per item.
// ======================================================================================================================
template <typename T>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv_stream_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                     const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                     const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                                     const mico_attn_params p) {
    constexpr int HDP = 96, KM = 1, RS = 256, QR = RES_KR, ROWS_A = 128, ROWS_B = 160;
    constexpr int NKS = HDP / 16, NTD = HDP / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * (ROWS_A + ROWS_B) * RS + 2 * 2 * QR * 4 + 8 * 2 * HDP * 4 + 2 * 2 * 16 * 16];
    LDS_AS char* qA = (LDS_AS char*)smem;                // Q rows 0..127 | dO rows 0..127 | Q rows 128..287 | dO rows 128..287
    LDS_AS char* dA = qA + ROWS_A * RS;
    LDS_AS char* qB = dA + ROWS_A * RS;
    LDS_AS char* dB = qB + ROWS_B * RS;
    LDS_AS float* stats = (LDS_AS float*)(dB + ROWS_B * RS);   // [item parity][lse | delta][QR]
    LDS_AS float* part = stats + 2 * 2 * QR;                      // [wave][dK | dV][HDP]: partial rows of key 256
    LDS_AS char* kv256 = (LDS_AS char*)(part + 8 * 2 * HDP);      // [item parity][K | V][16 chunks of 16 bytes]: row 256 of K and V
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const int nitems = p.B * p.H;
    const int nt32 = (p.Sq + 31) / 32;
    // per-lane LDS offsets (see attn_bwd_dkv_res32_kernel)
    const int r31 = lane & 31, hh = lane >> 5, key_r = swz_key<KM>(r31);
    int rfo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) rfo[ks] = r31 * RS + (((ks * 2 + hh) ^ key_r) << 4);
    // transposed fragments: first read per 32-dim tile, the second one (8 query rows further: the swizzle key flips chunk bit 1) at + tf2
    int tfo[NTD], tf2;
    {
        const int pp = lane & 15, u = lane >> 4;
        const int row0 = 4 * (u >> 1) + (pp >> 2);
        const int cl = (u & 1) * 2 + ((pp & 3) >> 1);
#pragma unroll
        for (int td = 0; td < NTD; ++td) tfo[td] = row0 * RS + (((td * 4 + cl) ^ swz_key<KM>(row0)) << 4) + (pp & 1) * 8;
        tf2 = 8 * RS + ((((cl ^ swz_key<KM>(row0 + 8)) & 3) - ((cl ^ swz_key<KM>(row0)) & 3)) << 4);
    }
    // DMA source of this thread's piece of a 32-row slab: row tid >> 4, LDS chunk slot tid & 15 <- source chunk slot ^ key(row) (recomputed at
    // every issue: two registers less in the tile loop)
    const int qbytes = (int)(((int64_t)(p.Sq - 1) * p.q_rs + p.hd) * 2), obytes = (int)(((int64_t)(p.Sq - 1) * p.o_rs + p.hd) * 2);

    const int nwg = gridDim.x;
    int item0 = (int)blockIdx.x, item_step = nwg, n_my;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {   // XCD-contiguous runs of items (see attn_fwd_res_kernel)
        n_my = nitems / nwg;
        item0 = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * n_my;
        item_step = 1;
    } else {
        n_my = item0 < nitems ? (nitems - item0 + nwg - 1) / nwg : 0;
    }
    // half hf of local item li: rows 128 hf .. of Q and dO -> buffer A / B.  Thread tid moves pieces c = it * 512 + tid: row c >> 4, chunk slot c & 15
    auto issue_half = [&](int li, int hf) {
        if (MICO_DKV_ABL == 2 || MICO_DKV_ABL == 4) return;   // ablation: no DMA
        const int it_ = item0 + li * item_step;
        const int b_ = it_ / p.H, h_ = it_ - b_ * p.H;
        // (descriptor inputs through readfirstlane: under SGPR pressure hipcc parks them in VGPRs and then wraps every DMA in a waterfall
        // loop with a vmcnt(0) inside - cdna_hip_programming.md T20)
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(q + (int64_t)b_ * p.q_bs + h_ * p.hd), 0, __builtin_amdgcn_readfirstlane(qbytes), 0x00020000);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(d_o + (int64_t)b_ * p.o_bs + h_ * p.hd), 0, __builtin_amdgcn_readfirstlane(obytes), 0x00020000);
        LDS_AS char* qd = hf ? qB : qA;
        LDS_AS char* dd = hf ? dB : dA;
        const int drow = tid >> 4, dch = (tid & 15) ^ swz_key<KM>(drow);
        const bool dlive = dch * 8 < p.hd;
        const unsigned qoff0 = dlive ? (unsigned)(drow * p.q_rs * 2 + dch * 16) : 0xFFFFFFF0u;
        const unsigned ooff0 = dlive ? (unsigned)(drow * p.o_rs * 2 + dch * 16) : 0xFFFFFFF0u;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            if (it == 4 && !hf) break;
            const int rowg = hf * ROWS_A + it * 32;
            const unsigned qo = qoff0 == 0xFFFFFFF0u ? 0xFFFFFFF0u : qoff0 + (unsigned)(rowg * p.q_rs * 2);
            const unsigned oo = ooff0 == 0xFFFFFFF0u ? 0xFFFFFFF0u : ooff0 + (unsigned)(rowg * p.o_rs * 2);
            // The DMA as inline assembly: with the builtin, hipcc knows an LDS write is pending and waits vmcnt(0) in front of the first transposing
            // read that follows (it cannot tell that read's address from the DMA destination) - in the first tile of every half, i.e. it would
            // wait for the NEXT half to land before multiplying this one.  The asm form is invisible to its counters; completion is OUR
            // vmcnt(0) + barrier at the half boundaries (the compiler's own counted waits can only over-wait: the counter is in order).
            // M0 (the LDS destination) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
            lds_dma16(rq, (unsigned)(uintptr_t)(qd + (it * 512 + wave * 64) * 16), qo);
            lds_dma16(rd, (unsigned)(uintptr_t)(dd + (it * 512 + wave * 64) * 16), oo);
        }
    };
    if (n_my > 0) issue_half(0, 0);

    for (int li = 0; li < n_my; ++li) {
        const int item = item0 + li * item_step;
        const int b = item / p.H, h = item - b * p.H;
        const T* kb = k + (int64_t)b * p.k_bs + h * p.hd;
        const T* vb = v + (int64_t)b * p.v_bs + h * p.hd;
        const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;
        LDS_AS float* lse_t = stats + (li & 1) * 2 * QR;
        LDS_AS float* del_t = lse_t + QR;
        // ---- this wave's K / V fragments (keys 32 wave + (l & 31)), the item's row statistics, key 256's K / V chunk for the VALU update ----
        const int j = wave * 32 + r31;
        s16x8 kf[NKS], vf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 16 + hh * 8;
            s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
            if (d < p.hd) {
                a = *(const s16x8*)(kb + (int64_t)j * p.k_rs + d);
                c = *(const s16x8*)(vb + (int64_t)j * p.v_rs + d);
            }
            kf[ks] = a;
            vf[ks] = c;
        }
        const int xc = lane & 31, xq = lane >> 5;   // ragged-key update: 4 head dims 4 xc .. (24 of the 32 lanes of a query carry data), query slot
        LDS_AS char* kv_t = kv256 + (li & 1) * 512;
        if (tid < 32) {   // row 256 of K (threads 0..15) and of V (16..31), chunk tid & 15, parked in LDS for the per-tile VALU update
            s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((tid & 15) * 8 < p.hd) a = *(const s16x8*)((tid < 16 ? kb : vb) + (int64_t)256 * p.k_rs + (tid & 15) * 8);
            *(LDS_AS s16x8*)(kv_t + tid * 16) = a;
        }
        if (tid < QR) {
            // rows beyond Sq: lse = +big makes their probabilities (and dS) exactly zero - no masking in the tile loop
            float sl = 1.0e30f, sd = 0.f;
            if (tid < p.Sq) { sl = lse[stat_base + tid] * LOG2E; sd = delta[stat_base + tid] * p.scale; }
            lse_t[tid] = sl;
            del_t[tid] = sd;
        }
        f32x16 dkacc[NTD], dvacc[NTD];
#pragma unroll
        for (int td = 0; td < NTD; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dkacc[td][r] = 0.f; dvacc[td][r] = 0.f; }
        float dkx[4], dvx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { dkx[e] = 0.f; dvx[e] = 0.f; }

        for (int t = 0; t < nt32; ++t) {
            if (t == 0 || t == 4) {
                // this half has landed (every thread's pieces; the barrier publishes them) and every wave has left the OTHER buffer: refill it
                __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0), as a builtin: the compiler then knows its own K / V fragment loads have
                                                      // landed and does not wait for them again BEHIND the DMA it cannot see (that wait would
                                                      // cover the whole next half)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0) issue_half(li, 1);
                else if (li + 1 < n_my) issue_half(li + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MICO_DKV_ABL == 1) continue;   // ablation: no tile arithmetic
            // (tile offsets kept opaque scalars: hipcc otherwise peels the loop and holds per-lane fragment addresses of several tiles in registers)
            int qto = t < 4 ? t * 32 * RS : (2 * ROWS_A + (t - 4) * 32) * RS;
            int dto = t < 4 ? (ROWS_A + t * 32) * RS : (2 * ROWS_A + ROWS_B + (t - 4) * 32) * RS;
            asm volatile("" : "+s"(qto), "+s"(dto));
            LDS_AS const char* qtile = qA + qto;
            LDS_AS const char* dtile = qA + dto;
            // ---- key 256 on the VALU: queries t * 32 + 4 wave + 2 ps + xq (two passes), head dims 4 xc .. 4 xc + 3 ----
            if (MICO_DKV_ABL != 3 && MICO_DKV_ABL != 4) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int qrow = wave * 4 + ps * 2 + xq;
                    const int off = qrow * RS + (((xc >> 1) ^ swz_key<KM>(qrow)) << 4) + (xc & 1) * 8;
                    const f32x4 qv = unpack4<T>(*(LDS_AS const s16x4*)(qtile + off));
                    const f32x4 dov = unpack4<T>(*(LDS_AS const s16x4*)(dtile + off));
                    const f32x4 k4 = unpack4<T>(*(LDS_AS const s16x4*)(kv_t + xc * 8));
                    const f32x4 v4 = unpack4<T>(*(LDS_AS const s16x4*)(kv_t + 256 + xc * 8));
                    float sx = 0.f, dpx = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sx = fmaf(qv[e], k4[e], sx); dpx = fmaf(dov[e], v4[e], dpx); }
                    sx = xor16_sum(row16_sum(sx));      // (lanes 24..31 of a query read the zero chunks 12..15 of its row)
                    dpx = xor16_sum(row16_sum(dpx));
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sx, sc2, -lse_t[t * 32 + qrow]));
                    const float dsx = pv * fmaf(dpx, p.scale, -del_t[t * 32 + qrow]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dvx[e] = fmaf(pv, dov[e], dvx[e]); dkx[e] = fmaf(dsx, qv[e], dkx[e]); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (the VALU update first: its unpacked rows and the score tiles are not live together)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                s = mfma32<T>(*(LDS_AS const s16x8*)(qtile + rfo[ks]), kf[ks], s);
                dp = mfma32<T>(*(LDS_AS const s16x8*)(dtile + rfo[ks]), vf[ks], dp);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float pr[8], ds[8];
#pragma unroll
                for (int rq = 0; rq < 2; ++rq) {
                    const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + t * 32 + 16 * m + 8 * rq + 4 * hh);
                    const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + t * 32 + 16 * m + 8 * rq + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = m * 8 + rq * 4 + e;
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lv[e]));
                        pr[rq * 4 + e] = pv;
                        ds[rq * 4 + e] = pv * fmaf(dp[r], p.scale, -dv4[e]);
                    }
                }
                const s16x8 pf = pack8<T>(pr);
                const s16x8 df = pack8<T>(ds);
#pragma unroll
                for (int td = 0; td < NTD; ++td) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dtile + m * 16 * RS + tfo[td]));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dtile + m * 16 * RS + tfo[td] + tf2));
                    dvacc[td] = mfma32<T>((s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, pf, dvacc[td]);
                    const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(qtile + m * 16 * RS + tfo[td]));
                    const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(qtile + m * 16 * RS + tfo[td] + tf2));
                    dkacc[td] = mfma32<T>((s16x8){c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, df, dkacc[td]);
                }
            }
        }
        // ---- this wave's 32 keys: dK^T / dV^T tile td holds head dims 32 td + 8 (r >> 2) + 4 hh + (r & 3) of key j ----
        {
            T* dkb = dk + (int64_t)b * p.k_bs + (int64_t)j * p.k_rs + h * p.hd;
            T* dvb = dv + (int64_t)b * p.v_bs + (int64_t)j * p.v_rs + h * p.hd;
#pragma unroll
            for (int td = 0; td < NTD; ++td)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = td * 32 + 8 * rq + 4 * hh;
                    if (d < p.hd) {
                        *(s16x4*)(dkb + d) = pack4<T>(dkacc[td][rq * 4], dkacc[td][rq * 4 + 1], dkacc[td][rq * 4 + 2], dkacc[td][rq * 4 + 3]);
                        *(s16x4*)(dvb + d) = pack4<T>(dvacc[td][rq * 4], dvacc[td][rq * 4 + 1], dvacc[td][rq * 4 + 2], dvacc[td][rq * 4 + 3]);
                    }
                }
        }
        // ---- key 256: fold the two query slots of the wave, then the eight waves through LDS ----
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dkx[e] = xor32_sum(dkx[e]);
            dvx[e] = xor32_sum(dvx[e]);
        }
        if (lane < 24) {
            LDS_AS float* pw = part + wave * 2 * HDP + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pw[e] = dkx[e]; pw[HDP + e] = dvx[e]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * HDP / 4) {
            const int which = tid / (HDP / 4), d = (tid - which * (HDP / 4)) * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < 8; ++w) {
                LDS_AS const float* prow = part + w * 2 * HDP + which * HDP + d;
                acc[0] += prow[0]; acc[1] += prow[1]; acc[2] += prow[2]; acc[3] += prow[3];
            }
            if (d < p.hd) {
                T* out = (which ? dv + (int64_t)b * p.v_bs + (int64_t)256 * p.v_rs : dk + (int64_t)b * p.k_bs + (int64_t)256 * p.k_rs) + h * p.hd + d;
                *(s16x4*)out = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        // (the partial rows are rewritten at the END of the next item: at least one tile barrier lies in between)
    }
}

#endif   // MICO_ATTN_EXPERIMENTS
// ======================================================================================================================
// backward for SHORT query sequences (BERT: 77 text tokens against 1285 condition tokens, hd 64): dQ, dK and dV in ONE pass.
// The two tiled kernels above give every 64-query block and every 64-key block a workgroup of its own: at 77 x 1285 that is 2 + 21
// workgroups per (b, h), each of which stages the other operand through LDS again, and the scores, the softmax and - in training - the
// dropout decision of every (query, key) pair are computed twice (1.2 ms for the ITM triplet's 192 x 12 cross-attention heads).  Here ONE
// 4-wave workgroup owns a (b, h): Q and dO (<= 80 rows) sit in LDS for its whole life and the keys are dealt out in rounds of four 32-key
// strips, one per wave (K / V rows straight from global into registers, prefetched one round ahead).  Per round:
//   every wave, for its strip:
//     S = Q K^T, dP = dO V^T               acc lane = (key l & 15, queries 4 g + r)     [A = Q / dO rows from LDS, B = K / V in registers]
//     P, dS (softmax from the saved lse, mask, dropout regenerated by drop_hash_t) - branch free, every score produced ONCE
//     dV^T = dO^T P, dK^T = Q^T dS         complete for the strip (all queries are here) -> stored at once, no accumulators carried
//     K and dS^T -> the wave's LDS strip images
//   barrier; then wave w owns QUERY block w (and the d-slice w of the fifth block, queries 64..79):
//     dQ^T[d][query] += K^T[d][key] dS^T[key][query] over the round's 128 keys, both operands by transposing reads of the strip images
//   barrier (the strips are rewritten next round).
// dQ therefore needs 20 accumulator registers per lane instead of 80, no cross-wave reduction and no atomics (bit-reproducible).
// LDS: 2 x 12 KiB (Q, dO: 96 rows x 128 B, rows >= Sq zero) + 4 x (4 KiB K strip + 8 KiB dS strip) + statistics = 73.5 KiB -> two
// workgroups per CU.
// ======================================================================================================================
struct SqCfg {
    static constexpr int QMAX = 80, QR = 96;        // rows 80..95: the zero half of the third 32-deep reduction step over the queries
    static constexpr int RS = 128;                  // compact 64-wide rows; 16-byte chunk index XOR-ed by the key of sq_off()
    static constexpr int QT = QR * RS;
    static constexpr int KT = 32 * RS, DST = 32 * 256, WAVE = KT + DST;
    static constexpr int STAT = 2 * QR * 4;
    static constexpr int LDS = 2 * QT + STAT + 4 * WAVE;
};

// swizzle key of the compact rows: 2 ((row >> 1) & 3) + ((row >> 3) & 1).  Two rows share 256 bytes (all 64 banks); a ds_read_b128 is
// served 16 lanes at a time - eight consecutive rows x two adjacent chunks (lane groups g, g + 1), or sixteen rows x one chunk - and either
// set must spread over all 16 four-bank groups: rows of equal parity get keys whose upper two bits differ within 8 rows and whose low bit
// tells rows r and r + 8 apart (measured with (row >> 1) & 7: 0.4 conflict cycles per LDS cycle, SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int sq_off(int row, int ch) { return row * SqCfg::RS + ((ch ^ ((((row >> 1) & 3) << 1) | ((row >> 3) & 1))) << 4); }
__device__ __forceinline__ s16x8 sq_row_frag(LDS_AS const char* tile, int r0, int ks, int lane) {
    return *(LDS_AS const s16x8*)(tile + sq_off(r0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// the transposed fragment (see lds_tr_frag): output index d = td * 16 + (lane & 15), reduction slots over tile rows (2 s2 + r2) * 16 + 4 g + {0..3}
__device__ __forceinline__ s16x8 sq_tr_frag(LDS_AS const char* tile, int td, int s2, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int r_lo = (2 * s2) * 16 + g * 4 + (p >> 2);   // rows r_lo and r_lo + 16 have the same swizzle key
    const int off = sq_off(r_lo, td * 2 + ((p >> 1) & 1)) + (p & 1) * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(tile + off + 16 * SqCfg::RS));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <typename T, bool DROP, int MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_smallq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                 const T* __restrict__ o, const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                 T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, const mico_attn_params p) {
    using S = SqCfg;
    constexpr int HD = 64;
    __shared__ __attribute__((aligned(16))) char smem[S::LDS];
    LDS_AS char* qt = (LDS_AS char*)smem;
    LDS_AS char* dot = qt + S::QT;
    LDS_AS float* lse_t = (LDS_AS float*)(dot + S::QT);   // lse * log2(e)
    LDS_AS float* del_t = lse_t + S::QR;                    // delta * scale
    LDS_AS char* strips = (LDS_AS char*)smem + 2 * S::QT + S::STAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int b = blockIdx.y, h = blockIdx.x;
    LDS_AS char* kt = strips + wave * S::WAVE;
    LDS_AS char* dst = kt + S::KT;
    const T* qb = q + (int64_t)b * p.q_bs + h * HD;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const T* kb = k + (int64_t)kvb * p.k_bs + h * HD;
    const T* vb = v + (int64_t)kvb * p.v_bs + h * HD;
    const T* ob = o + (int64_t)b * p.o_bs + h * HD;
    const T* dob = d_o + (int64_t)b * p.o_bs + h * HD;
    const int64_t stat_base = ((int64_t)b * p.H + h) * p.Sq;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const int nstrips = (p.Sk + 31) >> 5, nrounds = (nstrips + 3) >> 2;

    // the first round's K / V rows (zero beyond Sk, also for a wave without a strip)
    s16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
        row_frags<T, HD>(kf[kbk], kb, p.k_rs, wave * 32 + kbk * 16 + (lane & 15), p.Sk, HD, lane);
        row_frags<T, HD>(vf[kbk], vb, p.v_rs, wave * 32 + kbk * 16 + (lane & 15), p.Sk, HD, lane);
    }
    // ---- Q, dO -> LDS (rows >= Sq zero); delta = rowsum(O * dO) from the dO chunk in hand ----
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int c = it * 256 + tid, row = c >> 3, ch = c & 7;   // 96 rows x 8 chunks = 768 = 3 x 256
        s16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0}, dv8 = qv, ov = qv;
        if (row < p.Sq) {
            qv = *(const s16x8*)(qb + (int64_t)row * p.q_rs + ch * 8);
            dv8 = *(const s16x8*)(dob + (int64_t)row * p.o_rs + ch * 8);
            ov = *(const s16x8*)(ob + (int64_t)row * p.o_rs + ch * 8);
        }
        *(LDS_AS s16x8*)(qt + sq_off(row, ch)) = qv;
        *(LDS_AS s16x8*)(dot + sq_off(row, ch)) = dv8;
        float a[8], c8[8], dl = 0.f;
        unpack8<T>(ov, a);
        unpack8<T>(dv8, c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += a[e] * c8[e];
        dl += __shfl_xor(dl, 1);
        dl += __shfl_xor(dl, 2);
        dl += __shfl_xor(dl, 4);
        if (ch == 0) {
            del_t[row] = dl * p.scale;
            lse_t[row] = row < p.Sq ? lse[stat_base + row] * LOG2E : 0.f;
        }
    }
    __syncthreads();

    f32x4 dqw[4], dq4 = {0.f, 0.f, 0.f, 0.f};     // dQ^T of query block `wave` (4 d-blocks) and of queries 64..79, d-block `wave`
#pragma unroll
    for (int td = 0; td < 4; ++td) dqw[td] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nti = (p.Sq + 15) >> 4;                  // 16-query blocks holding real queries (<= 5)
    // dropout: idx = ((b H + h) Sq + i) Sk + j, below 2^32 for every launch routed here (mico_attn_bwd) -> the incremental form of the hash
    const unsigned ibase = (unsigned)((((unsigned long long)(b + p.batch0) * p.H + h) * p.Sq) * (unsigned long long)p.Sk);
    const unsigned h0 = p.drop_seed ^ ((unsigned)p.drop_site * 0x9E3779B9u), tstep = (unsigned)p.Sk * 0x85EBCA6Bu;
    const unsigned thr = drop_threshold(p.drop_p);
    const float inv_keep = 1.f / (1.f - p.drop_p);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int rd = 0; rd < nrounds; ++rd) {
        const int k0 = (rd * 4 + wave) * 32;
        s16x8 nkf[2][2], nvf[2][2];
        if (rd + 1 < nrounds) {        // next round's rows fly under this round's work
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                row_frags<T, HD>(nkf[kbk], kb, p.k_rs, k0 + 128 + kbk * 16 + (lane & 15), p.Sk, HD, lane);
                row_frags<T, HD>(nvf[kbk], vb, p.v_rs, k0 + 128 + kbk * 16 + (lane & 15), p.Sk, HD, lane);
            }
        }
        if (k0 < p.Sk) {
            // K strip -> LDS image (the A operand of dQ^T = K^T dS^T is read from it transposed)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) *(LDS_AS s16x8*)(kt + sq_off(kbk * 16 + (lane & 15), ks * 4 + g)) = kf[kbk][ks];
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk) {
                __builtin_amdgcn_sched_barrier(0);              // one key block at a time: interleaving the two doubles the live score registers
                const int j = k0 + kbk * 16 + (lane & 15);      // this lane's key
                const bool jv = j < p.Sk;
                const int jc = jv ? j : p.Sk - 1;
                // (round 6) an accumulating launch (the shared own set's second / third reader) adds to what the rows hold: their old values are
                // requested HERE, a key block's worth of MFMA / softmax work ahead of the store that needs them - loaded next to the store, every key
                // block waited a memory round trip for them.  MASK == 0 only (cross-attention: the one place that accumulates; the masked
                // instantiations have no registers to spare)
                T* const dkb = dk + (int64_t)b * p.k_bs + (int64_t)jc * p.k_rs + h * HD;
                T* const dvb = dv + (int64_t)b * p.v_bs + (int64_t)jc * p.v_rs + h * HD;
                s16x4 okd[4], ovd[4];
                if (MASK == 0 && p.dkv_accumulate == 1) {      // (2: the A/B form of the launcher's MICO_SMALLQ_NOPF - loaded next to the store)
#pragma unroll
                    for (int td = 0; td < 4; ++td) {
                        okd[td] = *(const s16x4*)(dkb + td * 16 + g * 4);
                        ovd[td] = *(const s16x4*)(dvb + td * 16 + g * 4);
                    }
                }
                float mk1 = 0.f;
                if (MASK == 1) mk1 = p.mask[(int64_t)b * p.Sk + jc] * LOG2E;
                f32x4 s[5], dp[5];
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    s[ti] = zero4;
                    dp[ti] = zero4;
                    if (ti < nti) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            s[ti] = T16<T>::mfma(sq_row_frag(qt, ti * 16, ks, lane), kf[kbk][ks], s[ti]);
                            dp[ti] = T16<T>::mfma(sq_row_frag(dot, ti * 16, ks, lane), vf[kbk][ks], dp[ti]);
                        }
                    }
                }
                const unsigned tq = (ibase + (unsigned)j + (unsigned)(4 * g) * (unsigned)p.Sk) * 0x85EBCA6Bu;
                f32x4 pr[5];
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    const f32x4 lv = *(LDS_AS const f32x4*)(lse_t + ti * 16 + g * 4);
                    const f32x4 dv4 = *(LDS_AS const f32x4*)(del_t + ti * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = ti * 16 + g * 4 + r;
                        const bool ok = jv && i < p.Sq;
                        float x = s[ti][r] * sc2;
                        if (MASK == 1) x += mk1;
                        if (MASK == 2) x += p.mask[((int64_t)b * p.Sq + (i < p.Sq ? i : p.Sq - 1)) * p.Sk + jc] * LOG2E;
                        float pv = __builtin_amdgcn_exp2f(x - lv[r]);
                        float dm = 1.f;
                        if (DROP) dm = drop_hash_t(h0, tq + (unsigned)(ti * 16 + r) * tstep) >= thr ? inv_keep : 0.f;
                        const float ds = pv * fmaf(dp[ti][r] * dm, p.scale, -dv4[r]);
                        pv *= dm;      // dV sees the dropped probabilities
                        pr[ti][r] = ok ? pv : 0.f;
                        s[ti][r] = ok ? ds : 0.f;
                    }
                    // dS^T of this key block -> the wave's strip image [key][query] (256-byte rows, the tiles' swizzle)
                    const int row = kbk * 16 + (lane & 15), ch = ti * 2 + (g >> 1);
                    *(LDS_AS s16x4*)(dst + row * 256 + ((ch ^ swz_key<0>(row)) << 4) + (g & 1) * 8) = pack4<T>(s[ti][0], s[ti][1], s[ti][2], s[ti][3]);
                }
                // dV^T = dO^T P, dK^T = Q^T dS over all queries: complete for these 16 keys
                f32x4 dva[4], dka[4];
#pragma unroll
                for (int td = 0; td < 4; ++td) {
                    dva[td] = zero4;
                    dka[td] = zero4;
                }
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    if (2 * s2 >= nti) continue;
                    const s16x8 pf = pack_pair<T>(pr[2 * s2], s2 < 2 ? pr[2 * s2 + 1] : zero4);
                    const s16x8 df = pack_pair<T>(s[2 * s2], s2 < 2 ? s[2 * s2 + 1] : zero4);
#pragma unroll
                    for (int td = 0; td < 4; ++td) {
                        dva[td] = T16<T>::mfma(sq_tr_frag(dot, td, s2, lane), pf, dva[td]);
                        dka[td] = T16<T>::mfma(sq_tr_frag(qt, td, s2, lane), df, dka[td]);
                    }
                }
                if (jv) {
#pragma unroll
                    for (int td = 0; td < 4; ++td) {
                        const int d = td * 16 + g * 4;
                        if (p.dkv_accumulate) {   // (launch-uniform) the rows already hold another batch entry's gradient for the same K/V set
                            if (MASK == 0 && p.dkv_accumulate == 1) {
                                dka[td] += unpack4<T>(okd[td]);
                                dva[td] += unpack4<T>(ovd[td]);
                            } else {
                                dka[td] += unpack4<T>(*(const s16x4*)(dkb + d));
                                dva[td] += unpack4<T>(*(const s16x4*)(dvb + d));
                            }
                        }
                        *(s16x4*)(dkb + d) = pack4<T>(dka[td][0], dka[td][1], dka[td][2], dka[td][3]);
                        *(s16x4*)(dvb + d) = pack4<T>(dva[td][0], dva[td][1], dva[td][2], dva[td][3]);
                    }
                }
            }
        }
        __syncthreads();
        // dQ^T[d][query] += K^T[d][key] dS^T[key][query]: this wave's query block over the round's strips
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if ((rd * 4 + st) * 32 >= p.Sk) continue;
            LDS_AS const char* skt = strips + st * S::WAVE;
            LDS_AS const char* sdst = skt + S::KT;
            if (wave < nti) {
                const s16x8 dsf = lds_tr_frag<64, 0>(sdst, wave, 0, lane);
#pragma unroll
                for (int td = 0; td < 4; ++td) dqw[td] = T16<T>::mfma(sq_tr_frag(skt, td, 0, lane), dsf, dqw[td]);
            }
            if (nti == 5) dq4 = T16<T>::mfma(sq_tr_frag(skt, wave, 0, lane), lds_tr_frag<64, 0>(sdst, 4, 0, lane), dq4);
        }
        __syncthreads();
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[kbk][ks] = nkf[kbk][ks];
                vf[kbk][ks] = nvf[kbk][ks];
            }
    }

    // ---- dQ: lane = (query l & 15 of the block, d = 16 td + 4 g + r) ----
    {
        const int i = wave * 16 + (lane & 15);
        if (i < p.Sq) {
            T* dqb = dq + (int64_t)b * p.q_bs + (int64_t)i * p.q_rs + h * HD;
#pragma unroll
            for (int td = 0; td < 4; ++td) *(s16x4*)(dqb + td * 16 + g * 4) = pack4<T>(dqw[td][0], dqw[td][1], dqw[td][2], dqw[td][3]);
        }
        const int i4 = 64 + (lane & 15);
        if (i4 < p.Sq)
            *(s16x4*)(dq + (int64_t)b * p.q_bs + (int64_t)i4 * p.q_rs + h * HD + wave * 16 + g * 4) = pack4<T>(dq4[0], dq4[1], dq4[2], dq4[3]);
    }
}

// ======================================================================================================================
// backward for the towers' unmasked self-attention in ONE pass (round 4): dQ, dK and dV of a (b, h) item from scores
// that are computed once, every operand fetched from HBM once.  The two resident kernels above each stage half of the operands for the whole
// item, recompute S and dP (7 matmuls for 5, every exponential twice), read Q / K / V / dO twice and serialise fetch -> barrier -> arithmetic ->
// store per item on a CU whose LDS admits one workgroup.  Here a persistent 8-wave workgroup STREAMS the queries:
//   * wave w owns keys 32 w .. 32 w + 31 (two 16-key blocks): their V rows sit in registers, dK^T / dV^T accumulate in registers over the
//     whole item (no cross-wave reduction); K of the item is an LDS image (row fragments for S, transposed fragments for dQ);
//   * Q and dO arrive in chunks of 32 queries through a three-stage LDS ring: every thread carries one 16-byte piece of Q, dO and O per chunk
//     through registers (requested three chunks ahead of its use, committed one barrier later), and the sixteen threads of a row reduce
//     delta = rowsum(dO * O) on the way - the flat chunk sequence runs across item boundaries, so the next item's first chunks are in LDS
//     before the current item ends;
//   * phase 1 of a chunk (all waves): S = Q K^T, dP = dO V^T for the wave's keys, P and dS (lse saved by the forward), dV^T += dO^T P,
//     dK^T += Q^T dS (transposing reads of the ring tiles), dS -> a 16-bit LDS image [key][query] (double buffered);
//   * ONE barrier;
//   * phase 2 (the barrier's other side): the waves share the twelve (head-dim tile, query tile) pairs of dQ^T[d][query] = K^T[d][key] dS^T[key][query],
//     reduced over keys 0..255 with transposing reads of the K image and the dS image, and store the chunk's dQ rows; every thread also finishes
//     chunk G + 2 (below) and requests chunk G + 3.
//   * key 256 (the 257th token) never becomes a 17th key block - a block with one real row made the two waves that ran it the critical path
//     (+36 % against 256 keys).  It is a rank-one problem, and the loader threads hold exactly its operands: the thread of (query row, 16-byte
//     piece) multiplies its Q / dO pieces with the matching pieces of K[256] / V[256] (registers, per item), the row's sixteen threads reduce
//     s, dP and delta together, all of them know P and dS of (row, key 256), and dK[256] += dS Q, dV[256] += P dO accumulate in fp32 registers per
//     thread over the item (per item: lane-swap sums over the wave's four rows, one partial row per wave in LDS, eight of them added at the item's end); dS[.][256] rides in the stage's statistics and the
//     dQ waves apply it as a rank-one update from row 256 of the K image.
// Zero rows make masks unnecessary except in ragged key blocks: Q / dO rows beyond Sq and K / V rows beyond Sk are zero in LDS, so whatever
// P and dS hold there multiplies zeros (they stay finite: lse of a dead query is 0), and dead rows are never stored.
// LDS: K image 272 x 256 B, ring 3 x 16 KiB, dS 2 x 16 KiB, statistics, the key-256 row = 149.9 KiB.
// ======================================================================================================================
template <int HDP> struct OpCfg {
    using C = Cfg<HDP>;
    static constexpr int NST = 3;                         // ring stages
    static constexpr int QT = 32 * C::RS;                 // one [32][HDP] tile image
    static constexpr int STAGE = 2 * QT;                  // Q chunk | dO chunk
    static constexpr int KROWS = 272;
    static constexpr int KT = KROWS * C::RS;
    static constexpr int DSK = 32 * 64;                   // dS image of one 32-key step: [32 keys][32 queries] 16-bit, 64-byte rows
    static constexpr int DSB = 8 * DSK;                   // keys 0..255 (key 256 travels as one fp32 column, see below)
    static constexpr int SST = 96;                        // floats per stage: lse * log2 e | delta * scale | dS[.][256]
    static constexpr int STAT = NST * SST * 4;
    static constexpr int X16 = 8 * 2 * HDP * 4;           // [wave][dK | dV][HDP]: the key-256 row of the item, one partial row per wave
    static constexpr int LDS = KT + NST * STAGE + 2 * DSB + STAT + X16;
};
// an opaque copy of a lane-dependent value: address arithmetic derived from it is redone where it is used instead of being hoisted out of
// the item / chunk loops into registers that live (spilled) through them
__device__ __forceinline__ int launder(int x) { asm volatile("" : "+v"(x)); return x; }
// byte offset of the 8-byte piece `pc` (4 queries) of (key row kl of the step, query tile qt) in a dS step image: 64-byte rows of four 16-byte
// granules, the granule index XOR-ed with bits 2..3 of the row.  Rows r, r + 4, r + 8, r + 12 start in the same banks (64-byte stride), so both
// access patterns - the score tile's 8-byte writes (16 rows x 32 bytes per lane-group pair) and the transposing reads (4 rows x 32 bytes per lane
// group, four groups with different bits 2..3) - spread over all 64 banks (with a one-bit swap of the row halves SQ_LDS_BANK_CONFLICT was 0.16
// per LDS cycle: rows r and r + 8 collided)
__device__ __forceinline__ int ds_off(int kl, int qt, int pc) { return kl * 64 + ((((qt << 1) | (pc >> 1)) ^ ((kl >> 2) & 3)) << 4) + (pc & 1) * 8; }

// KMODE 2: Sk == 257 (the towers), 1: Sk == 256, 0: Sk < 256 (ragged / dead key blocks: masks and run-time loops, no 17th block)
template <typename T, int HDP, int KMODE>
__global__ __launch_bounds__(512, 1) void attn_bwd_onepass_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                  const T* __restrict__ o, const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                  T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, const mico_attn_params p) {
    using C = Cfg<HDP>;
    using O = OpCfg<HDP>;
    constexpr int CPR = HDP / 8;
    __shared__ __attribute__((aligned(16))) char smem[O::LDS];
    LDS_AS char* kimg = (LDS_AS char*)smem;
    LDS_AS char* ring = kimg + O::KT;
    LDS_AS char* dsb = ring + O::NST * O::STAGE;
    LDS_AS float* stat = (LDS_AS float*)(dsb + 2 * O::DSB);
    LDS_AS float* x16 = stat + O::NST * O::SST;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, l15 = lane & 15;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nitems = p.B * p.H;
    const int NC = (p.Sq + 31) >> 5;             // 32-query chunks per item
    const int nks = KMODE == 2 ? 9 : (KMODE == 1 ? 8 : (p.Sk + 31) >> 5);   // 32-key steps of the dQ reduction
    constexpr bool has16 = KMODE == 2, FULLK = KMODE != 0;
    // roles in phase 2: wave w < 4 owns head-dim tile w of dQ^T for both query tiles of the chunk, waves 4..7 one (tile 4 or 5, query tile) pair
    // each - three of the twelve pairs per SIMD (waves w and w + 4 share one)
    const int dq_td = wave < 4 ? wave : 4 + ((wave - 4) >> 1);
    const int dq_qt = wave < 4 ? -1 : (wave & 1);   // -1: both

    // XCD-contiguous runs of items (see attn_fwd_res_kernel)
    const int nwg = gridDim.x;
    int item0 = (int)blockIdx.x, item_step = nwg, n_my = (nitems - (int)blockIdx.x + nwg - 1) / nwg;
    if ((nwg & 7) == 0 && nitems % nwg == 0) {
        n_my = nitems / nwg;
        item0 = ((int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3)) * n_my;
        item_step = 1;
    }
    const int total = n_my * NC;                 // this workgroup's flat chunk sequence

    for (int c = tid; c < 2 * O::DSB / 16; c += 512) *(LDS_AS s16x8*)(dsb + c * 16) = zero8;   // rows of key blocks nobody owns stay zero

    // ---- the loader: thread = (chunk row tid >> 4, 16-byte slot tid & 15 of the 256-byte LDS row) ----
    // Q and dO go straight into the ring by LDS-DMA (no registers): a wave's instruction fills four consecutive rows, lane -> (row, slot), and
    // the slot holds the head-dim piece slot ^ key(row) - the tiles' swizzle, applied on the global side.  O (only needed for delta) and lse
    // travel through registers; they are requested AFTER the two DMA instructions, so the moment the compiler has waited for O the thread's
    // own DMA pieces have landed as well (loads return in order) and the next barrier publishes the stage.
    struct Piece { s16x8 o; float l; };
    // lch: the head-dim piece this thread's DMA lane fetches (it lands in slot tid & 15 of the row); ach = tid & 15: the piece it accumulates
    // (O, the K[256] / V[256] pieces, and the Q / dO pieces read back from slot ach ^ key(row) - written by a lane of the same wave's DMA
    // instructions, which have landed as a whole once the wave's wait for O is over).  One piece index per lane position in every row and wave.
    const int lrow = tid >> 4, lch = (tid & 15) ^ (((tid >> 4) & 7) << 1), ach = tid & 15;
    const bool lch_ok = lch * 8 < p.hd, ach_ok = ach * 8 < p.hd;
    const int qbytes = (int)(((int64_t)(p.Sq - 1) * p.q_rs + p.hd) * 2), obytes = (int)(((int64_t)(p.Sq - 1) * p.o_rs + p.hd) * 2);
    // the next chunk to request: chunk pf_c of item (pf_b, pf_h), pf_left items of this workgroup still to come after it
    int pf_c = 0, pf_left = n_my - 1, pf_b = item0 / p.H, pf_h = item0 - (item0 / p.H) * p.H;
    const unsigned ring_u = (unsigned)(uintptr_t)ring + (unsigned)(wave * 4 * C::RS);
    auto issue = [&](Piece& pc, int stage) {
        const int t = launder(tid);
        const int row = pf_c * 32 + (t >> 4);
        const bool rl = pf_left >= 0 && row < p.Sq, rp = rl && lch_ok, ra = rl && ach_ok;
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(q + (int64_t)pf_b * p.q_bs + pf_h * p.hd), 0, qbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(d_o + (int64_t)pf_b * p.o_bs + pf_h * p.hd), 0, obytes, 0x00020000);
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(o + (int64_t)pf_b * p.o_bs + pf_h * p.hd), 0, obytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + ((int64_t)pf_b * p.H + pf_h) * p.Sq), 0, p.Sq * 4, 0x00020000);
        const unsigned qo = rp ? (unsigned)(row * p.q_rs * 2 + lch * 16) : 0xFFFFFFF0u, oo = rp ? (unsigned)(row * p.o_rs * 2 + lch * 16) : 0xFFFFFFF0u;
        lds_dma16(rq, ring_u + stage * O::STAGE, qo);
        lds_dma16(rd, ring_u + stage * O::STAGE + O::QT, oo);
        pc.o = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(ro, ra ? (unsigned)(row * p.o_rs * 2 + ach * 16) : 0xFFFFFFF0u, 0, 0));
        pc.l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, rl ? (unsigned)(row * 4) : 0xFFFFFFF0u, 0, 0));   // (all 16 lanes of the row: one dword)
        if (++pf_c == NC) {
            pf_c = 0;
            if (--pf_left >= 0) {
                pf_h += item_step;
                if (pf_h >= p.H) { const int nb = pf_h / p.H; pf_b += nb; pf_h -= nb * p.H; }
            }
        }
    };
    // The commit side runs two chunks ahead of the arithmetic and keeps its own item state: chunk cm_c of item (cm_b, cm_h) is the next one to
    // finish; xk / xv = this thread's piece of K[256] / V[256] of that item; dk8 / dv8 = its share of dK[256] / dV[256].
    int cm_c = 0, cm_left = n_my - 1, cm_b = pf_b, cm_h = pf_h;
    s16x8 xk = zero8, xv = zero8;
    float dk8[8], dv8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dk8[e] = 0.f; dv8[e] = 0.f; }
    auto fetch256 = [&]() {
        if (KMODE != 2) return;
        const bool ok = cm_left >= 0 && ach_ok;
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)cm_b * p.k_bs + cm_h * p.hd), 0, (int)(((int64_t)(p.Sk - 1) * p.k_rs + p.hd) * 2), 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)cm_b * p.v_bs + cm_h * p.hd), 0, (int)(((int64_t)(p.Sk - 1) * p.v_rs + p.hd) * 2), 0x00020000);
        xk = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, ok ? (unsigned)(256 * p.k_rs * 2 + ach * 16) : 0xFFFFFFF0u, 0, 0));
        xv = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, ok ? (unsigned)(256 * p.v_rs * 2 + ach * 16) : 0xFFFFFFF0u, 0, 0));
    };
    // delta * scale and lse * log2(e) of the chunk in `stage` (whose Q / dO pieces this thread's own DMA has delivered), and key 256 against
    // the chunk's 32 queries
    auto commit = [&](const Piece& pc, int stage) {
        // the read-back of the thread's own pieces is ordered behind the arrival of O (hence of the older DMA) through a data dependence
        // the compiler can see: the LDS address passes through a statement that consumes O
        int t = tid;
        asm volatile("" : "+v"(t) : "v"(pc.o));
        const int roff = (t >> 4) * C::RS + (((t & 15) ^ (((t >> 4) & 7) << 1)) << 4);   // piece ach of the row
        const s16x8 dpc = *(LDS_AS const s16x8*)(ring + stage * O::STAGE + O::QT + roff);
        float a[8], c8[8], dl = 0.f;
        unpack8<T>(pc.o, a);
        unpack8<T>(dpc, c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += a[e] * c8[e];
        dl = row16_sum(dl) * p.scale;
        const float l2 = pc.l * LOG2E;
        LDS_AS float* ss = stat + stage * O::SST;
        if ((t & 15) == 0) { ss[32 + (t >> 4)] = dl; ss[t >> 4] = l2; }
        if (KMODE == 2) {
            const s16x8 qpc = *(LDS_AS const s16x8*)(ring + stage * O::STAGE + roff);
            float q8[8], k8[8], v8[8], sp = 0.f, dpp = 0.f;
            unpack8<T>(qpc, q8);
            unpack8<T>(xk, k8);
            unpack8<T>(xv, v8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sp += q8[e] * k8[e]; dpp += c8[e] * v8[e]; }
            sp = row16_sum(sp);
            dpp = row16_sum(dpp);
            const float pv = __builtin_amdgcn_exp2f(fmaf(sp, sc2, -l2));     // a dead row (Q = dO = 0, lse = 0): P = 1 times dO = 0, dS = 0
            const float ds = pv * fmaf(dpp, p.scale, -dl);
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk8[e] = fmaf(ds, q8[e], dk8[e]); dv8[e] = fmaf(pv, c8[e], dv8[e]); }
            if ((t & 15) == 1) ss[64 + (t >> 4)] = ds;
            if (++cm_c == NC) {   // the item's last chunk: the 32 rows' shares of its key-256 row meet in LDS (read out by item_end two barriers later)
                cm_c = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // the wave's four rows (lanes l, l + 16, l + 32, l + 48 hold the same piece)
                    dk8[e] = group_sum(dk8[e]);
                    dv8[e] = group_sum(dv8[e]);
                }
                if (t < 16 + (t & ~63) && ach_ok) {   // the first row of the wave writes the wave's partial row
                    LDS_AS float* xw = x16 + wave * 2 * HDP + ach * 8;
                    *(LDS_AS f32x4*)xw = (f32x4){dk8[0], dk8[1], dk8[2], dk8[3]};
                    *(LDS_AS f32x4*)(xw + 4) = (f32x4){dk8[4], dk8[5], dk8[6], dk8[7]};
                    *(LDS_AS f32x4*)(xw + HDP) = (f32x4){dv8[0], dv8[1], dv8[2], dv8[3]};
                    *(LDS_AS f32x4*)(xw + HDP + 4) = (f32x4){dv8[4], dv8[5], dv8[6], dv8[7]};
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { dk8[e] = 0.f; dv8[e] = 0.f; }
                if (--cm_left >= 0) {
                    cm_h += item_step;
                    if (cm_h >= p.H) { const int nb = cm_h / p.H; cm_b += nb; cm_h -= nb * p.H; }
                }
                fetch256();
            }
        }
    };

    // ---- per-item state ----
    s16x8 vf[2][C::KS];                          // V rows of this wave's keys (B operand of dP)
    f32x4 dkacc[2][C::TD], dvacc[2][C::TD];
    int b = 0, h = 0;

    // score tile (16 queries x 16 keys) -> P (into s), dS (into dp); packed fp32 arithmetic, two elements per instruction
    auto probs = [&](f32x4& s, f32x4& dp, const f32x4& lv, const f32x4& dl4) {
        const f32x2 sc = {sc2, sc2}, sl = {p.scale, p.scale};
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2 x = __builtin_elementwise_fma((f32x2){s[r], s[r + 1]}, sc, -(f32x2){lv[r], lv[r + 1]});
            const f32x2 pv = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            const f32x2 t = __builtin_elementwise_fma((f32x2){dp[r], dp[r + 1]}, sl, -(f32x2){dl4[r], dl4[r + 1]});
            const f32x2 ds = pv * t;
            s[r] = pv[0]; s[r + 1] = pv[1];
            dp[r] = ds[0]; dp[r + 1] = ds[1];
        }
    };

    // the general form (Sk < 256: ragged or dead key blocks, masks; the compiler's schedule)
    auto phase1_generic = [&](int stage, int dsbuf) {
        constexpr bool FULL = false;
        LDS_AS const char* qs = ring + stage * O::STAGE;
        LDS_AS const char* dos = qs + O::QT;
        LDS_AS const float* ss = stat + stage * O::SST;
        LDS_AS char* dsi = dsb + dsbuf * O::DSB + wave * O::DSK;
        const bool live0 = FULL || wave * 32 < p.Sk, live1 = FULL || wave * 32 + 16 < p.Sk;   // wave-uniform
        s16x4 plo[2], dlo[2];
        s16x8 pf[2], df[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // one operand family at a time (S, then dP): a query fragment is read once and serves both key blocks, and only one is live
            f32x4 s[2] = {zero4, zero4}, dp[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const s16x8 qf = lds_row_frag<HDP>(qs, qt * 16, ks, lane);
                if (live0) s[0] = T16<T>::mfma(qf, lds_row_frag<HDP>(kimg, wave * 32, ks, lane), s[0]);
                if (live1) s[1] = T16<T>::mfma(qf, lds_row_frag<HDP>(kimg, wave * 32 + 16, ks, lane), s[1]);
            }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const s16x8 dof = lds_row_frag<HDP>(dos, qt * 16, ks, lane);
                if (live0) dp[0] = T16<T>::mfma(dof, vf[0][ks], dp[0]);
                if (live1) dp[1] = T16<T>::mfma(dof, vf[1][ks], dp[1]);
            }
            const f32x4 lv = *(LDS_AS const f32x4*)(ss + qt * 16 + g * 4);
            const f32x4 dl4 = *(LDS_AS const f32x4*)(ss + 32 + qt * 16 + g * 4);
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise pulls the next query tile's fragment reads up here and spills)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int kb0 = wave * 32 + rb * 16;
                probs(s[rb], dp[rb], lv, dl4);
                if (!FULL && kb0 + l15 >= p.Sk) { s[rb] = zero4; dp[rb] = zero4; }
                const s16x4 p4 = pack4<T>(s[rb][0], s[rb][1], s[rb][2], s[rb][3]), d4 = pack4<T>(dp[rb][0], dp[rb][1], dp[rb][2], dp[rb][3]);
                if (rb == 0 ? live0 : live1) *(LDS_AS s16x4*)(dsi + ds_off(rb * 16 + l15, qt, g)) = d4;
                if (qt == 0) { plo[rb] = p4; dlo[rb] = d4; }
                else {
                    pf[rb] = (s16x8){plo[rb][0], plo[rb][1], plo[rb][2], plo[rb][3], p4[0], p4[1], p4[2], p4[3]};
                    df[rb] = (s16x8){dlo[rb][0], dlo[rb][1], dlo[rb][2], dlo[rb][3], d4[0], d4[1], d4[2], d4[3]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            const s16x8 ado = lds_tr_frag<HDP>(dos, td, 0, lane), aq = lds_tr_frag<HDP>(qs, td, 0, lane);
            if (live0) {
                dvacc[0][td] = T16<T>::mfma(ado, pf[0], dvacc[0][td]);
                dkacc[0][td] = T16<T>::mfma(aq, df[0], dkacc[0][td]);
            }
            if (live1) {
                dvacc[1][td] = T16<T>::mfma(ado, pf[1], dvacc[1][td]);
                dkacc[1][td] = T16<T>::mfma(aq, df[1], dkacc[1][td]);
            }
        }
    };

    // Sk >= 256: both key blocks of every wave hold 16 real keys - no masks, and the order of LDS reads and MFMAs is pinned (sched_barrier):
    // left alone hipcc reuses one fragment register set and waits for every read in front of the MFMA that consumes it (the function sits near
    // its register budget).  S for the four (query tile, key block) pairs advances one head-dim step at a time - four independent MFMA chains,
    // the K fragments of the next step requested under them - then dP (V fragments live in registers: no LDS), the softmax arithmetic, and the
    // dV / dK accumulation with the transposed fragments of the next head-dim tile in flight.
    auto phase1 = [&](int stage, int dsbuf) {
        if (!FULLK) { phase1_generic(stage, dsbuf); return; }
        LDS_AS const char* qs = ring + stage * O::STAGE;
        LDS_AS const char* dos = qs + O::QT;
        LDS_AS const float* ss = stat + stage * O::SST;
        LDS_AS char* dsi = dsb + dsbuf * O::DSB + wave * O::DSK;
        LDS_AS const char* kw = kimg + wave * 32 * C::RS;
        f32x4 sc[2][2] = {{zero4, zero4}, {zero4, zero4}}, dp[2][2] = {{zero4, zero4}, {zero4, zero4}};   // [query tile][key block]
        s16x8 qf[2][C::KS], dof[2][C::KS];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) qf[qt][ks] = lds_row_frag<HDP>(qs, qt * 16, ks, lane);
        s16x8 kc[2] = {lds_row_frag<HDP>(kw, 0, 0, lane), lds_row_frag<HDP>(kw, 16, 0, lane)};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            s16x8 kn[2] = {zero8, zero8};
            if (ks + 1 < C::KS) {
                kn[0] = lds_row_frag<HDP>(kw, 0, ks + 1, lane);
                kn[1] = lds_row_frag<HDP>(kw, 16, ks + 1, lane);
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dof[qt][ks] = lds_row_frag<HDP>(dos, qt * 16, ks, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) sc[qt][rb] = T16<T>::mfma(qf[qt][ks], kc[rb], sc[qt][rb]);
            __builtin_amdgcn_sched_barrier(0);
            kc[0] = kn[0];
            kc[1] = kn[1];
        }
        f32x4 lv[2], dl4[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            lv[qt] = *(LDS_AS const f32x4*)(ss + qt * 16 + g * 4);
            dl4[qt] = *(LDS_AS const f32x4*)(ss + 32 + qt * 16 + g * 4);
        }
        s16x8 ado = lds_tr_frag<HDP>(dos, 0, 0, lane), aq = lds_tr_frag<HDP>(qs, 0, 0, lane);   // head-dim tile 0 of the dV / dK step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) dp[qt][rb] = T16<T>::mfma(dof[qt][ks], vf[rb][ks], dp[qt][rb]);
        __builtin_amdgcn_sched_barrier(0);
        s16x8 pf[2], df[2];
        {
            s16x4 p4[2][2], d4[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    probs(sc[qt][rb], dp[qt][rb], lv[qt], dl4[qt]);
                    p4[qt][rb] = pack4<T>(sc[qt][rb][0], sc[qt][rb][1], sc[qt][rb][2], sc[qt][rb][3]);
                    d4[qt][rb] = pack4<T>(dp[qt][rb][0], dp[qt][rb][1], dp[qt][rb][2], dp[qt][rb][3]);
                    *(LDS_AS s16x4*)(dsi + ds_off(rb * 16 + l15, qt, g)) = d4[qt][rb];
                }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                pf[rb] = (s16x8){p4[0][rb][0], p4[0][rb][1], p4[0][rb][2], p4[0][rb][3], p4[1][rb][0], p4[1][rb][1], p4[1][rb][2], p4[1][rb][3]};
                df[rb] = (s16x8){d4[0][rb][0], d4[0][rb][1], d4[0][rb][2], d4[0][rb][3], d4[1][rb][0], d4[1][rb][1], d4[1][rb][2], d4[1][rb][3]};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int td = 0; td < C::TD; ++td) {
            s16x8 ado_n = zero8, aq_n = zero8;
            if (td + 1 < C::TD) {
                ado_n = lds_tr_frag<HDP>(dos, td + 1, 0, lane);
                aq_n = lds_tr_frag<HDP>(qs, td + 1, 0, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            dvacc[0][td] = T16<T>::mfma(ado, pf[0], dvacc[0][td]);
            dkacc[0][td] = T16<T>::mfma(aq, df[0], dkacc[0][td]);
            dvacc[1][td] = T16<T>::mfma(ado, pf[1], dvacc[1][td]);
            dkacc[1][td] = T16<T>::mfma(aq, df[1], dkacc[1][td]);
            __builtin_amdgcn_sched_barrier(0);
            ado = ado_n;
            aq = aq_n;
        }
    };

    // dQ^T tile td of the chunk's query tiles (QT: both, or one) over keys 0..255 (+ key 256 as a rank-one update); rows q0.. of item (b, h)
    auto phase2_dq = [&](auto both_tag, int dsbuf, int stage, int td, int qsel, int q0) {
        constexpr bool BOTH = decltype(both_tag)::value;
        const int lane = launder((int)threadIdx.x) & 63, l15 = lane & 15, g = lane >> 4;
        LDS_AS const char* dsi = dsb + dsbuf * O::DSB;
        const int rlo = g * 4 + (l15 >> 2);
        const int koff = rlo * C::RS + (((td * 2 + ((l15 >> 1) & 1)) ^ ((rlo & 7) << 1)) << 4) + (l15 & 1) * 8;   // rows rlo + 16 n share the key
        const int doff0 = ds_off(rlo, BOTH ? 0 : qsel, l15 & 3), doff1 = ds_off(rlo, 1, l15 & 3);
        f32x4 acc0 = zero4, acc1 = zero4;
        const int d = td * 16 + g * 4;
        // key 256's operands first: they are needed last and have the whole loop to arrive
        s16x4 k4r = {0, 0, 0, 0};
        float ds0 = 0.f, ds1 = 0.f;
        if (KMODE == 2) {
            k4r = *(LDS_AS const s16x4*)(kimg + 256 * C::RS + (d >> 3) * 16 + (d & 7) * 2);   // (row 256: swizzle key 0)
            LDS_AS const float* ss = stat + stage * O::SST + 64;
            ds0 = ss[(BOTH ? 0 : qsel * 16) + l15];
            if (BOTH) ds1 = ss[16 + l15];
        }
        struct Fr { s16x4 alo, ahi, b0lo, b0hi, b1lo, b1hi; };
        auto rd = [&](Fr& f, const int s) {
            f.alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(kimg + s * 32 * C::RS + koff));
            f.ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(kimg + (s * 32 + 16) * C::RS + koff));
            f.b0lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dsi + s * O::DSK + doff0));
            f.b0hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dsi + s * O::DSK + 1024 + doff0));
            if (BOTH) {
                f.b1lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dsi + s * O::DSK + doff1));
                f.b1hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(dsi + s * O::DSK + 1024 + doff1));
            }
        };
        auto mm = [&](const Fr& f) {
            const s16x8 a = {f.alo[0], f.alo[1], f.alo[2], f.alo[3], f.ahi[0], f.ahi[1], f.ahi[2], f.ahi[3]};
            const s16x8 b0 = {f.b0lo[0], f.b0lo[1], f.b0lo[2], f.b0lo[3], f.b0hi[0], f.b0hi[1], f.b0hi[2], f.b0hi[3]};
            acc0 = T16<T>::mfma(a, b0, acc0);
            if (BOTH) {
                const s16x8 b1 = {f.b1lo[0], f.b1lo[1], f.b1lo[2], f.b1lo[3], f.b1hi[0], f.b1hi[1], f.b1hi[2], f.b1hi[3]};
                acc1 = T16<T>::mfma(a, b1, acc1);
            }
        };
        if (KMODE != 0) {   // eight steps, the fragments of three steps in flight ahead of the MFMAs (the order is pinned: left alone the
                            // compiler reuses one register set and waits for every read)
            constexpr int AH = 3;
            Fr ringf[AH];
#pragma unroll
            for (int i = 0; i < AH; ++i) rd(ringf[i], i);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const Fr cur = ringf[s % AH];
                if (s + AH < 8) rd(ringf[s % AH], s + AH);
                __builtin_amdgcn_sched_barrier(0);
                mm(cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < nks; ++s) {
                Fr f;
                rd(f, s);
                mm(f);
            }
        }
        if (KMODE == 2) {   // key 256: dQ[q][d] += dS[q][256] K[256][d]
            const f32x4 k4 = unpack4<T>(k4r);
            acc0 += k4 * ds0;
            if (BOTH) acc1 += k4 * ds1;
        }
        if (d < p.hd) {
            T* dqb = dq + (int64_t)b * p.q_bs + h * p.hd + d;
            const int i0 = q0 + (BOTH ? 0 : qsel * 16) + l15, i1 = q0 + 16 + l15;
            if (i0 < p.Sq) *(s16x4*)(dqb + (int64_t)i0 * p.q_rs) = pack4<T>(acc0[0], acc0[1], acc0[2], acc0[3]);
            if (BOTH && i1 < p.Sq) *(s16x4*)(dqb + (int64_t)i1 * p.q_rs) = pack4<T>(acc1[0], acc1[1], acc1[2], acc1[3]);
        }
    };

    // K rows of an item (all threads), its V row 256.. and this wave's V rows: requested into registers ...
    constexpr int NKL = (O::KROWS * CPR + 511) / 512;
    s16x8 kr[NKL];
    auto item_fetch = [&](int item) {
        const int fb = item / p.H, fh = item - fb * p.H;
        const int tid = launder((int)threadIdx.x), l15 = tid & 15, g = (tid >> 4) & 3;
        __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(k + (int64_t)fb * p.k_bs + fh * p.hd), 0, (int)(((int64_t)(p.Sk - 1) * p.k_rs + p.hd) * 2), 0x00020000);
        __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(v + (int64_t)fb * p.v_bs + fh * p.hd), 0, (int)(((int64_t)(p.Sk - 1) * p.v_rs + p.hd) * 2), 0x00020000);
#pragma unroll
        for (int it = 0; it < NKL; ++it) {
            const int c = it * 512 + tid;
            const int row = c / CPR, ch = c - row * CPR;
            kr[it] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, (row < p.Sk && ch * 8 < p.hd) ? (unsigned)(row * p.k_rs * 2 + ch * 16) : 0xFFFFFFF0u, 0, 0));
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const int row = wave * 32 + rb * 16 + l15, d = ks * 32 + g * 8;
                vf[rb][ks] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (row < p.Sk && d < p.hd) ? (unsigned)(row * p.v_rs * 2 + d * 2) : 0xFFFFFFF0u, 0, 0));
            }
    };
    // ... and moved into LDS once every wave has left the previous item's phase 2
    auto item_start = [&](int item) {
        b = item / p.H;
        h = item - b * p.H;
        const int tid = launder((int)threadIdx.x);
        // the V rows requested with the K rows are consumed HERE as far as the compiler's wait counters go: first used inside the chunk loop, their
        // wait (vmcnt(0): they are the youngest loads of item_fetch) would sit in every chunk's phase 1 and drain the ring requests in flight
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(vf[rb][ks]));
        __syncthreads();   // the K image is free
#pragma unroll
        for (int it = 0; it < NKL; ++it) {
            const int c = it * 512 + tid;
            const int row = c / CPR, ch = c - row * CPR;
            if (row < O::KROWS) *(LDS_AS s16x8*)(kimg + row * C::RS + ((ch ^ ((row & 7) << 1)) << 4)) = kr[it];
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int td = 0; td < C::TD; ++td) {
                dkacc[rb][td] = zero4;
                dvacc[rb][td] = zero4;
            }
        __syncthreads();
    };

    auto item_end = [&]() {
        const int lane = launder((int)threadIdx.x) & 63, l15 = lane & 15, g = lane >> 4;
        // 16-byte stores: a lane holds 4 head dims of a key row per tile; v_permlane16_swap trades one tile's quartet with the neighbouring
        // lane group for the other tile's, so that even groups hold 8 consecutive head dims of tile td and odd groups of tile td + 1 - half the
        // store instructions (each touches 16 rows; the address path, not the bytes, is what they cost)
        static_assert(C::TD % 2 == 0, "tile pairs");
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int j = wave * 32 + rb * 16 + l15;
            T* dkb = dk + (int64_t)b * p.k_bs + (int64_t)j * p.k_rs + h * p.hd;
            T* dvb = dv + (int64_t)b * p.v_bs + (int64_t)j * p.v_rs + h * p.hd;
#pragma unroll
            for (int tp = 0; tp < C::TD; tp += 2) {
                const int d = (tp + (g & 1)) * 16 + (g >> 1) * 8;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x4 ta = m ? dvacc[rb][tp] : dkacc[rb][tp], tb = m ? dvacc[rb][tp + 1] : dkacc[rb][tp + 1];
                    u32x2 a = __builtin_bit_cast(u32x2, pack4<T>(ta[0], ta[1], ta[2], ta[3])), c = __builtin_bit_cast(u32x2, pack4<T>(tb[0], tb[1], tb[2], tb[3]));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a[0]), "+v"(c[0]));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a[1]), "+v"(c[1]));
                    if (j < p.Sk && d < p.hd) *(u32x4*)((m ? dvb : dkb) + d) = (u32x4){a[0], a[1], c[0], c[1]};
                }
            }
        }
        if (has16 && wave == 7 && lane < 2 * (HDP / 4)) {   // the key-256 row: every loader thread's share arrived two barriers ago
            const int which = lane / (HDP / 4), d = (lane - which * (HDP / 4)) * 4;
            f32x4 a0 = zero4;
#pragma unroll
            for (int w = 0; w < 8; ++w) a0 += *(LDS_AS const f32x4*)(x16 + w * 2 * HDP + which * HDP + d);
            if (d < p.hd) {
                T* out = (which ? dv + (int64_t)b * p.v_bs + (int64_t)256 * p.v_rs : dk + (int64_t)b * p.k_bs + (int64_t)256 * p.k_rs) + h * p.hd + d;
                *(s16x4*)out = pack4<T>(a0[0], a0[1], a0[2], a0[3]);
            }
        }
    };

    // ---- prologue: chunks 0, 1, 2 requested into the three ring stages, chunks 0 and 1 finished, the first item's K / V requested ----
    Piece pc;
    fetch256();
    issue(pc, 0);
    commit(pc, 0);
    issue(pc, 1);
    commit(pc, 1);
    issue(pc, 2);
    item_fetch(item0);

    int G = 0, st = 0;                           // flat chunk index of this workgroup and its ring stage (G % 3)
    PH_DECL;
    // one chunk: c of the item in hand; LAST: the item's last chunk (its phase 2 also stores dK / dV and requests the next item's K / V, so
    // that those registers are live from here to the next item_start only)
    auto chunk = [&](auto last_tag, const int c, const int next_item) {
        constexpr bool LAST = decltype(last_tag)::value;
#if MICO_ATTN_PRIO
        // waves w and w + 4 share a SIMD; left alone the first of them wins every issue tie, runs ahead and then idles at the barrier while its
        // partner finishes alone at single-wave latency: alternate the priority per chunk so that the pair advances together
        if (((G ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        phase1(st, G & 1);
        PH(2);
        __syncthreads();
        PH(3);
        // chunk G + 2 finished (requested one barrier ago; published by the next barrier), then chunk G + 3 into the stage chunk G just left
        // (order: the stores of this phase are OLDER than the requests that follow them - the wait for O at the next commit then passes stores that
        // had a whole chunk to retire; issued after the requests they sat between the wait and the loads it was for)
        commit(pc, st == 0 ? 2 : st - 1);
        PH(4);
        if (LAST) item_end();
        PH(7);
        if (dq_td < C::TD) {
            if (dq_qt < 0) phase2_dq(std::true_type{}, G & 1, st, dq_td, 0, c * 32);
            else phase2_dq(std::false_type{}, G & 1, st, dq_td, dq_qt, c * 32);
        }
        PH(6);
        issue(pc, st);
        if (LAST) item_fetch(next_item);
        PH(5);
        st = st == 2 ? 0 : st + 1;
        ++G;
    };
    for (int ii = 0; ii < n_my; ++ii) {
        item_start(item0 + ii * item_step);
        PH(0);
        for (int c = 0; c < NC - 1; ++c) chunk(std::false_type{}, c, 0);
        chunk(std::true_type{}, NC - 1, item0 + (ii + 1 < n_my ? ii + 1 : ii) * item_step);   // (the last item re-requests itself: no branch around the loads)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring requests beyond the last chunk (zero-fill DMA) must not outlive the workgroup's LDS
#ifdef MICO_ATTN_PHASES
    if (lane == 0 && blockIdx.x < 512) for (int e_ = 0; e_ < 8; ++e_) g_attn_phase[(blockIdx.x * 8 + wave) * 8 + e_] = ph_acc[e_];   // per wave
#endif
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
    MICO_CHECK(p->kv_batch_mod >= 0, "%s: kv_batch_mod must be >= 0", who);
    MICO_CHECK(p->batch0 >= 0, "%s: batch0 must be >= 0", who);
    return MICO_OK;
}

}  // namespace

#define ATTN_DISPATCH_HD(hd, ...)                        \
    do {                                                 \
        if ((hd) <= 64) { constexpr int HDP = 64; __VA_ARGS__; } \
        else if ((hd) <= 96) { constexpr int HDP = 96; __VA_ARGS__; } \
        else { constexpr int HDP = 128; __VA_ARGS__; }   \
    } while (0)

#ifdef MICO_ATTN_PHASES
extern "C" int mico_debug_attn_phases(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_phase), sizeof(unsigned long long) * n);
}
#endif

extern "C" int mico_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const mico_attn_params* p,
                             int dtype, void* stream) {
    MICO_CHECK(dtype_ok(dtype) && q && k && v && o && lse, "mico_attn_fwd: bad args");
    int rc = check_params(p, "mico_attn_fwd");
    if (rc) return rc;
    const dim3 block(256);
    hipStream_t st = (hipStream_t)stream;
    // two query blocks per wave beyond 64 query rows
    static const bool no_res = getenv("MICO_ATTN_NORES") != nullptr;   // A/B switch for tools/attn_bench.py, tools/probes/attn_phases.py
    // K/V-resident persistent kernel: unmasked self-attention of the ViT towers (hd 128 would spill next to the prefetch registers)
    if (!no_res && p->kv_batch_mod == 0 && p->mask_mode == 0 && p->drop_p <= 0.f && p->hd <= 96 && p->k_rs == p->v_rs && p->Sq > 128 && p->Sk <= 272 && p->Sq <= 256 + ResCfg<96>::NXMAX) {
        static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int nitems = p->B * p->H;
        const dim3 grid(nitems < n_cu ? nitems : n_cu);
        DISPATCH_T16(dtype, {
            if (p->hd <= 64) MICO_LAUNCH((attn_fwd_res_kernel<T, 64>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p);
            else MICO_LAUNCH((attn_fwd_res_kernel<T, 96>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p);
        });
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    // (also BERT's 77 text rows without dropout - evaluation, caption decoding: one workgroup with waves of 32 + 32 + 13 rows stages
    // each key tile once, two workgroups of 64 + 13 rows staged it twice and the second one ran its whole latency chain for 13 rows:
    // cross-attention forward 0.219 -> 0.164 ms.  With dropout the two-block variant needs 227 registers instead of 116 and is slower
    // in situ (134 vs 119 us), as is the same change in the dQ kernel (192 vs 124 registers: -10 %): occupancy wins there.)
    const bool two = p->Sq > 64 && p->drop_p <= 0.f;
    // (round 6) 65 .. 80 query rows with dropout - BERT's 77 text rows in training: five waves of 16 rows, one workgroup per (b, h) (see the kernel)
    static const bool no_five = getenv("MICO_ATTN_NOFIVE") != nullptr;   // A/B switch
    const bool five = !two && !no_five && p->drop_p > 0.f && p->Sq > 64 && p->Sq <= 80 && p->hd <= 64;
    const int qpw = two ? 128 : (five ? 80 : 64);
    const dim3 grid((p->Sq + qpw - 1) / qpw, p->H, p->B);
#define ATTN_FWD_LAUNCH(DROP, RB) MICO_LAUNCH((attn_fwd_kernel<T, HDP, DROP, RB>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p)
    if (five) {
        DISPATCH_T16(dtype, MICO_LAUNCH((attn_fwd_kernel<T, 64, true, 1, 5>), grid, dim3(320), 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, *p));
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, {
        if (two) ATTN_FWD_LAUNCH(false, 2);
        else if (p->drop_p > 0.f) ATTN_FWD_LAUNCH(true, 1);
        else ATTN_FWD_LAUNCH(false, 1);
    }));
#undef ATTN_FWD_LAUNCH
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
    // short query sequences at hd 64 (BERT's self- and cross-attention): the fused one-pass kernel
    static const bool no_smallq = getenv("MICO_ATTN_NOSMALLQ") != nullptr;   // A/B switch (tools/probes/drop_cost.py)
    if (!no_smallq && p->hd == 64 && p->Sq <= SqCfg::QMAX && (p->drop_p <= 0.f || (unsigned long long)(p->B + p->batch0) * p->H * p->Sq * p->Sk <= 0xFFFFFFFFull)) {
        const dim3 grid(p->H, p->B);
        static const bool no_pf = getenv("MICO_SMALLQ_NOPF") != nullptr;      // A/B switch: the accumulated rows' old values loaded next to their store (round 5's form)
        mico_attn_params pp = *p;
        pp.dkv_accumulate = p->dkv_accumulate ? (no_pf ? 2 : 1) : 0;
#define SQ_LAUNCH(DROP, MASK) MICO_LAUNCH((attn_bwd_smallq_kernel<T, DROP, MASK>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, (T*)dk, (T*)dv, pp)
        DISPATCH_T16(dtype, {
            if (p->drop_p > 0.f) { if (p->mask_mode == 0) SQ_LAUNCH(true, 0); else if (p->mask_mode == 1) SQ_LAUNCH(true, 1); else SQ_LAUNCH(true, 2); }
            else { if (p->mask_mode == 0) SQ_LAUNCH(false, 0); else if (p->mask_mode == 1) SQ_LAUNCH(false, 1); else SQ_LAUNCH(false, 2); }
        });
#undef SQ_LAUNCH
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    MICO_CHECK(!p->dkv_accumulate, "mico_attn_bwd: dkv_accumulate is implemented by the short-query kernel only (Sq <= %d at hd 64)", SqCfg::QMAX);
    const dim3 gq((p->Sq + 63) / 64, p->H, p->B), gk((p->Sk + 63) / 64, p->H, p->B);
    static const bool no_res = getenv("MICO_ATTN_NORES") != nullptr;
    const bool res = !no_res && p->kv_batch_mod == 0 && p->mask_mode == 0 && p->drop_p <= 0.f && p->hd <= 96 && p->k_rs == p->v_rs && p->Sq > 128 && p->Sk <= 272 &&
                     p->Sq <= 256 + ResCfg<96>::NXMAX;
    // the towers' self-attention (g/14: 257 tokens, hd 88): one pass, scores computed once (MICO_ATTN_NOONEPASS=1: the two resident kernels, for A/B runs)
    static const bool no_onepass = getenv("MICO_ATTN_NOONEPASS") != nullptr;
    if (res && !no_onepass && p->Sk <= 257) {
        static const int n_cu1 = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int nitems = p->B * p->H;
        const dim3 grid(nitems < n_cu1 ? nitems : n_cu1);
#define OP_LAUNCH(HDP, KMODE) MICO_LAUNCH((attn_bwd_onepass_kernel<T, HDP, KMODE>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, (T*)dk, (T*)dv, *p)
        // (hd <= 64: the EVA02 towers - L/14 at 257 tokens, B/16 at 197; the dQ waves of head-dim tiles 4 and 5 idle there)
        DISPATCH_T16(dtype, {
            if (p->hd <= 64) { if (p->Sk == 257) OP_LAUNCH(64, 2); else if (p->Sk == 256) OP_LAUNCH(64, 1); else OP_LAUNCH(64, 0); }
            else { if (p->Sk == 257) OP_LAUNCH(96, 2); else if (p->Sk == 256) OP_LAUNCH(96, 1); else OP_LAUNCH(96, 0); }
        });
#undef OP_LAUNCH
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    if (res) {
        static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int nitems = p->B * p->H;
        const dim3 grid(nitems < n_cu ? nitems : n_cu);
        DISPATCH_T16(dtype, {
            if (p->hd <= 64) MICO_LAUNCH((attn_bwd_dq_res_kernel<T, 64>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p);
            else MICO_LAUNCH((attn_bwd_dq_res_kernel<T, 96>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p);
        });
    } else
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, { if (p->drop_p > 0.f) MICO_LAUNCH((attn_bwd_dq_kernel<T, HDP, true>), gq, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p); else MICO_LAUNCH((attn_bwd_dq_kernel<T, HDP, false>), gq, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, (T*)dq, delta, *p); }));
    MICO_LAUNCH_CHECK();
    static const bool no_dkv_res = getenv("MICO_ATTN_NODKVRES") != nullptr;   // A/B switch (tools/attn_bench.py)
    if (res && !no_dkv_res && p->Sk <= 257 && p->hd > 64) {   // (keys beyond 256: one ragged key row is what the partial-row merge writes)
        static const int n_cu2 = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        const int nitems = p->B * p->H;
        const dim3 grid(nitems < n_cu2 ? nitems : n_cu2);
        // Round 3 experiments, selectable for A/B runs (tools/attn_bench.py; numbers in tools/probes/README.md): MICO_ATTN_DKV=res32 - the same
        // residency on 32x32x16 MFMAs; MICO_ATTN_DKV=stream - Q / dO by LDS-DMA in two halves per item, double buffered.  Both pass the kernel
        // tests; neither beats the 16x16 kernel by more than 2-3 % (1.27-1.29 vs 1.31 ms backward at 320 frames), so it keeps the launch.
#ifdef MICO_ATTN_EXPERIMENTS
        static const char* dkv_env = getenv("MICO_ATTN_DKV");
        static const int dkv_mode = !dkv_env ? 0 : (dkv_env[0] == 'r' ? 1 : (dkv_env[0] == 's' ? 2 : 0));
        if (dkv_mode == 2 && p->Sk == 257 && p->Sq > 128 && p->Sq <= RES_KR)
            DISPATCH_T16(dtype, MICO_LAUNCH((attn_bwd_dkv_stream_kernel<T>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta,
                                            (T*)dk, (T*)dv, *p));
        else if (dkv_mode == 1) DISPATCH_T16(dtype, MICO_LAUNCH((attn_bwd_dkv_res32_kernel<T>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta,
                                                                (T*)dk, (T*)dv, *p));
        else
#endif
        DISPATCH_T16(dtype, MICO_LAUNCH((attn_bwd_dkv_res_kernel<T, 96>), grid, dim3(512), 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta,
                                             (T*)dk, (T*)dv, *p));
        MICO_LAUNCH_CHECK();
        return MICO_OK;
    }
    DISPATCH_T16(dtype, ATTN_DISPATCH_HD(p->hd, { if (p->drop_p > 0.f) MICO_LAUNCH((attn_bwd_dkv_kernel<T, HDP, true>), gk, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, *p); else MICO_LAUNCH((attn_bwd_dkv_kernel<T, HDP, false>), gk, block, 0, st, (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, *p); }));
    MICO_LAUNCH_CHECK();
    return MICO_OK;
}
