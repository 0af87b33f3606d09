// Split-KV decode attention core (flash-decoding) for gfx950, shared by the op-level
// sdpa_causal_fixed_cache (ops_attention.hip) and the fused decode step (engine.hip).
//
// One workgroup = 4 waves owns one KV head and one contiguous chunk of cached positions and
// serves all G = Hq/Hkv query heads of that KV head from a single pass over K and V (the
// reference stores the cache GQA-expanded and re-reads it per query head: K17/K10 in SURVEY.md).
// K/V rows go straight from HBM to VGPRs, 16 bytes per lane: D/8 lanes cover one row, so one
// wave-instruction fetches 64/(D/8) consecutive positions = 1 KiB contiguous.  Each lane-group keeps
// its own online-softmax state (m, l, o[8]) per query head; groups and waves are merged once at
// the end through LDS, and chunks are merged by a second tiny kernel (or by the consumer).
#pragma once

#include "pgk_device.hip.h"

// diagnostic builds (-DPGK_PHASE_STAMPS): stamps inside the walk, parked like TLStamp::phase (engine_common.hip.h)
#ifdef PGK_PHASE_STAMPS
static __device__ unsigned long long* g_phase_tl;
#define PGK_PHASE(i)                                                                                              \
    do {                                                                                                          \
        unsigned long long* p_ = g_phase_tl;                                                                      \
        const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                      \
        if (p_ && threadIdx.x == 0 && wg_ < 256u) p_[2 * (256 + 4 * wg_) + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PGK_PHASE(i)
#endif

namespace pgk {

template <class T> struct KVLoad;
template <> struct KVLoad<bf16> {
    __device__ static __forceinline__ void load8(const bf16* p, float (&f)[8]) {
        Vec<bf16> v; v.load(p); v.to_float(f);
    }
};
template <> struct KVLoad<f16> {
    __device__ static __forceinline__ void load8(const f16* p, float (&f)[8]) {
        Vec<f16> v; v.load(p); v.to_float(f);
    }
};
template <> struct KVLoad<float> {
    __device__ static __forceinline__ void load8(const float* p, float (&f)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};

// Partial record per (query head, chunk): [m, l, o[0..D-1]] fp32, o un-normalised (sum p*v).
template <int D> constexpr int partial_stride() { return D + 2; }

// Running state of one lane-group for G heads.
template <int G>
struct DecodeState {
    float m[G], l[G], o[G][8];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
        }
    }
    // fold one position: s[g] = scaled score (already reduced over the row), v = this lane's 8 values
    __device__ __forceinline__ void update(const float (&s)[G], const float (&v)[8]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mn = fmaxf(m[g], s[g]);
            const float alpha = __expf(m[g] - mn);  // m = -inf first time: exp(-inf) = 0
            const float p = __expf(s[g] - mn);
            l[g] = l[g] * alpha + p;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = fmaf(p, v[j], o[g][j] * alpha);
            m[g] = mn;
        }
    }
};

// Walk positions [c0, c1) of one KV head's cache rows (row stride D elements).
// qf[g][8]: this lane's slice of the G pre-scaled queries.  Wave `wid` of 4 takes every 4th
// position-group.  LPR = D/8 lanes per row.
template <class T, int D, int G, int NWV = 4>
__device__ __forceinline__ void decode_walk(const T* kbase, const T* vbase, int c0, int c1, const float (&qf)[G][8],
                                            int lane, int wid, DecodeState<G>& st) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, U = 4, STRIDE = NWV * PPW;
    const int grp = lane / LPR, sub = lane % LPR;
    // U position-groups per trip: all 2*U 16-byte loads are issued before the first is consumed, so a
    // short chunk costs ONE memory round trip instead of one per position.
    for (int p0 = c0 + wid * PPW; p0 < c1; p0 += U * STRIDE) {
        float kf[U][8], vf[U][8];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pos = p0 + u * STRIDE + grp;
            valid[u] = pos < c1;
            const int pc = valid[u] ? pos : c1 - 1;  // clamp: loads stay in bounds, result discarded
            KVLoad<T>::load8(kbase + (size_t)pc * D + sub * 8, kf[u]);
            KVLoad<T>::load8(vbase + (size_t)pc * D + sub * 8, vf[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) d = fmaf(qf[g][j], kf[u][j], d);
                d = group_sum<LPR>(d);
                s[g] = d;
            }
            if (valid[u]) st.update(s, vf[u]);  // lane-group uniform
        }
    }
}

// Two-phase form for bf16 caches: issue the raw 16-byte loads of U position-groups (addresses clamped to
// `clamp_max`, so they can be issued before the context length is known), consume them later.
template <int U> struct KVBatch { uint4 k[U], v[U]; };

template <int D, int U, int NWV = 4>
__device__ __forceinline__ void kv_issue(KVBatch<U>& kb, const bf16* kbase, const bf16* vbase, int p0, int clamp_max, int lane) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, STRIDE = NWV * PPW;
    const int grp = lane / LPR, sub = lane % LPR;
    // every K row first, then the V rows: vector memory returns in issue order, so the scores (K only) run while the V
    // half of the batch is still arriving, instead of starting when the last byte of the batch has landed
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int pc = min(p0 + u * STRIDE + grp, clamp_max);
        kb.k[u] = *reinterpret_cast<const uint4*>(kbase + (size_t)pc * D + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int pc = min(p0 + u * STRIDE + grp, clamp_max);
        kb.v[u] = *reinterpret_cast<const uint4*>(vbase + (size_t)pc * D + sub * 8);
    }
}

// bf16 pairs for v_dot2_f32_bf16 (products exact in fp32, fp32 accumulate)
typedef __bf16 attn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float attn_dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a), __builtin_bit_cast(attn_bf16x2, b), acc, false);
}

// Scores and P.V on the packed bf16 dot instruction, straight from the cache's bf16 words - no widening of K or V:
//   score  = 4 x dot2(k pair, q pair)           q rounded to bf16 once per step (the model's dtype; flash prefill does the same)
//   o[j]  += dot2((v_u[j], v_u'[j]), (p_u, p_u'))  two cached positions per instruction: the V words of two rows are
//            interleaved by v_perm_b32 (shared by the G heads), the two probabilities packed to bf16
// 28 VALU instructions per position and lane-group at G = 2 where the fp32 version (8 + 8 widenings, 16 + 16 FMAs ...) took 75:
// the kernel's attention phase went from 1.96 to ~1 us (in-kernel stamps, tools/phase_stamps.py).  Position-groups that lie
// wholly beyond the context (the first batch is issued for U0 groups before the length is known) are skipped by a
// wave-uniform branch.  Batch-wise softmax as before: all scores first, ONE running-max update and one rescale per batch.
template <int D, int G, int U, int NWV = 4>
__device__ __forceinline__ void kv_consume(const KVBatch<U>& kb, int p0, int c1, const uint4 (&qb)[G], int lane,
                                           DecodeState<G>& st) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, STRIDE = NWV * PPW;
    static_assert(U % 2 == 0, "positions are consumed in pairs");
    const int grp = lane / LPR;
    float s[U][G];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool live = p0 + u * STRIDE < c1;            // wave-uniform: some lane-group of this wave has a valid position
#pragma unroll
        for (int g = 0; g < G; ++g) s[u][g] = -INFINITY;
        if (live) {
            const bool valid = p0 + u * STRIDE + grp < c1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = attn_dot2(kb.k[u].x, qb[g].x, 0.f);
                d = attn_dot2(kb.k[u].y, qb[g].y, d);
                d = attn_dot2(kb.k[u].z, qb[g].z, d);
                d = attn_dot2(kb.k[u].w, qb[g].w, d);
                d = group_sum<LPR>(d);
                s[u][g] = valid ? d : -INFINITY;
            }
        }
    }
    PGK_PHASE(5);
    float mx[G], alpha[G], lsum[G];
    bool any[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mx[g] = st.m[g];
#pragma unroll
        for (int u = 0; u < U; ++u) mx[g] = fmaxf(mx[g], s[u][g]);
        any[g] = mx[g] != -INFINITY;                        // nothing valid yet for this lane-group: leave the state alone
        alpha[g] = any[g] ? __expf(st.m[g] - mx[g]) : 1.f;
        lsum[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) st.o[g][j] *= alpha[g];
    }
    PGK_PHASE(6);
#pragma unroll
    for (int u = 0; u < U; u += 2) {
        if (p0 + u * STRIDE < c1) {                         // wave-uniform
            // (v_u[j], v_u'[j]) pairs: word i of a row holds dims 2i (low half) and 2i + 1 (high half)
            const uint32_t a[4] = {kb.v[u].x, kb.v[u].y, kb.v[u].z, kb.v[u].w};
            const uint32_t b[4] = {kb.v[u + 1].x, kb.v[u + 1].y, kb.v[u + 1].z, kb.v[u + 1].w};
            uint32_t ve[4], vo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ve[i] = __builtin_amdgcn_perm(b[i], a[i], 0x05040100u);   // {a.lo16, b.lo16}
                vo[i] = __builtin_amdgcn_perm(b[i], a[i], 0x07060302u);   // {a.hi16, b.hi16}
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float pa = any[g] ? __expf(s[u][g] - mx[g]) : 0.f;        // exp(-inf) = 0 for masked positions
                const float pb = any[g] ? __expf(s[u + 1][g] - mx[g]) : 0.f;
                const uint32_t pk = pack_bf16x2(pa, pb);
                // the denominator sums the ROUNDED probabilities, the ones the numerator uses
                lsum[g] += __uint_as_float(pk << 16) + __uint_as_float(pk & 0xFFFF0000u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st.o[g][2 * i] = attn_dot2(ve[i], pk, st.o[g][2 * i]);
                    st.o[g][2 * i + 1] = attn_dot2(vo[i], pk, st.o[g][2 * i + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (any[g]) {
            st.l[g] = st.l[g] * alpha[g] + lsum[g];
            st.m[g] = mx[g];
        }
    }
}

// decode_walk for bf16 caches in trips of U position-groups through kv_issue / kv_consume: all 2 U loads of a trip in
// flight together, ONE softmax rescale per trip.  U = 8 covers 128 positions (D = 128) per trip, so a split-KV slice of
// the usual 64-100 positions is one memory round trip (the per-position walk above made two, the second nearly empty).
template <int D, int G, int U = 8, int NWV = 4>
__device__ __forceinline__ void decode_walk_trips(const bf16* kbase, const bf16* vbase, int c0, int c1, const uint4 (&qb)[G],
                                                  int lane, int wid, DecodeState<G>& st) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, STRIDE = NWV * PPW;
    for (int b0 = c0; b0 < c1; b0 += U * STRIDE) {          // workgroup-uniform trip count
        const int p0 = b0 + wid * PPW;                      // this wave's first position of the trip
        KVBatch<U> kb;
        kv_issue<D, U, NWV>(kb, kbase, vbase, p0, c1 - 1, lane);
        kv_consume<D, G, U, NWV>(kb, p0, c1, qb, lane, st);
    }
}

// Merge the 4 waves x PPW lane-groups of a workgroup and write the chunk's partial records.
// lds: >= 4*PPW*G*(D+2) floats.  part points at record (head g=0) for this chunk; consecutive
// heads are `head_stride` floats apart.
template <int D, int G>
__device__ __forceinline__ void decode_block_merge(const DecodeState<G>& st, float* lds, float* part,
                                                   size_t head_stride, int lane, int wid) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, NS = 4 * PPW, RS = D + 2;
    const int grp = lane / LPR, sub = lane % LPR;
    const int slot = wid * PPW + grp;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* rec = lds + ((size_t)slot * G + g) * RS;
        if (sub == 0) { rec[0] = st.m[g]; rec[1] = st.l[g]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) rec[2 + sub * 8 + j] = st.o[g][j];
    }
    __syncthreads();
    // thread t handles output element (g, d) pairs strided over the block
    for (int e = threadIdx.x; e < G * D; e += blockDim.x) {
        const int g = e / D, d = e % D;
        float mx = -INFINITY;
        for (int s = 0; s < NS; ++s) mx = fmaxf(mx, lds[((size_t)s * G + g) * RS]);
        float l = 0.f, o = 0.f;
        for (int s = 0; s < NS; ++s) {
            const float* rec = lds + ((size_t)s * G + g) * RS;
            const float w = (rec[0] == -INFINITY) ? 0.f : __expf(rec[0] - mx);
            l = fmaf(rec[1], w, l);
            o = fmaf(rec[2 + d], w, o);
        }
        float* outrec = part + (size_t)g * head_stride;
        outrec[2 + d] = o;
        if (d == 0) { outrec[0] = mx; outrec[1] = l; }
    }
}

// Same merge, but the workgroup saw the WHOLE context: write the normalised attention output
// attn[g*D + d] into LDS (`attn_out`, G*D floats) for a consumer inside the same kernel.
// (Tried: folding the lane-groups of a wave with xor-shuffles first, so that only one record per wave crosses LDS - 40
// dependent ds_bpermutes per wave took longer than the 16-slot LDS pass they replaced, 1.04 vs 0.78 us in the fused kernel.)
template <int D, int G, int NWV = 4>
__device__ __forceinline__ void decode_block_merge_lds(const DecodeState<G>& st, float* lds, float* attn_out, int lane,
                                                       int wid) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, NS = NWV * PPW, RS = D + 2;
    const int grp = lane / LPR, sub = lane % LPR;
    const int slot = wid * PPW + grp;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float* rec = lds + ((size_t)slot * G + g) * RS;
        if (sub == 0) { rec[0] = st.m[g]; rec[1] = st.l[g]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) rec[2 + sub * 8 + j] = st.o[g][j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < G * D; e += blockDim.x) {
        const int g = e / D, d = e % D;
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) mx = fmaxf(mx, lds[((size_t)s * G + g) * RS]);
        float l = 0.f, o = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* rec = lds + ((size_t)s * G + g) * RS;
            const float w = (rec[0] == -INFINITY) ? 0.f : __expf(rec[0] - mx);
            l = fmaf(rec[1], w, l);
            o = fmaf(rec[2 + d], w, o);
        }
        attn_out[e] = l > 0.f ? o / l : 0.f;
    }
    __syncthreads();
}

// Combine `nsplit` chunk records of one head into the normalised output element d.
template <int D>
__device__ __forceinline__ float decode_combine(const float* recs, int nsplit, int d) {
    constexpr int RS = D + 2;
    float mx = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, recs[(size_t)s * RS]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* rec = recs + (size_t)s * RS;
        const float w = (rec[0] == -INFINITY) ? 0.f : __expf(rec[0] - mx);
        l = fmaf(rec[1], w, l);
        o = fmaf(rec[2 + d], w, o);
    }
    return l > 0.f ? o / l : 0.f;
}

// Chunking rule shared by producer and consumer: nsplit fixed at capture time from max_seq,
// chunk length derived from the live context length (read from device memory under a graph).
__device__ __host__ __forceinline__ int decode_chunk_len(int ctx, int nsplit, int gran = 32) {
    int c = (ctx + nsplit - 1) / nsplit;
    c = (c + gran - 1) / gran * gran;  // whole position-group steps (4 waves x 64 / (D/8) rows: 32 at D = 64, 16 at D = 128)
    return c < gran ? gran : c;
}

}  // namespace pgk
