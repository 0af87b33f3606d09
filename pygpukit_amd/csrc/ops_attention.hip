// Causal attention: MFMA flash-attention prefill and split-KV flash-decoding.
//
// Prefill (bf16/f16): workgroup = 4 waves = 64 query rows of one head (16 per wave), KV tiles of
// 64 positions.  S = Q.K^T and O += P.V on v_mfma_f32_16x16x32; online softmax in fp32 in the
// accumulator layout (row = (lane>>4)*4 + reg, col = lane&15); K tile row-major and V tile
// transposed in XOR-swizzled LDS; P goes through a per-wave LDS tile to become the A operand.
// GQA is handled by indexing (kv head = q head / (Hq/Hkv)) - no repeat_interleave copy, and
// element strides let the caller pass [S,H,D] projections without transposing to [H,S,D].
// Mask: kv_pos <= (kv_len - q_len) + q_pos  (reference: native/ops/nn/attention_kernels.cuh:32-148).

#include <type_traits>

#include <cstring>

#include "attn_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <class T> __device__ __forceinline__ f32x4_t mfma16a(const uint4& a, const uint4& b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16a<bf16>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16a<f16>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <class T> __device__ __forceinline__ uint16_t to_bits16(float f);
template <> __device__ __forceinline__ uint16_t to_bits16<bf16>(float f) { return f_to_bf16_bits(f); }
template <> __device__ __forceinline__ uint16_t to_bits16<f16>(float f) { return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f)); }

struct AttnStrides { long long qh, qs, kh, ks, oh, os; };

constexpr int FA_BQ = 64, FA_BKV = 64;

// [rows][64 x 16-bit] tile, 128-byte rows, 16-byte chunk kc XOR-swizzled by the row
__device__ __forceinline__ int t64_off(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }
// [rows][D x 16-bit] tile, chunk index XOR (row & (NC-1)) with NC = D/8 chunks per row
template <int D> __device__ __forceinline__ int tD_off(int row, int kc) {
    constexpr int NC = D / 8;
    return row * (D * 2) + ((kc ^ (row & (NC - 1))) << 4);
}

template <class T, int D>
__global__ __launch_bounds__(256) void flash_prefill_kernel(const T* q, const T* k, const T* v, T* out, int hq, int hkv,
                                                            int q_len, int kv_len, float scale, AttnStrides sd) {
    constexpr int NC = D / 8;          // 16-byte chunks per K row
    constexpr int KS = D / 32;         // k-steps for Q.K^T
    constexpr int DT = D / 16;         // output column tiles
    __shared__ __attribute__((aligned(16))) char k_lds[FA_BKV * D * 2];
    __shared__ __attribute__((aligned(16))) char vt_lds[D * 128];
    __shared__ __attribute__((aligned(16))) char p_lds[4 * 16 * 128];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int head = blockIdx.y, kvh = head / (hq / hkv);
    const int q0 = blockIdx.x * FA_BQ + wid * 16;
    const int causal_off = kv_len - q_len;
    const T* qh = q + (size_t)head * sd.qh;
    const T* kh = k + (size_t)kvh * sd.kh;
    const T* vh = v + (size_t)kvh * sd.kh;

    // Q fragments: A operand, row = lane & 15, k = ks*32 + (lane>>4)*8 .. +8
    uint4 qf[KS];
    {
        const int qr = min(q0 + (lane & 15), q_len - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = *reinterpret_cast<const uint4*>(qh + (size_t)qr * sd.qs + ks * 32 + (lane >> 4) * 8);
    }
    f32x4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    // kv positions needed by this workgroup's 64 query rows
    const int q_last = min(blockIdx.x * FA_BQ + FA_BQ - 1, q_len - 1);
    const int kv_end = min(kv_len, causal_off + q_last + 1);
    char* pw = p_lds + wid * (16 * 128);

    for (int kv0 = 0; kv0 < kv_end; kv0 += FA_BKV) {
        __syncthreads();  // previous tile fully consumed
        // stage K tile (row-major, swizzled) and V tile (transposed, swizzled)
        for (int c = threadIdx.x; c < FA_BKV * NC; c += 256) {
            const int r = c / NC, kc = c % NC;
            const int pos = kv0 + r;
            uint4 kvv = make_uint4(0, 0, 0, 0), vvv = make_uint4(0, 0, 0, 0);
            if (pos < kv_len) {
                kvv = *reinterpret_cast<const uint4*>(kh + (size_t)pos * sd.ks + kc * 8);
                vvv = *reinterpret_cast<const uint4*>(vh + (size_t)pos * sd.ks + kc * 8);
            }
            *reinterpret_cast<uint4*>(k_lds + tD_off<D>(r, kc)) = kvv;
            const uint32_t w[4] = {vvv.x, vvv.y, vvv.z, vvv.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = kc * 8 + j;
                const uint16_t e = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu));
                *reinterpret_cast<uint16_t*>(vt_lds + t64_off(d, r >> 3) + (r & 7) * 2) = e;
            }
        }
        __syncthreads();

        // S = Q K^T : 4 column tiles of 16 kv positions
        f32x4_t s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4 kb = *reinterpret_cast<const uint4*>(k_lds + tD_off<D>(t * 16 + (lane & 15), ks * 4 + (lane >> 4)));
                s[t] = mfma16a<T>(qf[ks], kb, s[t]);
            }
        // mask + online softmax; this lane holds rows (lane>>4)*4 + r, column lane&15 of each tile
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qpos = q0 + (lane >> 4) * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kvpos = kv0 + t * 16 + (lane & 15);
                const bool ok = kvpos < kv_len && kvpos <= causal_off + qpos;
                s[t][r] = ok ? s[t][r] * scale : -INFINITY;
                mx = fmaxf(mx, s[t][r]);
            }
            mx = group16_max(mx);
            const float mn = fmaxf(m_run[r], mx);
            alpha[r] = (mn == -INFINITY) ? 1.f : __expf(m_run[r] - mn);
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float p = (mn == -INFINITY) ? 0.f : __expf(s[t][r] - mn);
                s[t][r] = p;
                ls += p;
            }
            ls = group16_sum(ls);
            l_run[r] = l_run[r] * alpha[r] + ls;
            m_run[r] = mn;
        }
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha[r];
        // P -> per-wave LDS tile [16 q][64 kv] (A-operand layout source)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) * 4 + r, col = t * 16 + (lane & 15);
                *reinterpret_cast<uint16_t*>(pw + t64_off(row, col >> 3) + (col & 7) * 2) = to_bits16<T>(s[t][r]);
            }
        __syncthreads();  // P visible (also orders the wave's own LDS write -> read)
        // O += P V : A = P[16 x 64], B = V^T tile rows d, k = kv
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const uint4 pa = *reinterpret_cast<const uint4*>(pw + t64_off(lane & 15, ks2 * 4 + (lane >> 4)));
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const uint4 vb = *reinterpret_cast<const uint4*>(vt_lds + t64_off(i * 16 + (lane & 15), ks2 * 4 + (lane >> 4)));
                o[i] = mfma16a<T>(pa, vb, o[i]);
            }
        }
    }
    // normalise and store
    T* oh = out + (size_t)head * sd.oh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qpos = q0 + (lane >> 4) * 4 + r;
        if (qpos >= q_len) continue;
        const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i) oh[(size_t)qpos * sd.os + i * 16 + (lane & 15)] = from_f<T>(o[i][r] * inv);
    }
}

// ---- whole context in one tile: kv_len <= 128, head_dim 128, bf16 -------------------------------------------------
// A prompt of up to 128 tokens needs no online softmax and no K/V loop, and the 64-row flash kernel above spends its ~10 us
// on three barriers per 64-position tile and on transposing V with 2-byte LDS writes.  Here:
//   * 32 query rows per workgroup (2 waves; 4 x heads workgroups), K and V rows [0, needed) staged once, row-major;
//   * TRANSPOSED scores S^T = K.Q^T: the C layout then gives every lane ONE query row (column = lane & 15) and four
//     consecutive positions per tile, so max / sum are lane-local plus two 16-lane-stride shuffles, and the probabilities
//     of tiles 2s, 2s+1 packed to bf16 ARE the operand of the second product with no LDS round trip - the k-slot of
//     element j of lane quarter q is position 32 s + 16 (j >> 2) + 4 q + (j & 3), and V is read with the same slots;
//   * V comes out of its row-major image through ds_read_b64_tr_b16 (gfx950's transposing LDS read), two per fragment;
//   * O^T = V^T.P^T: a lane ends with 4 consecutive dims of its query row per tile = one 8-byte store, 1/l per lane.
// Layout (b) of the CDNA guide (T10) for the V image: 256-byte rows, chunk ch at ch ^ (((row & 3) << 2) | ((row >> 2) & 3)).
typedef short sa_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sa_voff(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_short_kernel(const bf16* q, const bf16* k, const bf16* v, bf16* out, int hq, int hkv,
                                                             int q_len, int kv_len, float scale, AttnStrides sd) {
    constexpr int D = 128, NT = 8;
    __shared__ __attribute__((aligned(16))) char k_lds[128 * 256];
    __shared__ __attribute__((aligned(16))) char v_lds[128 * 256];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, q4 = lane >> 4;
    const int head = blockIdx.y, kvh = head / (hq / hkv);
    const int qb0 = blockIdx.x * (16 * NW), q0 = qb0 + wid * 16;
    const int causal_off = kv_len - q_len;
    const bf16* qh = q + (size_t)head * sd.qh;
    const bf16* kh = k + (size_t)kvh * sd.kh;
    const bf16* vh = v + (size_t)kvh * sd.kh;
    // query fragments first (the B operand of S^T): row l15 of this wave's tile, dims 32 ks + 8 q4 .. + 8
    uint4 qf[4];
    const int qrow = q0 + l15;
    {
        const bf16* qp = qh + (size_t)min(qrow, q_len - 1) * sd.qs + 8 * q4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qp + 32 * ks);
    }
    // positions this workgroup can see, in whole 32-position steps (rows past kv_len come from a clamped address and are masked).
    // Staging is LDS-DMA: one instruction moves 4 rows x 256 bytes, lane i -> row 4 j + (i >> 4), chunk POSITION i & 15, and
    // the swizzle of each image is applied on the source side (both are XORs of the chunk index: involutions).  A loop of
    // load -> ds_write pairs here costs one memory round trip per iteration (the first version: 10.6 us, no faster than
    // the flash kernel); the DMAs are all in flight at once and hold no registers.
    const int kv_need = min(kv_len, causal_off + min(qb0 + 16 * NW, q_len));
    const int nstep = (kv_need + 31) >> 5;
    {
        const uint32_t k0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)k_lds;
        const uint32_t v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)v_lds;
        auto dma = [](const void* src, uint32_t lds_addr) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
        };
        const int ninst = nstep * 8;                       // 4 rows each
        for (int j = wid; j < ninst; j += NW) {
            const int r = 4 * j + q4;
            dma(kh + (size_t)min(r, kv_len - 1) * sd.ks + ((l15 ^ (r & 15)) << 3), k0 + j * 1024);
        }
        for (int j = wid; j < ninst; j += NW) {
            const int r = 4 * j + q4;
            dma(vh + (size_t)min(r, kv_len - 1) * sd.ks + ((l15 ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3), v0 + j * 1024);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs (and its query fragments); the barrier makes the images everyone's
    __syncthreads();
    // this wave's own horizon (wave-uniform): tiles of 16 positions it needs
    const int my_need = min(kv_len, causal_off + min(q0 + 16, q_len));
    const int my_steps = (my_need + 31) >> 5;
    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if ((t >> 1) < my_steps) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 ka = *reinterpret_cast<const uint4*>(k_lds + tD_off<D>(t * 16 + l15, ks * 4 + q4));
                s[t] = mfma16a<bf16>(ka, qf[ks], s[t]);          // rows = positions 16 t + 4 q4 + r, column = query row l15
            }
        }
    }
    const int last_ok = min(kv_len - 1, causal_off + qrow);       // last position query row `qrow` may attend
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = (t >> 1) < my_steps && 16 * t + 4 * q4 + r <= last_ok;
            s[t][r] = ok ? s[t][r] * scale : -INFINITY;
            mx = fmaxf(mx, s[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
    uint4 pf[NT / 2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = (mx == -INFINITY) ? 0.f : __expf(s[t][r] - mx);
        const uint32_t lo = pack_bf16x2(p[0], p[1]), hi = pack_bf16x2(p[2], p[3]);
        // the sum runs over the ROUNDED probabilities: numerator and denominator see the same weights
        ls += (__uint_as_float(lo << 16) + __uint_as_float(lo & 0xFFFF0000u)) + (__uint_as_float(hi << 16) + __uint_as_float(hi & 0xFFFF0000u));
        if (t & 1) { pf[t >> 1].z = lo; pf[t >> 1].w = hi; } else { pf[t >> 1].x = lo; pf[t >> 1].y = hi; }
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    // O^T[d][q] += V^T[d][kv] P^T[kv][q] over steps of 32 positions; lane (l15, q4) of the A operand supplies the address of
    // row (l15 >> 2) of its 16-lane group's 4 x 16 block, columns 4 (l15 & 3) .. + 4
    f32x4_t o[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int tq = l15 >> 2, tp = l15 & 3;
#pragma unroll
    for (int st2 = 0; st2 < NT / 2; ++st2) {
        if (st2 < my_steps) {
            const int r0 = 32 * st2 + 4 * q4 + tq;
#pragma unroll
            for (int i = 0; i < D / 16; ++i) {
                const sa_v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) sa_v4s*)(v_lds + sa_voff(r0, 2 * i + (tp >> 1)) + 8 * (tp & 1)));
                const sa_v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) sa_v4s*)(v_lds + sa_voff(r0 + 16, 2 * i + (tp >> 1)) + 8 * (tp & 1)));
                const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
                o[i] = mfma16a<bf16>(make_uint4(u0.x, u0.y, u1.x, u1.y), pf[st2], o[i]);   // rows = dims 16 i + 4 q4 + r, column = query row l15
            }
        }
    }
    if (qrow < q_len) {
        const float inv = ls > 0.f ? 1.f / ls : 0.f;
        bf16* op = out + (size_t)head * sd.oh + (size_t)qrow * sd.os + 4 * q4;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            uint2 w;
            w.x = pack_bf16x2(o[i][0] * inv, o[i][1] * inv);
            w.y = pack_bf16x2(o[i][2] * inv, o[i][3] * inv);
            *reinterpret_cast<uint2*>(op + 16 * i) = w;
        }
    }
}

// fp32 / odd head_dim fallback: one workgroup per (head, query row), scores kept in LDS.
template <class T>
__global__ __launch_bounds__(256) void sdpa_naive_kernel(const T* q, const T* k, const T* v, T* out, int hq, int hkv,
                                                         int q_len, int kv_len, int d, float scale, AttnStrides sd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [kv_len]
    __shared__ float red[16];
    const int head = blockIdx.x, qi = blockIdx.y, kvh = head / (hq / hkv);
    const T* qr = q + (size_t)head * sd.qh + (size_t)qi * sd.qs;
    const int n_att = min(kv_len, (kv_len - q_len) + qi + 1);
    float mx = -INFINITY;
    for (int p = threadIdx.x; p < n_att; p += blockDim.x) {
        const T* kr = k + (size_t)kvh * sd.kh + (size_t)p * sd.ks;
        float dot = 0.f;
        for (int j = 0; j < d; ++j) dot = fmaf(to_f(qr[j]), to_f(kr[j]), dot);
        dot *= scale;
        sc[p] = dot;
        mx = fmaxf(mx, dot);
    }
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 4; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
    for (int p = threadIdx.x; p < n_att; p += blockDim.x) {
        const float e = expf(sc[p] - mx);
        sc[p] = e;
        sum += e;
    }
    sum = block_sum(sum, red + 8);
    __syncthreads();
    const float inv = 1.f / sum;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        float acc = 0.f;
        for (int p = 0; p < n_att; ++p) acc = fmaf(sc[p], to_f(v[(size_t)kvh * sd.kh + (size_t)p * sd.ks + j]), acc);
        out[(size_t)head * sd.oh + (size_t)qi * sd.os + j] = from_f<T>(acc * inv);
    }
}

// ---- flash-decoding (q_len == 1) ---------------------------------------------------------------
template <class T, int D, int G>
__global__ __launch_bounds__(256) void decode_phase1_kernel(const T* q, const T* kc, const T* vc, float* ws, int hq,
                                                            int hc, int max_seq, float scale, int host_ctx,
                                                            const int32_t* ctx_buf, int nsplit) {
    constexpr int LPR = D / 8, PPW = 64 / LPR, RS = D + 2;
    __shared__ float lds[4 * PPW * G * RS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ctx = min(ctx_buf ? ctx_buf[0] : host_ctx, max_seq);
    const int hgroup = blockIdx.y;                  // group of G consecutive query heads
    const int h0 = hgroup * G;
    const int cache_head = h0 / (hq / hc);          // all G heads share it (G divides Hq/Hc, or Hc == Hq and G == 1)
    const int chunk = decode_chunk_len(ctx, nsplit);
    const int c0 = min(blockIdx.x * chunk, ctx), c1 = min(c0 + chunk, ctx);
    float qf[G][8];
    const int sub = lane % LPR;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        KVLoad<T>::load8(q + (size_t)(h0 + g) * D + sub * 8, qf[g]);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[g][j] *= scale;
    }
    DecodeState<G> st;
    st.init();
    decode_walk<T, D, G>(kc + (size_t)cache_head * max_seq * D, vc + (size_t)cache_head * max_seq * D, c0, c1, qf, lane,
                         wid, st);
    decode_block_merge<D, G>(st, lds, ws + ((size_t)h0 * nsplit + blockIdx.x) * RS, (size_t)nsplit * RS, lane, wid);
}

template <class T, int D>
__global__ void decode_phase2_kernel(const float* ws, T* out, int nsplit) {
    const int h = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        out[(size_t)h * D + d] = from_f<T>(decode_combine<D>(ws + (size_t)h * nsplit * (D + 2), nsplit, d));
}

static int decode_nsplit(int max_seq) {
    int n = (max_seq + 255) / 256;  // ~256 positions per chunk at full context
    if (n < 1) n = 1;
    return n > 64 ? 64 : n;
}

template <class T, int D>
static pgk_status launch_prefill(const T* q, const T* k, const T* v, T* out, int hq, int hkv, int q_len, int kv_len,
                                 float scale, const AttnStrides& sd, hipStream_t st) {
    dim3 grid(ceil_div(q_len, FA_BQ), hq);
    flash_prefill_kernel<T, D><<<grid, 256, 0, st>>>(q, k, v, out, hq, hkv, q_len, kv_len, scale, sd);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

pgk_status flash_prefill(const void* q, const void* k, const void* v, void* out, int hq, int hkv, int q_len, int kv_len, int d,
                         float scale, long long qh, long long qs, long long kh, long long ks, long long oh, long long os,
                         int dt16, hipStream_t st);

// The reference's A/B switches, same names and values (native/ops/nn/attention/sdpa_causal.inl:380-447), read per call:
//   PYGPUKIT_FLASH_ATTENTION  "0"/"false": never the tiled MFMA flash kernels (one-workgroup-per-row fallback, kv_len <= 15360);
//                             "1"/"true" or unset/"auto": flash whenever the layout allows it (there is no length threshold
//                             here: the fallback is never faster on this chip)
//   PYGPUKIT_FLASH_DECODING   0: single-token attention over a fixed cache takes the general path instead of split-KV
//                             flash-decoding (only possible with a host context length); 1 / -1 / unset: split-KV
static bool flash_attention_off() {
    const char* e = getenv("PYGPUKIT_FLASH_ATTENTION");
    return e && (strcmp(e, "0") == 0 || strcmp(e, "false") == 0);
}
static bool flash_decoding_off() {
    const char* e = getenv("PYGPUKIT_FLASH_DECODING");
    return e && atoi(e) == 0 && strcmp(e, "auto") != 0;
}

bool sdpa_flash_enabled() { return !flash_attention_off(); }   // for the engine's direct use of ops_flash.hip

template <class T>
static pgk_status sdpa_dispatch(const void* q, const void* k, const void* v, void* out, int hq, int hkv, int q_len,
                                int kv_len, int d, float scale, const AttnStrides& sd, hipStream_t st) {
    const bool mfma_ok = !std::is_same<T, float>::value && (d == 64 || d == 128) && aligned16(q) && aligned16(k) &&
                         aligned16(v) && sd.qs % 8 == 0 && sd.ks % 8 == 0 && sd.qh % 8 == 0 && sd.kh % 8 == 0 && !flash_attention_off();
    if constexpr (!std::is_same<T, float>::value) {
        // second generation: 128-row query tiles, transposed-score orientation (ops_flash.hip); the first-generation
        // kernel keeps the short prompts, where its 64-row tiles give twice the workgroups
        const bool gen2_ok = mfma_ok && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && sd.os % 4 == 0 && sd.oh % 4 == 0;
        if (gen2_ok && q_len > 128)
            return flash_prefill(q, k, v, out, hq, hkv, q_len, kv_len, d, scale, sd.qh, sd.qs, sd.kh, sd.ks, sd.oh, sd.os,
                                 std::is_same<T, f16>::value ? 1 : 0, st);
        if constexpr (std::is_same<T, bf16>::value) {
            // the whole context in one tile (attn_short_kernel)
            if (gen2_ok && d == 128 && kv_len <= 128) {
                attn_short_kernel<2><<<dim3(ceil_div(q_len, 32), hq), 128, 0, st>>>((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, hq, hkv,
                                                                                  q_len, kv_len, scale, sd);
                PGK_CHECK_HIP(hipGetLastError());
                return PGK_OK;
            }
        }
        if (mfma_ok) {
            if (d == 128) return launch_prefill<T, 128>((const T*)q, (const T*)k, (const T*)v, (T*)out, hq, hkv, q_len, kv_len, scale, sd, st);
            return launch_prefill<T, 64>((const T*)q, (const T*)k, (const T*)v, (T*)out, hq, hkv, q_len, kv_len, scale, sd, st);
        }
    }
    const size_t lds = (size_t)kv_len * 4;
    if (lds > 60 * 1024) return set_error(PGK_ERR_UNSUPPORTED, "sdpa: fallback kernel supports kv_len <= 15360 (got %d)", kv_len);
    dim3 grid(hq, q_len);
    sdpa_naive_kernel<T><<<grid, 256, lds, st>>>((const T*)q, (const T*)k, (const T*)v, (T*)out, hq, hkv, q_len, kv_len, d, scale, sd);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_sdpa_causal(const void* q, const void* k, const void* v, void* out, int hq, int hkv, int q_len,
                           int kv_len, int d, float scale, int64_t q_stride_h, int64_t q_stride_s, int64_t kv_stride_h,
                           int64_t kv_stride_s, int64_t o_stride_h, int64_t o_stride_s, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(q && k && v && out, "pgk_sdpa_causal: null pointer");
    PGK_REQUIRE(hq > 0 && hkv > 0 && hq % hkv == 0, "pgk_sdpa_causal: n_heads mismatch (Hq=%d, Hkv=%d)", hq, hkv);
    PGK_REQUIRE(q_len > 0 && kv_len > 0 && d > 0, "pgk_sdpa_causal: bad shape q_len=%d kv_len=%d D=%d", q_len, kv_len, d);
    PGK_REQUIRE(kv_len >= q_len, "pgk_sdpa_causal: kv_len %d < q_len %d", kv_len, q_len);
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)d);
    const AttnStrides sd{q_stride_h, q_stride_s, kv_stride_h, kv_stride_s, o_stride_h, o_stride_s};
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_sdpa_causal", return (sdpa_dispatch<T>(q, k, v, out, hq, hkv, q_len, kv_len, d, scale, sd, st)));
    return PGK_OK;
}

size_t pgk_sdpa_decode_workspace_bytes(int hq, int d, int max_seq) {
    return (size_t)hq * decode_nsplit(max_seq) * (d + 2) * sizeof(float);
}

pgk_status pgk_sdpa_fixed_cache(const void* q, const void* k_cache, const void* v_cache, void* out, int hq, int hc,
                                int q_len, int max_seq, int d, float scale, int h_context_len, const int32_t* ctx_buf,
                                void* workspace, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(q && k_cache && v_cache && out, "pgk_sdpa_fixed_cache: null pointer");
    PGK_REQUIRE(hq > 0 && hc > 0 && hq % hc == 0, "pgk_sdpa_fixed_cache: n_heads mismatch (Hq=%d, cache heads=%d)", hq, hc);
    PGK_REQUIRE(ctx_buf || (h_context_len > 0 && h_context_len <= max_seq), "sdpa: invalid context_len %d (cache rows %d)",
                h_context_len, max_seq);
    PGK_REQUIRE(q_len >= 1, "pgk_sdpa_fixed_cache: q_len=%d", q_len);
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)d);
    hipStream_t st = resolve_stream(s);
    const int rep = hq / hc;
    const bool fast = q_len == 1 && (d == 128 || d == 64) && workspace && aligned16(q) && aligned16(k_cache) && aligned16(v_cache) &&
                      !(flash_decoding_off() && !ctx_buf);
    if (fast) {
        const int nsplit = decode_nsplit(max_seq);
        int G = 1;
        if (rep % 4 == 0) G = 4; else if (rep % 2 == 0) G = 2;
        dim3 grid(nsplit, hq / G);
        float* ws = (float*)workspace;
#define PGK_DEC(DD, GG)                                                                                               \
    if (d == DD && G == GG) {                                                                                         \
        PGK_DISPATCH_FLOAT(dt, "pgk_sdpa_fixed_cache", {                                                              \
            decode_phase1_kernel<T, DD, GG><<<grid, 256, 0, st>>>((const T*)q, (const T*)k_cache, (const T*)v_cache, ws, hq, hc, \
                                                                  max_seq, scale, h_context_len, ctx_buf, nsplit);   \
            decode_phase2_kernel<T, DD><<<hq, DD, 0, st>>>(ws, (T*)out, nsplit);                                      \
        });                                                                                                           \
        PGK_LAUNCH_CHECK();                                                                                           \
        return PGK_OK;                                                                                                \
    }
        PGK_DEC(128, 1) PGK_DEC(128, 2) PGK_DEC(128, 4) PGK_DEC(64, 1) PGK_DEC(64, 2) PGK_DEC(64, 4)
#undef PGK_DEC
    }
    // general path (q_len > 1, other head dims, or no workspace): strided causal SDPA over the cache prefix.
    // A device-resident context length needs the decode path above.
    PGK_REQUIRE(!ctx_buf, "pgk_sdpa_fixed_cache: context_len_buf requires q_len == 1, head_dim 64/128 and a workspace");
    PGK_REQUIRE(h_context_len >= q_len, "sdpa: context_len %d < q_len %d", h_context_len, q_len);
    const AttnStrides sd{(long long)q_len * d, d, (long long)max_seq * d, d, (long long)q_len * d, d};
    PGK_DISPATCH_FLOAT(dt, "pgk_sdpa_fixed_cache",
                       return (sdpa_dispatch<T>(q, k_cache, v_cache, out, hq, hc, q_len, h_context_len, d, scale, sd, st)));
    return PGK_OK;
}

}  // extern "C"
