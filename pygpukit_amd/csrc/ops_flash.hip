// Flash-attention prefill for gfx950 (bf16 / f16, head_dim 64 or 128), second generation.
//
// Orientation: the scores are computed TRANSPOSED, S^T[kv][q] = K . Q^T on v_mfma_f32_32x32x16, so a lane owns one
// query column (q = lane & 31) and its 16 accumulator registers per 32-kv sub-tile are kv rows.  Consequences:
//   * the online softmax (max, exp2, sum) of a query row is lane-local arithmetic over registers plus ONE exchange
//     with lane ^ 32 (the other half of the rows) - no LDS, no 16-lane butterflies;
//   * the probabilities are already laid out as the B operand of the second product O^T[d][q] += V^T[d][kv] . P^T[kv][q]:
//     registers 8s..8s+7 of a sub-tile, packed pairwise to 16-bit, ARE the k-step-s fragment (k order permuted:
//     element j of lane half h is kv = 16s + 8(j>>2) + 4h + (j&3)); the V^T operand is read in the same permuted
//     order (two 8-byte LDS reads per fragment), so P never touches LDS;
//   * alpha (the running-max rescale) and 1/l are per-lane scalars for the O^T accumulators (q is again the lane).
// V arrives transposed: a small pre-pass writes V^T [Hkv][D][kv_pad] (zero padded to a multiple of 64 positions) so
// that both LDS images are filled with 16-byte row-contiguous chunks.
// Workgroup = 4 waves = 128 query rows of one head; KV tiles of 64 positions, K [64][D] and V^T [D][64] double
// buffered in LDS (64 KiB at D = 128: two workgroups per CU); the global loads of tile t+1 are issued before the
// MFMAs of tile t and written to the other buffer after them: one barrier per tile.
// Workgroups are ordered heavy-first (causal: late query tiles see more keys) and all query heads of a kv head
// share id % Hkv, i.e. an XCD when Hkv is a multiple of 8, so the K/V stream of a head is served by one L2.
// Mask: kv_pos <= (kv_len - q_len) + q_pos  (reference: native/ops/nn/attention_kernels.cuh:32-148).

#include <type_traits>

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

typedef __bf16 bf16x8_fl __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_fl __attribute__((ext_vector_type(8)));
typedef float f32x16_fl __attribute__((ext_vector_type(16)));

template <class T> __device__ __forceinline__ f32x16_fl mfma32(const uint4& a, const uint4& b, f32x16_fl c);
template <> __device__ __forceinline__ f32x16_fl mfma32<bf16>(const uint4& a, const uint4& b, f32x16_fl c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_fl, a), __builtin_bit_cast(bf16x8_fl, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_fl mfma32<f16>(const uint4& a, const uint4& b, f32x16_fl c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_fl, a), __builtin_bit_cast(f16x8_fl, b), c, 0, 0, 0);
}
template <class T> __device__ __forceinline__ uint32_t pack16x2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack16x2<bf16>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack16x2<f16>(float lo, float hi) {
    return (uint32_t)__builtin_bit_cast(uint16_t, static_cast<_Float16>(lo)) | ((uint32_t)__builtin_bit_cast(uint16_t, static_cast<_Float16>(hi)) << 16);
}

struct FlashStrides { long long qh, qs, kh, ks, oh, os; };

constexpr int FL_BQ = 128, FL_BKV = 64, FL_THREADS = 256;

// K image: [64 kv][D] 16-bit, 16-byte chunk c (0..D/8-1) of row r at r*2D + ((c ^ (r & (D/8-1))) << 4)
template <int D> __device__ __forceinline__ int fl_k_off(int r, int c) { return r * (2 * D) + ((c ^ (r & (D / 8 - 1))) << 4); }
// V^T image: [D][64 kv] 16-bit = 128-byte rows, 8-byte chunk c8 (0..15) of row d at d*128 + ((c8 ^ ((d>>1) & 15)) << 3):
// the 32 rows a half-wave reads at one c8 then fall on 32 different bank pairs
__device__ __forceinline__ int fl_v_off(int d, int c8) { return d * 128 + ((c8 ^ ((d >> 1) & 15)) << 3); }

// KV split (short prompts): nsplit > 1 cuts a query tile's KV tiles into nsplit contiguous runs, one workgroup each; every
// workgroup leaves a NORMALISED partial output (type T) and its (m, l) per row, and flash_merge_kernel combines them.
// Without it the causal imbalance sets the time of a short prompt: at S = 2048 with 16 heads there are 256 workgroups -
// one per CU, the heaviest walks 32 KV tiles, the lightest 2 - and the launch lasts as long as the heaviest (52 us for
// 17 GFLOP).  Cut in two and dealt heavy-first, two half-runs share a CU and the CUs finish together.
struct FlashSplit {
    int nsplit;
    float* ml;      // [nsplit][Hq][q_len][2]  (m in the exp2 domain, l)
    void* o;        // [nsplit][q_len][Hq][D] of T
    // fp8 x fp8 prefill (D = 128, bf16): instead of out, e4m3 codes [q_len][Hq * D] and one scale per (row, head) [q_len][Hq]
    // - a head's 128 output dims ARE one 128-wide scale block of the o_proj's A operand (quantize_fp8_rows' contract:
    // scale = absmax / 448 of the bf16-rounded values, 1 for an all-zero block), written by whichever kernel holds the final
    // row: this one (nsplit == 1) or flash_merge_kernel
    uint8_t* q8;
    float* q8s;
};

template <class T, int D>
__global__ __launch_bounds__(FL_THREADS, 2) void flash_fwd_kernel(const T* q, const T* k, const T* vt, T* out, int hq, int hkv,
                                                              int q_len, int kv_len, int kv_pad, float scale_log2e, FlashStrides sd,
                                                              FlashSplit sp) {
    constexpr int NC = D / 8;            // 16-byte chunks per K row
    constexpr int KS = D / 16;           // k-steps of Q.K^T
    constexpr int DT = D / 32;           // 32-row tiles of O^T
    constexpr int KCH = 64 * NC / FL_THREADS;   // K chunks staged per thread
    constexpr int VCH = D * 8 / FL_THREADS;     // V^T chunks staged per thread
    constexpr int K_BYTES = 64 * D * 2, V_BYTES = D * 128;
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];   // K[2] | V^T[2]
    auto Ks = [&](int buf) -> char* { return fl_smem + buf * K_BYTES; };
    auto Vs = [&](int buf) -> char* { return fl_smem + 2 * K_BYTES + buf * V_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, ql = lane & 31, h = lane >> 5;
    // workgroup id -> (kv head, query head in its group, query tile, KV run), heavy tiles first
    const int rep = hq / hkv, nqt = (q_len + FL_BQ - 1) / FL_BQ;
    const int id = blockIdx.x;
    const int kvh = id % hkv, rest = id / hkv;
    const int head = kvh * rep + rest % rep;
    const int rest2 = rest / rep;
    // heavy tiles first - and, when the launch is one round of two workgroups per CU (KV split), the second half of the ids in
    // ASCENDING weight: a CU's two residents are then a heavy and a light run (ids i and i + grid / 2 sit on the same CU when
    // the dispatcher deals one workgroup to every CU before the second)
    int order = rest2;
    const int nord = nqt * sp.nsplit;
    if (sp.nsplit > 1 && order >= nord / 2) order = nord / 2 + (nord - 1 - order);
    const int split = order % sp.nsplit;
    const int qt = nqt - 1 - order / sp.nsplit;
    const int qw0 = qt * FL_BQ + wid * 32;          // first query row of this wave
    const int causal_off = kv_len - q_len;
    const T* qh = q + (size_t)head * sd.qh;
    const T* kh = k + (size_t)kvh * sd.kh;
    const T* vh = vt + (size_t)kvh * D * kv_pad;

    // Q^T fragments (B operand): lane = query column, k = d
    uint4 qf[KS];
    {
        // Q is pre-multiplied by scale * log2(e) here (one rounding to 16 bits more, inside the 1e-2 bar): the scores come
        // out of the MFMA already in the exp2 domain and the per-tile scaling pass (32 multiplies per lane) disappears
        const T* qrow = qh + (size_t)min(qw0 + ql, q_len - 1) * sd.qs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Vec<T> v;
            v.load(qrow + ks * 16);
            float f[8];
            v.to_float(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= scale_log2e;
            v.from_float(f);
            qf[ks] = v.raw;
        }
    }
    f32x16_fl o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;    // running max (scaled, log2 domain) and this lane's half of the row sum

    const int q_last = min(qt * FL_BQ + FL_BQ - 1, q_len - 1);
    const int kv_end = min(kv_len, causal_off + q_last + 1);
    const int nt_all = (kv_end + FL_BKV - 1) / FL_BKV;
    // this workgroup's run of KV tiles [t0, t1)
    const int t0 = (int)((long long)nt_all * split / sp.nsplit), t1 = (int)((long long)nt_all * (split + 1) / sp.nsplit);

    // staging in NAMED registers (an array here ends up in scratch memory at this register pressure, see
    // ops_fp8_gemm.hip): chunk i of this thread is c = tid + 256 i; KCH = VCH = D / 32 chunks each
    static_assert(KCH == VCH && (KCH == 2 || KCH == 4), "staging is written for D = 64 / 128");
    uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    const int kr = tid / NC, kc16 = tid % NC;               // K: row kr + (256/NC) i, chunk kc16
    constexpr int KR_STEP = FL_THREADS / NC;
    const int vd = tid >> 3, vc = tid & 7;                  // V^T: row vd + 32 i, chunk vc
    auto k_src = [&](int kv0, int i) { return kh + (size_t)min(kv0 + kr + KR_STEP * i, kv_len - 1) * sd.ks + kc16 * 8; };
    // Whole tiles address their rows as (wave-uniform tile base) + (per-lane element offset fixed for the kernel): the
    // per-tile address arithmetic is scalar.  (Recomputing min(row, kv_len - 1) * stride per chunk cost 48 vector
    // instructions per tile - 64-bit multiplies included - in a loop that is bound by vector issue.)  Only the ragged
    // last tile clamps its rows.
    const uint32_t ko0 = (uint32_t)(kr * sd.ks + kc16 * 8), ko_step = (uint32_t)(KR_STEP * sd.ks);
    const uint32_t vo0 = (uint32_t)(vd * kv_pad + vc * 8), vo_step = (uint32_t)(32 * kv_pad);
    auto load_k = [&](int t) {
        const int kv0 = t * FL_BKV;
        if (kv0 + FL_BKV <= kv_len) {      // wave-uniform
            const T* kt = kh + (size_t)kv0 * sd.ks;
            rk0 = *reinterpret_cast<const uint4*>(kt + ko0);
            rk1 = *reinterpret_cast<const uint4*>(kt + ko0 + ko_step);
            if constexpr (KCH == 4) {
                rk2 = *reinterpret_cast<const uint4*>(kt + ko0 + 2 * ko_step);
                rk3 = *reinterpret_cast<const uint4*>(kt + ko0 + 3 * ko_step);
            }
        } else {
            rk0 = *reinterpret_cast<const uint4*>(k_src(kv0, 0));
            rk1 = *reinterpret_cast<const uint4*>(k_src(kv0, 1));
            if constexpr (KCH == 4) {
                rk2 = *reinterpret_cast<const uint4*>(k_src(kv0, 2));
                rk3 = *reinterpret_cast<const uint4*>(k_src(kv0, 3));
            }
        }
    };
    auto load_v = [&](int t) {
        const T* vtile = vh + t * FL_BKV;
        rv0 = *reinterpret_cast<const uint4*>(vtile + vo0);
        rv1 = *reinterpret_cast<const uint4*>(vtile + vo0 + vo_step);
        if constexpr (VCH == 4) {
            rv2 = *reinterpret_cast<const uint4*>(vtile + vo0 + 2 * vo_step);
            rv3 = *reinterpret_cast<const uint4*>(vtile + vo0 + 3 * vo_step);
        }
    };
    auto put_v = [&](char* base, int d, const uint4& x) {
        *reinterpret_cast<uint2*>(base + fl_v_off(d, 2 * vc)) = make_uint2(x.x, x.y);
        *reinterpret_cast<uint2*>(base + fl_v_off(d, 2 * vc + 1)) = make_uint2(x.z, x.w);
    };
    auto store_k = [&](int buf) {
        char* kb = Ks(buf);
        *reinterpret_cast<uint4*>(kb + fl_k_off<D>(kr, kc16)) = rk0;
        *reinterpret_cast<uint4*>(kb + fl_k_off<D>(kr + KR_STEP, kc16)) = rk1;
        if constexpr (KCH == 4) {
            *reinterpret_cast<uint4*>(kb + fl_k_off<D>(kr + 2 * KR_STEP, kc16)) = rk2;
            *reinterpret_cast<uint4*>(kb + fl_k_off<D>(kr + 3 * KR_STEP, kc16)) = rk3;
        }
    };
    auto store_v = [&](int buf) {
        char* vb = Vs(buf);
        put_v(vb, vd, rv0);
        put_v(vb, vd + 32, rv1);
        if constexpr (VCH == 4) {
            put_v(vb, vd + 64, rv2);
            put_v(vb, vd + 96, rv3);
        }
    };
    // S^T = K . Q^T of one tile: two 32-kv sub-tiles (rows = positions, column = this lane's query)
    auto qk = [&](int buf, f32x16_fl& s0, f32x16_fl& s1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(Ks(buf) + fl_k_off<D>(ql, 2 * ks + h));
            const uint4 a1 = *reinterpret_cast<const uint4*>(Ks(buf) + fl_k_off<D>(32 + ql, 2 * ks + h));
            s0 = mfma32<T>(a0, qf[ks], s0);
            s1 = mfma32<T>(a1, qf[ks], s1);
        }
    };
    // tiles entirely above this wave's diagonal contribute nothing (wave-uniform)
    auto live = [&](int t) -> bool { return t < t1 && t * FL_BKV <= causal_off + qw0 + 31; };

    // One KV tile per iteration: S = K . Q^T, softmax, O += V . P; the global loads of tile t + 1 are issued before the MFMAs of
    // tile t and parked in the other LDS buffer after them: one barrier per tile.  (Tried in round 3: multiplying S(t + 1) under
    // the softmax of S(t).  As two conditional blocks hipcc keeps them apart - QK block, 64 register copies, softmax - and
    // the kernel lost 2 %; it needs a branch-free, two-tile-unrolled body to interleave.)
    f32x16_fl sc0, sc1;
    if (t0 < t1) {
        load_k(t0);
        load_v(t0);
        store_k(t0 & 1);
        store_v(t0 & 1);
    }
    // Retire the Q loads HERE: if they are still counted as pending at the loop header, hipcc's conservative
    // vmcnt bookkeeping waits for them in every iteration - behind the tile prefetch issued at the top of the
    // loop, i.e. it drains the prefetch before the first MFMA.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks].x), "v"(qf[ks].y), "v"(qf[ks].z), "v"(qf[ks].w));
    __syncthreads();
    typedef uint32_t fl_u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const volatile fl_u32x2* fl_lds_cv64;     // volatile drops the inferred address space: say it
    for (int t = t0; t < t1; ++t) {
        const int buf = t & 1, kv0 = t * FL_BKV;
        if (t + 1 < t1) { load_k(t + 1); load_v(t + 1); }
        const bool cur = live(t);
        if (cur) qk(buf, sc0, sc1);
        if (cur) {
            // ---- mask, online softmax (this lane: query qw0 + ql, kv rows (r&3) + 8(r>>2) + 4h of each sub-tile) ----
            const bool need_mask = kv0 + FL_BKV - 1 > causal_off + qw0 || kv0 + FL_BKV > kv_len;   // wave-uniform
            const int lim = min(causal_off + qw0 + ql, kv_len - 1) - kv0 - 4 * h;                  // local kv row <= lim is visible
            float mx = -INFINITY;
            if (need_mask) {   // diagonal / ragged tiles only: a real wave-uniform branch, not 32 predicated selects per tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kvl = (r & 3) + 8 * (r >> 2);
                    sc0[r] = kvl <= lim ? sc0[r] : -INFINITY;
                    sc1[r] = 32 + kvl <= lim ? sc1[r] : -INFINITY;
                }
            }
            // (this file is built with -fno-honor-nans: without it every operand of a max is first canonicalised with a
            // v_max_f32 x, x, x of its own - 60 instructions for these 32 values instead of 16 v_max3_f32)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(fmaxf(sc0[r], sc1[r]), fmaxf(sc0[r + 1], sc1[r + 1])), mx);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Deferred rescale: the reference point of a row only moves when its maximum grew by more than 6 (log2
            // domain), so probabilities stay <= 64 (exact in fp32 sums, fine in 16-bit P) and on most tiles - for the whole
            // wave - alpha is exactly 1 and the 64-multiply rescale of O is skipped.
            const float m_new = (mx > m_run + 6.0f || m_run == -INFINITY) ? fmaxf(m_run, mx) : m_run;
            const float m_use = m_new == -INFINITY ? 0.f : m_new;        // a fully masked row so far: p = exp2(-inf) = 0
            const bool moved = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0;   // wave-uniform
            const float alpha = moved ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;   // m_run = -inf -> 0 (accumulators are 0 anyway)
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc0[r] = __builtin_amdgcn_exp2f(sc0[r] - m_use);
                sc1[r] = __builtin_amdgcn_exp2f(sc1[r] - m_use);
                ls += sc0[r] + sc1[r];
            }
            l_run = l_run * alpha + ls;
            m_run = m_new;
            if (moved) {
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            // ---- O^T += V^T . P^T : 4 k-steps of 16 kv ----
            // V^T fragment addresses: row d = 32 i + ql, 8-byte chunk c8 = 4 s + 2 u + h at d * 128 + ((c8 ^ ((d >> 1) & 15)) << 3).
            // 32 i is a multiple of 32, so the swizzle term depends on the lane and on c8 only: ONE address register per
            // (s, u) for this tile's buffer and the tile row i as the instruction's immediate offset (4096 i).  The reads are
            // VOLATILE: left ordinary, hipcc pairs those of tiles i and i + 2 into ds_read2st64_b64, which is served 16 lanes
            // at a time on a 32-bank modulus (2-way conflicts, half the rate); the first version kept them apart by laundering
            // every address through an empty asm - a v_mov + v_add per read, 64 vector instructions per tile.
            const char* vrow = Vs(buf) + ql * 128;
            const int vsw = (ql >> 1) & 15;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uint4 pb;
                if (s < 2) {
                    pb = make_uint4(pack16x2<T>(sc0[8 * s + 0], sc0[8 * s + 1]), pack16x2<T>(sc0[8 * s + 2], sc0[8 * s + 3]),
                                    pack16x2<T>(sc0[8 * s + 4], sc0[8 * s + 5]), pack16x2<T>(sc0[8 * s + 6], sc0[8 * s + 7]));
                } else {
                    pb = make_uint4(pack16x2<T>(sc1[8 * (s - 2) + 0], sc1[8 * (s - 2) + 1]), pack16x2<T>(sc1[8 * (s - 2) + 2], sc1[8 * (s - 2) + 3]),
                                    pack16x2<T>(sc1[8 * (s - 2) + 4], sc1[8 * (s - 2) + 5]), pack16x2<T>(sc1[8 * (s - 2) + 6], sc1[8 * (s - 2) + 7]));
                }
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    // each 8-byte read stays a ds_read_b64 (two 32-lane halves, 64-bank modulus: the layout is conflict-free for it)
                    const fl_u32x2 lo = *(fl_lds_cv64)(vrow + (((4 * s + h) ^ vsw) << 3) + i * 4096);
                    const fl_u32x2 hi = *(fl_lds_cv64)(vrow + (((4 * s + 2 + h) ^ vsw) << 3) + i * 4096);
                    o[i] = mfma32<T>(make_uint4(lo.x, lo.y, hi.x, hi.y), pb, o[i]);
                }
            }
        }
        if (t + 1 < t1) { store_k(buf ^ 1); store_v(buf ^ 1); }
        __syncthreads();
    }

    // ---- normalise and store: lane = query row, registers 4g..4g+3 of tile i are d = 32i + 8g + 4h .. +3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int qrow = qw0 + ql;
    if constexpr (std::is_same<T, bf16>::value && D == 128) {
        if (sp.q8 != nullptr && sp.nsplit == 1) {       // workgroup-uniform
            // the lane's 64 dims (the partner lane ^ 32 holds the other 64), rounded to bf16 as the plain store would
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    o[i][e] = __uint_as_float(pack_bf16x2(o[i][e] * inv, 0.f) << 16);
                    amax = fmaxf(amax, fabsf(o[i][e]));
                }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            if (qrow < q_len) {
                uint8_t* qrow8 = sp.q8 + ((size_t)qrow * hq + head) * D;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<uint32_t*>(qrow8 + i * 32 + 8 * g + 4 * h) =
                            pack_fp8x4(o[i][4 * g] / sc, o[i][4 * g + 1] / sc, o[i][4 * g + 2] / sc, o[i][4 * g + 3] / sc);
                if (h == 0) sp.q8s[(size_t)qrow * hq + head] = sc;
            }
            return;
        }
    }
    if (qrow < q_len) {
        T* orow = sp.nsplit > 1 ? reinterpret_cast<T*>(sp.o) + (((size_t)split * q_len + qrow) * hq + head) * D
                                : out + (size_t)head * sd.oh + (size_t)qrow * sd.os;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack16x2<T>(o[i][4 * g] * inv, o[i][4 * g + 1] * inv);
                w.y = pack16x2<T>(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + i * 32 + 8 * g + 4 * h) = w;
            }
        if (sp.nsplit > 1 && h == 0) {
            float* ml = sp.ml + (((size_t)split * hq + head) * q_len + qrow) * 2;
            ml[0] = m_run;
            ml[1] = l_tot;
        }
    }
}

// out[q][head][:] = sum_s w_s o_s / sum_s w_s,  w_s = l_s 2^(m_s - max m): one thread per 8 output elements
template <class T, int D>
__global__ __launch_bounds__(256) void flash_merge_kernel(const float* ml, const T* po, T* out, int hq, int q_len, int nsplit, long long oh, long long os,
                                                          uint8_t* q8 = nullptr, float* q8s = nullptr) {
    constexpr int CPR = D / 8;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nrow = (size_t)q_len * hq;
    if (gid >= nrow * CPR) return;          // nrow * CPR is a multiple of 16: the 16 lanes of a (row, head) stay together
    const size_t rowi = gid / CPR;
    const int c = (int)(gid % CPR), qrow = (int)(rowi / hq), head = (int)(rowi % hq);
    float mstar = -INFINITY;
    for (int s2 = 0; s2 < nsplit; ++s2) mstar = fmaxf(mstar, ml[(((size_t)s2 * hq + head) * q_len + qrow) * 2]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, den = 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2) {
        const float* r = ml + (((size_t)s2 * hq + head) * q_len + qrow) * 2;
        const float w = (r[1] > 0.f) ? r[1] * __builtin_amdgcn_exp2f(r[0] - mstar) : 0.f;
        if (w > 0.f) {
            Vec<T> v;
            v.load(po + (((size_t)s2 * q_len + qrow) * hq + head) * D + c * 8);
            float f[8];
            v.to_float(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, f[j], acc[j]);
            den += w;
        }
    }
    const float inv = den > 0.f ? 1.f / den : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    Vec<T> v;
    v.from_float(acc);
    if constexpr (std::is_same<T, bf16>::value && D == 128) {
        if (q8 != nullptr) {                // see FlashSplit::q8: 16 lanes x 8 dims = one (row, head) = one scale block
            v.to_float(acc);
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(acc[j]));
            amax = group16_max(amax);
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            uint2 o8;
            o8.x = pack_fp8x4(acc[0] / sc, acc[1] / sc, acc[2] / sc, acc[3] / sc);
            o8.y = pack_fp8x4(acc[4] / sc, acc[5] / sc, acc[6] / sc, acc[7] / sc);
            *reinterpret_cast<uint2*>(q8 + rowi * D + c * 8) = o8;
            if (c == 0) q8s[rowi] = sc;
            return;
        }
    }
    v.store(out + (size_t)head * oh + (size_t)qrow * os + c * 8);
}

// V [Hkv][kv][D] (element strides kh, ks) -> V^T [Hkv][D][kv_pad], zero beyond kv_len.  One workgroup per
// (64-position tile, kv head); the tile goes through LDS so both sides move 16-byte chunks.
template <class T, int D>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* v, T* vt, int kv_len, int kv_pad, long long kh, long long ks) {
    __shared__ uint16_t tile[64][D + 2];
    const int kv0 = blockIdx.x * 64, head = blockIdx.y;
    const T* vh = v + (size_t)head * kh;
    constexpr int NC = D / 8;
    for (int c = threadIdx.x; c < 64 * NC; c += 256) {
        const int r = c / NC, kc = c % NC;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (kv0 + r < kv_len) x = *reinterpret_cast<const uint4*>(vh + (size_t)(kv0 + r) * ks + kc * 8);
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[r][kc * 8 + 2 * j] = (uint16_t)(w[j] & 0xFFFFu);
            tile[r][kc * 8 + 2 * j + 1] = (uint16_t)(w[j] >> 16);
        }
    }
    __syncthreads();
    T* oh = vt + (size_t)head * D * kv_pad;
    for (int c = threadIdx.x; c < D * 8; c += 256) {
        const int d = c >> 3, kc = c & 7;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[kc * 8 + 2 * j][d] | ((uint32_t)tile[kc * 8 + 2 * j + 1][d] << 16);
        *reinterpret_cast<uint4*>(oh + (size_t)d * kv_pad + kv0 + kc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <class T, int D>
static pgk_status flash_launch(const T* q, const T* k, const T* v, T* out, int hq, int hkv, int q_len, int kv_len, float scale,
                               const FlashStrides& sd, hipStream_t st, uint8_t* q8 = nullptr, float* q8s = nullptr) {
    const int kv_pad = ceil_div(kv_len, 64) * 64;
    const int nqt = ceil_div(q_len, FL_BQ);
    // KV runs per query tile: enough workgroups for two per CU (the kernel's occupancy), never more runs than the
    // shortest useful run of two KV tiles allows for the heaviest query tile
    int nsplit = 1;
    {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const long long wgs = (long long)nqt * hq;
        while (nsplit < 4 && wgs * nsplit < 2LL * cus && ceil_div(kv_len, FL_BKV) >= 4 * (nsplit * 2)) nsplit *= 2;
    }
    const size_t vt_bytes = (size_t)hkv * D * kv_pad * sizeof(T);
    const size_t po_bytes = nsplit > 1 ? (size_t)nsplit * q_len * hq * D * sizeof(T) : 0;
    const size_t ml_bytes = nsplit > 1 ? (size_t)nsplit * hq * q_len * 2 * sizeof(float) : 0;
    void* ws = nullptr;
    if (pgk_status r = pgk_malloc(&ws, vt_bytes + po_bytes + ml_bytes + 512)) return r;
    char* base = (char*)ws;
    T* vt = (T*)base;
    FlashSplit sp{nsplit, nullptr, nullptr, q8, q8s};
    if (nsplit > 1) {
        sp.o = base + ((vt_bytes + 255) & ~(size_t)255);
        sp.ml = (float*)((char*)sp.o + ((po_bytes + 255) & ~(size_t)255));
    }
    transpose_v_kernel<T, D><<<dim3(kv_pad / 64, hkv), 256, 0, st>>>(v, vt, kv_len, kv_pad, sd.kh, sd.ks);
    constexpr size_t LDS = 2 * (size_t)(64 * D * 2) + 2 * (size_t)(D * 128);
    static bool attr_done = false;
    if (LDS > 48 * 1024 && !attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_fwd_kernel<T, D>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    flash_fwd_kernel<T, D><<<nqt * hq * nsplit, FL_THREADS, LDS, st>>>(q, k, (const T*)vt, out, hq, hkv, q_len, kv_len, kv_pad,
                                                                      scale * 1.4426950408889634f, sd, sp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && nsplit > 1) {
        const size_t work = (size_t)q_len * hq * (D / 8);
        flash_merge_kernel<T, D><<<(unsigned)((work + 255) / 256), 256, 0, st>>>(sp.ml, (const T*)sp.o, out, hq, q_len, nsplit, sd.oh, sd.os, q8, q8s);
        e = hipGetLastError();
    }
    pgk_free(ws);   // stream-ordered reuse: later work on this stream runs after the kernels above
    PGK_CHECK_HIP(e);
    return PGK_OK;
}

// entry used by ops_attention.hip's dispatcher; dt16: 0 = bf16, 1 = f16
pgk_status flash_prefill(const void* q, const void* k, const void* v, void* out, int hq, int hkv, int q_len, int kv_len, int d,
                         float scale, long long qh, long long qs, long long kh, long long ks, long long oh, long long os,
                         int dt16, hipStream_t st) {
    const FlashStrides sd{qh, qs, kh, ks, oh, os};
    if (dt16 == 0) {
        if (d == 128) return flash_launch<bf16, 128>((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, hq, hkv, q_len, kv_len, scale, sd, st);
        return flash_launch<bf16, 64>((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, hq, hkv, q_len, kv_len, scale, sd, st);
    }
    if (d == 128) return flash_launch<f16, 128>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, hq, hkv, q_len, kv_len, scale, sd, st);
    return flash_launch<f16, 64>((const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, hq, hkv, q_len, kv_len, scale, sd, st);
}

// engine entry (fp8 x fp8 prefill, bf16, head_dim 128, q_len > 128): causal attention whose result leaves as the o_proj's fp8
// operand - codes [q_len][hq * 128] + scales [q_len][hq] - instead of bf16 rows (FlashSplit::q8)
pgk_status flash_prefill_q8(const void* q, const void* k, const void* v, uint8_t* q8, float* q8s, int hq, int hkv, int q_len, int kv_len,
                            float scale, long long qh, long long qs, long long kh, long long ks, hipStream_t st) {
    PGK_REQUIRE(q8 && q8s && q_len > 128, "flash_prefill_q8: needs output buffers and q_len > 128 (got %d)", q_len);
    const FlashStrides sd{qh, qs, kh, ks, 0, 0};
    return flash_launch<bf16, 128>((const bf16*)q, (const bf16*)k, (const bf16*)v, nullptr, hq, hkv, q_len, kv_len, scale, sd, st, q8, q8s);
}

}  // namespace pgk
