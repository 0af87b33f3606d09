// fp8 (OCP e4m3) x fp8 MFMA GEMM with 128-wide block scales, and the quantisers that feed it (config 5:
// Llama-3-8B-shape prefill).
//
//   C[m][n] = sum_kb  sa[m][kb] * sw[n/128][kb] * ( sum_{k in block kb} A8[m][k] * W8[n][k] )
//
//   A8 [M,K]   e4m3 codes, sa [M, K/128] fp32   - activations, one scale per row per 128 k (quantised on the fly
//                                                 by pgk_quantize_fp8_rows)
//   W8 [N,K]   e4m3 codes, sw [N/128, K/128] bf16 - the LinearFP8 weight layout (src/pygpukit/llm/layers/linear.py:149-160)
//
// The inner sum of one 128-k block is ONE v_mfma_f32_16x16x128_f8f6f4 per 16x16 output tile (fp8 products are
// exact in fp32; 2x the bf16 MFMA rate), issued with a zero accumulator; its result is folded into the running
// fp32 accumulator with the block's scale product by 4 VALU FMAs.  The MX hardware scale operand is not used:
// it only takes power-of-two (E8M0) scales and the checkpoint's scales are arbitrary bf16.
// Both operands use the same lane -> k assignment (lane l holds the 32 bytes k = 32*(l>>4) .. +31 of row l&15),
// so the contraction is over matching k whatever order the hardware walks them in.
//
// The reference reaches CUTLASS for this (src/pygpukit/ops/matmul/fp8.py:270-343, native absent from the
// checkout: parity for this op is pinned on the oracle's restatement of the formula above).

#include "gemv_core.hip.h"
#include "gemm_epilogues.hip.h"
#include "pgk_internal.h"

namespace pgk {

typedef int i32x8_q __attribute__((ext_vector_type(8)));
typedef float f32x4_q __attribute__((ext_vector_type(4)));

constexpr int F8_BM = 128, F8_BN = 128, F8_THREADS = 256;
constexpr int F8_TILE = 128 * 128;   // bytes of one operand tile: 128 rows x 128 fp8

// byte offset of 16-byte chunk kc (0..7) of row `row` in a [128][128 B] tile
__device__ __forceinline__ int f8_off(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }

template <int EPI>   // 0: bf16 C store; 1: fp32 C += (the engine's residual stream)
__global__ __launch_bounds__(F8_THREADS) void gemm_fp8_kernel(const uint8_t* A, const float* sa, const uint8_t* W,
                                                              const bf16* sw, void* Cv, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A[2] | B[2] | scale[2][128] f32
    auto As = [&](int buf) -> char* { return smem + buf * F8_TILE; };
    auto Bs = [&](int buf) -> char* { return smem + 2 * F8_TILE + buf * F8_TILE; };
    auto Ss = [&](int buf) -> float* { return reinterpret_cast<float*>(smem + 4 * F8_TILE) + buf * F8_BM; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, q = lane >> 4;
    const int m0 = blockIdx.y * F8_BM, n0 = blockIdx.x * F8_BN;
    const int KB = K >> 7;

    // staging registers: 4 + 4 chunks of 16 B, and this thread's row scale.  Named scalars, not arrays: at this
    // register pressure hipcc leaves a staging ARRAY in scratch memory (a load-wait-scratch_store per chunk).
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float rs_a;
    bf16 rs_w;   // raw: multiplied in store_tiles, touching them at load time would drain the whole load queue
    const int srow = tid >> 3, skc = tid & 7;
    const uint8_t* a_p0 = A + (size_t)min(m0 + srow, M - 1) * K + skc * 16;
    const uint8_t* a_p1 = A + (size_t)min(m0 + 32 + srow, M - 1) * K + skc * 16;
    const uint8_t* a_p2 = A + (size_t)min(m0 + 64 + srow, M - 1) * K + skc * 16;
    const uint8_t* a_p3 = A + (size_t)min(m0 + 96 + srow, M - 1) * K + skc * 16;
    const uint8_t* b_p0 = W + (size_t)min(n0 + srow, N - 1) * K + skc * 16;
    const uint8_t* b_p1 = W + (size_t)min(n0 + 32 + srow, N - 1) * K + skc * 16;
    const uint8_t* b_p2 = W + (size_t)min(n0 + 64 + srow, N - 1) * K + skc * 16;
    const uint8_t* b_p3 = W + (size_t)min(n0 + 96 + srow, N - 1) * K + skc * 16;
    const float* sa_p = sa + (size_t)min(m0 + (tid & 127), M - 1) * KB;
    const bf16* sw_row = sw + (size_t)blockIdx.x * KB;
    const int st_off = f8_off(srow, skc);   // rows srow + 32 i share the swizzle (32 % 8 == 0)
    auto load_tiles = [&](int kt) {
        const size_t kb = (size_t)kt * 128;
        ra0 = *reinterpret_cast<const uint4*>(a_p0 + kb); ra1 = *reinterpret_cast<const uint4*>(a_p1 + kb);
        ra2 = *reinterpret_cast<const uint4*>(a_p2 + kb); ra3 = *reinterpret_cast<const uint4*>(a_p3 + kb);
        rb0 = *reinterpret_cast<const uint4*>(b_p0 + kb); rb1 = *reinterpret_cast<const uint4*>(b_p1 + kb);
        rb2 = *reinterpret_cast<const uint4*>(b_p2 + kb); rb3 = *reinterpret_cast<const uint4*>(b_p3 + kb);
        rs_a = sa_p[kt];
        rs_w = sw_row[kt];
    };
    auto store_tiles = [&](int buf) {
        char* a = As(buf) + st_off;
        char* b = Bs(buf) + st_off;
        *reinterpret_cast<uint4*>(a) = ra0; *reinterpret_cast<uint4*>(a + 32 * 128) = ra1;
        *reinterpret_cast<uint4*>(a + 64 * 128) = ra2; *reinterpret_cast<uint4*>(a + 96 * 128) = ra3;
        *reinterpret_cast<uint4*>(b) = rb0; *reinterpret_cast<uint4*>(b + 32 * 128) = rb1;
        *reinterpret_cast<uint4*>(b + 64 * 128) = rb2; *reinterpret_cast<uint4*>(b + 96 * 128) = rb3;
        if (tid < F8_BM) Ss(buf)[tid] = rs_a * to_f(rs_w);
    };

    f32x4_q acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_q{0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < KB; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KB) load_tiles(kt + 1);
        i32x8_q fa[4], fb[4];
        f32x4_q sc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ar = wm * 64 + i * 16 + (lane & 15), br = wn * 64 + i * 16 + (lane & 15);
            const uint4 a0 = *reinterpret_cast<const uint4*>(As(buf) + f8_off(ar, 2 * q));
            const uint4 a1 = *reinterpret_cast<const uint4*>(As(buf) + f8_off(ar, 2 * q + 1));
            const uint4 b0 = *reinterpret_cast<const uint4*>(Bs(buf) + f8_off(br, 2 * q));
            const uint4 b1 = *reinterpret_cast<const uint4*>(Bs(buf) + f8_off(br, 2 * q + 1));
            fa[i] = i32x8_q{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
            fb[i] = i32x8_q{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
            // C rows of tile i held by this lane: q*4 .. q*4+3
            sc[i] = *reinterpret_cast<const f32x4_q*>(Ss(buf) + wm * 64 + i * 16 + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_q t = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], f32x4_q{0.f, 0.f, 0.f, 0.f},
                                                                                  0, 0, 0, 0, 0, 0);
                acc[i][j] += t * sc[i];
            }
        if (kt + 1 < KB) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // C/D map of the 16x16 MFMA shapes: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + q * 4 + r;
                if (row < M && col < N) {
                    if constexpr (EPI == 0) reinterpret_cast<bf16*>(Cv)[(size_t)row * N + col] = from_f<bf16>(acc[i][j][r]);
                    else reinterpret_cast<float*>(Cv)[(size_t)row * N + col] += acc[i][j][r];
                }
            }
        }
}

// ---- quantisers --------------------------------------------------------------------------------
// (pack_fp8x4: RNE f32 -> e4m3, pgk_device.hip.h)

// Activations: one scale per row per 128 k.  16 lanes x 8 elements cover one (row, block); scale = absmax/448
// (1 when the block is all zero), codes = RNE(x / scale).
template <class T>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const T* x, uint8_t* out, float* scale, long long nblocks, int K) {
    const long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, block) index
    const int sub = threadIdx.x & 15, KB = K >> 7;
    const long long gg = g < nblocks ? g : nblocks - 1;
    const long long row = gg / KB;
    const int kb = (int)(gg % KB);
    const size_t off = (size_t)row * K + (size_t)kb * 128 + sub * 8;
    float f[8];
    if constexpr (sizeof(T) == 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(x + off), v1 = *reinterpret_cast<const float4*>(x + off + 4);
        f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
    } else {
        Vec<T> v;
        v.load(x + off);
        v.to_float(f);
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    amax = group16_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] / s;
    if (g >= nblocks) return;
    uint2 o;
    o.x = pack_fp8x4(f[0], f[1], f[2], f[3]);
    o.y = pack_fp8x4(f[4], f[5], f[6], f[7]);
    *reinterpret_cast<uint2*>(out + off) = o;
    if (sub == 0) scale[gg] = s;
}

// Weights: one bf16 scale per 128x128 block (the oracle's synthetic quantiser, oracle/cpu_ref.py
// quantize_fp8_e4m3_block): scale = bf16(absmax/448), codes = RNE(w / scale).  One workgroup per block.
__global__ __launch_bounds__(256) void quantize_blocks_kernel(const bf16* w, uint8_t* out, bf16* scale, int N, int K) {
    __shared__ float red[16];
    const int nb = blockIdx.y, kb = blockIdx.x, KB = K >> 7;
    const int sub = threadIdx.x & 15, r0 = threadIdx.x >> 4;   // 16 rows per pass, 8 passes
    float f[8][8];
    float amax = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = min(nb * 128 + p * 16 + r0, N - 1);
        Vec<bf16> v;
        v.load(w + (size_t)row * K + (size_t)kb * 128 + sub * 8);
        v.to_float(f[p]);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, nb * 128 + p * 16 + r0 < N ? fabsf(f[p][j]) : 0.f);
    }
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bf16 sb = from_f<bf16>(amax > 0.f ? amax / 448.0f : 1.0f);
    const float s = to_f(sb);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = nb * 128 + p * 16 + r0;
        if (row >= N) continue;
        uint2 o;
        o.x = pack_fp8x4(f[p][0] / s, f[p][1] / s, f[p][2] / s, f[p][3] / s);
        o.y = pack_fp8x4(f[p][4] / s, f[p][5] / s, f[p][6] / s, f[p][7] / s);
        *reinterpret_cast<uint2*>(out + (size_t)row * K + (size_t)kb * 128 + sub * 8) = o;
    }
    if (threadIdx.x == 0) scale[(size_t)nb * KB + kb] = sb;
}

pgk_status gemm256_fp8_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, void* c, bool accum_f32, int M, int N,
                          int K, hipStream_t st);
pgk_status gemm256_fp8_swiglu_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, uint8_t* q_out, float* s_out, int M,
                                 int I, int K, hipStream_t st);
bool want_gemm256(int M, int N);      // ops_gemm.hip

pgk_status gemm256_fp8_qkv_heads_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, bf16* qkv, int M, int N, int K,
                                    const QkvHeadArgs& hd, hipStream_t st);
// QKV projection with per-head norm + RoPE + cache write as its epilogue (256-tile kernel: whole tiles, enough of them)
bool gemm_fp8_qkv_heads_ok(int M, int N, int K) { return K % 128 == 0 && M % 256 == 0 && N % 256 == 0 && want_gemm256(M, N); }
pgk_status gemm_fp8_qkv_heads_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, bf16* qkv, int M, int N, int K,
                                 const QkvHeadArgs& hd, hipStream_t st) {
    PGK_REQUIRE(gemm_fp8_qkv_heads_ok(M, N, K), "gemm_fp8_qkv_heads: M=%d N=%d K=%d outside the fused kernel's shapes", M, N, K);
    return gemm256_fp8_qkv_heads_nt(a, sa, w, sw, qkv, M, N, K, hd, st);
}

// gate / up projection with the SwiGLU + e4m3 quantisation epilogue (ops_gemm256.hip): whole 256-row tiles, enough of them
bool gemm_fp8_swiglu_ok(int M, int I, int K) { return K % 128 == 0 && I % 128 == 0 && M % 256 == 0 && want_gemm256(M, 2 * I); }
pgk_status gemm_fp8_swiglu_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, uint8_t* q_out, float* s_out, int M, int I,
                              int K, hipStream_t st) {
    PGK_REQUIRE(gemm_fp8_swiglu_ok(M, I, K), "gemm_fp8_swiglu: M=%d I=%d K=%d outside the fused kernel's shapes", M, I, K);
    return gemm256_fp8_swiglu_nt(a, sa, w, sw, q_out, s_out, M, I, K, st);
}

// internal entry used by the engine's fp8-activation prefill
pgk_status gemm_fp8_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, void* c, bool accum_f32, int M,
                       int N, int K, hipStream_t st) {
    PGK_REQUIRE(M >= 1 && N >= 1 && K >= 128 && K % 128 == 0, "gemm_fp8: K=%d must be a positive multiple of 128 (M=%d N=%d)", K, M, N);
    // enough 256 x 256 tiles to fill the chip: the LDS-DMA structure (ops_gemm256.hip)
    if (want_gemm256(M, N) && M % 256 == 0 && N % 256 == 0) return gemm256_fp8_nt(a, sa, w, sw, c, accum_f32, M, N, K, st);
    constexpr size_t LDS = 4 * (size_t)F8_TILE + 2 * F8_BM * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    dim3 grid(ceil_div(N, F8_BN), ceil_div(M, F8_BM));
    if (accum_f32) gemm_fp8_kernel<1><<<grid, F8_THREADS, LDS, st>>>(a, sa, w, sw, c, M, N, K);
    else gemm_fp8_kernel<0><<<grid, F8_THREADS, LDS, st>>>(a, sa, w, sw, c, M, N, K);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

pgk_status quantize_fp8_rows_bf16(const bf16* x, uint8_t* out, float* scale, int M, int K, hipStream_t st) {
    const long long nblocks = (long long)M * (K >> 7);
    quantize_rows_kernel<bf16><<<(unsigned)ceil_div(nblocks, 16), 256, 0, st>>>(x, out, scale, nblocks, K);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_gemm_fp8_nt(const uint8_t* a_fp8, const float* a_scale, const uint8_t* w_fp8_nk, const void* w_scale, void* c,
                           int m, int n, int k, pgk_stream s) {
    PGK_REQUIRE(a_fp8 && a_scale && w_fp8_nk && w_scale && c, "pgk_gemm_fp8_nt: null argument");
    return gemm_fp8_nt(a_fp8, a_scale, w_fp8_nk, (const bf16*)w_scale, c, false, m, n, k, resolve_stream(s));
}

pgk_status pgk_quantize_fp8_rows(const void* x, uint8_t* out_fp8, float* out_scale, int m, int k, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && out_fp8 && out_scale, "pgk_quantize_fp8_rows: null argument");
    PGK_REQUIRE(m >= 1 && k >= 128 && k % 128 == 0, "pgk_quantize_fp8_rows: k=%d must be a positive multiple of 128 (m=%d)", k, m);
    hipStream_t st = resolve_stream(s);
    const long long nblocks = (long long)m * (k >> 7);
    const unsigned grid = (unsigned)ceil_div(nblocks, 16);
    switch (dt) {
        case PGK_BF16: quantize_rows_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)x, out_fp8, out_scale, nblocks, k); break;
        case PGK_F16: quantize_rows_kernel<f16><<<grid, 256, 0, st>>>((const f16*)x, out_fp8, out_scale, nblocks, k); break;
        case PGK_F32: quantize_rows_kernel<float><<<grid, 256, 0, st>>>((const float*)x, out_fp8, out_scale, nblocks, k); break;
        default: return set_error(PGK_ERR_UNSUPPORTED, "pgk_quantize_fp8_rows: dtype %d", (int)dt);
    }
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

pgk_status pgk_quantize_fp8_blocks(const void* w_bf16, uint8_t* out_fp8, void* out_scale_bf16, int n, int k, pgk_stream s) {
    PGK_REQUIRE(w_bf16 && out_fp8 && out_scale_bf16, "pgk_quantize_fp8_blocks: null argument");
    PGK_REQUIRE(n >= 1 && k >= 128 && k % 128 == 0, "pgk_quantize_fp8_blocks: k=%d must be a positive multiple of 128 (n=%d)", k, n);
    quantize_blocks_kernel<<<dim3(k >> 7, ceil_div(n, 128)), 256, 0, resolve_stream(s)>>>((const bf16*)w_bf16, out_fp8,
                                                                                       (bf16*)out_scale_bf16, n, k);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // extern "C"
