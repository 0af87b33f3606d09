// Weight-streaming GEMV core for gfx950, shared by the op-level GEMV (ops_gemv.hip) and
// the fused decode kernels (engine.hip).
//
//   y[m][n] = sum_k x[m][k] * W[n][k]        W is [N,K] row-major (PyTorch [out,in]), M small.
//
// Design (wave64, HBM-bound):
//   * the activation rows x[M][K] are staged ONCE per workgroup in LDS; weights are read
//     exactly once from HBM, straight to VGPRs (no LDS round trip for a read-once operand),
//     16 bytes per lane = 1 KiB per wave-instruction, fully coalesced along K, non-temporal;
//   * one wave owns R consecutive output rows at a time: R independent 16-B loads in flight
//     per lane per k-step, and one LDS read of the x fragment is reused by the R rows;
//   * fp32 accumulation; cross-lane reduction by 6 xor-shuffles per output.
// The reference's kernel (native/ops/matmul/gemv/bf16_bf16/sm120/bf16_opt.cuh:56-231) is
// "one 32-lane warp per output row, A in shared memory"; this is a re-derivation for
// 64-lane waves, not a transliteration.
#pragma once

#include <type_traits>

#include "pgk_device.hip.h"

namespace pgk {

struct fp8e4m3 { uint8_t b; };

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 load_nt16(const void* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- 16 bytes of weights -> NW floats -----------------------------------------------------
template <class WT> struct WTraits;
template <> struct WTraits<bf16> {
    static constexpr int NW = 8;
    __device__ static __forceinline__ void decode(const uint4& r, float (&f)[8]) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
};
template <> struct WTraits<f16> {
    static constexpr int NW = 8;
    __device__ static __forceinline__ void decode(const uint4& r, float (&f)[8]) {
        Vec<f16> v;
        v.raw = r;
        v.to_float(f);
    }
};
template <> struct WTraits<float> {
    static constexpr int NW = 4;
    __device__ static __forceinline__ void decode(const uint4& r, float (&f)[4]) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
        f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
};
// OCP e4m3 (gfx950's native fp8; MI300's fnuz is a different encoding): v_cvt_pk_f32_fp8.
template <> struct WTraits<fp8e4m3> {
    static constexpr int NW = 16;
    __device__ static __forceinline__ void decode(const uint4& r, float (&f)[16]) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
            const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
            f[4 * i] = lo.x; f[4 * i + 1] = lo.y; f[4 * i + 2] = hi.x; f[4 * i + 3] = hi.y;
        }
    }
};

// ---- NW consecutive activations from LDS -> floats ----------------------------------------
template <class XT, int NW> struct XLoad;
template <int NW> struct XLoad<float, NW> {
    __device__ static __forceinline__ void load(const float* xs, float (&f)[NW]) {
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(xs + 4 * i);
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
    }
};
template <int NW> struct XLoad<bf16, NW> {
    __device__ static __forceinline__ void load(const bf16* xs, float (&f)[NW]) {
        if constexpr (NW == 4) {
            const uint2 v = *reinterpret_cast<const uint2*>(xs);
            f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
            f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
        } else {
#pragma unroll
            for (int i = 0; i < NW / 8; ++i) {
                float t[8];
                WTraits<bf16>::decode(*reinterpret_cast<const uint4*>(xs + 8 * i), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[8 * i + j] = t[j];
            }
        }
    }
};
template <int NW> struct XLoad<f16, NW> {
    __device__ static __forceinline__ void load(const f16* xs, float (&f)[NW]) {
        if constexpr (NW == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = static_cast<float>(xs[j].v);
        } else {
#pragma unroll
            for (int i = 0; i < NW / 8; ++i) {
                float t[8];
                WTraits<f16>::decode(*reinterpret_cast<const uint4*>(xs + 8 * i), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[8 * i + j] = t[j];
            }
        }
    }
};

// bf16 x bf16: v_dot2c_f32_bf16 consumes two packed pairs per instruction (products exact in fp32, fp32
// accumulate) - 4 instructions per 16-byte chunk instead of 8 unpacks + 8 FMAs; this is what keeps the
// M >= 4 batched GEMV from going VALU-bound.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8_bf16(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.x), __builtin_bit_cast(bf16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.y), __builtin_bit_cast(bf16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.z), __builtin_bit_cast(bf16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.w), __builtin_bit_cast(bf16x2_t, x.w), acc, false);
    return acc;
}

// Partial dot products of R weight rows against M activation rows over the K range this lane
// owns (k = lane*NW + i*64*NW).  wrow[r] points at row r's first element; xs is x[M][ldx] in LDS.
// Results are per-lane partials: the caller reduces across the wave.
template <class WT, class XT, int M, int R>
__device__ __forceinline__ void gemv_rows(const WT* const (&wrow)[R], const XT* xs, int ldx, int K, int lane,
                                          float (&acc)[R][M]) {
    constexpr int NW = WTraits<WT>::NW;
    if constexpr (std::is_same<WT, bf16>::value && std::is_same<XT, bf16>::value) {
#pragma unroll 2
        for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
            uint4 raw[R], xr[M];
#pragma unroll
            for (int r = 0; r < R; ++r) raw[r] = load_nt16(wrow[r] + k0);
#pragma unroll
            for (int m = 0; m < M; ++m) xr[m] = *reinterpret_cast<const uint4*>(xs + (size_t)m * ldx + k0);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) acc[r][m] = dot8_bf16(raw[r], xr[m], acc[r][m]);
        }
        return;
    }
#pragma unroll 2
    for (int k0 = lane * NW; k0 < K; k0 += 64 * NW) {
        uint4 raw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) raw[r] = load_nt16(wrow[r] + k0);
        float xf[M][NW];
#pragma unroll
        for (int m = 0; m < M; ++m) XLoad<XT, NW>::load(xs + (size_t)m * ldx + k0, xf[m]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[NW];
            WTraits<WT>::decode(raw[r], wf);
#pragma unroll
            for (int m = 0; m < M; ++m) {
#pragma unroll
                for (int j = 0; j < NW; ++j) acc[r][m] = fmaf(wf[j], xf[m][j], acc[r][m]);
            }
        }
    }
}

// fp8 variant with one bf16 scale per 128x128 weight block: scale row pointer per weight row,
// index k0/128.  (16 | 128, so a lane's 16 codes never straddle a block.)
template <class XT, int M, int R>
__device__ __forceinline__ void gemv_rows_fp8(const fp8e4m3* const (&wrow)[R], const bf16* const (&srow)[R],
                                              const XT* xs, int ldx, int K, int lane, float (&acc)[R][M]) {
    constexpr int NW = 16;
#pragma unroll 2
    for (int k0 = lane * NW; k0 < K; k0 += 64 * NW) {
        uint4 raw[R];
        float sc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            raw[r] = load_nt16(wrow[r] + k0);
            sc[r] = to_f(srow[r][k0 >> 7]);
        }
        float xf[M][NW];
#pragma unroll
        for (int m = 0; m < M; ++m) XLoad<XT, NW>::load(xs + (size_t)m * ldx + k0, xf[m]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[NW];
            WTraits<fp8e4m3>::decode(raw[r], wf);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < NW; ++j) p = fmaf(wf[j], xf[m][j], p);
                acc[r][m] = fmaf(sc[r], p, acc[r][m]);
            }
        }
    }
}

}  // namespace pgk
