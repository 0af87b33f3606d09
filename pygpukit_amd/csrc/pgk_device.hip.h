// Device-side helpers for gfx950 (CDNA4): 64-lane wavefronts, 16-byte vector access,
// bf16/f16 <-> fp32 conversion, wave/block reductions.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgk {

constexpr int WAVE = 64;

struct bf16 { uint16_t bits; };
struct f16 { _Float16 v; };

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
// RNE; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint16_t f_to_bf16_bits(float f) {
    __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, b);
}
// two floats -> one dword of bf16 (lo in bits 0-15): a 2-wide vector conversion lowers to ONE v_cvt_pk_bf16_f32
// (RNE, NaN stays NaN); converting the halves separately and OR-ing them costs four instructions
typedef float pgk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pgk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const pgk_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pgk_bf16x2));
}

template <class T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<f16>(f16 v) { return static_cast<float>(v.v); }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return bf16_bits_to_f(v.bits); }

template <class T> __device__ __forceinline__ T from_f(float f);
template <> __device__ __forceinline__ float from_f<float>(float f) { return f; }
template <> __device__ __forceinline__ f16 from_f<f16>(float f) { return f16{static_cast<_Float16>(f)}; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float f) { return bf16{f_to_bf16_bits(f)}; }

// ---- 16-byte vectors ----------------------------------------------------------------------
// Vec<T>::N elements of T in one 16-byte access: 4 x fp32 or 8 x 16-bit.
template <class T> struct Vec {
    static constexpr int N = 16 / sizeof(T);
    uint4 raw;
    __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    __device__ __forceinline__ void to_float(float (&f)[N]) const;
    __device__ __forceinline__ void from_float(const float (&f)[N]);
};

template <> __device__ __forceinline__ void Vec<float>::to_float(float (&f)[4]) const {
    f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
    f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
}
template <> __device__ __forceinline__ void Vec<float>::from_float(const float (&f)[4]) {
    raw = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ void Vec<bf16>::to_float(float (&f)[8]) const {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
template <> __device__ __forceinline__ void Vec<bf16>::from_float(const float (&f)[8]) {
    raw = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                     pack_bf16x2(f[6], f[7]));
}
template <> __device__ __forceinline__ void Vec<f16>::to_float(float (&f)[8]) const {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] & 0xFFFFu)));
        f[2 * i + 1] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(w[i] >> 16)));
    }
}
template <> __device__ __forceinline__ void Vec<f16>::from_float(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * i]));
        const uint32_t hi = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f[2 * i + 1]));
        w[i] = lo | (hi << 16);
    }
    raw = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- reductions ---------------------------------------------------------------------------
// Cross-lane traffic goes through DPP (register-to-register, a few cycles) instead of ds_bpermute
// (an LDS-pipe round trip per step): a decode kernel's critical path is a chain of these.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i within each 16
constexpr int DPP_ROR8 = 0x128;        // rotate by 8 within each 16 (== xor 8)

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// all-reduce over aligned groups of 8 / 16 lanes
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v = group8_sum(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
// RNE f32 -> OCP e4m3, four codes per dword: v_cvt_pk_fp8_f32 (OCP encoding on gfx950)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return v;
}
template <int LANES> __device__ __forceinline__ float group_sum(float v) {  // LANES in {8,16,32,64}, aligned groups
    if constexpr (LANES == 8) return group8_sum(v);
    v = group16_sum(v);
    if constexpr (LANES == 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LANES == 64) v = readlane_f(v, 0) + readlane_f(v, 16) + readlane_f(v, 32) + readlane_f(v, 48);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
    v = group16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// value held by the lane LANES/2 away inside an aligned group of LANES (rotate-half partner)
template <int LANES> __device__ __forceinline__ float xor_half(float v) {
    if constexpr (LANES == 16) return dpp_f<DPP_ROR8>(v);
    else return __shfl_xor(v, LANES / 2, 64);
}

// One dword through the SCALAR cache, waited for on the spot.  For wave-uniform device state (a sequence's position) that
// hipcc would otherwise fetch with a vector load - possible aliasing with the kernel's own stores keeps it off the scalar
// path - and that a vector load would queue BEHIND every weight / cache load already in flight (vector memory returns in
// order): with the scalar load the value is there ~1 us earlier and nothing else is held up by waiting for it.
// The memory must not be written by this launch.
__device__ __forceinline__ int load_uniform_i32(const int32_t* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// Block-wide sum for blocks of up to 1024 threads (16 waves); `scratch` holds >= 16 floats.
// Every thread gets the result.  Two barriers.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

}  // namespace pgk
