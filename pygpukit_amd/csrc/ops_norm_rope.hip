// RMSNorm / RMSNorm+residual / LayerNorm / RoPE.
// Row norms: one wave per row when the row fits in registers (features <= 64*VEC*MAXV),
// else one 256-thread block per row; fp32 math, one rounding on store
// (reference: native/ops/nn/norm_kernels.cuh:32-584, 32-lane shuffles there, 64 here).

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

constexpr int NORM_WAVES = 4;  // rows per block in the wave-per-row kernels
constexpr int NORM_MAXV = 8;   // 16-byte vectors per lane held in registers

// mode: 0 rmsnorm, 1 rmsnorm(x + residual), 2 layernorm
template <class T, int MODE>
__global__ __launch_bounds__(NORM_WAVES * 64) void norm_wave_kernel(const T* x, const T* res, const T* gamma,
                                                                    const T* beta, T* out, int rows,
                                                                    int features, float eps) {
    constexpr int N = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * features;
    const int nv = features / N;  // features % N == 0 guaranteed by the host
    float v[NORM_MAXV][N];
    float sum = 0.f, sumsq = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
            Vec<T> t;
            t.load(xr + vi * N);
            t.to_float(v[i]);
            if (MODE == 1) {
                Vec<T> r;
                r.load(res + (size_t)row * features + vi * N);
                float rf[N];
                r.to_float(rf);
#pragma unroll
                for (int j = 0; j < N; ++j) v[i][j] += rf[j];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) { sum += v[i][j]; sumsq += v[i][j] * v[i][j]; }
        }
    }
    float mean = 0.f, inv;
    if (MODE == 2) {
        mean = wave_sum(sum) / features;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAXV; ++i)
            if (lane + i * 64 < nv) {
#pragma unroll
                for (int j = 0; j < N; ++j) { const float d = v[i][j] - mean; var += d * d; }
            }
        inv = 1.0f / sqrtf(wave_sum(var) / features + eps);
    } else {
        inv = 1.0f / sqrtf(wave_sum(sumsq) / features + eps);
    }
    T* orow = out + (size_t)row * features;
#pragma unroll
    for (int i = 0; i < NORM_MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
            Vec<T> g;
            g.load(gamma + vi * N);
            float gf[N], o[N];
            g.to_float(gf);
            if (MODE == 2) {
                Vec<T> b;
                b.load(beta + vi * N);
                float bf[N];
                b.to_float(bf);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = (v[i][j] - mean) * inv * gf[j] + bf[j];
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = v[i][j] * inv * gf[j];
            }
            Vec<T> ov;
            ov.from_float(o);
            ov.store(orow + vi * N);
        }
    }
}

// Generic fallback: one block per row, scalar accesses, any feature count.
template <class T, int MODE>
__global__ __launch_bounds__(256) void norm_block_kernel(const T* x, const T* res, const T* gamma, const T* beta,
                                                         T* out, int rows, int features, float eps) {
    __shared__ float scratch[16];
    const int row = blockIdx.x;
    const T* xr = x + (size_t)row * features;
    const T* rr = MODE == 1 ? res + (size_t)row * features : nullptr;
    float sum = 0.f, sumsq = 0.f;
    for (int i = threadIdx.x; i < features; i += blockDim.x) {
        float v = to_f(xr[i]);
        if (MODE == 1) v += to_f(rr[i]);
        sum += v;
        sumsq += v * v;
    }
    float mean = 0.f, inv;
    if (MODE == 2) {
        mean = block_sum(sum, scratch) / features;
        float var = 0.f;
        for (int i = threadIdx.x; i < features; i += blockDim.x) {
            const float d = to_f(xr[i]) - mean;
            var += d * d;
        }
        inv = 1.0f / sqrtf(block_sum(var, scratch) / features + eps);
    } else {
        inv = 1.0f / sqrtf(block_sum(sumsq, scratch) / features + eps);
    }
    T* orow = out + (size_t)row * features;
    for (int i = threadIdx.x; i < features; i += blockDim.x) {
        float v = to_f(xr[i]);
        if (MODE == 1) v += to_f(rr[i]);
        float o = (MODE == 2) ? (v - mean) * inv * to_f(gamma[i]) + to_f(beta[i]) : v * inv * to_f(gamma[i]);
        orow[i] = from_f<T>(o);
    }
}

template <class T, int MODE>
static pgk_status launch_norm(const void* x, const void* res, const void* gamma, const void* beta, void* out,
                              int rows, int features, float eps, hipStream_t st) {
    constexpr int N = Vec<T>::N;
    const bool vec = (features % N == 0) && features <= 64 * N * NORM_MAXV && aligned16(x) && aligned16(out) &&
                     aligned16(gamma) && (MODE != 1 || aligned16(res)) && (MODE != 2 || aligned16(beta)) &&
                     ((size_t)features * sizeof(T)) % 16 == 0;
    if (vec) {
        norm_wave_kernel<T, MODE><<<ceil_div(rows, NORM_WAVES), NORM_WAVES * 64, 0, st>>>(
            (const T*)x, (const T*)res, (const T*)gamma, (const T*)beta, (T*)out, rows, features, eps);
    } else {
        norm_block_kernel<T, MODE><<<rows, 256, 0, st>>>((const T*)x, (const T*)res, (const T*)gamma,
                                                         (const T*)beta, (T*)out, rows, features, eps);
    }
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

// RoPE (rotate-half), in place on q[S,Hq,D] and k[S,Hk,D]; table row per sequence position,
// only columns d < D/2 are read (reference: native/ops/nn/elementwise_kernels.cuh:82-137,253-307).
// One thread handles PAIRS consecutive (x[d], x[d+D/2]) pairs.
template <class T, class TT>
__global__ void rope_kernel(T* q, T* k, const TT* cos_t, const TT* sin_t, int seq, int hq, int hk, int d) {
    const int half = d >> 1;
    const size_t total = (size_t)seq * (hq + hk) * half;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int dd = (int)(i % half);
        const size_t sh = i / half;
        const int h = (int)(sh % (hq + hk));
        const int s = (int)(sh / (hq + hk));
        T* base = (h < hq) ? q + ((size_t)s * hq + h) * d : k + ((size_t)s * hk + (h - hq)) * d;
        const float c = to_f(cos_t[(size_t)s * d + dd]);
        const float sn = to_f(sin_t[(size_t)s * d + dd]);
        const float x0 = to_f(base[dd]);
        const float x1 = to_f(base[dd + half]);
        base[dd] = from_f<T>(x0 * c - x1 * sn);
        base[dd + half] = from_f<T>(x1 * c + x0 * sn);
    }
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_rmsnorm(const void* x, const void* gamma, void* out, int rows, int features, float eps, pgk_dtype dt,
                       pgk_stream s) {
    PGK_REQUIRE(x && gamma && out, "pgk_rmsnorm: null pointer");
    PGK_REQUIRE(rows >= 0 && features > 0, "pgk_rmsnorm: bad shape [%d,%d]", rows, features);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_rmsnorm", return (launch_norm<T, 0>(x, nullptr, gamma, nullptr, out, rows, features, eps, st)));
    return PGK_OK;
}

pgk_status pgk_rmsnorm_residual(const void* x, const void* residual, const void* gamma, void* out, int rows,
                                int features, float eps, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && residual && gamma && out, "pgk_rmsnorm_residual: null pointer");
    PGK_REQUIRE(rows >= 0 && features > 0, "pgk_rmsnorm_residual: bad shape [%d,%d]", rows, features);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_rmsnorm_residual",
                       return (launch_norm<T, 1>(x, residual, gamma, nullptr, out, rows, features, eps, st)));
    return PGK_OK;
}

pgk_status pgk_layernorm(const void* x, const void* gamma, const void* beta, void* out, int rows, int features,
                         float eps, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && gamma && beta && out, "pgk_layernorm: null pointer");
    PGK_REQUIRE(rows >= 0 && features > 0, "pgk_layernorm: bad shape [%d,%d]", rows, features);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_layernorm", return (launch_norm<T, 2>(x, nullptr, gamma, beta, out, rows, features, eps, st)));
    return PGK_OK;
}

pgk_status pgk_rope_inplace(void* q, void* k, const void* cos, const void* sin, int seq, int hq, int hk, int d,
                            pgk_dtype dt, int f32_table, pgk_stream s) {
    PGK_REQUIRE(q && k && cos && sin, "pgk_rope_inplace: null pointer");
    PGK_REQUIRE(seq >= 0 && hq > 0 && hk >= 0 && d > 0 && (d % 2) == 0, "pgk_rope_inplace: bad shape S=%d Hq=%d Hk=%d D=%d",
                seq, hq, hk, d);
    if (!seq) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const size_t total = (size_t)seq * (hq + hk) * (d / 2);
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    PGK_DISPATCH_FLOAT(dt, "pgk_rope_inplace", {
        if (f32_table) rope_kernel<T, float><<<grid, 256, 0, st>>>((T*)q, (T*)k, (const float*)cos, (const float*)sin, seq, hq, hk, d);
        else rope_kernel<T, T><<<grid, 256, 0, st>>>((T*)q, (T*)k, (const T*)cos, (const T*)sin, seq, hq, hk, d);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
