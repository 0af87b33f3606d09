// Elementwise ops: binary add/sub/mul/div (+ in-place), bias add, activations, GLU, casts.
// All HBM-bound: 16-byte accesses per lane, grid capped at 2048 blocks and grid-strided
// (cdna_hip_programming.md Guideline 11/13).  fp32 math, one rounding on store.

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

constexpr int EW_BLOCK = 256;

static inline int ew_grid(size_t work_items) {
    size_t g = (work_items + EW_BLOCK - 1) / EW_BLOCK;
    if (g < 1) g = 1;
    return (int)(g > 2048 ? 2048 : g);
}

__device__ __forceinline__ float binop(float a, float b, int op) {
    switch (op) {
        case 0: return a + b;
        case 1: return a - b;
        case 2: return a * b;
        default: return a / b;
    }
}

// gelu: tanh form with the reference's constants (native/ops/nn/activation_kernels.cuh:110-171)
__device__ __forceinline__ float act_fn(float x, int act) {
    switch (act) {
        case 0: return x / (1.0f + expf(-x));                                              // silu
        case 1: return x * 0.5f * (1.0f + tanhf(0.7978845608f * (x + 0.044715f * x * x * x)));  // gelu
        case 2: return 1.0f / (1.0f + expf(-x));                                           // sigmoid
        case 3: return tanhf(x);
        case 4: { float r = fmaxf(x, 0.f); return r * r; }                                    // relu2
        // native/ops/unary (ops.cuh:60-101): exp log relu sin cos sqrt rsqrt abs neg
        case 5: return expf(x);
        case 6: return logf(x);
        case 7: return fmaxf(x, 0.f);
        case 8: return sinf(x);
        case 9: return cosf(x);
        case 10: return sqrtf(x);
        case 11: return 1.0f / sqrtf(x);
        case 12: return fabsf(x);
        default: return -x;
    }
}

template <class T, bool VECTOR>
__global__ void binary_kernel(const T* a, const T* b, T* c, size_t n, int op) {
    constexpr int N = Vec<T>::N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (VECTOR) {
        const size_t nv = n / N;
        for (size_t i = tid; i < nv; i += stride) {
            Vec<T> va, vb, vc;
            va.load(a + i * N);
            vb.load(b + i * N);
            float fa[N], fb[N], fc[N];
            va.to_float(fa);
            vb.to_float(fb);
#pragma unroll
            for (int j = 0; j < N; ++j) fc[j] = binop(fa[j], fb[j], op);
            vc.from_float(fc);
            vc.store(c + i * N);
        }
        done = nv * N;
    }
    for (size_t i = done + tid; i < n; i += stride) c[i] = from_f<T>(binop(to_f(a[i]), to_f(b[i]), op));
}

template <class T, bool VECTOR>
__global__ void bias_add_kernel(T* out, const T* bias, size_t rows, int features) {
    constexpr int N = Vec<T>::N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VECTOR) {  // features % N == 0
        const int fv = features / N;
        const size_t nv = rows * fv;
        for (size_t i = tid; i < nv; i += stride) {
            const int col = (int)(i % fv) * N;
            Vec<T> vo, vb;
            vo.load(out + i * N);
            vb.load(bias + col);
            float fo[N], fb[N];
            vo.to_float(fo);
            vb.to_float(fb);
#pragma unroll
            for (int j = 0; j < N; ++j) fo[j] += fb[j];
            vo.from_float(fo);
            vo.store(out + i * N);
        }
    } else {
        const size_t n = rows * features;
        for (size_t i = tid; i < n; i += stride) out[i] = from_f<T>(to_f(out[i]) + to_f(bias[i % features]));
    }
}

template <class T, bool VECTOR>
__global__ void act_kernel(const T* x, T* y, size_t n, int act) {
    constexpr int N = Vec<T>::N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (VECTOR) {
        const size_t nv = n / N;
        for (size_t i = tid; i < nv; i += stride) {
            Vec<T> v;
            v.load(x + i * N);
            float f[N];
            v.to_float(f);
#pragma unroll
            for (int j = 0; j < N; ++j) f[j] = act_fn(f[j], act);
            v.from_float(f);
            v.store(y + i * N);
        }
        done = nv * N;
    }
    for (size_t i = done + tid; i < n; i += stride) y[i] = from_f<T>(act_fn(to_f(x[i]), act));
}

template <class T, bool VECTOR>
__global__ void glu_kernel(const T* g, const T* u, T* o, size_t n, int act) {
    constexpr int N = Vec<T>::N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (VECTOR) {
        const size_t nv = n / N;
        for (size_t i = tid; i < nv; i += stride) {
            Vec<T> vg, vu;
            vg.load(g + i * N);
            vu.load(u + i * N);
            float fg[N], fu[N];
            vg.to_float(fg);
            vu.to_float(fu);
#pragma unroll
            for (int j = 0; j < N; ++j) fg[j] = act_fn(fg[j], act) * fu[j];
            vg.from_float(fg);
            vg.store(o + i * N);
        }
        done = nv * N;
    }
    for (size_t i = done + tid; i < n; i += stride) o[i] = from_f<T>(act_fn(to_f(g[i]), act) * to_f(u[i]));
}

// out[r][i] = act(gu[r][i]) * gu[r][inter + i]
template <class T, bool VECTOR>
__global__ void glu_packed_kernel(const T* gu, T* o, size_t rows, int inter, int act) {
    constexpr int N = Vec<T>::N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VECTOR) {  // inter % N == 0
        const int iv = inter / N;
        for (size_t t = tid; t < rows * iv; t += stride) {
            const size_t r = t / iv, c = (t % iv) * N;
            Vec<T> vg, vu;
            vg.load(gu + r * 2 * inter + c);
            vu.load(gu + r * 2 * inter + inter + c);
            float fg[N], fu[N];
            vg.to_float(fg);
            vu.to_float(fu);
#pragma unroll
            for (int j = 0; j < N; ++j) fg[j] = act_fn(fg[j], act) * fu[j];
            vg.from_float(fg);
            vg.store(o + r * inter + c);
        }
    } else {
        for (size_t t = tid; t < rows * inter; t += stride) {
            const size_t r = t / inter, c = t % inter;
            o[t] = from_f<T>(act_fn(to_f(gu[r * 2 * inter + c]), act) * to_f(gu[r * 2 * inter + inter + c]));
        }
    }
}

template <class S, class D>
__global__ void cast_kernel(const S* src, D* dst, size_t n) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // 4 elements per thread per trip; the compiler merges the loads/stores of one trip
    const size_t n4 = n / 4;
    for (size_t i = tid; i < n4; i += stride) {
        S s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = src[i * 4 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[i * 4 + j] = from_f<D>(to_f(s[j]));
    }
    for (size_t i = n4 * 4 + tid; i < n; i += stride) dst[i] = from_f<D>(to_f(src[i]));
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_binary(const void* a, const void* b, void* c, size_t n, int op, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(a && b && c, "pgk_binary: null pointer");
    PGK_REQUIRE(op >= 0 && op <= 3, "pgk_binary: bad op %d", op);
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const bool vec = aligned16(a) && aligned16(b) && aligned16(c);
    PGK_DISPATCH_FLOAT(dt, "pgk_binary", {
        const int grid = ew_grid(vec ? n / Vec<T>::N + 1 : n);
        if (vec) binary_kernel<T, true><<<grid, EW_BLOCK, 0, st>>>((const T*)a, (const T*)b, (T*)c, n, op);
        else binary_kernel<T, false><<<grid, EW_BLOCK, 0, st>>>((const T*)a, (const T*)b, (T*)c, n, op);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_binary_inplace(void* a, const void* b, size_t n, int op, pgk_dtype dt, pgk_stream s) {
    return pgk_binary(a, b, a, n, op, dt, s);
}

pgk_status pgk_bias_add_inplace(void* out, const void* bias, int rows, int features, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(out && bias, "pgk_bias_add_inplace: null pointer");
    PGK_REQUIRE(rows >= 0 && features > 0, "pgk_bias_add_inplace: bad shape [%d,%d]", rows, features);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_bias_add_inplace", {
        const bool vec = aligned16(out) && aligned16(bias) && (features % Vec<T>::N == 0);
        const size_t n = (size_t)rows * features;
        const int grid = ew_grid(vec ? n / Vec<T>::N : n);
        if (vec) bias_add_kernel<T, true><<<grid, EW_BLOCK, 0, st>>>((T*)out, (const T*)bias, rows, features);
        else bias_add_kernel<T, false><<<grid, EW_BLOCK, 0, st>>>((T*)out, (const T*)bias, rows, features);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_activation(const void* x, void* y, size_t n, int act, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && y, "pgk_activation: null pointer");
    PGK_REQUIRE(act >= 0 && act <= 13, "pgk_activation: bad activation %d", act);
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const bool vec = aligned16(x) && aligned16(y);
    PGK_DISPATCH_FLOAT(dt, "pgk_activation", {
        const int grid = ew_grid(vec ? n / Vec<T>::N + 1 : n);
        if (vec) act_kernel<T, true><<<grid, EW_BLOCK, 0, st>>>((const T*)x, (T*)y, n, act);
        else act_kernel<T, false><<<grid, EW_BLOCK, 0, st>>>((const T*)x, (T*)y, n, act);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_glu(const void* gate, const void* up, void* out, size_t n, int act, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(gate && up && out, "pgk_glu: null pointer");
    PGK_REQUIRE(act == 0 || act == 1, "pgk_glu: bad activation %d", act);
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const bool vec = aligned16(gate) && aligned16(up) && aligned16(out);
    PGK_DISPATCH_FLOAT(dt, "pgk_glu", {
        const int grid = ew_grid(vec ? n / Vec<T>::N + 1 : n);
        if (vec) glu_kernel<T, true><<<grid, EW_BLOCK, 0, st>>>((const T*)gate, (const T*)up, (T*)out, n, act);
        else glu_kernel<T, false><<<grid, EW_BLOCK, 0, st>>>((const T*)gate, (const T*)up, (T*)out, n, act);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_glu_packed(const void* gate_up, void* out, int rows, int inter, int act, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(gate_up && out, "pgk_glu_packed: null pointer");
    PGK_REQUIRE(rows >= 0 && inter > 0, "pgk_glu_packed: bad shape [%d, 2*%d]", rows, inter);
    PGK_REQUIRE(act == 0 || act == 1, "pgk_glu_packed: bad activation %d", act);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_glu_packed", {
        const bool vec = aligned16(gate_up) && aligned16(out) && (inter % Vec<T>::N == 0);
        const size_t n = (size_t)rows * inter;
        const int grid = ew_grid(vec ? n / Vec<T>::N : n);
        if (vec) glu_packed_kernel<T, true><<<grid, EW_BLOCK, 0, st>>>((const T*)gate_up, (T*)out, rows, inter, act);
        else glu_packed_kernel<T, false><<<grid, EW_BLOCK, 0, st>>>((const T*)gate_up, (T*)out, rows, inter, act);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_cast(const void* src, pgk_dtype sdt, void* dst, pgk_dtype ddt, size_t n, pgk_stream s) {
    PGK_REQUIRE(src && dst, "pgk_cast: null pointer");
    PGK_REQUIRE(is_float_dtype(sdt) && is_float_dtype(ddt), "pgk_cast: only f32/f16/bf16 (got %d -> %d)", (int)sdt, (int)ddt);
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    if (sdt == ddt) return pgk_memcpy_d2d(dst, src, n * dtype_size(sdt), s);
    const int grid = ew_grid(n / 4 + 1);
#define PGK_CAST_CASE(SD, DD, ST, DT)                                                                      \
    if (sdt == SD && ddt == DD) {                                                                          \
        cast_kernel<ST, DT><<<grid, EW_BLOCK, 0, st>>>((const ST*)src, (DT*)dst, n);                       \
        PGK_LAUNCH_CHECK();                                                                                \
        return PGK_OK;                                                                                     \
    }
    PGK_CAST_CASE(PGK_F32, PGK_BF16, float, bf16)
    PGK_CAST_CASE(PGK_F32, PGK_F16, float, f16)
    PGK_CAST_CASE(PGK_BF16, PGK_F32, bf16, float)
    PGK_CAST_CASE(PGK_F16, PGK_F32, f16, float)
    PGK_CAST_CASE(PGK_BF16, PGK_F16, bf16, f16)
    PGK_CAST_CASE(PGK_F16, PGK_BF16, f16, bf16)
#undef PGK_CAST_CASE
    return set_error(PGK_ERR_INVALID, "pgk_cast: unsupported pair");
}

}  // extern "C"
