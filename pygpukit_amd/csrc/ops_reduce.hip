// Whole-array reductions, row softmax, axis sums, clamp and where: the remainder of the reference's `ops.basic`
// surface (src/pygpukit/ops/reduction.py:16-300, ops/elementwise.py:254-308; native ops.cuh:92-131).
//
// Every reduction is a fixed two-level tree (per-workgroup partials in fp32, then one workgroup over the partials),
// so a result does not depend on scheduling: same bits on every run.

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

constexpr int RD_BLOCK = 256;
constexpr int RD_MAX_BLOCKS = 1024;

// op: 0 sum, 1 mean (sum here, scaled at the end), 2 max, 3 min
__device__ __forceinline__ float rd_identity(int op) { return op == 2 ? -INFINITY : (op == 3 ? INFINITY : 0.f); }
__device__ __forceinline__ float rd_combine(float a, float b, int op) { return op == 2 ? fmaxf(a, b) : (op == 3 ? fminf(a, b) : a + b); }

__device__ __forceinline__ float rd_block(float v, int op, float* red /* [RD_BLOCK/64] */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = rd_combine(v, __shfl_xor(v, off, 64), op);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)blockDim.x / 64; ++w) r = rd_combine(r, red[w], op);
    return r;
}

template <class T>
__global__ __launch_bounds__(RD_BLOCK) void reduce_partial_kernel(const T* x, float* part, size_t n, int op) {
    __shared__ float red[RD_BLOCK / 64];
    float v = rd_identity(op);
    const size_t stride = (size_t)gridDim.x * RD_BLOCK;
    for (size_t i = (size_t)blockIdx.x * RD_BLOCK + threadIdx.x; i < n; i += stride) v = rd_combine(v, to_f(x[i]), op);
    v = rd_block(v, op, red);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}

template <class T>
__global__ __launch_bounds__(RD_BLOCK) void reduce_final_kernel(const float* part, int nparts, T* out, int op, float inv_n) {
    __shared__ float red[RD_BLOCK / 64];
    float v = rd_identity(op);
    for (int i = threadIdx.x; i < nparts; i += RD_BLOCK) v = rd_combine(v, part[i], op);
    v = rd_block(v, op, red);
    if (threadIdx.x == 0) out[0] = from_f<T>(op == 1 ? v * inv_n : v);
}

// y[r, :] = softmax(x[r, :]) in fp32 (max-subtracted), one workgroup per row
template <class T>
__global__ __launch_bounds__(RD_BLOCK) void softmax_rows_kernel(const T* x, T* y, int n) {
    __shared__ float red[RD_BLOCK / 64];
    const T* xr = x + (size_t)blockIdx.x * n;
    T* yr = y + (size_t)blockIdx.x * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += RD_BLOCK) m = fmaxf(m, to_f(xr[i]));
    m = rd_block(m, 2, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += RD_BLOCK) s += expf(to_f(xr[i]) - m);
    s = rd_block(s, 0, red);
    for (int i = threadIdx.x; i < n; i += RD_BLOCK) yr[i] = from_f<T>(expf(to_f(xr[i]) - m) / s);
}

// axis 1: out[m] = sum_n x[m, n] (a workgroup per row); axis 0: out[n] = sum_m x[m, n] (a thread per column, rows in order)
template <class T>
__global__ __launch_bounds__(RD_BLOCK) void sum_rows_kernel(const T* x, T* out, int N) {
    __shared__ float red[RD_BLOCK / 64];
    const T* xr = x + (size_t)blockIdx.x * N;
    float v = 0.f;
    for (int i = threadIdx.x; i < N; i += RD_BLOCK) v += to_f(xr[i]);
    v = rd_block(v, 0, red);
    if (threadIdx.x == 0) out[blockIdx.x] = from_f<T>(v);
}
template <class T>
__global__ __launch_bounds__(RD_BLOCK) void sum_cols_kernel(const T* x, T* out, int M, int N) {
    const int c = blockIdx.x * RD_BLOCK + threadIdx.x;
    if (c >= N) return;
    float v = 0.f;
    for (int r = 0; r < M; ++r) v += to_f(x[(size_t)r * N + c]);
    out[c] = from_f<T>(v);
}

template <class T>
__global__ void clamp_kernel(const T* x, T* y, size_t n, float lo, float hi) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = from_f<T>(fminf(fmaxf(to_f(x[i]), lo), hi));
}
template <class T>
__global__ void where_kernel(const uint8_t* cond, const T* a, const T* b, T* y, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = cond[i] ? a[i] : b[i];
}

__global__ void widen_i32_i64_kernel(const int32_t* in, int64_t* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

static inline int rd_grid(size_t n) {
    const size_t g = (n + RD_BLOCK - 1) / RD_BLOCK;
    return (int)(g < 1 ? 1 : (g > RD_MAX_BLOCKS ? RD_MAX_BLOCKS : g));
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_reduce(const void* x, void* out, size_t n, int op, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && out, "pgk_reduce: null pointer");
    PGK_REQUIRE(op >= 0 && op <= 3, "pgk_reduce: bad op %d (0 sum, 1 mean, 2 max, 3 min)", op);
    PGK_REQUIRE(n >= 1, "pgk_reduce: empty input");
    hipStream_t st = resolve_stream(s);
    const int grid = rd_grid(n);
    float* part = nullptr;
    if (pgk_status r = pgk_malloc((void**)&part, (size_t)grid * sizeof(float))) return r;
    PGK_DISPATCH_FLOAT(dt, "pgk_reduce", {
        reduce_partial_kernel<T><<<grid, RD_BLOCK, 0, st>>>((const T*)x, part, n, op);
        reduce_final_kernel<T><<<1, RD_BLOCK, 0, st>>>(part, grid, (T*)out, op, 1.0f / (float)n);
    });
    pgk_free(part);   // stream-ordered reuse
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_softmax_rows(const void* x, void* y, int rows, int n, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && y, "pgk_softmax_rows: null pointer");
    PGK_REQUIRE(rows >= 0 && n >= 1, "pgk_softmax_rows: bad shape rows=%d n=%d", rows, n);
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_softmax_rows", (softmax_rows_kernel<T><<<rows, RD_BLOCK, 0, st>>>((const T*)x, (T*)y, n)));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_sum_axis(const void* x, void* out, int m, int n, int axis, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && out, "pgk_sum_axis: null pointer");
    PGK_REQUIRE(m >= 1 && n >= 1 && (axis == 0 || axis == 1), "pgk_sum_axis: bad arguments m=%d n=%d axis=%d", m, n, axis);
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_sum_axis", {
        if (axis == 1) sum_rows_kernel<T><<<m, RD_BLOCK, 0, st>>>((const T*)x, (T*)out, n);
        else sum_cols_kernel<T><<<(n + RD_BLOCK - 1) / RD_BLOCK, RD_BLOCK, 0, st>>>((const T*)x, (T*)out, m, n);
    });
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_widen_i32_i64(const int32_t* src, int64_t* dst, size_t n, pgk_stream s) {
    PGK_REQUIRE(src && dst, "pgk_widen_i32_i64: null pointer");
    if (!n) return PGK_OK;
    widen_i32_i64_kernel<<<rd_grid(n), RD_BLOCK, 0, resolve_stream(s)>>>(src, dst, n);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_clamp(const void* x, void* y, size_t n, float lo, float hi, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(x && y, "pgk_clamp: null pointer");
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_clamp", (clamp_kernel<T><<<rd_grid(n), RD_BLOCK, 0, st>>>((const T*)x, (T*)y, n, lo, hi)));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_where(const uint8_t* cond, const void* a, const void* b, void* y, size_t n, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(cond && a && b && y, "pgk_where: null pointer");
    if (!n) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_where", (where_kernel<T><<<rd_grid(n), RD_BLOCK, 0, st>>>(cond, (const T*)a, (const T*)b, (T*)y, n)));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
