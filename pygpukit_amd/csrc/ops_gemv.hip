// Op-level GEMV entry points (M = 1..8 activation rows against an [N,K] weight).
// See gemv_core.hip.h for the kernel design.

#include "gemv_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

constexpr int GEMV_BLOCK = 256;  // 4 waves
constexpr int GEMV_R = 4;        // rows per wave per trip

// x[M,K] (dtype T) -> LDS, then stream W.  out[m*N + n] = y (+ bias[n]).
template <class T, int M>
__global__ __launch_bounds__(GEMV_BLOCK) void gemv_kernel(const T* x, const T* w, const T* bias, T* out, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);
    constexpr int NV = Vec<T>::N;
    const int nvec = M * K / NV;  // K % NV == 0
    for (int i = threadIdx.x; i < nvec; i += GEMV_BLOCK)
        reinterpret_cast<uint4*>(xs)[i] = reinterpret_cast<const uint4*>(x)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (GEMV_BLOCK / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (GEMV_BLOCK / 64);
    for (int g = wave; g * GEMV_R < N; g += nwaves) {
        const int n0 = g * GEMV_R;
        const T* wrow[GEMV_R];
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r) wrow[r] = w + (size_t)min(n0 + r, N - 1) * K;
        float acc[GEMV_R][M];
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
        gemv_rows<T, T, M, GEMV_R>(wrow, xs, K, K, lane, acc);
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(acc[r][m]);
                if (lane == 0 && n0 + r < N) {
                    const float b = bias ? to_f(bias[n0 + r]) : 0.f;
                    out[(size_t)m * N + n0 + r] = from_f<T>(v + b);
                }
            }
    }
}

// Any K (no alignment requirement): one wave per output row, scalar loads.
template <class T>
__global__ __launch_bounds__(GEMV_BLOCK) void gemv_generic_kernel(const T* x, const T* w, const T* bias, T* out,
                                                                  int M, int K, int N) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (GEMV_BLOCK / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (GEMV_BLOCK / 64);
    for (long long o = wave; o < (long long)M * N; o += nwaves) {
        const int m = (int)(o / N), n = (int)(o % N);
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(to_f(w[(size_t)n * K + k]), to_f(x[(size_t)m * K + k]), acc);
        acc = wave_sum(acc);
        if (lane == 0) out[(size_t)m * N + n] = from_f<T>(acc + (bias ? to_f(bias[n]) : 0.f));
    }
}

// fp8-e4m3 weights [N,K] + bf16 block scales [N/128, K/128]; bf16 activations and output.
template <int M>
__global__ __launch_bounds__(GEMV_BLOCK) void gemv_fp8_kernel(const bf16* x, const fp8e4m3* w, const bf16* scale,
                                                              bf16* out, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);
    const int nvec = M * K / 8;
    for (int i = threadIdx.x; i < nvec; i += GEMV_BLOCK)
        reinterpret_cast<uint4*>(xs)[i] = reinterpret_cast<const uint4*>(x)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (GEMV_BLOCK / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (GEMV_BLOCK / 64);
    const int kb = K >> 7;
    for (int g = wave; g * GEMV_R < N; g += nwaves) {
        const int n0 = g * GEMV_R;
        const fp8e4m3* wrow[GEMV_R];
        const bf16* srow[GEMV_R];
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r) {
            const int n = min(n0 + r, N - 1);
            wrow[r] = w + (size_t)n * K;
            srow[r] = scale + (size_t)(n >> 7) * kb;
        }
        float acc[GEMV_R][M];
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
        gemv_rows_fp8<bf16, M, GEMV_R>(wrow, srow, xs, K, K, lane, acc);
#pragma unroll
        for (int r = 0; r < GEMV_R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = wave_sum(acc[r][m]);
                if (lane == 0 && n0 + r < N) out[(size_t)m * N + n0 + r] = from_f<bf16>(v);
            }
    }
}

static int gemv_grid(int N) {
    // >= 2 blocks per CU where the row count allows; 256 CUs
    const int groups = ceil_div(N, GEMV_R * (GEMV_BLOCK / 64));
    return groups < 1 ? 1 : (groups > 1024 ? 1024 : groups);
}

// Launch helper used by pgk_gemv and by the M<=8 path of pgk_gemm_nt.
template <class T>
pgk_status launch_gemv(const T* x, const T* w, const T* bias, T* out, int M, int K, int N, hipStream_t st) {
    constexpr int NV = Vec<T>::N;
    const size_t lds = (size_t)M * K * sizeof(T);
    const bool fast = (K % NV == 0) && aligned16(x) && aligned16(w) && lds <= 64 * 1024 && M <= 8;
    const int grid = gemv_grid(N);
    if (!fast) {
        gemv_generic_kernel<T><<<grid, GEMV_BLOCK, 0, st>>>(x, w, bias, out, M, K, N);
    } else {
        switch (M) {
#define PGK_GEMV_CASE(MM) case MM: gemv_kernel<T, MM><<<grid, GEMV_BLOCK, lds, st>>>(x, w, bias, out, K, N); break;
            PGK_GEMV_CASE(1) PGK_GEMV_CASE(2) PGK_GEMV_CASE(3) PGK_GEMV_CASE(4)
            PGK_GEMV_CASE(5) PGK_GEMV_CASE(6) PGK_GEMV_CASE(7) PGK_GEMV_CASE(8)
#undef PGK_GEMV_CASE
        }
    }
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

template pgk_status launch_gemv<float>(const float*, const float*, const float*, float*, int, int, int, hipStream_t);
template pgk_status launch_gemv<f16>(const f16*, const f16*, const f16*, f16*, int, int, int, hipStream_t);
template pgk_status launch_gemv<bf16>(const bf16*, const bf16*, const bf16*, bf16*, int, int, int, hipStream_t);

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_gemv(const void* a, const void* b_nk, void* c, int k, int n, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(a && b_nk && c, "pgk_gemv: null pointer");
    PGK_REQUIRE(k > 0 && n > 0, "pgk_gemv: bad shape K=%d N=%d", k, n);
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_gemv", return (launch_gemv<T>((const T*)a, (const T*)b_nk, nullptr, (T*)c, 1, k, n, st)));
    return PGK_OK;
}

pgk_status pgk_gemv_fp8_bf16(const void* a, const uint8_t* b_nk, const void* scale, void* c, int m, int k, int n,
                             pgk_stream s) {
    PGK_REQUIRE(a && b_nk && scale && c, "pgk_gemv_fp8_bf16: null pointer");
    PGK_REQUIRE(m >= 1 && k > 0 && n > 0, "pgk_gemv_fp8_bf16: bad shape M=%d K=%d N=%d", m, k, n);
    PGK_REQUIRE(k % 128 == 0 && n % 128 == 0, "pgk_gemv_fp8_bf16: K=%d and N=%d must be multiples of the 128x128 scale block", k, n);
    PGK_REQUIRE(aligned16(a) && aligned16(b_nk), "pgk_gemv_fp8_bf16: operands must be 16-byte aligned");
    hipStream_t st = resolve_stream(s);
    const int grid = gemv_grid(n);
    // M rows in passes of <= 8 (weights are re-read per pass; M > 8 belongs to pgk_w8a16_gemm_kn / MFMA)
    for (int m0 = 0; m0 < m; m0 += 8) {
        const int mm = (m - m0) < 8 ? (m - m0) : 8;
        const bf16* x = (const bf16*)a + (size_t)m0 * k;
        bf16* o = (bf16*)c + (size_t)m0 * n;
        const size_t lds = (size_t)mm * k * 2;
        PGK_REQUIRE(lds <= 64 * 1024, "pgk_gemv_fp8_bf16: K=%d too large for %d rows per pass", k, mm);
        switch (mm) {
#define PGK_CASE(MM) case MM: gemv_fp8_kernel<MM><<<grid, GEMV_BLOCK, lds, st>>>(x, (const fp8e4m3*)b_nk, (const bf16*)scale, o, k, n); break;
            PGK_CASE(1) PGK_CASE(2) PGK_CASE(3) PGK_CASE(4) PGK_CASE(5) PGK_CASE(6) PGK_CASE(7) PGK_CASE(8)
#undef PGK_CASE
        }
        PGK_LAUNCH_CHECK();
    }
    return PGK_OK;
}

}  // extern "C"
