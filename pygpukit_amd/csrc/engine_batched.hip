// Batched decode projections on MFMA (3..64 sequences per launch): kernels in engine_batched.hip.h, run-time dispatch here.
// Its own translation unit so that the ~150 template instances compile beside engine.hip instead of inside it.

#include <cstdlib>
#include <type_traits>

#include "engine_common.hip.h"

namespace pgk {

typedef __bf16 bf16x8_b __attribute__((ext_vector_type(8)));
typedef float f32x4_b __attribute__((ext_vector_type(4)));
#include "engine_batched.hip.h"

// x16[m][:] = bf16(h[m][:] * rsqrt(mean(h[m]^2) + eps) * gamma)   (gamma == nullptr: plain fp32 -> bf16 rows)
// One 256-thread workgroup per row; rows up to 4096 columns stay in registers between the two passes.
__global__ __launch_bounds__(256) void norm_rows_bf16_kernel(unsigned long long* tl, const float* h, const bf16* gamma, bf16* out, int K, float eps) {
    const TLStamp tls(tl);
    __shared__ float red[16];
    const float* hr = h + (size_t)blockIdx.x * K;
    bf16* orow = out + (size_t)blockIdx.x * K;
    constexpr int MAXT = 4;
    float4 v[MAXT];
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int i = (threadIdx.x + t * 256) * 4;
        v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < K) {
            v[t] = *reinterpret_cast<const float4*>(hr + i);
            ss += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
        }
    }
    for (int i = (threadIdx.x + MAXT * 256) * 4; i < K; i += 1024) {   // columns past 4096: summed here, re-read below
        const float4 u = *reinterpret_cast<const float4*>(hr + i);
        ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    }
    float inv = 1.0f;
    if (gamma) {
        ss = block_sum(ss, red);
        inv = 1.0f / sqrtf(ss / K + eps);
    }
    auto emit = [&](int i, const float4& u) {
        float g0 = 1.f, g1 = 1.f, g2 = 1.f, g3 = 1.f;
        if (gamma) {
            const uint2 g = *reinterpret_cast<const uint2*>(gamma + i);
            g0 = __uint_as_float(g.x << 16); g1 = __uint_as_float(g.x & 0xFFFF0000u);
            g2 = __uint_as_float(g.y << 16); g3 = __uint_as_float(g.y & 0xFFFF0000u);
        }
        uint2 o;
        o.x = pack_bf16x2(u.x * inv * g0, u.y * inv * g1);
        o.y = pack_bf16x2(u.z * inv * g2, u.w * inv * g3);
        *reinterpret_cast<uint2*>(orow + i) = o;
    };
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int i = (threadIdx.x + t * 256) * 4;
        if (i < K) emit(i, v[t]);
    }
    for (int i = (threadIdx.x + MAXT * 256) * 4; i < K; i += 1024) emit(i, *reinterpret_cast<const float4*>(hr + i));
    tls.end();
}


pgk_status norm_rows_bf16(const float* h, const bf16* gamma, bf16* x16, int M, int K, float eps, hipStream_t st) {
    PGK_REQUIRE(h && x16 && M >= 1 && K >= 4 && K % 4 == 0, "norm_rows_bf16: bad arguments (M=%d, K=%d)", M, K);
    PGK_CHECK_HIP(launch_k(norm_rows_bf16_kernel, dim3(M), dim3(256), 0, st, h, gamma, x16, K, eps));
    return PGK_OK;
}

template <class WT>
static pgk_status batched_proj_t(int pro, int epi, const FusedArgs& a, int M, hipStream_t st, int nblk) {
    if (M > 16) {
        PGK_REQUIRE(pro == PRO_PLAIN, "batched_proj: more than 16 sequences take pre-normalised bf16 rows (PRO_PLAIN)");
        switch (epi) {
            case EPI_STORE: return launch_batched_tiled<WT, EPI_STORE>(a, M, st);
            case EPI_RESID: return launch_batched_tiled<WT, EPI_RESID>(a, M, st);
            case EPI_SWIGLU: return launch_batched_tiled<WT, EPI_SWIGLU>(a, M, st);
            case EPI_LOGITS: return launch_batched_tiled<WT, EPI_LOGITS>(a, M, st, nblk);
        }
        return set_error(PGK_ERR_INVALID, "batched_proj: epilogue %d", epi);
    }
    if (pro == PRO_NORM) {
        switch (epi) {
            case EPI_STORE: return launch_batched<WT, PRO_NORM, EPI_STORE>(a, M, st);
            case EPI_SWIGLU: return launch_batched<WT, PRO_NORM, EPI_SWIGLU>(a, M, st);
            case EPI_LOGITS: return launch_batched<WT, PRO_NORM, EPI_LOGITS>(a, M, st, nblk);
        }
    } else if (pro == PRO_PLAIN) {
        switch (epi) {
            case EPI_STORE: return launch_batched<WT, PRO_PLAIN, EPI_STORE>(a, M, st);
            case EPI_RESID: return launch_batched<WT, PRO_PLAIN, EPI_RESID>(a, M, st);
            case EPI_SWIGLU: return launch_batched<WT, PRO_PLAIN, EPI_SWIGLU>(a, M, st);
        }
    }
    return set_error(PGK_ERR_INVALID, "batched_proj: prologue %d with epilogue %d is not instantiated", pro, epi);
}

pgk_status batched_proj(bool fp8, int pro, int epi, const FusedArgs& a, int M, hipStream_t st, int nblk_logits) {
    if (fp8) return batched_proj_t<fp8e4m3>(pro, epi, a, M, st, nblk_logits);
    return batched_proj_t<bf16>(pro, epi, a, M, st, nblk_logits);
}

}  // namespace pgk
