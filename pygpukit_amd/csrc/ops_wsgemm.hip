// Weight-streaming skinny MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  with M <= 128.
//
// Prefill at S <= 128 (and batched decode) is WEIGHT-bound: 128 FLOP per weight byte vs the chip's 312, so
// the kernel is organised like the GEMV - every weight byte is read exactly once, straight from HBM into the
// MFMA B-operand registers - and not like a tiled GEMM:
//   * a wave owns one 16-row slab of W and walks K; lane l loads W[n0 + (l&15)][k + 8*(l>>4) .. +8], which IS the
//     B fragment of v_mfma_f32_16x16x32_bf16 (no LDS round trip for the read-once operand); 8 k-steps of loads
//     are in flight per wave before the first is consumed;
//   * the activations (the re-used operand) sit in LDS as [M_pad][256] K-tiles, XOR-swizzled so the 16 rows of an
//     A-fragment read hit 16 different 16-byte slots; one fragment read feeds MT = M_pad/16 MFMAs;
//   * fp32 accumulators stay in registers across the whole K range of the workgroup;
//   * small-N projections (N = hidden) are split along K over workgroups so the whole chip streams; partial
//     results go to fp32 slabs that the consumer (the next RMSNorm) sums - deterministic, no atomics.
// The reference has nothing of this shape (M < 16 goes to cuBLASLt, M >= 16 to 128x128 CUTLASS tiles,
// native/ops/matmul/matmul.cu:142-235).

#include "gemv_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

typedef __bf16 bf16x8_w __attribute__((ext_vector_type(8)));
typedef float f32x4_w __attribute__((ext_vector_type(4)));

constexpr int WS_KT = 256;        // K elements per LDS tile (512-byte rows)
constexpr int WS_THREADS = 256;   // 4 waves, 16 weight rows each

// byte offset of 16-byte chunk c (0..31) of row r in the [rows][256 x bf16] tile
__device__ __forceinline__ int ws_off(int r, int c) { return r * (WS_KT * 2) + (((c & ~15) | ((c ^ r) & 15)) << 4); }

struct WsArgs {
    const bf16* a;      // [M][lda]
    int lda;
    const void* w;      // [N][K] bf16, or fp8 codes with wscale
    const bf16* wscale; // fp8: [N/128][K/128]
    int M, N, K;
    int k_per_split;    // multiple of WS_KT
    void* c;            // mode 0: bf16 [M][N]; mode 1: fp32 slabs [ksplits][M][N]; mode 2: fp32 [M][N] += (ksplits == 1)
    const bf16* bias;   // mode 0 only
    int mode;
};

template <int MT, bool FP8, int MODE>
__global__ __launch_bounds__(WS_THREADS) void wsgemm_kernel(WsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char a_lds[];   // [MT*16][256] bf16, swizzled
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wid) * 16;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(kbeg + g.k_per_split, g.K);
    const int nrow = min(n0 + (lane & 15), g.N - 1);            // clamp: out-of-range rows are computed, never stored
    const int kl = 8 * (lane >> 4);

    f32x4_w acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_w{0.f, 0.f, 0.f, 0.f};

    // weight fragments of one K tile: 8 k-steps x 16 bytes per lane
    uint4 wreg[8];
    float wsc[8];
    auto load_w = [&](int kt) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = kt + s * 32 + kl;
            if constexpr (FP8) {
                // 8 codes per lane per k-step; dequantised to bf16 pairs when consumed
                const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(g.w) + (size_t)nrow * g.K + min(k, g.K - 8));
                wreg[s] = make_uint4(v.x, v.y, 0, 0);
                wsc[s] = to_f(g.wscale[(size_t)(nrow >> 7) * (g.K >> 7) + (min(k, g.K - 8) >> 7)]);
            } else {
                wreg[s] = load_nt16(reinterpret_cast<const bf16*>(g.w) + (size_t)nrow * g.K + min(k, g.K - 8));
            }
        }
    };
    // A tile staging: M_pad x 32 chunks of 16 bytes, MT*2 per thread
    constexpr int ACH = MT * 16 * 32 / WS_THREADS;
    uint4 areg[ACH];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int ch = threadIdx.x + i * WS_THREADS;
            const int r = ch >> 5, c = ch & 31;
            const int k = kt + c * 8;
            // unconditional load from a clamped address, masked afterwards: a guarded load would be waited
            // for on the spot (16 serialised round trips per tile)
            // (the mask is applied in store_a: touching the value here would make the compiler wait for the next
            // tile's loads before this tile's MFMAs)
            areg[i] = *reinterpret_cast<const uint4*>(g.a + (size_t)min(r, g.M - 1) * g.lda + min(k, g.K - 8));
        }
    };
    auto store_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int ch = threadIdx.x + i * WS_THREADS;
            const int r = ch >> 5, c = ch & 31;
            const bool ok = r < g.M && kt + c * 8 < kend;
            const uint4 v = areg[i];
            *reinterpret_cast<uint4*>(a_lds + ws_off(r, c)) = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
    };

    load_w(kbeg);                        // HBM first: the longest latency in the kernel
    __builtin_amdgcn_sched_barrier(0);
    load_a(kbeg);
    for (int kt = kbeg; kt < kend; kt += WS_KT) {
        __syncthreads();          // previous tile fully consumed
        store_a(kt);
        __syncthreads();
        uint4 wcur[8];
        float scur[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { wcur[s] = wreg[s]; scur[s] = wsc[s]; }
        if (kt + WS_KT < kend) { load_w(kt + WS_KT); load_a(kt + WS_KT); }   // next tile in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        // A fragments are read one k-step ahead of the MFMAs that use them: with one wave per SIMD nothing else
        // hides the LDS latency.  No K-tail branch: the A tile is zero beyond kend and the clamped W loads are
        // finite, so tail k-steps add exact zeros.
        uint4 af[2][MT];
        auto read_a = [&](int s, uint4 (&dst)[MT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                dst[mt] = *reinterpret_cast<const uint4*>(a_lds + ws_off(mt * 16 + (lane & 15), s * 4 + (lane >> 4)));
        };
        read_a(0, af[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) read_a(s + 1, af[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the MT reads ahead of this k-step's MFMAs
            uint4 bfrag;
            if constexpr (FP8) {
                float f[8];
                const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wcur[s].x, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wcur[s].x, true);
                const f32x2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wcur[s].y, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wcur[s].y, true);
                f[0] = a0.x; f[1] = a0.y; f[2] = a1.x; f[3] = a1.y; f[4] = a2.x; f[5] = a2.y; f[6] = a3.x; f[7] = a3.y;
                bfrag = make_uint4(pack_bf16x2(f[0] * scur[s], f[1] * scur[s]), pack_bf16x2(f[2] * scur[s], f[3] * scur[s]),
                                   pack_bf16x2(f[4] * scur[s], f[5] * scur[s]), pack_bf16x2(f[6] * scur[s], f[7] * scur[s]));
            } else {
                bfrag = wcur[s];
            }
            const bf16x8_w b = __builtin_bit_cast(bf16x8_w, bfrag);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_w, af[s & 1][mt]), b, acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C/D map: col (n) = lane & 15, row (m) = (lane >> 4) * 4 + reg.  Straight-line epilogue: the bias /
    // read-modify-write loads are issued together, never one per element under a branch.
    const int n = min(n0 + (lane & 15), g.N - 1);
    const bool n_ok = n0 + (lane & 15) < g.N;
    if constexpr (MODE == 0) {
        const float b = g.bias ? to_f(g.bias[n]) : 0.f;
        bf16* c = reinterpret_cast<bf16*>(g.c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r;
                if (n_ok && m < g.M) c[(size_t)m * g.N + n] = from_f<bf16>(acc[mt][r] + b);
            }
    } else if constexpr (MODE == 1) {
        float* c = reinterpret_cast<float*>(g.c) + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r;
                if (n_ok && m < g.M) c[(size_t)m * g.N + n] = acc[mt][r];
            }
    } else {
        float* c = reinterpret_cast<float*>(g.c);
        float old[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[mt][r] = c[(size_t)min(mt * 16 + (lane >> 4) * 4 + r, g.M - 1) * g.N + n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r;
                if (n_ok && m < g.M) c[(size_t)m * g.N + n] = old[mt][r] + acc[mt][r];
            }
    }
}

// ksplits chosen so that ~256+ workgroups stream; returns the number of splits used (slabs to sum).
int wsgemm_pick_splits(int N, int K, bool allow_split) {
    const int nblk = ceil_div(N, 64);
    if (!allow_split || nblk >= 192) return 1;
    int s = ceil_div(256, nblk);
    const int max_s = K / WS_KT;
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    // normalise to a split count that tiles K in whole LDS tiles (what wsgemm_nt will actually launch)
    const int kps = ceil_div(ceil_div(K, s), WS_KT) * WS_KT;
    return ceil_div(K, kps);
}

// A[M,K] bf16 (row stride lda), W[N,K]; mode 0: bf16 C (+bias); mode 1: fp32 slabs [splits][M][N]; mode 2: fp32 C += .
pgk_status wsgemm_nt(const bf16* a, int lda, const void* w, const bf16* wscale, bool fp8, void* c, const bf16* bias, int mode,
                     int splits, int M, int N, int K, hipStream_t st) {
    PGK_REQUIRE(M >= 1 && M <= 128, "wsgemm: M=%d outside [1,128]", M);
    PGK_REQUIRE(K % 8 == 0 && lda % 8 == 0, "wsgemm: K=%d / lda=%d must be multiples of 8", K, lda);
    PGK_REQUIRE(mode == 1 || splits == 1, "wsgemm: split-K needs slab output");
    WsArgs g{a, lda, w, wscale, M, N, K, 0, c, bias, mode};
    int kps = ceil_div(K, splits);
    kps = ceil_div(kps, WS_KT) * WS_KT;
    g.k_per_split = kps;
    const int real_splits = ceil_div(K, kps);
    PGK_REQUIRE(real_splits == splits, "wsgemm: %d splits do not tile K=%d (use %d)", splits, K, real_splits);
    const int mt = ceil_div(M, 16);
    dim3 grid(ceil_div(N, 64), splits);
#define PGK_WS_LAUNCH(MTV, F8, MD)                                                                               \
    {                                                                                                            \
        static bool done = false;                                                                                \
        if (lds > 48 * 1024 && !done) {                                                                          \
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wsgemm_kernel<MTV, F8, MD>),         \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            done = true;                                                                                         \
        }                                                                                                        \
        wsgemm_kernel<MTV, F8, MD><<<grid, WS_THREADS, lds, st>>>(g);                                            \
    }
#define PGK_WS(MTV)                                                                                              \
    if (mt <= MTV) {                                                                                             \
        const size_t lds = (size_t)MTV * 16 * WS_KT * 2;                                                         \
        if (fp8) {                                                                                               \
            if (mode == 0) PGK_WS_LAUNCH(MTV, true, 0) else if (mode == 1) PGK_WS_LAUNCH(MTV, true, 1) else PGK_WS_LAUNCH(MTV, true, 2) \
        } else {                                                                                                 \
            if (mode == 0) PGK_WS_LAUNCH(MTV, false, 0) else if (mode == 1) PGK_WS_LAUNCH(MTV, false, 1) else PGK_WS_LAUNCH(MTV, false, 2) \
        }                                                                                                        \
        PGK_CHECK_HIP(hipGetLastError());                                                                        \
        return PGK_OK;                                                                                           \
    }
    PGK_WS(1) PGK_WS(2) PGK_WS(4) PGK_WS(8)
#undef PGK_WS
#undef PGK_WS_LAUNCH
    return set_error(PGK_ERR_INVALID, "wsgemm: no tile for M=%d", M);
}

}  // namespace pgk
