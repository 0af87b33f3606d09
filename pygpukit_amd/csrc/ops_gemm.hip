// MFMA GEMMs for gfx950 (prefill / batched projections):
//   pgk_gemm_nt : C[M,N] = A[M,K] . W[N,K]^T (+bias)   - the Linear layer on the PyTorch-layout weight
//   pgk_gemm_nn : C[M,N] = A[M,K] . B[K,N]             - the generic ops.matmul
//   pgk_w8a16_gemm_kn : bf16 A, fp8-e4m3 B[K,N] with 128x128 block scales (reference K4 layout)
//   (internal) nt with fp8 W[N,K] + block scales, used by the engine's fp8 prefill
//
// Structure (v1): BM x BN x 64 block tile, 256 threads = 2x2 waves, v_mfma_f32_16x16x32_{bf16,f16},
// fp32 accumulate; A/B tiles staged global -> VGPR -> LDS (16-byte chunks, XOR-swizzled so the
// ds_read_b128 fragment reads are bank-conflict free), two LDS buffers: the global loads of tile t+1
// are issued before the MFMAs of tile t and written to LDS after them (one barrier per K tile).
// The reference reaches for CUTLASS / cuBLASLt here (native/ops/matmul/matmul.cu:43-354); neither
// exists on this target and nothing is linked in their place.

#include "gemm_epilogues.hip.h"
#include "gemv_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

template <class T> pgk_status launch_gemv(const T*, const T*, const T*, T*, int, int, int, hipStream_t);
pgk_status wsgemm_nt(const bf16* a, int lda, const void* w, const bf16* wscale, bool fp8, void* c, const bf16* bias, int mode,
                     int splits, int M, int N, int K, hipStream_t st);
// (packed: W is the fragment-major bf16 copy of ops_pkgemm.hip instead of the row-major weight - the staged kernels' DMA
// instruction moves 16 rows x 32 k, which is one block of that layout)
pgk_status gemm256_bf16_nt(const bf16* A, const bf16* W, const bf16* bias, void* C, bool accum_f32, int M, int N, int K,
                           hipStream_t st, bool packed = false);
pgk_status gemm256_bf16_swiglu_nt(const bf16* A, const bf16* W, bf16* act, int M, int I, int K, hipStream_t st, bool packed = false);
bool gemm128s_ok(int M, int N, int K);    // ops_gemm256.hip: 128 x 128 tiles on the staged LDS-DMA pipeline (bf16, M > 128, K % 64 == 0, N % 8 == 0)
pgk_status gemm128s_bf16_nt(const bf16* A, const bf16* W, const bf16* bias, void* C, int mode, int splits, int M, int N, int K, hipStream_t st,
                            const QkvHeadArgs* heads = nullptr, bool packed = false);
// the 256 x 256 structure needs enough tiles to fill the chip and whole 128-byte K rows
// PGK_GEMM256 = 0 / 1 forces the choice (read per call: tests flip it to drive small shapes through both kernels); shared
// with the fp8 x fp8 GEMM (ops_fp8_gemm.hip)
bool want_gemm256(int M, int N) {
    const char* e = getenv("PGK_GEMM256");
    const int force = e ? atoi(e) : -1;
    if (force == 0) return false;
    if (force == 1) return true;
    return (long long)ceil_div(M, 256) * ceil_div(N, 256) >= 192;
}
static bool use_gemm256(int M, int N, int K) { return K % 64 == 0 && want_gemm256(M, N); }

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <class T> __device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mfma16<bf16>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<f16>(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

constexpr int GEMM_BK = 64;       // K elements per tile (128 bytes per LDS row)
constexpr int GEMM_THREADS = 256;

// byte offset of 16-byte chunk `kc` (0..7) of row `row` in a [rows][64 x 16-bit] swizzled LDS tile
__device__ __forceinline__ int lds_off(int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); }

enum BMode { B_NT = 0, B_NN = 1, B_NT_FP8 = 2, B_KN_FP8 = 3 };

// ---- A-side (and NT B-side) tile: rows x 64 elements, k contiguous in memory -----------------
template <int ROWS>
struct TileRegsK {
    static constexpr int CH = ROWS * 8 / GEMM_THREADS;  // 16-byte chunks per thread
    uint4 v[CH > 0 ? CH : 1];
    template <class T>
    __device__ __forceinline__ void load(const T* base, int row0, int nrows, int k0, int K, int ld) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            const int r = c >> 3, kc = c & 7;
            const int gr = row0 + r, gk = k0 + kc * 8;
            if (gr < nrows && gk < K) v[i] = *reinterpret_cast<const uint4*>(base + (size_t)gr * ld + gk);
            else v[i] = make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            *reinterpret_cast<uint4*>(lds + lds_off(c >> 3, c & 7)) = v[i];
        }
    }
};

// ---- NN B-side tile: B[K,N] row-major; tile is 64 k-rows x BN columns, transposed into LDS ----
template <int BN>
struct TileRegsN {
    static constexpr int CH = 64 * (BN / 8) / GEMM_THREADS;
    uint4 v[CH];
    template <class T>
    __device__ __forceinline__ void load(const T* base, int n0, int N, int k0, int K) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            const int k = c / (BN / 8), nc = c % (BN / 8);
            const int gk = k0 + k, gn = n0 + nc * 8;
            if (gk < K && gn < N) v[i] = *reinterpret_cast<const uint4*>(base + (size_t)gk * N + gn);  // N % 8 == 0
            else v[i] = make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            const int k = c / (BN / 8), nc = c % (BN / 8);
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = nc * 8 + j;
                const uint16_t e = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu));
                *reinterpret_cast<uint16_t*>(lds + lds_off(n, k >> 3) + (k & 7) * 2) = e;
            }
        }
    }
};

// ---- fp8 B tiles, dequantised to bf16 on the way into LDS -----------------------------------
// NT: W[N,K] u8, scale[N/128, K/128] bf16.  One 16-byte load = 16 k-values of one row.
template <int BN>
struct TileRegsFp8NT {
    static constexpr int CH = (BN * 4 + GEMM_THREADS - 1) / GEMM_THREADS;  // 4 x 16-code chunks per row
    uint4 v[CH];
    float sc[CH];
    __device__ __forceinline__ void load(const uint8_t* w, const bf16* scale, int n0, int N, int k0, int K) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            const int r = c >> 2, q = c & 3;
            const int gn = n0 + r, gk = k0 + q * 16;
            if (c < BN * 4 && gn < N && gk < K) {
                v[i] = *reinterpret_cast<const uint4*>(w + (size_t)gn * K + gk);
                sc[i] = to_f(scale[(size_t)(gn >> 7) * (K >> 7) + (gk >> 7)]);
            } else {
                v[i] = make_uint4(0, 0, 0, 0);
                sc[i] = 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            if (c >= BN * 4) continue;
            const int r = c >> 2, q = c & 3;
            float f[16];
            WTraits<fp8e4m3>::decode(v[i], f);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 o = make_uint4(pack_bf16x2(f[8 * h] * sc[i], f[8 * h + 1] * sc[i]),
                                           pack_bf16x2(f[8 * h + 2] * sc[i], f[8 * h + 3] * sc[i]),
                                           pack_bf16x2(f[8 * h + 4] * sc[i], f[8 * h + 5] * sc[i]),
                                           pack_bf16x2(f[8 * h + 6] * sc[i], f[8 * h + 7] * sc[i]));
                *reinterpret_cast<uint4*>(lds + lds_off(r, q * 2 + h)) = o;
            }
        }
    }
};
// KN: B[K,N] u8, scale[K/128, N/128].  One 16-byte load = 16 n-values of one k.
template <int BN>
struct TileRegsFp8KN {
    static constexpr int PER_ROW = BN / 16;
    static constexpr int CH = (64 * PER_ROW + GEMM_THREADS - 1) / GEMM_THREADS;
    uint4 v[CH];
    float sc[CH];
    __device__ __forceinline__ void load(const uint8_t* b, const bf16* scale, int n0, int N, int k0, int K) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            const int k = c / PER_ROW, nc = c % PER_ROW;
            const int gk = k0 + k, gn = n0 + nc * 16;
            if (c < 64 * PER_ROW && gk < K && gn < N) {
                v[i] = *reinterpret_cast<const uint4*>(b + (size_t)gk * N + gn);  // N % 16 == 0
                sc[i] = to_f(scale[(size_t)(gk >> 7) * (N >> 7) + (gn >> 7)]);
            } else {
                v[i] = make_uint4(0, 0, 0, 0);
                sc[i] = 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = threadIdx.x + i * GEMM_THREADS;
            if (c >= 64 * PER_ROW) continue;
            const int k = c / PER_ROW, nc = c % PER_ROW;
            float f[16];
            WTraits<fp8e4m3>::decode(v[i], f);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = nc * 16 + j;
                *reinterpret_cast<uint16_t*>(lds + lds_off(n, k >> 3) + (k & 7) * 2) = f_to_bf16_bits(f[j] * sc[i]);
            }
        }
    }
};

template <class T, int BM, int BN, int MODE, int EPI>
// kps > 0 (B_NT only, EPI 2): split-K - workgroup z multiplies k in [z * kps, (z + 1) * kps) and stores its fp32 partial tile into
// slab z of Cv ([splits][M][N]); the consumer sums the slabs (the engine's RMSNorm kernel does, as for the skinny GEMMs).
__global__ __launch_bounds__(GEMM_THREADS) void gemm_mfma_kernel(const T* A, const void* Bv, const bf16* bscale,
                                                                const T* bias, void* Cv, int M, int N, int K, int kps) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto As = [&](int buf) -> char* { return smem + buf * A_BYTES; };
    auto Bs = [&](int buf) -> char* { return smem + 2 * A_BYTES + buf * B_BYTES; };

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    TileRegsK<BM> ra;
    TileRegsK<BN> rb_nt;
    TileRegsN<BN> rb_nn;
    TileRegsFp8NT<BN> rb_f8nt;
    TileRegsFp8KN<BN> rb_f8kn;

    const int kbeg = kps > 0 ? (int)blockIdx.z * kps : 0, kend = kps > 0 ? min(K, kbeg + kps) : K;
    auto load_tiles = [&](int k0) {
        ra.load(A, m0, M, k0, kend, K);
        if constexpr (MODE == B_NT) rb_nt.load((const T*)Bv, n0, N, k0, kend, K);
        else if constexpr (MODE == B_NN) rb_nn.load((const T*)Bv, n0, N, k0, K);
        else if constexpr (MODE == B_NT_FP8) rb_f8nt.load((const uint8_t*)Bv, bscale, n0, N, k0, K);
        else rb_f8kn.load((const uint8_t*)Bv, bscale, n0, N, k0, K);
    };
    auto store_tiles = [&](int buf) {
        ra.store(As(buf));
        if constexpr (MODE == B_NT) rb_nt.store(Bs(buf));
        else if constexpr (MODE == B_NN) rb_nn.store(Bs(buf));
        else if constexpr (MODE == B_NT_FP8) rb_f8nt.store(Bs(buf));
        else rb_f8kn.store(Bs(buf));
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * GEMM_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fa[TM], fb[TN];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const uint4*>(As(buf) + lds_off(wm * WM + i * 16 + (lane & 15), kc));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const uint4*>(Bs(buf) + lds_off(wn * WN + j * 16 + (lane & 15), kc));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<T>(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // C/D map of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 16 + (lane & 15);
            if (col >= N) continue;
            const float b = bias ? to_f(bias[col]) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
                if (row >= M) continue;
                if constexpr (EPI == 0) reinterpret_cast<T*>(Cv)[(size_t)row * N + col] = from_f<T>(acc[i][j][r] + b);
                else if constexpr (EPI == 2) reinterpret_cast<float*>(Cv)[((size_t)blockIdx.z * M + row) * N + col] = acc[i][j][r];   // split-K slab
                else reinterpret_cast<float*>(Cv)[(size_t)row * N + col] += acc[i][j][r] + b;  // fp32 residual stream
            }
        }
}

// fp32 (and odd-shape) fallback: 64x64 tile, 16x16 threads x 4x4 outputs, LDS-staged.
template <class T, bool B_IS_NT>
__global__ __launch_bounds__(256) void gemm_simple_kernel(const T* A, const T* B, const T* bias, T* C, int M, int N, int K) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, k = i & 15;  // r: row in tile, k: k index
            const int gm = m0 + r, gn = n0 + r, gk = k0 + k;
            As[k][r] = (gm < M && gk < K) ? to_f(A[(size_t)gm * K + gk]) : 0.f;
            if (B_IS_NT) Bs[k][r] = (gn < N && gk < K) ? to_f(B[(size_t)gn * K + gk]) : 0.f;
        }
        if (!B_IS_NT) {
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int k = i >> 6, c = i & 63;
                const int gk = k0 + k, gn = n0 + c;
                Bs[k][c] = (gk < K && gn < N) ? to_f(B[(size_t)gk * N + gn]) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
            if (gm < M && gn < N) C[(size_t)gm * N + gn] = from_f<T>(acc[i][j] + (bias ? to_f(bias[gn]) : 0.f));
        }
}

template <class T, int BM, int BN, int MODE, int EPI>
static pgk_status launch_mfma(const T* A, const void* B, const bf16* bscale, const T* bias, void* C, int M, int N, int K,
                              hipStream_t st, int splits = 1) {
    constexpr size_t LDS = 2 * (size_t)(BM + BN) * 128;
    static bool attr_done = false;
    if (LDS > 48 * 1024 && !attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<T, BM, BN, MODE, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    dim3 grid(ceil_div(N, BN), ceil_div(M, BM), splits);
    const int kps = splits > 1 ? ceil_div(ceil_div(K, splits), GEMM_BK) * GEMM_BK : 0;
    gemm_mfma_kernel<T, BM, BN, MODE, EPI><<<grid, GEMM_THREADS, LDS, st>>>(A, B, bscale, bias, C, M, N, K, kps);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// Tile choice: smallest BM covering M (<=128); BN as large as keeps >= ~256 workgroups in flight.
template <class T, int MODE, int EPI = 0>
static pgk_status dispatch_mfma(const T* A, const void* B, const bf16* bscale, const T* bias, void* C, int M, int N, int K,
                                hipStream_t st) {
    int bm = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
    long long mblocks = (M + bm - 1) / bm;
    int bn = 128;
    while (bn > 32 && mblocks * ((N + bn - 1) / bn) < 256) bn >>= 1;
    // 128 x 64 tiles that only just cover the chip (one 4-wave workgroup per CU, nothing to overlap its barriers with)
    // lose to twice as many 64 x 64 tiles: M=2048, N=1024, K=2048/3072 measured 23.9 / 33.9 us against 28.3 / 39.9
    if (bm == 128 && bn == 64 && mblocks * ((N + 63) / 64) < 512) { bm = 64; mblocks = (M + 63) / 64; }
    if (MODE == B_KN_FP8 && bn < 64) bn = 64;  // keep whole 16-code chunks per thread
#define PGK_TILE(BM_, BN_) if (bm == BM_ && bn == BN_) return launch_mfma<T, BM_, BN_, MODE, EPI>(A, B, bscale, bias, C, M, N, K, st);
    PGK_TILE(128, 128) PGK_TILE(128, 64) PGK_TILE(128, 32)
    PGK_TILE(64, 128) PGK_TILE(64, 64) PGK_TILE(64, 32)
    PGK_TILE(32, 128) PGK_TILE(32, 64) PGK_TILE(32, 32)
#undef PGK_TILE
    return set_error(PGK_ERR_INVALID, "gemm: no tile for bm=%d bn=%d", bm, bn);
}

// fp8 codes [N,K] + 128x128 bf16 block scales -> bf16 [N,K] (exact: an e4m3 value times a bf16 scale rounds once)
__global__ __launch_bounds__(256) void dequant_fp8_blocks_kernel(const uint8_t* w8, const bf16* sw, bf16* out, int N, int K) {
    const size_t chunks = (size_t)N * K / 16, stride = (size_t)gridDim.x * blockDim.x;
    const int KB = K >> 7, cpr = K / 16;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < chunks; t += stride) {
        const int n = (int)(t / cpr), c = (int)(t % cpr);
        const uint4 raw = *reinterpret_cast<const uint4*>(w8 + (size_t)n * K + (size_t)c * 16);
        const float sc = to_f(sw[(size_t)(n >> 7) * KB + (c >> 3)]);
        float f[16];
        WTraits<fp8e4m3>::decode(raw, f);
        uint4 lo, hi;
        lo.x = pack_bf16x2(f[0] * sc, f[1] * sc); lo.y = pack_bf16x2(f[2] * sc, f[3] * sc);
        lo.z = pack_bf16x2(f[4] * sc, f[5] * sc); lo.w = pack_bf16x2(f[6] * sc, f[7] * sc);
        hi.x = pack_bf16x2(f[8] * sc, f[9] * sc); hi.y = pack_bf16x2(f[10] * sc, f[11] * sc);
        hi.z = pack_bf16x2(f[12] * sc, f[13] * sc); hi.w = pack_bf16x2(f[14] * sc, f[15] * sc);
        uint4* o = reinterpret_cast<uint4*>(out + (size_t)n * K + (size_t)c * 16);
        o[0] = lo;
        o[1] = hi;
    }
}

// internal (engine prefill): N = hidden projections of a long prompt whose 128 x 128 tiles do not cover the chip (M = 2048,
// N = 1024: 128 tiles; the 64 x 64 tiles that do cover it run at 200-290 TFLOP/s) - the 128-tile kernel split along K into
// `splits` fp32 slabs [splits][M][N] that the next RMSNorm sums.  Returns the number of slabs through the argument; 1 = not used.
int engine_gemm_pick_splits(int M, int N, int K) {
    const long long tiles = (long long)ceil_div(M, 128) * ceil_div(N, 128);
    if (M <= 128 || tiles >= 192 || use_gemm256(M, N, K)) return 1;
    int s = (int)(256 / tiles);      // (512 workgroups - two per CU - measured the same to 3 %: S = 2048 4.86 vs 4.72 ms, S = 512 2.50 vs 2.52)
    if (s > 4) s = 4;
    while (s > 1 && (K / s < 512 || (gemm128s_ok(M, N, K) && K % (64 * s) != 0))) --s;
    return s < 1 ? 1 : s;
}
pgk_status engine_gemm_nt_slabs(const bf16* A, const void* W, float* slabs, int splits, int M, int N, int K, hipStream_t st, bool packed) {
    PGK_REQUIRE(splits >= 2 && K % 8 == 0, "engine_gemm_nt_slabs: splits=%d K=%d", splits, K);
    if (gemm128s_ok(M, N, K) && K % (64 * splits) == 0) return gemm128s_bf16_nt(A, (const bf16*)W, nullptr, slabs, 2, splits, M, N, K, st, nullptr, packed);
    PGK_REQUIRE(!packed, "engine_gemm_nt_slabs: M=%d N=%d K=%d splits=%d has no kernel that reads the fragment-major copy", M, N, K, splits);
    return launch_mfma<bf16, 128, 128, B_NT, 2>(A, W, nullptr, nullptr, slabs, M, N, K, st, splits);
}

// internal (engine prefill): bf16 A against a bf16 or fp8 (+128x128 bf16 block scales) weight W[N,K];
// either a bf16 result or an fp32 "+=" into the residual stream.
// the shapes whose kernels can read the fragment-major bf16 copy (engine_gemm_nt and friends with packed = true)
bool engine_gemm_packed_ok(int M, int N, int K) { return N % 16 == 0 && K % 64 == 0 && (use_gemm256(M, N, K) || gemm128s_ok(M, N, K)); }

pgk_status engine_gemm_nt(const bf16* A, const void* W, const bf16* wscale, bool fp8, void* C, bool accum_f32, int M,
                          int N, int K, hipStream_t st, bool packed) {
    if (packed) {
        PGK_REQUIRE(!fp8 && engine_gemm_packed_ok(M, N, K), "engine_gemm_nt: M=%d N=%d K=%d has no kernel that reads the fragment-major copy", M, N, K);
        if (use_gemm256(M, N, K)) return gemm256_bf16_nt(A, (const bf16*)W, nullptr, C, accum_f32, M, N, K, st, true);
        return gemm128s_bf16_nt(A, (const bf16*)W, nullptr, C, accum_f32 ? 1 : 0, 1, M, N, K, st, nullptr, true);
    }
    if (fp8 && use_gemm256(M, N, K) && K % 128 == 0 && N % 128 == 0) {
        // large w8a16 products: dequantise the weight once into a bf16 scratch (a ~10 % extra pass over memory) and run
        // the LDS-DMA bf16 kernel, instead of dequantising in the staging path of the 128-tile kernel (0.58 vs ~1 PFLOP/s)
        void* wb = nullptr;
        if (pgk_status r = pgk_malloc(&wb, (size_t)N * K * sizeof(bf16))) return r;
        const size_t chunks = (size_t)N * K / 16;
        dequant_fp8_blocks_kernel<<<(unsigned)(ceil_div((long long)chunks, 256) > 8192 ? 8192 : ceil_div((long long)chunks, 256)), 256, 0, st>>>(
            (const uint8_t*)W, wscale, (bf16*)wb, N, K);
        const pgk_status r = gemm256_bf16_nt(A, (const bf16*)wb, nullptr, C, accum_f32, M, N, K, st);
        pgk_free(wb);   // stream-ordered reuse
        return r;
    }
    if (fp8) {
        if (accum_f32) return dispatch_mfma<bf16, B_NT_FP8, 1>(A, W, wscale, nullptr, C, M, N, K, st);
        return dispatch_mfma<bf16, B_NT_FP8, 0>(A, W, wscale, nullptr, C, M, N, K, st);
    }
    if (use_gemm256(M, N, K)) return gemm256_bf16_nt(A, (const bf16*)W, nullptr, C, accum_f32, M, N, K, st);
    if (gemm128s_ok(M, N, K)) return gemm128s_bf16_nt(A, (const bf16*)W, nullptr, C, accum_f32 ? 1 : 0, 1, M, N, K, st);
    if (accum_f32) return dispatch_mfma<bf16, B_NT, 1>(A, W, nullptr, nullptr, C, M, N, K, st);
    return dispatch_mfma<bf16, B_NT, 0>(A, W, nullptr, nullptr, C, M, N, K, st);
}

// internal (engine prefill, bf16 weights, head_dim 128): the QKV projection with per-head RMSNorm + RoPE + KV-cache write as its
// epilogue (QkvHeadArgs) - on the 128-tile kernel, whose tile columns are whole heads; where engine_gemm_nt would pick the
// 256-tile kernel the separate pass stays
bool engine_gemm_qkv_heads_ok(int M, int N, int K) { return N % 128 == 0 && gemm128s_ok(M, N, K) && !use_gemm256(M, N, K); }
pgk_status engine_gemm_qkv_heads_nt(const bf16* A, const bf16* W, bf16* qkv, int M, int N, int K, const QkvHeadArgs& hd, hipStream_t st, bool packed) {
    PGK_REQUIRE(engine_gemm_qkv_heads_ok(M, N, K), "engine_gemm_qkv_heads: M=%d N=%d K=%d outside the fused kernel's shapes", M, N, K);
    return gemm128s_bf16_nt(A, W, nullptr, qkv, 3, 1, M, N, K, st, &hd, packed);
}

// internal (engine prefill): act[M][I] = bf16(silu(A . Wg^T) * (A . Wu^T)) on the fused [2 I, K] gate / up weight (bf16, or fp8 with
// block scales: dequantised once into a bf16 scratch as in engine_gemm_nt), SwiGLU in the 256-tile kernel's epilogue - the
// [M][2 I] intermediate never goes to HBM.  Only where engine_gemm_nt would pick the 256-tile kernel anyway.
// (below 192 tiles of 256 x 256, bf16 weights: the same epilogue on the 128-tile kernel, 64 gate + 64 up rows per tile)
bool engine_gemm_swiglu_ok(int M, int I, int K, bool fp8) {
    if (I % 128 == 0 && use_gemm256(M, 2 * I, K) && (!fp8 || K % 128 == 0)) return true;
    return !fp8 && I % 64 == 0 && gemm128s_ok(M, I, K);
}
pgk_status engine_gemm_swiglu_nt(const bf16* A, const void* W, const bf16* wscale, bool fp8, bf16* act, int M, int I, int K, hipStream_t st, bool packed) {
    PGK_REQUIRE(engine_gemm_swiglu_ok(M, I, K, fp8) && !(packed && fp8), "engine_gemm_swiglu: M=%d I=%d K=%d outside the fused kernel's shapes", M, I, K);
    if (!(I % 128 == 0 && use_gemm256(M, 2 * I, K))) return gemm128s_bf16_nt(A, (const bf16*)W, nullptr, act, 4, 1, M, I, K, st, nullptr, packed);
    if (!fp8) return gemm256_bf16_swiglu_nt(A, (const bf16*)W, act, M, I, K, st, packed);
    void* wb = nullptr;
    if (pgk_status r = pgk_malloc(&wb, (size_t)2 * I * K * sizeof(bf16))) return r;
    const size_t chunks = (size_t)2 * I * K / 16;
    dequant_fp8_blocks_kernel<<<(unsigned)(ceil_div((long long)chunks, 256) > 8192 ? 8192 : ceil_div((long long)chunks, 256)), 256, 0, st>>>(
        (const uint8_t*)W, wscale, (bf16*)wb, 2 * I, K);
    const pgk_status r = gemm256_bf16_swiglu_nt(A, (const bf16*)wb, act, M, I, K, st);
    pgk_free(wb);   // stream-ordered reuse
    return r;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_gemm_nt(const void* a, const void* w, const void* bias, void* c, int m, int n, int k, pgk_dtype dt,
                       pgk_stream s) {
    PGK_REQUIRE(a && w && c, "pgk_gemm_nt: null pointer");
    PGK_REQUIRE(m >= 0 && n > 0 && k > 0, "pgk_gemm_nt: bad shape M=%d N=%d K=%d", m, n, k);
    if (!m) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const bool al = aligned16(a) && aligned16(w) && (k % 8 == 0);
    if (dt == PGK_F32 || !al) {
        if (m <= 8) {
            PGK_DISPATCH_FLOAT(dt, "pgk_gemm_nt", return (launch_gemv<T>((const T*)a, (const T*)w, (const T*)bias, (T*)c, m, k, n, st)));
        }
        dim3 grid(ceil_div(n, 64), ceil_div(m, 64));
        PGK_DISPATCH_FLOAT(dt, "pgk_gemm_nt", (gemm_simple_kernel<T, true><<<grid, 256, 0, st>>>((const T*)a, (const T*)w, (const T*)bias, (T*)c, m, n, k)));
        PGK_LAUNCH_CHECK();
        return PGK_OK;
    }
    if (m <= 8 && (size_t)m * k * 2 <= 64 * 1024) {  // weight-streaming GEMV path
        if (dt == PGK_BF16) return launch_gemv<bf16>((const bf16*)a, (const bf16*)w, (const bf16*)bias, (bf16*)c, m, k, n, st);
        return launch_gemv<f16>((const f16*)a, (const f16*)w, (const f16*)bias, (f16*)c, m, k, n, st);
    }
    if (dt == PGK_BF16 && m <= 128)   // weight-bound regime: stream W once through the skinny MFMA kernel
        return wsgemm_nt((const bf16*)a, k, w, nullptr, false, c, (const bf16*)bias, 0, 1, m, n, k, st);
    if (dt == PGK_BF16 && use_gemm256(m, n, k))
        return gemm256_bf16_nt((const bf16*)a, (const bf16*)w, (const bf16*)bias, c, false, m, n, k, st);
    if (dt == PGK_BF16 && gemm128s_ok(m, n, k)) return gemm128s_bf16_nt((const bf16*)a, (const bf16*)w, (const bf16*)bias, c, 0, 1, m, n, k, st);
    if (dt == PGK_BF16) return dispatch_mfma<bf16, B_NT>((const bf16*)a, w, nullptr, (const bf16*)bias, (bf16*)c, m, n, k, st);
    return dispatch_mfma<f16, B_NT>((const f16*)a, w, nullptr, (const f16*)bias, (f16*)c, m, n, k, st);
}

pgk_status pgk_gemm_nn(const void* a, const void* b, void* c, int m, int n, int k, pgk_dtype dt, pgk_stream s) {
    PGK_REQUIRE(a && b && c, "pgk_gemm_nn: null pointer");
    PGK_REQUIRE(m >= 0 && n > 0 && k > 0, "pgk_gemm_nn: bad shape M=%d N=%d K=%d", m, n, k);
    if (!m) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const bool al = aligned16(a) && aligned16(b) && (k % 8 == 0) && (n % 8 == 0);
    if (dt == PGK_F32 || !al) {
        dim3 grid(ceil_div(n, 64), ceil_div(m, 64));
        PGK_DISPATCH_FLOAT(dt, "pgk_gemm_nn", (gemm_simple_kernel<T, false><<<grid, 256, 0, st>>>((const T*)a, (const T*)b, nullptr, (T*)c, m, n, k)));
        PGK_LAUNCH_CHECK();
        return PGK_OK;
    }
    if (dt == PGK_BF16) return dispatch_mfma<bf16, B_NN>((const bf16*)a, b, nullptr, nullptr, (bf16*)c, m, n, k, st);
    return dispatch_mfma<f16, B_NN>((const f16*)a, b, nullptr, nullptr, (f16*)c, m, n, k, st);
}

pgk_status pgk_w8a16_gemm_nk(const void* a, const uint8_t* w_nk, const void* scale, void* c, int m, int n, int k,
                             pgk_stream s) {
    PGK_REQUIRE(a && w_nk && scale && c, "pgk_w8a16_gemm_nk: null pointer");
    PGK_REQUIRE(m >= 0 && n > 0 && k > 0, "pgk_w8a16_gemm_nk: bad shape M=%d N=%d K=%d", m, n, k);
    PGK_REQUIRE(k % 128 == 0 && n % 128 == 0, "pgk_w8a16_gemm_nk: K=%d, N=%d must be multiples of the 128x128 scale block", k, n);
    PGK_REQUIRE(aligned16(a) && aligned16(w_nk), "pgk_w8a16_gemm_nk: operands must be 16-byte aligned");
    if (!m) return PGK_OK;
    if (m <= 128) return wsgemm_nt((const bf16*)a, k, w_nk, (const bf16*)scale, true, c, nullptr, 0, 1, m, n, k, resolve_stream(s));
    return engine_gemm_nt((const bf16*)a, w_nk, (const bf16*)scale, true, c, false, m, n, k, resolve_stream(s), false);
}

pgk_status pgk_w8a16_gemm_kn(const void* a, const uint8_t* b_kn, const void* scale, void* c, int m, int n, int k,
                             pgk_stream s) {
    PGK_REQUIRE(a && b_kn && scale && c, "pgk_w8a16_gemm_kn: null pointer");
    PGK_REQUIRE(m >= 0 && n > 0 && k > 0, "pgk_w8a16_gemm_kn: bad shape M=%d N=%d K=%d", m, n, k);
    PGK_REQUIRE(k % 128 == 0 && n % 128 == 0, "pgk_w8a16_gemm_kn: K=%d, N=%d must be multiples of the 128x128 scale block", k, n);
    PGK_REQUIRE(aligned16(a) && aligned16(b_kn), "pgk_w8a16_gemm_kn: operands must be 16-byte aligned");
    if (!m) return PGK_OK;
    return dispatch_mfma<bf16, B_KN_FP8>((const bf16*)a, b_kn, (const bf16*)scale, nullptr, (bf16*)c, m, n, k, resolve_stream(s));
}

}  // extern "C"
