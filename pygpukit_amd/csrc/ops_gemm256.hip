// Large-tile MFMA GEMMs for gfx950 (prefill at S >= 256):  C[M,N] = A[M,K] . W[N,K]^T
//
//   bf16:  A, W bf16, fp32 accumulate, bf16 C or fp32 "+=" (the engine's residual stream)
//   fp8 :  A, W OCP e4m3 with 128-wide block scales (see ops_fp8_gemm.hip for the contract), same outputs
//
// Structure: 256 x 256 output tile per workgroup, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 8 x 4 MFMA tiles of
// 16 x 16 held in 128 accumulator registers; K tiles of 128 BYTES per row (64 bf16 / 128 fp8).  Both operand
// tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), 16 bytes
// per lane, one wave-instruction = 8 rows x 128 B written linearly; the bank swizzle is applied on the SOURCE
// address (lane p of a row fetches chunk p ^ (row & 7)) and again on the fragment reads, so the LDS image is the
// XOR-swizzled one without a scatter.  Two LDS buffers (128 KiB): the DMA of tile t+1 is issued before the MFMAs
// of tile t; one counted wait + one raw barrier per K tile.  All LDS lives in ONE dynamic array (a second
// __shared__ object makes hipcc wait vmcnt(0) before every fragment read).
// Workgroup ids are remapped so that the tiles an XCD works on form a compact patch of the output (shared A / W
// panels stay in that XCD's L2).
//
// The reference calls CUTLASS / cuBLASLt here (native/ops/matmul/matmul.cu:142-235); nothing of theirs is used.

#include "gemv_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

typedef __bf16 bf16x8_g __attribute__((ext_vector_type(8)));
typedef float f32x4_g __attribute__((ext_vector_type(4)));
typedef int i32x8_g __attribute__((ext_vector_type(8)));

constexpr int G2_BM = 256, G2_BN = 256, G2_THREADS = 512;
constexpr int G2_TILE = 256 * 128;   // bytes of one operand tile

// byte offset of 16-byte chunk c (0..7) of row r in a [rows][128 B] tile
__device__ __forceinline__ int g2_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_addr, +1024).
// Written as inline asm on purpose: issued through __builtin_amdgcn_global_load_lds, hipcc (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of the next ds_read of the kernel - it cannot tell the buffer being filled from the
// buffer being read - and the prefetch is drained the moment it is issued.  The waits that order these DMAs
// before the fragment reads are the explicit vmcnt + barrier pairs in the kernels below.
__device__ __forceinline__ void g2_dma16(const void* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
// the same with a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset
__device__ __forceinline__ void g2_dma16_so(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ uint32_t g2_lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// workgroup id -> output tile: ids are dealt round-robin to the 8 XCDs, so XCD x owns ids x, x+8, ...; give it a
// contiguous run of tiles, and walk the tiles in groups of 4 tile-rows so a run is a compact patch.
__device__ __forceinline__ void g2_tile_of(int id, int nwg, int ntm, int ntn, int& tm, int& tn) {
    const int q = nwg / 8, r = nwg % 8, xcd = id % 8, idx = id / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any nwg
    constexpr int GM = 4;
    const int per_group = GM * ntn, g = t / per_group, first = g * GM;
    const int gsz = min(ntm - first, GM), in = t % per_group;
    tm = first + in % gsz;
    tn = in / gsz;
}

template <int EPI>   // 0: bf16 C store (+bias); 1: fp32 C +=
__global__ __launch_bounds__(G2_THREADS) void gemm256_bf16_kernel(const bf16* A, const bf16* W, const bf16* bias, void* Cv,
                                                                   int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // A[2] | W[2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;

    // DMA sources: wave w moves rows 32w .. 32w+31 of each tile, 8 rows per instruction; lane -> (row l>>3, chunk p = l&7)
    const int drow = wid * 32 + (lane >> 3);
    const int dchunk = (lane & 7) ^ (lane >> 3);            // logical chunk that lands at linear position l&7
    const bf16* a_src[4];
    const bf16* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_src[i] = A + (size_t)min(m0 + drow + 8 * i, M - 1) * K + dchunk * 8;
        w_src[i] = W + (size_t)min(n0 + drow + 8 * i, N - 1) * K + dchunk * 8;
    }
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    auto stage = [&](int kt, int buf) {
        const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G2_TILE + wid * 4096);
        const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds0 + (2 + buf) * G2_TILE + wid * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g2_dma16(a_src[i] + (size_t)kt * 64, a_dst + i * 1024);
            g2_dma16(w_src[i] + (size_t)kt * 64, w_dst + i * 1024);
        }
    };

    f32x4_g acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + l15 (base a multiple of 16), chunk = ks*4 + q
    const int f_off0 = l15 * 128 + (((0 + q) ^ (l15 & 7)) << 4);
    const int f_off1 = l15 * 128 + (((4 + q) ^ (l15 & 7)) << 4);
    const int a_base = wr * 128 * 128, w_base = wc * 64 * 128;

    const int nk = K / 64;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* As = g2_smem + buf * G2_TILE + a_base;
        const char* Ws = g2_smem + (2 + buf) * G2_TILE + w_base;
        // Both k-steps' fragments are requested before the first MFMA: the second set's LDS latency runs under the
        // first 32 MFMAs (one wave cannot rely on its SIMD neighbour for that).  The sched_barriers pin this order;
        // left alone, hipcc keeps only two reads in flight and waits for one before every group of 4 MFMAs.
        uint4 fa0[8], fb0[4], fa1[8], fb1[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) fa0[i] = *reinterpret_cast<const uint4*>(As + i * 2048 + f_off0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const uint4*>(Ws + j * 2048 + f_off0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa1[i] = *reinterpret_cast<const uint4*>(As + i * 2048 + f_off1);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const uint4*>(Ws + j * 2048 + f_off1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, fa0[i]),
                                                                    __builtin_bit_cast(bf16x8_g, fb0[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, fa1[i]),
                                                                    __builtin_bit_cast(bf16x8_g, fb1[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + l15;
        if (col >= N) continue;
        float b = 0.f;
        if constexpr (EPI == 0) b = bias ? to_f(bias[col]) : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 128 + i * 16 + q * 4 + r;
                if (row >= M) continue;
                if constexpr (EPI == 0) reinterpret_cast<bf16*>(Cv)[(size_t)row * N + col] = from_f<bf16>(acc[i][j][r] + b);
                else reinterpret_cast<float*>(Cv)[(size_t)row * N + col] += acc[i][j][r];
            }
    }
}

// ---- bf16, staggered phases: the two waves of a SIMD alternate between "read + DMA issue" and "32 MFMAs" ---------
// In the kernel above both waves of a SIMD leave each barrier together, request fragments together and want the MFMA
// pipe together: the pipe idles while both wait (57 % busy).  Here the stage is half a K tile ([256 rows][32 k] per
// operand, 64-byte rows; four stages in the same 128 KiB, three in flight) and a wave's work on stage s is two phases,
//     A(s): request the 12 fragments of stage s, issue the DMA of stage s+3, wait for the fragments  | barrier
//     B(s): 32 MFMAs                                                                                  | barrier
// with waves 4-7 (the second wave of every SIMD) running ONE PHASE BEHIND waves 0-3 (one extra barrier up front, one
// at the end for the others): whenever one wave of a SIMD multiplies, the other one reads.  A stage is refilled one
// phase after its last reader finished (lgkmcnt(0) before that phase's barrier); its arrival is ordered by a counted
// vmcnt(8) - two younger stages may stay in flight - at the end of the phase before the first group reads it, which
// is B(s) for waves 0-3 and A(s) for waves 4-7.  DMAs past the end of K re-read the last stage (never consumed) so the
// count is constant.  64-byte rows: chunk c of row r at r*64 + ((c ^ ((r >> 2) & 3)) << 4) (conflict-free b128 reads).
template <int EPI>
__global__ __launch_bounds__(G2_THREADS) void gemm256s_bf16_kernel(const bf16* A, const bf16* W, const bf16* bias, void* Cv,
                                                                    int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // 4 stages x (A 16 KiB | W 16 KiB)
    constexpr int HALF = 256 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    const int late = __builtin_amdgcn_readfirstlane(wr);     // wave-uniform by construction: scalar branches around barriers
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;

    const int drow = wid * 32 + (lane >> 2);                 // one DMA instruction = 16 rows x 64 B
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const bf16* a_src0 = A + (size_t)min(m0 + drow, M - 1) * K + dchunk * 8;
    const bf16* a_src1 = A + (size_t)min(m0 + drow + 16, M - 1) * K + dchunk * 8;
    const bf16* w_src0 = W + (size_t)min(n0 + drow, N - 1) * K + dchunk * 8;
    const bf16* w_src1 = W + (size_t)min(n0 + drow + 16, N - 1) * K + dchunk * 8;
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    const int nh = K / 32;
    auto stage = [&](int h, int buf) {
        const int k = min(h, nh - 1) * 32;
        const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds0 + buf * 2 * HALF + wid * 2048);
        g2_dma16(a_src0 + k, a_dst);
        g2_dma16(a_src1 + k, a_dst + 1024);
        g2_dma16(w_src0 + k, a_dst + HALF);
        g2_dma16(w_src1 + k, a_dst + HALF + 1024);
    };

    f32x4_g acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};
    const int f_off = l15 * 64 + ((q ^ ((l15 >> 2) & 3)) << 4);
    const int a_base = wr * 128 * 64, w_base = HALF + wc * 64 * 64;
    uint4 fa[8], fb[4];

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();                  // waves 4-7 start one phase behind
#define G2S_STEP(S, BUF)                                                                                     \
    {   /* phase A */                                                                                        \
        const char* As = g2_smem + (BUF) * 2 * HALF + a_base + f_off;                                        \
        const char* Ws = g2_smem + (BUF) * 2 * HALF + w_base + f_off;                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + i * 1024); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Ws + j * 1024); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        stage((S) + 3, ((BUF) + 3) & 3);                                                                     \
        if (late) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* phase B */                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, fa[i]),     \
                                                                    __builtin_bit_cast(bf16x8_g, fb[j]), acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (!late) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                          \
        __builtin_amdgcn_s_barrier();                                                                        \
    }
    int s = 0;
    for (; s + 4 <= nh; s += 4) {
        G2S_STEP(s, 0) G2S_STEP(s + 1, 1) G2S_STEP(s + 2, 2) G2S_STEP(s + 3, 3)
    }
    if (s < nh) {   // K % 64 == 0: an even number of stages, so two remain at most
        G2S_STEP(s, 0) G2S_STEP(s + 1, 1)
    }
#undef G2S_STEP
    if (!late) __builtin_amdgcn_s_barrier();                 // match the extra barrier of waves 4-7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // retire the phantom DMAs before the LDS is given back

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + l15;
        if (col >= N) continue;
        float b = 0.f;
        if constexpr (EPI == 0) b = bias ? to_f(bias[col]) : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 128 + i * 16 + q * 4 + r;
                if (row >= M) continue;
                if constexpr (EPI == 0) reinterpret_cast<bf16*>(Cv)[(size_t)row * N + col] = from_f<bf16>(acc[i][j][r] + b);
                else reinterpret_cast<float*>(Cv)[(size_t)row * N + col] += acc[i][j][r];
            }
    }
}

// ---- fp8 x fp8 with 128-wide block scales on the same structure --------------------------------------------
// K tile = 128 fp8 = one scale block: per tile the 256 row scales of A (fp32) and the two weight-block scales of
// the tile's 256 columns (bf16) are DMA'd into a small LDS array next to the operand tiles - every global load
// in the loop is an LDS-DMA, so no compiler-counted vmcnt ever drains the prefetch.  One v_mfma_f32_16x16x128_f8f6f4
// per 16x16 output tile and K tile, issued on a zero accumulator and folded in with the scale product.
__device__ __forceinline__ void g2_dma4(const void* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void g2_dma2(const void* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_ushort %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}

constexpr int G2_SCALE_BYTES = 256 * 4 + 256;   // per buffer: 256 fp32 row scales | 64 lanes x 4 B of weight-scale slots (2 used)

template <int EPI>   // 0: bf16 C store; 1: fp32 C +=
__global__ __launch_bounds__(G2_THREADS) void gemm256_fp8_kernel(const uint8_t* A, const float* sa, const uint8_t* W, const bf16* sw,
                                                                  void* Cv, int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // A[2] | W[2] | scales[2]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;
    const int KB = K >> 7, NB = (N + 127) >> 7;

    // DMA sources as (uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): 8 offset registers instead of 16
    // pointer registers in a kernel that lives at the 256-register line, and the per-K-tile advance is one scalar add.
    // Offsets are relative to the tile's first row (operand tiles span at most 255 rows x K <= 2^31 bytes).
    const int drow = wid * 32 + (lane >> 3);
    const int dchunk = (lane & 7) ^ (lane >> 3);
    const uint8_t* a_tile = A + (size_t)m0 * K;
    const uint8_t* w_tile = W + (size_t)n0 * K;
    uint32_t a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (uint32_t)(min(drow + 8 * i, M - 1 - m0) * K + dchunk * 16);
        w_off[i] = (uint32_t)(min(drow + 8 * i, N - 1 - n0) * K + dchunk * 16);
    }
    // scale DMAs: waves 0-3 fetch 64 row scales each; wave 4 fetches the (up to) two weight-block scales
    const float* sa_src = sa + (size_t)min(m0 + (wid & 3) * 64 + lane, M - 1) * KB;
    const bf16* sw_src = sw + (size_t)min(2 * tn + (lane & 1), NB - 1) * KB;
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    auto stage = [&](int kt, int buf) {
        const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G2_TILE + wid * 4096);
        const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds0 + (2 + buf) * G2_TILE + wid * 4096);
        const uint32_t s_dst = __builtin_amdgcn_readfirstlane(lds0 + 4 * G2_TILE + buf * G2_SCALE_BYTES + (wid < 4 ? wid * 256 : 1024));
        const uint8_t* ab = a_tile + (size_t)kt * 128;
        const uint8_t* wb = w_tile + (size_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g2_dma16_so(ab, a_off[i], a_dst + i * 1024);
            g2_dma16_so(wb, w_off[i], w_dst + i * 1024);
        }
        if (wid < 4) g2_dma4(sa_src + kt, s_dst);
        else if (wid == 4) g2_dma2(sw_src + kt, s_dst);
    };

    f32x4_g acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};

    // fragment reads: row = base + l15, the lane's 32 bytes are chunks 2q and 2q+1
    const int f_lo = l15 * 128 + (((2 * q) ^ (l15 & 7)) << 4);
    const int f_hi = l15 * 128 + (((2 * q + 1) ^ (l15 & 7)) << 4);
    const int a_base = wr * 128 * 128, w_base = wc * 64 * 128;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The 32 MFMAs of a K tile run as ONE software pipeline.  (1) The scale-and-add of product n is issued behind MFMA
    // n + 2, each statement fenced by a sched_barrier of its own: hipcc otherwise sinks every product's FMAs directly under
    // its MFMA (s_nop 11 + two v_pk_fma per MFMA in the first version's ISA - the matrix pipe idle for a third of every
    // 32-cycle slot, the packed FMAs costing more issue time than four scalar ones).  (2) The A fragments and row scales come
    // in four groups of two 16-row tiles; group g + 1 is requested before group g's MFMAs, so only the first group's LDS
    // latency is exposed per K tile (the first version read four tiles, waited, multiplied, twice per K tile).
    // (3) Scalar FMAs on purpose: beside MFMAs a v_pk_fma_f32 costs more than the two v_fma_f32 it replaces.
    for (int kt = 0; kt < KB; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KB) stage(kt + 1, buf ^ 1);
        const char* As = g2_smem + buf * G2_TILE + a_base;
        const char* Ws = g2_smem + (2 + buf) * G2_TILE + w_base;
        const char* Ss = g2_smem + 4 * G2_TILE + buf * G2_SCALE_BYTES;
        i32x8_g fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 lo = *reinterpret_cast<const uint4*>(Ws + j * 2048 + f_lo), hi = *reinterpret_cast<const uint4*>(Ws + j * 2048 + f_hi);
            fb[j] = i32x8_g{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
        // (a sub-dword LDS-DMA still strides the lanes by 4 bytes: lane l's 16 bits land at +4l, zero-extended)
        const float swv = to_f(*reinterpret_cast<const bf16*>(Ss + 1024 + (wc >> 1) * 4));   // this wave's 64 columns lie in one block
        i32x8_g fa[2][2];
        f32x4_g sraw[2];        // one set: a group's raw row scales are consumed (x the block scale) before the next group's are requested
        auto load_group = [&](int g, int slot) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int off = (2 * g + u) * 2048;
                const uint4 lo = *reinterpret_cast<const uint4*>(As + off + f_lo), hi = *reinterpret_cast<const uint4*>(As + off + f_hi);
                fa[slot][u] = i32x8_g{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                sraw[u] = *reinterpret_cast<const f32x4_g*>(Ss + (wr * 128 + (2 * g + u) * 16 + q * 4) * 4);
            }
        };
        load_group(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f32x4_g t[3];
        float sc[2][2][4];      // [slot][tile of the group][row]
#define G2F_FMA(NPREV)                                                                                              \
    {                                                                                                               \
        constexpr int g_ = (NPREV) >> 3, u_ = ((NPREV) >> 2) & 1, j_ = (NPREV) & 3, i_ = 2 * g_ + u_;               \
        asm volatile("" : "+v"(t[(NPREV) % 3]));      /* ordered behind the MFMA issued just above (volatile asm statements keep their order) */ \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i_][j_][r] = fmaf(t[(NPREV) % 3][r], sc[g_ & 1][u_][r], acc[i_][j_][r]); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#define G2F_STEP(N)                                                                                                 \
    {                                                                                                               \
        constexpr int g_ = (N) >> 3, u_ = ((N) >> 2) & 1, j_ = (N) & 3;                                            \
        if constexpr (((N) & 7) == 0) {                                                                             \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                           \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) sc[g_ & 1][u][r] = sraw[u][r] * swv;                  \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            if constexpr (g_ < 3) { load_group(g_ + 1, (g_ + 1) & 1); __builtin_amdgcn_sched_barrier(0); }          \
        }                                                                                                           \
        t[(N) % 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[g_ & 1][u_], fb[j_], f32x4_g{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0); \
        asm volatile("" : "+v"(t[(N) % 3]));          /* pins this MFMA here: instruction selection otherwise sinks every product's FMAs under its own MFMA, sched_barriers notwithstanding */ \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if constexpr ((N) >= 2) G2F_FMA((N) - 2)                                                                    \
    }
        __builtin_amdgcn_s_setprio(1);
        G2F_STEP(0) G2F_STEP(1) G2F_STEP(2) G2F_STEP(3) G2F_STEP(4) G2F_STEP(5) G2F_STEP(6) G2F_STEP(7)
        G2F_STEP(8) G2F_STEP(9) G2F_STEP(10) G2F_STEP(11) G2F_STEP(12) G2F_STEP(13) G2F_STEP(14) G2F_STEP(15)
        G2F_STEP(16) G2F_STEP(17) G2F_STEP(18) G2F_STEP(19) G2F_STEP(20) G2F_STEP(21) G2F_STEP(22) G2F_STEP(23)
        G2F_STEP(24) G2F_STEP(25) G2F_STEP(26) G2F_STEP(27) G2F_STEP(28) G2F_STEP(29) G2F_STEP(30) G2F_STEP(31)
        G2F_FMA(30) G2F_FMA(31)
        __builtin_amdgcn_s_setprio(0);
#undef G2F_STEP
#undef G2F_FMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wc * 64 + j * 16 + l15;
        if (col >= N) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 128 + i * 16 + q * 4 + r;
                if (row >= M) continue;
                if constexpr (EPI == 0) reinterpret_cast<bf16*>(Cv)[(size_t)row * N + col] = from_f<bf16>(acc[i][j][r]);
                else reinterpret_cast<float*>(Cv)[(size_t)row * N + col] += acc[i][j][r];
            }
    }
}

pgk_status gemm256_fp8_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, void* c, bool accum_f32, int M, int N,
                          int K, hipStream_t st) {
    PGK_REQUIRE(K % 128 == 0 && K >= 128, "gemm256 fp8: K=%d must be a multiple of 128", K);
    constexpr size_t LDS = 4 * (size_t)G2_TILE + 2 * G2_SCALE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, G2_BM), ntn = ceil_div(N, G2_BN);
    if (accum_f32) gemm256_fp8_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, c, M, N, K, ntm, ntn);
    else gemm256_fp8_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, c, M, N, K, ntm, ntn);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// bf16 NT on the 256^2 structure; caller guarantees K % 64 == 0 and 16-byte aligned rows
pgk_status gemm256_bf16_nt(const bf16* A, const bf16* W, const bf16* bias, void* C, bool accum_f32, int M, int N, int K,
                           hipStream_t st) {
    PGK_REQUIRE(K % 64 == 0 && K >= 64, "gemm256: K=%d must be a multiple of 64", K);
    constexpr size_t LDS = 4 * (size_t)G2_TILE;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, G2_BM), ntn = ceil_div(N, G2_BN);
    const char* e = getenv("PGK_GEMM256S");          // 0: two full stages, waves in lockstep; default: staggered phases
    if (!e || atoi(e) != 0) {
        static bool attr_s = false;
        if (!attr_s) {
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            attr_s = true;
        }
        if (accum_f32) gemm256s_bf16_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn);
        else gemm256s_bf16_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn);
        PGK_CHECK_HIP(hipGetLastError());
        return PGK_OK;
    }
    if (accum_f32) gemm256_bf16_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn);
    else gemm256_bf16_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk
