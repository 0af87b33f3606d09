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

#include "gemm_epilogues.hip.h"
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

// fp8 image swizzle (see gemm256_fp8_kernel): (r & 7) ^ (5 for rows 8-15 of every 16)
__device__ __forceinline__ int g2_sw8(int r) { return (r & 7) ^ ((r & 8) ? 5 : 0); }

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

// ---- SwiGLU epilogue (EPI 2 of the kernels below): W is the fused [2 I, K] gate / up weight, the workgroup's 256 B-tile rows are
// gate rows [128 tn, +128) followed by up rows [I + 128 tn, +128), so output tile tn holds act columns [128 tn, +128): waves
// wc 0 / 1 hold gate columns, wc 2 / 3 the matching up columns.  The accumulators go to LDS as the bf16 tile the plain kernel
// would have stored ([256][256] bf16 = the 128 KiB the operand stages occupied; 2-byte writes, the 32-byte run of a lane
// quarter XOR-ed by q so the four quarters of a wave hit different banks), then 16 consecutive lanes take one row's 128
// act columns, 8 each: act = bf16(silu(g) * u) from the bf16-ROUNDED g and u - the arithmetic of swiglu_rows_kernel
// (engine.hip) on the values the plain epilogue stores, so the result is bit-identical to GEMM + swiglu_rows_kernel - and
// store 16 bytes of bf16, or (QOUT) 8 e4m3 codes with the row's scale for this 128-column block (absmax / 448 over the 16
// lanes, quantize_fp8_rows' contract): exactly the fp8 x fp8 down projection's A operand.  Saves the [M][2 I] bf16 round
// trip through HBM (Llama-3-8B shape, S = 4096: 235 MB written + read back = ~75 us of a 1.4 ms layer) and a launch.
template <bool QOUT, int TN = 4>   // TN = n-fragments per wave column: 4 = 128 act columns per tile; 3 = 96 (bf16 output only)
__device__ __forceinline__ void g2_swiglu_epilogue(char* smem, const f32x4_g (&acc)[8][TN], int tid, int m0, int tn, int M, int I,
                                                   void* outv, float* out_scales) {
    static_assert(TN == 4 || !QOUT, "the e4m3 epilogue needs whole 128-column scale blocks");
    constexpr int WN = TN * 16, ACT = 2 * WN, ROWB = 4 * WN * 2, CH = ACT / 8;       // act columns per tile, bytes per LDS row, 8-column chunks per row
    const int lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    __builtin_amdgcn_s_barrier();                    // every wave is past its last operand read (DMAs retired by the caller)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 128 + i * 16 + q * 4 + r, colb = (wc * WN + j * 16 + l15) * 2;
                *reinterpret_cast<bf16*>(smem + row * ROWB + (colb ^ (q << 5))) = from_f<bf16>(acc[i][j][r]);   // (row >> 2) & 3 == q; the XOR stays inside a 128-byte group
            }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 256 * CH / G2_THREADS; ++it) {
        const int item = it * G2_THREADS + tid, row = item / CH, c = item % CH, sw = ((row >> 2) & 3) << 5;
        Vec<bf16> g, u;
        g.raw = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((c * 16) ^ sw));
        u.raw = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((ACT * 2 + c * 16) ^ sw));
        float gf[8], uf[8];
        g.to_float(gf);
        u.to_float(uf);
#pragma unroll
        for (int e = 0; e < 8; ++e) gf[e] = gf[e] / (1.0f + __expf(-gf[e])) * uf[e];
        g.from_float(gf);
        const size_t grow = (size_t)m0 + row;
        if constexpr (QOUT) {
            g.to_float(gf);
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(gf[e]));
            amax = group16_max(amax);
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            uint2 o;
            o.x = pack_fp8x4(gf[0] / sc, gf[1] / sc, gf[2] / sc, gf[3] / sc);
            o.y = pack_fp8x4(gf[4] / sc, gf[5] / sc, gf[6] / sc, gf[7] / sc);
            if (grow < (size_t)M) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(outv) + grow * I + tn * ACT + c * 8) = o;
                if (c == 0) out_scales[grow * (I >> 7) + tn] = sc;
            }
        } else {
            if (grow < (size_t)M) g.store(reinterpret_cast<bf16*>(outv) + grow * I + tn * ACT + c * 8);
        }
    }
}

// ---- fp32 "+=" epilogue (EPI 1) through LDS: the residual stream is read and written as whole 512-byte row segments ---------
// The direct form - every lane adds its 128 accumulators to 128 separate words, 16 lanes per 64 contiguous bytes - cost
// ~30-40 us per launch at the Llama shapes (o_proj 161 us against 121 for the bf16-store kernel of the same product), all of
// it exposed: a one-wave grid has no second tile to hide an epilogue behind.  Here, per 64-column quarter of the tile: the
// residual words are requested first (8 float4 per lane), the owning wave column puts its accumulators into LDS ([256][64]
// fp32 over the operand stages, lane quarters XOR-ed apart), and every lane adds and stores 8 float4 (256-byte row segments).
__device__ __forceinline__ void g2_accum_epilogue(char* smem, const f32x4_g (&acc)[8][4], int tid, int m0, int n0, int M, int N, float* C) {
    const int lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    // one pass per 64-column quarter (= one wave column): 8 float4 per lane in flight (16 - halves - spilled in these kernels)
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
        float4 old[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = it * G2_THREADS + tid, row = item >> 4, c = item & 15;
            old[it] = *reinterpret_cast<const float4*>(C + (size_t)min(m0 + row, M - 1) * N + min(n0 + h * 64 + c * 4, N - 4));
        }
        __syncthreads();                             // operand stages (h = 0) / the previous quarter's tile are no longer read
        if (wc == h) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wr * 128 + i * 16 + q * 4 + r, colb = (j * 16 + l15) * 4;
                        *reinterpret_cast<float*>(smem + row * 256 + (colb ^ (q << 6))) = acc[i][j][r];      // (row >> 2) & 3 == q
                    }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = it * G2_THREADS + tid, row = item >> 4, c = item & 15, sw = ((row >> 2) & 3) << 6;
            float4 v = *reinterpret_cast<const float4*>(smem + row * 256 + ((c * 16) ^ sw));
            const int grow = m0 + row, gcol = n0 + h * 64 + c * 4;
            if (grow >= M || gcol >= N) continue;
            v.x += old[it].x; v.y += old[it].y; v.z += old[it].z; v.w += old[it].w;
            *reinterpret_cast<float4*>(C + (size_t)grow * N + gcol) = v;
        }
    }
}

// ---- QKV-heads epilogue (EPI 3): the accumulators go to LDS as the bf16 tile the plain kernel would have stored (the layout of
// g2_swiglu_epilogue), a tile = 256 token rows x two head slots, 16 consecutive lanes take one (row, head) and finish it
// (qkv_head_finish: per-head RMSNorm, RoPE, q -> qkv buffer, k / v -> cache): bit-identical to GEMM + qknorm_rope_kvwrite_kernel.
__device__ __forceinline__ void g2_qkv_heads_epilogue(char* smem, const f32x4_g (&acc)[8][4], int tid, int m0, int tn, int M, int N, bf16* qkv,
                                                      const QkvHeadArgs& hd) {
    const int lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 128 + i * 16 + q * 4 + r, colb = (wc * 64 + j * 16 + l15) * 2;
                *reinterpret_cast<bf16*>(smem + row * 512 + (colb ^ (q << 5))) = from_f<bf16>(acc[i][j][r]);
            }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int item = it * G2_THREADS + tid, row = item >> 5, c = item & 31, sw = ((row >> 2) & 3) << 5;
        Vec<bf16> raw;
        raw.raw = *reinterpret_cast<const uint4*>(smem + row * 512 + ((c * 16) ^ sw));
        float x[8];
        raw.to_float(x);
        qkv_head_finish(x, c & 15, 2 * tn + (c >> 4), m0 + row, M, hd, qkv, N);
    }
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
// count is constant.  64-byte rows: chunk c of row r at r*64 + ((c ^ ((r >> 2) & 2)) << 4).  ds_read_b128 is served in the
// lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), not in runs of 16 lanes: the first version's swizzle
// (r >> 2) & 3 was conflict-free for runs of 16 and 2-way conflicted for the real groups (SQ_LDS_BANK_CONFLICT = half of
// SQ_LDS_IDX_ACTIVE on the round-2 kernel); (r >> 2) & 2 is conflict-free for them (exhaustive check; counter 0 in
// profiles/r03_gemm256s_pmc.txt).
// TN = n-fragments per wave column: 4 = 256-column tiles; 3 = 192-column tiles (EPI 0 only), for shapes whose 256-tiles leave a
// half-empty last round: Llama's QKV (4096 x 6144) is 384 tiles = 1.5 rounds of the 256 CUs, 512 tiles of 256 x 192 are two
// full rounds of three quarters the work.  The W stage then has 192 rows = 12 DMA instructions: waves 0-3 move two (rows 0-127),
// waves 4-7 one (rows 128-191), so the "two younger stages in flight" count is vmcnt(8) for the former, vmcnt(6) for the latter.
template <int EPI, int TN = 4>   // EPI 0: bf16 C store (+bias); 1: fp32 C +=; 2: SwiGLU (W = fused gate / up rows, N = I act columns, C = bf16 act [M][I])
__global__ __launch_bounds__(G2_THREADS) void gemm256s_bf16_kernel(const bf16* A, const bf16* W, const bf16* bias, void* Cv,
                                                                    int M, int N, int K, int ntm, int ntn, int packed) {
    static_assert(TN == 4 || (TN == 3 && (EPI == 0 || EPI == 2)), "192-column tiles: bf16 store and SwiGLU only");
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // 4 stages x (A 16 KiB | W 16 KiB)
    constexpr int HALF = 256 * 64, BN = TN * 64, WN = TN * 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    const int late = __builtin_amdgcn_readfirstlane(wr);     // wave-uniform by construction: scalar branches around barriers
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * G2_BM, n0 = tn * BN;

    const int drow = wid * 32 + (lane >> 2);                 // one DMA instruction = 16 rows x 64 B
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 2);      // = (lane & 3) ^ g2s_sw(row & 15), row & 15 = lane >> 2
    const bf16* a_src0 = A + (size_t)min(m0 + drow, M - 1) * K + dchunk * 8;
    const bf16* a_src1 = A + (size_t)min(m0 + drow + 16, M - 1) * K + dchunk * 8;
    // B-tile row r of an EPI-2 workgroup: gate row 128 tn + r for r < 128, up row N + 128 tn + (r - 128) after that (N = I)
    const int wtile_row = (TN == 4 || !late) ? drow : 128 + (wid - 4) * 16 + (lane >> 2);
    constexpr int GATE = TN * 32;     // EPI 2: gate rows per B tile (128 / 96), the matching up rows follow
    const int wrow = (EPI == 2) ? (wtile_row < GATE ? tn * GATE + wtile_row : N + tn * GATE + wtile_row - GATE) : n0 + wtile_row;
    const int wlast = (EPI == 2) ? 2 * N - 1 : N - 1;
    // W row-major [rows][K], or (packed) the fragment-major copy of ops_pkgemm.hip: block (16-row tile nt, k-step ks) = 1 KiB at
    // ((nt K/32 + ks) 64 + chunk 16 + row) 8 elements.  One DMA instruction moves 16 rows x 32 k = exactly one such block; only
    // the per-lane source offset inside it differs (this lane's logical chunk and row), and a stage advances 512 elements.
    const bf16 *w_src0, *w_src1;
    const int wk = packed ? 512 : 32;
    if (packed) {
        const int r_in = lane >> 2, nt_last = (wlast >> 4);
        w_src0 = W + ((size_t)min((wrow - r_in) >> 4, nt_last) * (K >> 5) * 64 + dchunk * 16 + r_in) * 8;
        w_src1 = W + ((size_t)min((wrow - r_in + 16) >> 4, nt_last) * (K >> 5) * 64 + dchunk * 16 + r_in) * 8;
    } else {
        w_src0 = W + (size_t)min(wrow, wlast) * K + dchunk * 8;
        w_src1 = W + (size_t)min(wrow + 16, wlast) * K + dchunk * 8;
    }
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    const int nh = K / 32;
    auto stage = [&](int h, int buf) {
        const int hc = min(h, nh - 1), k = hc * 32;
        const size_t kw = (size_t)hc * wk;
        const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds0 + buf * 2 * HALF + wid * 2048);
        g2_dma16(a_src0 + k, a_dst);
        g2_dma16(a_src1 + k, a_dst + 1024);
        if constexpr (TN == 4) {
            g2_dma16(w_src0 + kw, a_dst + HALF);
            g2_dma16(w_src1 + kw, a_dst + HALF + 1024);
        } else if (!late) {
            g2_dma16(w_src0 + kw, a_dst + HALF);
            g2_dma16(w_src1 + kw, a_dst + HALF + 1024);
        } else {
            g2_dma16(w_src0 + kw, __builtin_amdgcn_readfirstlane(lds0 + buf * 2 * HALF + HALF + 128 * 64 + (wid - 4) * 1024));
        }
    };
    // "this wave's share of the oldest of three stages in flight has landed"
    auto wait_two_younger = [&]() {
        if constexpr (TN == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (late) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    f32x4_g acc[8][TN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};
    const int f_off = l15 * 64 + ((q ^ ((l15 >> 2) & 2)) << 4);
    const int a_base = wr * 128 * 64, w_base = HALF + wc * WN * 64;
    uint4 fa[8], fb[TN];

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    wait_two_younger();
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();                  // waves 4-7 start one phase behind
#define G2S_STEP(S, BUF)                                                                                     \
    {   /* phase A */                                                                                        \
        const char* As = g2_smem + (BUF) * 2 * HALF + a_base + f_off;                                        \
        const char* Ws = g2_smem + (BUF) * 2 * HALF + w_base + f_off;                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + i * 1024); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(Ws + j * 1024); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        stage((S) + 3, ((BUF) + 3) & 3);                                                                     \
        if (late) wait_two_younger();                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* phase B */                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, fa[i]),     \
                                                                    __builtin_bit_cast(bf16x8_g, fb[j]), acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (!late) wait_two_younger();                                                                       \
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

    if constexpr (EPI == 2) {
        g2_swiglu_epilogue<false, TN>(g2_smem, acc, tid, m0, tn, M, N, Cv, nullptr);
        return;
    }
    if constexpr (EPI == 1 && TN == 4) {
        if ((N & 3) == 0) {       // whole float4s per row (every engine shape); otherwise the word-wise form below
            g2_accum_epilogue(g2_smem, acc, tid, m0, n0, M, N, reinterpret_cast<float*>(Cv));
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * WN + j * 16 + l15;
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

// ---- bf16, 128 x 128 tiles on the same staged LDS-DMA pipeline: the shapes too small for 192 tiles of 256 x 256 -------------
// (Qwen3-0.6B at S = 2048: N = 1024 projections = 128 tiles, QKV 512 tiles with K = 1024.)  The register-staged 128-tile
// kernel of ops_gemm.hip keeps ONE K step in flight (global -> registers -> LDS -> barrier): with one or two workgroups
// per CU every step costs a memory round trip - 1.9 us per 64 k at M = 2048, N = 1024 (30 us for 8.6 GFLOP).  Here: 4 waves
// (2 x 2, 64 x 64 each = 64 accumulator registers), stages of 32 k ([128 rows][64 B] per operand, 16 KiB per stage, the
// swizzle of gemm256s), four stages of which three are in flight behind a counted vmcnt(8), ONE barrier per stage; 64 KiB
// of LDS and < 128 registers, so two workgroups share a CU and one's MFMAs cover the other's barrier and fragment reads -
// no stagger needed.  blockIdx.y = K split (EPI 2: fp32 slab z of [splits][M][N], summed by the consumer).
// Epilogue through LDS: the fp32 tile (64 KiB = the four stages) is written with the lane quarters XOR-ed apart, then read
// back as rows - 32 lanes x 16 B = one 512-byte row segment per instruction - for coalesced stores (EPI 1: float4
// read-modify-write of the residual stream, all loads issued before the first store).
constexpr int G1_THREADS = 256, G1_HALF = 128 * 64;
template <int EPI>   // 0: bf16 C store (+bias); 1: fp32 C +=; 2: fp32 split-K slab; 3: QKV heads (QkvHeadArgs: tile column = head slot);
                     // 4: SwiGLU (W = fused gate / up rows, N = I act columns, C = bf16 act [M][I]; tile tn = act columns [64 tn, +64))
__global__ __launch_bounds__(G1_THREADS, 2) void gemm128s_bf16_kernel(const bf16* A, const bf16* W, const bf16* bias, void* Cv,
                                                                       int M, int N, int K, int ntm, int ntn, int kps, QkvHeadArgs hd, int packed) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // 4 stages x (A 8 KiB | W 8 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, q = lane >> 4, l15 = lane & 15;
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = (int)blockIdx.y * kps, kend = min(K, kbeg + kps);

    const int drow = wid * 32 + (lane >> 2);                 // one DMA instruction = 16 rows x 64 B
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 2);
    const bf16* a_src0 = A + (size_t)min(m0 + drow, M - 1) * K + kbeg + dchunk * 8;
    const bf16* a_src1 = A + (size_t)min(m0 + drow + 16, M - 1) * K + kbeg + dchunk * 8;
    // EPI 4: B-tile rows 0-63 = gate rows 64 tn .., rows 64-127 = the matching up rows N + 64 tn .. (N = I); a wave's 32 rows stay in one half
    const int wrow = (EPI == 4) ? (drow < 64 ? tn * 64 + drow : N + tn * 64 + drow - 64) : n0 + drow;
    const int wlast = (EPI == 4) ? 2 * N - 1 : N - 1;
    // W row-major, or (packed) the fragment-major copy: see gemm256s_bf16_kernel
    const bf16 *w_src0, *w_src1;
    const int wk = packed ? 512 : 32;
    if (packed) {
        const int r_in = lane >> 2, nt_last = (wlast >> 4);
        w_src0 = W + (((size_t)min((wrow - r_in) >> 4, nt_last) * (K >> 5) + (kbeg >> 5)) * 64 + dchunk * 16 + r_in) * 8;
        w_src1 = W + (((size_t)min((wrow - r_in + 16) >> 4, nt_last) * (K >> 5) + (kbeg >> 5)) * 64 + dchunk * 16 + r_in) * 8;
    } else {
        w_src0 = W + (size_t)min(wrow, wlast) * K + kbeg + dchunk * 8;
        w_src1 = W + (size_t)min(wrow + 16, wlast) * K + kbeg + dchunk * 8;
    }
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    const int nh = (kend - kbeg) / 32;
    auto stage = [&](int h, int buf) {
        const int hc = min(h, nh - 1), k = hc * 32;          // past the end: re-read the last stage (never consumed), the count stays constant
        const size_t kw = (size_t)hc * wk;
        const uint32_t a_dst = __builtin_amdgcn_readfirstlane(lds0 + buf * 2 * G1_HALF + wid * 2048);
        g2_dma16(a_src0 + k, a_dst);
        g2_dma16(a_src1 + k, a_dst + 1024);
        g2_dma16(w_src0 + kw, a_dst + G1_HALF);
        g2_dma16(w_src1 + kw, a_dst + G1_HALF + 1024);
    };

    f32x4_g acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};
    const int f_off = l15 * 64 + ((q ^ ((l15 >> 2) & 2)) << 4);
    const int a_base = wr * 64 * 64, w_base = G1_HALF + wc * 64 * 64;
    uint4 fa[4], fb[4];

    stage(0, 0);
    stage(1, 1);
    stage(2, 2);
    // iteration s: own share of stage s landed (two younger stages may stay in flight) | barrier: everyone's share landed AND
    // everyone finished reading stage s - 1, whose buffer the DMA of stage s + 3 - issued right behind the reads - refills
#define G1S_STEP(S, BUF)                                                                                      \
    {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                      \
        __builtin_amdgcn_s_barrier();                                                                         \
        const char* As = g2_smem + (BUF) * 2 * G1_HALF + a_base + f_off;                                      \
        const char* Ws = g2_smem + (BUF) * 2 * G1_HALF + w_base + f_off;                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + i * 1024); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const uint4*>(Ws + j * 1024); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        stage((S) + 3, ((BUF) + 3) & 3);                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, fa[i]),      \
                                                                    __builtin_bit_cast(bf16x8_g, fb[j]), acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
    int s = 0;
    for (; s + 4 <= nh; s += 4) {
        G1S_STEP(s, 0) G1S_STEP(s + 1, 1) G1S_STEP(s + 2, 2) G1S_STEP(s + 3, 3)
    }
    if (s < nh) {   // (kend - kbeg) % 64 == 0: an even number of stages, so two remain at most
        G1S_STEP(s, 0) G1S_STEP(s + 1, 1)
    }
#undef G1S_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // retire the phantom DMAs: the stages become the C tile
    __builtin_amdgcn_s_barrier();

    // fp32 tile [128][128]: row R's 64-byte group g at g ^ ((R >> 2) & 3) - (R >> 2) & 3 == q for the accumulator layout
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * 64 + i * 16 + q * 4 + r, colb = (wc * 64 + j * 16 + l15) * 4;
                *reinterpret_cast<float*>(g2_smem + row * 512 + (colb ^ (q << 6))) = acc[i][j][r];
            }
    __syncthreads();
    if constexpr (EPI == 4) {
        // act = bf16(silu(g) * u) from the bf16-rounded g and u (the arithmetic of swiglu_rows_kernel / g2_swiglu_epilogue): 8 lanes per row
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = it * G1_THREADS + tid, row = item >> 3, c = item & 7, sw = ((row >> 2) & 3) << 6;
            const char* rp = g2_smem + row * 512;
            const float4 g0 = *reinterpret_cast<const float4*>(rp + ((c * 32) ^ sw)), g1 = *reinterpret_cast<const float4*>(rp + ((c * 32 + 16) ^ sw));
            const float4 u0 = *reinterpret_cast<const float4*>(rp + ((256 + c * 32) ^ sw)), u1 = *reinterpret_cast<const float4*>(rp + ((256 + c * 32 + 16) ^ sw));
            const float gfr[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, ufr[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            Vec<bf16> g, u;
            g.from_float(gfr);
            u.from_float(ufr);
            float gf[8], uf[8];
            g.to_float(gf);
            u.to_float(uf);
#pragma unroll
            for (int e = 0; e < 8; ++e) gf[e] = gf[e] / (1.0f + __expf(-gf[e])) * uf[e];
            g.from_float(gf);
            const int grow = m0 + row, gcol = tn * 64 + c * 8;
            if (grow < M && gcol < N) g.store(reinterpret_cast<bf16*>(Cv) + (size_t)grow * N + gcol);
        }
    } else if constexpr (EPI == 3) {
        // 16 consecutive lanes = one (token row, head): the arithmetic of qknorm_rope_kvwrite_kernel on the bf16-rounded row,
        // so the result is bit-identical to GEMM (bf16 store) + that kernel.  Tile column tn is the head slot.
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = it * G1_THREADS + tid, row = item >> 4, c = item & 15, sw = ((row >> 2) & 3) << 6;
            const float4 v0 = *reinterpret_cast<const float4*>(g2_smem + row * 512 + ((c * 32) ^ sw));
            const float4 v1 = *reinterpret_cast<const float4*>(g2_smem + row * 512 + ((c * 32 + 16) ^ sw));
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            Vec<bf16> raw;
            raw.from_float(f);
            float x[8];
            raw.to_float(x);
            qkv_head_finish(x, c, tn, m0 + row, M, hd, reinterpret_cast<bf16*>(Cv), N);
        }
    } else if constexpr (EPI == 0) {
        // 8 columns per lane: two float4 -> 16 bytes of bf16
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = it * G1_THREADS + tid, row = item >> 4, c = item & 15, sw = ((row >> 2) & 3) << 6;
            const float4 v0 = *reinterpret_cast<const float4*>(g2_smem + row * 512 + ((c * 32) ^ sw));
            const float4 v1 = *reinterpret_cast<const float4*>(g2_smem + row * 512 + ((c * 32 + 16) ^ sw));
            const int grow = m0 + row, gcol = n0 + c * 8;
            if (grow >= M || gcol >= N) continue;
            float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += to_f(bias[gcol + e]);
            }
            Vec<bf16> o;
            o.from_float(f);
            o.store(reinterpret_cast<bf16*>(Cv) + (size_t)grow * N + gcol);
        }
    } else {
        float* C = reinterpret_cast<float*>(Cv) + (EPI == 2 ? (size_t)blockIdx.y * M * N : 0);
        float4 old[16];
        if constexpr (EPI == 1) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int item = it * G1_THREADS + tid, row = item >> 5, c = item & 31;
                const int grow = min(m0 + row, M - 1), gcol = min(n0 + c * 4, N - 4);
                old[it] = *reinterpret_cast<const float4*>(C + (size_t)grow * N + gcol);
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int item = it * G1_THREADS + tid, row = item >> 5, c = item & 31, sw = ((row >> 2) & 3) << 6;
            float4 v = *reinterpret_cast<const float4*>(g2_smem + row * 512 + ((c * 16) ^ sw));
            const int grow = m0 + row, gcol = n0 + c * 4;
            if (grow >= M || gcol >= N) continue;
            if constexpr (EPI == 1) { v.x += old[it].x; v.y += old[it].y; v.z += old[it].z; v.w += old[it].w; }
            *reinterpret_cast<float4*>(C + (size_t)grow * N + gcol) = v;
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

__device__ __forceinline__ void g2_dma4_so(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void g2_dma2_so(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

constexpr int G2_SCALE_BYTES = 256 * 4 + 256;   // per buffer: 256 fp32 row scales | 64 lanes x 4 B of weight-scale slots (2 used)

template <int EPI>   // 0: bf16 C store; 1: fp32 C +=; 2: SwiGLU + e4m3 quantisation (W = fused gate / up rows, N = I; Cv = codes [M][I], Cs = scales [M][I/128]);
                     // 3: QKV heads (QkvHeadArgs; a tile holds two head slots)
__global__ __launch_bounds__(G2_THREADS) void gemm256_fp8_kernel(const uint8_t* A, const float* sa, const uint8_t* W, const bf16* sw,
                                                                  void* Cv, float* Cs, int M, int N, int K, int ntm, int ntn, QkvHeadArgs hd) {
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];   // A[2] | W[2] | scales[2]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // in an SGPR: it enters the DMAs' scalar bases
    const int wr = wid >> 2, wc = wid & 3, q = lane >> 4, l15 = lane & 15;
    int tm, tn;
    g2_tile_of(blockIdx.x, ntm * ntn, ntm, ntn, tm, tn);
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;
    const int KB = K >> 7, NB = (N + 127) >> 7;

    // DMA sources as (uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): 8 offset registers instead of 16
    // pointer registers in a kernel that lives at the 256-register line, and the per-K-tile advance is one scalar add.
    // Offsets are relative to the tile's first row (operand tiles span at most 255 rows x K <= 2^31 bytes).
    // LDS image: [256 rows][128 B], 16-byte chunk c of row r at position c ^ g2_sw8(r).  ds_read_b128 is served in the lane
    // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) - not in runs of 16 lanes - and a fragment read has lanes (q =
    // lane >> 4, row lane & 15) fetch chunks 2 q / 2 q + 1: with the plain c ^ (r & 7) swizzle of the bf16 image, rows r and
    // r + 8 of such a group meet in the same banks (SQ_LDS_BANK_CONFLICT = 42 % of the LDS cycles of the first version);
    // XOR-ing 5 into the rows 8-15 of every 16 separates them for both reads (exhaustive check over the four lane groups).
    //
    // DMA sources: (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset).  Instruction i of a wave moves rows
    // 32 wid + 8 i + (lane >> 3); the row step 8 i K goes into the scalar base, so a lane needs TWO offsets for all eight
    // instructions of a stage (even / odd i differ in the swizzle only) - this kernel lives at the 256-register line, and
    // sixteen 64-bit source pointers (the first version) or eight offsets spill.  Whole tiles only: the caller sends shapes
    // with M % 256 or N % 256 != 0 to the 128-tile kernel (clamped row offsets are loop-invariant, so hipcc computes all
    // sixteen up front and spills them - and with them a few accumulators).
    // The per-lane offsets are RECOMPUTED from the lane id at every stage (a handful of vector instructions per tile): held
    // across the loop they are spilled at its fullest point, and every reload waits on the vector-memory counter.
    const uint8_t* a_tile = A + ((size_t)m0 + wid * 32) * K;
    // EPI 2: B-tile rows 0-127 = gate rows 128 tn .., rows 128-255 = up rows N + 128 tn .. (N = I): waves 0-3 stage gate rows, 4-7 up rows,
    // and the tile's two weight-scale blocks are block rows tn and I / 128 + tn
    const uint8_t* w_tile = W + (size_t)((EPI == 2) ? (wid < 4 ? tn * 128 + wid * 32 : N + tn * 128 + (wid - 4) * 32) : n0 + wid * 32) * K;
    const float* sa_tile = sa + ((size_t)m0 + (wid & 3) * 64) * KB;      // waves 0-3: 64 row scales each (one per lane)
    const bf16* sw_tile = sw + (size_t)((EPI == 2) ? tn : 2 * tn) * KB;  // wave 4: the tile's two weight-block scales (lanes 0 / 1)
    const uint32_t sw_step = (uint32_t)((EPI == 2) ? (N >> 7) : 1) * (uint32_t)(KB * 2);   // bytes between the two
    const uint32_t lds0 = g2_lds_addr(g2_smem);
    // operand tiles of K tile kt -> stage buf, as eight 1-KiB pieces (piece p: rows 8 (p >> 1) .. + 8 of this wave's 32, A for
    // even p, W for odd p).  Branch-free: the pieces are issued from the middle of the MFMA pipeline, one behind each MFMA of
    // a tile's last group - a DMA costs ~60 cycles of issue, which back to back (the first version) is ~500 cycles in which
    // neither wave of a SIMD issues an MFMA, and behind an MFMA mostly runs under it.
    auto stage_piece = [&](int kt, int buf, int p, uint32_t off_e, uint32_t off_o) {
        const int i = p >> 1;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + ((p & 1) ? 2 + buf : buf) * G2_TILE + wid * 4096 + i * 1024);
        const uint8_t* base = ((p & 1) ? w_tile : a_tile) + (size_t)kt * 128 + (size_t)8 * i * K;
        g2_dma16_so(base, (i & 1) ? off_o : off_e, dst);
    };
    auto stage_offsets = [&](uint32_t& off_e, uint32_t& off_o) {
        uint32_t ln = (uint32_t)lane;
        asm volatile("" : "+v"(ln));                                     // opaque: keeps the offsets out of the loop-invariant set
        const uint32_t r8 = ln >> 3, c8 = ln & 7;
        off_e = r8 * (uint32_t)K + ((c8 ^ r8) << 4);                     // rows 0-7 of 16: swizzle r & 7
        off_o = r8 * (uint32_t)K + ((c8 ^ r8 ^ 5u) << 4);                // rows 8-15: ^ 5 (g2_sw8)
    };
    auto stage = [&](int kt, int buf) {
        uint32_t off_e, off_o;
        stage_offsets(off_e, off_o);
#pragma unroll
        for (int p = 0; p < 8; ++p) stage_piece(kt, buf, p, off_e, off_o);
    };
    // scales of K tile kt -> scale stage buf (per-wave branches: issued at the top of a tile, outside the pipeline)
    auto stage_scales = [&](int kt, int buf) {
        const uint32_t s_dst = __builtin_amdgcn_readfirstlane(lds0 + 4 * G2_TILE + buf * G2_SCALE_BYTES + (wid < 4 ? wid * 256 : 1024));
        if (wid < 4) g2_dma4_so(sa_tile + kt, (uint32_t)lane * (uint32_t)(KB * 4), s_dst);
        else if (wid == 4) g2_dma2_so(sw_tile + kt, ((uint32_t)lane & 1u) * sw_step, s_dst);
    };

    f32x4_g acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_g{0.f, 0.f, 0.f, 0.f};

    // fragment reads: row = base + l15, the lane's 32 bytes are chunks 2q and 2q+1
    const int f_lo = l15 * 128 + (((2 * q) ^ g2_sw8(l15)) << 4);
    const int f_hi = l15 * 128 + (((2 * q + 1) ^ g2_sw8(l15)) << 4);
    const int a_base = wr * 128 * 128, w_base = wc * 64 * 128;

    // The MFMAs run as ONE software pipeline over the whole K loop, 32 per K tile (4 groups of 2 A row-tiles x 4 B column-tiles):
    // (1) the scale-and-add of product n is issued behind MFMA n + 2, every MFMA pinned by an empty volatile asm on its result:
    //     instruction selection otherwise sinks each product's FMAs directly under its own MFMA (s_nop 11 + two v_pk_fma per
    //     MFMA in the first version's ISA; sched_barrier only binds the later machine scheduler).  Scalar FMAs on purpose:
    //     beside MFMAs a v_pk_fma_f32 costs more issue time than the two v_fma_f32 it replaces;
    // (2) A fragments and row scales of group g + 1 are requested before group g's MFMAs;
    // (3) ONE barrier per K tile, at the start of its last group: by then every wave holds all fragments of tile kt in
    //     registers (stage kt & 1 is free) and its own share of tile kt + 1 - requested a whole tile earlier - has landed.
    //     Behind it the DMA of tile kt + 2 goes out, one piece behind each MFMA of the group, and the NEXT tile's first
    //     fragments are requested: group 0's A fragments at once, its row scales two steps later, each B fragment right
    //     after its last MFMA of this tile (the last group runs column-major for that).  No tile starts with an LDS round
    //     trip, a burst of DMA issue or a barrier.  Everything inside the pipeline is branch-free - one conditional DMA
    //     in it cut the basic block in two and made hipcc spill a hundred registers per tile - so the scales of tile
    //     kt + 1 (1 KiB under per-wave branches) are requested at the top of tile kt; their stage was last read in tile kt - 1.
    stage(0, 0);
    stage_scales(0, 0);
    stage(min(1, KB - 1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    i32x8_g fb[4], fa[2][2];
    f32x4_g t[3];
    f32x4_g sc[2][2];        // [slot][row-tile of the group]: raw row scales as they come from LDS, multiplied by the block scale in place
    float swv;
    uint32_t off_e = 0, off_o = 0;
    auto ldfrag = [&](const char* base) -> i32x8_g {
        const uint4 lo = *reinterpret_cast<const uint4*>(base + f_lo), hi = *reinterpret_cast<const uint4*>(base + f_hi);
        return i32x8_g{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    auto load_frags = [&](int buf, int g, int slot) {
        const char* As = g2_smem + buf * G2_TILE + a_base;
#pragma unroll
        for (int u = 0; u < 2; ++u) fa[slot][u] = ldfrag(As + (2 * g + u) * 2048);
    };
    // a group's row scales go into the slot the group BEFORE the previous one used: that one's last two scale-and-adds run
    // two steps into the next group, so the scales are requested at step 2 of a group, the fragments at step 0
    auto load_scales = [&](int buf, int g, int slot) {
        const char* Ss = g2_smem + 4 * G2_TILE + buf * G2_SCALE_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) sc[slot][u] = *reinterpret_cast<const f32x4_g*>(Ss + (wr * 128 + (2 * g + u) * 16 + q * 4) * 4);
    };
    // (a sub-dword LDS-DMA still strides the lanes by 4 bytes: lane l's 16 bits land at +4l, zero-extended)
    auto load_swv = [&](int buf) -> float { return to_f(*reinterpret_cast<const bf16*>(g2_smem + 4 * G2_TILE + buf * G2_SCALE_BYTES + 1024 + (wc >> 1) * 4)); };
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = ldfrag(g2_smem + 2 * G2_TILE + w_base + j * 2048);
    load_frags(0, 0, 0);
    load_scales(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // step n of a tile: group g = n / 8; groups 0-2 walk (row-tile u, column-tile j) row-major, group 3 column-major
#define G2F_U(N) (((N) < 24) ? (((N) >> 2) & 1) : ((N) & 1))
#define G2F_J(N) (((N) < 24) ? ((N) & 3) : (((N) & 7) >> 1))
#define G2F_FMA(NPREV)                                                                                              \
    {                                                                                                               \
        constexpr int g_ = (NPREV) >> 3, u_ = G2F_U(NPREV), j_ = G2F_J(NPREV), i_ = 2 * g_ + u_;                    \
        asm volatile("" : "+v"(t[(NPREV) % 3]));      /* ordered behind the MFMA issued just above (volatile asm statements keep their order) */ \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i_][j_][r] = fmaf(t[(NPREV) % 3][r], sc[g_ & 1][u_][r], acc[i_][j_][r]); \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#define G2F_STEP(N)                                                                                                 \
    {                                                                                                               \
        constexpr int g_ = (N) >> 3, u_ = G2F_U(N), j_ = G2F_J(N);                                                  \
        if constexpr ((N) == 24) {                                                                                  \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                             \
            __builtin_amdgcn_s_barrier();                                                                           \
            stage_offsets(off_e, off_o);                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if constexpr (((N) & 7) == 0) {                                                                             \
            if constexpr ((N) == 0) swv = load_swv(buf);                                                            \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                           \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) sc[g_ & 1][u][r] *= swv;                              \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            if constexpr (g_ < 3) load_frags(buf, g_ + 1, (g_ + 1) & 1);                                            \
            else load_frags(buf ^ 1, 0, 0);                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if constexpr (((N) & 7) == 2) {                                                                             \
            if constexpr (g_ < 3) load_scales(buf, g_ + 1, (g_ + 1) & 1);                                           \
            else load_scales(buf ^ 1, 0, 0);                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        t[(N) % 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[g_ & 1][u_], fb[j_], f32x4_g{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0); \
        asm volatile("" : "+v"(t[(N) % 3]));          /* pins this MFMA here */                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if constexpr ((N) >= 24) {                    /* one DMA piece of tile kt + 2 behind each MFMA of the last group (past the end: re-reads the last tile into a stage nobody reads any more) */ \
            stage_piece(min(kt + 2, KB - 1), buf, (N) - 24, off_e, off_o);                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if constexpr ((N) >= 24 && ((N) & 1) == 1) {  /* last use of column-tile j_ in this tile: fetch the next tile's */ \
            fb[j_] = ldfrag(g2_smem + (2 + (buf ^ 1)) * G2_TILE + w_base + j_ * 2048);                              \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        if constexpr ((N) >= 2) G2F_FMA((N) - 2)                                                                    \
    }
    __builtin_amdgcn_s_setprio(1);
    for (int kt = 0; kt < KB; ++kt) {
        const int buf = kt & 1;
        stage_scales(min(kt + 1, KB - 1), buf ^ 1);
        G2F_STEP(0) G2F_STEP(1) G2F_STEP(2) G2F_STEP(3) G2F_STEP(4) G2F_STEP(5) G2F_STEP(6) G2F_STEP(7)
        G2F_STEP(8) G2F_STEP(9) G2F_STEP(10) G2F_STEP(11) G2F_STEP(12) G2F_STEP(13) G2F_STEP(14) G2F_STEP(15)
        G2F_STEP(16) G2F_STEP(17) G2F_STEP(18) G2F_STEP(19) G2F_STEP(20) G2F_STEP(21) G2F_STEP(22) G2F_STEP(23)
        G2F_STEP(24) G2F_STEP(25) G2F_STEP(26) G2F_STEP(27) G2F_STEP(28) G2F_STEP(29) G2F_STEP(30) G2F_STEP(31)
        G2F_FMA(30) G2F_FMA(31)
    }
    __builtin_amdgcn_s_setprio(0);
#undef G2F_STEP
#undef G2F_FMA
#undef G2F_U
#undef G2F_J
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // retire the phantom DMAs before the LDS is given back

    if constexpr (EPI == 2) {
        g2_swiglu_epilogue<true>(g2_smem, acc, tid, m0, tn, M, N, Cv, Cs);
        return;
    }
    if constexpr (EPI == 1) {     // N % 256 == 0 here
        g2_accum_epilogue(g2_smem, acc, tid, m0, n0, M, N, reinterpret_cast<float*>(Cv));
        return;
    }
    if constexpr (EPI == 3) {
        g2_qkv_heads_epilogue(g2_smem, acc, tid, m0, tn, M, N, reinterpret_cast<bf16*>(Cv), hd);
        return;
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
    PGK_REQUIRE(K % 128 == 0 && K >= 128 && M % G2_BM == 0 && N % G2_BN == 0, "gemm256 fp8: M=%d N=%d must be multiples of 256 and K=%d of 128", M, N, K);
    constexpr size_t LDS = 4 * (size_t)G2_TILE + 2 * G2_SCALE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, G2_BM), ntn = ceil_div(N, G2_BN);
    if (accum_f32) gemm256_fp8_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, c, nullptr, M, N, K, ntm, ntn, QkvHeadArgs{});
    else gemm256_fp8_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, c, nullptr, M, N, K, ntm, ntn, QkvHeadArgs{});
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// QKV projection finished per head in the epilogue (QkvHeadArgs): N = (Hq + 2 Hkv) x 128, two head slots per tile
pgk_status gemm256_fp8_qkv_heads_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, bf16* qkv, int M, int N, int K,
                                    const QkvHeadArgs& hd, hipStream_t st) {
    PGK_REQUIRE(K % 128 == 0 && K >= 128 && M % G2_BM == 0 && N % G2_BN == 0 && N == (hd.hq + 2 * hd.hkv) * 128,
                "gemm256 fp8 qkv heads: M=%d, N=%d must be multiples of 256, K=%d of 128, N = (Hq + 2 Hkv) x 128", M, N, K);
    constexpr size_t LDS = 4 * (size_t)G2_TILE + 2 * G2_SCALE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = M / G2_BM, ntn = N / G2_BN;
    gemm256_fp8_kernel<3><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, qkv, nullptr, M, N, K, ntm, ntn, hd);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// act codes / scales = quantise(silu(a . Wg^T) * (a . Wu^T)), w = fused [2 I, K] gate / up codes with [2 I / 128][K / 128] block scales
pgk_status gemm256_fp8_swiglu_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, uint8_t* q_out, float* s_out, int M,
                                 int I, int K, hipStream_t st) {
    PGK_REQUIRE(K % 128 == 0 && K >= 128 && M % G2_BM == 0 && I % 128 == 0, "gemm256 fp8 swiglu: M=%d must be a multiple of 256, I=%d and K=%d of 128", M, I, K);
    PGK_REQUIRE((const void*)a != (const void*)q_out, "gemm256 fp8 swiglu: output aliases the activation operand");
    constexpr size_t LDS = 4 * (size_t)G2_TILE + 2 * G2_SCALE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_fp8_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = M / G2_BM, ntn = I / 128;
    gemm256_fp8_kernel<2><<<ntm * ntn, G2_THREADS, LDS, st>>>(a, sa, w, sw, q_out, s_out, M, I, K, ntm, ntn, QkvHeadArgs{});
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// bf16 NT on the 256^2 structure; caller guarantees K % 64 == 0 and 16-byte aligned rows
pgk_status gemm256_bf16_nt(const bf16* A, const bf16* W, const bf16* bias, void* C, bool accum_f32, int M, int N, int K,
                           hipStream_t st, bool packed) {
    PGK_REQUIRE(K % 64 == 0 && K >= 64, "gemm256: K=%d must be a multiple of 64", K);
    PGK_REQUIRE(!packed || N % 16 == 0, "gemm256: the fragment-major weight copy has whole 16-row tiles (N=%d)", N);
    constexpr size_t LDS = 4 * (size_t)G2_TILE;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, G2_BM), ntn = ceil_div(N, G2_BN);
    const char* e = getenv("PGK_GEMM256S");          // 0: two full stages, waves in lockstep; default: staggered phases
    if (packed || !e || atoi(e) != 0) {
        static bool attr_s = false;
        if (!attr_s) {
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
            attr_s = true;
        }
        // 192-column tiles when they fill the rounds of the chip better (cost = rounds x work per tile)
        const int ntn3 = N / 192;
        const bool narrow = !accum_f32 && N % 192 == 0 &&
                            0.75 * ceil_div(ntm * ntn3, 256) < (double)ceil_div(ntm * ntn, 256) - 0.01;
        if (narrow) {
            static bool attr_n = false;
            if (!attr_n) {
                PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
                attr_n = true;
            }
            gemm256s_bf16_kernel<0, 3><<<ntm * ntn3, G2_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn3, packed ? 1 : 0);
        } else if (accum_f32) gemm256s_bf16_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn, packed ? 1 : 0);
        else gemm256s_bf16_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn, packed ? 1 : 0);
        PGK_CHECK_HIP(hipGetLastError());
        return PGK_OK;
    }
    if (accum_f32) gemm256_bf16_kernel<1><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn);
    else gemm256_bf16_kernel<0><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// act[M][I] = bf16(silu(A . Wg^T) * (A . Wu^T)), W = fused [2 I, K] gate / up weight (staggered kernel, SwiGLU epilogue)
pgk_status gemm256_bf16_swiglu_nt(const bf16* A, const bf16* W, bf16* act, int M, int I, int K, hipStream_t st, bool packed) {
    PGK_REQUIRE(K % 64 == 0 && K >= 64 && I % 128 == 0, "gemm256 swiglu: K=%d must be a multiple of 64 and I=%d of 128", K, I);
    constexpr size_t LDS = 4 * (size_t)G2_TILE;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_bf16_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, G2_BM), ntn = I / 128;
    // 96 act columns per tile (192-column B tiles) when that fills the rounds of the chip better: Qwen3-0.6B at S = 2048 is
    // 8 x 24 = 192 tiles of 128 (three quarters of the CUs, one round) or 8 x 32 = 256 tiles of 96 (all of them)
    if (I % 96 == 0 && 0.75 * ceil_div(ntm * (I / 96), 256) < (double)ceil_div(ntm * ntn, 256) - 0.01) {
        gemm256s_bf16_kernel<2, 3><<<ntm * (I / 96), G2_THREADS, LDS, st>>>(A, W, nullptr, act, M, I, K, ntm, I / 96, packed ? 1 : 0);
    } else {
        gemm256s_bf16_kernel<2><<<ntm * ntn, G2_THREADS, LDS, st>>>(A, W, nullptr, act, M, I, K, ntm, ntn, packed ? 1 : 0);
    }
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// 128 x 128 tiles; mode 0: bf16 C (+bias), 1: fp32 C +=, 2: fp32 slabs [splits][M][N] (K split into `splits` runs of whole 64-k steps)
bool gemm128s_ok(int M, int N, int K) { return M > 128 && K % 64 == 0 && N % 8 == 0 && N >= 8; }
pgk_status gemm128s_bf16_nt(const bf16* A, const bf16* W, const bf16* bias, void* C, int mode, int splits, int M, int N, int K, hipStream_t st,
                            const QkvHeadArgs* heads, bool packed) {
    PGK_REQUIRE(!packed || N % 16 == 0, "gemm128s: the fragment-major weight copy has whole 16-row tiles (N=%d)", N);
    const int pk = packed ? 1 : 0;
    PGK_REQUIRE(gemm128s_ok(M, N, K) && mode >= 0 && mode <= 4 && splits >= 1 && (splits == 1 || mode == 2), "gemm128s: M=%d N=%d K=%d mode=%d splits=%d", M, N, K, mode, splits);
    PGK_REQUIRE(mode != 4 || N % 64 == 0, "gemm128s: the SwiGLU epilogue needs I=%d to be a multiple of 64", N);
    PGK_REQUIRE(mode != 3 || (heads && N == (heads->hq + 2 * heads->hkv) * 128), "gemm128s: the QKV-heads epilogue needs N=%d = (Hq + 2 Hkv) x 128", N);
    const QkvHeadArgs hd = heads ? *heads : QkvHeadArgs{};
    constexpr size_t LDS = 8 * (size_t)G1_HALF;
    static bool attr_done = false;
    if (!attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128s_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128s_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128s_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128s_bf16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128s_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_done = true;
    }
    const int ntm = ceil_div(M, 128), ntn = mode == 4 ? N / 64 : ceil_div(N, 128);      // SwiGLU: N = I act columns, 64 per tile
    const int kps = ceil_div(ceil_div(K, splits), 64) * 64;
    const dim3 grid(ntm * ntn, ceil_div(K, kps));
    PGK_REQUIRE((int)grid.y == splits, "gemm128s: K=%d does not split into %d runs of whole 64-k steps", K, splits);
    if (mode == 0) gemm128s_bf16_kernel<0><<<grid, G1_THREADS, LDS, st>>>(A, W, bias, C, M, N, K, ntm, ntn, kps, hd, pk);
    else if (mode == 1) gemm128s_bf16_kernel<1><<<grid, G1_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn, kps, hd, pk);
    else if (mode == 2) gemm128s_bf16_kernel<2><<<grid, G1_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn, kps, hd, pk);
    else if (mode == 3) gemm128s_bf16_kernel<3><<<grid, G1_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn, kps, hd, pk);
    else gemm128s_bf16_kernel<4><<<grid, G1_THREADS, LDS, st>>>(A, W, nullptr, C, M, N, K, ntm, ntn, kps, hd, pk);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk
