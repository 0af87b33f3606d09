// Packed-weight skinny GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T,  M <= 128 (prefill of a short prompt).
//
// Why a second weight layout.  The MFMA B fragment of a ROW-MAJOR weight matrix puts lane l on row l & 15: the 64 lanes
// of one 16-byte load touch 64 separate 16-byte pieces, and a CU's texture addresser takes those about one lane per
// clock - 70 cycles per wave-instruction against 20-25 for 1 KiB of consecutive bytes (tools/micro/ta_probe.hip:
// 8.9 vs 25-31 TB/s chip-wide from L2).  A workgroup that wants its whole weight slice in flight at once pays that per
// instruction, serially, on its one CU.  So the engine keeps a second copy of the layer weights in FRAGMENT-MAJOR
// order: block (n-tile, k-step) = 16 rows x 32 k = 1 KiB, lane l's 16 bytes at offset 16 l, blocks of one n-tile
// consecutive along k.  A wave streaming its n-tile reads one contiguous run (K / 32 KiB), every instruction fully
// coalesced, no LDS round trip for the read-once operand.  288 GB of HBM is what pays for the copy (bf16 layers only).
//
// Shape of a workgroup (4 waves):
//   * m-block of 32 rows (MT = 2 m-tiles): the grid is (n-blocks x m-blocks x K-splits), with the m-blocks of one
//     n-block 8 ids apart so they land on the same XCD and share the weight bytes in its L2;
//   * the whole activation block [32][K_wg] goes global -> LDS by LDS-DMA up front (XOR-swizzled on the source side),
//     the first 16 k-steps of every wave's weight run are requested right behind it: ONE memory round trip, then a
//     register ring refilled 16 k-steps ahead of use;
//   * a wave owns one or two n-tiles (two for the fused epilogues) and reads each A fragment once for both.
// Epilogues: bf16 store, fp32 split-K slabs, fp32 +=, SwiGLU (gate tile and its up tile in the same wave), and the
// QKV head epilogue - per-head RMSNorm + RoPE + KV-cache write, a workgroup owning one head of 128 with tiles d and
// d + 64 in the same wave so the rotate-half pair sits in one lane.
// Reference: the same projections go through cuBLASLt / CUTLASS 128x128 tiles with separate rmsnorm / rope / silu /
// mul launches (src/pygpukit/llm/layers/attention.py, mlp.py; native/ops/matmul/matmul.cu:142-235).

#include "gemv_core.hip.h"
#include "pgk_internal.h"
#include "pkgemm.hip.h"

namespace pgk {

typedef __bf16 bf16x8_p __attribute__((ext_vector_type(8)));
typedef float f32x4_p __attribute__((ext_vector_type(4)));

constexpr int PK_THREADS = 256;
constexpr int PK_MT = 2;           // m-tiles per workgroup
constexpr int PK_MB = PK_MT * 16;  // rows per m-block
constexpr int PK_RING = 16;        // k-steps of weights in flight per n-tile

__device__ __forceinline__ void pk_dma16(const void* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ uint32_t pk_lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// 16-byte chunk c of a row lives at chunk position pk_swz(c, row): involution inside aligned groups of 16 chunks
__device__ __forceinline__ int pk_swz(int c, int row) { return (c & ~15) | ((c ^ row) & 15); }

__global__ void pack_weights_kernel(const bf16* w, bf16* wp, int N, int K) {
    const size_t chunks = (size_t)N * K / 8, stride = (size_t)gridDim.x * blockDim.x;
    const int ksn = K / 32;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < chunks; t += stride) {
        const int l = (int)(t & 63);
        const size_t blk = t >> 6;
        const int ks = (int)(blk % ksn), nt = (int)(blk / ksn);
        const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)(nt * 16 + (l & 15)) * K + ks * 32 + 8 * (l >> 4));
        *reinterpret_cast<uint4*>(wp + t * 8) = v;
    }
}

pgk_status pack_weights_bf16(const void* w, void* wp, int N, int K, hipStream_t st) {
    PGK_REQUIRE(N % 16 == 0 && K % 32 == 0, "pack_weights: N=%d must be a multiple of 16 and K=%d of 32", N, K);
    const size_t chunks = (size_t)N * K / 8;
    const int grid = (int)(chunks / 256 > 4096 ? 4096 : (chunks + 255) / 256);
    pack_weights_kernel<<<grid, 256, 0, st>>>((const bf16*)w, (bf16*)wp, N, K);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// fp8-weight (w8a16) engines get the SAME bf16 fragment-major working copy, dequantised once at engine creation:
// wp = bf16(lut[code] * block scale), the value the reference's w8a16 GEMM puts in front of its MMA
// (native/ops/matmul/gemm/w8a16_bf16/sm120/w8a16_gemm.cu:187-203) and the M > 1 kernels of engine_batched.hip.h form on the
// fly.  The batch-1 GEMV - what BASELINE config 3 measures - still streams the one-byte codes; the skinny-GEMM paths
// (prompts of <= 128 tokens, decode steps of 17-64 sequences) are latency- and texture-addresser-bound, not byte-bound,
// so they take the kernels above unchanged (fp8 batch 64: 1.77 ms per step on the row-major kernels).
__global__ void pack_weights_fp8_kernel(const uint8_t* w, const bf16* scale, bf16* wp, int N, int K) {
    const size_t chunks = (size_t)N * K / 8, stride = (size_t)gridDim.x * blockDim.x;
    const int ksn = K / 32, KB = K >> 7;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < chunks; t += stride) {
        const int l = (int)(t & 63);
        const size_t blk = t >> 6;
        const int ks = (int)(blk % ksn), nt = (int)(blk / ksn);
        const int row = nt * 16 + (l & 15), k = ks * 32 + 8 * (l >> 4);
        const uint2 v = *reinterpret_cast<const uint2*>(w + (size_t)row * K + k);
        const float sc = to_f(scale[(size_t)(row >> 7) * KB + (k >> 7)]);
        const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), c1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
        const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), c3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
        *reinterpret_cast<uint4*>(wp + t * 8) = make_uint4(pack_bf16x2(c0.x * sc, c0.y * sc), pack_bf16x2(c1.x * sc, c1.y * sc),
                                                          pack_bf16x2(c2.x * sc, c2.y * sc), pack_bf16x2(c3.x * sc, c3.y * sc));
    }
}

pgk_status pack_weights_fp8(const void* w, const void* scale, void* wp, int N, int K, hipStream_t st) {
    PGK_REQUIRE(N % 16 == 0 && K % 128 == 0 && scale != nullptr, "pack_weights_fp8: N=%d must be a multiple of 16, K=%d of 128, scales present", N, K);
    const size_t chunks = (size_t)N * K / 8;
    const int grid = (int)(chunks / 256 > 4096 ? 4096 : (chunks + 255) / 256);
    pack_weights_fp8_kernel<<<grid, 256, 0, st>>>((const uint8_t*)w, (const bf16*)scale, (bf16*)wp, N, K);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

#ifdef PGK_PHASE_STAMPS
// diagnostic build only (tools/pk_stamps.py): 100 MHz stamps of workgroup phases, last launch of each epilogue kind
__device__ unsigned long long g_pk_stamps[5][512][8];
#define PK_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_pk_stamps[EPI][blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PK_STAMP(i) do { } while (0)
#endif

// MT = m-tiles per workgroup: 2 (32-row m-blocks) for prompts, 1 (16 rows) for M <= 96 - decode batches, short prompts - where
// 32-row blocks would leave CUs without a workgroup.
template <int NTW, int EPI, int MT>
__global__ __launch_bounds__(PK_THREADS) void pkgemm_kernel(PkArgs g) {
    constexpr int PK_MT = MT, PK_MB = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) char a_lds[];   // [PK_MB][ksteps * 32] bf16, chunk-swizzled rows
    __shared__ float ssred[4][PK_MB];
    PK_STAMP(0);
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    // workgroup id -> (n-block, m-block, split).  Ids are dealt round-robin to the 8 XCDs; n-blocks go in groups of 8, one per
    // XCD, and the m-blocks (and K-splits) of an n-block sit 8 ids apart: they land on the same XCD and share the weight
    // bytes in its L2.  The grid is padded to whole groups; the padding workgroups leave at once.
    int cb, mb, sp;
    {
        const int per = 8 * g.mblk * g.splits;
        const int grp = blockIdx.x / per, r = blockIdx.x - grp * per;
        cb = grp * 8 + (r & 7);
        int rest = r >> 3;
        mb = rest % g.mblk;
        sp = rest / g.mblk;
        if (cb >= g.nblk) return;
    }
    const int m0 = mb * PK_MB;
    const int ks0 = sp * g.ksteps, KS = g.ksteps, ksn = g.K / 32;
    const int rb = KS * 64;                                   // bytes per LDS row
    const int tile_a = cb * g.tiles_per_cb + wid;
    const bf16* wrun[NTW];
    wrun[0] = g.wp + ((size_t)tile_a * ksn + ks0) * 512 + lane * 8;
    if constexpr (NTW == 2) wrun[1] = g.wp + ((size_t)(tile_a + g.tile_b_off) * ksn + ks0) * 512 + lane * 8;

    // head epilogue: the m-block's RoPE rows (32 positions x 64 floats, cos and sin) go to LDS by 16 DMA instructions and
    // the gammas are requested first of all (first in = first out).  (Per-lane loads of the 16 entries a lane needs were
    // 64 more wave-instructions through the CU's texture addresser, which is what bounds this kernel's first 2.8 us.)
    __shared__ __attribute__((aligned(16))) float rope_lds[EPI == PK_EPI_QKV ? 2 * PK_MB * 64 : 4];
    uint16_t gbits[2] = {0, 0};
    bool is_q = false, is_k = false, has_gamma = false;
    if constexpr (EPI == PK_EPI_QKV) {
        is_q = cb < g.hq;
        is_k = !is_q && cb < g.hq + g.hkv;
        const int d = 16 * wid + l15;                          // < 64; the wave's second tile holds d + 64
        // unconditional loads from an always-valid address, converted where they are used: under a branch the compiler
        // finishes the conversion before the join, i.e. waits for the loads right here, ahead of every DMA of the kernel
        const bf16* gm = is_q ? g.q_gamma : g.k_gamma;
        has_gamma = gm != nullptr && (is_q || is_k);
        const bf16* gsafe = has_gamma ? gm : reinterpret_cast<const bf16*>(g.rope_cos);
        gbits[0] = gsafe[d].bits; gbits[1] = gsafe[d + 64].bits;
        if (is_q || is_k) {
            const uint32_t r0 = pk_lds_addr(reinterpret_cast<const char*>(rope_lds));
            constexpr int IPT = PK_MB / 4;                    // instructions per table: 4 rows each
            for (int j = wid; j < 2 * IPT; j += 4) {           // instruction j: table j / IPT, rows 4 (j % IPT) + q, 16 bytes per lane
                const int row = 4 * (j % IPT) + q;
                const int pos = min(g.start_pos + min(m0 + row, g.M - 1), g.max_seq - 1);
                pk_dma16(((j / IPT) ? g.rope_sin : g.rope_cos) + (size_t)pos * 64 + 4 * l15, r0 + j * 1024);
            }
        }
    }

    // rows that arrive un-normalised (pkgemm_resid_nt): 1 / rms of this m-block's rows from the producer's partial sums,
    // 8 threads per row; visible to everyone after the barrier that follows the DMA wait
    __shared__ float inv_lds[PK_MB];
    const bool scaled = g.ss_in != nullptr;
    // The row statistics are requested here and reduced only after this wave's DMAs and weight ring are issued (reducing on
    // the spot held those back by an L2 round trip).  Unconditional load, valid dummy address when there is nothing to scale:
    // under a branch the compiler completes the loaded values before the join - the same wait.
    float4 ssu, ssv;
    {
        const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
        const float* sp = scaled ? g.ss_in + (size_t)min(m0 + min(row, PK_MB - 1), g.M - 1) * PK_SS_LD + 8 * sub   // a row's PK_SS_LD floats: 8 per thread
                                 : reinterpret_cast<const float*>(g.wp) + 8 * sub;
        ssu = *reinterpret_cast<const float4*>(sp);
        ssv = *reinterpret_cast<const float4*>(sp + 4);
    }
    // activation block by LDS-DMA: instruction j fills LDS bytes [1024 j, +1024) = four 256-byte swizzle groups; lane i sits
    // in group 4 j + (i >> 4) at chunk position i & 15.  row = group / groups-per-row through a 16-bit reciprocal (exact for
    // the <= 2048 groups of a block, no integer division in the issue path).
    {
        const uint32_t lds0 = pk_lds_addr(a_lds);
        const int ndma = PK_MB * rb / 1024, gpr = KS >> 2;
        const uint32_t magic = (65536u + gpr - 1) / gpr;
        const bf16* abase = g.a + (size_t)ks0 * 32;
        for (int j = wid; j < ndma; j += 4) {
            const int grp = 4 * j + q;
            const int row = (int)(((uint32_t)grp * magic) >> 16), gi = grp - row * gpr;
            const int c = gi * 16 + ((l15 ^ row) & 15);
            pk_dma16(abase + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8, lds0 + j * 1024);
        }
    }
    // weight ring: exactly NTW * PK_RING loads behind the DMAs (clamped k-step: the explicit wait below counts them)
    uint4 wr[NTW][PK_RING];
#pragma unroll
    for (int s = 0; s < PK_RING; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t) wr[t][s] = *reinterpret_cast<const uint4*>(wrun[t] + (size_t)min(s, KS - 1) * 512);   // default cache policy: the other m-blocks re-read these bytes from L2
    if (scaled) {
        const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
        const float p8[8] = {ssu.x, ssu.y, ssu.z, ssu.w, ssv.x, ssv.y, ssv.z, ssv.w};
        float ssum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ssum += (8 * sub + i < g.ss_n) ? p8[i] : 0.f;
        ssum = group8_sum(ssum);
        if (sub == 0 && row < PK_MB) inv_lds[row] = 1.0f / sqrtf(ssum / (float)g.K + g.ss_eps);
    }
    PK_STAMP(1);
    if constexpr (NTW == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // in-order return: every DMA has landed
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    PK_STAMP(2);

    f32x4_p acc[NTW][PK_MT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt) acc[t][mt] = f32x4_p{0.f, 0.f, 0.f, 0.f};

    // A fragment of k-step ks: row mt * 16 + l15, chunk 4 ks + q at its swizzled position.  With ks = 4 a + j the byte
    // offset is rowbase + 256 a + 16 ((4 j + q) ^ (row & 15)): the four XOR terms are per-lane constants
    uint4 af[4][PK_MT];
    int aoff[PK_MT][4];
#pragma unroll
    for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) aoff[mt][j] = (mt * 16 + l15) * rb + ((((4 * j + q) ^ l15) & 15) << 4);
    auto read_a = [&](int ks, int j, uint4 (&dst)[PK_MT]) {   // j == ks & 3 (compile-time at every call site)
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt) dst[mt] = *reinterpret_cast<const uint4*>(a_lds + aoff[mt][j] + ((ks >> 2) << 8));
    };
    auto step = [&](int s, const uint4 (&a_cur)[PK_MT]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const bf16x8_p b = __builtin_bit_cast(bf16x8_p, wr[t][s]);
#pragma unroll
            for (int mt = 0; mt < PK_MT; ++mt)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_p, a_cur[mt]), b, acc[t][mt], 0, 0, 0);
        }
    };
    // fragments are read TWO k-steps ahead of the MFMAs that use them: with one wave per SIMD nothing else hides the LDS
    // latency, and one step (4 MFMAs = 64 cycles) does not cover it (in-kernel stamps: 16 k-steps took 1.15 us against
    // 0.43 us of MFMA time with a one-step lead)
    auto ahead = [&](int ks, int s2) {   // fragment of k-step ks + 2 into slot (s2 + 2) & 3; s2 = ks mod 16 at every call site
        const int j = (s2 + 2) & 3;
        read_a(min(ks + 2, KS - 4 + j), j, af[(s2 + 2) & 3]);
    };
    read_a(0, 0, af[0]);
    read_a(1, 1, af[1]);
    const int nfull = KS / PK_RING, rem = KS % PK_RING;
    for (int blk = 0; blk < nfull; ++blk) {
#pragma unroll
        for (int s = 0; s < PK_RING; ++s) {
            const int ks = blk * PK_RING + s;
            ahead(ks, s);
            step(s, af[s & 3]);
            if (ks + PK_RING < KS) {                               // wave-uniform
#pragma unroll
                for (int t = 0; t < NTW; ++t) wr[t][s] = *reinterpret_cast<const uint4*>(wrun[t] + (size_t)(ks + PK_RING) * 512);
            }
        }
        if (blk == 0) PK_STAMP(3);
    }
#pragma unroll
    for (int s = 0; s < PK_RING; ++s) {
        if (s < rem) {
            const int ks = nfull * PK_RING + s;
            ahead(ks, s);
            step(s, af[s & 3]);
        }
    }

    PK_STAMP(4);
    if (scaled) {
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float inv = inv_lds[mt * 16 + 4 * q + r];
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t][mt][r] *= inv;
            }
    }
    // C/D map: column (weight row) = l15 of the tile, row m = mt * 16 + 4 q + r
    if constexpr (EPI == PK_EPI_BF16 || EPI == PK_EPI_SLAB || EPI == PK_EPI_ACCUM) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int n = (tile_a + t * g.tile_b_off) * 16 + l15;
            if constexpr (EPI == PK_EPI_ACCUM) {
                float* c = reinterpret_cast<float*>(g.c);
                float old[PK_MT][4];
#pragma unroll
                for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[mt][r] = c[(size_t)min(m0 + mt * 16 + 4 * q + r, g.M - 1) * g.ldc + n];
#pragma unroll
                for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + mt * 16 + 4 * q + r;
                        if (m < g.M) c[(size_t)m * g.ldc + n] = old[mt][r] + acc[t][mt][r];
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + mt * 16 + 4 * q + r;
                        if (m < g.M) {
                            if constexpr (EPI == PK_EPI_BF16) reinterpret_cast<bf16*>(g.c)[(size_t)m * g.ldc + n] = from_f<bf16>(acc[t][mt][r]);
                            else (reinterpret_cast<float*>(g.c) + (size_t)sp * g.M * g.ldc)[(size_t)m * g.ldc + n] = acc[t][mt][r];
                        }
                    }
            }
        }
    } else if constexpr (EPI == PK_EPI_SWIGLU) {
        // act = silu(gate) * up on the bf16-rounded projections (what a bf16 gate_up store followed by the activation
        // kernel computes), one act column per lane
        const int n = tile_a * 16 + l15;
        bf16* c = reinterpret_cast<bf16*>(g.c);
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mt * 16 + 4 * q + r;
                const float gf = to_f(from_f<bf16>(acc[0][mt][r])), uf = to_f(from_f<bf16>(acc[1][mt][r]));
                if (m < g.M) c[(size_t)m * g.ldc + n] = from_f<bf16>(gf / (1.0f + __expf(-gf)) * uf);
            }
    } else {
        // one head of 128: lane holds dims d = 16 wid + l15 (tile 0) and d + 64 (tile 1) of rows m
        const int d = 16 * wid + l15;
        float x0[PK_MT][4], x1[PK_MT][4];
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[mt][r] = to_f(from_f<bf16>(acc[0][mt][r]));
                x1[mt][r] = to_f(from_f<bf16>(acc[1][mt][r]));
            }
        if (is_q || is_k) {
            const bool has_norm = (is_q ? g.q_gamma : g.k_gamma) != nullptr;
            if (has_norm) {
#pragma unroll
                for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float ss = fmaf(x0[mt][r], x0[mt][r], x1[mt][r] * x1[mt][r]);
                        ss = group16_sum(ss);
                        if (l15 == 0) ssred[wid][mt * 16 + 4 * q + r] = ss;
                    }
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mi = mt * 16 + 4 * q + r;
                        const float ss = (ssred[0][mi] + ssred[1][mi]) + (ssred[2][mi] + ssred[3][mi]);
                        const float inv = 1.0f / sqrtf(ss / 128.f + g.eps);
                        x0[mt][r] = x0[mt][r] * inv * bf16_bits_to_f(gbits[0]);
                        x1[mt][r] = x1[mt][r] * inv * bf16_bits_to_f(gbits[1]);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = mt * 16 + 4 * q + r;
                    const float cs = rope_lds[mi * 64 + d], sn = rope_lds[PK_MB * 64 + mi * 64 + d];
                    const float a0 = x0[mt][r], a1 = x1[mt][r];
                    x0[mt][r] = a0 * cs - a1 * sn;
                    x1[mt][r] = a1 * cs + a0 * sn;
                }
        }
#pragma unroll
        for (int mt = 0; mt < PK_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mt * 16 + 4 * q + r;
                if (m >= g.M) continue;
                if (is_q) {
                    bf16* dst = reinterpret_cast<bf16*>(g.c) + (size_t)m * g.ldc + (size_t)cb * 128;
                    dst[d] = from_f<bf16>(x0[mt][r]);
                    dst[d + 64] = from_f<bf16>(x1[mt][r]);
                } else {
                    const int pos = g.start_pos + m;
                    if (pos < g.max_seq) {
                        const int kvh = is_k ? cb - g.hq : cb - g.hq - g.hkv;
                        bf16* dst = (is_k ? g.kcache : g.vcache) + ((size_t)kvh * g.max_seq + pos) * 128;
                        dst[d] = from_f<bf16>(x0[mt][r]);
                        dst[d + 64] = from_f<bf16>(x1[mt][r]);
                    }
                }
            }
    }
    PK_STAMP(5);
}

#ifdef PGK_PHASE_STAMPS
extern "C" int pgk_debug_pk_stamps(unsigned long long* out) {   // [5][512][8]
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_stamps), sizeof(g_pk_stamps));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// N = hidden projections WITHOUT a K split over workgroups: h += A . W^T with the next RMSNorm's inputs as by-products.
// The split-K form above needs a reducer launch (the RMSNorm kernel sums the slabs): ~5 us + a dependent-launch gap, twice
// per layer.  Here a workgroup owns NTW n-tiles x MB rows over the WHOLE K: its four waves take a K quarter each (a wave
// reads only its quarter of the activation block from LDS), meet in LDS, and the workgroup adds the residual, stores
// h_new, bf16(h_new * gamma_next) and the sum of squares of its columns per row.  The consumer (QKV / gate_up above)
// multiplies with the un-normalised rows and scales its result rows by 1 / rms.  Costs more texture-addresser time per
// workgroup than the split form (the activation block is MB x K instead of MB x K / splits) and saves the launch.
struct PkResidArgs {
    const bf16* a;
    int lda;
    const bf16* wp;
    int M, N, K;
    int nblk, mblk;
    float* h;
    const bf16* gamma_next;
    bf16* xpre;
    float* ss_out;
};

template <int MB, int NTW>
__global__ __launch_bounds__(PK_THREADS) void pkgemm_resid_kernel(PkResidArgs g) {
    constexpr int MT = MB / 16, NC = 16 * NTW;
    constexpr int EPT = MB * NC / PK_THREADS;                       // output elements per thread (1 or 2)
    static_assert(MB * NC == 512 || MB * NC == 256, "one workgroup owns 32 x 16, 16 x 32 or 16 x 16 outputs");
    extern __shared__ __attribute__((aligned(16))) char a_lds[];   // [MB][K] bf16, chunk-swizzled rows
    __shared__ __attribute__((aligned(16))) float4 kpart[4 * NTW * MT * 64];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    int cb, mb;
    {
        const int per = 8 * g.mblk;
        const int grp = blockIdx.x / per, r = blockIdx.x - grp * per;
        cb = grp * 8 + (r & 7);
        mb = r >> 3;
        if (cb >= g.nblk) return;
    }
    const int m0 = mb * MB, n0 = cb * NC;
    const int KS = g.K / 32, KSW = KS / 4, ksw0 = wid * KSW;       // this wave's k-steps
    const int rb = KS * 64;
    // the residual values of this thread's outputs first: they are needed last, and first in = first out
    float hold[EPT];
    int em[EPT], en[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + PK_THREADS * i;
        em[i] = e / NC; en[i] = e % NC;
        hold[i] = g.h[(size_t)min(m0 + em[i], g.M - 1) * g.N + n0 + en[i]];
    }
    // (unconditional loads, converted where they are used: under a branch the compiler waits for them right here - two
    // serialised L2 round trips in front of every DMA of the kernel in the first version)
    uint16_t gnb[EPT];
    {
        const bf16* gsafe = g.gamma_next ? g.gamma_next + n0 : g.a;
#pragma unroll
        for (int i = 0; i < EPT; ++i) gnb[i] = gsafe[en[i]].bits;
    }
    {
        const uint32_t lds0 = pk_lds_addr(a_lds);
        const int ndma = MB * rb / 1024, gpr = KS >> 2;
        const uint32_t magic = (65536u + gpr - 1) / gpr;
        for (int j = wid; j < ndma; j += 4) {
            const int grp = 4 * j + q;
            const int row = (int)(((uint32_t)grp * magic) >> 16), gi = grp - row * gpr;
            const int c = gi * 16 + ((l15 ^ row) & 15);
            pk_dma16(g.a + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8, lds0 + j * 1024);
        }
    }
    const int ksn = KS;
    const bf16* wrun[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) wrun[t] = g.wp + ((size_t)(cb * NTW + t) * ksn + ksw0) * 512 + lane * 8;
    uint4 wr[NTW][PK_RING];
#pragma unroll
    for (int s = 0; s < PK_RING; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t) wr[t][s] = *reinterpret_cast<const uint4*>(wrun[t] + (size_t)min(s, KSW - 1) * 512);
    if constexpr (NTW == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();

    f32x4_p acc[NTW][MT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4_p{0.f, 0.f, 0.f, 0.f};
    uint4 af[4][MT];
    int aoff[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) aoff[mt][j] = (mt * 16 + l15) * rb + ((((4 * j + q) ^ l15) & 15) << 4);
    auto read_a = [&](int ks, int j, uint4 (&dst)[MT]) {       // ks: k-step of the whole block; j == ks & 3
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dst[mt] = *reinterpret_cast<const uint4*>(a_lds + aoff[mt][j] + ((ks >> 2) << 8));
    };
    auto step = [&](int s, const uint4 (&a_cur)[MT]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const bf16x8_p b = __builtin_bit_cast(bf16x8_p, wr[t][s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_p, a_cur[mt]), b, acc[t][mt], 0, 0, 0);
        }
    };
    auto ahead = [&](int ks, int s2) {                          // local k-step ks, s2 = ks mod 16; KSW and ksw0 are multiples of 4
        const int j = (s2 + 2) & 3;
        read_a(ksw0 + min(ks + 2, KSW - 4 + j), j, af[(s2 + 2) & 3]);
    };
    read_a(ksw0, 0, af[0]);
    read_a(ksw0 + 1, 1, af[1]);
    const int nfull = KSW / PK_RING, rem = KSW % PK_RING;
    for (int blk = 0; blk < nfull; ++blk) {
#pragma unroll
        for (int s = 0; s < PK_RING; ++s) {
            const int ks = blk * PK_RING + s;
            ahead(ks, s);
            step(s, af[s & 3]);
            if (ks + PK_RING < KSW) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) wr[t][s] = *reinterpret_cast<const uint4*>(wrun[t] + (size_t)(ks + PK_RING) * 512);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < PK_RING; ++s) {
        if (s < rem) {
            const int ks = nfull * PK_RING + s;
            ahead(ks, s);
            step(s, af[s & 3]);
        }
    }
    // the four K quarters meet: [wave][tile][m-tile][lane] float4, then thread e sums its element's four partials
    {
        float4* mine = kpart + (size_t)wid * NTW * MT * 64 + lane;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mine[(t * MT + mt) * 64] = make_float4(acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]);
    }
    __syncthreads();
    const float* kp = reinterpret_cast<const float*>(kpart);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int m = em[i], n = en[i];
        // C/D map: tile n >> 4, column lane & 15 = n & 15, row = 4 (lane >> 4) + reg -> lane = 16 ((m & 15) >> 2) + (n & 15), reg = m & 3
        const int idx = ((((n >> 4) * MT + (m >> 4)) * 64) + 16 * ((m & 15) >> 2) + (n & 15)) * 4 + (m & 3);
        float v = hold[i];
        v += (kp[idx] + kp[idx + NTW * MT * 256]) + (kp[idx + 2 * NTW * MT * 256] + kp[idx + 3 * NTW * MT * 256]);
        const bool ok = m0 + m < g.M;
        if (ok) g.h[(size_t)(m0 + m) * g.N + n0 + n] = v;
        if (g.gamma_next) {
            if (ok) g.xpre[(size_t)(m0 + m) * g.N + n0 + n] = from_f<bf16>(v * bf16_bits_to_f(gnb[i]));
            const float sq = group_sum<NC>(v * v);                 // the NC threads of a row are consecutive lanes
            if (n == 0 && ok) g.ss_out[(size_t)(m0 + m) * PK_SS_LD + cb] = sq;
        }
    }
}

bool pkgemm_resid_ok(int N, int K) { return N % 32 == 0 && N / 16 <= PK_SS_LD && K % 512 == 0 && 16 * (K / 32) * 64 <= 128 * 1024; }

pgk_status pkgemm_resid_nt(const bf16* a, int lda, const void* wp, float* h, int M, int N, int K, const bf16* gamma_next, bf16* xpre,
                           float* ss_out, int* ss_n, hipStream_t st) {
    PGK_REQUIRE(M >= 1 && M <= 128 && pkgemm_resid_ok(N, K) && lda % 8 == 0, "pkgemm_resid: M=%d N=%d K=%d lda=%d not supported", M, N, K, lda);
    PGK_REQUIRE(!gamma_next || (xpre && ss_out && ss_n), "pkgemm_resid: the next norm's outputs are missing");
    PkResidArgs g{a, lda, (const bf16*)wp, M, N, K, 0, 0, h, gamma_next, xpre, ss_out};
    // 16 rows x one n-tile while that still leaves at most one workgroup per CU (fewest bytes per workgroup); otherwise
    // 32 rows x one n-tile while the 32-row activation block fits the LDS budget, else 16 rows x two n-tiles
    const bool small = (N / 16) * ceil_div(M, 16) <= 256;
    const bool wide = !small && (size_t)32 * (K / 32) * 64 <= 128 * 1024;
    const int mb = wide ? 32 : 16, ntw = (small || wide) ? 1 : 2;
    g.nblk = N / (16 * ntw);
    g.mblk = ceil_div(M, mb);
    if (ss_n) *ss_n = g.nblk;
    const size_t lds = (size_t)mb * (K / 32) * 64;
    const int grid = ceil_div(g.nblk, 8) * 8 * g.mblk;
#define PGK_PKR_LAUNCH(MBV, NTWV)                                                                                  \
    {                                                                                                              \
        static bool attr = false;                                                                                  \
        if (lds > 48 * 1024 && !attr) {                                                                            \
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pkgemm_resid_kernel<MBV, NTWV>),       \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(128 * 1024)));     \
            attr = true;                                                                                           \
        }                                                                                                          \
        pkgemm_resid_kernel<MBV, NTWV><<<grid, PK_THREADS, lds, st>>>(g);                                          \
    }
    if (small) PGK_PKR_LAUNCH(16, 1) else if (wide) PGK_PKR_LAUNCH(32, 1) else PGK_PKR_LAUNCH(16, 2)
#undef PGK_PKR_LAUNCH
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

// K splits for the N = hidden projections: enough workgroups to cover the chip, and a K range per workgroup whose
// activation block (32 rows) fits the LDS budget.  0: no split count works for this K.
int pkgemm_pick_splits(int M, int N, int K) {
    const int nblk = N / 64, mblk = ceil_div(M, M <= 96 ? 16 : 32);
    int best = 0;
    for (int s = 1; s <= 16; ++s) {
        if (K % (s * 128) != 0 || K / s > 2048) continue;
        best = s;
        if (nblk * mblk * s >= 256) break;
    }
    return best;
}

// splittable: the projection may be cut along K (slab epilogue); otherwise one workgroup walks the whole K
bool pkgemm_shape_ok(int N, int K, bool splittable) {
    return N % 64 == 0 && K % 128 == 0 && (splittable ? pkgemm_pick_splits(128, N, K) > 0 : K <= 2048);
}

// Packed-weight projection.  epi: PK_EPI_*; `splits` > 1 only with PK_EPI_SLAB.  SwiGLU: N = 2 * I weight rows, c = act [M][I].
pgk_status pkgemm_nt(const bf16* a, int lda, const void* wp, void* c, int ldc, int epi, int splits, int M, int N, int K, const PkArgs* head,
                     hipStream_t st) {
    PGK_REQUIRE(M >= 1 && M <= 128, "pkgemm: M=%d outside [1,128]", M);
    PGK_REQUIRE(N % 64 == 0 && K % 128 == 0 && lda % 8 == 0, "pkgemm: N=%d K=%d lda=%d not supported", N, K, lda);
    PGK_REQUIRE(splits >= 1 && K % (splits * 128) == 0 && (epi == PK_EPI_SLAB || splits == 1), "pkgemm: %d K-splits not usable here", splits);
    PkArgs g = head ? *head : PkArgs{};
    g.a = a; g.lda = lda; g.wp = (const bf16*)wp; g.M = M; g.N = N; g.K = K; g.c = c; g.ldc = ldc;
    g.ksteps = K / splits / 32;
    g.splits = splits;
    // (m-tile count chosen below, once the n-block count of the epilogue is known)
    g.tiles_per_cb = 4;
    g.tile_b_off = 0;
    if (epi == PK_EPI_SWIGLU) {
        PGK_REQUIRE(N % 128 == 0, "pkgemm: SwiGLU needs 2 I = %d weight rows with I a multiple of 64", N);
        g.nblk = N / 2 / 64;
        g.tile_b_off = N / 2 / 16;
    } else if (epi == PK_EPI_QKV) {
        PGK_REQUIRE(head && N == (g.hq + 2 * g.hkv) * 128, "pkgemm: head epilogue needs head_dim 128 (N=%d)", N);
        g.nblk = N / 128;
        g.tiles_per_cb = 8;
        g.tile_b_off = 4;
    } else {
        g.nblk = N / 64;
    }
    // 16-row m-blocks (twice the workgroups, fewer bytes each) while they still fit one round at one workgroup per CU, and
    // always up to 96 rows; else 32-row blocks, which re-read the weights half as often (measured at 128 rows: QKV 256 x 16-row
    // workgroups instead of 128 x 32-row ones pays, gate_up with 384 does not)
    const int mt = (M <= 96 || g.nblk * splits * ceil_div(M, 16) <= 256) ? 1 : 2;
    g.mblk = ceil_div(M, 16 * mt);
    const size_t lds = (size_t)16 * mt * g.ksteps * 64;
    PGK_REQUIRE(lds <= 128 * 1024, "pkgemm: K per workgroup %d too long for the LDS activation block", g.ksteps * 32);
    const int grid = ceil_div(g.nblk, 8) * 8 * g.mblk * splits;     // whole groups of 8 n-blocks (see the id mapping in the kernel)
#define PGK_PK_LAUNCH2(NTWV, EPIV, MTV)                                                                            \
    {                                                                                                              \
        static size_t attr = 0;                                                                                    \
        if (lds > 48 * 1024 && lds > attr) {                                                                       \
            PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pkgemm_kernel<NTWV, EPIV, MTV>),       \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(128 * 1024)));     \
            attr = 128 * 1024;                                                                                     \
        }                                                                                                          \
        pkgemm_kernel<NTWV, EPIV, MTV><<<grid, PK_THREADS, lds, st>>>(g);                                          \
    }
#define PGK_PK_LAUNCH(NTWV, EPIV)                                                                                  \
    if (mt == 1) PGK_PK_LAUNCH2(NTWV, EPIV, 1) else PGK_PK_LAUNCH2(NTWV, EPIV, 2)
    switch (epi) {
        case PK_EPI_BF16: PGK_PK_LAUNCH(1, PK_EPI_BF16) break;
        case PK_EPI_SLAB: PGK_PK_LAUNCH(1, PK_EPI_SLAB) break;
        case PK_EPI_ACCUM: PGK_PK_LAUNCH(1, PK_EPI_ACCUM) break;
        case PK_EPI_SWIGLU: PGK_PK_LAUNCH(2, PK_EPI_SWIGLU) break;
        case PK_EPI_QKV: PGK_PK_LAUNCH(2, PK_EPI_QKV) break;
        default: return set_error(PGK_ERR_INVALID, "pkgemm: unknown epilogue %d", epi);
    }
#undef PGK_PK_LAUNCH2
#undef PGK_PK_LAUNCH
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk
