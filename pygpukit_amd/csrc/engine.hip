// Native decode / prefill engine for Qwen3 / Llama-shaped causal transformers on gfx950.
//
// A decode step is a short chain of fused weight-streaming kernels, enqueued by ONE C call and captured
// into ONE hipGraph whose per-sequence token id and position live in device memory:
//
//   per layer (fused path, contexts <= 512):
//     norm_qkv          qkv[b]      = Wqkv . rmsnorm(h[b])
//     attn_oproj        part[b][kv] = Wo[:, heads of kv] . attention(q,k,v of kv head)   (QK-norm, RoPE, KV write inside)
//     norm_gateup       h2[b] = h[b] + sum_kv part[b][kv] ;  act[b] = silu(Wg x) * (Wu x),  x = rmsnorm(h2[b])
//     down_residual     h[b]  = h2[b] + Wd . act[b]
//   per layer (split path, long contexts): norm_qkv, split-KV attn, oproj_residual, norm_gateup, down_residual
//   norm_lmhead         logits[b]   = E . rmsnorm(h[b])  (+ per-workgroup argmax partials)
//   finalize            token[b] = argmax (lowest index on ties); position[b] += 1; log; h[b] = E[token[b]]
//
// i.e. 4L+2 launches per token versus ~21 launches + 21 device syncs per layer in the reference's eager
// step and 2L+2 graphs in its "graph" step (src/pygpukit/llm/decode/m1.py:40-121,
// m1_graph.py:463-589).  On this chip a dependent kernel boundary (~1.7 us in a graph) is the cheapest
// chip-wide synchronisation there is, so the design minimises the NUMBER of boundaries and, inside each
// kernel, issues the weight loads BEFORE the activation prologue so the two memory round trips overlap.
// The residual stream, q/k/v and the MLP activation stay fp32 between kernels; only the KV cache (bf16)
// and the weights are rounded.  KV cache layout: [layer][seq][Hkv][max_seq][D] - un-expanded GQA.
//
// Prefill runs the MFMA GEMM / flash-attention kernels on bf16 activations with an fp32 residual stream.

#include <cstdlib>
#include <type_traits>
#include <vector>

#include "attn_core.hip.h"
#include "engine_common.hip.h"
#include "gemm_epilogues.hip.h"
#include "pkgemm.hip.h"

namespace pgk {

constexpr int SHORT_CTX = 512;   // contexts up to here take the whole-context attention kernels (direct batch attention); one sequence: SHORT_CTX_B1
constexpr int SHORT_CTX_B1 = 384; // a single sequence's fused attention + o_proj kernel walks the context in chunks of 192 rows (AM_CHUNK) in EVERY one of its
                                  // 256 workgroups: two chunks still beat the split-KV sequence (context 300: 0.616 vs 0.625 ms per step), three do not (400: 0.678 vs 0.627)

pgk_status engine_gemm_nt(const bf16* A, const void* W, const bf16* wscale, bool fp8, void* C, bool accum_f32, int M,
                          int N, int K, hipStream_t st, bool packed = false);      // packed: W = the fragment-major bf16 copy
bool engine_gemm_packed_ok(int M, int N, int K);
int wsgemm_pick_splits(int N, int K, bool allow_split);
int engine_gemm_pick_splits(int M, int N, int K);
pgk_status engine_gemm_nt_slabs(const bf16* A, const void* W, float* slabs, int splits, int M, int N, int K, hipStream_t st, bool packed = false);
pgk_status gemm_fp8_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, void* c, bool accum_f32, int M,
                       int N, int K, hipStream_t st);
pgk_status quantize_fp8_rows_bf16(const bf16* x, uint8_t* out, float* scale, int M, int K, hipStream_t st);
bool sdpa_flash_enabled();                                     // ops_attention.hip: PYGPUKIT_FLASH_ATTENTION
pgk_status flash_prefill_q8(const void* q, const void* k, const void* v, uint8_t* q8, float* q8s, int hq, int hkv, int q_len, int kv_len,
                            float scale, long long qh, long long qs, long long kh, long long ks, hipStream_t st);   // ops_flash.hip
bool engine_gemm_qkv_heads_ok(int M, int N, int K);            // ops_gemm.hip: QKV projection with per-head norm + RoPE + cache write as its epilogue
pgk_status engine_gemm_qkv_heads_nt(const bf16* A, const bf16* W, bf16* qkv, int M, int N, int K, const QkvHeadArgs& hd, hipStream_t st, bool packed = false);
bool engine_gemm_swiglu_ok(int M, int I, int K, bool fp8);     // ops_gemm.hip: gate / up projection with the SwiGLU epilogue
pgk_status engine_gemm_swiglu_nt(const bf16* A, const void* W, const bf16* wscale, bool fp8, bf16* act, int M, int I, int K, hipStream_t st, bool packed = false);
bool gemm_fp8_qkv_heads_ok(int M, int N, int K);               // ops_fp8_gemm.hip: the same epilogue on the fp8 x fp8 256-tile kernel
pgk_status gemm_fp8_qkv_heads_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, bf16* qkv, int M, int N, int K,
                                 const QkvHeadArgs& hd, hipStream_t st);
bool gemm_fp8_swiglu_ok(int M, int I, int K);                  // ops_fp8_gemm.hip: ... and the e4m3 quantisation of the result
pgk_status gemm_fp8_swiglu_nt(const uint8_t* a, const float* sa, const uint8_t* w, const bf16* sw, uint8_t* q_out, float* s_out, int M, int I,
                              int K, hipStream_t st);
pgk_status wsgemm_nt(const bf16* a, int lda, const void* w, const bf16* wscale, bool fp8, void* c, const bf16* bias, int mode,
                     int splits, int M, int N, int K, hipStream_t st);

// --------------------------------------------------------------------------------------------
// Fused GEMV kernel: prologue builds x[M][K] in LDS, body streams W, epilogue consumes y.
// --------------------------------------------------------------------------------------------
template <class XT> __device__ __forceinline__ void store_x(XT* xs, int i, float v);
template <> __device__ __forceinline__ void store_x<float>(float* xs, int i, float v) { xs[i] = v; }
template <> __device__ __forceinline__ void store_x<bf16>(bf16* xs, int i, float v) { xs[i] = from_f<bf16>(v); }

// C = number of 16-byte chunks per weight row held per lane.  C > 0 fixes K = C * 64 * NW at COMPILE
// time: the whole row set of the wave's first trip is preloaded before the prologue touches the
// activations, and every prologue loop has an exact trip count - straight-line code, no guarded loads.
// (A load under a per-lane guard, or accumulated inside a conditional, is waited for on the spot by
// hipcc: that serialised dozens of memory round trips per kernel in the first version.)  C == 0 is the
// generic any-K path.
template <class WT, class XT, int M, int R, int PRO, int EPI, int C>
__global__ __launch_bounds__(256) void fused_gemv_kernel(unsigned long long* tl, const void* w_, const bf16* wscale_, const float* x_, const bf16* gamma_,
                                                         const float* aux_, int N_, int naux_, FusedArgs a) {
    // The first 14 dwords of the kernel arguments - everything the load-issue phase needs - arrive PRELOADED in SGPRs
    // (-mllvm -amdgpu-kernarg-preload-count=14, see the Makefile): x_ = the fp32 input rows (FusedArgs::h for the norm
    // prologues, FusedArgs::xin for PRO_PLAIN), aux_ / naux_ = the o_proj partial vectors and their count (PRO_NORM_SUM) or the residual rows and
    // their leading dimension (EPI_RESID).  What is left in the by-value struct (eps, out, ld_out, h_out, argmax slots) is
    // fetched by scalar loads that complete under the weight stream.  Before, every load of the kernel waited for the
    // struct's s_load through a scalar cache the dispatch had just invalidated.
    static_assert(!(PRO == PRO_NORM_SUM && EPI == EPI_RESID), "aux_ cannot carry partial vectors and residual rows at once");
    const TLStamp tls(tl);
    constexpr int NW = WTraits<WT>::NW;
    constexpr bool FP8 = std::is_same<WT, fp8e4m3>::value;
    constexpr int KC = C * 64 * NW;           // compile-time K (0 = runtime)
    constexpr int KJ = KC / 256;              // activation elements per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    XT* xs = reinterpret_cast<XT*>(smem);  // [M][K]
    __shared__ float red[16];
    __shared__ float s_bv[4][M];
    __shared__ int s_bi[4][M];
    const int K = (C > 0) ? KC : a.K;
    const int N = N_;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int OUT_PER_TRIP = (EPI == EPI_SWIGLU) ? R / 2 : R;
    const int wave = blockIdx.x * 4 + wid, nwaves = gridDim.x * 4;

    auto row_of = [&](int n0, int r) -> int {
        if constexpr (EPI == EPI_SWIGLU) return (r < R / 2) ? min(n0 + r, N - 1) : N + min(n0 + r - R / 2, N - 1);
        else return min(n0 + r, N - 1);
    };

    uint4 pre[R][C > 0 ? C : 1];
    float psc[R][C > 0 ? C : 1];
    float resv[R][M];
    if constexpr (C > 0) {
        // ---- all global loads of the first trip, issued back to back; nothing is waited for until the prologue's ALU ----
        // Vector memory returns in ISSUE order.  The activation vectors are a few KB that the previous kernel left in L2,
        // the weight rows come from HBM: issued first, the activations are usable ~1 us before the weights land and the
        // whole prologue (norm statistic, barrier, LDS image) runs under the weight latency.  (The first version issued
        // the weights first: the prologue then started only after the last weight chunk had arrived - in-kernel stamps,
        // tools/phase_stamps.py.)
        const int nf = min(wave * OUT_PER_TRIP, N - 1);  // waves beyond N recompute the last rows (never stored)
        auto load_weights = [&]() {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row_of(nf, r);
                const WT* wr = reinterpret_cast<const WT*>(w_) + (size_t)row * KC;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int k0 = lane * NW + c * 64 * NW;
                    pre[r][c] = load_nt16(wr + k0);
                    if constexpr (FP8) psc[r][c] = to_f(wscale_[(size_t)(row >> 7) * (KC >> 7) + (k0 >> 7)]);
                }
            }
        };
        constexpr bool NORM = PRO == PRO_NORM || PRO == PRO_NORM_SUM;
        constexpr int NP = (PRO == PRO_NORM_SUM) ? 8 : 1;
        const int np = (PRO == PRO_NORM_SUM) ? naux_ : 1;
        float hv[M][KJ], gv[NORM ? KJ : 1], pvs[PRO == PRO_NORM_SUM ? M : 1][PRO == PRO_NORM_SUM ? KJ : 1][NP];
        // A thread owns KJ / VW runs of VW consecutive elements (run v starts at element (256 v + thread) * VW): every load of
        // the prologue is one 16-byte (K % 1024 == 0) or 8-byte access per run - a quarter of the instructions of the
        // element-per-load form, and the texture addresser moves 1 KiB instead of 256 B per wave-instruction.  The gate/up
        // kernel reads nine such vectors (h + 8 o_proj partials) in every workgroup: more bytes through a CU's addresser
        // than its share of the weights.
        constexpr int VW = (KJ % 4 == 0) ? 4 : 2, NV = KJ / VW;
        auto run0 = [&](int v) -> int { return (256 * v + (int)threadIdx.x) * VW; };
        auto ldrun = [&](const float* base, int v, float* dst) {
            if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(base + run0(v)); dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w; }
            else { const float2 t = *reinterpret_cast<const float2*>(base + run0(v)); dst[0] = t.x; dst[1] = t.y; }
        };
        auto strun = [&](float* base, int v, const float* src) {
            if constexpr (VW == 4) *reinterpret_cast<float4*>(base + run0(v)) = make_float4(src[0], src[1], src[2], src[3]);
            else *reinterpret_cast<float2*>(base + run0(v)) = make_float2(src[0], src[1]);
        };
        auto stx = [&](int m, int v, const float* src) {        // the LDS image of row m
            if constexpr (std::is_same<XT, float>::value) strun(reinterpret_cast<float*>(xs) + (size_t)m * KC, v, src);
            else if constexpr (VW == 4) *reinterpret_cast<uint2*>(xs + (size_t)m * KC + run0(v)) = make_uint2(pack_bf16x2(src[0], src[1]), pack_bf16x2(src[2], src[3]));
            else *reinterpret_cast<uint32_t*>(xs + (size_t)m * KC + run0(v)) = pack_bf16x2(src[0], src[1]);
        };
        // ---- activation loads ----
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int m = 0; m < M; ++m) resv[r][m] = *(aux_ + (size_t)m * naux_ + min(nf + r, N - 1));
        }
        if constexpr (NORM) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if constexpr (VW == 4) {
                    const uint2 g = *reinterpret_cast<const uint2*>(gamma_ + run0(v));
                    gv[4 * v] = __uint_as_float(g.x << 16); gv[4 * v + 1] = __uint_as_float(g.x & 0xFFFF0000u);
                    gv[4 * v + 2] = __uint_as_float(g.y << 16); gv[4 * v + 3] = __uint_as_float(g.y & 0xFFFF0000u);
                } else {
                    const uint32_t g = *reinterpret_cast<const uint32_t*>(gamma_ + run0(v));
                    gv[2 * v] = __uint_as_float(g << 16); gv[2 * v + 1] = __uint_as_float(g & 0xFFFF0000u);
                }
#pragma unroll
                for (int m = 0; m < M; ++m) ldrun(x_ + (size_t)m * KC, v, &hv[m][VW * v]);
            }
            if constexpr (PRO == PRO_NORM_SUM) {
                // partial vectors: unconditional clamped loads, masked adds below (one round trip for up to 8)
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            float t[VW];
                            ldrun(aux_ + ((size_t)m * np + min(p, np - 1)) * KC, v, t);
#pragma unroll
                            for (int e = 0; e < VW; ++e) pvs[m][VW * v + e][p] = t[e];
                        }
            }
        } else if constexpr (PRO == PRO_PLAIN) {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int v = 0; v < NV; ++v) ldrun(x_ + (size_t)m * KC, v, &hv[m][VW * v]);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the compiler from hoisting the weight stream above the small loads
        load_weights();
        __builtin_amdgcn_sched_barrier(0);
        // ---- prologue ALU, exact trip counts ----
        if constexpr (NORM) {
            if constexpr (PRO == PRO_NORM_SUM) {
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int j = 0; j < KJ; ++j)
#pragma unroll
                        for (int p = 0; p < NP; ++p) hv[m][j] += (p < np) ? pvs[m][j][p] : 0.f;
                // more than NP partial vectors (models with more than 8 kv heads): the rest in a second trip.  (Until this loop
                // existed partials 8.. were silently dropped: batch-1 / batch-2 decode of a 16-kv-head model was wrong.)
                for (int p = NP; p < np; ++p)
#pragma unroll
                    for (int m = 0; m < M; ++m)
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            float t[VW];
                            ldrun(aux_ + ((size_t)m * np + p) * KC, v, t);
#pragma unroll
                            for (int e = 0; e < VW; ++e) hv[m][VW * v + e] += t[e];
                        }
            }
            float ss[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                ss[m] = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) ss[m] = fmaf(hv[m][j], hv[m][j], ss[m]);
                ss[m] = wave_sum(ss[m]);
            }
            if (lane == 0) {
#pragma unroll
                for (int m = 0; m < M; ++m) s_bv[wid][m] = ss[m];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float tot = s_bv[0][m] + s_bv[1][m] + s_bv[2][m] + s_bv[3][m];
                const float inv = 1.0f / sqrtf(tot / KC + a.eps);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if constexpr (PRO == PRO_NORM_SUM) { if (blockIdx.x == 0) strun(a.h_out + (size_t)m * KC, v, &hv[m][VW * v]); }
                    float t[VW];
#pragma unroll
                    for (int e = 0; e < VW; ++e) t[e] = hv[m][VW * v + e] * inv * gv[VW * v + e];
                    stx(m, v, t);
                }
            }
        } else if constexpr (PRO == PRO_PLAIN) {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int v = 0; v < NV; ++v) stx(m, v, &hv[m][VW * v]);
        }
    }
    if constexpr (C == 0) {
        // ---- generic prologue (any K) ----
        if constexpr (PRO == PRO_NORM || PRO == PRO_NORM_SUM) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float* hr = x_ + (size_t)m * K;
                auto xin = [&](int i) -> float {
                    float v = hr[i];
                    if constexpr (PRO == PRO_NORM_SUM) {
                        for (int p = 0; p < naux_; ++p) v += aux_[((size_t)m * naux_ + p) * K + i];
                    }
                    return v;
                };
                float ss = 0.f;
                for (int i = threadIdx.x; i < K; i += 256) { const float v = xin(i); ss = fmaf(v, v, ss); }
                ss = block_sum(ss, red);
                const float inv = 1.0f / sqrtf(ss / K + a.eps);
                for (int i = threadIdx.x; i < K; i += 256) {
                    const float v = xin(i);
                    if constexpr (PRO == PRO_NORM_SUM) { if (blockIdx.x == 0) a.h_out[(size_t)m * K + i] = v; }
                    store_x<XT>(xs, m * K + i, v * inv * to_f(gamma_[i]));
                }
            }
        } else if constexpr (PRO == PRO_PLAIN) {
            for (int i = threadIdx.x; i < M * K; i += 256) store_x<XT>(xs, i, x_[i]);
        }
    }
    __syncthreads();

    // ---- body ----
    float best_v[M];
    int best_i[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { best_v[m] = -INFINITY; best_i[m] = 0x7FFFFFFF; }

    for (int g = wave; g * OUT_PER_TRIP < N; g += nwaves) {
        const int n0 = g * OUT_PER_TRIP;
        float acc[R][M];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
        if (C > 0 && g == wave) {
            // consume the preloaded chunks
#pragma unroll
            for (int c = 0; c < (C > 0 ? C : 1); ++c) {
                const int k0 = lane * NW + c * 64 * NW;
                if constexpr (std::is_same<WT, bf16>::value && std::is_same<XT, bf16>::value) {
                    uint4 xr[M];
#pragma unroll
                    for (int m = 0; m < M; ++m) xr[m] = *reinterpret_cast<const uint4*>(xs + (size_t)m * K + k0);
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int m = 0; m < M; ++m) acc[r][m] = dot8_bf16(pre[r][c], xr[m], acc[r][m]);
                    continue;
                }
                float xf[M][NW];
#pragma unroll
                for (int m = 0; m < M; ++m) XLoad<XT, NW>::load(xs + (size_t)m * K + k0, xf[m]);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float wf[NW];
                    WTraits<WT>::decode(pre[r][c], wf);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        if constexpr (FP8) {
                            float p = 0.f;
#pragma unroll
                            for (int j = 0; j < NW; ++j) p = fmaf(wf[j], xf[m][j], p);
                            acc[r][m] = fmaf(psc[r][c], p, acc[r][m]);
                        } else {
#pragma unroll
                            for (int j = 0; j < NW; ++j) acc[r][m] = fmaf(wf[j], xf[m][j], acc[r][m]);
                        }
                    }
                }
            }
        } else {
            const WT* wrow[R];
            const bf16* srow[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row_of(n0, r);
                wrow[r] = reinterpret_cast<const WT*>(w_) + (size_t)row * K;
                srow[r] = wscale_ ? wscale_ + (size_t)(row >> 7) * (K >> 7) : nullptr;
            }
            if constexpr (FP8) gemv_rows_fp8<XT, M, R>(wrow, srow, xs, K, K, lane, acc);
            else gemv_rows<WT, XT, M, R>(wrow, xs, K, K, lane, acc);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
        // ---- epilogue (lane 0 of the wave) ----
        if (lane == 0) {
            if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int r = 0; r < R / 2; ++r)
                    if (n0 + r < N) {
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const float gt = acc[r][m], up = acc[r + R / 2][m];
                            *(a.out + (size_t)m * a.ld_out + n0 + r) = gt / (1.0f + __expf(-gt)) * up;
                        }
                    }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (n0 + r < N) {
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const size_t o = (size_t)m * a.ld_out + n0 + r;
                            if constexpr (EPI == EPI_RESID) {
                                const float base = (C > 0 && g == wave) ? resv[r][m] : *(aux_ + o);
                                *(a.out + o) = base + acc[r][m];
                            } else if constexpr (EPI == EPI_LOGITS) {
                                a.out[o] = acc[r][m];          // read by later launches only: ordinary stores
                            } else {
                                *(a.out + o) = acc[r][m];
                            }
                            if constexpr (EPI == EPI_LOGITS) {
                                if (acc[r][m] > best_v[m]) { best_v[m] = acc[r][m]; best_i[m] = n0 + r; }
                            }
                        }
                    }
            }
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
        __syncthreads();  // s_bv may still be read by the prologue reduction of a slower wave
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) { s_bv[wid][m] = best_v[m]; s_bi[wid][m] = best_i[m]; }
        }
        __syncthreads();
        if (threadIdx.x < M) {
            const int m = threadIdx.x;
            float bv = s_bv[0][m];
            int bi = s_bi[0][m];
            for (int w = 1; w < 4; ++w)
                if (s_bv[w][m] > bv || (s_bv[w][m] == bv && s_bi[w][m] < bi)) { bv = s_bv[w][m]; bi = s_bi[w][m]; }
            a.amax_val[(size_t)m * gridDim.x + blockIdx.x] = bv;
            a.amax_idx[(size_t)m * gridDim.x + blockIdx.x] = bi;
        }
    }
    tls.end();
}

// h[b][:] = E[token[b]][:]   (step entry: pgk_engine_set_state; afterwards finalize_kernel keeps h current)
__global__ void embed_kernel(const bf16* embed, const int32_t* tokens, float* h, int H, const int32_t* positions,
                             const float* rope_cos, const float* rope_sin, float* cur_cos, float* cur_sin, int half,
                             int max_seq) {
    const int b = blockIdx.x;
    const bf16* row = embed + (size_t)tokens[b] * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) h[(size_t)b * H + i] = to_f(row[i]);
    const int pos = min(positions[b], max_seq - 1);
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        cur_cos[(size_t)b * half + i] = rope_cos[(size_t)pos * half + i];
        cur_sin[(size_t)b * half + i] = rope_sin[(size_t)pos * half + i];
    }
}

// token[b] = argmax over workgroup partials (ties -> lowest index); position[b] += 1; log the token;
// h[b] = E[token] for the next step; the last workgroup of the step's last chunk bumps the step counter.
// One workgroup per sequence (grid = M).  Every workgroup reads the step counter before it takes an arrival ticket
// (step_counter[1]); the bump is made by whoever draws the last ticket, so it cannot overtake a read.  (One workgroup
// for the whole chunk, a wave per sequence, took 25 us at M = 8 and 57 us at M = 16 - a dependent walk per sequence.)
__global__ __launch_bounds__(256) void finalize_kernel(unsigned long long* tl, const float* amax_val, const int* amax_idx, int nblk,
                                                       int32_t* tokens, int32_t* positions, int32_t* token_log,
                                                       int32_t* step_counter, int log_width, int log_cap,
                                                       const bf16* embed, float* h, int H, int bump,
                                                       unsigned long long* clk_log, const float* rope_cos,
                                                       const float* rope_sin, float* cur_cos, float* cur_sin, int half,
                                                       int max_seq, int M, const int32_t* sampled) {
    const TLStamp tls(tl);
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_tok, s_pos;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, b = blockIdx.x;
    const int step = step_counter[0];
    int bi;
    if (sampled) {
        bi = sampled[b];                                  // temperature / top-k / top-p draw (ops_sampling.hip)
    } else {
        float bv = -INFINITY;
        bi = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < nblk; i += 256) {
            const float v = amax_val[(size_t)b * nblk + i];
            const int ix = amax_idx[(size_t)b * nblk + i];
            if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
        __syncthreads();
        bv = sv[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (bi == 0x7FFFFFFF) bi = 0;
    }
    if (threadIdx.x == 0) {
        s_tok = bi;
        tokens[b] = bi;
        const int npos = positions[b] + 1;
        positions[b] = npos;
        s_pos = min(npos, max_seq - 1);
        if (step < log_cap) token_log[(size_t)step * log_width + b] = bi;
        if (b == 0 && step < log_cap && clk_log) {  // shader-clock / 100 MHz wall-clock stamps (diagnostic only)
            clk_log[2 * (size_t)step] = __builtin_amdgcn_s_memtime();
            clk_log[2 * (size_t)step + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
    __syncthreads();
    // next step's inputs of this sequence: its embedding row (8 bf16 per lane) and the RoPE row of its next position
    const bf16* erow = embed + (size_t)s_tok * H;
    float* hrow = h + (size_t)b * H;
    if ((H & 7) == 0) {
        for (int v = threadIdx.x; v < (H >> 3); v += 256) {
            const uint4 raw = *reinterpret_cast<const uint4*>(erow + v * 8);
            float f[8];
            WTraits<bf16>::decode(raw, f);
            *reinterpret_cast<float4*>(hrow + v * 8) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(hrow + v * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
        }
    } else {
        for (int i = threadIdx.x; i < H; i += 256) hrow[i] = to_f(erow[i]);
    }
    for (int i = threadIdx.x; i < half; i += 256) {
        cur_cos[(size_t)b * half + i] = rope_cos[(size_t)s_pos * half + i];
        cur_sin[(size_t)b * half + i] = rope_sin[(size_t)s_pos * half + i];
    }
    if (bump && threadIdx.x == 0) {
        if (M == 1) {
            step_counter[0] = step + 1;
        } else if (atomicAdd(&step_counter[1], 1) == M - 1) {   // every workgroup has read `step` before its own ticket
            atomicExch(&step_counter[1], 0);
            step_counter[0] = step + 1;
        }
    }
    tls.end();
}

// --------------------------------------------------------------------------------------------
// Decode attention.  Shared front end: QK-norm + RoPE of the new token's q/k, bf16 rounding of k/v.
// --------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* qkv;     // [B][(Hq+2Hkv)*D] fp32, pre-norm
    int qkv_ld;
    const bf16 *q_gamma, *k_gamma;
    float eps;
    const float *rope_cos, *rope_sin;   // [B][D/2]: the table rows of each sequence's CURRENT position
    bf16 *kcache, *vcache;              // this layer: [B][Hkv][max_seq][D]
    const int32_t* positions;
    int hq, hkv, max_seq;
    int span;             // split path: the positions [0, span) are what the slices cover (<= max_seq: the step's context tier, Engine::step_span)
    float scale;
    // split path
    float* part;          // [B][Hq][nsplit][D+2]
    int nsplit;
    float* attn_direct;   // whole-context variant (nsplit == 1): normalised output [B][Hq][D], no merge launch
    // fused o_proj path
    const bf16* w_o;      // [H][Hq*D] (fp8 codes on the merged o_proj path with fp8 weights)
    const bf16* w_o_scale; // fp8 W_o: [H/128][Hq*D/128] block scales
    int H, rows_per_block;
    float* opart;         // [B][Hkv][H]
    bf16* attn_direct16;  // whole-context variant: bf16 output instead of attn_direct (batched MFMA o_proj reads it)
    // GQA groups other than the instantiated 1 / 2 / 4 query heads per kv head run as several launches over head chunks:
    // this launch serves query heads kvh * g_total + g_off + [0, G) of every kv head (ordinary launch: g_total = G, g_off = 0)
    int g_total, g_off;
};

template <int D, int G>
struct NewToken {
    float qf[G][8], kn[8], vn[8];   // q (pre-scaled) and k, v of the new token, all as the bf16-rounded values every consumer sees
    uint4 qb[G], kbits, vbits;      // the same as packed bf16
};

// The new token's q/k/v, in two steps so that a kernel can put other loads between them: (1) every load - the fp32 q/k/v
// row slices of this lane, the QK-norm gammas, the RoPE row - issued back to back, nothing waited for; (2) pure ALU.
// Vector memory returns in issue order, so whatever is loaded FIRST is usable first: the fused kernel issues these small
// L2-resident loads ahead of its K/V and W_o streams and runs step (2) while those are still in flight.
template <int G>
struct NewTokenRaw {
    float4 lo[G + 2], hi[G + 2];   // q heads, k, v: this lane's 8 dims
    uint4 gq, gk;                  // 8 bf16 gammas each
    float4 cs[2], sn[2];           // RoPE row slice
};

template <int D, int G>
__device__ __forceinline__ void new_token_load(const AttnArgs& a, int b, int kvh, int lane, NewTokenRaw<G>& r) {
    constexpr int LPR = D / 8, HALF = D / 2;
    const int sub = lane % LPR;
    const float* row = a.qkv + (size_t)b * a.qkv_ld;
#pragma unroll
    for (int g = 0; g < G + 2; ++g) {
        const unsigned eoff = (g < G) ? (unsigned)(kvh * a.g_total + a.g_off + g) * D : (g == G ? (unsigned)(a.hq + kvh) * D : (unsigned)(a.hq + a.hkv + kvh) * D);
        r.lo[g] = *reinterpret_cast<const float4*>(row + eoff + sub * 8);
        r.hi[g] = *reinterpret_cast<const float4*>(row + eoff + sub * 8 + 4);
    }
    r.gq = r.gk = make_uint4(0, 0, 0, 0);
    if (a.q_gamma != nullptr) {
        r.gq = *reinterpret_cast<const uint4*>(a.q_gamma + sub * 8);
        r.gk = *reinterpret_cast<const uint4*>(a.k_gamma + sub * 8);
    }
    const int dd = (sub * 8) % HALF;                    // the lane's 8 dims stay inside one half (8 | HALF)
    const float* cs = a.rope_cos + (size_t)b * HALF + dd;   // address independent of the position: no extra round trip
    const float* sn = a.rope_sin + (size_t)b * HALF + dd;
    r.cs[0] = *reinterpret_cast<const float4*>(cs); r.cs[1] = *reinterpret_cast<const float4*>(cs + 4);
    r.sn[0] = *reinterpret_cast<const float4*>(sn); r.sn[1] = *reinterpret_cast<const float4*>(sn + 4);
}

template <int D, int G>
__device__ __forceinline__ void new_token_finish(const AttnArgs& a, int lane, const NewTokenRaw<G>& r, NewToken<D, G>& t) {
    constexpr int LPR = D / 8;
    const int sub = lane % LPR;
    float raw[G + 2][8], gq[8], gk[8];
#pragma unroll
    for (int g = 0; g < G + 2; ++g) {
        raw[g][0] = r.lo[g].x; raw[g][1] = r.lo[g].y; raw[g][2] = r.lo[g].z; raw[g][3] = r.lo[g].w;
        raw[g][4] = r.hi[g].x; raw[g][5] = r.hi[g].y; raw[g][6] = r.hi[g].z; raw[g][7] = r.hi[g].w;
    }
    const bool has_norm = a.q_gamma != nullptr;
    if (has_norm) {
        WTraits<bf16>::decode(r.gq, gq);
        WTraits<bf16>::decode(r.gk, gk);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) gq[j] = gk[j] = 1.f;
    }
    const float csv[8] = {r.cs[0].x, r.cs[0].y, r.cs[0].z, r.cs[0].w, r.cs[1].x, r.cs[1].y, r.cs[1].z, r.cs[1].w};
    const float snv[8] = {r.sn[0].x, r.sn[0].y, r.sn[0].z, r.sn[0].w, r.sn[1].x, r.sn[1].y, r.sn[1].z, r.sn[1].w};
    // norm + rope of one head vector; this lane holds dims sub*8..+8, the rotate-half partner dims live
    // LPR/2 lanes away.
    auto norm_rope = [&](const float (&xin)[8], const float (&gamma)[8], float (&o)[8]) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = xin[j];
        if (has_norm) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(x[j], x[j], ss);
            ss = group_sum<LPR>(ss);
            const float inv = 1.0f / sqrtf(ss / D + a.eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] * inv * gamma[j];
        }
        const bool lo = sub < LPR / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float other = xor_half<LPR>(x[j]);
            o[j] = lo ? (x[j] * csv[j] - other * snv[j]) : (x[j] * csv[j] + other * snv[j]);
        }
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
        norm_rope(raw[g], gq, t.qf[g]);
#pragma unroll
        for (int j = 0; j < 8; ++j) t.qf[g][j] *= a.scale;
        Vec<bf16> qv;                   // q is bf16 from here on (the model's dtype): scores run on the packed bf16 dot
        qv.from_float(t.qf[g]);
        qv.to_float(t.qf[g]);
        t.qb[g] = qv.raw;
    }
    norm_rope(raw[G], gk, t.kn);
#pragma unroll
    for (int j = 0; j < 8; ++j) t.vn[j] = raw[G + 1][j];
    // the cache holds bf16: this step uses the rounded values too (identical to reading them back)
    Vec<bf16> kb, vb;
    kb.from_float(t.kn);
    vb.from_float(t.vn);
    kb.to_float(t.kn);
    vb.to_float(t.vn);
    t.kbits = kb.raw;
    t.vbits = vb.raw;
}

template <int D, int G>
__device__ __forceinline__ void prepare_new_token(const AttnArgs& a, int b, int kvh, int pos, int lane, NewToken<D, G>& t) {
    NewTokenRaw<G> r;
    new_token_load<D, G>(a, b, kvh, lane, r);
    new_token_finish<D, G>(a, lane, r, t);
}

template <int D, int G>
__device__ __forceinline__ void fold_new_token(const NewToken<D, G>& t, DecodeState<G>& st) {
    constexpr int LPR = D / 8;
    float s[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum = fmaf(t.qf[g][j], t.kn[j], dsum);
        dsum = group_sum<LPR>(dsum);
        s[g] = dsum;
    }
    st.update(s, t.vn);
}

// split path: grid (nsplit, Hkv, batch)
template <int D, int G, bool DIRECT>
__global__ __launch_bounds__(256) void attn_decode_kernel(unsigned long long* tl, AttnArgs a) {
    const TLStamp tls(tl);
    constexpr int LPR = D / 8, PPW = 64 / LPR, RS = D + 2;
    __shared__ __attribute__((aligned(16))) float lds[4 * PPW * G * RS];
    __shared__ float attn_out[DIRECT ? G * D : 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane % LPR;
    const int kvh = blockIdx.y, b = blockIdx.z;
    const size_t head_off = (((size_t)b * a.hkv + kvh) * a.max_seq) * D;
    if constexpr (DIRECT) {
        // One workgroup owns the whole (short) context.  As in attn_oproj_kernel, the first U0 position-groups per wave
        // are loaded from clamped addresses BEFORE the position is known, so the K/V bytes, the q/k/v row and the
        // position share one memory round trip (the split path below learns the position first, then walks:
        // two dependent trips - 9.2 us against 6.x for 8 sequences).
        // The position-independent part is the first 128 rows; rows 128-191 are requested as soon as the position is there
        // (a scalar load that overtakes the vector loads in flight), clamped to the LAST CACHED ROW instead of the cache's
        // last row: with 64 sequences at context ~150 every workgroup used to pull 192 rows whatever the context - 50 MB per
        // layer for 39 MB of live K/V, and this kernel is bandwidth-bound at that batch (3.2 TB/s of live bytes).
        constexpr int U0 = 8, UB = 4;
        NewTokenRaw<G> raw;                 // issue order = arrival order: the few L2-resident q/k/v bytes first, then the K/V rows
        new_token_load<D, G>(a, b, kvh, lane, raw);
        __builtin_amdgcn_sched_barrier(0);
        KVBatch<U0> kb0;
        kv_issue<D, U0, 4>(kb0, a.kcache + head_off, a.vcache + head_off, wid * PPW, a.max_seq - 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        const int pos = load_uniform_i32(a.positions + b);   // scalar path (pgk_device.hip.h): not queued behind the vector loads in flight
        const int c1 = min(pos, a.max_seq);
        KVBatch<UB> kb1;
        kv_issue<D, UB, 4>(kb1, a.kcache + head_off, a.vcache + head_off, U0 * 4 * PPW + wid * PPW, max(c1 - 1, 0), lane);
        __builtin_amdgcn_sched_barrier(0);
        NewToken<D, G> t;
        new_token_finish<D, G>(a, lane, raw, t);
        if (pos < a.max_seq && wid == 0 && lane < LPR && a.g_off == 0) {
            *reinterpret_cast<uint4*>(a.kcache + head_off + (size_t)pos * D + sub * 8) = t.kbits;
            *reinterpret_cast<uint4*>(a.vcache + head_off + (size_t)pos * D + sub * 8) = t.vbits;
        }
        DecodeState<G> st;
        st.init();
        kv_consume<D, G, U0, 4>(kb0, wid * PPW, c1, t.qb, lane, st);
        kv_consume<D, G, UB, 4>(kb1, U0 * 4 * PPW + wid * PPW, c1, t.qb, lane, st);
        if (c1 > (U0 + UB) * 4 * PPW) decode_walk_trips<D, G>(a.kcache + head_off, a.vcache + head_off, (U0 + UB) * 4 * PPW, c1, t.qb, lane, wid, st);
        if (pos < a.max_seq && wid == 0 && lane < LPR) fold_new_token<D, G>(t, st);
        decode_block_merge_lds<D, G>(st, lds, attn_out, lane, wid);
        if (a.attn_direct16) {
            for (int e = threadIdx.x; e < G * D; e += 256) a.attn_direct16[((size_t)b * a.hq + (size_t)kvh * a.g_total + a.g_off) * D + e] = from_f<bf16>(attn_out[e]);
        } else {
            for (int e = threadIdx.x; e < G * D; e += 256) a.attn_direct[((size_t)b * a.hq + (size_t)kvh * a.g_total + a.g_off) * D + e] = attn_out[e];
        }
        tls.end();
        return;
    }
    // Slices are cut by ABSOLUTE position (slice s = cache rows [s * chunk, (s + 1) * chunk), chunk from the step's context tier a.span <= cache length):
    // no address depends on the context length, so the new token's q/k/v and the slice's first 128 K/V rows are requested
    // before the position is even known (clamped addresses; masked later).  Before, the walk started one scalar and one
    // vector round trip later (position -> slice bounds -> addresses).  Slices beyond the context write empty records.
    constexpr int U1 = 8;
    const int chunk = decode_chunk_len(a.span, a.nsplit, 4 * PPW);
    const int c0 = (int)blockIdx.x * chunk;
    NewTokenRaw<G> raw;
    new_token_load<D, G>(a, b, kvh, lane, raw);
    __builtin_amdgcn_sched_barrier(0);
    KVBatch<U1> kb0;
    kv_issue<D, U1, 4>(kb0, a.kcache + head_off, a.vcache + head_off, c0 + wid * PPW, a.max_seq - 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    const int pos = load_uniform_i32(a.positions + b);   // scalar path (pgk_device.hip.h): not queued behind the vector loads in flight
    const int ctx = min(pos + 1, a.max_seq);
    tls.phase(0);
    NewToken<D, G> t;
    new_token_finish<D, G>(a, lane, raw, t);
    tls.phase(1);
    // the LAST slice runs to the end of the context wherever that is: the tier (a.span) comes from a host-side bound on the
    // position, and a caller that moved the device-resident positions past it behind the library's back must lose speed, not rows
    const bool last_slice = (int)blockIdx.x == a.nsplit - 1;
    const int c1 = last_slice ? ctx : min(c0 + chunk, ctx);
    const bool owns_new = (pos < a.max_seq) && (pos >= c0) && (last_slice || pos < c0 + chunk);
    if (owns_new && wid == 0 && lane < LPR && a.g_off == 0) {
        *reinterpret_cast<uint4*>(a.kcache + head_off + (size_t)pos * D + sub * 8) = t.kbits;
        *reinterpret_cast<uint4*>(a.vcache + head_off + (size_t)pos * D + sub * 8) = t.vbits;
    }
    DecodeState<G> st;
    st.init();
    const int cend = owns_new ? min(c1, pos) : c1;       // cached rows of this slice; the new token's row is folded from registers
    kv_consume<D, G, U1, 4>(kb0, c0 + wid * PPW, cend, t.qb, lane, st);
    if (cend > c0 + U1 * 4 * PPW)
        decode_walk_trips<D, G>(a.kcache + head_off, a.vcache + head_off, c0 + U1 * 4 * PPW, cend, t.qb, lane, wid, st);
    if (owns_new && wid == 0 && lane < LPR) fold_new_token<D, G>(t, st);
    tls.phase(2);
    if constexpr (DIRECT) {
        // this workgroup saw the whole context: normalise here and skip the merge launch
        decode_block_merge_lds<D, G>(st, lds, attn_out, lane, wid);
        for (int e = threadIdx.x; e < G * D; e += 256) a.attn_direct[((size_t)b * a.hq + (size_t)kvh * a.g_total + a.g_off) * D + e] = attn_out[e];
    } else {
        decode_block_merge<D, G>(st, lds, a.part + (((size_t)b * a.hq + (size_t)kvh * a.g_total + a.g_off) * a.nsplit + blockIdx.x) * RS,
                                 (size_t)a.nsplit * RS, lane, wid);
    }
    tls.end();
}

// split path, step 2: merge the nsplit (<= 64) chunk records of every head into the normalised attention
// vector attn[b][h*D + d] (fp32).  grid (Hq, batch), D threads (D = 64 or 128: whole waves).  Lane s of wave 0
// owns record s's (m, l): one load each, a wave max and a wave sum give the weights; then every thread sums its
// element over the records with independent loads.
template <int D>
__global__ void attn_merge_kernel(unsigned long long* tl, const float* part, float* attn, int hq, int nsplit) {
    const TLStamp tls(tl);
    __shared__ float w_s[64];
    __shared__ float inv_l;
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* recs = part + ((size_t)b * hq + h) * nsplit * (D + 2);
    if (threadIdx.x < 64) {
        const int s = threadIdx.x;
        const int sc = min(s, nsplit - 1);
        float m = recs[(size_t)sc * (D + 2)], l = recs[(size_t)sc * (D + 2) + 1];
        if (s >= nsplit) { m = -INFINITY; l = 0.f; }
        const float mx = wave_max(m);
        const float w = (m == -INFINITY) ? 0.f : __expf(m - mx);
        const float tot = wave_sum(w * l);
        w_s[s] = w;
        if (s == 0) inv_l = tot > 0.f ? 1.0f / tot : 0.f;
    }
    __syncthreads();
    float o = 0.f;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = recs[(size_t)(s + u) * (D + 2) + 2 + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) o = fmaf(w_s[s + u], v[u], o);
    }
    for (; s < nsplit; ++s) o = fmaf(w_s[s], recs[(size_t)s * (D + 2) + 2 + d], o);
    attn[((size_t)b * hq + h) * D + d] = o * inv_l;
    tls.end();
}

// fused path: grid ((H / rows_per_block) * Hkv, 1, batch), 256 threads.  Every workgroup of a KV head recomputes
// that head's (short-context) attention from L2-resident K/V, then multiplies it with ITS slice of W_o
// (rows_per_block output rows x G*D columns), whose loads were issued before anything else.
template <int D, int G>
__global__ __launch_bounds__(256) void attn_oproj_kernel(unsigned long long* tl, AttnArgs a) {
    const TLStamp tls(tl);
#ifdef PGK_PHASE_STAMPS
    if (threadIdx.x == 0) g_phase_tl = tl;      // same value from every workgroup of the launch
    __syncthreads();
#endif
    constexpr int NWV = 4;   // 8 waves measured slower: the kernel is issue-bound per SIMD, not per wave
    constexpr int LPR = D / 8, PPW = 64 / LPR, RS = D + 2;
    constexpr int GD = G * D, LPW = GD / 8;          // lanes covering one W_o row slice
    constexpr int RPP = NWV * 64 / LPW;              // rows per pass of the workgroup
    constexpr int PRE = 4;                           // preloaded passes
    constexpr int U0 = 12;                           // position-groups per wave in the first KV batch
    __shared__ __attribute__((aligned(16))) float lds[NWV * PPW * G * RS];
    __shared__ __attribute__((aligned(16))) float attn[GD];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPR;   // wid in an SGPR: per-wave branches stay scalar
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs in linear order, so with the kv head as the
    // FASTEST index all row slices of kv head h land on XCD h % 8 and its K/V rows are fetched into ONE L2 instead of eight
    // (PMC: 9.6 MB of HBM traffic per launch for 4.9 MB of algorithmic bytes with the row slice fastest).  Speed only.
    const int kvh = blockIdx.x % a.hkv, rb = blockIdx.x / a.hkv, b = blockIdx.z;
    const int r0 = rb * a.rows_per_block;
    const int lr = threadIdx.x % LPW, rip = threadIdx.x / LPW;
    const int npass = a.rows_per_block / RPP;
    const bf16* wbase = a.w_o + (size_t)kvh * GD + lr * 8;
    const int ldw = a.hq * D;
    // Issue order = arrival order (vector memory returns in order): first the few L2-resident bytes the new token's
    // q/k/v need, then the cached K/V rows, last the W_o slice that is only consumed at the very end.  (The first
    // version issued W_o and K/V first: the q/k/v row then arrived behind ~1.4 KiB per lane of HBM traffic and the
    // norm / RoPE work started 2.9 us into the workgroup - in-kernel stamps, tools/phase_stamps.py.)
    const size_t head_off = (((size_t)b * a.hkv + kvh) * a.max_seq) * D;
    uint4 pre[PRE];
    KVBatch<U0> kb0;
    NewTokenRaw<G> raw;
    new_token_load<D, G>(a, b, kvh, lane, raw);
    __builtin_amdgcn_sched_barrier(0);
    // first KV batch: U0 position-groups per wave = positions [0, U0*NWV*PPW); addresses do not depend on
    // the context length (clamped), so these loads share the round trip of everything else in this kernel
    kv_issue<D, U0, NWV>(kb0, a.kcache + head_off, a.vcache + head_off, wid * PPW, a.max_seq - 1, lane);
#pragma unroll
    for (int p = 0; p < PRE; ++p)  // unconditional (clamped) so nothing waits on these until the GEMV
        pre[p] = load_nt16(wbase + (size_t)(r0 + min(p, npass - 1) * RPP + rip) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    const int pos = load_uniform_i32(a.positions + b);     // scalar path: not queued behind the 50-odd vector loads above
    tls.phase(0);
    NewToken<D, G> t;
    new_token_finish<D, G>(a, lane, raw, t);
    if (rb == 0 && pos < a.max_seq && wid == 0 && lane < LPR) {
        *reinterpret_cast<uint4*>(a.kcache + head_off + (size_t)pos * D + sub * 8) = t.kbits;
        *reinterpret_cast<uint4*>(a.vcache + head_off + (size_t)pos * D + sub * 8) = t.vbits;
    }
    tls.phase(1);
    DecodeState<G> st;
    st.init();
    const int c1 = min(pos, a.max_seq);
    kv_consume<D, G, U0, NWV>(kb0, wid * PPW, c1, t.qb, lane, st);
    if (c1 > U0 * NWV * PPW)
        decode_walk_trips<D, G, 8, NWV>(a.kcache + head_off, a.vcache + head_off, U0 * NWV * PPW, c1, t.qb, lane, wid, st);
    if (wid == 0 && lane < LPR) fold_new_token<D, G>(t, st);
    tls.phase(2);
    decode_block_merge_lds<D, G, NWV>(st, lds, attn, lane, wid);
    tls.phase(3);

    float xf[8];
    {
        const float4 u = *reinterpret_cast<const float4*>(attn + lr * 8), v = *reinterpret_cast<const float4*>(attn + lr * 8 + 4);
        xf[0] = u.x; xf[1] = u.y; xf[2] = u.z; xf[3] = u.w; xf[4] = v.x; xf[5] = v.y; xf[6] = v.z; xf[7] = v.w;
    }
    float* outp = a.opart + ((size_t)b * a.hkv + kvh) * a.H;
    for (int p = 0; p < npass; ++p) {
        const int row = r0 + p * RPP + rip;
        uint4 w = (p < PRE) ? pre[p < PRE ? p : 0] : load_nt16(wbase + (size_t)row * ldw);
        float wf[8];
        WTraits<bf16>::decode(w, wf);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(wf[j], xf[j], acc);
        acc = group_sum<LPW>(acc);
        if (lr == 0) outp[row] = acc;
    }
    tls.phase(4);
    tls.end();
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused batch-1 attention + o_proj with both products on the matrix pipe (head_dim 128).
//
// attn_oproj_kernel above spends 2.1 of its 5.5 us in the score / P.V loop: one wave per SIMD, ~750 VALU instructions per 48
// positions (v_dot2c at ~10 cycles of issue, DPP reductions).  The first attempt to move Q.K^T to MFMA loaded the K rows
// from global memory in A-fragment shape (lane = row): 64 separate 16-byte pieces per instruction, and the kernel lost
// more in its load issue phase than the MFMAs won.  Here the cached rows are staged ROW-MAJOR by LDS-DMA -
// the same bytes per instruction as the register loads they replace, 1 KiB contiguous each - and the fragments come out
// of LDS: the one-tile prefill kernel's scheme (ops_attention.hip, attn_short_kernel) with the G query heads of the kv
// head as the only live columns:
//   * chunks of 192 positions: K and V rows [c0, c0 + 192) -> LDS (K: 16-byte chunks XOR row & 15; V: layout (b) of the
//     CDNA guide for ds_read_b64_tr_b16); chunk 0 is requested before the position is known (clamped rows);
//   * wave w takes tiles w, w + 4, w + 8 of the chunk (16 positions each): S^T = K.Q^T - rows = positions, columns =
//     heads - so a lane holds ONE head's scores of 4 consecutive positions per tile; max / sum are lane-local + two
//     shuffles; exp'd and packed to bf16 they are the B operand of O^T = V^T.P^T with no LDS round trip (k-slot j of lane
//     quarter q <-> position 16 tile(j >> 2) + 4 q + (j & 3), V read with the same slots through the transposing read);
//   * every wave keeps a running (m, l, O^T) across chunks; at the end the four waves' states and the new token's
//     (score, 1, v) meet in LDS and are combined per output element, then the W_o slice product as before.
// The new token's k/v never enter the LDS images (the DMA of its cache row would race the write): it is a fifth partial.
constexpr int AM_CHUNK = 192;   // positions per staged chunk: 12 tiles, 3 per wave

__device__ __forceinline__ int am_koff(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
__device__ __forceinline__ int am_voff(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// OPROJ = false: the whole-context BATCH attention (one workgroup per (sequence, kv head), grid (Hkv, 1, batch)): the same
// kernel without the W_o slice - the normalised heads leave as bf16 (attn_direct16) or fp32 (attn_direct) rows.
template <int G, bool OPROJ = true>
__global__ __launch_bounds__(256) void attn_oproj_mfma_kernel(unsigned long long* tl, AttnArgs a) {
    const TLStamp tls(tl);
    typedef __bf16 am_bf16x8 __attribute__((ext_vector_type(8)));
    typedef float am_f32x4 __attribute__((ext_vector_type(4)));
    typedef short am_v4s __attribute__((ext_vector_type(4)));
    constexpr int D = 128, NWV = 4, LPR = 16, RS = D + 4;   // a state record: o[128], then m, l (16-byte aligned rows)
    constexpr int GD = G * D, LPW = GD / 8, RPP = NWV * 64 / LPW, PRE = OPROJ ? 4 : 0;
    extern __shared__ __attribute__((aligned(16))) char am_lds[];          // K image | V image (AM_CHUNK rows x 256 bytes each)
    char* k_lds = am_lds;
    char* v_lds = am_lds + AM_CHUNK * 256;
    __shared__ __attribute__((aligned(16))) float part[NWV + 1][G][RS];    // (o[128], m, l) of the four waves + the new token
    __shared__ __attribute__((aligned(16))) float attn[GD];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int kvh = blockIdx.x % a.hkv, rb = blockIdx.x / a.hkv, b = blockIdx.z;   // kv head fastest: XCD-aware (attn_oproj_kernel)
    const int r0 = rb * a.rows_per_block;
    const int lr = threadIdx.x % LPW, rip = threadIdx.x / LPW;
    const int npass = a.rows_per_block / RPP;
    const bf16* wbase = a.w_o + (size_t)kvh * GD + lr * 8;
    const int ldw = a.hq * D;
    const size_t head_off = (((size_t)b * a.hkv + kvh) * a.max_seq) * D;
    const bf16* kc = a.kcache + head_off;
    const bf16* vc = a.vcache + head_off;

    // Issue order = arrival order, and every wait on it is written out here: the new token's inputs, the cache chunk and
    // nothing else go through LDS-DMA issued as inline asm, so the compiler neither counts them nor - as it does for the
    // builtin form - answers any vector load older than them with vmcnt(0) (which made the norm / RoPE work wait for
    // the whole chunk).  Per wave: NRAW instructions for ITS copy of the new token's fp32 q/k/v slices, the two gammas
    // and the RoPE row (3 KiB for G = 2: three instructions instead of 14 register loads), 24 for the chunk, then the four
    // W_o preloads as ordinary loads.
    auto dma = [](const void* src, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
    };
    auto lds_u32 = [](const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; };
    constexpr int HB = (G + 2) * 512;                     // bytes of the head slices in a wave's slot; then gq, gk, cos, sin (256 each)
    constexpr int NRAW = (HB + 1024 + 1023) / 1024;
    __shared__ __attribute__((aligned(16))) char raw_lds[NWV][NRAW * 1024];
    {
        const char* row = reinterpret_cast<const char*>(a.qkv + (size_t)b * a.qkv_ld);
        const char* cosr = reinterpret_cast<const char*>(a.rope_cos + (size_t)b * 64);
        const char* sinr = reinterpret_cast<const char*>(a.rope_sin + (size_t)b * 64);
        const char* gqp = a.q_gamma ? reinterpret_cast<const char*>(a.q_gamma) : cosr;      // no QK-norm: any valid bytes
        const char* gkp = a.k_gamma ? reinterpret_cast<const char*>(a.k_gamma) : cosr;
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int o = 1024 * i + 16 * lane;
            const char* src;
            if (o < HB) {
                const int hs = o >> 9, within = o & 511;
                const int elem = (hs < G) ? (kvh * a.g_total + a.g_off + hs) * D : (hs == G ? (a.hq + kvh) * D : (a.hq + a.hkv + kvh) * D);
                src = row + (size_t)elem * 4 + within;
            } else {
                const int o2 = min(o - HB, 1023), seg = o2 >> 8, within = o2 & 255;
                src = (seg == 0 ? gqp : seg == 1 ? gkp : seg == 2 ? cosr : sinr) + within;
            }
            dma(src, lds_u32(&raw_lds[wid][0]) + 1024 * i);
        }
    }
    auto stage = [&](int c0) {      // rows [c0, c0 + AM_CHUNK) of K and V: instruction j = 4 rows, lane i -> row 4 j + (i >> 4), chunk position i & 15
#pragma unroll
        for (int i = 0; i < AM_CHUNK / 4 / NWV; ++i) {
            const int j = wid + NWV * i;
            const int rl = 4 * j + q4, rg = min(c0 + rl, a.max_seq - 1);
            dma(kc + (size_t)rg * D + ((l15 ^ (rl & 15)) << 3), lds_u32(k_lds) + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < AM_CHUNK / 4 / NWV; ++i) {
            const int j = wid + NWV * i;
            const int rl = 4 * j + q4, rg = min(c0 + rl, a.max_seq - 1);
            dma(vc + (size_t)rg * D + ((l15 ^ (((rl & 3) << 2) | ((rl >> 2) & 3))) << 3), lds_u32(v_lds) + j * 1024);
        }
    };
    stage(0);
    uint4 pre[PRE > 0 ? PRE : 1];
#pragma unroll
    for (int p = 0; p < PRE; ++p) pre[p] = load_nt16(wbase + (size_t)(r0 + min(p, npass - 1) * RPP + rip) * ldw);
    __builtin_amdgcn_sched_barrier(0);
    const int pos = load_uniform_i32(a.positions + b);
    tls.phase(0);
    // in-order return: once at most the operations issued after them are outstanding, this wave's copy of the new token's inputs is in its slot
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AM_CHUNK / 4 / NWV) + PRE) : "memory");
    // The new token's G + 2 head vectors (q heads, k, v) are SPLIT over the waves - item i goes to wave i % 4 - instead of
    // every wave normalising and rotating all of them (0.74 us of ALU per wave in attn_oproj_kernel): each wave reads its
    // item from its own copy of the inputs, and the results meet in LDS behind the barrier that the cache chunk needs anyway.
    __shared__ __attribute__((aligned(16))) uint4 q_sh[G][16], k_sh[16], v_sh[16];   // bf16, 16 chunks of 8 dims per vector
    {
        const char* slot = &raw_lds[wid][0];
        const int sub = l15;                                 // the lane's 8 dims: sub * 8 .. + 8 (all four lane quarters compute the same)
        const int dd = (sub * 8) % 64;
        float csv[8], snv[8];
        {
            const float4 c0v = *reinterpret_cast<const float4*>(slot + HB + 512 + dd * 4), c1v = *reinterpret_cast<const float4*>(slot + HB + 512 + dd * 4 + 16);
            const float4 s0v = *reinterpret_cast<const float4*>(slot + HB + 768 + dd * 4), s1v = *reinterpret_cast<const float4*>(slot + HB + 768 + dd * 4 + 16);
            csv[0] = c0v.x; csv[1] = c0v.y; csv[2] = c0v.z; csv[3] = c0v.w; csv[4] = c1v.x; csv[5] = c1v.y; csv[6] = c1v.z; csv[7] = c1v.w;
            snv[0] = s0v.x; snv[1] = s0v.y; snv[2] = s0v.z; snv[3] = s0v.w; snv[4] = s1v.x; snv[5] = s1v.y; snv[6] = s1v.z; snv[7] = s1v.w;
        }
        const bool has_norm = a.q_gamma != nullptr;
        for (int item = wid; item < G + 2; item += NWV) {    // wave-uniform
            const float4 lo = *reinterpret_cast<const float4*>(slot + item * 512 + sub * 32), hi = *reinterpret_cast<const float4*>(slot + item * 512 + sub * 32 + 16);
            float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (item < G + 1) {                              // q heads and k: QK-norm (optional) + RoPE, the arithmetic of new_token_finish
                if (has_norm) {
                    float gm[8];
                    WTraits<bf16>::decode(*reinterpret_cast<const uint4*>(slot + HB + (item < G ? 0 : 256) + sub * 16), gm);
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss = fmaf(x[j], x[j], ss);
                    ss = group_sum<LPR>(ss);
                    const float inv = 1.0f / sqrtf(ss / D + a.eps);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = x[j] * inv * gm[j];
                }
                const bool lo_half = sub < LPR / 2;
                float o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float other = xor_half<LPR>(x[j]);
                    o8[j] = lo_half ? (x[j] * csv[j] - other * snv[j]) : (x[j] * csv[j] + other * snv[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (item < G) ? o8[j] * a.scale : o8[j];
            }
            Vec<bf16> vb;
            vb.from_float(x);
            if (lane < 16) {
                if (item < G) q_sh[item][lane] = vb.raw;
                else if (item == G) k_sh[lane] = vb.raw;
                else v_sh[lane] = vb.raw;
            }
        }
    }
    tls.phase(1);
    uint4 qf[4];
    const int c1 = min(pos, a.max_seq);                 // cached positions [0, c1)
    float m_run = -INFINITY, l_run = 0.f;                // of head l15 (lanes l15 >= G carry dummies)
    am_f32x4 o[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) o[i] = am_f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = l15 >> 2, tp = l15 & 3;
    for (int c0 = 0; c0 == 0 || c0 < c1; c0 += AM_CHUNK) {
        if (c0 > 0) {
            __syncthreads();                               // everyone is done with the previous chunk's images
            stage(c0);
        }
        // chunk 0: everything but the four W_o preloads issued behind it
        if (c0 > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRE) : "memory");
        __syncthreads();
        if (c0 == 0) {
            // q as B fragments: lane (l15 = head, q4) of k-step ks holds dims 32 ks + 8 q4 .. + 8 of head l15
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 v = q_sh[min(l15, G - 1)][4 * ks + q4];
                qf[ks] = l15 < G ? v : make_uint4(0, 0, 0, 0);
            }
            // the new token: fifth partial (score = q . k_new per head, weight 1, value v_new) and its cache row - the last wave,
            // which had the fewest items above
            if (wid == NWV - 1 && lane < LPR) {
                float kn[8], vn[8];
                Vec<bf16> kb, vb2;
                kb.raw = k_sh[lane]; vb2.raw = v_sh[lane];
                kb.to_float(kn); vb2.to_float(vn);
                const bool live = pos < a.max_seq;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    Vec<bf16> qb;
                    qb.raw = q_sh[g][lane];
                    float qv[8];
                    qb.to_float(qv);
                    float dsum = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dsum = fmaf(qv[j], kn[j], dsum);
                    dsum = group_sum<LPR>(dsum);
                    if (lane == 0) { part[NWV][g][D] = live ? dsum : -INFINITY; part[NWV][g][D + 1] = live ? 1.f : 0.f; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) part[NWV][g][lane * 8 + j] = vn[j];
                }
                if (rb == 0 && live) {
                    *reinterpret_cast<uint4*>(a.kcache + head_off + (size_t)pos * D + lane * 8) = kb.raw;
                    *reinterpret_cast<uint4*>(a.vcache + head_off + (size_t)pos * D + lane * 8) = vb2.raw;
                }
            }
        }
        // S^T tiles of this wave: rows = positions c0 + 16 t + 4 q4 + r, column = head l15
        am_f32x4 s[3];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tile = wid + NWV * u;
            s[u] = am_f32x4{0.f, 0.f, 0.f, 0.f};
            if (c0 + 16 * tile < c1) {                     // wave-uniform
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const uint4 ka = *reinterpret_cast<const uint4*>(k_lds + am_koff(16 * tile + l15, 4 * ks + q4));
                    s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(am_bf16x8, ka), __builtin_bit_cast(am_bf16x8, qf[ks]), s[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = c0 + 16 * tile + 4 * q4 + r < c1;
                s[u][r] = ok ? s[u][r] : -INFINITY;
                mx = fmaxf(mx, s[u][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
        float ls = 0.f;
        uint32_t pk[3][2];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = (m_new == -INFINITY) ? 0.f : __expf(s[u][r] - m_new);
            pk[u][0] = pack_bf16x2(p[0], p[1]);
            pk[u][1] = pack_bf16x2(p[2], p[3]);
            ls += (__uint_as_float(pk[u][0] << 16) + __uint_as_float(pk[u][0] & 0xFFFF0000u)) + (__uint_as_float(pk[u][1] << 16) + __uint_as_float(pk[u][1] & 0xFFFF0000u));
        }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < D / 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha;
        // O^T += V^T . P^T: step 0 pairs tiles (wid, wid + 4), step 1 tile wid + 8 with zeros
#pragma unroll
        for (int st2 = 0; st2 < 2; ++st2) {
            const int ta = wid + NWV * (2 * st2), tb = (st2 == 0) ? wid + NWV : ta;     // tb of step 1: any staged tile (weights zero)
            if (c0 + 16 * ta < c1) {                       // wave-uniform
                const uint4 pf = make_uint4(pk[2 * st2][0], pk[2 * st2][1], st2 == 0 ? pk[1][0] : 0u, st2 == 0 ? pk[1][1] : 0u);
                const int ra = 16 * ta + 4 * q4 + tq, rbv = 16 * tb + 4 * q4 + tq;
#pragma unroll
                for (int i = 0; i < D / 16; ++i) {
                    const am_v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_v4s*)(v_lds + am_voff(ra, 2 * i + (tp >> 1)) + 8 * (tp & 1)));
                    const am_v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_v4s*)(v_lds + am_voff(rbv, 2 * i + (tp >> 1)) + 8 * (tp & 1)));
                    const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
                    const uint4 va = make_uint4(u0.x, u0.y, u1.x, u1.y);
                    o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(am_bf16x8, va), __builtin_bit_cast(am_bf16x8, pf), o[i], 0, 0, 0);
                }
            }
        }
    }
    tls.phase(2);
    // the waves' states -> LDS: lane (l15 = head g, q4) holds dims 16 i + 4 q4 + r of head g
    if (l15 < G) {
        if (q4 == 0) { part[wid][l15][D] = m_run; part[wid][l15][D + 1] = l_run; }
#pragma unroll
        for (int i = 0; i < D / 16; ++i)
            *reinterpret_cast<float4*>(&part[wid][l15][16 * i + 4 * q4]) = make_float4(o[i][0], o[i][1], o[i][2], o[i][3]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GD; e += 256) {
        const int g = e / D, d = e % D;
        float mstar = -INFINITY;
#pragma unroll
        for (int w = 0; w <= NWV; ++w) mstar = fmaxf(mstar, part[w][g][D]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w <= NWV; ++w) {
            const float mw = part[w][g][D];
            const float wgt = (mw == -INFINITY) ? 0.f : __expf(mw - mstar);
            num = fmaf(wgt, part[w][g][d], num);
            den = fmaf(wgt, part[w][g][D + 1], den);
        }
        attn[e] = den > 0.f ? num / den : 0.f;
    }
    __syncthreads();
    tls.phase(3);
    if constexpr (OPROJ) {
        float xf[8];
        {
            const float4 u = *reinterpret_cast<const float4*>(attn + lr * 8), v = *reinterpret_cast<const float4*>(attn + lr * 8 + 4);
            xf[0] = u.x; xf[1] = u.y; xf[2] = u.z; xf[3] = u.w; xf[4] = v.x; xf[5] = v.y; xf[6] = v.z; xf[7] = v.w;
        }
        float* outp = a.opart + ((size_t)b * a.hkv + kvh) * a.H;
        for (int p = 0; p < npass; ++p) {
            const int row = r0 + p * RPP + rip;
            uint4 w = (p < PRE) ? pre[p < PRE ? p : 0] : load_nt16(wbase + (size_t)row * ldw);
            float wf[8];
            WTraits<bf16>::decode(w, wf);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(wf[j], xf[j], acc);
            acc = group_sum<LPW>(acc);
            if (lr == 0) outp[row] = acc;
        }
    } else {
        const size_t ob = ((size_t)b * a.hq + (size_t)kvh * G) * D;
        if (a.attn_direct16) {
            for (int e = threadIdx.x; e < GD; e += 256) a.attn_direct16[ob + e] = from_f<bf16>(attn[e]);
        } else {
            for (int e = threadIdx.x; e < GD; e += 256) a.attn_direct[ob + e] = attn[e];
        }
    }
    tls.phase(4);
    tls.end();
}

// long-context path, steps 2 + 3 in ONE launch: merge the split-KV records of a kv head's G query heads (the arithmetic
// of attn_merge_kernel, same order) and multiply the result with this workgroup's slice of W_o.  Grid as attn_oproj_kernel
// ((H / rows_per_block) * Hkv, kv head fastest: XCD-aware), output the same per-kv-head partial vectors, which the
// gate/up kernel's PRO_NORM_SUM prologue adds to the residual stream.  Replaces attn_merge_kernel + the o_proj GEMV:
// one launch and one dependent-kernel gap less per layer (context 2048, w8a16: the pair took 1.95 + 1.3 + 1.74 us of
// every 23.8 us layer; profiles/r02_config3_timeline.json).  W_o bf16 or fp8 (16 codes per lane, block scale in registers).
template <int D, int G, bool FP8>
__global__ __launch_bounds__(256) void attn_merge_oproj_kernel(unsigned long long* tl, AttnArgs a) {
    const TLStamp tls(tl);
    constexpr int RS = D + 2, GD = G * D;
    constexpr int NWT = FP8 ? 16 : 8;                // weights per 16-byte load
    constexpr int LPW = GD / NWT;                    // lanes covering one W_o row slice
    constexpr int RPP = 256 / LPW;                   // rows per pass of the workgroup
    constexpr int PRE = 4;
    static_assert(LPW <= 64 && 256 % LPW == 0, "row slice must fit a wave");
    __shared__ float w_s[G][64];
    __shared__ float inv_l[G];
    __shared__ __attribute__((aligned(16))) float attn[GD];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kvh = blockIdx.x % a.hkv, rb = blockIdx.x / a.hkv, b = blockIdx.z;
    const int r0 = rb * a.rows_per_block;
    const int lr = threadIdx.x % LPW, rip = threadIdx.x / LPW;
    const int npass = a.rows_per_block / RPP;
    const int ldw = a.hq * D;
    const int col0 = kvh * GD + lr * NWT;
    const char* wbase = reinterpret_cast<const char*>(a.w_o) + (size_t)col0 * (FP8 ? 1 : 2);
    const size_t row_bytes = (size_t)ldw * (FP8 ? 1 : 2);
    // the record words first (L2, written by the launch before), then the W_o stream: arrival order = issue order
    const float* hrecs = a.part + ((size_t)b * a.hq + (size_t)kvh * G) * a.nsplit * RS;
    float m = -INFINITY, l = 0.f;
    if (wid < G) {
        const int sc = min(lane, a.nsplit - 1);
        m = hrecs[((size_t)wid * a.nsplit + sc) * RS];
        l = hrecs[((size_t)wid * a.nsplit + sc) * RS + 1];
    }
    // ... and the records' value words of this thread's output element(s): up to 32 slices per element straight into
    // registers, BEFORE the W_o stream and before the barrier below.  (They used to be read after the barrier, eight at a
    // time: three to four dependent L2 round trips on the critical path of every layer at context 2048.)
    constexpr int EPT = (GD + 255) / 256, RPRE = 32;
    float rv[EPT][RPRE];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = min((int)threadIdx.x + 256 * i, GD - 1), g = e / D, d = e % D;
        const float* recs = hrecs + (size_t)g * a.nsplit * RS + 2 + d;
#pragma unroll
        for (int u = 0; u < RPRE; ++u) rv[i][u] = recs[(size_t)min(u, a.nsplit - 1) * RS];
    }
    __builtin_amdgcn_sched_barrier(0);
    uint4 pre[PRE];
    float psc[PRE];
#pragma unroll
    for (int p = 0; p < PRE; ++p) {
        const int row = r0 + min(p, npass - 1) * RPP + rip;
        pre[p] = load_nt16(wbase + (size_t)row * row_bytes);
        if constexpr (FP8) psc[p] = to_f(a.w_o_scale[(size_t)(row >> 7) * (ldw >> 7) + (col0 >> 7)]);
    }
    __builtin_amdgcn_sched_barrier(0);
    tls.phase(0);
    if (wid < G) {
        if (lane >= a.nsplit) { m = -INFINITY; l = 0.f; }
        const float mx = wave_max(m);
        const float w = (m == -INFINITY) ? 0.f : __expf(m - mx);
        const float tot = wave_sum(w * l);
        w_s[wid][lane] = w;
        if (lane == 0) inv_l[wid] = tot > 0.f ? 1.0f / tot : 0.f;
    }
    __syncthreads();
    tls.phase(1);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        if (e < GD) {
            const int g = e / D, d = e % D;
            const float* recs = hrecs + (size_t)g * a.nsplit * RS;
            float o = 0.f;
#pragma unroll
            for (int u = 0; u < RPRE; ++u) o = fmaf(u < a.nsplit ? w_s[g][u] : 0.f, rv[i][u], o);   // same order as attn_merge_kernel
            for (int s2 = RPRE; s2 < a.nsplit; ++s2) o = fmaf(w_s[g][s2], recs[(size_t)s2 * RS + 2 + d], o);
            attn[e] = o * inv_l[g];
        }
    }
    __syncthreads();
    tls.phase(2);
    float xf[NWT];
#pragma unroll
    for (int i = 0; i < NWT / 4; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(attn + lr * NWT + 4 * i);
        xf[4 * i] = u.x; xf[4 * i + 1] = u.y; xf[4 * i + 2] = u.z; xf[4 * i + 3] = u.w;
    }
    float* outp = a.opart + ((size_t)b * a.hkv + kvh) * a.H;
    for (int p = 0; p < npass; ++p) {
        const int row = r0 + p * RPP + rip;
        uint4 w;
        float sc = 1.f;
        if (p < PRE) {
            w = pre[p < PRE ? p : 0];
            if constexpr (FP8) sc = psc[p < PRE ? p : 0];
        } else {
            w = load_nt16(wbase + (size_t)row * row_bytes);
            if constexpr (FP8) sc = to_f(a.w_o_scale[(size_t)(row >> 7) * (ldw >> 7) + (col0 >> 7)]);
        }
        float wf[NWT];
        if constexpr (FP8) WTraits<fp8e4m3>::decode(w, wf);
        else WTraits<bf16>::decode(w, wf);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NWT; ++j) acc = fmaf(wf[j], xf[j], acc);
        acc = group_sum<LPW>(acc * sc);      // scale per lane: a row slice of G*D columns may span several 128-column scale blocks
        if (lr == 0) outp[row] = acc;
    }
    tls.end();
}

// --------------------------------------------------------------------------------------------
// Prefill helpers (bf16 activations, fp32 residual stream)
// --------------------------------------------------------------------------------------------
__global__ void embed_rows_kernel(const bf16* embed, const int32_t* tokens, float* h, int H) {
    const int s = blockIdx.x;
    const bf16* row = embed + (size_t)tokens[s] * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) h[(size_t)s * H + i] = to_f(row[i]);
}

// h32[s] += sum of the split-K slabs of the projection that precedes this norm (if any), written back;
// x_bf16[s] = rmsnorm(h32[s]) * gamma.  One 256-thread workgroup per row, 4 elements per thread per trip;
// the slab loads are unconditional (clamped slab index, masked add) so they share one memory round trip.
// With q8 != nullptr (fp8-activation prefill, H % 128 == 0) the row leaves as e4m3 codes + one fp32 scale per 128
// columns instead of bf16: the values quantised are the bf16-rounded ones, so this is bit-identical to
// rmsnorm -> quantize_rows_kernel without the second pass over the activations.
template <int NS>   // slab loads issued per trip (>= nslabs): 4 on the packed path, 16 covers every split count of wsgemm
__global__ __launch_bounds__(256) void rmsnorm_f32_bf16_kernel(float* h, const bf16* gamma, bf16* out, int rows, int H,
                                                               float eps, const float* slabs, int nslabs,
                                                               uint8_t* q8 = nullptr, float* q8s = nullptr) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    float* hr = h + (size_t)row * H;
    constexpr int MAXT = 4;                     // H <= 4096 handled in registers
    float4 v[MAXT];
    uint2 gm[MAXT];                             // gamma requested with the row: not a second round trip after the reduction
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int i = (threadIdx.x + t * 256) * 4;
        gm[t] = *reinterpret_cast<const uint2*>(gamma + min(i, H - 4));
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int i = (threadIdx.x + t * 256) * 4;
        if (i < H) {                            // block-uniform per t when H % 1024 == 0; otherwise per-lane tail
            float4 acc = *reinterpret_cast<const float4*>(hr + i);
            if (nslabs > 0) {
                float4 p[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    p[s] = *reinterpret_cast<const float4*>(slabs + ((size_t)min(s, nslabs - 1) * rows + row) * H + i);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float w = s < nslabs ? 1.f : 0.f;
                    acc.x = fmaf(w, p[s].x, acc.x); acc.y = fmaf(w, p[s].y, acc.y);
                    acc.z = fmaf(w, p[s].z, acc.z); acc.w = fmaf(w, p[s].w, acc.w);
                }
                *reinterpret_cast<float4*>(hr + i) = acc;
            }
            v[t] = acc;
            ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        }
    }
    ss = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(ss / H + eps);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int i = (threadIdx.x + t * 256) * 4;
        if (i < H) {
            const uint2 g = gm[t];
            const float g0 = __uint_as_float(g.x << 16), g1 = __uint_as_float(g.x & 0xFFFF0000u);
            const float g2 = __uint_as_float(g.y << 16), g3 = __uint_as_float(g.y & 0xFFFF0000u);
            uint2 o;
            o.x = pack_bf16x2(v[t].x * inv * g0, v[t].y * inv * g1);
            o.y = pack_bf16x2(v[t].z * inv * g2, v[t].w * inv * g3);
            if (q8) {   // 32 lanes x 4 columns = one 128-column scale block
                const float f0 = __uint_as_float(o.x << 16), f1 = __uint_as_float(o.x & 0xFFFF0000u);
                const float f2 = __uint_as_float(o.y << 16), f3 = __uint_as_float(o.y & 0xFFFF0000u);
                float amax = fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3)));
                amax = group16_max(amax);
                amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
                *reinterpret_cast<uint32_t*>(q8 + (size_t)row * H + i) = pack_fp8x4(f0 / sc, f1 / sc, f2 / sc, f3 / sc);
                if ((threadIdx.x & 31) == 0) q8s[(size_t)row * (H >> 7) + (i >> 7)] = sc;
            } else {
                *reinterpret_cast<uint2*>(out + (size_t)row * H + i) = o;
            }
        }
    }
}

// Per (token s, head slot hh) of qkv[n][(Hq+2Hkv)*D] bf16: q heads -> norm+rope in place;
// k heads -> norm+rope -> cache row; v heads -> cache row.  One lane-group of D/8 lanes per vector.
template <int D>
__global__ __launch_bounds__(256) void qknorm_rope_kvwrite_kernel(bf16* qkv, const bf16* q_gamma, const bf16* k_gamma,
                                                                  float eps, const float* rope_cos,
                                                                  const float* rope_sin, bf16* kcache, bf16* vcache,
                                                                  int n, int hq, int hkv, int max_seq, int start_pos,
                                                                  const float* slabs, int nslabs) {
    constexpr int LPR = D / 8, HALF = D / 2, VPB = 256 / LPR;
    const int nslots = hq + 2 * hkv;
    const long long vec = (long long)blockIdx.x * VPB + threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const bool live = vec < (long long)n * nslots;
    const long long vv = live ? vec : 0;
    const int s = (int)(vv / nslots), hh = (int)(vv % nslots);
    bf16* src = qkv + (size_t)s * nslots * D + (size_t)hh * D + sub * 8;
    float x[8];
    Vec<bf16> raw;
    if (nslabs > 0) {
        // the projection arrived as split-K fp32 partials: sum them and round to bf16, as the projection's
        // own bf16 store would have
        const size_t off = (size_t)s * nslots * D + (size_t)hh * D + sub * 8, stride = (size_t)n * nslots * D;
        float4 a0 = *reinterpret_cast<const float4*>(slabs + off), a1 = *reinterpret_cast<const float4*>(slabs + off + 4);
        for (int k = 1; k < nslabs; ++k) {
            const float4 b0 = *reinterpret_cast<const float4*>(slabs + k * stride + off);
            const float4 b1 = *reinterpret_cast<const float4*>(slabs + k * stride + off + 4);
            a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
            a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
        }
        const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        raw.from_float(f);
    } else {
        raw.load(src);
    }
    raw.to_float(x);
    const int pos = start_pos + s;
    const bool is_q = hh < hq, is_k = !is_q && hh < hq + hkv;
    if (is_q || is_k) {
        const bf16* gamma = is_q ? q_gamma : k_gamma;
        if (gamma) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(x[j], x[j], ss);
            ss = group_sum<LPR>(ss);
            const float inv = 1.0f / sqrtf(ss / D + eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] * inv * to_f(gamma[sub * 8 + j]);
        }
        const bool lo = sub < LPR / 2;
        const float* cs = rope_cos + (size_t)min(pos, max_seq - 1) * HALF;
        const float* sn = rope_sin + (size_t)min(pos, max_seq - 1) * HALF;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float other = xor_half<LPR>(x[j]);
            const int dd = (sub * 8 + j) % HALF;
            o[j] = lo ? (x[j] * cs[dd] - other * sn[dd]) : (x[j] * cs[dd] + other * sn[dd]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = o[j];
    }
    if (!live) return;
    Vec<bf16> ov;
    ov.from_float(x);
    if (is_q) {
        ov.store(src);
    } else if (pos < max_seq) {
        const int kvh = is_k ? hh - hq : hh - hq - hkv;
        bf16* dst = (is_k ? kcache : vcache) + ((size_t)kvh * max_seq + pos) * D + sub * 8;
        ov.store(dst);
    }
}

// act[s][i] = silu(gu[s][i]) * gu[s][I+i]   (bf16 in/out, fp32 math)
// With nslabs > 0 the gate_up projection arrives as split-K fp32 partials [nslabs][n][2I] (summed, rounded to bf16).
// q8 != nullptr (I % 128 == 0): e4m3 codes + per-(row, 128 columns) scales of the bf16-rounded result instead of bf16
__global__ void swiglu_rows_kernel(const bf16* gu, bf16* act, int n, int I, const float* slabs, int nslabs, uint8_t* q8 = nullptr,
                                   float* q8s = nullptr) {
    const size_t total = (size_t)n * I / 8;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += stride) {
        const size_t s = t / (I / 8), c = t % (I / 8);
        Vec<bf16> g, u;
        if (nslabs > 0) {
            const size_t og = s * 2 * I + c * 8, ou = og + I, sst = (size_t)n * 2 * I;
            float sg[8], su[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { sg[j] = 0.f; su[j] = 0.f; }
            for (int k = 0; k < nslabs; ++k) {
                const float4 g0 = *reinterpret_cast<const float4*>(slabs + k * sst + og), g1 = *reinterpret_cast<const float4*>(slabs + k * sst + og + 4);
                const float4 u0 = *reinterpret_cast<const float4*>(slabs + k * sst + ou), u1 = *reinterpret_cast<const float4*>(slabs + k * sst + ou + 4);
                sg[0] += g0.x; sg[1] += g0.y; sg[2] += g0.z; sg[3] += g0.w; sg[4] += g1.x; sg[5] += g1.y; sg[6] += g1.z; sg[7] += g1.w;
                su[0] += u0.x; su[1] += u0.y; su[2] += u0.z; su[3] += u0.w; su[4] += u1.x; su[5] += u1.y; su[6] += u1.z; su[7] += u1.w;
            }
            g.from_float(sg);
            u.from_float(su);
        } else {
            g.load(gu + s * 2 * I + c * 8);
            u.load(gu + s * 2 * I + I + c * 8);
        }
        float gf[8], uf[8];
        g.to_float(gf);
        u.to_float(uf);
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[j] = gf[j] / (1.0f + __expf(-gf[j])) * uf[j];
        g.from_float(gf);
        if (q8) {   // 16 lanes x 8 columns = one scale block; total and stride are multiples of 16, so groups stay whole
            g.to_float(gf);
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(gf[j]));
            amax = group16_max(amax);
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            uint2 o;
            o.x = pack_fp8x4(gf[0] / sc, gf[1] / sc, gf[2] / sc, gf[3] / sc);
            o.y = pack_fp8x4(gf[4] / sc, gf[5] / sc, gf[6] / sc, gf[7] / sc);
            *reinterpret_cast<uint2*>(q8 + s * I + c * 8) = o;
            if ((c & 15) == 0) q8s[s * (I >> 7) + (c >> 4)] = sc;
        } else {
            g.store(act + s * I + c * 8);
        }
    }
}

__global__ void bf16_rows_to_f32_kernel(const bf16* in, float* out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = to_f(in[i]);
}

// --------------------------------------------------------------------------------------------
thread_local Probe* g_probe = nullptr;   // measurement hooks: engine_common.hip.h

struct Engine {
    pgk_model_config_t cfg;
    const bf16 *embed, *lm_head, *final_norm;
    std::vector<pgk_layer_weights_t> layers;
    int nsplit = 1, lm_blocks = 1, lm_cap = 1, log_cap = 4096;
    bool batched_mfma = true;   // chunks of 3 and 5..16 sequences use engine_batched.hip.h (PGK_BATCHED_MFMA=0: GEMV kernels only, =2: from 3 up)
    int batched_min = 5, batched_max = 64;   // PGK_BATCHED_MAX=16: chunks of at most 16 sequences (one weight pass per chunk), the A/B switch of the tiled kernels
    int cu_count = 256;
    bool short_path = true;     // contexts <= SHORT_CTX take the whole-context attention kernels (PGK_FUSED_ATTN=0: the split-KV sequence at every context)
    int pos_hi = -1;            // host-side upper bound of the largest position the next step sees (-1: unknown); selects the sequence, see step_is_short
    // in-graph stochastic sampling (pgk_engine_set_sampling): temperature <= 0 keeps greedy argmax
    float sample_temperature = 0.f, sample_top_p = 1.f;
    int sample_top_k = 0, u_cap = 0, u_alloc_rows = 0;   // u_cap: rows in use (ring length); u_alloc_rows: rows allocated
    float* u_ring = nullptr;       // [u_cap][max_batch] uniforms, row = step counter % u_cap
    void* sample_scratch = nullptr;   // top-k candidate keys (ops_sampling.hip), sized for max_batch rows
    size_t sample_scratch_cap = 0;
    int32_t* sampled = nullptr;    // [max_batch]
    bool fused_attn = false;   // one sequence at short context: attn + o_proj in one kernel (bf16 W_o, shapes that tile)
    bool attn_mfma = false;    // ... with Q.K^T and P.V on the matrix pipe from LDS-staged K/V (head_dim 128; PGK_ATTN_MFMA=0: the dot2 kernels); also the whole-context
                               // batch attention while its workgroups (96 KB of LDS: one per CU) fit one round - batch x Hkv <= CUs (beyond: attn_decode_kernel, batch 64 1.258 vs 1.316 ms)
    bool merged_oproj = false; // long contexts / fp8 W_o, one or two sequences: split-KV merge + o_proj in one kernel (PGK_MERGED_OPROJ=0: merge kernel + GEMV)
    int moproj_rows = 32;
    int oproj_rows = 32;       // W_o rows per workgroup on the fused path
    // device state
    bf16 *kcache = nullptr, *vcache = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr, *cur_cos = nullptr, *cur_sin = nullptr;
    int32_t *tokens = nullptr, *positions = nullptr, *token_log = nullptr, *step_counter = nullptr;
    bf16 *act16 = nullptr, *attnv16 = nullptr;   // batched MFMA path: bf16 hand-off of SwiGLU output and attention output
    bf16* x16 = nullptr;                         // 17..64 sequences: the next RMSNorm's input rows in bf16 (layer 0: normalised by norm_rows_bf16; then un-normalised, written by o_proj / down)
    float* ss_part = nullptr;                    // ... and its statistic: per-workgroup sums of squares [64][1024]
    float *h = nullptr, *h2 = nullptr, *qkv = nullptr, *part = nullptr, *opart = nullptr, *attnv = nullptr, *act = nullptr, *logits = nullptr,
          *amax_val = nullptr;
    int* amax_idx = nullptr;
    unsigned long long* clk_log = nullptr;
    size_t kv_bytes = 0, ws_bytes = 0;
    // fragment-major copies of the layer weights for prompts of <= 128 tokens (ops_pkgemm.hip); PGK_PACKED_PREFILL=0: none
    struct PackedLayer { bf16 *qkv = nullptr, *o = nullptr, *gate_up = nullptr, *down = nullptr; };
    std::vector<PackedLayer> packed;
    bool packed_ok = false;     // the skinny-GEMM kernels of ops_pkgemm.hip can use the copy (their shape limits)
    bool packed_have = false;   // the copy exists (w8a16 engines: also for shapes beyond those kernels - the long-prompt GEMMs read it)
    size_t packed_bytes = 0;
    bool packed_resid = false;      // o_proj / down_proj without K split, carrying the next RMSNorm (pkgemm_resid_nt); PGK_PACKED_RESID=0: split-K slabs + norm launches
    float* pk_ss = nullptr;         // its sum-of-squares table [128][PK_SS_LD]
    bf16* packed_lm = nullptr;      // fragment-major lm_head for the batched (3..64 sequences) lm_head kernels; PGK_PACKED_LMHEAD=0: none
    float* dec_slabs = nullptr;     // 17..64 sequences on the packed kernels: split-K slabs of o_proj / down_proj [splits][M][H] (PGK_PACKED_DECODE=0: engine_batched kernels)
    bool packed_decode = false;
    // prefill workspace (grown on demand, outside capture)
    void* pf = nullptr;
    size_t pf_bytes = 0;
    int32_t* pf_tokens = nullptr;
    int pf_tokens_cap = 0;
    // captured step: [0] the long-context launch sequence, [1] the short-context one (when the engine has both)
    // captured steps: tier 0 = the short-context sequence (span 0); the others = the split-KV sequence with its slices cut for
    // contexts up to `span` positions (1024, 2048, ... and the cache length).  pgk_engine_replay picks per step (pick_tier).
    struct Tier { int span = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int launches = 0; };
    std::vector<Tier> tiers;
    int graph_batch = 0;
    int step_span = 0;          // the split-KV slicing of the step being enqueued (launch_attn); 0: the whole cache
    int launches_per_step = 0;
    std::vector<void*> allocs;

    size_t kv_layer_elems() const { return (size_t)cfg.max_batch * cfg.num_kv_heads * cfg.max_seq_len * cfg.head_dim; }
    int qkv_dim() const { return (cfg.num_heads + 2 * cfg.num_kv_heads) * cfg.head_dim; }
};

static pgk_status dev_alloc(Engine* e, void** p, size_t bytes, size_t* acct) {
    pgk_status r = pgk_malloc(p, bytes);
    if (r != PGK_OK) return r;
    e->allocs.push_back(*p);
    if (acct) *acct += bytes;
    return PGK_OK;
}


template <class WT, class XT, int M, int R, int PRO, int EPI, int C>
static pgk_status launch_fused_c(const FusedArgs& a, int n_out, hipStream_t st, int force_grid) {
    constexpr int OUT_PER_TRIP = (EPI == EPI_SWIGLU) ? R / 2 : R;
    const size_t lds = (size_t)M * a.K * sizeof(XT);
    PGK_REQUIRE(lds <= 156 * 1024, "engine: %d activation rows of K=%d do not fit LDS", M, a.K);
    auto kfn = &fused_gemv_kernel<WT, XT, M, R, PRO, EPI, C>;
    static bool attr_done = false;
    if (lds > 48 * 1024 && !attr_done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        attr_done = true;
    }
    int grid = force_grid ? force_grid : ceil_div(n_out, OUT_PER_TRIP * 4);
    if (grid > 1024) grid = 1024;
    const float* x = (PRO == PRO_PLAIN) ? a.xin : a.h;
    const float* aux = (PRO == PRO_NORM_SUM) ? a.part : a.res;
    const int naux = (PRO == PRO_NORM_SUM) ? a.nsplit : a.ld_out;
    PGK_CHECK_HIP(launch_k(kfn, dim3(grid), dim3(256), lds, st, a.w, a.wscale, x, a.gamma, aux, a.N, naux, a));
    return PGK_OK;
}

// Pick the preload depth C = K / (64 * NW) when the row is short enough to sit in registers.
template <class WT, class XT, int M, int R, int PRO, int EPI>
static pgk_status launch_fused(const FusedArgs& a, int n_out, hipStream_t st, int force_grid = 0) {
    constexpr int NW = WTraits<WT>::NW;
    const int c = (a.K % (64 * NW) == 0) ? a.K / (64 * NW) : 0;
    constexpr int BUDGET = 12 / R;  // R*C*4 preload VGPRs <= 48
    if constexpr (1 <= BUDGET) { if (c == 1) return launch_fused_c<WT, XT, M, R, PRO, EPI, 1>(a, n_out, st, force_grid); }
    if constexpr (2 <= BUDGET) { if (c == 2) return launch_fused_c<WT, XT, M, R, PRO, EPI, 2>(a, n_out, st, force_grid); }
    if constexpr (3 <= BUDGET) { if (c == 3) return launch_fused_c<WT, XT, M, R, PRO, EPI, 3>(a, n_out, st, force_grid); }
    if constexpr (4 <= BUDGET) { if (c == 4) return launch_fused_c<WT, XT, M, R, PRO, EPI, 4>(a, n_out, st, force_grid); }
    if constexpr (6 <= BUDGET) { if (c == 6) return launch_fused_c<WT, XT, M, R, PRO, EPI, 6>(a, n_out, st, force_grid); }
    return launch_fused_c<WT, XT, M, R, PRO, EPI, 0>(a, n_out, st, force_grid);
}

// rows-per-wave heuristic: enough workgroups to cover 256 CUs even for the N = hidden projections
template <class WT, class XT, int M, int PRO, int EPI>
static pgk_status launch_fused_auto(const FusedArgs& a, int n_out, hipStream_t st) {
    if constexpr (EPI == EPI_SWIGLU) {
        // One sequence, mid-sized gate/up (Qwen3-0.6B: 3072 pairs): 3 pairs per wave = 256 workgroups, one per CU.  Every
        // workgroup's prologue re-reads h and the 8 o_proj partial vectors (36 KB from L2); with 768 two-pair workgroups that
        // was 108 KB per CU through the texture addresser against 49 KB of weights (gate/up span 4.32 -> 3.69 us, step 0.607
        // -> 0.592 ms; 4 pairs per wave = 384 workgroups: 4.28 us).  Needs the 6 rows x C chunks to fit the preload budget.
        if constexpr (M == 1) {
            constexpr int NW = WTraits<WT>::NW;
            if (n_out >= 2048 && n_out < 4096 && a.K % (64 * NW) == 0 && a.K / (64 * NW) <= 2) return launch_fused<WT, XT, M, 6, PRO, EPI>(a, n_out, st);
        }
        if (n_out >= 4096) return launch_fused<WT, XT, M, 4, PRO, EPI>(a, n_out, st);
        return launch_fused<WT, XT, M, 2, PRO, EPI>(a, n_out, st);
    } else {
        if (n_out >= 4096) return launch_fused<WT, XT, M, 4, PRO, EPI>(a, n_out, st);
        if (n_out >= 2048) return launch_fused<WT, XT, M, 2, PRO, EPI>(a, n_out, st);
        return launch_fused<WT, XT, M, 1, PRO, EPI>(a, n_out, st);
    }
}

// attention launches per layer for a GQA group of G query heads per kv head (chunks of 4 / 2 / 1 beyond the instantiated sizes)
static int gqa_chunks(int G) {
    if (G == 1 || G == 2 || G == 4) return 1;
    int n = 0;
    for (int off = 0; off < G; ++n) off += (G - off >= 4) ? 4 : ((G - off >= 2) ? 2 : 1);
    return n;
}

template <int G, bool OPROJ = true>
static hipError_t launch_attn_mfma(dim3 grid, hipStream_t st, const AttnArgs& a) {
    constexpr int lds = 2 * AM_CHUNK * 256;
    static bool attr = false;
    if (!attr) {
        const hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_oproj_mfma_kernel<G, OPROJ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (he != hipSuccess) return he;
        attr = true;
    }
    return launch_k(attn_oproj_mfma_kernel<G, OPROJ>, grid, dim3(256), lds, st, a);
}

// `fused`: attention + o_proj partials in one kernel (one sequence at short context); `direct`: one workgroup per
// (sequence, kv head) walks the whole (short) context and writes the normalised output - no merge launch; otherwise
// split-KV slices, then `merged` (merge + o_proj partials in one launch) or the merge kernel.
template <int D>
static pgk_status launch_attn(Engine* e, int layer, int b0, int m, bool fused, bool direct, hipStream_t st, bool direct_bf16 = false,
                              bool merged = false) {
    const auto& c = e->cfg;
    const auto& L = e->layers[layer];
    const int G = c.num_heads / c.num_kv_heads;
    const size_t lofs = (size_t)layer * e->kv_layer_elems() + (size_t)b0 * c.num_kv_heads * c.max_seq_len * c.head_dim;
    AttnArgs a{};
    a.qkv = e->qkv + (size_t)b0 * e->qkv_dim();
    a.qkv_ld = e->qkv_dim();
    a.q_gamma = c.use_qk_norm ? (const bf16*)L.q_norm : nullptr;
    a.k_gamma = c.use_qk_norm ? (const bf16*)L.k_norm : nullptr;
    a.eps = c.norm_eps;
    a.rope_cos = e->cur_cos + (size_t)b0 * (D / 2); a.rope_sin = e->cur_sin + (size_t)b0 * (D / 2);
    a.kcache = e->kcache + lofs; a.vcache = e->vcache + lofs;
    a.positions = e->positions + b0;
    a.hq = c.num_heads; a.hkv = c.num_kv_heads; a.max_seq = c.max_seq_len;
    const int span = (e->step_span > 0 && e->step_span < c.max_seq_len) ? e->step_span : c.max_seq_len;
    a.span = span;
    a.scale = 1.0f / sqrtf((float)D);
    a.part = e->part + (size_t)b0 * c.num_heads * e->nsplit * (D + 2);
    // KV slices per (sequence, kv head): as many as fit ONE wave of workgroups over the chip (a 257th workgroup waits
    // for a free CU and adds a tail: 33 slices x 8 heads measured 5 % slower than 32 at context 2048), never more than
    // the workspace was sized for (e->nsplit: ~64 positions per slice at the full cache length)
    // (batches stream enough KV bytes to want two workgroups per CU: measured 22.1 vs 24.6 us at 8 x 2048 positions)
    int ns = e->cu_count * (m >= 4 ? 2 : 1) / (c.num_kv_heads * m);
    const int ns_cap = ceil_div(span, 64) < e->nsplit ? ceil_div(span, 64) : e->nsplit;     // ~64 positions per slice at least, by the TIER (not the cache: a tier slices alike on every cache)
    ns = ns < 1 ? 1 : (ns > ns_cap ? ns_cap : ns);
    // slices are cut by absolute position in whole position-group steps: launch only as many as the step's context tier needs
    // (the tier, not the cache length: with slices of a 4096-row cache a context of 400 kept 3 of 27 slices busy - 0.671 ms per
    // step against 0.623 on a 1024-row cache)
    a.nsplit = ceil_div(span, decode_chunk_len(span, ns, 4 * (64 / (D / 8))));
    a.w_o = (const bf16*)L.w_o; a.w_o_scale = (const bf16*)L.s_o; a.H = c.hidden_size; a.rows_per_block = e->oproj_rows;
    a.opart = e->opart ? e->opart + (size_t)b0 * c.num_kv_heads * c.hidden_size : nullptr;
    if (direct) {
        a.nsplit = 1;
        a.attn_direct = e->attnv + (size_t)b0 * c.num_heads * D;
        if (direct_bf16) a.attn_direct16 = e->attnv16 + (size_t)b0 * c.num_heads * D;
    }
    dim3 grid = fused ? dim3((c.hidden_size / e->oproj_rows) * c.num_kv_heads, 1, m) : dim3(a.nsplit, c.num_kv_heads, m);
    hipError_t he = hipSuccess;
    a.g_total = G;
    a.g_off = 0;
    if (G != 1 && G != 2 && G != 4) {
        // any other group size (Qwen2.5-7B: 28 / 4 = 7): chunks of 4, 2 and 1 query heads per kv head, one launch each -
        // every chunk re-reads the kv head's K/V rows, which is what the reference's GQA-expanded cache costs for ALL heads
        PGK_REQUIRE(!fused, "engine: fused attention needs a GQA group of 1, 2 or 4");
        for (int off = 0; off < G;) {
            const int gc = (G - off >= 4) ? 4 : ((G - off >= 2) ? 2 : 1);
            a.g_off = off;
            if (gc == 4) he = direct ? launch_k(attn_decode_kernel<D, 4, true>, grid, dim3(256), 0, st, a) : launch_k(attn_decode_kernel<D, 4, false>, grid, dim3(256), 0, st, a);
            else if (gc == 2) he = direct ? launch_k(attn_decode_kernel<D, 2, true>, grid, dim3(256), 0, st, a) : launch_k(attn_decode_kernel<D, 2, false>, grid, dim3(256), 0, st, a);
            else he = direct ? launch_k(attn_decode_kernel<D, 1, true>, grid, dim3(256), 0, st, a) : launch_k(attn_decode_kernel<D, 1, false>, grid, dim3(256), 0, st, a);
            PGK_CHECK_HIP(he);
            off += gc;
        }
    } else {
#define PGK_ATTN(GG)                                                               \
    case GG:                                                                       \
        if (fused && D == 128 && e->attn_mfma) he = launch_attn_mfma<GG>(grid, st, a);              \
        else if (fused) he = launch_k(attn_oproj_kernel<D, GG>, grid, dim3(256), 0, st, a);         \
        else if (direct && D == 128 && e->attn_mfma && m * (int)c.num_kv_heads <= e->cu_count) he = launch_attn_mfma<GG, false>(dim3(c.num_kv_heads, 1, m), st, a); \
        else if (direct) he = launch_k(attn_decode_kernel<D, GG, true>, grid, dim3(256), 0, st, a); \
        else he = launch_k(attn_decode_kernel<D, GG, false>, grid, dim3(256), 0, st, a);            \
        break;
    switch (G) {
        PGK_ATTN(1) PGK_ATTN(2) PGK_ATTN(4)
        default: break;
    }
#undef PGK_ATTN
    PGK_CHECK_HIP(he);
    }
    if (merged) {
        // split-KV merge + o_proj partial products in one launch (attn_merge_oproj_kernel)
        PGK_REQUIRE(!fused && !direct && (G == 1 || G == 2 || G == 4), "engine: merged o_proj on an unsupported attention path");
        mark(KC_OPROJ);
        a.rows_per_block = e->moproj_rows;
        const dim3 g2((c.hidden_size / e->moproj_rows) * c.num_kv_heads, 1, m);
        const bool f8 = c.weight_format != 0;
#define PGK_MO(GG)                                                                                             \
    case GG:                                                                                                   \
        he = f8 ? launch_k(attn_merge_oproj_kernel<D, GG, true>, g2, dim3(256), 0, st, a)                       \
                : launch_k(attn_merge_oproj_kernel<D, GG, false>, g2, dim3(256), 0, st, a);                     \
        break;
        switch (G) { PGK_MO(1) PGK_MO(2) PGK_MO(4) default: break; }
#undef PGK_MO
        PGK_CHECK_HIP(he);
    } else if (!fused && !direct) {
        PGK_CHECK_HIP(launch_k(attn_merge_kernel<D>, dim3(c.num_heads, m), dim3(D), 0, st, (const float*)a.part,
                               e->attnv + (size_t)b0 * c.num_heads * D, (int)c.num_heads, (int)a.nsplit));
    }
    return PGK_OK;
}

// One decode step for sequences [b0, b0+M); `last` = this is the step's last chunk (bumps the step counter)
pgk_status sample_rows_ring(const float* logits, int rows, int vocab, float temperature, int top_k, float top_p, const float* u_ring,
                            int u_cap, int u_stride, const int32_t* step_counter, int32_t* out, void* scratch, hipStream_t st);
size_t sample_scratch_bytes(int rows, int vocab, int top_k, float top_p);

// one draw per sequence of the chunk from the fp32 logits the lm_head kernel just wrote; the uniform numbers come from
// the device ring row (step counter % u_cap), so a captured graph replays with fresh randomness the host queued up
static pgk_status engine_sample(Engine* e, int b0, int M, hipStream_t st) {
    return sample_rows_ring(e->logits + (size_t)b0 * e->cfg.vocab_size, M, e->cfg.vocab_size, e->sample_temperature, e->sample_top_k,
                            e->sample_top_p, e->u_ring + b0, e->u_cap, e->cfg.max_batch, e->step_counter, e->sampled + b0, e->sample_scratch, st);
}

// One decode step for sequences [b0, b0+M); `last` = this is the step's last chunk (bumps the step counter).
// `short_ctx`: every sequence of the step has at most SHORT_CTX positions - a single sequence then takes the fused
// attention + o_proj kernel (4 L + 2 launches); otherwise split-KV slices + merge/o_proj (5 L + 2).  Both are correct at
// any context: the choice follows the context of the step, not the capacity of the cache (pgk_engine_replay).
template <class WT, class XT, int M>
static pgk_status decode_chunk(Engine* e, int b0, bool last, hipStream_t st, int* launches, bool short_ctx) {
    const auto& c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, D = c.head_dim, QD = c.num_heads * D, NQKV = e->qkv_dim();
    float* h = e->h + (size_t)b0 * H;
    float* h2 = e->h2 + (size_t)b0 * H;
    // fused attention+o_proj recomputes a KV head's attention in every row-slice workgroup: right for one
    // sequence at short context, wasteful for a batch.
    // (two sequences ran the fused kernel too until round 3: its 96-KB workgroups are one per CU, so 2 x 256 of them took two
    // rounds - 9.4 us per layer against 3.7 + 2.4 for whole-context attention + an o_proj GEMV: 0.80 -> 0.66 ms per step)
    const bool fused = e->fused_attn && short_ctx && M == 1;
    const bool direct = !fused && short_ctx && M >= 2;
    // long contexts (or fp8 W_o): split-KV slices, then merge + o_proj partials in one launch; the gate/up prologue adds them
    const bool merged = !fused && !direct && M <= 2 && e->merged_oproj;
    const bool partials = fused || merged;
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& L = e->layers[l];
        FusedArgs a{};
        // 1. qkv = Wqkv . rmsnorm(h)
        mark(KC_NORM_QKV);
        a.w = L.w_qkv; a.wscale = (const bf16*)L.s_qkv; a.N = NQKV; a.K = H;
        a.h = h; a.gamma = (const bf16*)L.attn_norm; a.eps = c.norm_eps;
        a.out = e->qkv + (size_t)b0 * NQKV; a.ld_out = NQKV;
        if (pgk_status r = launch_fused_auto<WT, XT, M, PRO_NORM, EPI_STORE>(a, NQKV, st)) return r;
        mark(KC_ATTN);
        // 2. attention (QK-norm, RoPE, KV write fused; on the fused path also the o_proj partial products)
        if (D == 128) { if (pgk_status r = launch_attn<128>(e, l, b0, M, fused, direct, st, false, merged)) return r; }
        else { if (pgk_status r = launch_attn<64>(e, l, b0, M, fused, direct, st, false, merged)) return r; }
        const float* mlp_in = h;
        if (merged) *launches += 1;          // the merge + o_proj launch
        if (!partials) {
            mark(KC_OPROJ);
            // 3. h += Wo . attn   (attn = merged split-KV records, written by attn_merge_kernel)
            a = FusedArgs{};
            a.w = L.w_o; a.wscale = (const bf16*)L.s_o; a.N = H; a.K = QD;
            a.xin = e->attnv + (size_t)b0 * QD;
            a.res = h; a.out = h; a.ld_out = H;
            if (pgk_status r = launch_fused_auto<WT, XT, M, PRO_PLAIN, EPI_RESID>(a, H, st)) return r;
            *launches += (direct ? 1 : 2) + gqa_chunks(c.num_heads / c.num_kv_heads) - 1;   // o_proj (+ the merge kernel unless attention normalised in place)
        }
        // 4. act = silu(Wg x) * (Wu x), x = rmsnorm(h [+ sum of o_proj partials])
        mark(KC_GATEUP);
        a = FusedArgs{};
        a.w = L.w_gate_up; a.wscale = (const bf16*)L.s_gate_up; a.N = I; a.K = H;
        a.h = h; a.gamma = (const bf16*)L.mlp_norm; a.eps = c.norm_eps;
        a.out = e->act + (size_t)b0 * I; a.ld_out = I;
        bool done_gateup = false;
        if constexpr (M <= 2) {   // per-kv-head o_proj partials only ever exist for one or two sequences per chunk
            if (partials) {
                a.part = e->opart + (size_t)b0 * c.num_kv_heads * H; a.nsplit = c.num_kv_heads; a.h_out = h2;
                if (pgk_status r = launch_fused_auto<WT, XT, M, PRO_NORM_SUM, EPI_SWIGLU>(a, I, st)) return r;
                mlp_in = h2;
                done_gateup = true;
            }
        }
        if (!done_gateup) {
            if (pgk_status r = launch_fused_auto<WT, XT, M, PRO_NORM, EPI_SWIGLU>(a, I, st)) return r;
        }
        // 5. h = mlp_in + Wd . act
        mark(KC_DOWN);
        a = FusedArgs{};
        a.w = L.w_down; a.wscale = (const bf16*)L.s_down; a.N = H; a.K = I;
        a.xin = e->act + (size_t)b0 * I;
        a.res = mlp_in; a.out = h; a.ld_out = H;
        if (pgk_status r = launch_fused_auto<WT, XT, M, PRO_PLAIN, EPI_RESID>(a, H, st)) return r;
        *launches += 4;
    }
    // logits = E . rmsnorm(h)  (lm_head stays bf16 even when the linears are fp8)
    mark(KC_LMHEAD);
    FusedArgs a{};
    a.w = e->lm_head; a.N = c.vocab_size; a.K = H;
    a.h = h; a.gamma = e->final_norm; a.eps = c.norm_eps;
    a.out = e->logits + (size_t)b0 * c.vocab_size; a.ld_out = c.vocab_size;
    a.amax_val = e->amax_val + (size_t)b0 * e->lm_blocks; a.amax_idx = e->amax_idx + (size_t)b0 * e->lm_blocks;
    if (pgk_status r = launch_fused<bf16, XT, M, 4, PRO_NORM, EPI_LOGITS>(a, c.vocab_size, st, e->lm_blocks)) return r;
    mark(KC_ARGMAX);
    const int32_t* sampled = nullptr;
    if (e->sample_temperature > 0.f) {
        if (pgk_status r = engine_sample(e, b0, M, st)) return r;
        sampled = e->sampled + b0;
        *launches += 1;
    }
    PGK_CHECK_HIP(launch_k(finalize_kernel, dim3(M), dim3(256), 0, st, e->amax_val + (size_t)b0 * e->lm_blocks, e->amax_idx + (size_t)b0 * e->lm_blocks,
                                       e->lm_blocks, e->tokens + b0, e->positions + b0, e->token_log + b0, e->step_counter,
                                       e->cfg.max_batch, e->log_cap, e->embed, h, H, last ? 1 : 0, b0 == 0 ? e->clk_log : nullptr,
                                       e->rope_cos, e->rope_sin, e->cur_cos + (size_t)b0 * (D / 2), e->cur_sin + (size_t)b0 * (D / 2),
                                       D / 2, c.max_seq_len, M, sampled));
    *launches += 2;
    return PGK_OK;
}

// 3..64 sequences per chunk: every projection on the MFMA kernels of engine_batched.hip.  Up to 16 sequences the cost of
// a projection does not depend on M and RMSNorm is fused into the consumer's prologue (5 L + 2 launches at short
// context); 17..64 sequences run the M-tiled kernels - each weight byte is still read ONCE per step - on rows that one
// small launch per norm has already normalised to bf16 (7 L + 3 launches).  Attention is per sequence either way.
template <class WT>
static pgk_status decode_chunk_batched(Engine* e, int b0, int M, bool last, hipStream_t st, int* launches, bool short_ctx) {
    const auto& c = e->cfg;
    constexpr bool FP8 = std::is_same<WT, fp8e4m3>::value;
    const int H = c.hidden_size, I = c.intermediate_size, D = c.head_dim, QD = c.num_heads * D, NQKV = e->qkv_dim();
    float* h = e->h + (size_t)b0 * H;
    bf16* x16 = e->x16 + (size_t)b0 * H;
    const bool tiled = M > 16;
    const bool direct = short_ctx && M >= 3;
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& L = e->layers[l];
        FusedArgs a{};
        mark(KC_NORM_QKV);
        a.w = L.w_qkv; a.wscale = (const bf16*)L.s_qkv; a.N = NQKV; a.K = H;
        a.h = h; a.gamma = (const bf16*)L.attn_norm; a.eps = c.norm_eps;
        a.out = e->qkv + (size_t)b0 * NQKV; a.ld_out = NQKV;
        if (tiled) {
            if (pgk_status r = norm_rows_bf16(h, a.gamma, x16, M, H, c.norm_eps, st)) return r;
            *launches += 1;
            a.xin16 = x16;
        }
        // 16-row workgroups (N / 16 >= 192) stream the fragment-major copy where the engine holds one: one coalesced KiB per
        // load instead of 64 separate 16-byte pieces of the row-major matrix (the batched lm_head's gain)
        if (!FP8 && e->packed_ok) a.wp = e->packed[l].qkv;
        if (pgk_status r = batched_proj(FP8, tiled ? PRO_PLAIN : PRO_NORM, EPI_STORE, a, M, st)) return r;
        mark(KC_ATTN);
        if (D == 128) { if (pgk_status r = launch_attn<128>(e, l, b0, M, false, direct, st, true)) return r; }
        else { if (pgk_status r = launch_attn<64>(e, l, b0, M, false, direct, st, true)) return r; }
        mark(KC_OPROJ);
        a = FusedArgs{};
        a.w = L.w_o; a.wscale = (const bf16*)L.s_o; a.N = H; a.K = QD;
        a.xin = e->attnv + (size_t)b0 * QD;
        if (direct) a.xin16 = e->attnv16 + (size_t)b0 * QD;   // the whole-context attention kernel wrote bf16
        a.res = h; a.out = h; a.ld_out = H;
        if (tiled && !direct) {   // long contexts: the merge kernel leaves fp32 rows; the tiled kernels read bf16 fragments
            if (pgk_status r = norm_rows_bf16(a.xin, nullptr, e->attnv16 + (size_t)b0 * QD, M, QD, 0.f, st)) return r;
            a.xin16 = e->attnv16 + (size_t)b0 * QD;
            *launches += 1;
        }
        if (pgk_status r = batched_proj(FP8, PRO_PLAIN, EPI_RESID, a, M, st)) return r;
        mark(KC_GATEUP);
        a = FusedArgs{};
        a.w = L.w_gate_up; a.wscale = (const bf16*)L.s_gate_up; a.N = I; a.K = H;
        a.h = h; a.gamma = (const bf16*)L.mlp_norm; a.eps = c.norm_eps;
        a.out = e->act + (size_t)b0 * I; a.ld_out = I;
        a.out16 = e->act16 + (size_t)b0 * I;        // SiLU(g) * u leaves as bf16: down_proj rounds it to bf16 anyway
        if (tiled) {
            if (pgk_status r = norm_rows_bf16(h, a.gamma, x16, M, H, c.norm_eps, st)) return r;
            *launches += 1;
            a.xin16 = x16;
        }
        if (!FP8 && e->packed_ok) a.wp = e->packed[l].gate_up;
        if (pgk_status r = batched_proj(FP8, tiled ? PRO_PLAIN : PRO_NORM, EPI_SWIGLU, a, M, st)) return r;
        mark(KC_DOWN);
        a = FusedArgs{};
        a.w = L.w_down; a.wscale = (const bf16*)L.s_down; a.N = H; a.K = I;
        a.xin = e->act + (size_t)b0 * I;
        a.xin16 = e->act16 + (size_t)b0 * I;
        a.res = h; a.out = h; a.ld_out = H;
        if (pgk_status r = batched_proj(FP8, PRO_PLAIN, EPI_RESID, a, M, st)) return r;
        *launches += (direct ? 5 : 6) + gqa_chunks(c.num_heads / c.num_kv_heads) - 1;
    }
    const int nblk = ceil_div(c.vocab_size, 16) < 2048 ? ceil_div(c.vocab_size, 16) : 2048;
    mark(KC_LMHEAD);
    FusedArgs a{};
    a.w = e->lm_head; a.N = c.vocab_size; a.K = H;
    a.h = h; a.gamma = e->final_norm; a.eps = c.norm_eps;
    a.out = e->logits + (size_t)b0 * c.vocab_size; a.ld_out = c.vocab_size;
    a.amax_val = e->amax_val + (size_t)b0 * e->lm_cap; a.amax_idx = e->amax_idx + (size_t)b0 * e->lm_cap;
    a.wp = e->packed_lm;
    if (tiled) {
        // the final norm keeps its launch: 2048 lm_head workgroups re-deriving the row statistic would read 128 MB of partials
        if (pgk_status r = norm_rows_bf16(h, a.gamma, x16, M, H, c.norm_eps, st)) return r;
        *launches += 1;
        a.xin16 = x16;
    }
    if (pgk_status r = batched_proj(false, tiled ? PRO_PLAIN : PRO_NORM, EPI_LOGITS, a, M, st, nblk)) return r;
    mark(KC_ARGMAX);
    const int32_t* sampled = nullptr;
    if (e->sample_temperature > 0.f) {
        if (pgk_status r = engine_sample(e, b0, M, st)) return r;
        sampled = e->sampled + b0;
        *launches += 1;
    }
    PGK_CHECK_HIP(launch_k(finalize_kernel, dim3(M), dim3(256), 0, st, a.amax_val, a.amax_idx, nblk, e->tokens + b0, e->positions + b0, e->token_log + b0, e->step_counter,
                                       e->cfg.max_batch, e->log_cap, e->embed, h, H, last ? 1 : 0, b0 == 0 ? e->clk_log : nullptr,
                                       e->rope_cos, e->rope_sin, e->cur_cos + (size_t)b0 * (D / 2), e->cur_sin + (size_t)b0 * (D / 2),
                                       D / 2, c.max_seq_len, M, sampled));
    *launches += 2;
    return PGK_OK;
}

// 17..64 sequences, bf16 layers: the step on the fragment-major weight copy (ops_pkgemm.hip) - the prefill's kernels with
// one row per SEQUENCE.  Seven launches per layer as on the tiled path, but the projections stream coalesced 1 KiB
// fragments with the activation block in LDS by DMA, SwiGLU sits in the gate_up epilogue, and the N = hidden projections
// are split along K over 256 workgroups with the next RMSNorm summing their slabs (rmsnorm_f32_bf16_kernel, as in
// pgk_engine_prefill).  Attention (per-sequence positions, new-token norm / RoPE / cache write) is the batch kernel.
static pgk_status decode_chunk_packed(Engine* e, int b0, int M, bool last, hipStream_t st, int* launches, bool short_ctx) {
    const auto& c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, D = c.head_dim, QD = c.num_heads * D, NQKV = e->qkv_dim();
    float* h = e->h + (size_t)b0 * H;
    bf16* x16 = e->x16 + (size_t)b0 * H;
    const bool direct = short_ctx && M >= 3;
    const int s_o = pkgemm_pick_splits(M, H, QD), s_d = pkgemm_pick_splits(M, H, I);
    int pending = 0;
    auto norm = [&](const bf16* gamma) -> pgk_status {
        mark(KC_NORM_QKV);
        // (plain launches, like the projections of this path: the per-launch probe and the in-kernel timeline cover the
        // kernels that take a timeline pointer - attention, lm_head, finalize)
        if (pending <= 4) rmsnorm_f32_bf16_kernel<4><<<M, 256, 0, st>>>(h, gamma, x16, M, H, c.norm_eps, e->dec_slabs, pending);
        else rmsnorm_f32_bf16_kernel<16><<<M, 256, 0, st>>>(h, gamma, x16, M, H, c.norm_eps, e->dec_slabs, pending);
        pending = 0;
        PGK_LAUNCH_CHECK();
        return PGK_OK;
    };
    const bool carried = e->packed_resid;      // o_proj / down_proj carry the next RMSNorm: 5 launches per layer instead of 7
    int ss_n = 0;
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& L = e->layers[l];
        const auto& P = e->packed[l];
        PkArgs nrm{};
        if (carried && l > 0) { nrm.ss_in = e->pk_ss; nrm.ss_n = ss_n; nrm.ss_eps = c.norm_eps; }
        else if (pgk_status r = norm((const bf16*)L.attn_norm)) return r;
        if (pgk_status r = pkgemm_nt(x16, H, P.qkv, e->qkv + (size_t)b0 * NQKV, NQKV, PK_EPI_SLAB, 1, M, NQKV, H, &nrm, st)) return r;
        mark(KC_ATTN);
        if (D == 128) { if (pgk_status r = launch_attn<128>(e, l, b0, M, false, direct, st, true)) return r; }
        else { if (pgk_status r = launch_attn<64>(e, l, b0, M, false, direct, st, true)) return r; }
        mark(KC_OPROJ);
        bf16* attn16 = e->attnv16 + (size_t)b0 * QD;
        if (!direct) {   // long contexts: the merge kernel leaves fp32 rows
            if (pgk_status r = norm_rows_bf16(e->attnv + (size_t)b0 * QD, nullptr, attn16, M, QD, 0.f, st)) return r;
            *launches += 1;
        }
        bf16* act16 = e->act16 + (size_t)b0 * I;
        if (carried) {
            if (pgk_status r = pkgemm_resid_nt(attn16, QD, P.o, h, M, H, QD, (const bf16*)L.mlp_norm, x16, e->pk_ss, &ss_n, st)) return r;
            mark(KC_GATEUP);
            PkArgs gn{};
            gn.ss_in = e->pk_ss; gn.ss_n = ss_n; gn.ss_eps = c.norm_eps;
            if (pgk_status r = pkgemm_nt(x16, H, P.gate_up, act16, I, PK_EPI_SWIGLU, 1, M, 2 * I, H, &gn, st)) return r;
            mark(KC_DOWN);
            const bf16* gnext = l + 1 < c.num_layers ? (const bf16*)e->layers[l + 1].attn_norm : nullptr;
            if (pgk_status r = pkgemm_resid_nt(act16, I, P.down, h, M, H, I, gnext, x16, e->pk_ss, &ss_n, st)) return r;
            *launches += (direct ? 5 : 6) + (l == 0 ? 1 : 0) + gqa_chunks(c.num_heads / c.num_kv_heads) - 1;
            continue;
        }
        if (pgk_status r = pkgemm_nt(attn16, QD, P.o, e->dec_slabs, H, PK_EPI_SLAB, s_o, M, H, QD, nullptr, st)) return r;
        pending = s_o;
        if (pgk_status r = norm((const bf16*)L.mlp_norm)) return r;
        mark(KC_GATEUP);
        if (pgk_status r = pkgemm_nt(x16, H, P.gate_up, act16, I, PK_EPI_SWIGLU, 1, M, 2 * I, H, nullptr, st)) return r;
        mark(KC_DOWN);
        if (pgk_status r = pkgemm_nt(act16, I, P.down, e->dec_slabs, H, PK_EPI_SLAB, s_d, M, H, I, nullptr, st)) return r;
        pending = s_d;
        *launches += (direct ? 7 : 8) + gqa_chunks(c.num_heads / c.num_kv_heads) - 1;
    }
    const int nblk = ceil_div(c.vocab_size, 16) < 2048 ? ceil_div(c.vocab_size, 16) : 2048;
    if (pgk_status r = norm(e->final_norm)) return r;     // also folds the last down_proj's slabs into the residual stream
    mark(KC_LMHEAD);
    FusedArgs a{};
    a.w = e->lm_head; a.N = c.vocab_size; a.K = H;
    a.h = h; a.gamma = e->final_norm; a.eps = c.norm_eps;
    a.out = e->logits + (size_t)b0 * c.vocab_size; a.ld_out = c.vocab_size;
    a.amax_val = e->amax_val + (size_t)b0 * e->lm_cap; a.amax_idx = e->amax_idx + (size_t)b0 * e->lm_cap;
    a.xin16 = x16;
    a.wp = e->packed_lm;                                  // the M-tiled lm_head streams the fragment-major copy too
    // (lm_head on a packed copy with an argmax epilogue was built and measured: 1.342 ms per step against 1.331 at 64
    // sequences - its 39 MB of fp32 logits stores, not the weight loads, are what the row-major kernel's 95 us are made of)
    if (pgk_status r = batched_proj(false, PRO_PLAIN, EPI_LOGITS, a, M, st, nblk)) return r;
    mark(KC_ARGMAX);
    const int32_t* sampled = nullptr;
    if (e->sample_temperature > 0.f) {
        if (pgk_status r = engine_sample(e, b0, M, st)) return r;
        sampled = e->sampled + b0;
        *launches += 1;
    }
    PGK_CHECK_HIP(launch_k(finalize_kernel, dim3(M), dim3(256), 0, st, a.amax_val, a.amax_idx, nblk, e->tokens + b0, e->positions + b0, e->token_log + b0, e->step_counter,
                                       e->cfg.max_batch, e->log_cap, e->embed, h, H, last ? 1 : 0, b0 == 0 ? e->clk_log : nullptr,
                                       e->rope_cos, e->rope_sin, e->cur_cos + (size_t)b0 * (D / 2), e->cur_sin + (size_t)b0 * (D / 2),
                                       D / 2, c.max_seq_len, M, sampled));
    *launches += 3;
    return PGK_OK;
}

template <class WT>
static pgk_status decode_step_impl(Engine* e, int batch, hipStream_t st, int* launches, bool short_ctx) {
    int b0 = 0;
    while (b0 < batch) {
        const int rem = batch - b0;
        pgk_status r;
        // the MFMA projections cost the same for 3 as for 16 sequences (~1.0-1.2 ms per step on Qwen3-0.6B); the GEMV
        // kernels exist for M = 1, 2, 4, 8 only, so 3 / 5 / 6 / 7 sequences would take two or three weight passes there
        // (measured: 7 sequences 2.46 ms against 1.05).  GEMV stays for exactly 1, 2 and 4 (0.70 / 0.93 / 0.82 ms).
        const bool mfma_ok = e->batched_mfma && (rem >= e->batched_min || (e->batched_min == 5 && rem == 3));
        if (mfma_ok) {
            const int m = rem > e->batched_max ? e->batched_max : rem;
            if (m > 16 && e->packed_decode) r = decode_chunk_packed(e, b0, m, rem == m, st, launches, short_ctx);
            else r = decode_chunk_batched<WT>(e, b0, m, rem == m, st, launches, short_ctx);
            b0 += m;
        }
        else if (rem >= 8) { r = decode_chunk<WT, bf16, 8>(e, b0, rem == 8, st, launches, short_ctx); b0 += 8; }
        else if (rem >= 4) { r = decode_chunk<WT, bf16, 4>(e, b0, rem == 4, st, launches, short_ctx); b0 += 4; }
        else if (rem >= 2) { r = decode_chunk<WT, float, 2>(e, b0, rem == 2, st, launches, short_ctx); b0 += 2; }
        else { r = decode_chunk<WT, float, 1>(e, b0, true, st, launches, short_ctx); b0 += 1; }
        if (r != PGK_OK) return r;
    }
    return PGK_OK;
}

// short_ctx: the step's launch sequence for contexts <= SHORT_CTX (ignored - long sequence - when the engine has no such path)
static pgk_status decode_step(Engine* e, int batch, hipStream_t st, int* launches, bool short_ctx) {
    short_ctx = short_ctx && e->short_path;
    if (e->cfg.weight_format != 0) return decode_step_impl<fp8e4m3>(e, batch, st, launches, short_ctx);
    return decode_step_impl<bf16>(e, batch, st, launches, short_ctx);
}

// Which launch sequence the NEXT step takes: the short-context one while the host-side bound on the step's largest
// position (set by pgk_engine_set_state, advanced by every step this library enqueues) stays below SHORT_CTX.  The bound
// is a speed hint only - both sequences are correct at any context - so a caller that rewrites the device-resident
// positions behind the library's back loses speed, never correctness; an unknown bound selects the long sequence.
static int short_limit(int batch) { return batch == 1 ? SHORT_CTX_B1 : SHORT_CTX; }
static bool step_is_short(const Engine* e, int batch) { return e->short_path && e->pos_hi >= 0 && e->pos_hi + 1 <= short_limit(batch); }

static void drop_graphs(Engine* e) {
    for (auto& t : e->tiers) {
        if (t.exec) (void)hipGraphExecDestroy(t.exec);
        if (t.graph) (void)hipGraphDestroy(t.graph);
    }
    e->tiers.clear();
    e->graph_batch = 0;
}

// The split-KV slicing a step at the host-side position bound needs: the smallest of 1024, 2048, ... that covers the context, capped at
// the cache length (an unknown bound: the cache length).
static int span_for(const Engine* e) {
    const int cap = e->cfg.max_seq_len;
    if (e->pos_hi < 0) return cap;
    int span = 1024;
    while (span < e->pos_hi + 1 && span < cap) span *= 2;
    return span < cap ? span : cap;
}

// The captured step the next replay takes: the short-context sequence while the position bound allows it (and it was captured),
// otherwise the split-KV tier whose slices cover the context; both kinds are correct at any context they cover.
static size_t pick_tier(const Engine* e) {
    const bool has_short = !e->tiers.empty() && e->tiers[0].span == 0;
    if (has_short && (e->tiers.size() == 1 || step_is_short(e, e->graph_batch))) return 0;
    const int want = span_for(e);
    for (size_t i = has_short ? 1 : 0; i < e->tiers.size(); ++i)
        if (e->tiers[i].span >= want) return i;
    return e->tiers.size() - 1;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_engine_create(const pgk_model_config_t* cfg, const void* embed, const void* lm_head,
                             const void* final_norm, const pgk_layer_weights_t* layers, pgk_engine* out) {
    PGK_REQUIRE(cfg && embed && final_norm && layers && out, "pgk_engine_create: null argument");
    const auto& c = *cfg;
    PGK_REQUIRE(c.head_dim == 128 || c.head_dim == 64, "pgk_engine_create: head_dim %d not in {64,128}", c.head_dim);
    PGK_REQUIRE(c.num_heads % c.num_kv_heads == 0, "pgk_engine_create: Hq %d %% Hkv %d", c.num_heads, c.num_kv_heads);
    const int G = c.num_heads / c.num_kv_heads;
    PGK_REQUIRE(G >= 1, "pgk_engine_create: GQA group %d", G);
    PGK_REQUIRE(c.hidden_size % 16 == 0 && c.intermediate_size % 16 == 0, "pgk_engine_create: sizes must be multiples of 16");
    PGK_REQUIRE(c.weight_format == 0 || (c.hidden_size % 128 == 0 && c.intermediate_size % 128 == 0),
                "pgk_engine_create: fp8 weights need 128-multiple dims");
    PGK_REQUIRE(c.max_batch >= 1 && c.max_seq_len >= 1 && c.num_layers >= 1, "pgk_engine_create: bad sizes");
    Engine* e = new Engine();
    e->cfg = c;
    e->embed = (const bf16*)embed;
    e->lm_head = (const bf16*)(lm_head ? lm_head : embed);
    e->final_norm = (const bf16*)final_norm;
    e->layers.assign(layers, layers + c.num_layers);
    int nsplit = (c.max_seq_len + 63) / 64;   // ~64 cached positions per workgroup: one KV batch per wave
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            e->cu_count = prop.multiProcessorCount;
    }
    e->nsplit = nsplit < 1 ? 1 : (nsplit > 64 ? 64 : nsplit);
    e->lm_blocks = 1024;
    // fused attention + o_proj: short contexts, bf16 W_o, and a row slicing that tiles the workgroup
    {
        const int gd = G * c.head_dim, rpp = 256 / (gd / 8);
        int rows = c.hidden_size / 32;
        while (rows > 4 * rpp && rows % 2 == 0) rows /= 2;   // <= 4 preloaded passes per workgroup
        while (rows < rpp) rows *= 2;
        const bool tiles = rows % rpp == 0 && c.hidden_size % rows == 0 && gd / 8 <= 64;
        // PGK_FUSED_ATTN=0: no short-context launch sequence at all (the split-KV sequence at every context; A/B and tests)
        const char* env = getenv("PGK_FUSED_ATTN");
        e->short_path = env ? atoi(env) != 0 : true;
        e->fused_attn = tiles && c.weight_format == 0 && (G == 1 || G == 2 || G == 4);
        e->oproj_rows = rows;
        {
            const char* am = getenv("PGK_ATTN_MFMA");
            e->attn_mfma = c.head_dim == 128 && !(am && atoi(am) == 0);
        }
        // merged o_proj (long contexts, fp8 W_o): same slicing rule with 16 codes per lane for fp8
        const int nwt = c.weight_format != 0 ? 16 : 8, lpw = gd / nwt, rpp2 = lpw > 0 ? 256 / lpw : 256;
        int rows2 = c.hidden_size / 32;
        while (rows2 > 4 * rpp2 && rows2 % 2 == 0) rows2 /= 2;
        while (rows2 < rpp2) rows2 *= 2;
        const char* emo = getenv("PGK_MERGED_OPROJ");
        e->merged_oproj = !(emo && atoi(emo) == 0) && (G == 1 || G == 2 || G == 4) && lpw >= 8 && lpw <= 64 && 256 % lpw == 0 &&
                          rows2 % rpp2 == 0 && c.hidden_size % rows2 == 0 && (c.weight_format == 0 || (gd % 128 == 0 || 128 % gd == 0));
        e->moproj_rows = rows2;
    }
    const int B = c.max_batch, H = c.hidden_size, D = c.head_dim;
    pgk_status r = PGK_OK;
    auto A = [&](void** p, size_t bytes, size_t* acct) { if (r == PGK_OK) r = dev_alloc(e, p, bytes, acct); };
    const size_t kvb = (size_t)c.num_layers * e->kv_layer_elems() * sizeof(bf16);
    A((void**)&e->kcache, kvb, &e->kv_bytes);
    A((void**)&e->vcache, kvb, &e->kv_bytes);
    A((void**)&e->rope_cos, (size_t)c.max_seq_len * (D / 2) * 4, &e->ws_bytes);
    A((void**)&e->rope_sin, (size_t)c.max_seq_len * (D / 2) * 4, &e->ws_bytes);
    A((void**)&e->cur_cos, (size_t)B * (D / 2) * 4, &e->ws_bytes);
    A((void**)&e->cur_sin, (size_t)B * (D / 2) * 4, &e->ws_bytes);
    A((void**)&e->tokens, (size_t)B * 4, &e->ws_bytes);
    A((void**)&e->positions, (size_t)B * 4, &e->ws_bytes);
    A((void**)&e->token_log, (size_t)e->log_cap * B * 4, &e->ws_bytes);
    A((void**)&e->step_counter, 16, &e->ws_bytes);
    A((void**)&e->h, (size_t)B * H * 4, &e->ws_bytes);
    A((void**)&e->h2, (size_t)B * H * 4, &e->ws_bytes);
    A((void**)&e->qkv, (size_t)B * e->qkv_dim() * 4, &e->ws_bytes);
    A((void**)&e->part, (size_t)B * c.num_heads * e->nsplit * (D + 2) * 4, &e->ws_bytes);
    A((void**)&e->opart, (size_t)B * c.num_kv_heads * H * 4, &e->ws_bytes);
    A((void**)&e->act16, (size_t)B * c.intermediate_size * 2, &e->ws_bytes);
    A((void**)&e->attnv16, (size_t)B * c.num_heads * c.head_dim * 2, &e->ws_bytes);
    A((void**)&e->x16, (size_t)B * H * 2, &e->ws_bytes);
    A((void**)&e->ss_part, (size_t)64 * 1024 * 4, &e->ws_bytes);
    A((void**)&e->attnv, (size_t)B * c.num_heads * D * 4, &e->ws_bytes);
    A((void**)&e->act, (size_t)B * c.intermediate_size * 4, &e->ws_bytes);
    A((void**)&e->logits, (size_t)B * c.vocab_size * 4, &e->ws_bytes);
    e->lm_cap = e->lm_blocks > ceil_div(c.vocab_size, 16) ? e->lm_blocks : ceil_div(c.vocab_size, 16);   // per-sequence argmax partial slots
    {
        const char* ev = getenv("PGK_BATCHED_MFMA");
        e->batched_min = (ev && atoi(ev) == 2) ? 3 : 5;   // 2: from 3 up
        // every projection's K must suit the MFMA decode kernels: K = 128 S with S in {8, 16, 24, 32} (activation
        // fragments in registers) or an LDS image of K x 16 bf16 that fits (K <= 4096); Llama-3-8B's down_proj
        // (K = 14336) does neither, so such models decode batches in GEMV chunks of 8 / 4 / 2 / 1
        auto k_ok = [](int K) { return K % 128 == 0 && K <= 4096; };
        e->batched_mfma = !(ev && atoi(ev) == 0) && k_ok(c.hidden_size) && k_ok(c.intermediate_size) && k_ok(c.num_heads * c.head_dim);
        // 17..64 sequences in one weight pass (batched_mt_kernel) need K = 128 S with S instantiated; otherwise chunks of 16
        auto k_tiled = [](int K) { const int s = K / 128; return K % 128 == 0 && (s == 2 || s == 4 || s == 8 || s == 16 || s == 24 || s == 32); };
        e->batched_max = (k_tiled(c.hidden_size) && k_tiled(c.intermediate_size) && k_tiled(c.num_heads * c.head_dim)) ? 64 : 16;
        if (const char* em = getenv("PGK_BATCHED_MAX")) { const int v = atoi(em); if (v >= 16 && v < e->batched_max) e->batched_max = v; }
    }
    A((void**)&e->amax_val, (size_t)B * e->lm_cap * 4, &e->ws_bytes);
    A((void**)&e->amax_idx, (size_t)B * e->lm_cap * 4, &e->ws_bytes);
    A((void**)&e->clk_log, (size_t)e->log_cap * 16, &e->ws_bytes);
    {
        // Second, fragment-major copy of the bf16 layer weights: what the short-prompt prefill streams (ops_pkgemm.hip).
        // Costs the layers' bytes again; skipped when that is more than a quarter of the device's free memory.
        const int QDp = c.num_heads * D, NQ = e->qkv_dim(), I = c.intermediate_size;
        const char* ep = getenv("PGK_PACKED_PREFILL");
        const size_t per_layer = ((size_t)NQ * H + (size_t)H * QDp + (size_t)2 * I * H + (size_t)H * I) * 2;
        size_t free_b = 0, total_b = 0;
        const bool fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && per_layer * c.num_layers < free_b / 4;
        const bool shapes = pkgemm_shape_ok(NQ, H, false) && pkgemm_shape_ok(H, QDp, true) && pkgemm_shape_ok(2 * I, H, false) && pkgemm_shape_ok(H, I, true) && I % 64 == 0;
        // bf16 layers, or fp8 codes + block scales (w8a16): the copy is bf16 either way (ops_pkgemm.hip)
        const bool f8w = c.weight_format == 1;
        // w8a16 engines keep the copy also where the skinny kernels cannot use it (Llama-3-8B: K = 4096 / 14336 is beyond them):
        // their long-prompt GEMMs read it (packed_have), instead of dequantising every weight again in front of every call
        const bool long_only = f8w && !shapes && NQ % 16 == 0 && H % 64 == 0 && QDp % 64 == 0 && I % 64 == 0;
        if (r == PGK_OK && (c.weight_format == 0 || f8w) && (shapes || long_only) && fits && !(ep && atoi(ep) == 0)) {
            e->packed.resize(c.num_layers);
            hipStream_t st = resolve_stream(nullptr);
            for (int l = 0; l < c.num_layers && r == PGK_OK; ++l) {
                auto& P = e->packed[l];
                const auto& L = e->layers[l];
                A((void**)&P.qkv, (size_t)NQ * H * 2, &e->packed_bytes);
                A((void**)&P.o, (size_t)H * QDp * 2, &e->packed_bytes);
                A((void**)&P.gate_up, (size_t)2 * I * H * 2, &e->packed_bytes);
                A((void**)&P.down, (size_t)H * I * 2, &e->packed_bytes);
                if (r == PGK_OK) r = f8w ? pack_weights_fp8(L.w_qkv, L.s_qkv, P.qkv, NQ, H, st) : pack_weights_bf16(L.w_qkv, P.qkv, NQ, H, st);
                if (r == PGK_OK) r = f8w ? pack_weights_fp8(L.w_o, L.s_o, P.o, H, QDp, st) : pack_weights_bf16(L.w_o, P.o, H, QDp, st);
                if (r == PGK_OK) r = f8w ? pack_weights_fp8(L.w_gate_up, L.s_gate_up, P.gate_up, 2 * I, H, st) : pack_weights_bf16(L.w_gate_up, P.gate_up, 2 * I, H, st);
                if (r == PGK_OK) r = f8w ? pack_weights_fp8(L.w_down, L.s_down, P.down, H, I, st) : pack_weights_bf16(L.w_down, P.down, H, I, st);
            }
            if (r == PGK_OK && hipStreamSynchronize(st) != hipSuccess) r = set_error(PGK_ERR_HIP, "pgk_engine_create: packing the prefill weights failed");
            e->packed_have = r == PGK_OK;
            e->packed_ok = e->packed_have && shapes;
            if (e->packed_ok && pkgemm_resid_ok(H, QDp) && pkgemm_resid_ok(H, I)) {
                A((void**)&e->pk_ss, (size_t)128 * PK_SS_LD * 4, &e->ws_bytes);
                e->packed_resid = r == PGK_OK;
            }
            if (e->packed_ok && c.max_batch >= 3 && c.vocab_size % 16 == 0 && H % 32 == 0) {
                A((void**)&e->packed_lm, (size_t)c.vocab_size * H * 2, &e->packed_bytes);
                if (r == PGK_OK) r = pack_weights_bf16(e->lm_head, e->packed_lm, c.vocab_size, H, st);
                if (r == PGK_OK && hipStreamSynchronize(st) != hipSuccess) r = set_error(PGK_ERR_HIP, "pgk_engine_create: packing the lm_head failed");
            }
            const char* pd = getenv("PGK_PACKED_DECODE");
            if (e->packed_ok && c.max_batch > 16 && !(pd && atoi(pd) == 0)) {
                A((void**)&e->dec_slabs, (size_t)16 * 64 * H * 4, &e->ws_bytes);
                e->packed_decode = r == PGK_OK;
            }
        }
    }
    if (r != PGK_OK) { pgk_engine_destroy(e); return r; }
    // RoPE tables in fp32, same formula as the reference (src/pygpukit/llm/layers/rope.py:13-24):
    // freqs = 1/theta^(2i/D) in fp32, angle = float(t) * freq in fp32, cos/sin of that.
    {
        std::vector<float> hc((size_t)c.max_seq_len * (D / 2)), hs(hc.size());
        for (int i = 0; i < D / 2; ++i) {
            const float expo = (float)(2 * i) / (float)D;
            const float freq = 1.0f / powf(c.rope_theta, expo);
            for (int t = 0; t < c.max_seq_len; ++t) {
                const float ang = (float)t * freq;
                hc[(size_t)t * (D / 2) + i] = (float)cos((double)ang);
                hs[(size_t)t * (D / 2) + i] = (float)sin((double)ang);
            }
        }
        hipStream_t st = resolve_stream(nullptr);
        hipError_t he = hipMemcpyAsync(e->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipMemcpyAsync(e->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->kcache, 0, kvb, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->vcache, 0, kvb, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->tokens, 0, (size_t)B * 4, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->positions, 0, (size_t)B * 4, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->step_counter, 0, 16, st);
        if (he == hipSuccess) he = hipMemsetAsync(e->h, 0, (size_t)B * H * 4, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he != hipSuccess) { pgk_engine_destroy(e); return set_error(PGK_ERR_HIP, "pgk_engine_create: %s", hipGetErrorString(he)); }
    }
    *out = e;
    return PGK_OK;
}

pgk_status pgk_engine_destroy(pgk_engine eh) {
    if (!eh) return PGK_OK;
    Engine* e = (Engine*)eh;
    drop_graphs(e);
    for (void* p : e->allocs) (void)pgk_free(p);
    if (e->pf) (void)pgk_free(e->pf);
    if (e->pf_tokens) (void)pgk_free(e->pf_tokens);
    if (e->u_ring) (void)pgk_free(e->u_ring);
    if (e->sampled) (void)pgk_free(e->sampled);
    if (e->sample_scratch) (void)pgk_free(e->sample_scratch);
    delete e;
    return PGK_OK;
}

pgk_status pgk_engine_bytes(pgk_engine eh, size_t* kv_bytes, size_t* workspace_bytes) {
    PGK_REQUIRE(eh, "pgk_engine_bytes: null engine");
    Engine* e = (Engine*)eh;
    if (kv_bytes) *kv_bytes = e->kv_bytes;
    if (workspace_bytes) *workspace_bytes = e->ws_bytes + e->pf_bytes + e->packed_bytes;
    return PGK_OK;
}

pgk_status pgk_engine_prefill(pgk_engine eh, int seq, const int32_t* h_tokens, int n, int start_pos, void* all_logits,
                              float* h_last_logits, pgk_stream s) {
    PGK_REQUIRE(eh && h_tokens, "pgk_engine_prefill: null argument");
    Engine* e = (Engine*)eh;
    const auto& c = e->cfg;
    PGK_REQUIRE(seq >= 0 && seq < c.max_batch, "pgk_engine_prefill: sequence slot %d outside [0,%d)", seq, c.max_batch);
    PGK_REQUIRE(n >= 1 && start_pos >= 0 && start_pos + n <= c.max_seq_len, "pgk_engine_prefill: positions %d..%d outside cache of %d",
                start_pos, start_pos + n, c.max_seq_len);
    hipStream_t st = resolve_stream(s);
    // (prompts of 129..256 tokens used to run as two chunks of <= 128 through the packed-weight kernels; since the staged 128-tile
    // GEMM and the epilogue fusions of round 3 the long-prompt path is faster at every such length: 144 tokens 1.94 vs 2.38 ms,
    // 256 tokens 2.27 vs 2.62)
    const int H = c.hidden_size, I = c.intermediate_size, D = c.head_dim, QD = c.num_heads * D, NQKV = e->qkv_dim();
    // workspace: h32 [n,H] f32 | x [n,H] | qkv [n,NQKV] | attn [n,QD] | gu [n,2I] | act [n,I]  (bf16)
    // + split-K slabs of the N = hidden projections on the weight-streaming path (n <= 128)
    const bool ws = n <= 128;
    const int s_o = ws ? wsgemm_pick_splits(H, QD, true) : 1, s_d = ws ? wsgemm_pick_splits(H, I, true) : 1;
    const int s_qkv_ws = ws ? wsgemm_pick_splits(NQKV, H, true) : 1, s_qkv = s_qkv_ws, s_gu = ws ? wsgemm_pick_splits(2 * I, H, true) : 1;
    const int maxk = I > QD ? (I > H ? I : H) : (QD > H ? QD : H);
    // packed-weight path (ops_pkgemm.hip): bf16 layers, n <= 128; its own split counts for the N = hidden projections
    const bool pk = ws && e->packed_ok;
    const int pk_so = pk ? pkgemm_pick_splits(n, H, QD) : 1, pk_sd = pk ? pkgemm_pick_splits(n, H, I) : 1;
    const bool pk_heads = pk && D == 128;       // QKV epilogue: per-head norm + RoPE + cache write inside the projection
    // long prompts, bf16 weights: the N = hidden projections as split-K slabs when their 128-tiles do not cover the chip
    // (QKV / gate_up were tried too - their consumers can sum slabs - and measured slightly slower: 3.06 vs 2.99 ms at S = 512)
    // w8a16 engines, long prompts: the staged bf16 GEMMs read the DEQUANTISED fragment-major copy the engine already holds
    // (pack_weights_fp8: bf16(code x scale), the value the reference's w8a16 GEMM multiplies) - same kernels, epilogues and
    // times as a bf16 engine (S = 512: 4.23 -> 2.44 ms) instead of the in-staging-dequant 128-tile kernel / a per-call
    // dequantisation pass in front of the 256-tile kernel
    const bool pkd = !ws && c.weight_format == 1 && e->packed_have && engine_gemm_packed_ok(n, NQKV, H) && engine_gemm_packed_ok(n, H, QD) &&
                     engine_gemm_packed_ok(n, 2 * I, H) && engine_gemm_packed_ok(n, H, I);
    const bool gsplit = !ws && (c.weight_format == 0 || pkd);
    const int g_so = gsplit ? engine_gemm_pick_splits(n, H, QD) : 1, g_sd = gsplit ? engine_gemm_pick_splits(n, H, I) : 1;
    const bool use_slabs = ws || g_so > 1 || g_sd > 1;
    size_t slab_elems = (size_t)(s_o > s_d ? s_o : s_d) * n * H;
    if ((size_t)(g_so > g_sd ? g_so : g_sd) * n * H > slab_elems) slab_elems = (size_t)(g_so > g_sd ? g_so : g_sd) * n * H;
    if (pk && (size_t)(pk_so > pk_sd ? pk_so : pk_sd) * n * H > slab_elems) slab_elems = (size_t)(pk_so > pk_sd ? pk_so : pk_sd) * n * H;
    if (s_qkv > 1 && (size_t)s_qkv * n * NQKV > slab_elems) slab_elems = (size_t)s_qkv * n * NQKV;
    if (s_gu > 1 && (size_t)s_gu * n * 2 * I > slab_elems) slab_elems = (size_t)s_gu * n * 2 * I;
    const size_t need = (size_t)n * H * 4 + ((size_t)n * H + (size_t)n * NQKV + (size_t)n * QD + (size_t)n * 2 * I + (size_t)n * I) * 2 +
                        (use_slabs ? slab_elems * 4 : 0) + 512 + (c.weight_format == 2 ? 2 * ((size_t)n * maxk + (size_t)n * (maxk / 128) * 4 + 512) : 0);
    if (need > e->pf_bytes) {
        if (e->pf) PGK_CHECK_HIP(hipStreamSynchronize(st));
        if (e->pf) pgk_free(e->pf);
        e->pf = nullptr;
        if (pgk_status r = pgk_malloc(&e->pf, need)) return r;
        e->pf_bytes = need;
    }
    if (n > e->pf_tokens_cap) {
        if (e->pf_tokens) { PGK_CHECK_HIP(hipStreamSynchronize(st)); pgk_free(e->pf_tokens); }
        if (pgk_status r = pgk_malloc((void**)&e->pf_tokens, (size_t)n * 4)) return r;
        e->pf_tokens_cap = n;
    }
    for (int i = 0; i < n; ++i)
        PGK_REQUIRE(h_tokens[i] >= 0 && h_tokens[i] < c.vocab_size, "pgk_engine_prefill: token %d out of range", h_tokens[i]);
    PGK_CHECK_HIP(hipMemcpyAsync(e->pf_tokens, h_tokens, (size_t)n * 4, hipMemcpyHostToDevice, st));
    PGK_CHECK_HIP(hipStreamSynchronize(st));  // h_tokens may be pageable: make the copy complete before returning control
    const bool fp8 = c.weight_format != 0;
    const bool fp8act = c.weight_format == 2 && n > 128;   // fp8 x fp8 MFMA projections, activations quantised on the fly
    char* p = (char*)e->pf;
    float* h32 = (float*)p; p += (size_t)n * H * 4;
    bf16* x = (bf16*)p; p += (size_t)n * H * 2;
    bf16* qkv = (bf16*)p; p += (size_t)n * NQKV * 2;
    bf16* attn = (bf16*)p; p += (size_t)n * QD * 2;
    bf16* gu = (bf16*)p; p += (size_t)n * 2 * I * 2;
    bf16* act = (bf16*)p; p += (size_t)n * I * 2;
    float* slabs = (float*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    uint8_t* q8 = (uint8_t*)(((uintptr_t)slabs + (use_slabs ? slab_elems * 4 : 0) + 255) & ~(uintptr_t)255);   // fp8 activations [n][maxk]
    float* q8s = (float*)(q8 + (size_t)n * maxk);                                                          // their scales [n][maxk/128]
    uint8_t* q8b = (uint8_t*)(((uintptr_t)(q8s + (size_t)n * (maxk / 128)) + 255) & ~(uintptr_t)255);      // second pair: the gate / up GEMM's
    float* q8bs = (float*)(q8b + (size_t)n * maxk);                                                        // SwiGLU epilogue writes while q8 is its operand
    int pending = 0;   // split-K slabs of the previous projection still to be added into h32 by the next norm
    // fp8act: RMSNorm and SwiGLU leave their result in q8/q8s themselves (x_in == nullptr); attention output is
    // quantised here (its rows span all heads, a flash workgroup only sees one)
    // (PGK_FUSED_EPILOGUES=0 keeps the separate passes - quantise, SwiGLU: the A/B switch of the bit-identity tests)
    const char* fq_env = getenv("PGK_FUSED_EPILOGUES");
    const bool fuse_epi = !(fq_env && atoi(fq_env) == 0);
    const bool fuse_q = fp8act && H % 128 == 0 && I % 128 == 0 && H <= 4096 && fuse_epi;
    // SwiGLU in the gate / up GEMM's epilogue (256-tile kernels; fp8 x fp8: with the quantisation of its result)
    const bool fuse_sw8 = fuse_q && gemm_fp8_swiglu_ok(n, I, H);
    const bool fuse_sw16 = !fp8act && !ws && fuse_epi && engine_gemm_swiglu_ok(n, I, H, fp8 && !pkd);
    // per-head norm + RoPE + cache write in the QKV GEMM's epilogue (bf16 weights, head_dim 128, 128-tile kernel: tile column = head)
    const bool fuse_heads8 = fuse_q && D == 128 && gemm_fp8_qkv_heads_ok(n, NQKV, H);      // fp8 x fp8: x's codes are already in q8
    const bool fuse_heads = fuse_heads8 || (!ws && !pk && fuse_epi && (c.weight_format == 0 || pkd) && D == 128 && engine_gemm_qkv_heads_ok(n, NQKV, H));
    auto proj_accum = [&](const bf16* x_in, const void* w, const void* sc, int N_, int K_, int splits, const bf16* wp = nullptr) -> pgk_status {
        if (fp8act) {
            if (x_in)
                if (pgk_status r = quantize_fp8_rows_bf16(x_in, q8, q8s, n, K_, st)) return r;
            return gemm_fp8_nt(q8, q8s, (const uint8_t*)w, (const bf16*)sc, h32, true, n, N_, K_, st);
        }
        if (!ws) {
            const bool usep = pkd && wp != nullptr;
            const int gs = gsplit ? engine_gemm_pick_splits(n, N_, K_) : 1;
            if (gs > 1) { pending = gs; return engine_gemm_nt_slabs(x_in, usep ? (const void*)wp : w, slabs, gs, n, N_, K_, st, usep); }
            if (usep) return engine_gemm_nt(x_in, wp, nullptr, false, h32, true, n, N_, K_, st, true);
            return engine_gemm_nt(x_in, w, (const bf16*)sc, fp8, h32, true, n, N_, K_, st);
        }
        if (splits == 1) return wsgemm_nt(x_in, K_, w, (const bf16*)sc, fp8, h32, nullptr, 2, 1, n, N_, K_, st);
        pending = splits;
        return wsgemm_nt(x_in, K_, w, (const bf16*)sc, fp8, slabs, nullptr, 1, splits, n, N_, K_, st);
    };
    // with splits > 1 the result is left as fp32 split-K slabs for the consumer kernel to sum
    auto proj_store = [&](const bf16* x_in, const void* w, const void* sc, bf16* out_, int N_, int K_, int splits, const bf16* wp = nullptr) -> pgk_status {
        if (fp8act) {
            if (x_in)
                if (pgk_status r = quantize_fp8_rows_bf16(x_in, q8, q8s, n, K_, st)) return r;
            return gemm_fp8_nt(q8, q8s, (const uint8_t*)w, (const bf16*)sc, out_, false, n, N_, K_, st);
        }
        if (!ws && pkd && wp != nullptr) return engine_gemm_nt(x_in, wp, nullptr, false, out_, false, n, N_, K_, st, true);
        if (!ws) return engine_gemm_nt(x_in, w, (const bf16*)sc, fp8, out_, false, n, N_, K_, st);
        if (splits > 1) return wsgemm_nt(x_in, K_, w, (const bf16*)sc, fp8, slabs, nullptr, 1, splits, n, N_, K_, st);
        return wsgemm_nt(x_in, K_, w, (const bf16*)sc, fp8, out_, nullptr, 0, 1, n, N_, K_, st);
    };
    auto norm = [&](const bf16* gamma, bool to_fp8 = false) -> pgk_status {
        if (pending <= 4)
            rmsnorm_f32_bf16_kernel<4><<<n, 256, 0, st>>>(h32, gamma, x, n, H, c.norm_eps, slabs, pending, to_fp8 ? q8 : nullptr,
                                                          to_fp8 ? q8s : nullptr);
        else
            rmsnorm_f32_bf16_kernel<16><<<n, 256, 0, st>>>(h32, gamma, x, n, H, c.norm_eps, slabs, pending, to_fp8 ? q8 : nullptr,
                                                           to_fp8 ? q8s : nullptr);
        pending = 0;
        PGK_LAUNCH_CHECK();
        return PGK_OK;
    };
    embed_rows_kernel<<<n, 256, 0, st>>>(e->embed, e->pf_tokens, h32, H);
    PGK_LAUNCH_CHECK();
    const int kv_len = start_pos + n;
    int ss_n = 0;    // partial sums per row in pk_ss (carried norms)
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& L = e->layers[l];
        bf16* kc = e->kcache + (size_t)l * e->kv_layer_elems() + (size_t)seq * c.num_kv_heads * c.max_seq_len * D;
        bf16* vc = e->vcache + (size_t)l * e->kv_layer_elems() + (size_t)seq * c.num_kv_heads * c.max_seq_len * D;
        // packed path with carried norms: layer 0 normalises with a launch; afterwards x holds bf16(h * gamma) and pk_ss the
        // row statistics, both left by the previous layer's down_proj
        const bool carried = pk && e->packed_resid;
        PkArgs nrm{};                                   // how the consumer of x scales its rows (all null: x is normalised)
        if (carried && l > 0) { nrm.ss_in = e->pk_ss; nrm.ss_n = ss_n; nrm.ss_eps = c.norm_eps; }
        else if (pgk_status r = norm((const bf16*)L.attn_norm, fuse_q)) return r;
        if (pk_heads) {
            PkArgs hd = nrm;
            hd.q_gamma = c.use_qk_norm ? (const bf16*)L.q_norm : nullptr;
            hd.k_gamma = c.use_qk_norm ? (const bf16*)L.k_norm : nullptr;
            hd.eps = c.norm_eps; hd.rope_cos = e->rope_cos; hd.rope_sin = e->rope_sin; hd.kcache = kc; hd.vcache = vc;
            hd.hq = c.num_heads; hd.hkv = c.num_kv_heads; hd.max_seq = c.max_seq_len; hd.start_pos = start_pos;
            if (pgk_status r = pkgemm_nt(x, H, e->packed[l].qkv, qkv, NQKV, PK_EPI_QKV, 1, n, NQKV, H, &hd, st)) return r;
        } else if (pk) {
            if (pgk_status r = pkgemm_nt(x, H, e->packed[l].qkv, qkv, NQKV, PK_EPI_BF16, 1, n, NQKV, H, &nrm, st)) return r;
        } else if (fuse_heads) {
            QkvHeadArgs hd{};
            hd.q_gamma = c.use_qk_norm ? (const bf16*)L.q_norm : nullptr;
            hd.k_gamma = c.use_qk_norm ? (const bf16*)L.k_norm : nullptr;
            hd.eps = c.norm_eps; hd.rope_cos = e->rope_cos; hd.rope_sin = e->rope_sin; hd.kcache = kc; hd.vcache = vc;
            hd.hq = c.num_heads; hd.hkv = c.num_kv_heads; hd.max_seq = c.max_seq_len; hd.start_pos = start_pos;
            if (fuse_heads8) {
                if (pgk_status r = gemm_fp8_qkv_heads_nt(q8, q8s, (const uint8_t*)L.w_qkv, (const bf16*)L.s_qkv, qkv, n, NQKV, H, hd, st)) return r;
            } else if (pgk_status r = engine_gemm_qkv_heads_nt(x, pkd ? e->packed[l].qkv : (const bf16*)L.w_qkv, qkv, n, NQKV, H, hd, st, pkd)) return r;
        } else {
            if (pgk_status r = proj_store(fuse_q ? nullptr : x, L.w_qkv, L.s_qkv, qkv, NQKV, H, s_qkv, pkd ? e->packed[l].qkv : nullptr)) return r;
        }
        if (!pk_heads && !fuse_heads) {
            const int s_qkv = pk ? 1 : s_qkv_ws;
            const int nslots = c.num_heads + 2 * c.num_kv_heads;
            const bf16* qg = c.use_qk_norm ? (const bf16*)L.q_norm : nullptr;
            const bf16* kg = c.use_qk_norm ? (const bf16*)L.k_norm : nullptr;
            if (D == 128)
                qknorm_rope_kvwrite_kernel<128><<<ceil_div((long long)n * nslots, 16), 256, 0, st>>>(
                    qkv, qg, kg, c.norm_eps, e->rope_cos, e->rope_sin, kc, vc, n, c.num_heads, c.num_kv_heads, c.max_seq_len, start_pos,
                    slabs, s_qkv > 1 ? s_qkv : 0);
            else
                qknorm_rope_kvwrite_kernel<64><<<ceil_div((long long)n * nslots, 32), 256, 0, st>>>(
                    qkv, qg, kg, c.norm_eps, e->rope_cos, e->rope_sin, kc, vc, n, c.num_heads, c.num_kv_heads, c.max_seq_len, start_pos,
                    slabs, s_qkv > 1 ? s_qkv : 0);
            PGK_LAUNCH_CHECK();
        }
        // fp8 x fp8: a head's 128 output dims are one scale block of the o_proj operand, so the flash kernel quantises them itself
        const bool attn_q8 = fuse_q && D == 128 && n > 128 && sdpa_flash_enabled();
        if (attn_q8) {
            if (pgk_status r = flash_prefill_q8(qkv, kc, vc, q8, q8s, c.num_heads, c.num_kv_heads, n, kv_len, 1.0f / sqrtf((float)D), D, NQKV,
                                                (long long)c.max_seq_len * D, D, st))
                return r;
        } else if (pgk_status r = pgk_sdpa_causal(qkv, kc, vc, attn, c.num_heads, c.num_kv_heads, n, kv_len, D, 0.f, D, NQKV,
                                                  (int64_t)c.max_seq_len * D, D, D, QD, PGK_BF16, st))
            return r;
        // N = hidden projections of the packed path: fp32 split-K slabs summed by the next norm (or h32 += with one split)
        auto pk_accum = [&](const bf16* x_in, const bf16* wp, int K_, int splits) -> pgk_status {
            if (splits == 1) return pkgemm_nt(x_in, K_, wp, h32, H, PK_EPI_ACCUM, 1, n, H, K_, nullptr, st);
            pending = splits;
            return pkgemm_nt(x_in, K_, wp, slabs, H, PK_EPI_SLAB, splits, n, H, K_, nullptr, st);
        };
        if (carried) {
            // o_proj adds the residual itself and leaves bf16(h * gamma_mlp) + row statistics; gate_up scales by 1 / rms;
            // down_proj does the same for the next layer's attention norm: 5 launches per layer
            if (pgk_status r = pkgemm_resid_nt(attn, QD, e->packed[l].o, h32, n, H, QD, (const bf16*)L.mlp_norm, x, e->pk_ss, &ss_n, st)) return r;
            PkArgs gn{};
            gn.ss_in = e->pk_ss; gn.ss_n = ss_n; gn.ss_eps = c.norm_eps;
            if (pgk_status r = pkgemm_nt(x, H, e->packed[l].gate_up, act, I, PK_EPI_SWIGLU, 1, n, 2 * I, H, &gn, st)) return r;
            const bf16* gnext = l + 1 < c.num_layers ? (const bf16*)e->layers[l + 1].attn_norm : nullptr;
            if (pgk_status r = pkgemm_resid_nt(act, I, e->packed[l].down, h32, n, H, I, gnext, x, e->pk_ss, &ss_n, st)) return r;
            continue;
        }
        if (pk) { if (pgk_status r = pk_accum(attn, e->packed[l].o, QD, pk_so)) return r; }
        else if (pgk_status r = proj_accum(attn_q8 ? nullptr : attn, L.w_o, L.s_o, H, QD, s_o, pkd ? e->packed[l].o : nullptr)) return r;
        if (pgk_status r = norm((const bf16*)L.mlp_norm, fuse_q)) return r;
        if (pk) {
            // SwiGLU inside the gate_up projection: the gate tile and its up tile live in the same wave
            if (pgk_status r = pkgemm_nt(x, H, e->packed[l].gate_up, act, I, PK_EPI_SWIGLU, 1, n, 2 * I, H, nullptr, st)) return r;
            if (pgk_status r = pk_accum(act, e->packed[l].down, I, pk_sd)) return r;
            continue;
        }
        if (fuse_sw8) {
            // x's codes in q8 -> act's codes in q8b; the down projection reads q8b
            if (pgk_status r = gemm_fp8_swiglu_nt(q8, q8s, (const uint8_t*)L.w_gate_up, (const bf16*)L.s_gate_up, q8b, q8bs, n, I, H, st)) return r;
            if (pgk_status r = gemm_fp8_nt(q8b, q8bs, (const uint8_t*)L.w_down, (const bf16*)L.s_down, h32, true, n, H, I, st)) return r;
            continue;
        }
        if (fuse_sw16) {
            if (pgk_status r = engine_gemm_swiglu_nt(x, pkd ? (const void*)e->packed[l].gate_up : L.w_gate_up, (const bf16*)L.s_gate_up, fp8 && !pkd, act, n, I, H, st, pkd)) return r;
            if (pgk_status r = proj_accum(act, L.w_down, L.s_down, H, I, s_d, pkd ? e->packed[l].down : nullptr)) return r;
            continue;
        }
        if (pgk_status r = proj_store(fuse_q ? nullptr : x, L.w_gate_up, L.s_gate_up, gu, 2 * I, H, s_gu, pkd ? e->packed[l].gate_up : nullptr)) return r;
        swiglu_rows_kernel<<<ceil_div((long long)n * I / 8, 256) > 2048 ? 2048 : ceil_div((long long)n * I / 8, 256), 256, 0, st>>>(
            gu, act, n, I, slabs, s_gu > 1 ? s_gu : 0, fuse_q ? q8 : nullptr, fuse_q ? q8s : nullptr);
        PGK_LAUNCH_CHECK();
        if (pgk_status r = proj_accum(fuse_q ? nullptr : act, L.w_down, L.s_down, H, I, s_d, pkd ? e->packed[l].down : nullptr)) return r;
    }
    if (pgk_status r = norm(e->final_norm)) return r;
    if (all_logits) {
        if (pgk_status r = engine_gemm_nt(x, e->lm_head, nullptr, false, all_logits, false, n, c.vocab_size, H, st)) return r;
    }
    if (h_last_logits) {
        // last row through the fp32-output GEMV (the decode lm_head kernel with a plain prologue)
        float* xin = h32;  // reuse: widen the last normed row
        bf16_rows_to_f32_kernel<<<4, 256, 0, st>>>(x + (size_t)(n - 1) * H, xin, H);
        PGK_LAUNCH_CHECK();
        FusedArgs a{};
        a.w = e->lm_head; a.N = c.vocab_size; a.K = H; a.xin = xin;
        a.out = e->logits + (size_t)seq * c.vocab_size; a.ld_out = c.vocab_size;
        if (pgk_status r = launch_fused<bf16, float, 1, 4, PRO_PLAIN, EPI_STORE>(a, c.vocab_size, st)) return r;
        PGK_CHECK_HIP(hipMemcpyAsync(h_last_logits, a.out, (size_t)c.vocab_size * 4, hipMemcpyDeviceToHost, st));
        PGK_CHECK_HIP(hipStreamSynchronize(st));
    }
    return PGK_OK;
}

pgk_status pgk_engine_set_state(pgk_engine eh, const int32_t* h_tokens, const int32_t* h_positions, int batch, pgk_stream s) {
    PGK_REQUIRE(eh && h_tokens && h_positions, "pgk_engine_set_state: null argument");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch, "pgk_engine_set_state: batch %d outside [1,%d]", batch, e->cfg.max_batch);
    for (int b = 0; b < batch; ++b) {
        PGK_REQUIRE(h_tokens[b] >= 0 && h_tokens[b] < e->cfg.vocab_size, "pgk_engine_set_state: token %d out of range", h_tokens[b]);
        PGK_REQUIRE(h_positions[b] >= 0 && h_positions[b] < e->cfg.max_seq_len, "pgk_engine_set_state: position %d outside cache of %d",
                    h_positions[b], e->cfg.max_seq_len);
    }
    hipStream_t st = resolve_stream(s);
    e->pos_hi = 0;
    for (int b = 0; b < batch; ++b) e->pos_hi = h_positions[b] > e->pos_hi ? h_positions[b] : e->pos_hi;
    PGK_CHECK_HIP(hipMemcpyAsync(e->tokens, h_tokens, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    PGK_CHECK_HIP(hipMemcpyAsync(e->positions, h_positions, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    // the residual stream enters a step already holding the embeddings of the state tokens
    embed_kernel<<<batch, 256, 0, st>>>(e->embed, e->tokens, e->h, e->cfg.hidden_size, e->positions, e->rope_cos, e->rope_sin,
                                        e->cur_cos, e->cur_sin, e->cfg.head_dim / 2, e->cfg.max_seq_len);
    PGK_LAUNCH_CHECK();
    PGK_CHECK_HIP(hipStreamSynchronize(st));
    return PGK_OK;
}

pgk_status pgk_engine_decode_step(pgk_engine eh, int batch, pgk_stream s) {
    PGK_REQUIRE(eh, "pgk_engine_decode_step: null engine");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch, "pgk_engine_decode_step: batch %d outside [1,%d]", batch, e->cfg.max_batch);
    int launches = 0;
    e->step_span = span_for(e);
    pgk_status r = decode_step(e, batch, resolve_stream(s), &launches, step_is_short(e, batch));
    e->launches_per_step = launches;
    if (e->pos_hi >= 0) ++e->pos_hi;
    return r;
}

pgk_status pgk_engine_profile_step(pgk_engine eh, int batch, int n_iters, float* h_ms_sum, int* h_count, pgk_stream s) {
    PGK_REQUIRE(eh && h_ms_sum && h_count, "pgk_engine_profile_step: null argument");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch && n_iters >= 1, "pgk_engine_profile_step: bad arguments");
    hipStream_t st = resolve_stream(s);
    for (int i = 0; i < KC_COUNT; ++i) { h_ms_sum[i] = 0.f; h_count[i] = 0; }
    // Eager steps whose every launch carries its own start / stop event (hipExtLaunchKernelGGL): the elapsed time of a
    // pair is the dispatch's begin -> end interval, the quantity rocprofv3 --kernel-trace reports - no launch gap, nothing
    // subtracted.
    Probe probe;
    probe.timing = true;
    pgk_status r = PGK_OK;
    for (int it = 0; it < n_iters && r == PGK_OK; ++it) {
        probe.used = 0;
        probe.info.clear();
        g_probe = &probe;
        int launches = 0;
        e->step_span = span_for(e);
        r = decode_step(e, batch, st, &launches, step_is_short(e, batch));
        g_probe = nullptr;
        if (e->pos_hi >= 0) ++e->pos_hi;
        if (r != PGK_OK) break;
        hipError_t he = hipStreamSynchronize(st);
        if (he != hipSuccess) { r = set_error(PGK_ERR_HIP, "pgk_engine_profile_step: %s", hipGetErrorString(he)); break; }
        for (size_t i = 0; i < probe.used; ++i) {
            float ms = 0.f;
            const int cls = probe.cls[i];
            if (cls >= 0 && cls < KC_COUNT && hipEventElapsedTime(&ms, probe.ev[2 * i], probe.ev[2 * i + 1]) == hipSuccess) {
                h_ms_sum[cls] += ms;
                h_count[cls] += 1;
            }
        }
    }
    for (hipEvent_t ev : probe.ev) (void)hipEventDestroy(ev);
    return r;
}

// Timeline of ONE replayed step (diagnostic): the step is captured into a temporary graph with every kernel's `tl` slot
// set, replayed `warm` times and then once more; per launch the host gets
//   h_out[6 i .. 6 i + 5] = { kernel class (KC_*), workgroups, first start, last start, first end, last end }
// with the four times in ticks of the 100 MHz s_memrealtime counter relative to the step's first start.  The engine's
// own captured graph is left alone; the sequence state advances by warm + 1 real steps.
pgk_status pgk_engine_timeline(pgk_engine eh, int batch, int warm, uint64_t* h_out, int max_launches, int* n_launches, pgk_stream s) {
    PGK_REQUIRE(eh && h_out && n_launches && max_launches >= 1 && warm >= 0, "pgk_engine_timeline: bad arguments");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch, "pgk_engine_timeline: batch %d outside [1,%d]", batch, e->cfg.max_batch);
    hipStream_t st = resolve_stream(s);
    const int cap = 16 * e->cfg.num_layers + 64;
    Probe probe;
    probe.tl_cap = cap;
    const size_t bytes = (size_t)cap * TL_MAXWG * 2 * sizeof(unsigned long long);
    if (pgk_status r = pgk_malloc((void**)&probe.tl, bytes)) return r;
    pgk_status r = PGK_OK;
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    hipError_t he = hipMemsetAsync(probe.tl, 0, bytes, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (he == hipSuccess) {
        g_probe = &probe;
        int launches = 0;
        e->step_span = span_for(e);
        r = decode_step(e, batch, st, &launches, step_is_short(e, batch));
        g_probe = nullptr;
        he = hipStreamEndCapture(st, &g);
    }
    if (r == PGK_OK && he == hipSuccess) he = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    for (int i = 0; r == PGK_OK && he == hipSuccess && i <= warm; ++i) he = hipGraphLaunch(ex, st);
    if (e->pos_hi >= 0) e->pos_hi += warm + 1;
    if (r == PGK_OK && he == hipSuccess) he = hipStreamSynchronize(st);
    const int n = (int)probe.info.size() < cap ? (int)probe.info.size() : cap;
    if (r == PGK_OK && he == hipSuccess) {
        std::vector<unsigned long long> host((size_t)n * TL_MAXWG * 2);
        he = hipMemcpy(host.data(), probe.tl, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (he == hipSuccess) {
            unsigned long long origin = ~0ull;
            std::vector<unsigned long long> agg((size_t)n * 4);
            for (int i = 0; i < n; ++i) {
                const int nwg = probe.info[i].nwg < TL_MAXWG ? probe.info[i].nwg : TL_MAXWG;
                unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long a = host[((size_t)i * TL_MAXWG + w) * 2], b = host[((size_t)i * TL_MAXWG + w) * 2 + 1];
                    if (a == 0 && b == 0) continue;   // slot never written
                    s0 = a < s0 ? a : s0; s1 = a > s1 ? a : s1; e0 = b < e0 ? b : e0; e1 = b > e1 ? b : e1;
                }
                agg[4 * i] = s0; agg[4 * i + 1] = s1; agg[4 * i + 2] = e0; agg[4 * i + 3] = e1;
                if (s0 < origin) origin = s0;
            }
#ifdef PGK_PHASE_STAMPS
            {
                double sum[KC_COUNT][8] = {}, cnt[KC_COUNT][8] = {}, dur[KC_COUNT] = {};
                for (int i = 0; i < n; ++i) {
                    const int cls = probe.info[i].cls, nwg = probe.info[i].nwg;
                    if (cls < 0 || cls >= KC_COUNT || nwg > 256) continue;
                    for (int w = 0; w < nwg; ++w) {
                        const unsigned long long t0 = host[((size_t)i * TL_MAXWG + w) * 2], t1 = host[((size_t)i * TL_MAXWG + w) * 2 + 1];
                        dur[cls] += (double)(t1 - t0);
                        for (int k = 0; k < 8; ++k) {
                            const unsigned long long v = host[((size_t)i * TL_MAXWG + 256 + 4 * w) * 2 + k];
                            if (v) { sum[cls][k] += (double)(v - t0); cnt[cls][k] += 1; }
                        }
                    }
                }
                for (int cls = 0; cls < KC_COUNT; ++cls)
                    for (int k = 0; k < 8; ++k)
                        if (cnt[cls][k] > 0) fprintf(stderr, "phase stamps: class %d phase %d at %.2f us after its workgroup's start (n=%.0f)\n", cls, k, sum[cls][k] / cnt[cls][k] / 100.0, cnt[cls][k]);
            }
#endif
            *n_launches = n < max_launches ? n : max_launches;
            for (int i = 0; i < *n_launches; ++i) {
                h_out[6 * i] = (uint64_t)probe.info[i].cls;
                h_out[6 * i + 1] = (uint64_t)probe.info[i].nwg;
                for (int k = 0; k < 4; ++k) h_out[6 * i + 2 + k] = agg[4 * i] == ~0ull ? 0 : agg[4 * i + k] - origin;
            }
        }
    }
    if (ex) (void)hipGraphExecDestroy(ex);
    if (g) (void)hipGraphDestroy(g);
    (void)pgk_free(probe.tl);
    if (r != PGK_OK) return r;
    if (he != hipSuccess) return set_error(PGK_ERR_HIP, "pgk_engine_timeline: %s", hipGetErrorString(he));
    return PGK_OK;
}

pgk_status pgk_engine_capture(pgk_engine eh, int batch, pgk_stream s) {
    PGK_REQUIRE(eh, "pgk_engine_capture: null engine");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch, "pgk_engine_capture: batch %d outside [1,%d]", batch, e->cfg.max_batch);
    hipStream_t st = resolve_stream(s);
    drop_graphs(e);
    // tier 0: the short-context sequence; then the split-KV sequence once per context tier (1024, 2048, ... positions and the
    // cache length) - needed whenever a context can exceed the short limit (or the short sequences are switched off)
    std::vector<int> spans;
    if (e->short_path) spans.push_back(0);
    if (!e->short_path || e->cfg.max_seq_len > short_limit(batch)) {
        for (int sp = 1024; sp < e->cfg.max_seq_len; sp *= 2) spans.push_back(sp);
        spans.push_back(e->cfg.max_seq_len);
    }
    for (int span : spans) {
        PGK_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int launches = 0;
        e->step_span = span;
        pgk_status r = decode_step(e, batch, st, &launches, span == 0);
        hipGraph_t g = nullptr;
        hipError_t he = hipStreamEndCapture(st, &g);
        if (r != PGK_OK) { if (g) (void)hipGraphDestroy(g); drop_graphs(e); return r; }
        if (he != hipSuccess || !g) { drop_graphs(e); return set_error(PGK_ERR_HIP, "pgk_engine_capture: hipStreamEndCapture: %s", hipGetErrorString(he)); }
        Engine::Tier t;
        t.span = span; t.graph = g; t.launches = launches;
        he = hipGraphInstantiate(&t.exec, g, nullptr, nullptr, 0);
        e->tiers.push_back(t);      // owned from here on (drop_graphs)
        if (he != hipSuccess) { drop_graphs(e); return set_error(PGK_ERR_HIP, "pgk_engine_capture: hipGraphInstantiate: %s", hipGetErrorString(he)); }
    }
    e->graph_batch = batch;
    e->launches_per_step = e->tiers[pick_tier(e)].launches;
    return PGK_OK;
}

pgk_status pgk_engine_replay(pgk_engine eh, int n_steps, pgk_stream s) {
    PGK_REQUIRE(eh, "pgk_engine_replay: null engine");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(!e->tiers.empty(), "pgk_engine_replay: no captured graph (call pgk_engine_capture first)");
    hipStream_t st = resolve_stream(s);
    for (int i = 0; i < n_steps; ++i) {
        const auto& t = e->tiers[pick_tier(e)];
        PGK_CHECK_HIP(hipGraphLaunch(t.exec, st));
        e->launches_per_step = t.launches;
        if (e->pos_hi >= 0) ++e->pos_hi;
    }
    return PGK_OK;
}

pgk_status pgk_engine_logits_ptr(pgk_engine eh, void** logits_f32) {
    PGK_REQUIRE(eh && logits_f32, "pgk_engine_logits_ptr: null argument");
    *logits_f32 = ((Engine*)eh)->logits;
    return PGK_OK;
}

pgk_status pgk_engine_read_tokens(pgk_engine eh, int32_t* h_out, int batch, int n_steps, pgk_stream s) {
    PGK_REQUIRE(eh && h_out, "pgk_engine_read_tokens: null argument");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(n_steps >= 0 && n_steps <= e->log_cap, "pgk_engine_read_tokens: %d steps exceed the log capacity %d", n_steps, e->log_cap);
    PGK_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch, "pgk_engine_read_tokens: bad batch %d", batch);
    hipStream_t st = resolve_stream(s);
    // log rows are max_batch wide; return the first `batch` columns, step-major
    std::vector<int32_t> tmp((size_t)n_steps * e->cfg.max_batch);
    if (n_steps) PGK_CHECK_HIP(hipMemcpyAsync(tmp.data(), e->token_log, tmp.size() * 4, hipMemcpyDeviceToHost, st));
    PGK_CHECK_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < n_steps; ++t)
        for (int b = 0; b < batch; ++b) h_out[(size_t)t * batch + b] = tmp[(size_t)t * e->cfg.max_batch + b];
    return PGK_OK;
}

pgk_status pgk_engine_read_clock(pgk_engine eh, uint64_t* h_out, int n_steps, pgk_stream s) {
    PGK_REQUIRE(eh && h_out, "pgk_engine_read_clock: null argument");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(n_steps >= 0 && n_steps <= e->log_cap, "pgk_engine_read_clock: %d steps exceed the log capacity %d", n_steps, e->log_cap);
    hipStream_t st = resolve_stream(s);
    if (n_steps) {
        PGK_CHECK_HIP(hipMemcpyAsync(h_out, e->clk_log, (size_t)n_steps * 16, hipMemcpyDeviceToHost, st));
        PGK_CHECK_HIP(hipStreamSynchronize(st));
    }
    return PGK_OK;
}

// In-graph stochastic sampling.  temperature <= 0 restores greedy argmax.  `h_uniforms` [n_rows][max_batch] floats in
// [0,1) are copied into the device ring: step s of the log uses row s % n_rows.  The sampling node and its arguments
// (ring pointer and length, scratch pointer, temperature, top-k, top-p) are baked into a captured graph, so ANY change
// of them - switching sampling on or off, another temperature / top-k / top-p, another n_rows, a scratch buffer that had
// to grow - drops the engine's captured graph: pgk_engine_replay then fails with "no captured graph" until
// pgk_engine_capture is called again.  A refill with the same n_rows and the same parameters only overwrites the ring's
// contents and keeps the graph.
pgk_status pgk_engine_set_sampling(pgk_engine eh, float temperature, int top_k, float top_p, const float* h_uniforms, int n_rows,
                                   pgk_stream s) {
    PGK_REQUIRE(eh, "pgk_engine_set_sampling: null engine");
    Engine* e = (Engine*)eh;
    hipStream_t st = resolve_stream(s);
    auto drop_graph = [&]() {
        if (!e->tiers.empty()) (void)hipStreamSynchronize(st);
        drop_graphs(e);
    };
    if (temperature <= 0.f) {
        if (e->sample_temperature > 0.f) drop_graph();          // a captured sampling node must not be replayed as "greedy"
        e->sample_temperature = 0.f;
        return PGK_OK;
    }
    PGK_REQUIRE(top_k >= 0 && top_p > 0.f && top_p <= 1.f, "pgk_engine_set_sampling: need top_k >= 0 and 0 < top_p <= 1");
    PGK_REQUIRE(h_uniforms && n_rows >= 1, "pgk_engine_set_sampling: uniforms missing");
    const int B = e->cfg.max_batch;
    bool changed = e->sample_temperature != temperature || e->sample_top_k != top_k || e->sample_top_p != top_p || n_rows != e->u_cap;
    if (n_rows > e->u_alloc_rows) {
        drop_graph();                                           // its nodes hold the old ring pointer
        PGK_CHECK_HIP(hipStreamSynchronize(st));
        if (e->u_ring) pgk_free(e->u_ring);
        e->u_ring = nullptr;
        e->u_alloc_rows = 0;
        if (pgk_status r = pgk_malloc((void**)&e->u_ring, (size_t)n_rows * B * 4)) return r;
        e->u_alloc_rows = n_rows;
    }
    if (!e->sampled) {
        if (pgk_status r = pgk_malloc((void**)&e->sampled, (size_t)B * 4)) return r;
    }
    if (const size_t need = sample_scratch_bytes(B, e->cfg.vocab_size, top_k, top_p); need > e->sample_scratch_cap) {
        drop_graph();
        PGK_CHECK_HIP(hipStreamSynchronize(st));
        if (e->sample_scratch) pgk_free(e->sample_scratch);
        e->sample_scratch = nullptr;
        e->sample_scratch_cap = 0;
        if (pgk_status r = pgk_malloc(&e->sample_scratch, need)) return r;
        e->sample_scratch_cap = need;
    }
    if (changed) drop_graph();
    PGK_CHECK_HIP(hipMemcpyAsync(e->u_ring, h_uniforms, (size_t)n_rows * B * 4, hipMemcpyHostToDevice, st));
    PGK_CHECK_HIP(hipStreamSynchronize(st));
    e->u_cap = n_rows;                                          // ring length = rows queued: row index is step % n_rows
    e->sample_temperature = temperature;
    e->sample_top_k = top_k;
    e->sample_top_p = top_p;
    return PGK_OK;
}

pgk_status pgk_engine_reset_log(pgk_engine eh, pgk_stream s) {
    PGK_REQUIRE(eh, "pgk_engine_reset_log: null engine");
    PGK_CHECK_HIP(hipMemsetAsync(((Engine*)eh)->step_counter, 0, 16, resolve_stream(s)));
    return PGK_OK;
}

pgk_status pgk_engine_kv_ptr(pgk_engine eh, int layer, void** k, void** v) {
    PGK_REQUIRE(eh && k && v, "pgk_engine_kv_ptr: null argument");
    Engine* e = (Engine*)eh;
    PGK_REQUIRE(layer >= 0 && layer < e->cfg.num_layers, "pgk_engine_kv_ptr: layer %d", layer);
    *k = e->kcache + (size_t)layer * e->kv_layer_elems();
    *v = e->vcache + (size_t)layer * e->kv_layer_elems();
    return PGK_OK;
}

pgk_status pgk_engine_state_ptr(pgk_engine eh, void** tokens, void** positions) {
    PGK_REQUIRE(eh && tokens && positions, "pgk_engine_state_ptr: null argument");
    *tokens = ((Engine*)eh)->tokens;
    *positions = ((Engine*)eh)->positions;
    return PGK_OK;
}

pgk_status pgk_engine_launches_per_step(pgk_engine eh, int* n) {
    PGK_REQUIRE(eh && n, "pgk_engine_launches_per_step: null argument");
    *n = ((Engine*)eh)->launches_per_step;
    return PGK_OK;
}

}  // extern "C"
