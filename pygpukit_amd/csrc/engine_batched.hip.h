// Batched decode projections on MFMA (3 <= M <= 64 sequences per launch), included by engine_batched.hip.
//
// The GEMV kernels of engine.hip score every weight chunk against M activation rows with VALU dot products: 32 packed
// dot instructions per 16-byte weight chunk at M = 8, and the kernel turns instruction-bound (gate_up 9.3 us, lm_head
// 86 us at M = 8 against 5.7 / 50 us at M = 1).  Here the M rows are the (zero-padded) A operand of
// v_mfma_f32_16x16x32_bf16 and 16 weight rows are its B operand, loaded straight from HBM in fragment shape
// (lane l: W[n0 + (l&15)][k + 8*(l>>4) ..+8], the layout of ops_wsgemm.hip): one MFMA replaces those 32 instructions and
// the cost of a projection no longer depends on M.
//   workgroup = 4 waves = 16 output rows (SWIGLU: 16 gate rows and their 16 up rows); the waves split K four ways
//   (every lane's weight loads - K/128 of them - are issued before the prologue touches the activations), partial
//   16x16 tiles are summed through LDS; prologue = RMSNorm (or plain copy) of the M rows into an LDS image laid out
//   [k/8][MP] x 16 B so that an A-fragment read is 64 consecutive 16-byte slots; epilogues as in engine.hip
//   (store / residual add / SiLU(g)*u / logits + per-workgroup argmax partials, lowest index on ties).
// fp8 weights: 8 codes per lane per step, widened to bf16 with the block scale in registers (as ops_wsgemm.hip).

// MP: activation rows held in LDS (8 or 16).  S: 32-k steps per wave preloaded into registers (K = 128 S when it is one of
// the instantiated 8 / 16 / 24; any further steps stream through a plain loop).
template <class WT, int PRO, int EPI, int MP, int S>
__global__ __launch_bounds__(256) void batched_mfma_kernel(unsigned long long* tl, FusedArgs a, int M, int steps /* K / 128 per wave */, int nblk_logits) {
    const TLStamp tls(tl);
    constexpr bool FP8 = std::is_same<WT, fp8e4m3>::value;
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;       // weight row groups per workgroup
    constexpr int MAXS = S > 0 ? S : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // x image [K/8][MP] x 16 B | partial tiles
    __shared__ float s_ss[4][16];
    const int K = a.K, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, q = lane >> 4, l15 = lane & 15;
    const int ngroups = (N + 15) >> 4;                     // 16-row groups; a workgroup takes groups blockIdx.x, + gridDim.x, ...
    const int kw0 = wid * steps * 32;                      // first k of this wave
    uint4 wv[NT][MAXS];
    float wsc[NT][MAXS];
    auto load_w = [&](int n0) {
        const int nrow = min(n0 + l15, N - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t row = (size_t)(t == 0 ? nrow : N + nrow);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = kw0 + s * 32 + 8 * q;
                if constexpr (FP8) {
                    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(a.w) + row * K + k);
                    wv[t][s] = make_uint4(v.x, v.y, 0, 0);
                    wsc[t][s] = to_f(a.wscale[(row >> 7) * (size_t)(K >> 7) + (k >> 7)]);
                } else {
                    wv[t][s] = load_nt16(reinterpret_cast<const bf16*>(a.w) + row * K + k);
                }
            }
        }
    };
    // ---- weight loads of the first group's K quarter, all issued before the prologue touches the activations ----
    load_w(blockIdx.x * 16);
    const int em = tid >> 4, en = tid & 15;                // epilogue element of this thread: (row em, column n0 + en)

    // ---- prologue: M rows -> bf16 LDS image, chunk c (8 k) of row m at (c * MP + m) * 16 ----
    // Rows are unrolled to MP with clamped addresses and masked use (a runtime row loop would index registers
    // dynamically and wait per load); chunk loops have a uniform trip count for the same reason.
    const int nchunk = K >> 3;
    const float* src = (PRO == PRO_NORM) ? a.h : a.xin;
    if constexpr (PRO == PRO_NORM) {
        float ss[MP];
#pragma unroll
        for (int m = 0; m < MP; ++m) ss[m] = 0.f;
        for (int c0 = 0; c0 < nchunk; c0 += 256) {
            const int c = min(c0 + tid, nchunk - 1);
            const bool live = c0 + tid < nchunk;
            float4 v0[MP], v1[MP];
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                v0[m] = *reinterpret_cast<const float4*>(src + (size_t)min(m, M - 1) * K + c * 8);
                v1[m] = *reinterpret_cast<const float4*>(src + (size_t)min(m, M - 1) * K + c * 8 + 4);
            }
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const float p = v0[m].x * v0[m].x + v0[m].y * v0[m].y + v0[m].z * v0[m].z + v0[m].w * v0[m].w + v1[m].x * v1[m].x +
                                v1[m].y * v1[m].y + v1[m].z * v1[m].z + v1[m].w * v1[m].w;
                ss[m] += live ? p : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            const float t = wave_sum(ss[m]);
            if (lane == 0) s_ss[wid][m] = t;
        }
        __syncthreads();
    }
    for (int c0 = 0; c0 < nchunk; c0 += 256) {
        const int c = min(c0 + tid, nchunk - 1);
        float g[8];
        if constexpr (PRO == PRO_NORM) {
            Vec<bf16> gr;
            gr.load(a.gamma + c * 8);
            gr.to_float(g);
        }
        if constexpr (PRO != PRO_NORM) {
            if (a.xin16) {                                  // bf16 hand-off: copy the stored chunks into the image
#pragma unroll
                for (int m = 0; m < MP; ++m) {
                    uint4 raw = *reinterpret_cast<const uint4*>(a.xin16 + (size_t)min(m, M - 1) * K + c * 8);
                    if (m >= M) raw = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<uint4*>(smem + ((size_t)c * MP + m) * 16) = raw;
                }
                continue;
            }
        }
        float4 v0[MP], v1[MP];
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            v0[m] = *reinterpret_cast<const float4*>(src + (size_t)min(m, M - 1) * K + c * 8);
            v1[m] = *reinterpret_cast<const float4*>(src + (size_t)min(m, M - 1) * K + c * 8 + 4);
        }
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            float f[8] = {v0[m].x, v0[m].y, v0[m].z, v0[m].w, v1[m].x, v1[m].y, v1[m].z, v1[m].w};
            if constexpr (PRO == PRO_NORM) {
                const float inv = 1.0f / sqrtf((s_ss[0][m] + s_ss[1][m] + s_ss[2][m] + s_ss[3][m]) / K + a.eps);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = f[j] * inv * g[j];
            }
            Vec<bf16> o;
            o.from_float(f);
            if (m >= M) o.raw = make_uint4(0, 0, 0, 0);     // unused rows of the image are zero
            *reinterpret_cast<uint4*>(smem + ((size_t)c * MP + m) * 16) = o.raw;   // (duplicate writes of the clamped tail chunk are identical)
        }
    }
    __syncthreads();

    // ---- body: one 16-row group per trip; the LDS image of the activations stays, the partial tiles have their own area ----
    float* red = reinterpret_cast<float*>(smem + (size_t)nchunk * MP * 16);   // [NT][4 waves][16 m][16 n]
    const int am = l15 & (MP - 1);
    auto bfrag = [&](int t, int s) -> uint4 {
        if constexpr (FP8) {
            const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, false), c1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, true);
            const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, false), c3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, true);
            const float sc = wsc[t][s];
            return make_uint4(pack_bf16x2(c0.x * sc, c0.y * sc), pack_bf16x2(c1.x * sc, c1.y * sc), pack_bf16x2(c2.x * sc, c2.y * sc),
                              pack_bf16x2(c3.x * sc, c3.y * sc));
        } else {
            return wv[t][s];
        }
    };
    float bv = -INFINITY;          // EPI_LOGITS: best of this thread's elements over the workgroup's groups
    int bi = 0x7FFFFFFF;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int n0 = g * 16;
        if (g != (int)blockIdx.x) load_w(n0);
        float resv = 0.f;
        if constexpr (EPI == EPI_RESID) resv = a.res[(size_t)min(em, M - 1) * a.ld_out + min(n0 + en, N - 1)];
        f32x4_b acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int kc = (kw0 >> 3) + s * 4 + q;
            const uint4 af = *reinterpret_cast<const uint4*>(smem + ((size_t)kc * MP + am) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b, af), __builtin_bit_cast(bf16x8_b, bfrag(t, s)), acc[t], 0, 0, 0);
        }
        for (int s = S; s < steps; ++s) {          // steps beyond the preloaded ones stream through a plain loop
            const int k = kw0 + s * 32 + 8 * q;
            const uint4 af = *reinterpret_cast<const uint4*>(smem + ((size_t)((k >> 3)) * MP + am) * 16);
            const int nrow = min(n0 + l15, N - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const size_t row = (size_t)(t == 0 ? nrow : N + nrow);
                uint4 b;
                if constexpr (FP8) {
                    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(a.w) + row * K + k);
                    const float sc = to_f(a.wscale[(row >> 7) * (size_t)(K >> 7) + (k >> 7)]);
                    const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), c1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
                    const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), c3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
                    b = make_uint4(pack_bf16x2(c0.x * sc, c0.y * sc), pack_bf16x2(c1.x * sc, c1.y * sc), pack_bf16x2(c2.x * sc, c2.y * sc),
                                   pack_bf16x2(c3.x * sc, c3.y * sc));
                } else {
                    b = load_nt16(reinterpret_cast<const bf16*>(a.w) + row * K + k);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b, af), __builtin_bit_cast(bf16x8_b, b), acc[t], 0, 0, 0);
            }
        }
        // sum the four K quarters: C tile element (m = 4 q + r, n = l15)
        __syncthreads();                                    // the previous trip's partial tiles have been read
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((t * 4 + wid) * 16 + q * 4 + r) * 16 + l15] = acc[t][r];
        __syncthreads();
        float y[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* p = red + (size_t)t * 4 * 256 + em * 16 + en;
            y[t] = p[0] + p[256] + p[512] + p[768];
        }
        const bool ok = em < M && n0 + en < N;
        const size_t o = (size_t)em * a.ld_out + n0 + en;
        if constexpr (EPI == EPI_STORE) {
            if (ok) a.out[o] = y[0];
        } else if constexpr (EPI == EPI_RESID) {
            if (ok) a.out[o] = resv + y[0];
        } else if constexpr (EPI == EPI_SWIGLU) {
            if (ok) {
                const float act = y[0] / (1.0f + __expf(-y[0])) * y[NT - 1];
                if (a.out16) a.out16[o] = from_f<bf16>(act);
                else a.out[o] = act;
            }
        } else {   // EPI_LOGITS
            if (ok) {
                a.out[o] = y[0];
                if (y[0] > bv) { bv = y[0]; bi = n0 + en; }     // groups ascend: a later equal value never replaces an earlier one
            }
        }
    }
    if constexpr (EPI == EPI_LOGITS) {   // best of this workgroup's columns per row (lowest index on ties)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (en == 0 && em < M) {
            a.amax_val[(size_t)em * nblk_logits + blockIdx.x] = bv;
            a.amax_idx[(size_t)em * nblk_logits + blockIdx.x] = bi;
        }
    }
    tls.end();
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident variant (K = 128 S, S in {8, 16, 24, 32}): the default.
//
// The LDS-image kernel above spends its time before the first MFMA: every workgroup reads all M rows twice (sum of
// squares, then normalise), writes a K x 16 image to LDS and crosses two barriers - 9-13 us per projection at M = 16
// where the M = 1 GEMV kernels take 4.6-5.8.  A wave only ever multiplies ITS quarter of K, so here each wave loads that
// quarter of the activation rows straight from global memory in A-fragment shape (lane l: row l & 15, k = 32 s + 8 (l >> 4)
// .. + 8), keeps the raw fp32 values in registers while the row sums of squares cross the waves (64 floats of LDS,
// one barrier), and converts in place: no activation image, one pass over the rows.
// RPG = weight rows per workgroup (16, 8 or 4): the N = hidden projections have only N / 16 = 64 row groups at
// N = 1024, far too few workgroups to pull HBM bandwidth; with RPG < 16 the lanes of the unused B columns load nothing
// and the epilogue ignores those columns (the MFMA computes them on whatever the registers hold - columns are
// independent).
template <class WT, int PRO, int EPI, int S, int RPG>
__global__ __launch_bounds__(256) void batched_reg_kernel(unsigned long long* tl, const void* w_, const bf16* wp_, const float* x_, const bf16* gamma_,
                                                          int N_, int K_, int M, int nblk_logits, FusedArgs a) {
    // the first 14 dwords of the arguments - everything the load-issue phase needs - arrive preloaded in SGPRs (see fused_gemv_kernel,
    // engine.hip): w_ / wp_ = FusedArgs::w / wp, x_ = the fp32 input rows (h for the norm prologue, xin otherwise), gamma_, N_, K_
    const TLStamp tls(tl);
    constexpr bool FP8 = std::is_same<WT, fp8e4m3>::value;
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    __shared__ float s_ss[4][16];
    __shared__ float red[NT * 4 * 256];                    // [NT][4 waves][16 m][16 n]
    const int K = K_, N = N_;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, q = lane >> 4, l15 = lane & 15;
    const int ngroups = (N + RPG - 1) / RPG;
    const int kw0 = wid * S * 32;
    const bool wlane = l15 < RPG;                          // this lane holds a real weight row
    uint4 wv[NT][S];
    float wsc[NT][S];
    auto load_w = [&](int n0) {
        const int nrow = min(n0 + l15, N - 1);
        if constexpr (!FP8 && RPG == 16) {
            // fragment-major copy (ops_pkgemm.hip): block (n-tile, k-step) is this very fragment, 1 KiB in lane order - one
            // coalesced load per k-step instead of 64 separate 16-byte pieces of a row-major matrix (3.5x the texture
            // addresser's time per instruction, tools/micro/ta_probe.hip).  Used where a workgroup streams many groups (lm_head).
            if (wp_ != nullptr && n0 + 16 <= N) {
                const int ksn = K >> 5;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16* run = wp_ + (((size_t)((t == 0 ? n0 : N + n0) >> 4) * ksn + (kw0 >> 5)) * 64 + lane) * 8;
#pragma unroll
                    for (int s = 0; s < S; ++s) wv[t][s] = load_nt16(run + (size_t)s * 512);
                }
                return;
            }
        }
        if (wlane) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const size_t row = (size_t)(t == 0 ? nrow : N + nrow);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int k = kw0 + s * 32 + 8 * q;
                    if constexpr (FP8) {
                        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(w_) + row * K + k);
                        wv[t][s] = make_uint4(v.x, v.y, 0, 0);
                        wsc[t][s] = to_f(a.wscale[(row >> 7) * (size_t)(K >> 7) + (k >> 7)]);
                    } else {
                        wv[t][s] = load_nt16(reinterpret_cast<const bf16*>(w_) + row * K + k);
                    }
                }
            }
        }
    };
    const int em = tid >> 4, en = tid & 15;

    // ---- activation fragments of this wave's K quarter ----
    // Issue order = arrival order: the activation rows (L2) are requested BEFORE the weight stream (HBM), so the RMSNorm
    // statistic, its barrier and the bf16 conversion run while the weights are still in flight (the other way round the
    // prologue started when the last weight byte had landed - the batch-1 kernels' lesson).
    const float* src = x_ + (size_t)min(l15, M - 1) * K + kw0 + 8 * q;
    uint4 af[S];
    constexpr bool HOLD_FIRST = PRO == PRO_NORM && S <= 8;      // raw rows held in registers across the reduction
    if constexpr (!HOLD_FIRST) load_w(blockIdx.x * RPG);        // (deep K: the rows are re-read after the barrier anyway)
    auto pack8 = [](const float4& u, const float4& v, float inv, const float* g) {
        return make_uint4(pack_bf16x2(u.x * inv * g[0], u.y * inv * g[1]), pack_bf16x2(u.z * inv * g[2], u.w * inv * g[3]),
                          pack_bf16x2(v.x * inv * g[4], v.y * inv * g[5]), pack_bf16x2(v.z * inv * g[6], v.w * inv * g[7]));
    };
    if constexpr (PRO == PRO_NORM) {
        constexpr bool HOLD = S <= 8;                      // raw rows stay in registers across the reduction
        float4 r0[HOLD ? S : 1], r1[HOLD ? S : 1];
        float ss = 0.f;
        if constexpr (HOLD) {
#pragma unroll
            for (int s = 0; s < S; ++s) { r0[s] = *reinterpret_cast<const float4*>(src + s * 32); r1[s] = *reinterpret_cast<const float4*>(src + s * 32 + 4); }
            __builtin_amdgcn_sched_barrier(0);
            load_w(blockIdx.x * RPG);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 u = r0[s], v = r1[s];
                ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 u = *reinterpret_cast<const float4*>(src + s * 32), v = *reinterpret_cast<const float4*>(src + s * 32 + 4);
                ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w + v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (q == 0) s_ss[wid][l15] = ss;
        __syncthreads();
        const float inv = 1.0f / sqrtf((s_ss[0][l15] + s_ss[1][l15] + s_ss[2][l15] + s_ss[3][l15]) / K + a.eps);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            Vec<bf16> gr;
            gr.load(gamma_ + kw0 + s * 32 + 8 * q);
            float g[8];
            gr.to_float(g);
            if constexpr (HOLD) {
                af[s] = pack8(r0[s], r1[s], inv, g);
            } else {
                const float4 u = *reinterpret_cast<const float4*>(src + s * 32), v = *reinterpret_cast<const float4*>(src + s * 32 + 4);
                af[s] = pack8(u, v, inv, g);
            }
        }
    } else if (a.xin16) {                                  // bf16 hand-off: the fragment is the stored 16-byte chunk
        const bf16* src16 = a.xin16 + (size_t)min(l15, M - 1) * K + kw0 + 8 * q;
#pragma unroll
        for (int s = 0; s < S; ++s) af[s] = *reinterpret_cast<const uint4*>(src16 + s * 32);
    } else {
        const float one[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(src + s * 32), v = *reinterpret_cast<const float4*>(src + s * 32 + 4);
            af[s] = pack8(u, v, 1.0f, one);
        }
    }
    if (l15 >= M) {
#pragma unroll
        for (int s = 0; s < S; ++s) af[s] = make_uint4(0, 0, 0, 0);      // rows past the batch contribute nothing
    }

    auto bfrag = [&](int t, int s) -> uint4 {
        if constexpr (FP8) {
            const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, false), c1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, true);
            const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, false), c3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, true);
            const float sc = wsc[t][s];
            return make_uint4(pack_bf16x2(c0.x * sc, c0.y * sc), pack_bf16x2(c1.x * sc, c1.y * sc), pack_bf16x2(c2.x * sc, c2.y * sc),
                              pack_bf16x2(c3.x * sc, c3.y * sc));
        } else {
            return wv[t][s];
        }
    };
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int n0 = g * RPG;
        float resv = 0.f;
        if constexpr (EPI == EPI_RESID) resv = a.res[(size_t)min(em, M - 1) * a.ld_out + min(n0 + en, N - 1)];
        f32x4_b acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b, af[s]), __builtin_bit_cast(bf16x8_b, bfrag(t, s)), acc[t], 0, 0, 0);
        // the next group's weights (lm_head: a workgroup walks several groups) travel while this one is reduced and stored
        if (g + (int)gridDim.x < ngroups) load_w((g + (int)gridDim.x) * RPG);
        __syncthreads();                                    // the previous trip's partial tiles have been read
        if (wlane) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((t * 4 + wid) * 16 + q * 4 + r) * 16 + l15] = acc[t][r];
        }
        __syncthreads();
        const bool ok = em < M && en < RPG && n0 + en < N;
        float y[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* p = red + (size_t)t * 4 * 256 + em * 16 + en;
            y[t] = ok ? p[0] + p[256] + p[512] + p[768] : 0.f;
        }
        const size_t o = (size_t)em * a.ld_out + n0 + en;
        if constexpr (EPI == EPI_STORE) {
            if (ok) a.out[o] = y[0];
        } else if constexpr (EPI == EPI_RESID) {
            if (ok) a.out[o] = resv + y[0];
        } else if constexpr (EPI == EPI_SWIGLU) {
            if (ok) {
                const float act = y[0] / (1.0f + __expf(-y[0])) * y[NT - 1];
                if (a.out16) a.out16[o] = from_f<bf16>(act);
                else a.out[o] = act;
            }
        } else {   // EPI_LOGITS
            if (ok) {
                a.out[o] = y[0];
                if (y[0] > bv) { bv = y[0]; bi = n0 + en; }
            }
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (en == 0 && em < M) {
            a.amax_val[(size_t)em * nblk_logits + blockIdx.x] = bv;
            a.amax_idx[(size_t)em * nblk_logits + blockIdx.x] = bi;
        }
    }
    tls.end();
}

// ---------------------------------------------------------------------------------------------------------------------
// M-tiled variant: 17..64 sequences per launch (MT = 2 or 4 tiles of 16 rows), every weight byte still read ONCE.
//
// Chunks of at most 16 sequences made a batch of 64 four full passes over the weights (4 x 1.08 ms per step on
// Qwen3-0.6B).  Here a wave keeps its K quarter of the workgroup's weight rows in registers exactly as above and runs
// MT MFMAs per weight fragment, one per 16-row tile of the batch.  The activation rows arrive as bf16 (the RMSNorm'ed
// rows from norm_rows_bf16, the attention output, the SwiGLU output): an A fragment is one 16-byte load, no conversion,
// no cross-wave reduction before the first MFMA.  S * MT <= 32 fragments stay resident (and are reused by every weight
// group an lm_head workgroup walks); beyond that they stream from L2 in chunks of CH = 4 k-steps, the next chunk's loads
// issued before the current chunk's MFMAs.  Rows >= M of the last tile repeat row M - 1: MFMA output rows are independent
// and the epilogue drops them.

template <class WT, int EPI, int S, int RPG, int MT>
__global__ __launch_bounds__(256) void batched_mt_kernel(unsigned long long* tl, FusedArgs a, int M, int nblk_logits) {
    const TLStamp tls(tl);
    constexpr bool FP8 = std::is_same<WT, fp8e4m3>::value;
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int CH = (S * MT <= 32) ? S : 4;             // k-steps per activation chunk
    constexpr int NCH = S / CH;
    static_assert(S % CH == 0, "chunking");
    __shared__ float red[NT * 4 * MT * 256];               // [NT][4 waves][MT][16 m][16 n]
    const int K = a.K, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, q = lane >> 4, l15 = lane & 15;
    const int ngroups = (N + RPG - 1) / RPG;
    const int kw0 = wid * S * 32;
    const bool wlane = l15 < RPG;
    uint4 wv[NT][S];
    float wsc[NT][S];
    auto load_w = [&](int n0) {
        const int nrow = min(n0 + l15, N - 1);
        if constexpr (!FP8 && RPG == 16) {
            if (a.wp != nullptr && n0 + 16 <= N) {          // fragment-major copy: see batched_reg_kernel
                const int ksn = K >> 5;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16* run = a.wp + (((size_t)((t == 0 ? n0 : N + n0) >> 4) * ksn + (kw0 >> 5)) * 64 + lane) * 8;
#pragma unroll
                    for (int s = 0; s < S; ++s) wv[t][s] = load_nt16(run + (size_t)s * 512);
                }
                return;
            }
        }
        if (wlane) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const size_t row = (size_t)(t == 0 ? nrow : N + nrow);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int k = kw0 + s * 32 + 8 * q;
                    if constexpr (FP8) {
                        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(a.w) + row * K + k);
                        wv[t][s] = make_uint4(v.x, v.y, 0, 0);
                        wsc[t][s] = to_f(a.wscale[(row >> 7) * (size_t)(K >> 7) + (k >> 7)]);
                    } else {
                        wv[t][s] = load_nt16(reinterpret_cast<const bf16*>(a.w) + row * K + k);
                    }
                }
            }
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    load_w(blockIdx.x * RPG);                              // in flight before the activations are touched
    const int em = tid >> 4, en = tid & 15;
    __builtin_amdgcn_sched_barrier(0);

    const bf16* xrow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xrow[t] = a.xin16 + (size_t)min(t * 16 + l15, M - 1) * K + kw0 + 8 * q;
    uint4 af[NCH > 1 ? 2 : 1][CH][MT];
    auto load_a = [&](int buf, int c) {
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) af[buf][j][t] = *reinterpret_cast<const uint4*>(xrow[t] + (c * CH + j) * 32);
    };
    if constexpr (NCH == 1) load_a(0, 0);
    auto bfrag = [&](int t, int s) -> uint4 {
        if constexpr (FP8) {
            const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, false), c1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].x, true);
            const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, false), c3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)wv[t][s].y, true);
            const float sc = wsc[t][s];
            return make_uint4(pack_bf16x2(c0.x * sc, c0.y * sc), pack_bf16x2(c1.x * sc, c1.y * sc), pack_bf16x2(c2.x * sc, c2.y * sc),
                              pack_bf16x2(c3.x * sc, c3.y * sc));
        } else {
            return wv[t][s];
        }
    };
    float bv[MT];
    int bi[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { bv[t] = -INFINITY; bi[t] = 0x7FFFFFFF; }
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int n0 = g * RPG;
        float resv[MT];
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int t = 0; t < MT; ++t) resv[t] = a.res[(size_t)min(t * 16 + em, M - 1) * a.ld_out + min(n0 + en, N - 1)];
        }
        f32x4_b acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[t][u] = f32x4_b{0.f, 0.f, 0.f, 0.f};
        if constexpr (NCH > 1) load_a(0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if constexpr (NCH > 1) { if (c + 1 < NCH) load_a((c + 1) & 1, c + 1); }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const uint4 b = bfrag(u, c * CH + j);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b, af[c & 1][j][t]),
                                                                             __builtin_bit_cast(bf16x8_b, b), acc[t][u], 0, 0, 0);
                }
            }
        }
        // the next group's weights (lm_head: a workgroup walks several groups) travel while this one is reduced and stored
        if (g + (int)gridDim.x < ngroups) load_w((g + (int)gridDim.x) * RPG);
        __syncthreads();                                    // the previous trip's partial tiles have been read
        if (wlane) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(((u * 4 + wid) * MT + t) * 16 + q * 4 + r) * 16 + l15] = acc[t][u][r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 16 + em;
            const bool ok = m < M && en < RPG && n0 + en < N;
            float y[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float* p = red + ((size_t)(u * 4) * MT + t) * 256 + em * 16 + en;
                y[u] = ok ? (p[0] + p[MT * 256] + p[2 * MT * 256] + p[3 * MT * 256]) : 0.f;
            }
            const size_t o = (size_t)m * a.ld_out + n0 + en;
            if constexpr (EPI == EPI_STORE) {
                if (ok) a.out[o] = y[0];
            } else if constexpr (EPI == EPI_RESID) {
                const float v = resv[t] + y[0];
                if (ok) a.out[o] = v;
            } else if constexpr (EPI == EPI_SWIGLU) {
                if (ok) {
                    const float act = y[0] / (1.0f + __expf(-y[0])) * y[NT - 1];
                    if (a.out16) a.out16[o] = from_f<bf16>(act);
                    else a.out[o] = act;
                }
            } else {   // EPI_LOGITS
                if (ok) {
                    a.out[o] = y[0];
                    if (y[0] > bv[t]) { bv[t] = y[0]; bi[t] = n0 + en; }
                }
            }
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float v = bv[t];
            int ix = bi[t];
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                const float ov = __shfl_xor(v, off, 64);
                const int oi = __shfl_xor(ix, off, 64);
                if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
            }
            const int m = t * 16 + em;
            if (en == 0 && m < M) {
                a.amax_val[(size_t)m * nblk_logits + blockIdx.x] = v;
                a.amax_idx[(size_t)m * nblk_logits + blockIdx.x] = ix;
            }
        }
    }
    tls.end();
}

// (Tried for the N = hidden projections at batch 64: 16 rows per workgroup - 64 workgroups, a quarter of the activation
// traffic: 17.2 / 12.1 us against 15.5 / 11.1; and the weights of 16 rows parked in LDS with one wave per 16-row batch tile,
// no cross-wave reduction: 16.4 us average.  A workgroup keeps at most ~64 KB of loads in flight, so 64 workgroups cannot
// pull the 6 MB of weights plus their activation rows faster than 256 four-row workgroups that each re-read the rows.)
static int mt_rows_per_group(int N, int epi) { return (epi == EPI_LOGITS || ceil_div(N, 16) >= 192) ? 16 : 4; }

template <class WT, int EPI, int S>
static pgk_status launch_batched_mt(const FusedArgs& a, int M, hipStream_t st, int nblk_logits) {
    const int rpg = mt_rows_per_group(a.N, EPI);
    const int grid = (EPI == EPI_LOGITS) ? nblk_logits : ceil_div(a.N, rpg);
    const bool two = M <= 32;
    hipError_t he = hipSuccess;
    if constexpr (EPI == EPI_LOGITS) {
        if (two) he = launch_k(batched_mt_kernel<WT, EPI, S, 16, 2>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
        else he = launch_k(batched_mt_kernel<WT, EPI, S, 16, 4>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
    } else {
        if (rpg == 16 && two) he = launch_k(batched_mt_kernel<WT, EPI, S, 16, 2>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
        else if (rpg == 16) he = launch_k(batched_mt_kernel<WT, EPI, S, 16, 4>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
        else if (two) he = launch_k(batched_mt_kernel<WT, EPI, S, 4, 2>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
        else he = launch_k(batched_mt_kernel<WT, EPI, S, 4, 4>, dim3(grid), dim3(256), 0, st, a, M, nblk_logits);
    }
    PGK_CHECK_HIP(he);
    return PGK_OK;
}

template <class WT, int EPI>
static pgk_status launch_batched_tiled(const FusedArgs& a, int M, hipStream_t st, int nblk_logits = 0) {
    PGK_REQUIRE(a.xin16 != nullptr, "batched decode projection: %d sequences need bf16 input rows", M);
    PGK_REQUIRE(M > 16 && M <= 64, "batched decode projection: tiled kernels take 17..64 sequences, got %d", M);
    switch (a.K / 128) {
        case 2: return launch_batched_mt<WT, EPI, 2>(a, M, st, nblk_logits);
        case 4: return launch_batched_mt<WT, EPI, 4>(a, M, st, nblk_logits);
        case 8: return launch_batched_mt<WT, EPI, 8>(a, M, st, nblk_logits);
        case 16: return launch_batched_mt<WT, EPI, 16>(a, M, st, nblk_logits);
        case 24: return launch_batched_mt<WT, EPI, 24>(a, M, st, nblk_logits);
        case 32: return launch_batched_mt<WT, EPI, 32>(a, M, st, nblk_logits);
    }
    return set_error(PGK_ERR_UNSUPPORTED, "batched decode projection: K=%d is not 128 x {2,4,8,16,24,32}", a.K);
}

template <class WT, int PRO, int EPI, int S>
static pgk_status launch_batched_reg(const FusedArgs& a, int M, hipStream_t st, int nblk_logits) {
    // enough workgroups to pull HBM bandwidth: fewer weight rows per workgroup when N / 16 would leave CUs idle
    int rpg = (EPI == EPI_LOGITS || ceil_div(a.N, 16) >= 192) ? 16 : (ceil_div(a.N, 8) >= 192 ? 8 : 4);
    const int grid = (EPI == EPI_LOGITS) ? nblk_logits : ceil_div(a.N, rpg);
    const float* x = (PRO == PRO_NORM) ? a.h : a.xin;
    hipError_t he;
    if (rpg == 16) he = launch_k(batched_reg_kernel<WT, PRO, EPI, S, 16>, dim3(grid), dim3(256), 0, st, a.w, a.wp, x, a.gamma, a.N, a.K, M, nblk_logits, a);
    else if (rpg == 8) he = launch_k(batched_reg_kernel<WT, PRO, EPI, S, 8>, dim3(grid), dim3(256), 0, st, a.w, a.wp, x, a.gamma, a.N, a.K, M, nblk_logits, a);
    else he = launch_k(batched_reg_kernel<WT, PRO, EPI, S, 4>, dim3(grid), dim3(256), 0, st, a.w, a.wp, x, a.gamma, a.N, a.K, M, nblk_logits, a);
    PGK_CHECK_HIP(he);
    return PGK_OK;
}

template <class WT, int PRO, int EPI, int MP, int S>
static pgk_status launch_batched_s(const FusedArgs& a, int M, int steps, int grid, size_t lds, hipStream_t st, int nblk_logits) {
    static bool done = false;
    if (lds > 48 * 1024 && !done) {
        PGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&batched_mfma_kernel<WT, PRO, EPI, MP, S>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        done = true;
    }
    PGK_CHECK_HIP(launch_k(batched_mfma_kernel<WT, PRO, EPI, MP, S>, dim3(grid), dim3(256), lds, st, a, M, steps, nblk_logits));
    return PGK_OK;
}

template <class WT, int PRO, int EPI>
static pgk_status launch_batched(const FusedArgs& a, int M, hipStream_t st, int nblk_logits = 0) {
    PGK_REQUIRE(a.K % 128 == 0 && M >= 1 && M <= 16, "batched decode projection: K=%d must be a multiple of 128 and M=%d in [1,16]", a.K, M);
    const int steps = a.K / 128, ngroups = ceil_div(a.N, 16);
    // register-resident activations at the instantiated widths; the LDS-image kernel below for every other K
    if (steps == 8) return launch_batched_reg<WT, PRO, EPI, 8>(a, M, st, nblk_logits);
    if (steps == 16) return launch_batched_reg<WT, PRO, EPI, 16>(a, M, st, nblk_logits);
    if (steps == 24) return launch_batched_reg<WT, PRO, EPI, 24>(a, M, st, nblk_logits);
    if (steps == 32) return launch_batched_reg<WT, PRO, EPI, 32>(a, M, st, nblk_logits);
    // the prologue (RMSNorm of the M rows) is per workgroup: large N (lm_head) runs 2048 workgroups over several groups each
    const int grid = (EPI == EPI_LOGITS) ? nblk_logits : ngroups;
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int mp = M <= 8 ? 8 : 16;
    const size_t lds = (size_t)(a.K / 8) * mp * 16 + (size_t)NT * 4 * 256 * 4;
    PGK_REQUIRE(lds <= 158 * 1024, "batched decode projection: K=%d does not fit the LDS image", a.K);
#define PGK_BS(MPV, SV) return launch_batched_s<WT, PRO, EPI, MPV, SV>(a, M, steps, grid, lds, st, nblk_logits);
    if (mp == 8) {
        if (steps == 8) PGK_BS(8, 8)
        if (steps == 16) PGK_BS(8, 16)
        if (steps == 24) PGK_BS(8, 24)
        PGK_BS(8, 0)
    }
    if (steps == 8) PGK_BS(16, 8)
    if (steps == 16) PGK_BS(16, 16)
    if (steps == 24) PGK_BS(16, 24)
    PGK_BS(16, 0)
#undef PGK_BS
}
