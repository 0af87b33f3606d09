// Argument block and prologue / epilogue codes shared by the fused decode kernels of engine.hip (GEMV, M <= 8) and
// engine_batched.hip (MFMA, 3..64 sequences per launch).
#pragma once

#include <hip/hip_ext.h>

#include <tuple>
#include <vector>

#include "gemv_core.cuh"
#include "pgk_internal.h"

namespace pgk {

// --------------------------------------------------------------------------------------------
// Measurement hooks shared by every decode-step kernel (engine.hip, engine_batched.hip).
//
// (1) Per-launch device time: with a Probe installed and `timing` set, a launch goes through hipExtLaunchKernelGGL with
//     a start and a stop event - the dispatch's own begin / end timestamps, the interval rocprofv3 --kernel-trace reports
//     for the same launch (no inter-launch gap, no event-packet cost to subtract).  bench.py's `roofline` uses these.
// (2) Timeline of a REPLAYED step: every kernel takes a trailing `tl` pointer (null outside the diagnostic capture);
//     when set, each workgroup stores the 100 MHz s_memrealtime value at its first instruction and after its last
//     barrier into its own 16-byte slot - plain stores to distinct addresses, no atomics, nothing another kernel reads.
//     pgk_engine_timeline reduces them per launch to first start / last start / first end / last end.
enum { KC_EMBED = 0, KC_NORM_QKV, KC_ATTN, KC_OPROJ, KC_GATEUP, KC_DOWN, KC_LMHEAD, KC_ARGMAX, KC_COUNT };
constexpr int TL_MAXWG = 2048;        // workgroup slots per launch (larger grids stamp their first 2048 workgroups)

struct TLInfo { int cls, nwg; };
struct Probe {
    bool timing = false;
    int cur_cls = 0;                  // class the following launches are attributed to
    std::vector<hipEvent_t> ev;       // (start, stop) pairs
    std::vector<int> cls;
    size_t used = 0;                  // pairs used
    unsigned long long* tl = nullptr; // device buffer [tl_cap][TL_MAXWG][2], or null
    int tl_cap = 0;
    std::vector<TLInfo> info;         // one entry per launch since the probe was installed
};
extern thread_local Probe* g_probe;

template <class... KArgs, class... Args>
inline hipError_t launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args) + 1, "launch_k: every decode-step kernel takes a trailing timeline pointer");
    Probe* p = g_probe;
    unsigned long long* tl = nullptr;
    auto go = [&](hipEvent_t e0, hipEvent_t e1) {
        std::tuple<KArgs...> formal{args..., tl};   // implicit conversions to the kernel's formal parameter types happen here
        std::apply([&](auto... ka) {
            if (e0) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, e0, e1, 0, ka...);
            else hipLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, ka...);
        }, formal);
    };
    if (p) {
        const int nwg = (int)(grid.x * grid.y * grid.z);
        if (p->tl && (int)p->info.size() < p->tl_cap) tl = p->tl + (size_t)p->info.size() * TL_MAXWG * 2;
        p->info.push_back(TLInfo{p->cur_cls, nwg});
        if (p->timing) {
            if (2 * p->used + 1 >= p->ev.size()) {
                hipEvent_t e0, e1;
                if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return hipErrorOutOfMemory;
                p->ev.push_back(e0);
                p->ev.push_back(e1);
                p->cls.push_back(0);
            }
            p->cls[p->used] = p->cur_cls;
            go(p->ev[2 * p->used], p->ev[2 * p->used + 1]);
            ++p->used;
            return hipGetLastError();
        }
    }
    go(nullptr, nullptr);
    return hipGetLastError();
}
static inline void mark(int cls) { if (g_probe) g_probe->cur_cls = cls; }

// in-kernel side of (2): construct first thing, call end() on every exit path (all threads of the workgroup together)
struct TLStamp {
    unsigned long long* p;
    unsigned long long t0;
    __device__ __forceinline__ explicit TLStamp(unsigned long long* tl) : p(tl), t0(0) {
        if (p) t0 = __builtin_amdgcn_s_memrealtime();
    }
    // -DPGK_PHASE_STAMPS diagnostic builds: up to 8 phase stamps per workgroup (grids <= 256 workgroups), parked in the
    // launch's unused slots [256 + 4 wg, ...); pgk_engine_timeline prints their means per kernel class to stderr
    __device__ __forceinline__ void phase(int i) const {
#ifdef PGK_PHASE_STAMPS
        if (p && threadIdx.x == 0) {
            const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            if (wg < 256u) p[2 * (256 + 4 * wg) + i] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
    __device__ __forceinline__ void end() const {
        if (p) {
            __syncthreads();
            const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            if (threadIdx.x == 0 && wg < (unsigned)TL_MAXWG) {
                p[2 * wg] = t0;
                p[2 * wg + 1] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
};

// --------------------------------------------------------------------------------------------
// Dual-chain decode step: overlap of dependent kernels (engine.hip, decode_chunk).
//
// A batch-1 token is a chain of 4 L dependent weight-streaming kernels of 1-2 us of streaming each; run back to back
// on one stream every link costs a dispatch boundary + the grid's ramp + one cold HBM round trip before the first
// useful byte (~3 us of a ~5 us kernel).  In the dual chain consecutive kernels alternate between two graph branches, so
// kernel k+1 is dispatched WHILE kernel k runs: its workgroups issue their weight loads (which depend on nothing),
// then wait for kernel k on a device counter, then read the activation vector and finish.  Protocol per the CDNA guide's
// inter-workgroup rules (Guideline 16, table row "one lane of each storing workgroup ... atomic add / sc1 poll"):
//   producer  every activation store is write-through (`sc1`), every storing wave drains (`s_waitcnt vmcnt(0)`), the
//             workgroup meets, ONE lane adds 1 to the kernel's arrival counter (8 shards on separate 64-byte lines);
//   consumer  one wave polls the 8 shards with relaxed agent-scope loads until they sum to (steps so far + 1) x the
//             producer's workgroup count, the workgroup meets, and EVERY load of producer-written bytes is an `sc1`
//             load (L1 is bypassed; no acquire fence needed).  Read-only data (weights, gammas) use ordinary loads.
// Counters only ever grow (the step count comes from a device word the step's last kernel bumps), so nothing is reset
// between replays.  Every spin is bounded: on timeout the waiter sets the error word and goes on (wrong numbers, no hang);
// the host checks the word whenever it synchronises.  Deadlock-freedom: a spinning kernel never occupies the whole chip
// (its grid x registers is < 70 % of the register file), and its producer was dispatched before it.
struct DepArgs {
    const unsigned* wait_cnt;   // predecessor's arrival counters (shard s at [16 s]), or null
    unsigned wait_per_step;     // workgroups of the predecessor per step
    unsigned* sig_cnt;          // this kernel's arrival counters, or null
    const unsigned* epoch;      // [0] = steps completed so far
    unsigned* err;              // [0] |= 1 when a wait timed out
};
constexpr int DEP_SHARDS = 8, DEP_STRIDE = 16;      // dwords between shards: one 64-byte line each
constexpr int DEP_SPIN_MAX = 1 << 14;               // x (one L2 round trip + s_sleep) ~ 10-20 ms before a waiter gives up

__device__ __forceinline__ float ld_sc1_f(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1_f(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte sc1 load: `base` wave-uniform, byte offset per lane (buffer_load_dwordx4 ... offen sc1)
__device__ __forceinline__ float4 ld_sc1_f4(const float* base, unsigned byte_off) {
    typedef unsigned dep_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    const dep_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);   // aux 16 = sc1
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// activation accessors: write-through / L1-bypassing only when the step is wired as a dual chain (`coh`); the ordinary
// single-stream step keeps ordinary loads and stores (an sc1 store drops its line from L2, so the next kernel's loads
// would go to the fabric for nothing)
__device__ __forceinline__ float ld_act(const float* p, bool coh) { return coh ? ld_sc1_f(p) : *p; }
__device__ __forceinline__ void st_act(float* p, float v, bool coh) { if (coh) st_sc1_f(p, v); else *p = v; }
__device__ __forceinline__ float4 ld_act4(const float* base, unsigned elem_off, bool coh) {
    return coh ? ld_sc1_f4(base, elem_off * 4u) : *reinterpret_cast<const float4*>(base + elem_off);
}
__device__ __forceinline__ void dep_wait(const DepArgs& d) {
    if (d.wait_cnt) {
        if (threadIdx.x < 64) {
            const unsigned target = (d.epoch[0] + 1u) * d.wait_per_step;
            const unsigned* p = d.wait_cnt + (threadIdx.x & (DEP_SHARDS - 1)) * DEP_STRIDE;
            bool ok = false;
            for (int spin = 0; spin < DEP_SPIN_MAX; ++spin) {
                unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                if ((int)(v - target) >= 0) { ok = true; break; }     // every lane holds the same sum: wave-uniform
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok && threadIdx.x == 0) atomicOr(d.err, 1u);
        }
        __syncthreads();
    }
}
// call after the kernel's last activation store, by ALL threads of the workgroup
__device__ __forceinline__ void dep_signal(const DepArgs& d) {
    if (d.sig_cnt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            __hip_atomic_fetch_add(d.sig_cnt + (wg & (DEP_SHARDS - 1)) * DEP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

enum { PRO_NORM = 0, PRO_PLAIN = 1, PRO_ATTN = 2, PRO_NORM_SUM = 3 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_LOGITS = 3 };

struct FusedArgs {
    const void* w;        // [N,K] (SWIGLU: [2*N,K], gate rows then up rows)
    const bf16* wscale;   // fp8 block scales or null
    const bf16* wp;       // batched_reg_kernel, bf16, 16 rows per workgroup: the same weights in fragment-major order (ops_pkgemm.hip) or null
    int N, K;
    const float* h;       // PRO_NORM / PRO_NORM_SUM: [M][K] residual stream
    const bf16* gamma;
    float eps;
    const float* xin;     // PRO_PLAIN: [M][K]
    const float* part;    // PRO_ATTN: [M][Hq][nsplit][D+2] ; PRO_NORM_SUM: [M][n_part][K]
    int nsplit, hq, d;    // PRO_NORM_SUM: nsplit = number of partial vectors to add
    float* h_out;         // PRO_NORM_SUM: workgroup 0 stores h + sum(part) here ([M][K])
    const float* res;     // EPI_RESID: out = res + y (res may alias out)
    float* out;           // [M][ld_out]
    int ld_out;
    float* amax_val;      // EPI_LOGITS: [M][gridDim.x]
    int* amax_idx;
    // batched MFMA path only: bf16 hand-off between projections (the consumer rounds to bf16 anyway, so the producer
    // does it once and every consuming workgroup reads half the bytes)
    const bf16* xin16;    // PRO_PLAIN: [M][K] bf16, used instead of xin when set
    bf16* out16;          // EPI_SWIGLU: [M][ld_out] bf16, written instead of out when set
    DepArgs dep;          // dual-chain step (GEMV kernels with compile-time K only); all null otherwise
    // M-tiled batched path (17..64 sequences): RMSNorm without a launch of its own.  A producer of the residual stream
    // (EPI_RESID) also leaves hb16 = bf16(h_new * gamma_next) and, per workgroup, the sum of squares of its columns of every
    // row; the consumer multiplies with the UN-normalised hb16 rows and scales its results by
    // inv[m] = rsqrt(sum over the producer's workgroups / K + eps) - a per-row scalar commutes with the product.
    bf16* hb16_out;           // EPI_RESID: [M][ld_out]
    const bf16* gamma_next;   // EPI_RESID: [N]
    float* ss_out;            // EPI_RESID: [64 rows][1024]: column = the producing workgroup
    const float* ss_in;       // consumer: the same table (null: the rows in xin16 are already normalised)
    int ss_n;                 // producer workgroups (<= 1024)
};

// engine_batched.hip: projections for 3..64 sequences on MFMA.  `pro`/`epi` are the codes above; `fp8` selects e4m3
// weights with 128x128 bf16 block scales.  M <= 16 accepts PRO_NORM (RMSNorm fused) or PRO_PLAIN; 17 <= M <= 64 takes
// PRO_PLAIN with a.xin16 set (rows already normalised to bf16 by norm_rows_bf16) and reads each weight byte ONCE for
// all M rows.
pgk_status batched_proj(bool fp8, int pro, int epi, const FusedArgs& a, int M, hipStream_t st, int nblk_logits = 0);
// workgroups the M-tiled kernel launches for an N-column projection with epilogue `epi` (= the columns of the
// sum-of-squares table an EPI_RESID producer fills)
int batched_tiled_groups(int N, int epi);
// x16[m][:] = bf16(rmsnorm(h[m][:]) * gamma): one workgroup per row
pgk_status norm_rows_bf16(const float* h, const bf16* gamma, bf16* x16, int M, int K, float eps, hipStream_t st);

}  // namespace pgk
