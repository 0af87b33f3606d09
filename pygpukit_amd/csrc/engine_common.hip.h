// Argument block and prologue / epilogue codes shared by the fused decode kernels of engine.hip (GEMV, M <= 8) and
// engine_batched.hip (MFMA, 3..64 sequences per launch).
#pragma once

#include <hip/hip_ext.h>

#include <tuple>
#include <vector>

#include "gemv_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

// --------------------------------------------------------------------------------------------
// Measurement hooks shared by every decode-step kernel (engine.hip, engine_batched.hip).
//
// (1) Per-launch device time: with a Probe installed and `timing` set, a launch goes through hipExtLaunchKernelGGL with
//     a start and a stop event - the dispatch's own begin / end timestamps, the interval rocprofv3 --kernel-trace reports
//     for the same launch (no inter-launch gap, no event-packet cost to subtract).  bench.py's `roofline` uses these.
// (2) Timeline of a REPLAYED step: every kernel takes a LEADING `tl` pointer (null outside the diagnostic capture; leading so
//     that it is among the kernel arguments the dispatcher preloads into SGPRs: the first-instruction stamp test then waits
//     for no scalar load);
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
    static_assert(sizeof...(KArgs) == sizeof...(Args) + 1, "launch_k: every decode-step kernel takes a leading timeline pointer");
    Probe* p = g_probe;
    unsigned long long* tl = nullptr;
    auto go = [&](hipEvent_t e0, hipEvent_t e1) {
        std::tuple<KArgs...> formal{tl, args...};   // implicit conversions to the kernel's formal parameter types happen here
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

enum { PRO_NORM = 0, PRO_PLAIN = 1, PRO_NORM_SUM = 3 };
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
    const float* part;    // PRO_NORM_SUM: [M][n_part][K]
    int nsplit;           // PRO_NORM_SUM: number of partial vectors to add
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
};

// engine_batched.hip: projections for 3..64 sequences on MFMA.  `pro`/`epi` are the codes above; `fp8` selects e4m3
// weights with 128x128 bf16 block scales.  M <= 16 accepts PRO_NORM (RMSNorm fused) or PRO_PLAIN; 17 <= M <= 64 takes
// PRO_PLAIN with a.xin16 set (rows already normalised to bf16 by norm_rows_bf16) and reads each weight byte ONCE for
// all M rows.
pgk_status batched_proj(bool fp8, int pro, int epi, const FusedArgs& a, int M, hipStream_t st, int nblk_logits = 0);
// x16[m][:] = bf16(rmsnorm(h[m][:]) * gamma): one workgroup per row
pgk_status norm_rows_bf16(const float* h, const bf16* gamma, bf16* x16, int M, int K, float eps, hipStream_t st);

}  // namespace pgk
