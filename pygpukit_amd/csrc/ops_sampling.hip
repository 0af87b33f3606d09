// Stochastic token sampling on the device: temperature / top-k / top-p (nucleus), one workgroup per logits row.
//
// Reference: native/ops/sampling/sampling_kernels.cuh:200-830 + sampling.cu (sample_multinomial / sample_topk /
// sample_topp / sample_topk_to_buf_ptr) and the host sampler src/pygpukit/llm/sampling.py:12-63.  The reference's
// device kernels are not a usable specification (its top-k fills a shared array with unsynchronised atomicExch, so
// the kept set depends on thread timing; its "top-p" only scales the random number), so this build implements the
// semantics its docstrings and its HOST sampler state, as a deterministic function of (logits, parameters, u):
//
//   z_i = logit_i / temperature                       (IEEE fp32 division, as the reference kernels)
//   mass_i = floor(exp(z_i - max z) * 2^32)           (integer masses: every sum below is exact and independent
//                                                      of the order threads add in - same token on every run)
//   top-k : keep the k largest z (ties: lowest index first)                       [llm/sampling.py:36-42]
//   top-p : within what top-k kept, order by z descending (ties: lowest index first) and keep the smallest prefix
//           whose mass reaches top_p * total                                      [llm/sampling.py:44-55]
//   draw  : the first kept token, in ASCENDING INDEX order, whose inclusive cumulative mass is >= u * (kept mass);
//           this is the inverse-CDF walk of sample_multinomial_*_kernel (sampling_kernels.cuh:255-268).
//
// u comes from the host (value) or from a device buffer (graph-replay compatible, sample_topk_to_buf_ptr).
// The oracle restates exactly this (oracle/cpu_ref.py sample_token_u).
//
// Thresholds are found by an 8-bit-per-pass radix descent on an order-preserving integer image of z, with LDS
// histograms of counts (top-k) or of integer masses (top-p): 4 passes over the row each, integer atomics only.

#include "pgk_device.cuh"
#include "pgk_internal.h"

namespace pgk {

constexpr int SMP_THREADS = 1024;

__device__ __forceinline__ uint32_t smp_key(float z) {   // larger z -> larger key; -0 < +0 is harmless
    const uint32_t b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long smp_mass(float z, float zmax) {
    return (unsigned long long)((double)expf(z - zmax) * 4294967296.0);
}

// block-wide exclusive scan of one value per thread (1024 threads); returns this thread's base, *total gets the sum
template <class V>
__device__ __forceinline__ V smp_scan(V v, V* wave_tot /* [17] */, V* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    V inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const V o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();   // wave_tot may still be read from a previous scan
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        V acc = 0;
        for (int w = 0; w < SMP_THREADS / 64; ++w) { const V t = wave_tot[w]; wave_tot[w] = acc; acc += t; }
        wave_tot[16] = acc;
    }
    __syncthreads();
    *total = wave_tot[16];
    return wave_tot[wid] + inc - v;
}

template <class T>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const T* logits_all, int V, float temperature, int top_k, float top_p,
                                                            float u_val, const float* u_buf, int32_t* out, const int32_t* step_counter = nullptr,
                                                            int u_cap = 0, int u_stride = 0) {
    __shared__ unsigned long long hist[256];
    __shared__ unsigned long long tot64[17];
    __shared__ int tot32[17];
    __shared__ float red[16];
    __shared__ unsigned long long sel_remaining;
    __shared__ int sel_bucket, owner, last_kept;

    const T* logits = logits_all + (size_t)blockIdx.x * V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int chunk = (V + SMP_THREADS - 1) / SMP_THREADS;
    const int lo = min(tid * chunk, V), hi = min(lo + chunk, V);
    auto Z = [&](int i) { return to_f(logits[i]) / temperature; };

    // ---- 1. max ----
    float mx = -INFINITY;
    for (int i = lo; i < hi; ++i) mx = fmaxf(mx, Z(i));
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);

    // radix descent: select the key T such that the weights of keys > T sum to < target <= including T.
    // weight(i, rank-aware) is supplied by the caller; returns T and what is still needed from the ties at T.
    uint32_t Tk = 0;            // top-k threshold key (0: everything passes)
    int need_k = 0x7fffffff;    // ties at Tk that are kept (in index order)
    int tie_base_k = 0;
    const bool use_k = top_k > 0 && top_k < V;
    if (use_k) {
        uint32_t prefix = 0, mask = 0;
        unsigned long long remaining = (unsigned long long)top_k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = lo; i < hi; ++i) {
                const uint32_t key = smp_key(Z(i));
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1ull);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long acc = 0;
                int b = 255;
                for (; b > 0; --b) {
                    if (acc + hist[b] >= remaining) break;
                    acc += hist[b];
                }
                sel_bucket = b;
                sel_remaining = remaining - acc;
            }
            __syncthreads();
            prefix |= (uint32_t)sel_bucket << shift;
            mask |= 255u << shift;
            remaining = sel_remaining;
            __syncthreads();
        }
        Tk = prefix;
        need_k = (int)remaining;
        int ties = 0;
        for (int i = lo; i < hi; ++i) ties += smp_key(Z(i)) == Tk;
        int tt;
        tie_base_k = smp_scan<int>(ties, tot32, &tt);
    }
    // kept-by-k predicate needs the running tie rank: walk the chunk in index order
    auto for_each_k = [&](auto&& f) {   // f(i, z, key) for every element top-k keeps
        int rank = tie_base_k;
        for (int i = lo; i < hi; ++i) {
            const float z = Z(i);
            const uint32_t key = smp_key(z);
            bool keep = key > Tk;
            if (key == Tk) { keep = rank < need_k; ++rank; }
            if (!use_k) keep = true;
            if (keep) f(i, z, key);
        }
    };

    // ---- 3. nucleus threshold inside what top-k kept ----
    uint32_t Tf = Tk;
    int need_f = need_k;
    bool use_f = use_k;          // a final threshold exists
    if (top_p < 1.0f) {
        unsigned long long mine = 0;
        for_each_k([&](int, float z, uint32_t) { mine += smp_mass(z, mx); });
        unsigned long long S;
        smp_scan<unsigned long long>(mine, tot64, &S);
        const double want = (double)top_p * (double)S;
        unsigned long long remaining = (unsigned long long)want;
        if ((double)remaining < want) ++remaining;      // ceil
        if (remaining < 1) remaining = 1;
        uint32_t prefix = 0, mask = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for_each_k([&](int, float z, uint32_t key) {
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], smp_mass(z, mx));
            });
            __syncthreads();
            if (tid == 0) {
                unsigned long long acc = 0;
                int b = 255;
                for (; b > 0; --b) {
                    if (acc + hist[b] >= remaining) break;
                    acc += hist[b];
                }
                sel_bucket = b;
                sel_remaining = remaining - acc;
            }
            __syncthreads();
            prefix |= (uint32_t)sel_bucket << shift;
            mask |= 255u << shift;
            remaining = sel_remaining;
            __syncthreads();
        }
        Tf = prefix;
        // every tie at Tf carries the same mass: how many of them reach the target
        const float zt = (Tf & 0x80000000u) ? __uint_as_float(Tf & 0x7fffffffu) : __uint_as_float(~Tf);
        const unsigned long long qt = smp_mass(zt, mx);
        unsigned long long n = qt ? (remaining + qt - 1) / qt : 0x7fffffffull;
        if (n < 1) n = 1;
        if (n > 0x7fffffffull) n = 0x7fffffffull;
        need_f = (int)n;
        if (use_k && Tf == Tk && need_f > need_k) need_f = need_k;
        use_f = true;
    }

    // ---- 4. final kept set, its mass, the draw ----
    int tie_base_f = 0;
    if (use_f) {
        int ties = 0;
        for (int i = lo; i < hi; ++i) ties += smp_key(Z(i)) == Tf;
        int tt;
        tie_base_f = smp_scan<int>(ties, tot32, &tt);
    }
    auto for_each_final = [&](auto&& f) {
        int rank = tie_base_f;
        for (int i = lo; i < hi; ++i) {
            const float z = Z(i);
            const uint32_t key = smp_key(z);
            bool keep = key > Tf;
            if (key == Tf) { keep = rank < need_f; ++rank; }
            if (!use_f) keep = true;
            if (keep) f(i, z);
        }
    };
    unsigned long long mine = 0;
    int my_last = -1;
    for_each_final([&](int i, float z) { mine += smp_mass(z, mx); my_last = i; });
    unsigned long long total;
    const unsigned long long base = smp_scan<unsigned long long>(mine, tot64, &total);
    // u: a host value, one device float shared by the rows, or (engine) row `step % u_cap` of a ring with one column per sequence
    const float u = u_buf ? (step_counter ? u_buf[(size_t)(step_counter[0] % u_cap) * u_stride + blockIdx.x] : *u_buf) : u_val;
    const double thr = (double)u * (double)total;
    if (tid == 0) { owner = SMP_THREADS; last_kept = -1; }
    __syncthreads();
    if (my_last >= 0) {
        atomicMax(&last_kept, my_last);
        if ((double)(base + mine) >= thr) atomicMin(&owner, tid);
    }
    __syncthreads();
    if (owner == SMP_THREADS) {            // u >= 1 and rounding: the last kept token (sampling_kernels.cuh:257)
        if (tid == 0) out[blockIdx.x] = last_kept >= 0 ? last_kept : V - 1;
        return;
    }
    if (tid == owner) {
        unsigned long long cum = base;
        int pick = my_last;
        bool done = false;
        for_each_final([&](int i, float z) {
            cum += smp_mass(z, mx);
            if (!done && (double)cum >= thr) { pick = i; done = true; }
        });
        out[blockIdx.x] = pick;
    }
}

// engine entry: fp32 logits rows, uniforms from a device ring indexed by the engine's step counter
pgk_status sample_rows_ring(const float* logits, int rows, int vocab, float temperature, int top_k, float top_p, const float* u_ring,
                            int u_cap, int u_stride, const int32_t* step_counter, int32_t* out, hipStream_t st) {
    PGK_REQUIRE(logits && u_ring && step_counter && out && u_cap > 0, "sample_rows_ring: sampling state not set up");
    sample_kernel<float><<<rows, SMP_THREADS, 0, st>>>(logits, vocab, temperature, top_k, top_p, 0.f, u_ring, out, step_counter, u_cap, u_stride);
    PGK_CHECK_HIP(hipGetLastError());
    return PGK_OK;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

pgk_status pgk_sample_token(const void* logits, int rows, int vocab, pgk_dtype dt, float temperature, int top_k, float top_p,
                            float u, const float* u_buf, int32_t* out_tokens, pgk_stream s) {
    PGK_REQUIRE(logits && out_tokens, "pgk_sample_token: null pointer");
    PGK_REQUIRE(rows >= 1 && vocab >= 1, "pgk_sample_token: bad shape rows=%d vocab=%d", rows, vocab);
    PGK_REQUIRE(temperature > 0.f, "pgk_sample_token: temperature must be > 0 (use pgk_argmax for greedy), got %g", (double)temperature);
    PGK_REQUIRE(top_k >= 0 && top_p > 0.f && top_p <= 1.f, "pgk_sample_token: need top_k >= 0 and 0 < top_p <= 1 (got %d, %g)", top_k, (double)top_p);
    PGK_REQUIRE(u_buf || (u >= 0.f && u <= 1.f), "pgk_sample_token: u=%g outside [0,1]", (double)u);
    hipStream_t st = resolve_stream(s);
    PGK_DISPATCH_FLOAT(dt, "pgk_sample_token",
                       (sample_kernel<T><<<rows, SMP_THREADS, 0, st>>>((const T*)logits, vocab, temperature, top_k, top_p, u, u_buf, out_tokens)));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
