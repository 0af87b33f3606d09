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
// Thresholds are found by an 8-bit-per-pass radix descent on a 64-bit key: the order-preserving integer image of z in
// the high word, ~index in the low word.  Keys are unique, so "ties: lowest index first" IS the key order and no tie
// bookkeeping exists; the low-word passes only run when the threshold bucket still holds more than it needs.
//
// Two launch shapes:
//   * 1 <= top_k <= 1024 (the decode default, top_k = 50): stage 1 cuts the row into 4096-token slices, one workgroup
//     per slice holds its slice in registers and writes its k best keys; stage 2 (one workgroup per row) selects the
//     k best of those, and does nucleus cut and draw on <= 1024 survivors in LDS.  ~10 us per draw at V = 151 936.
//   * top_k = 0 on a long row: slice maxima, then per-slice exact integer masses.  Pure multinomial picks the slice
//     that owns u * S and scans its 4096 tokens.  A nucleus (top_p < 1) is cut inside the best keys of the row
//     (SMP_WHOLE_K = 256) when their mass reaches top_p * S (decided on the device; the usual case for a language
//     model at top_p <= 0.95); a flatter row
//     falls through to the whole-row kernel, which the launch sequence always contains and which exits at once
//     for rows already drawn.
//   * otherwise (top_k > 1024, short rows, or no scratch buffer as inside a user's stream capture): one 1024-thread
//     workgroup walks the row with coalesced loads: max, count / mass descent, indexed draw.

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

constexpr int SMP_THREADS = 1024;
constexpr int SMP_SLICE_THREADS = 256;
constexpr int SMP_SLICE_E = 16;                                   // keys a stage-1 thread holds
constexpr int SMP_SLICE = SMP_SLICE_THREADS * SMP_SLICE_E;        // 4096 tokens per stage-1 workgroup
constexpr int SMP_MAX_K = 1024;
constexpr int SMP_WHOLE_K = 256;   // candidates the whole-vocabulary nucleus path keeps (per slice and overall)

using u64 = unsigned long long;

__device__ __forceinline__ uint32_t smp_key(float z) {   // larger z -> larger key; -0 < +0 is harmless
    const uint32_t b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float smp_unkey(uint32_t k) { return (k & 0x80000000u) ? __uint_as_float(k & 0x7fffffffu) : __uint_as_float(~k); }
__device__ __forceinline__ u64 smp_key64(float z, int i) { return ((u64)smp_key(z) << 32) | (uint32_t)~(uint32_t)i; }
__device__ __forceinline__ int smp_index(u64 key) { return (int)~(uint32_t)key; }
__device__ __forceinline__ u64 smp_mass(float z, float zmax) { return (u64)((double)expf(z - zmax) * 4294967296.0); }

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

struct SmpSel {
    u64 remaining, weight;
    int bucket;
    unsigned cnt;
};

// wave 0: walk the 256 buckets from the top and pick the one in which the running weight reaches `remaining`
// (bucket 0 if none does).  Four buckets per lane, one shuffle scan.
__device__ __forceinline__ void smp_pick(const u64* hist, const unsigned* cnt, u64 remaining, SmpSel* sh) {
    const int lane = threadIdx.x;
    u64 h[4], loc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = hist[255 - (lane * 4 + q)]; loc += h[q]; }
    u64 inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    const u64 m = __ballot(inc >= remaining);
    const int L = m ? __ffsll((long long)m) - 1 : 63;
    if (lane == L) {
        u64 acc = inc - loc;
        int q = 0;
        for (; q < 3; ++q) {
            if (m && acc + h[q] >= remaining) break;
            acc += h[q];
        }
        const int b = 255 - (lane * 4 + q);
        sh->bucket = b;
        sh->remaining = remaining - acc;
        sh->weight = h[q];
        sh->cnt = cnt ? cnt[b] : 0u;
    }
}

// Radix descent over this workgroup's keys.  for_each(g) must call g(key, weight_fn) once per live key of the calling
// thread.  Returns T such that {key >= T} is the smallest top set whose weight reaches `target`
// (MASS = false: weight 1 per key, i.e. the `target` largest keys; MASS = true: integer masses).
template <bool MASS, class FE>
__device__ __forceinline__ u64 smp_select(FE&& for_each, u64 target, u64* hist, unsigned* cnt, SmpSel* sh) {
    u64 prefix = 0, mask = 0, remaining = target;
    for (int shift = 56; shift >= 0; shift -= 8) {
        __syncthreads();
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            hist[t] = 0;
            if (MASS) cnt[t] = 0;
        }
        __syncthreads();
        for_each([&](u64 key, auto&& weight) {
            if ((key & mask) == prefix) {
                const int b = (int)(key >> shift) & 255;
                if (MASS) {
                    atomicAdd(&hist[b], weight());
                    atomicAdd(&cnt[b], 1u);
                } else {
                    atomicAdd(&hist[b], 1ull);
                }
            }
        });
        __syncthreads();
        if (threadIdx.x < 64) smp_pick(hist, MASS ? cnt : nullptr, remaining, sh);
        __syncthreads();
        prefix |= (u64)sh->bucket << shift;
        mask |= 255ull << shift;
        remaining = sh->remaining;
        // the whole bucket is needed (counts) / the bucket is one key (masses): every lower bit may stay 0
        if (MASS ? sh->cnt == 1u : sh->weight == remaining) break;
    }
    return prefix;
}

// In-place descending bitonic sort of N (a power of two, <= 1024) 64-bit keys in LDS by the whole workgroup.
// Ends with a barrier; the caller's writes to `a` need none before the call (the first stage starts with one).
__device__ __forceinline__ void smp_bitonic_desc(u64* a, int N) {
    if (N <= 128) {   // few keys (top_k = 50): rank by counting, two barriers instead of 28
        __syncthreads();
        const u64 mine = (int)threadIdx.x < N ? a[threadIdx.x] : 0ull;
        int r = 0;
        if (mine)
            for (int i = 0; i < N; ++i) r += a[i] > mine;
        __syncthreads();
        if (mine) a[r] = mine;          // keys are unique: a permutation of the non-zero slots, zeros stay behind them
        __syncthreads();
        return;
    }
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                const int j = i ^ stride;
                if (j > i) {
                    const u64 x = a[i], y = a[j];
                    const bool desc = (i & size) == 0;
                    if ((x < y) == desc) { a[i] = y; a[j] = x; }
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float smp_u(float u_val, const float* u_buf, const int32_t* step_counter, int u_cap, int u_stride, int row) {
    // a host value, one device float shared by the rows, or (engine) row `step % u_cap` of a ring with one column per sequence
    return u_buf ? (step_counter ? u_buf[(size_t)(step_counter[0] % u_cap) * u_stride + row] : *u_buf) : u_val;
}

// ---------------------------------------------------------------------------------------------------------------------
// top-k path, stage 1: the k best keys of one 4096-token slice
// ---------------------------------------------------------------------------------------------------------------------
// slice maxima of z (whole-vocabulary path: the integer masses need the row maximum before anything is summed)
template <class T>
__global__ __launch_bounds__(SMP_SLICE_THREADS) void sample_slice_max_kernel(const T* logits_all, int V, float temperature, float* smax) {
    __shared__ float red[SMP_SLICE_THREADS / 64];
    const int tid = threadIdx.x, slice = blockIdx.x, row = blockIdx.y;
    const T* logits = logits_all + (size_t)row * V;
    const int lo = slice * SMP_SLICE, n = min(V - lo, SMP_SLICE);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMP_SLICE_E; ++j) {
        const int o = tid + SMP_SLICE_THREADS * j;
        if (o < n) m = fmaxf(m, to_f(logits[lo + o]) / temperature);
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) smax[(size_t)row * gridDim.x + slice] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// WHOLE = false: the k best keys of the slice.  WHOLE = true (no top-k; smax holds the slice maxima): also the
// slice's exact integer mass, and candidates only when a nucleus will be cut (k > 0).
template <class T, bool WHOLE>
__global__ __launch_bounds__(SMP_SLICE_THREADS) void sample_slice_topk_kernel(const T* logits_all, int V, float temperature, int k,
                                                                              u64* cand /* [rows][slices][k] */, const float* smax = nullptr,
                                                                              u64* smass = nullptr) {
    __shared__ u64 hist[256];
    __shared__ SmpSel sh;
    __shared__ int out_pos;
    __shared__ u64 wsum[SMP_SLICE_THREADS / 64];
    const int tid = threadIdx.x, slice = blockIdx.x, row = blockIdx.y;
    const T* logits = logits_all + (size_t)row * V;
    const int lo = slice * SMP_SLICE, n = min(V - lo, SMP_SLICE);
    u64 key[SMP_SLICE_E];
#pragma unroll
    for (int j = 0; j < SMP_SLICE_E; ++j) {
        const int o = tid + SMP_SLICE_THREADS * j;
        key[j] = o < n ? smp_key64(to_f(logits[lo + o]) / temperature, lo + o) : 0ull;
    }
    if constexpr (WHOLE) {
        float zmax = -INFINITY;
        for (int i = 0; i < (int)gridDim.x; ++i) zmax = fmaxf(zmax, smax[(size_t)row * gridDim.x + i]);
        u64 mine = 0;
#pragma unroll
        for (int j = 0; j < SMP_SLICE_E; ++j)
            if (key[j]) mine += smp_mass(smp_unkey((uint32_t)(key[j] >> 32)), zmax);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
        if ((tid & 63) == 0) wsum[tid >> 6] = mine;
        __syncthreads();
        if (tid == 0) smass[(size_t)row * gridDim.x + slice] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (k <= 0) return;
    }
    u64 T64 = 0;
    if (n > k) {
        T64 = smp_select<false>(
            [&](auto&& g) {
#pragma unroll
                for (int j = 0; j < SMP_SLICE_E; ++j)
                    if (key[j]) g(key[j], [] { return 1ull; });
            },
            (u64)k, hist, nullptr, &sh);
    }
    if (tid == 0) out_pos = 0;
    __syncthreads();
    u64* dst = cand + ((size_t)row * gridDim.x + slice) * k;
#pragma unroll
    for (int j = 0; j < SMP_SLICE_E; ++j)
        if (key[j] && key[j] >= T64) dst[atomicAdd(&out_pos, 1)] = key[j];
    __syncthreads();
    for (int p = out_pos + tid; p < k; p += SMP_SLICE_THREADS) dst[p] = 0ull;
}

// stage 2: the k best of the slices' candidates, then nucleus cut and draw on those k keys in LDS
// WHOLE (no top-k, top_p < 1): the candidates are the 1024 best keys of the whole row; zmax and the total mass S come
// from the slice statistics.  If those 1024 keys carry the nucleus (their mass reaches top_p * S - the normal case
// for a language model's distribution) the draw is finished here and done[row] = 1; otherwise done[row] = 0 and the
// whole-row kernel that follows does the row.
template <bool WHOLE>
__global__ __launch_bounds__(SMP_THREADS) void sample_topk_draw_kernel(const u64* cand_all, int n /* slices*k */, int k, float top_p, float u_val,
                                                                       const float* u_buf, int32_t* out, const int32_t* step_counter, int u_cap,
                                                                       int u_stride, const float* smax = nullptr, const u64* smass = nullptr,
                                                                       int slices = 0, int* done = nullptr) {
    __shared__ u64 hist[256];
    __shared__ SmpSel sh;
    __shared__ u64 tot64[17];
    __shared__ u64 ka[SMP_MAX_K];
    __shared__ int nk_s, cut_s, pick_s;
    __shared__ u64 top_s;
    const int tid = threadIdx.x, row = blockIdx.x;
    const u64* cand = cand_all + (size_t)row * n;

    u64 T64 = 0;
    if (n > k) {
        T64 = smp_select<false>(
            [&](auto&& g) {
                for (int i = tid; i < n; i += SMP_THREADS) {
                    const u64 key = cand[i];
                    if (key) g(key, [] { return 1ull; });
                }
            },
            (u64)k, hist, nullptr, &sh);
    }
    if (tid == 0) { nk_s = 0; top_s = 0; cut_s = SMP_MAX_K; pick_s = SMP_MAX_K; }
    __syncthreads();
    for (int i = tid; i < n; i += SMP_THREADS) {
        const u64 key = cand[i];
        if (key && key >= T64) {
            ka[atomicAdd(&nk_s, 1)] = key;
            atomicMax(&top_s, key);
        }
    }
    __syncthreads();
    int nk = nk_s;                                  // == k (k < V is the caller's precondition)
    const float zmax = smp_unkey((uint32_t)(top_s >> 32));   // the row maximum is among the candidates in either mode
    u64 S_whole = 0;
    if constexpr (WHOLE)
        for (int i = 0; i < slices; ++i) S_whole += smass[(size_t)row * slices + i];
    if constexpr (WHOLE) {   // do the candidates carry the nucleus at all?  decided before anything is sorted
        u64 cs;
        smp_scan<u64>(tid < nk ? smp_mass(smp_unkey((uint32_t)(ka[tid] >> 32)), zmax) : 0ull, tot64, &cs);
        const double want = (double)top_p * (double)S_whole;
        u64 remaining = (u64)want;
        if ((double)remaining < want) ++remaining;
        if (remaining < 1) remaining = 1;
        const bool reach = cs >= remaining;              // block-uniform
        if (tid == 0) done[row] = reach ? 1 : 0;
        if (!reach) return;
    }
    // N = the power of two that holds the kept keys; unused slots are 0 and sort to the end
    int N = 1;
    while (N < nk) N <<= 1;
    for (int i = nk + tid; i < N; i += SMP_THREADS) ka[i] = 0ull;
    auto mass_of = [&](u64 key) { return key ? smp_mass(smp_unkey((uint32_t)(key >> 32)), zmax) : 0ull; };

    if (top_p < 1.0f) {
        // key-descending order, then the smallest prefix whose mass reaches top_p * S
        smp_bitonic_desc(ka, N);
        const u64 m = tid < nk ? mass_of(ka[tid]) : 0ull;
        u64 S;
        const u64 base = smp_scan<u64>(m, tot64, &S);
        const double want = (double)top_p * (double)(WHOLE ? S_whole : S);
        u64 remaining = (u64)want;
        if ((double)remaining < want) ++remaining;      // ceil
        if (remaining < 1) remaining = 1;
        if (tid < nk && base + m >= remaining) atomicMin(&cut_s, tid);
        __syncthreads();
        nk = min(nk, cut_s + 1);
    }
    // ascending index order == descending ~index: sort the kept keys with their halves swapped
    __syncthreads();
    for (int i = tid; i < N; i += SMP_THREADS) {
        const u64 key = i < nk ? ka[i] : 0ull;
        ka[i] = (key << 32) | (key >> 32);
    }
    smp_bitonic_desc(ka, N);
    const u64 sw = tid < nk ? ka[tid] : 0ull;                       // {~index, key of z}
    const u64 m = sw ? smp_mass(smp_unkey((uint32_t)sw), zmax) : 0ull;
    u64 total;
    const u64 base = smp_scan<u64>(m, tot64, &total);
    const double thr = (double)smp_u(u_val, u_buf, step_counter, u_cap, u_stride, row) * (double)total;
    if (tid < nk && (double)(base + m) >= thr) atomicMin(&pick_s, tid);
    __syncthreads();
    if (tid == 0) out[row] = (int)~(uint32_t)(ka[pick_s < nk ? pick_s : nk - 1] >> 32);   // u >= 1 and rounding: the last kept token
}

// ---------------------------------------------------------------------------------------------------------------------
// whole-row path: one workgroup per row
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const T* logits_all, int V, float temperature, int top_k, float top_p,
                                                            float u_val, const float* u_buf, int32_t* out, const int32_t* step_counter = nullptr,
                                                            int u_cap = 0, int u_stride = 0, const int* done = nullptr,
                                                            const float* smax = nullptr, const u64* smass = nullptr, int slices = 0) {
    if (done && done[blockIdx.x]) return;   // the sliced nucleus path already drew this row
    __shared__ u64 hist[256];
    __shared__ unsigned cnt[256];
    __shared__ SmpSel sh;
    __shared__ u64 tot64[17];
    __shared__ float red[16];
    __shared__ u64 wmass[16];
    __shared__ int wlast[16];
    __shared__ int pick_s;

    const T* logits = logits_all + (size_t)blockIdx.x * V;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto Z = [&](int i) { return to_f(logits[i]) / temperature; };

    // ---- 1. max (already known when the sliced passes ran first) ----
    float mx = -INFINITY;
    if (smax) {
        for (int i = 0; i < slices; ++i) mx = fmaxf(mx, smax[(size_t)blockIdx.x * slices + i]);
    } else {
#pragma unroll 8
        for (int i = tid; i < V; i += SMP_THREADS) mx = fmaxf(mx, Z(i));
        mx = wave_max(mx);
        if (lane == 0) red[wid] = mx;
        __syncthreads();
        mx = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    }

    // ---- 2. top-k threshold (k > 1024 or k >= V never reaches the sliced path) ----
    u64 Tk = 0;
    if (top_k > 0 && top_k < V) {
        Tk = smp_select<false>(
            [&](auto&& g) {
                for (int i = tid; i < V; i += SMP_THREADS) g(smp_key64(Z(i), i), [] { return 1ull; });
            },
            (u64)top_k, hist, nullptr, &sh);
    }

    // ---- 3. nucleus threshold inside what top-k kept ----
    u64 Tf = Tk;
    if (top_p < 1.0f) {
        u64 S = 0;
        if (smass && Tk == 0) {               // no top-k: the slice masses are the total
            for (int i = 0; i < slices; ++i) S += smass[(size_t)blockIdx.x * slices + i];
        } else {
            u64 mine = 0;
#pragma unroll 8
            for (int i = tid; i < V; i += SMP_THREADS) {
                const float z = Z(i);
                if (smp_key64(z, i) >= Tk) mine += smp_mass(z, mx);
            }
            smp_scan<u64>(mine, tot64, &S);
        }
        const double want = (double)top_p * (double)S;
        u64 remaining = (u64)want;
        if ((double)remaining < want) ++remaining;      // ceil
        if (remaining < 1) remaining = 1;
        Tf = smp_select<true>(
            [&](auto&& g) {
                for (int i = tid; i < V; i += SMP_THREADS) {
                    const float z = Z(i);
                    const u64 key = smp_key64(z, i);
                    if (key >= Tk) g(key, [&] { return smp_mass(z, mx); });
                }
            },
            remaining, hist, cnt, &sh);
    }

    // ---- 4. the draw, in ascending index order: wave w owns one contiguous 1/16 of the row, 64 tokens per step ----
    const int per_wave = ((V + 15) / 16 + 63) & ~63;
    const int wlo = min(wid * per_wave, V), whi = min(wlo + per_wave, V);
    u64 mine = 0;
    int my_last = -1;
#pragma unroll 8
    for (int i = wlo + lane; i < whi; i += 64) {
        const float z = Z(i);
        if (smp_key64(z, i) >= Tf) { mine += smp_mass(z, mx); my_last = i; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mine += __shfl_xor(mine, d, 64);
        my_last = max(my_last, __shfl_xor(my_last, d, 64));
    }
    if (lane == 0) { wmass[wid] = mine; wlast[wid] = my_last; }
    if (tid == 0) pick_s = -1;
    __syncthreads();
    u64 total = 0, base = 0;
    int last_kept = -1;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wid) base += wmass[w];
        total += wmass[w];
        last_kept = max(last_kept, wlast[w]);
    }
    const double thr = (double)smp_u(u_val, u_buf, step_counter, u_cap, u_stride, blockIdx.x) * (double)total;
    // the owner is the first wave that keeps something and whose inclusive mass reaches thr
    bool earlier = false;
    {
        u64 b = 0;
        for (int w = 0; w < wid; ++w) {
            b += wmass[w];
            earlier |= wlast[w] >= 0 && (double)b >= thr;
        }
    }
    if (!earlier && my_last >= 0 && (double)(base + mine) >= thr) {
        u64 cum = base;
        for (int i0 = wlo; i0 < whi; i0 += 64) {
            const int i = i0 + lane;
            bool keep = false;
            u64 m = 0;
            if (i < whi) {
                const float z = Z(i);
                keep = smp_key64(z, i) >= Tf;
                if (keep) m = smp_mass(z, mx);
            }
            u64 inc = m;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            const u64 hit = __ballot(keep && (double)(cum + inc) >= thr);
            if (hit) {
                if (lane == 0) pick_s = i0 + __ffsll((long long)hit) - 1;
                break;
            }
            cum += __shfl(inc, 63, 64);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = pick_s >= 0 ? pick_s : (last_kept >= 0 ? last_kept : V - 1);   // u >= 1 and rounding
}

// pure multinomial over the whole row: the slice that owns u * S, then that slice's 4096 tokens in index order
// (4 consecutive tokens per thread, one block scan)
template <class T>
__global__ __launch_bounds__(SMP_THREADS) void sample_multinomial_pick_kernel(const T* logits_all, int V, float temperature, const float* smax,
                                                                              const u64* smass, int slices, float u_val, const float* u_buf,
                                                                              int32_t* out, const int32_t* step_counter, int u_cap, int u_stride) {
    __shared__ u64 tot64[17];
    __shared__ int pick_s;
    const int tid = threadIdx.x, row = blockIdx.x;
    const T* logits = logits_all + (size_t)row * V;
    float zmax = -INFINITY;
    u64 total = 0;
    for (int i = 0; i < slices; ++i) {
        zmax = fmaxf(zmax, smax[(size_t)row * slices + i]);
        total += smass[(size_t)row * slices + i];
    }
    const double thr = (double)smp_u(u_val, u_buf, step_counter, u_cap, u_stride, row) * (double)total;
    int owner = -1;
    u64 base = 0;
    for (int i = 0; i < slices; ++i) {
        const u64 m = smass[(size_t)row * slices + i];
        if ((double)(base + m) >= thr) { owner = i; break; }
        base += m;
    }
    if (owner < 0) {                          // u >= 1 and rounding: the last token
        if (tid == 0) out[row] = V - 1;
        return;
    }
    if (tid == 0) pick_s = 0x7fffffff;
    const int lo = owner * SMP_SLICE + tid * 4;
    u64 m[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = lo + j < V ? smp_mass(to_f(logits[lo + j]) / temperature, zmax) : 0ull;
        mine += m[j];
    }
    u64 slice_total;
    u64 cum = base + smp_scan<u64>(mine, tot64, &slice_total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cum += m[j];
        if (lo + j < V && (double)cum >= thr) { atomicMin(&pick_s, lo + j); break; }
    }
    __syncthreads();
    if (tid == 0) out[row] = pick_s != 0x7fffffff ? pick_s : min(V, (owner + 1) * SMP_SLICE) - 1;
}

static inline bool smp_sliced(int top_k, int vocab) { return top_k >= 1 && top_k <= SMP_MAX_K && top_k < vocab; }
// whole-vocabulary sliced path: no top-k and a row long enough to be worth three launches
static inline bool smp_whole(int top_k, int vocab) { return top_k == 0 && vocab > 2 * SMP_SLICE; }
static inline int smp_slices(int vocab) { return (vocab + SMP_SLICE - 1) / SMP_SLICE; }

// scratch layout of the whole-vocabulary path: cand [rows][slices][SMP_WHOLE_K] u64 | smass [rows][slices] u64 | smax f32 | done i32
size_t sample_scratch_bytes(int rows, int vocab, int top_k, float top_p) {
    if (smp_sliced(top_k, vocab)) return (size_t)rows * smp_slices(vocab) * top_k * sizeof(u64);
    if (smp_whole(top_k, vocab)) {
        const size_t rs = (size_t)rows * smp_slices(vocab);
        return (top_p < 1.0f ? rs * SMP_WHOLE_K * sizeof(u64) : 0) + rs * (sizeof(u64) + sizeof(float)) + (size_t)rows * sizeof(int) + 64;
    }
    return 0;
}

template <class T>
static void sample_launch(const T* logits, int rows, int vocab, float temperature, int top_k, float top_p, float u, const float* u_buf,
                          const int32_t* step_counter, int u_cap, int u_stride, int32_t* out, void* scratch, hipStream_t st) {
    const int slices = smp_slices(vocab);
    if (smp_sliced(top_k, vocab) && scratch) {
        sample_slice_topk_kernel<T, false><<<dim3(slices, rows), SMP_SLICE_THREADS, 0, st>>>(logits, vocab, temperature, top_k, (u64*)scratch);
        sample_topk_draw_kernel<false><<<rows, SMP_THREADS, 0, st>>>((const u64*)scratch, slices * top_k, top_k, top_p, u, u_buf, out,
                                                                     step_counter, u_cap, u_stride);
    } else if (smp_whole(top_k, vocab) && scratch) {
        const size_t rs = (size_t)rows * slices;
        const bool nucleus = top_p < 1.0f;
        u64* cand = (u64*)scratch;
        u64* smass = cand + (nucleus ? rs * SMP_WHOLE_K : 0);
        float* smax = (float*)(smass + rs);
        int* done = (int*)(smax + rs);
        sample_slice_max_kernel<T><<<dim3(slices, rows), SMP_SLICE_THREADS, 0, st>>>(logits, vocab, temperature, smax);
        sample_slice_topk_kernel<T, true><<<dim3(slices, rows), SMP_SLICE_THREADS, 0, st>>>(logits, vocab, temperature, nucleus ? SMP_WHOLE_K : 0,
                                                                                            cand, smax, smass);
        if (nucleus) {
            sample_topk_draw_kernel<true><<<rows, SMP_THREADS, 0, st>>>(cand, slices * SMP_WHOLE_K, SMP_WHOLE_K, top_p, u, u_buf, out,
                                                                        step_counter, u_cap, u_stride, smax, smass, slices, done);
            sample_kernel<T><<<rows, SMP_THREADS, 0, st>>>(logits, vocab, temperature, top_k, top_p, u, u_buf, out, step_counter, u_cap, u_stride,
                                                           done, smax, smass, slices);
        } else {
            sample_multinomial_pick_kernel<T><<<rows, SMP_THREADS, 0, st>>>(logits, vocab, temperature, smax, smass, slices, u, u_buf, out,
                                                                            step_counter, u_cap, u_stride);
        }
    } else {
        sample_kernel<T><<<rows, SMP_THREADS, 0, st>>>(logits, vocab, temperature, top_k, top_p, u, u_buf, out, step_counter, u_cap, u_stride);
    }
}

// engine entry: fp32 logits rows, uniforms from a device ring indexed by the engine's step counter.  `scratch`
// (sample_scratch_bytes) lets 1 <= top_k <= 1024 take the sliced path; without it the whole-row kernel runs.
pgk_status sample_rows_ring(const float* logits, int rows, int vocab, float temperature, int top_k, float top_p, const float* u_ring,
                            int u_cap, int u_stride, const int32_t* step_counter, int32_t* out, void* scratch, hipStream_t st) {
    PGK_REQUIRE(logits && u_ring && step_counter && out && u_cap > 0, "sample_rows_ring: sampling state not set up");
    sample_launch<float>(logits, rows, vocab, temperature, top_k, top_p, 0.f, u_ring, step_counter, u_cap, u_stride, out, scratch, st);
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
    // the sliced top-k path needs a candidate buffer; a capturing stream cannot allocate, so a captured launch
    // (sample_topk_to_buf_ptr inside a user graph) takes the whole-row kernel.  The engine brings its own buffer.
    void* scratch = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    PGK_CHECK_HIP(hipStreamIsCapturing(st, &cap));
    if (const size_t bytes = sample_scratch_bytes(rows, vocab, top_k, top_p); bytes && cap == hipStreamCaptureStatusNone)
        if (pgk_status r = pgk_malloc(&scratch, bytes)) return r;
    PGK_DISPATCH_FLOAT(dt, "pgk_sample_token",
                       (sample_launch<T>((const T*)logits, rows, vocab, temperature, top_k, top_p, u, u_buf, nullptr, 0, 0, out_tokens, scratch, st)));
    if (scratch) pgk_free(scratch);   // stream-ordered reuse
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
