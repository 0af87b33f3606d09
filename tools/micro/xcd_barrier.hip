// Microbenchmark for the next-round plan: what does a barrier among the workgroups of ONE XCD
// cost, next to a barrier over all 256 workgroups, when neither uses an agent-scope fence (no L2 write-back / L1
// invalidate) - only agent-scope relaxed atomics and sc1 loads / stores, which meet in the XCD's own L2?
// Also checks that a payload stored with sc1 before such a barrier is read fresh with sc1 after it, by a workgroup of the
// same XCD (tag mismatches are counted).  Every spin is bounded; a timeout sets err and the kernel drains.
//   hipcc --offload-arch=gfx950 -O2 -o xcd_barrier xcd_barrier.hip && ./xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int PAD = 64;            // one counter per 256-byte line
constexpr unsigned SPIN_MAX = 4000000u;

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until *p >= target (monotonic counter); false on timeout
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* err) {
    for (unsigned it = 0; it < SPIN_MAX; ++it) {
        if (ld_sc1(p) >= target) return true;
        if (ld_sc1(err)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    st_sc1(err, 1u);
    return false;
}

// mode 0: barrier per XCD (counter of the workgroup's XCC); mode 1: one counter for the whole grid
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr /* [9][PAD] */, unsigned* members /* [8][PAD] */, unsigned* arrive,
                                                      unsigned* xcc_of, unsigned* slots /* [grid][PAD] */, unsigned* stale, unsigned* err,
                                                      unsigned long long* ticks, int rounds, int mode) {
    __shared__ unsigned s_n, s_ok, s_peer;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) {
        xcc_of[b] = xcc;
        const unsigned my = add_agent(&members[xcc * PAD], 1u);     // my rank inside the XCD
        slots[(size_t)b * PAD + 1] = my;
        add_agent(arrive, 1u);
        s_ok = wait_ge(arrive, gridDim.x, err) ? 1u : 0u;            // one-time census: everybody is resident and counted
        s_n = mode == 0 ? ld_sc1(&members[xcc * PAD]) : gridDim.x;
    }
    __syncthreads();
    if (!s_ok) return;
    const unsigned n = s_n;
    unsigned* c = ctr + (mode == 0 ? xcc : 8u) * PAD;
    unsigned long long t0 = 0, t1 = 0;
    unsigned bad = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        // payload: every thread of the workgroup writes one sc1 dword of its slot, tagged with the round
        __hip_atomic_store(&slots[(size_t)b * PAD + 2 + (tid & 31)], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            add_agent(c, 1u);
            s_ok = wait_ge(c, n * (unsigned)(r + 1), err) ? 1u : 0u;
        }
        __syncthreads();
        if (!s_ok) return;
        // read a neighbour's payload (next block with the same blockIdx % 8: same XCD under round-robin placement)
        const int peer = (b + 8) % (int)gridDim.x;
        const unsigned v = __hip_atomic_load(&slots[(size_t)peer * PAD + 2 + (tid & 31)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < (unsigned)(r + 1)) ++bad;
    }
    if (tid == 0) { t1 = __builtin_amdgcn_s_memrealtime(); ticks[b] = t1 - t0; }
    if (bad) atomicAdd(stale, bad);
}

int main() {
    const int grid = 256, rounds = 2000;
    unsigned *ctr, *members, *arrive, *xcc_of, *slots, *stale, *err;
    unsigned long long* ticks;
    CK(hipMalloc(&ctr, 9 * PAD * 4)); CK(hipMalloc(&members, 8 * PAD * 4)); CK(hipMalloc(&arrive, 256)); CK(hipMalloc(&xcc_of, grid * 4));
    CK(hipMalloc(&slots, (size_t)grid * PAD * 4)); CK(hipMalloc(&stale, 256)); CK(hipMalloc(&err, 256)); CK(hipMalloc(&ticks, grid * 8));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(ctr, 0, 9 * PAD * 4)); CK(hipMemset(members, 0, 8 * PAD * 4)); CK(hipMemset(arrive, 0, 256));
        CK(hipMemset(slots, 0, (size_t)grid * PAD * 4)); CK(hipMemset(stale, 0, 256)); CK(hipMemset(err, 0, 256)); CK(hipMemset(ticks, 0, grid * 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        barrier_kernel<<<grid, 256>>>(ctr, members, arrive, xcc_of, slots, stale, err, ticks, rounds, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> hx(grid), hm(8 * PAD);
        std::vector<unsigned long long> ht(grid);
        unsigned hstale = 0, herr = 0;
        CK(hipMemcpy(hx.data(), xcc_of, grid * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), members, 8 * PAD * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht.data(), ticks, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hstale, stale, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        int same = 0;
        for (int b = 0; b + 8 < grid; ++b) same += hx[b] == hx[b + 8];
        std::sort(ht.begin(), ht.end());
        printf("mode %d (%s): kernel %.3f ms for %d rounds = %.2f us per round (host events); in-kernel median %.2f us per round (100 MHz clock)\n", mode,
               mode == 0 ? "barrier per XCD" : "one barrier over the grid", ms, rounds, ms * 1e3 / rounds, ht[grid / 2] / 100.0 / rounds);
        printf("   members per XCC:"); for (int x = 0; x < 8; ++x) printf(" %u", hm[x * PAD]);
        printf("   blocks b and b+8 on the same XCC: %d of %d   stale payload reads: %u   timeout: %u\n", same, grid - 8, hstale, herr);
    }
    return 0;
}
