// Do two kernels that depend on each other only through device memory run side by side?
//   (1) two branches of ONE hipGraph (stream capture with fork / join), (2) two plain streams, (3) two graphs on two streams.
// The "consumer" spins (bounded) on a word the "producer" sets; if the runtime serialises consumer-before-producer the
// spin times out.  Prints per-scenario wall time and the number of timeouts.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dual_probe.hip -o /tmp/dual_probe && /tmp/dual_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned* flag, unsigned val, int work) {
    // a little work first so that the consumer has time to start
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)work) { __builtin_amdgcn_s_sleep(8); }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag + (blockIdx.x & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)val;
}
__global__ void consumer(const unsigned* flag, unsigned target, unsigned* timeouts, unsigned long long* waited) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    if (threadIdx.x < 64) {
        for (int spin = 0; spin < (1 << 15); ++spin) {
            unsigned v = __hip_atomic_load(flag + (threadIdx.x & 7) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if ((int)(v - target) >= 0) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && threadIdx.x == 0) atomicAdd(timeouts, 1u);
        if (threadIdx.x == 0 && blockIdx.x == 0) *waited = __builtin_amdgcn_s_memrealtime() - t0;
    }
    __syncthreads();
}

int main() {
    unsigned *flags, *timeouts; unsigned long long* waited;
    CK(hipMalloc(&flags, 65536)); CK(hipMalloc(&timeouts, 4)); CK(hipMalloc(&waited, 8));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int NP = 256, NC = 256, PAIRS = 20;
    auto report = [&](const char* name, double ms) {
        unsigned to = 0; unsigned long long w = 0;
        hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost); hipMemcpy(&w, waited, 8, hipMemcpyDeviceToHost);
        printf("%-46s %8.3f ms   timeouts %u   last wait %.2f us\n", name, ms, to, w / 100.0);
    };
    for (int order = 0; order < 2; ++order) {
        // ---- (1) one graph, two branches; `order` = which branch is enqueued first during capture ----
        CK(hipMemset(flags, 0, 65536)); CK(hipMemset(timeouts, 0, 4));
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeRelaxed));
        CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0));
        for (int i = 0; i < PAIRS; ++i) {
            // chain A: producer_i ; chain B: consumer_i waits for producer_i.  (consumer_i+1 follows consumer_i in its branch)
            if (order == 0) { producer<<<NP, 256, 0, sa>>>(flags + i * 128, 0, 300); consumer<<<NC, 256, 0, sb>>>(flags + i * 128, NP, timeouts, waited); }
            else { consumer<<<NC, 256, 0, sb>>>(flags + i * 128, NP, timeouts, waited); producer<<<NP, 256, 0, sa>>>(flags + i * 128, 0, 300); }
        }
        CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
        CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ex, sa)); CK(hipStreamSynchronize(sa));     // warm (counters now at NP: reset)
        CK(hipMemset(flags, 0, 65536)); CK(hipMemset(timeouts, 0, 4));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ex, sa)); CK(hipStreamSynchronize(sa));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        report(order == 0 ? "(1) one graph, two branches, producer first" : "(1) one graph, two branches, consumer first", ms);
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
        // ---- (2) two plain streams ----
        CK(hipMemset(flags, 0, 65536)); CK(hipMemset(timeouts, 0, 4));
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < PAIRS; ++i) {
            if (order == 0) { producer<<<NP, 256, 0, sa>>>(flags + i * 128, 0, 300); consumer<<<NC, 256, 0, sb>>>(flags + i * 128, NP, timeouts, waited); }
            else { consumer<<<NC, 256, 0, sb>>>(flags + i * 128, NP, timeouts, waited); producer<<<NP, 256, 0, sa>>>(flags + i * 128, 0, 300); }
        }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        report(order == 0 ? "(2) two streams, producer first" : "(2) two streams, consumer first", ms);
        // ---- (3) two graphs (one chain each) on two streams ----
        hipGraph_t ga, gb; hipGraphExec_t xa, xb;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < PAIRS; ++i) producer<<<NP, 256, 0, sa>>>(flags + i * 128, 0, 300);
        CK(hipStreamEndCapture(sa, &ga)); CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(sb, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < PAIRS; ++i) consumer<<<NC, 256, 0, sb>>>(flags + i * 128, NP, timeouts, waited);
        CK(hipStreamEndCapture(sb, &gb)); CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
        CK(hipMemset(flags, 0, 65536)); CK(hipMemset(timeouts, 0, 4));
        t0 = std::chrono::steady_clock::now();
        if (order == 0) { CK(hipGraphLaunch(xa, sa)); CK(hipGraphLaunch(xb, sb)); } else { CK(hipGraphLaunch(xb, sb)); CK(hipGraphLaunch(xa, sa)); }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        report(order == 0 ? "(3) two graphs on two streams, producer first" : "(3) two graphs on two streams, consumer first", ms);
        hipGraphExecDestroy(xa); hipGraphDestroy(ga); hipGraphExecDestroy(xb); hipGraphDestroy(gb);
    }
    return 0;
}
