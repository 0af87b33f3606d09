// Micro-benchmark: per-kernel cost of dependent trivial kernels on one stream, eager vs hipGraph
// (captured and manually built), to calibrate the decode step's launch floor on this box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny256(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
int main() {
    int* d; CK(hipMalloc(&d, 4096 * 4)); CK(hipMemset(d, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 200, REP = 20;
    auto report = [&](const char* name, float ms_total, int launches) { printf("%-44s %.3f us per kernel\n", name, ms_total * 1e3f / launches); };
    for (int grid : {1, 256, 1024}) {
        // eager
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, s));
            for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) { if (grid == 1) tiny<<<1, 64, 0, s>>>(d); else tiny256<<<grid, 256, 0, s>>>(d); }
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b)); char nm[64]; snprintf(nm, 64, "eager grid=%d", grid); report(nm, ms, N * REP);
        // captured graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < N; ++i) { if (grid == 1) tiny<<<1, 64, 0, s>>>(d); else tiny256<<<grid, 256, 0, s>>>(d); }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, s));
            for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        }
        CK(hipEventElapsedTime(&ms, a, b)); snprintf(nm, 64, "captured graph grid=%d", grid); report(nm, ms, N * REP);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // null stream eager
    {
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, 0));
            for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) tiny256<<<256, 256, 0, 0>>>(d);
            CK(hipEventRecord(b, 0)); CK(hipStreamSynchronize(0));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b)); report("eager null-stream grid=256", ms, N * REP);
    }
    // wall-clock for eager (host-bound?)
    {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) tiny256<<<256, 256, 0, s>>>(d);
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("%-44s %.3f us per kernel (host wall)\n", "eager grid=256", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * REP));
    }
    return 0;
}
