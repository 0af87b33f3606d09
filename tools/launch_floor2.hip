// Which ingredient of a fused decode kernel costs time beyond the ~1.6 us launch floor?  Graph of N
// dependent kernels; variants add: big kernarg struct, dynamic LDS + barrier, cross-kernel data
// dependency (read what the previous kernel wrote), constant-buffer read, wave reduction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { const float* in; float* out; const float* gamma; int n; int pad[25]; };
template <int V>
__global__ __launch_bounds__(256) void k(Big a) {
    extern __shared__ float sm[];
    float v = 1.0f;
    if (V >= 2) { sm[threadIdx.x] = threadIdx.x; __syncthreads(); v = sm[(threadIdx.x + 1) & 255]; }
    if (V >= 3) { for (int i = threadIdx.x; i < 1024; i += 256) v += a.in[i]; }            // cross-kernel dependency
    if (V >= 4) { for (int i = threadIdx.x; i < 1024; i += 256) v += a.gamma[i]; }         // read-only buffer
    if (V >= 5) { for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64); }
    if (V >= 6) { sm[threadIdx.x] = v; __syncthreads(); v = sm[0] + sm[64] + sm[128] + sm[192]; __syncthreads(); }
    if (threadIdx.x == 0) a.out[blockIdx.x * 4 % 1024] = v;
}
template <int V> int run(const char* name, int grid, size_t lds) {
    float *b0, *b1, *g; CK(hipMalloc(&b0, 4096 * 4)); CK(hipMalloc(&b1, 4096 * 4)); CK(hipMalloc(&g, 4096 * 4));
    CK(hipMemset(b0, 0, 4096 * 4)); CK(hipMemset(b1, 0, 4096 * 4)); CK(hipMemset(g, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 112, REP = 30;
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < N; ++i) { Big arg{(i & 1) ? b1 : b0, (i & 1) ? b0 : b1, g, 1024, {}}; k<V><<<grid, 256, lds, s>>>(arg); }
    CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(a, s));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%-64s grid=%4d  %.2f us/kernel\n", name, grid, ms * 1e3f / (N * REP));
    return 0;
}
int main() {
    for (int grid : {256, 768}) {
        run<1>("1 big kernarg struct only", grid, 0);
        run<2>("2 + 4 KB dynamic LDS, one barrier", grid, 4096);
        run<3>("3 + read 4 KB written by the previous kernel", grid, 4096);
        run<4>("4 + read 4 KB constant buffer", grid, 4096);
        run<5>("5 + wave xor-shuffle reduction", grid, 4096);
        run<6>("6 + block reduction (2 more barriers)", grid, 4096);
    }
    return 0;
}
