// How long does a workgroup take to pull X KiB of never-seen bytes from HBM when every load is issued up front?
// (What bounds a "one round trip" kernel: the latency, or what one CU can keep in flight?)
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NL>
__global__ __launch_bounds__(256) void pull(const uint4* __restrict__ src, uint4* out, size_t wg_stride16) {
    const uint4* p = src + (size_t)blockIdx.x * wg_stride16 + threadIdx.x;
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = p[(size_t)i * 256];
    uint4 a = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NL; ++i) { a.x ^= v[i].x; a.y ^= v[i].y; a.z ^= v[i].z; a.w ^= v[i].w; }
    if (a.x == 0x12345u) out[threadIdx.x] = a;
}

template <int NL>
int run(const uint4* src, uint4* out, size_t span16, int wgs) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t per_wg = (size_t)NL * 256;
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int rep = 0; rep < reps; ++rep) {
        const size_t off = ((size_t)rep * wgs * per_wg) % (span16 - (size_t)wgs * per_wg);   // fresh bytes every launch
        CK(hipEventRecord(a));
        pull<NL><<<wgs, 256>>>(src + off, out, per_wg);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("wgs %4d  %4d KiB per workgroup: avg %.2f us  best %.2f us  (%.2f TB/s at best)\n", wgs, NL * 4, sum / (reps - 2) * 1e3, best * 1e3,
           (double)wgs * NL * 4096 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t span = (size_t)3 << 30;
    uint4 *src, *out;
    CK(hipMalloc(&src, span));
    CK(hipMemset(src, 1, span));
    CK(hipMalloc(&out, 1 << 16));
    for (int wgs : {128, 256, 512}) {
        run<4>(src, out, span / 16, wgs);
        run<8>(src, out, span / 16, wgs);
        run<16>(src, out, span / 16, wgs);
        run<32>(src, out, span / 16, wgs);
        run<48>(src, out, span / 16, wgs);
    }
    return 0;
}
