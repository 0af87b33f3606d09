// How fast does one CU's texture addresser take 16-byte-per-lane loads whose lanes are (a) consecutive, (b) laid out as an
// MFMA B fragment of a row-major matrix (lane l -> row l & 15, 16-byte chunk l >> 4: every lane in its own 64-byte piece)?
//   hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip && ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PATTERN>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ src, uint4* out, int iters, int row_bytes, int span_bytes) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // every wave walks its own 16 rows x 512 bytes window again and again (L2 / L1 resident after the first pass)
    const char* base = reinterpret_cast<const char*>(src) + ((size_t)(blockIdx.x * 4 + wid) * 16 * row_bytes) % span_bytes;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            size_t off;
            if (PATTERN == 0) off = (size_t)(s & 7) * 1024 + lane * 16;                                  // 1 KiB contiguous per instruction
            else if (PATTERN == 1) off = (size_t)(lane & 15) * row_bytes + s * 64 + (lane >> 4) * 16;    // 16x16x32 B fragment
            else off = (size_t)(lane & 31) * row_bytes + s * 32 + (lane >> 5) * 16;                      // 32x32x16 B fragment
            const uint4 v = *reinterpret_cast<const uint4*>(base + off + (size_t)(it & 3) * 512);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
}

int main() {
    const int row_bytes = 2048, span = 64 << 20;
    uint4 *src, *out;
    CK(hipMalloc(&src, span + (1 << 20)));
    CK(hipMemset(src, 1, span + (1 << 20)));
    CK(hipMalloc(&out, 1 << 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 64;
    for (int wgs : {256, 1024}) {
        for (int pat = 0; pat < 3; ++pat) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(a));
                if (pat == 0) probe<0><<<wgs, 256>>>(src, out, iters, row_bytes, span);
                else if (pat == 1) probe<1><<<wgs, 256>>>(src, out, iters, row_bytes, span);
                else probe<2><<<wgs, 256>>>(src, out, iters, row_bytes, span);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            const double bytes = (double)wgs * 4 * iters * 8 * 1024;
            printf("wgs %4d pattern %d: %.3f ms  %.2f TB/s  %.1f cycles per wave-instruction per CU at 2.4 GHz\n", wgs, pat, best, bytes / best / 1e9,
                   best * 1e-3 * 2.4e9 / ((double)wgs / 256 * 4 * iters * 8));
        }
    }
    return 0;
}
