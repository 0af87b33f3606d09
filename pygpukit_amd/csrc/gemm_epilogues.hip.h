// Arguments of the GEMM epilogues that finish an engine step inside the projection (ops_gemm256.hip; callers: ops_gemm.hip, engine.hip).
#pragma once
#include "pgk_device.hip.h"

namespace pgk {

// QKV projection whose tile columns are whole heads (head_dim 128): per (token row, head slot) the bf16-rounded result gets the
// per-head RMSNorm (q / k heads, when a gamma is given) and RoPE; q heads are stored to the qkv buffer, k / v heads to the KV
// cache row of the token's position - what qknorm_rope_kvwrite_kernel (engine.hip) does in a pass of its own.
struct QkvHeadArgs {
    const bf16 *q_gamma, *k_gamma;        // null: no per-head norm
    float eps;
    const float *rope_cos, *rope_sin;     // [max_seq][64]
    bf16 *kcache, *vcache;                // this layer's, this sequence's: [Hkv][max_seq][128]
    int hq, hkv, max_seq, start_pos;
};

// The 16 lanes that hold one (token row, head slot) of the bf16-rounded QKV projection, 8 dims each (sub = lane's chunk):
// the arithmetic of qknorm_rope_kvwrite_kernel, operation for operation, then the store.  All 16 lanes must be active.
__device__ __forceinline__ void qkv_head_finish(float (&x)[8], int sub, int slot, int grow, int M, const QkvHeadArgs& hd, bf16* qkv, int ldq) {
    const bool is_q = slot < hd.hq, is_k = !is_q && slot < hd.hq + hd.hkv;
    const int pos = hd.start_pos + grow;
    if (is_q || is_k) {
        const bf16* gamma = is_q ? hd.q_gamma : hd.k_gamma;
        if (gamma) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(x[j], x[j], ss);
            ss = group_sum<16>(ss);
            const float inv = 1.0f / sqrtf(ss / 128 + hd.eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] * inv * to_f(gamma[sub * 8 + j]);
        }
        const bool lo = sub < 8;
        const float* cs = hd.rope_cos + (size_t)min(pos, hd.max_seq - 1) * 64;
        const float* sn = hd.rope_sin + (size_t)min(pos, hd.max_seq - 1) * 64;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float other = xor_half<16>(x[j]);
            const int dd = (sub * 8 + j) % 64;
            o[j] = lo ? (x[j] * cs[dd] - other * sn[dd]) : (x[j] * cs[dd] + other * sn[dd]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = o[j];
    }
    if (grow >= M) return;
    Vec<bf16> ov;
    ov.from_float(x);
    if (is_q) {
        ov.store(qkv + (size_t)grow * ldq + slot * 128 + sub * 8);
    } else if (pos < hd.max_seq) {
        const int kvh = is_k ? slot - hd.hq : slot - hd.hq - hd.hkv;
        ov.store((is_k ? hd.kcache : hd.vcache) + ((size_t)kvh * hd.max_seq + pos) * 128 + sub * 8);
    }
}

}  // namespace pgk
