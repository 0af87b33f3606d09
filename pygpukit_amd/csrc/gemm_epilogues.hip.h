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

}  // namespace pgk
