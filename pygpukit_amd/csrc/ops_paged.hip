// Paged KV cache and continuous-batching helpers.
//
// Reference: native/ops/attention/paged_attention.cuh:46-283 (paged_attention_v1, copy_to_paged_cache,
// reshape_and_cache), native/ops/batch/continuous_batching.cuh:70-245 (gather_embeddings,
// scatter_last_token_logits, prepare_position_ids, argmax_sample, check_eos), declarations ops.cuh:466-563.
// Cache layout is the reference's: K/V [num_blocks, num_kv_heads, block_size, head_dim]; block_tables
// [num_seqs, max_blocks_per_seq] int32; context_lens [num_seqs] int32.
//
// paged_attention_v1 here is split-KV flash-decoding over pages (the reference runs one block per (seq, head) with
// the whole score row in shared memory): a workgroup owns one (sequence, kv head, slice of pages); its 4 waves take
// pages round-robin; a 16-lane group reads one K/V row per instruction (16 bytes per lane) so a wave-instruction covers
// 4 rows; all G = Hq/Hkv query heads of the kv head are scored against each K row (the KV stream is read once per kv
// head, not once per query head); each lane group keeps its own online-softmax state, merged through LDS at the end;
// slices are merged by a second tiny kernel.  HBM-bound on the KV bytes of the live context.

#include "attn_core.hip.h"
#include "pgk_internal.h"

namespace pgk {

template <class T, int D, int G>
__global__ __launch_bounds__(256) void paged_attn_kernel(const T* q, const T* kc, const T* vc, const int32_t* block_tables,
                                                         const int32_t* context_lens, float* ws, T* out, int num_heads,
                                                         int num_kv_heads, int block_size, int max_blocks, float scale, int nsplit) {
    constexpr int LPR = D / 8;            // lanes per row (16 at D = 128, 8 at D = 64)
    constexpr int RPW = 64 / LPR;         // rows per wave-instruction
    constexpr int NG = 4;                 // partial states per workgroup after the in-wave merge: one per wave
    __shared__ float part_ml[NG][G][2];
    __shared__ float part_o[NG][G][D];

    const int split = blockIdx.x, kvh = blockIdx.y, seq = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
    const int ctx = context_lens[seq];
    const int npages = (ctx + block_size - 1) / block_size;
    const int per = (npages + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(p0 + per, npages);

    float qv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        Vec<T> raw;
        raw.load(q + ((size_t)seq * num_heads + kvh * G + g) * D + sub * 8);
        raw.to_float(qv[g]);
    }
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }
    const int32_t* table = block_tables + (size_t)seq * max_blocks;
    for (int p = p0 + wid; p < p1; p += 4) {
        const size_t base = ((size_t)table[p] * num_kv_heads + kvh) * block_size * D;
        for (int r0 = 0; r0 < block_size; r0 += RPW) {
            const int r = r0 + grp;
            const bool valid = r < block_size && p * block_size + r < ctx;
            const size_t off = base + (size_t)min(r, block_size - 1) * D + sub * 8;   // clamped, masked below
            Vec<T> kr, vr;
            kr.load(kc + off);
            vr.load(vc + off);
            float kf[8], vf[8];
            kr.to_float(kf);
            vr.to_float(vf);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(qv[g][j], kf[j], s);
                s = group_sum<LPR>(s) * scale;
                if (!valid) s = -INFINITY;
                const float mn = fmaxf(m[g], s);
                const float a = mn == -INFINITY ? 1.f : __expf(m[g] - mn);
                const float pr = mn == -INFINITY ? 0.f : __expf(s - mn);
                l[g] = l[g] * a + pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[g][j] = fmaf(o[g][j], a, pr * vf[j]);
                m[g] = mn;
            }
        }
    }
    // merge the RPW lane-group states of a wave (butterfly over the group index), then the 4 waves through LDS
#pragma unroll
    for (int step = LPR; step < 64; step <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mo = __shfl_xor(m[g], step, 64), lo = __shfl_xor(l[g], step, 64);
            const float mn = fmaxf(m[g], mo);
            const float a = m[g] == -INFINITY ? 0.f : __expf(m[g] - mn), b = mo == -INFINITY ? 0.f : __expf(mo - mn);
            l[g] = l[g] * a + lo * b;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * a + __shfl_xor(o[g][j], step, 64) * b;
            m[g] = mn;
        }
    }
    const int gi = wid;
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (sub == 0) { part_ml[gi][g][0] = m[g]; part_ml[gi][g][1] = l[g]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) part_o[gi][g][sub * 8 + j] = o[g][j];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < G * D; e += 256) {
        const int g = e / D, d = e % D;
        float mx = -INFINITY;
        for (int i = 0; i < NG; ++i) mx = fmaxf(mx, part_ml[i][g][0]);
        float lt = 0.f, ot = 0.f;
        for (int i = 0; i < NG; ++i) {
            const float w = part_ml[i][g][0] == -INFINITY ? 0.f : __expf(part_ml[i][g][0] - mx);
            lt = fmaf(part_ml[i][g][1], w, lt);
            ot = fmaf(part_o[i][g][d], w, ot);
        }
        const int head = kvh * G + g;
        if (nsplit == 1) {
            out[((size_t)seq * num_heads + head) * D + d] = from_f<T>(lt > 0.f ? ot / lt : 0.f);
        } else {
            float* rec = ws + (((size_t)seq * num_heads + head) * nsplit + split) * (D + 2);
            rec[2 + d] = ot;
            if (d == 0) { rec[0] = mx; rec[1] = lt; }
        }
    }
}

template <class T, int D>
__global__ void paged_combine_kernel(const float* ws, T* out, int nsplit) {
    const size_t sh = blockIdx.x;   // seq * num_heads + head
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        out[sh * D + d] = from_f<T>(decode_combine<D>(ws + sh * nsplit * (D + 2), nsplit, d));
}

// K_new/V_new rows -> cache slots.  slot = physical_block * block_size + offset; a negative slot is skipped (padding).
template <int ITEM>
__global__ void paged_cache_write_kernel(const char* k_new, const char* v_new, char* kc, char* vc, const int32_t* slot_mapping,
                                         int n_tokens, int num_kv_heads, int block_size, int row_bytes /* head_dim * itemsize */) {
    const int chunks = row_bytes / 16;
    const long long total = (long long)n_tokens * num_kv_heads * chunks;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % chunks);
        const int h = (int)((t / chunks) % num_kv_heads);
        const int tok = (int)(t / ((long long)chunks * num_kv_heads));
        const int slot = slot_mapping[tok];
        if (slot < 0) continue;
        const size_t src = ((size_t)tok * num_kv_heads + h) * row_bytes + (size_t)c * 16;
        const size_t dst = (((size_t)(slot / block_size) * num_kv_heads + h) * block_size + slot % block_size) * row_bytes + (size_t)c * 16;
        *reinterpret_cast<uint4*>(kc + dst) = *reinterpret_cast<const uint4*>(k_new + src);
        *reinterpret_cast<uint4*>(vc + dst) = *reinterpret_cast<const uint4*>(v_new + src);
    }
}

// logits [batch_tokens, vocab] -> out [batch, vocab]: row seq_start[b] + seq_len[b] - 1 of each sequence
__global__ void scatter_last_logits_kernel(const char* logits, char* out, const int32_t* seq_start, const int32_t* seq_lens, int row_bytes) {
    const int b = blockIdx.x;
    const size_t src = (size_t)(seq_start[b] + seq_lens[b] - 1) * row_bytes, dst = (size_t)b * row_bytes;
    for (int c = threadIdx.x * 16; c + 16 <= row_bytes; c += blockDim.x * 16)
        *reinterpret_cast<uint4*>(out + dst + c) = *reinterpret_cast<const uint4*>(logits + src + c);
    for (int c = (row_bytes & ~15) + threadIdx.x; c < row_bytes; c += blockDim.x) out[dst + c] = logits[src + c];
}

// continuous_batching.cuh:139-165: prefill tokens get their index in the sequence, decode tokens the context length
__global__ void prepare_position_ids_kernel(const int32_t* seq_start, const int32_t* seq_ctx, const int32_t* is_prefill,
                                            const int32_t* input_lens, int32_t* position_ids) {
    const int b = blockIdx.x;
    const int start = seq_start[b], ctx = seq_ctx[b], n = input_lens[b];
    const bool pf = is_prefill[b] != 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) position_ids[start + i] = pf ? i : ctx;
}

__global__ void check_eos_kernel(const int32_t* tokens, int32_t* finished, int n, int eos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) finished[i] = tokens[i] == eos ? 1 : 0;
}

// exclusive prefix sum of up to a few thousand int32 (sequence start offsets): one workgroup, serial tail on thread 0
__global__ void exclusive_cumsum_kernel(const int32_t* in, int32_t* out, int n) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < n; ++i) { const int v = in[i]; out[i] = acc; acc += v; }
    }
}

}  // namespace pgk

using namespace pgk;

extern "C" {

size_t pgk_paged_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim, int max_context) {
    int nsplit = (max_context + 511) / 512;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 32) nsplit = 32;
    return nsplit == 1 ? 0 : (size_t)num_seqs * num_heads * nsplit * (head_dim + 2) * sizeof(float);
}

pgk_status pgk_paged_attention_v1(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                                  const int32_t* context_lens, void* out, int num_seqs, int num_heads, int num_kv_heads, int head_dim,
                                  int block_size, int max_blocks_per_seq, int max_context, float scale, void* workspace, pgk_dtype dt,
                                  pgk_stream s) {
    PGK_REQUIRE(q && k_cache && v_cache && block_tables && context_lens && out, "pgk_paged_attention_v1: null pointer");
    PGK_REQUIRE(num_seqs >= 1 && num_heads >= 1 && num_kv_heads >= 1 && num_heads % num_kv_heads == 0,
                "pgk_paged_attention_v1: bad head counts (Hq=%d, Hkv=%d, seqs=%d)", num_heads, num_kv_heads, num_seqs);
    PGK_REQUIRE(head_dim == 64 || head_dim == 128, "pgk_paged_attention_v1: head_dim %d not in {64, 128}", head_dim);
    PGK_REQUIRE(block_size >= 1 && max_blocks_per_seq >= 1, "pgk_paged_attention_v1: bad paging (block_size=%d, max_blocks=%d)", block_size,
                max_blocks_per_seq);
    PGK_REQUIRE(dt == PGK_BF16 || dt == PGK_F16, "pgk_paged_attention_v1: 16-bit caches only (dtype %d)", (int)dt);
    const int G = num_heads / num_kv_heads;
    PGK_REQUIRE(G == 1 || G == 2 || G == 4 || G == 8, "pgk_paged_attention_v1: %d query heads per kv head unsupported", G);
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)head_dim);
    int nsplit = (max_context + 511) / 512;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 32) nsplit = 32;
    PGK_REQUIRE(nsplit == 1 || workspace, "pgk_paged_attention_v1: contexts beyond 512 need the workspace of pgk_paged_attention_workspace_bytes");
    hipStream_t st = resolve_stream(s);
    dim3 grid(nsplit, num_kv_heads, num_seqs);
#define PGK_PA(TT, DD, GG)                                                                                              \
    if (head_dim == DD && G == GG) {                                                                                    \
        paged_attn_kernel<TT, DD, GG><<<grid, 256, 0, st>>>((const TT*)q, (const TT*)k_cache, (const TT*)v_cache, block_tables,  \
                                                            context_lens, (float*)workspace, (TT*)out, num_heads, num_kv_heads,   \
                                                            block_size, max_blocks_per_seq, scale, nsplit);             \
        if (nsplit > 1) paged_combine_kernel<TT, DD><<<num_seqs * num_heads, DD, 0, st>>>((const float*)workspace, (TT*)out, nsplit); \
        PGK_LAUNCH_CHECK();                                                                                             \
        return PGK_OK;                                                                                                  \
    }
#define PGK_PA_T(TT) PGK_PA(TT, 128, 1) PGK_PA(TT, 128, 2) PGK_PA(TT, 128, 4) PGK_PA(TT, 128, 8) PGK_PA(TT, 64, 1) PGK_PA(TT, 64, 2) PGK_PA(TT, 64, 4) PGK_PA(TT, 64, 8)
    if (dt == PGK_BF16) { PGK_PA_T(bf16) } else { PGK_PA_T(f16) }
#undef PGK_PA_T
#undef PGK_PA
    return set_error(PGK_ERR_INVALID, "pgk_paged_attention_v1: no kernel for D=%d G=%d", head_dim, G);
}

// copy_to_paged_cache (one token per sequence) and reshape_and_cache (all prefill tokens) are the same scatter
pgk_status pgk_paged_cache_write(const void* k_new, const void* v_new, void* k_cache, void* v_cache, const int32_t* slot_mapping,
                                 int n_tokens, int num_kv_heads, int block_size, int head_dim, int itemsize, pgk_stream s) {
    PGK_REQUIRE(k_new && v_new && k_cache && v_cache && slot_mapping, "pgk_paged_cache_write: null pointer");
    PGK_REQUIRE(n_tokens >= 0 && num_kv_heads >= 1 && block_size >= 1 && (head_dim * itemsize) % 16 == 0,
                "pgk_paged_cache_write: bad shape (tokens=%d, Hkv=%d, block=%d, row bytes=%d)", n_tokens, num_kv_heads, block_size,
                head_dim * itemsize);
    if (!n_tokens) return PGK_OK;
    const long long total = (long long)n_tokens * num_kv_heads * (head_dim * itemsize / 16);
    const int blocks = (int)(ceil_div(total, 256) > 4096 ? 4096 : ceil_div(total, 256));
    paged_cache_write_kernel<16><<<blocks, 256, 0, resolve_stream(s)>>>((const char*)k_new, (const char*)v_new, (char*)k_cache, (char*)v_cache,
                                                                       slot_mapping, n_tokens, num_kv_heads, block_size, head_dim * itemsize);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_scatter_last_token_logits(const void* logits, void* out, const int32_t* seq_start, const int32_t* seq_lens, int batch,
                                         int vocab, int itemsize, pgk_stream s) {
    PGK_REQUIRE(logits && out && seq_start && seq_lens, "pgk_scatter_last_token_logits: null pointer");
    PGK_REQUIRE(batch >= 1 && vocab >= 1, "pgk_scatter_last_token_logits: bad shape");
    PGK_REQUIRE(((size_t)vocab * itemsize) % 16 == 0, "pgk_scatter_last_token_logits: rows of %zu bytes are not 16-byte multiples",
                (size_t)vocab * itemsize);
    scatter_last_logits_kernel<<<batch, 256, 0, resolve_stream(s)>>>((const char*)logits, (char*)out, seq_start, seq_lens, vocab * itemsize);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_prepare_position_ids(const int32_t* seq_start, const int32_t* seq_ctx, const int32_t* is_prefill, const int32_t* input_lens,
                                    int32_t* position_ids, int batch, pgk_stream s) {
    PGK_REQUIRE(seq_start && seq_ctx && is_prefill && input_lens && position_ids && batch >= 1, "pgk_prepare_position_ids: bad argument");
    prepare_position_ids_kernel<<<batch, 256, 0, resolve_stream(s)>>>(seq_start, seq_ctx, is_prefill, input_lens, position_ids);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_check_eos(const int32_t* tokens, int32_t* finished, int n, int eos_token_id, pgk_stream s) {
    PGK_REQUIRE(tokens && finished && n >= 1, "pgk_check_eos: bad argument");
    check_eos_kernel<<<ceil_div(n, 256), 256, 0, resolve_stream(s)>>>(tokens, finished, n, eos_token_id);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_exclusive_cumsum_i32(const int32_t* in, int32_t* out, int n, pgk_stream s) {
    PGK_REQUIRE(in && out && n >= 1, "pgk_exclusive_cumsum_i32: bad argument");
    exclusive_cumsum_kernel<<<1, 64, 0, resolve_stream(s)>>>(in, out, n);
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
