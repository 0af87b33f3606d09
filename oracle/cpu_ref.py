"""CPU oracle for the PyGPUkit LLM-inference hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy restatement of the reference's CPU/NumPy path for the ops and
the model forward on the hot path.  It exists to CHECK the
HIP implementation; it is never imported by the product package ``pygpukit_amd``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.

Pinning: every function below that has a runnable counterpart in the reference's
``CPUSimulationBackend`` path is checked bit-for-bit (fp32) against golden vectors
that were produced by importing the reference itself in the build container
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  Functions whose reference implementation exists
only as a CUDA kernel (no CPU branch: kv-cache scatter, fixed-cache SDPA, embedding
lookup, fp8 GEMV, fused swiglu / rmsnorm_residual, argmax tie-break) restate the
kernel source and are pinned by equivalences against the runnable path
(e.g. "decode step t == row t of a length-(t+1) prefill"); those are marked
[kernel-defined] below.

All citations are relative to /root/reference/.
"""

from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# bf16 <-> fp32 (storage convention: bf16 travels as uint16)
# ---------------------------------------------------------------------------


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bits.

    Follows src/pygpukit/core/array.py:386-395 (GPUArray.astype(bfloat16)).
    """
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    """bf16 bits -> fp32.  Follows src/pygpukit/core/array.py:373-376 and
    src/pygpukit/llm/models/causal.py:62-71 (_to_float32_logits)."""
    return (np.asarray(b).astype(np.uint32) << 16).view(np.float32)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 value, returned widened to fp32."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# ---------------------------------------------------------------------------
# Elementwise / norms / activations  (CPU branches of pygpukit.ops)
# ---------------------------------------------------------------------------


def matmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/matmul/generic.py:79-90 (_matmul_cpu): np.matmul."""
    return np.matmul(a, b)


def transpose(a: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/matmul/generic.py:149-152 (_transpose_cpu): contiguous copy of a.T."""
    return a.T.copy()


def add(a, b):
    """src/pygpukit/ops/elementwise.py (_add_cpu): a + b."""
    return a + b


def mul(a, b):
    """src/pygpukit/ops/elementwise.py (_mul_cpu): a * b."""
    return a * b


def rmsnorm(x: np.ndarray, gamma: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """src/pygpukit/ops/nn/norm.py:173-188 (_rmsnorm_cpu).  x is [rows, features]."""
    rms = np.sqrt(np.mean(x**2, axis=1, keepdims=True) + eps)
    return (x / rms) * gamma


def layernorm(x: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """src/pygpukit/ops/nn/norm.py:79-97 (_layernorm_cpu): population variance."""
    mean = x.mean(axis=1, keepdims=True)
    var = x.var(axis=1, keepdims=True)
    normalized = (x - mean) / np.sqrt(var + eps)
    return normalized * gamma + beta


def silu(x: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/nn/activation.py:87-92 (_silu_cpu)."""
    return x / (1.0 + np.exp(-x))


def gelu(a: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/nn/activation.py:40-48 (_gelu_cpu): tanh approximation."""
    x = a.astype(np.float32) if a.dtype in [np.float16] else a
    c1 = 0.7978845608
    c2 = 0.044715
    result = x * 0.5 * (1 + np.tanh(c1 * (x + c2 * x**3)))
    return result.astype(a.dtype)


# --- the rest of ops.basic (pinned by fixture G6, tests/golden/gen_golden.py gen_basic) ---------------------------
def unary(name: str, x: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/unary.py:38-260 (_exp_cpu ... _neg_cpu) and ops/nn/activation.py sigmoid / tanh / relu2:
    the NumPy expression each CPU branch evaluates."""
    f = {"exp": np.exp, "log": np.log, "relu": lambda v: np.maximum(v, 0), "sin": np.sin, "cos": np.cos, "sqrt": np.sqrt,
         "rsqrt": lambda v: 1.0 / np.sqrt(v), "abs": np.abs, "neg": lambda v: -v,
         "sigmoid": lambda v: 1.0 / (1.0 + np.exp(-v)), "tanh": np.tanh, "relu2": lambda v: np.maximum(v, 0) ** 2}[name]
    return f(x).astype(x.dtype)


def reduce_all(name: str, x: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/reduction.py:38-130,227-268: sum / mean / max / min -> shape [1] in the input dtype;
    argmax -> int64 [1] (np.argmax: lowest index on ties)."""
    if name == "argmax":
        return np.array([np.argmax(x)], dtype=np.int64)
    return np.array([{"sum": np.sum, "mean": np.mean, "max": np.max, "min": np.min}[name](x)], dtype=x.dtype)


def softmax_last(x: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/reduction.py:179-185 (_softmax_cpu_nd): max-subtracted softmax over the last axis."""
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def sum_axis(x: np.ndarray, axis: int) -> np.ndarray:
    """src/pygpukit/ops/reduction.py:298-300."""
    return np.sum(x, axis=axis)


def clamp(x: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """src/pygpukit/ops/elementwise.py:275-276."""
    return np.clip(x, lo, hi)


def where(cond: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """src/pygpukit/ops/elementwise.py:305-308."""
    return np.where(cond.astype(bool), a, b)


def swiglu(gate: np.ndarray, up: np.ndarray) -> np.ndarray:
    """[kernel-defined] native/ops/nn/fused_kernels.cuh:36-100: silu(gate) * up, fp32 math."""
    return silu(gate) * up


def rmsnorm_residual(x, residual, gamma, eps=1e-5):
    """[kernel-defined] native/ops/nn/fused_kernels.cuh:233-296: rmsnorm(x + residual) * gamma."""
    return rmsnorm(x + residual, gamma, eps)


def bias_add(out: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """[kernel-defined] native/ops/nn/elementwise_kernels.cuh:22-68: row-broadcast bias add.
    (The reference CPU branch raises AttributeError, src/pygpukit/ops/nn/linear.py:59.)"""
    return out + bias[None, :]


# ---------------------------------------------------------------------------
# RoPE
# ---------------------------------------------------------------------------


def precompute_freqs_cis(head_dim: int, max_seq_len: int, theta: float = 10000.0):
    """src/pygpukit/llm/layers/rope.py:13-24."""
    freqs = 1.0 / (theta ** (np.arange(0, head_dim, 2, dtype=np.float32) / head_dim))
    t = np.arange(max_seq_len, dtype=np.float32)
    freqs = np.outer(t, freqs)
    cos = np.cos(freqs)
    sin = np.sin(freqs)
    cos = np.concatenate([cos, cos], axis=-1)
    sin = np.concatenate([sin, sin], axis=-1)
    return cos, sin


def rope(q: np.ndarray, k: np.ndarray, cos: np.ndarray, sin: np.ndarray):
    """src/pygpukit/ops/nn/rope.py:49-89 (_rope_inplace_cpu), returned instead of in-place.

    q [S,Hq,D], k [S,Hk,D], cos/sin [S,D]; rotate-half with table index d < D/2.
    """
    q = q.copy()
    k = k.copy()
    half = q.shape[2] // 2
    c = cos[:, None, :half]
    s = sin[:, None, :half]
    for t in (q, k):
        t0 = t[:, :, :half].copy()
        t1 = t[:, :, half:].copy()
        t[:, :, :half] = t0 * c - t1 * s
        t[:, :, half:] = t1 * c + t0 * s
    return q, k


# ---------------------------------------------------------------------------
# Attention
# ---------------------------------------------------------------------------


def sdpa_causal(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float = 0.0) -> np.ndarray:
    """src/pygpukit/ops/nn/attention.py:89-131 (_sdpa_causal_cpu).

    q [H,q_len,D], k/v [H,kv_len,D].  Quirk reproduced on purpose: with scale<=0 the
    reference sets scale = 1.0/np.sqrt(head_dim), an np.float64 scalar, so scores,
    softmax and P.V are float64 and are cast back with .astype(q.dtype).
    """
    n_heads, q_len, head_dim = q.shape
    kv_len = k.shape[1]
    if scale <= 0:
        scale = 1.0 / np.sqrt(head_dim)
    scores = np.matmul(q, k.transpose(0, 2, 1)) * scale
    causal_offset = kv_len - q_len
    for i in range(q_len):
        max_attend = causal_offset + i + 1
        if max_attend < kv_len:
            scores[:, i, max_attend:] = -np.inf
    scores_max = scores.max(axis=-1, keepdims=True)
    exp_scores = np.exp(scores - scores_max)
    weights = exp_scores / exp_scores.sum(axis=-1, keepdims=True)
    output = np.matmul(weights, v)
    return output.astype(q.dtype)


def sdpa_causal_fixed_cache(q, k_cache, v_cache, context_len: int, scale: float = 0.0):
    """[kernel-defined] native/ops/nn/attention_kernels.cuh:32-148 with the host wrapper
    native/ops/nn/attention/sdpa_causal.inl:736-772: attention of q [H,q_len,D] over the
    first ``context_len`` rows of a fixed cache [Hc,max_seq,D]; causal offset
    ``context_len - q_len``.  Hc may be H (GQA-expanded cache, reference layout) or a
    divisor of H (un-expanded cache; kv_head = head // (H/Hc))."""
    H = q.shape[0]
    Hc = k_cache.shape[0]
    rep = H // Hc
    k = k_cache[:, :context_len]
    v = v_cache[:, :context_len]
    if rep > 1:
        k = np.repeat(k, rep, axis=0)
        v = np.repeat(v, rep, axis=0)
    return sdpa_causal(q, k, v, scale)


# ---------------------------------------------------------------------------
# Layout shuffles
# ---------------------------------------------------------------------------


def concat_axis0(a, b):
    """src/pygpukit/ops/tensor.py:52-57."""
    return np.concatenate([a, b], axis=0)


def repeat_interleave_axis1(x, repeats: int):
    """src/pygpukit/ops/tensor.py:104-109."""
    return np.repeat(x, repeats, axis=1)


def transpose_3d_021(x):
    """src/pygpukit/ops/tensor.py:168-172: [d0,d1,d2] -> [d1,d0,d2]."""
    return np.transpose(x, (1, 0, 2)).copy()


def split_qkv_batch(qkv, q_dim, k_dim, v_dim):
    """[kernel-defined] native/ops/nn/memory_kernels.cuh (split_qkv_batch_*): column split."""
    return (
        qkv[:, :q_dim].copy(),
        qkv[:, q_dim : q_dim + k_dim].copy(),
        qkv[:, q_dim + k_dim : q_dim + k_dim + v_dim].copy(),
    )


# ---------------------------------------------------------------------------
# Embedding / KV cache  [kernel-defined: no CPU branch in src/pygpukit/ops/embedding.py]
# ---------------------------------------------------------------------------


def paged_attention_v1(q, k_cache, v_cache, block_tables, context_lens, scale: float = 0.0):
    """native/ops/attention/paged_attention.cuh:46-200: per (sequence, head) softmax(q . K^T * scale) . V over the first
    context_len rows reached through the block table; GQA by kv_head = head // (Hq // Hkv).
    q [num_seqs, Hq, D]; caches [num_blocks, Hkv, block_size, D]; returns fp32 [num_seqs, Hq, D]."""
    num_seqs, hq, d = q.shape
    _, hkv, bs, _ = k_cache.shape
    if scale <= 0:
        scale = 1.0 / np.sqrt(d)
    out = np.zeros((num_seqs, hq, d), np.float32)
    for sidx in range(num_seqs):
        ctx = int(context_lens[sidx])
        pages = [int(b) for b in block_tables[sidx, :(ctx + bs - 1) // bs]]
        for h in range(hq):
            kvh = h // (hq // hkv)
            k = np.concatenate([k_cache[b, kvh] for b in pages], axis=0)[:ctx].astype(np.float64)
            v = np.concatenate([v_cache[b, kvh] for b in pages], axis=0)[:ctx].astype(np.float64)
            sc = (k @ q[sidx, h].astype(np.float64)) * scale
            p = np.exp(sc - sc.max())
            out[sidx, h] = ((p / p.sum()) @ v).astype(np.float32)
    return out


def paged_cache_write(k_new, v_new, k_cache, v_cache, slot_mapping):
    """paged_attention.cuh:206-283 (copy_to_paged_cache / reshape_and_cache): row t goes to block slot//bs, offset slot%bs."""
    bs = k_cache.shape[2]
    for t, slot in enumerate(slot_mapping):
        if slot < 0:
            continue
        k_cache[slot // bs, :, slot % bs, :] = k_new[t]
        v_cache[slot // bs, :, slot % bs, :] = v_new[t]


def prepare_position_ids(seq_start, seq_ctx, is_prefill, input_lens, total_tokens):
    """native/ops/batch/continuous_batching.cuh:139-165."""
    pos = np.zeros(total_tokens, np.int32)
    for b in range(len(seq_start)):
        for i in range(int(input_lens[b])):
            pos[seq_start[b] + i] = i if is_prefill[b] else seq_ctx[b]
    return pos


def embedding_lookup(embed: np.ndarray, token_id: int) -> np.ndarray:
    """native/ops/nn/embedding_kernels.cuh:27-66: out[0,:] = embed[token_id,:]."""
    return embed[token_id : token_id + 1].copy()


def kv_cache_update_gqa(new_kv: np.ndarray, cache: np.ndarray, num_heads: int, position: int):
    """native/ops/nn/kv_cache_kernels.cuh:176-246: scatter new_kv [1,Hkv,D] into
    cache [Hc,max_seq,D] at ``position``; head h takes kv head h // (Hc/Hkv)
    (Hc == num_heads is the reference's expanded layout; Hc == Hkv is un-expanded)."""
    Hc = cache.shape[0]
    Hkv = new_kv.shape[1]
    rep = Hc // Hkv
    for h in range(Hc):
        cache[h, position, :] = new_kv[0, h // rep, :]


def kv_cache_prefill_gqa(new_kv: np.ndarray, cache: np.ndarray, num_heads: int, start_pos: int = 0):
    """native/ops/nn/kv_cache_kernels.cuh:330-423: rows start_pos..start_pos+S-1."""
    S = new_kv.shape[0]
    Hc = cache.shape[0]
    Hkv = new_kv.shape[1]
    rep = Hc // Hkv
    for h in range(Hc):
        cache[h, start_pos : start_pos + S, :] = new_kv[:, h // rep, :]


# ---------------------------------------------------------------------------
# Sampling
# ---------------------------------------------------------------------------


def sample_token(logits: np.ndarray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
                 rng: np.random.Generator | None = None) -> int:
    """src/pygpukit/llm/sampling.py:12-63.  temperature == 0 -> np.argmax(probs)
    (first max wins) after softmax and optional top-k / top-p masks."""
    if temperature != 1.0 and temperature > 0:
        logits = logits / temperature
    logits_max = logits.max()
    exp_logits = np.exp(logits - logits_max)
    probs = exp_logits / exp_logits.sum()
    if top_k > 0 and top_k < len(probs):
        top_k_indices = np.argsort(probs)[-top_k:]
        mask = np.zeros_like(probs, dtype=bool)
        mask[top_k_indices] = True
        probs = np.where(mask, probs, 0.0)
        probs = probs / probs.sum()
    if top_p < 1.0:
        sorted_indices = np.argsort(probs)[::-1]
        sorted_probs = probs[sorted_indices]
        cumsum = np.cumsum(sorted_probs)
        cutoff_idx = np.searchsorted(cumsum, top_p) + 1
        cutoff_idx = min(cutoff_idx, len(sorted_probs))
        mask = np.zeros_like(probs, dtype=bool)
        mask[sorted_indices[:cutoff_idx]] = True
        probs = np.where(mask, probs, 0.0)
        probs = probs / probs.sum()
    if temperature == 0:
        return int(np.argmax(probs))
    if rng is None:
        return int(np.random.choice(len(probs), p=probs))
    return int(rng.choice(len(probs), p=probs))


def sample_token_u(logits: np.ndarray, temperature: float, top_k: int, top_p: float, u: float, return_margin: bool = False):
    """[build-defined; semantics of the reference's HOST sampler src/pygpukit/llm/sampling.py:12-63 (top-k, then a true
    nucleus, then the draw) made a deterministic function of the uniform number u, with the inverse-CDF walk in
    ascending index order of native/ops/sampling/sampling_kernels.cuh:255-268.  The reference's device top-k / top-p
    kernels are racy / approximate (sampling_kernels.cuh:419-497, 640-680) and cannot serve as a specification.]
    Integer masses floor(exp(z - max) * 2^32) make every sum exact; restated by csrc/ops_sampling.hip.
    return_margin: also return how far (relative to the kept mass) u * total is from the nearest decision boundary."""
    z = (np.asarray(logits, dtype=np.float32) / np.float32(temperature)).astype(np.float32)
    V = z.size
    q = np.floor(np.exp(z - z.max()).astype(np.float32).astype(np.float64) * 4294967296.0).astype(np.uint64)
    order = np.lexsort((np.arange(V), -z.astype(np.float64)))      # z descending, index ascending among ties
    kept = np.ones(V, bool)
    if 0 < top_k < V:
        kept[:] = False
        kept[order[:top_k]] = True
    if top_p < 1.0:
        o = order[kept[order]]
        cum = np.cumsum(q[o].astype(np.float64))                    # < 2^53: exact
        target = max(float(np.ceil(np.float64(np.float32(top_p)) * cum[-1])), 1.0)
        n = int(np.searchsorted(cum, target, side="left")) + 1
        kept[:] = False
        kept[o[:n]] = True
    idx = np.nonzero(kept)[0]
    cum = np.cumsum(q[idx].astype(np.float64))
    thr = np.float64(np.float32(u)) * cum[-1]
    pos = int(np.searchsorted(cum, thr, side="left"))
    pos = min(pos, idx.size - 1)
    tok = int(idx[pos])
    if return_margin:
        below = cum[pos - 1] if pos > 0 else -np.inf
        return tok, float(min(cum[pos] - thr, thr - below) / cum[-1])
    return tok


def argmax_lowest_index(logits: np.ndarray) -> int:
    """Greedy token: np.argmax semantics (lowest index among ties), which is what the
    reference's host sampler applies (src/pygpukit/llm/sampling.py:60-61)."""
    return int(np.argmax(logits))


# ---------------------------------------------------------------------------
# FP8 E4M3 (OCP) weights with 128x128 block scales
# ---------------------------------------------------------------------------

_FP8_TABLE = None


def fp8_e4m3_table() -> np.ndarray:
    """src/pygpukit/llm/quant.py:292-320 (_get_fp8_e4m3_table): 256-entry E4M3 -> fp32;
    0x7F / 0xFF are NaN here (the CUDA LUT native/ops/matmul/gemv/w8a16_bf16/fp8.cuh:57-122
    maps them to +-480, so synthetic weights never use those two codes)."""
    global _FP8_TABLE
    if _FP8_TABLE is None:
        table = np.zeros(256, dtype=np.float32)
        for i in range(256):
            sign = (i >> 7) & 1
            exp = (i >> 3) & 0xF
            mant = i & 0x7
            if exp == 0xF and mant == 0x7:
                table[i] = np.nan
            elif exp == 0:
                value = (mant / 8.0) * (2.0**-6)
                table[i] = -value if sign else value
            else:
                value = (1.0 + mant / 8.0) * (2.0 ** (exp - 7))
                table[i] = -value if sign else value
        _FP8_TABLE = table
    return _FP8_TABLE


def dequantize_fp8_e4m3_block(fp8_bytes: np.ndarray, scale_inv: np.ndarray,
                              block_size=(128, 128)) -> np.ndarray:
    """src/pygpukit/llm/quant.py:323-366: table lookup then per-block scale (bf16 bits or fp32)."""
    table = fp8_e4m3_table()
    f32 = table[fp8_bytes.ravel()].reshape(fp8_bytes.shape)
    H, W = f32.shape
    bh, bw = block_size
    if scale_inv.dtype != np.float32:
        if scale_inv.dtype == np.uint16:
            scale_f32 = bf16_bits_to_f32(scale_inv)
        else:
            scale_f32 = scale_inv.astype(np.float32)
    else:
        scale_f32 = scale_inv
    r = f32.reshape(H // bh, bh, W // bw, bw) * scale_f32[:, np.newaxis, :, np.newaxis]
    return r.reshape(H, W)


def dequantize_fp8_e4m3_block_w8a16_gemm(fp8_bytes: np.ndarray, scale_inv: np.ndarray, block_size=(128, 128)) -> np.ndarray:
    """[kernel-defined] native/ops/matmul/gemm/w8a16_bf16/sm120/w8a16_gemm.cu:187-203: the reference's w8a16 GEMM (every
    M > 1 product of a LinearFP8: prefill and batched decode) dequantises each weight as lut[code] * scale in fp32 and
    ROUNDS IT TO BF16 (`smB[...] = __float2bfloat16(dequant)`) before the bf16 MMA.  Its M = 1 GEMV
    (gemv/w8a16_bf16/sm120/fp8_opt.cuh:62-122) keeps the fp32 product (gemv_fp8_bf16 below).  Returned widened to fp32."""
    return bf16_round(dequantize_fp8_e4m3_block(fp8_bytes, scale_inv, block_size))


def quantize_fp8_e4m3_block(w: np.ndarray, block_size=(128, 128)):
    """Synthetic-weight quantiser used by tests/bench:
    per 128x128 block scale = absmax/448 rounded to bf16, codes = nearest E4M3 value of
    w/scale, never 0x7F/0xFF.  Not a reference function (the reference only loads
    pre-quantised checkpoints, src/pygpukit/llm/loader.py:228-252); its inverse is
    dequantize_fp8_e4m3_block above.  Returns (codes uint8 [H,W], scale_inv bf16-bits [H/bh,W/bw])."""
    H, W = w.shape
    bh, bw = block_size
    blocks = w.reshape(H // bh, bh, W // bw, bw).astype(np.float32)
    absmax = np.abs(blocks).max(axis=(1, 3))
    scale = np.where(absmax > 0, absmax / 448.0, 1.0).astype(np.float32)
    scale_bits = f32_to_bf16_bits(scale)
    scale = bf16_bits_to_f32(scale_bits)
    x = blocks / scale[:, None, :, None]
    table = fp8_e4m3_table()[:0x7F]  # non-negative finite codes 0x00..0x7E (max 448)
    mag = np.minimum(np.abs(x), 448.0)
    idx = np.searchsorted(table, mag.ravel(), side="left").reshape(mag.shape)
    idx = np.clip(idx, 0, 0x7E)
    lo = np.clip(idx - 1, 0, 0x7E)
    pick_lo = np.abs(table[lo] - mag) <= np.abs(table[idx] - mag)
    # ties -> even mantissa code
    tie = np.abs(table[lo] - mag) == np.abs(table[idx] - mag)
    pick_lo = np.where(tie, (lo % 2) == 0, pick_lo)
    code = np.where(pick_lo, lo, idx).astype(np.uint8)
    code = np.where(x < 0, code | 0x80, code).astype(np.uint8)
    code = np.where(code == 0x80, 0, code).astype(np.uint8)  # no negative zero needed
    return code.reshape(H, W), scale_bits


def _rne_e4m3_codes(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> OCP e4m3 code (|x| clamped to 448; negative zero kept as 0x80,
    as the hardware conversion leaves it)."""
    table = fp8_e4m3_table()[:0x7F]
    mag = np.minimum(np.abs(x), np.float32(448.0))
    idx = np.clip(np.searchsorted(table, mag.ravel(), side="left").reshape(mag.shape), 0, 0x7E)
    lo = np.clip(idx - 1, 0, 0x7E)
    d_lo, d_hi = np.abs(table[lo] - mag), np.abs(table[idx] - mag)
    pick_lo = np.where(d_lo == d_hi, (lo % 2) == 0, d_lo < d_hi)
    code = np.where(pick_lo, lo, idx).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def quantize_fp8_rows(x: np.ndarray):
    """[build-defined; the reference's auto-quantising matmul_fp8 (src/pygpukit/ops/matmul/fp8.py:20-70) has no
    native side in the checkout] activation quantiser of the fp8 x fp8 GEMM: per (row, 128-k block)
    scale = absmax/448 in fp32 (1 for an all-zero block), code = RNE e4m3 of x/scale (fp32 division).
    x fp32 [M,K]; returns (codes uint8 [M,K], scale fp32 [M,K/128])."""
    M, K = x.shape
    xb = x.astype(np.float32).reshape(M, K // 128, 128)
    absmax = np.abs(xb).max(axis=2)
    scale = np.where(absmax > 0, absmax / np.float32(448.0), np.float32(1.0)).astype(np.float32)
    q = (xb / scale[:, :, None]).astype(np.float32)
    return _rne_e4m3_codes(q).reshape(M, K), scale


def gemm_fp8_blockwise(a_codes: np.ndarray, a_scale: np.ndarray, w_codes: np.ndarray, w_scale_bits: np.ndarray) -> np.ndarray:
    """[build-defined; restates the blockwise-scaled contract of gemm_fp8_fp8_blockwise_sm120,
    src/pygpukit/ops/matmul/fp8.py:288-343]  C[m][n] = sum_kb sa[m][kb] * sw[n/128][kb] * sum_{k in kb} A[m][k] W[n][k],
    evaluated in float64.  a_scale fp32 [M,K/128]; w_scale_bits bf16 bits [ceil(N/128),K/128].  Returns fp32 [M,N]."""
    table = fp8_e4m3_table().astype(np.float64)
    M, K = a_codes.shape
    N = w_codes.shape[0]
    a = table[a_codes].reshape(M, K // 128, 128) * a_scale.astype(np.float64)[:, :, None]
    if w_scale_bits.dtype == np.uint16:
        sw_rows = np.repeat(bf16_bits_to_f32(w_scale_bits).astype(np.float64), 128, axis=0)[:N]   # [N, K/128]
    else:
        sw_rows = w_scale_bits.astype(np.float64)                  # already expanded per row: fp32 [N, K/128]
    w = table[w_codes].reshape(N, K // 128, 128) * sw_rows[:, :, None]
    return (a.reshape(M, K) @ w.reshape(N, K).T).astype(np.float32)


def gemv_bf16(a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
    """[kernel-defined] native/ops/matmul/gemv/bf16_bf16/sm120/bf16_opt.cuh:56-127:
    C[N] = A[K] . B[N,K]^T, bf16 in, fp32 accumulate, bf16 out.  Returns fp32 (unrounded)
    so callers can apply the 1e-2 relative-error bar of tests/test_gemv_correctness.py:144-149."""
    a = bf16_bits_to_f32(a_bits).astype(np.float64)
    b = bf16_bits_to_f32(b_bits).astype(np.float64)
    return (b @ a).astype(np.float32)


def gemv_fp8_bf16(a_bits: np.ndarray, w_codes: np.ndarray, scale_bits: np.ndarray) -> np.ndarray:
    """[kernel-defined] native/ops/matmul/gemv/w8a16_bf16/sm120/fp8_opt.cuh:62-122:
    C[N] = A[K] . dequant(B_fp8[N,K], scale[N/128,K/128])^T, fp32 accumulate."""
    w = dequantize_fp8_e4m3_block(w_codes, scale_bits).astype(np.float64)
    a = bf16_bits_to_f32(a_bits).astype(np.float64)
    return (w @ a).astype(np.float32)


# ---------------------------------------------------------------------------
# Model forward (restates llm/layers/* and llm/models/causal.py on the CPU path)
# ---------------------------------------------------------------------------


class RefLinear:
    """src/pygpukit/llm/layers/linear.py:25-99 (LinearBF16 on the CPU backend):
    y = matmul(x, transpose(W)) with the transposed copy cached (:59-60, :94)."""

    def __init__(self, weight: np.ndarray, bias: np.ndarray | None = None):
        self.weight = weight
        self.bias = bias
        self._weight_t = None

    def __call__(self, x):
        if self._weight_t is None:
            self._weight_t = transpose(self.weight)
        y = matmul(x, self._weight_t)
        if self.bias is not None:
            y = bias_add(y, self.bias)
        return y


class RefLinearFP8A8:
    """[build-defined] Linear on the fp8 x fp8 path (weight_format 2 of include/pgk_hip.h): the bf16 activations are
    quantised per (row, 128 k) by quantize_fp8_rows and multiplied with the block-scaled e4m3 weight by
    gemm_fp8_blockwise.  `scale_rows` is the 128x128 block scale expanded to one fp32 row per weight row, so a
    row slice of a fused (qkv / gate_up) matrix keeps the scales of the block it was quantised in."""

    def __init__(self, codes: np.ndarray, scale_rows: np.ndarray):
        self.codes, self.scale_rows = codes, scale_rows

    def __call__(self, x):
        a_codes, a_scale = quantize_fp8_rows(bf16_round(np.asarray(x, dtype=np.float32)))
        return gemm_fp8_blockwise(a_codes, a_scale, self.codes, self.scale_rows)


def fp8a8_linears(mats: list) -> list:
    """Quantise the row-concatenation of `mats` (as the engine fuses q|k|v and gate|up) per 128x128 block and return
    one RefLinearFP8A8 per input matrix."""
    fused = np.concatenate(mats, axis=0).astype(np.float32)
    n = fused.shape[0]
    pad = (-n) % 128
    codes, sbits = quantize_fp8_e4m3_block(np.concatenate([fused, np.zeros((pad, fused.shape[1]), np.float32)], axis=0))
    rows = np.repeat(bf16_bits_to_f32(sbits), 128, axis=0)[:n]
    out, r = [], 0
    for m in mats:
        out.append(RefLinearFP8A8(codes[r:r + m.shape[0]], rows[r:r + m.shape[0]]))
        r += m.shape[0]
    return out


class RefNorm:
    """src/pygpukit/llm/layers/norm.py:18-39."""

    def __init__(self, weight, bias=None, norm_type="rmsnorm", eps=1e-5):
        self.weight, self.bias, self.norm_type, self.eps = weight, bias, norm_type, eps

    def __call__(self, x):
        if self.norm_type == "rmsnorm":
            return rmsnorm(x, self.weight, self.eps)
        return layernorm(x, self.weight, self.bias, self.eps)


class RefAttention:
    """src/pygpukit/llm/layers/attention.py:43-277 (Attention.__call__/_forward_gpu)."""

    def __init__(self, q_proj, k_proj, v_proj, o_proj, *, num_heads, num_kv_heads, head_dim,
                 use_rope=True, rope_theta=10000.0, max_position_embeddings=2048,
                 q_norm: RefNorm | None = None, k_norm: RefNorm | None = None):
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (
            RefLinear(q_proj), RefLinear(k_proj), RefLinear(v_proj), RefLinear(o_proj))
        self.num_heads, self.num_kv_heads, self.head_dim = num_heads, num_kv_heads, head_dim
        self.num_kv_groups = num_heads // num_kv_heads
        self.q_norm, self.k_norm = q_norm, k_norm
        self.use_rope = use_rope
        if use_rope:
            self._cos, self._sin = precompute_freqs_cis(head_dim, max_position_embeddings, rope_theta)

    def __call__(self, x, position_ids, past_kv=None, use_cache=False):
        S = x.shape[0]
        q = self.q_proj(x).reshape(S, self.num_heads, self.head_dim)
        k = self.k_proj(x).reshape(S, self.num_kv_heads, self.head_dim)
        v = self.v_proj(x).reshape(S, self.num_kv_heads, self.head_dim)
        if self.q_norm is not None:
            q = self.q_norm(q.reshape(S * self.num_heads, self.head_dim)).reshape(q.shape)
        if self.k_norm is not None:
            k = self.k_norm(k.reshape(S * self.num_kv_heads, self.head_dim)).reshape(k.shape)
        if self.use_rope:
            cos = self._cos[position_ids].astype(np.float32)
            sin = self._sin[position_ids].astype(np.float32)
            q, k = rope(q, k, cos, sin)
        if past_kv is not None:
            k = concat_axis0(past_kv[0], k)
            v = concat_axis0(past_kv[1], v)
        present = (k, v) if use_cache else None
        if self.num_kv_groups > 1:
            ke = repeat_interleave_axis1(k, self.num_kv_groups)
            ve = repeat_interleave_axis1(v, self.num_kv_groups)
        else:
            ke, ve = k, v
        out = sdpa_causal(transpose_3d_021(q), transpose_3d_021(ke), transpose_3d_021(ve))
        out = transpose_3d_021(out).reshape(S, self.num_heads * self.head_dim)
        return self.o_proj(out), present


class RefMLP:
    """src/pygpukit/llm/layers/mlp.py:25-98."""

    def __init__(self, activation, *, fc1=None, fc2=None, gate=None, up=None, down=None):
        self.activation = activation
        if activation == "gelu":
            self.fc1, self.fc2 = RefLinear(fc1), RefLinear(fc2)
        else:
            self.gate_proj, self.up_proj, self.down_proj = RefLinear(gate), RefLinear(up), RefLinear(down)

    def __call__(self, x):
        if self.activation == "gelu":
            return self.fc2(gelu(self.fc1(x)))
        gate = silu(self.gate_proj(x))
        up = self.up_proj(x)
        return self.down_proj(mul(gate, up))


class RefBlock:
    """src/pygpukit/llm/layers/block.py:18-57."""

    def __init__(self, attn_norm, attn, mlp_norm, mlp):
        self.attn_norm, self.attn, self.mlp_norm, self.mlp = attn_norm, attn, mlp_norm, mlp

    def __call__(self, x, position_ids, past_kv=None, use_cache=False):
        residual = x
        x = self.attn_norm(x)
        a, present = self.attn(x, position_ids, past_kv, use_cache)
        x = add(residual, a)
        residual = x
        x = self.mlp(self.mlp_norm(x))
        x = add(residual, x)
        return x, present


class RefModel:
    """src/pygpukit/llm/models/causal.py:79-255 (CausalTransformerModel: __call__,
    get_logits, generate) on fp32 NumPy arrays."""

    def __init__(self, embed_tokens, blocks, final_norm, lm_head=None, position_embed=None):
        self.embed_tokens, self.blocks, self.final_norm = embed_tokens, blocks, final_norm
        self._lm_head, self.position_embed = lm_head, position_embed
        self._lm_head_t = None

    def __call__(self, input_ids, position_ids=None, past_key_values=None, use_cache=False):
        S = len(input_ids)
        if position_ids is None:
            if past_key_values is not None and past_key_values[0] is not None:
                past_len = past_key_values[0][0].shape[0]
                position_ids = list(range(past_len, past_len + S))
            else:
                position_ids = list(range(S))
        hidden = self.embed_tokens[input_ids]
        if self.position_embed is not None:
            hidden = hidden + self.position_embed[position_ids]
        hidden = hidden.astype(self.embed_tokens.dtype)
        presents = []
        for i, block in enumerate(self.blocks):
            past = past_key_values[i] if past_key_values else None
            hidden, present = block(hidden, position_ids, past, use_cache)
            presents.append(present)
        hidden = self.final_norm(hidden)
        return (hidden, presents) if use_cache else (hidden, None)

    def get_logits(self, hidden):
        if self._lm_head_t is None:
            lm = self._lm_head if self._lm_head is not None else self.embed_tokens
            self._lm_head_t = transpose(lm)
        return matmul(hidden, self._lm_head_t)

    def generate(self, input_ids, max_new_tokens=20, temperature=1.0, top_k=50, top_p=0.9,
                 eos_token_id=None, return_logits=False):
        """causal.py:179-239 (use_cache=True, gpu_sampling=False branch)."""
        tokens = list(input_ids)
        step_logits = []
        hidden, past = self(tokens, use_cache=True)
        logits = self.get_logits(hidden)
        last = logits[-1].astype(np.float32)
        step_logits.append(last)
        nxt = sample_token(last, temperature, top_k, top_p)
        tokens.append(nxt)
        if eos_token_id is not None and nxt == eos_token_id:
            return (tokens, step_logits) if return_logits else tokens
        for _ in range(max_new_tokens - 1):
            hidden, past = self([nxt], past_key_values=past, use_cache=True)
            logits = self.get_logits(hidden)
            last = logits[-1].astype(np.float32)
            step_logits.append(last)
            nxt = sample_token(last, temperature, top_k, top_p)
            tokens.append(nxt)
            if eos_token_id is not None and nxt == eos_token_id:
                break
        return (tokens, step_logits) if return_logits else tokens


# ---------------------------------------------------------------------------
# Synthetic weights (shared by golden generation, tests, smoke and bench)
# ---------------------------------------------------------------------------

QWEN3_0_6B = dict(vocab_size=151936, hidden_size=1024, num_layers=28, num_heads=16, num_kv_heads=8,
                  head_dim=128, intermediate_size=3072, rope_theta=1e6, norm_eps=1e-6)
LLAMA3_8B = dict(vocab_size=128256, hidden_size=4096, num_layers=32, num_heads=32, num_kv_heads=8,
                 head_dim=128, intermediate_size=14336, rope_theta=5e5, norm_eps=1e-5)
GPT2_SMALL = dict(vocab_size=50257, hidden_size=768, num_layers=12, num_heads=12, num_kv_heads=12,
                  head_dim=64, intermediate_size=3072, max_position_embeddings=1024, norm_eps=1e-5)


def make_qwen3_weights(cfg: dict, seed: int = 0, bf16: bool = True, std: float = 0.02) -> dict:
    """Random-init Qwen3-shaped weights, N(0, std^2) from np.random.default_rng(seed), in the
    fixed draw order embed, then per layer q,k,v,o,gate,up,down (SURVEY.md section 8d /
    Appendix B).  With bf16=True every matrix is rounded to bf16 (RNE) and returned widened
    to fp32, which is what both the oracle and (as bf16 bits) the GPU consume.  Norm gammas are 1."""
    rng = np.random.default_rng(seed)
    H, D, I, V = cfg["hidden_size"], cfg["head_dim"], cfg["intermediate_size"], cfg["vocab_size"]
    Hq, Hkv, L = cfg["num_heads"], cfg["num_kv_heads"], cfg["num_layers"]

    def W(*s):
        w = rng.standard_normal(s, dtype=np.float32) * np.float32(std)
        return bf16_round(w) if bf16 else w

    out = {"embed": W(V, H), "layers": []}
    for _ in range(L):
        out["layers"].append(dict(
            q=W(Hq * D, H), k=W(Hkv * D, H), v=W(Hkv * D, H), o=W(H, Hq * D),
            gate=W(I, H), up=W(I, H), down=W(H, I),
            attn_norm=np.ones(H, np.float32), mlp_norm=np.ones(H, np.float32),
            q_norm=np.ones(D, np.float32), k_norm=np.ones(D, np.float32)))
    out["final_norm"] = np.ones(H, np.float32)
    return out


def build_qwen3_ref(cfg: dict, weights: dict, max_pos: int = 2048) -> RefModel:
    """Oracle model wired like Appendix B of SURVEY.md (Qwen3: RMSNorm eps, QK-norm, GQA,
    SwiGLU, RoPE theta, tied lm_head)."""
    eps = cfg["norm_eps"]
    blocks = []
    for lw in weights["layers"]:
        attn = RefAttention(lw["q"], lw["k"], lw["v"], lw["o"], num_heads=cfg["num_heads"],
                            num_kv_heads=cfg["num_kv_heads"], head_dim=cfg["head_dim"],
                            use_rope=True, rope_theta=cfg["rope_theta"], max_position_embeddings=max_pos,
                            q_norm=RefNorm(lw["q_norm"], None, "rmsnorm", eps) if "q_norm" in lw else None,
                            k_norm=RefNorm(lw["k_norm"], None, "rmsnorm", eps) if "k_norm" in lw else None)
        mlp = RefMLP("silu", gate=lw["gate"], up=lw["up"], down=lw["down"])
        blocks.append(RefBlock(RefNorm(lw["attn_norm"], None, "rmsnorm", eps), attn,
                               RefNorm(lw["mlp_norm"], None, "rmsnorm", eps), mlp))
    return RefModel(weights["embed"], blocks, RefNorm(weights["final_norm"], None, "rmsnorm", eps))


def build_qwen3_ref_fp8a8(cfg: dict, weights: dict, max_pos: int = 2048) -> RefModel:
    """build_qwen3_ref with every projection on the fp8 x fp8 path (prefill of weight_format 2).  Note the padding
    caveat of fp8a8_linears: a fused matrix whose row count is not a multiple of 128 is quantised with zero rows
    appended, which leaves the block absmax - and so the codes - unchanged."""
    model = build_qwen3_ref(cfg, weights, max_pos)
    for blk, lw in zip(model.blocks, weights["layers"]):
        blk.attn.q_proj, blk.attn.k_proj, blk.attn.v_proj = fp8a8_linears([lw["q"], lw["k"], lw["v"]])
        (blk.attn.o_proj,) = fp8a8_linears([lw["o"]])
        blk.mlp.gate_proj, blk.mlp.up_proj = fp8a8_linears([lw["gate"], lw["up"]])
        (blk.mlp.down_proj,) = fp8a8_linears([lw["down"]])
    return model


def make_gpt2_weights(cfg: dict, seed: int = 0, std: float = 0.02) -> dict:
    """GPT-2-shaped fp32 weights (config 1): draw order wte, wpe, then per layer q,k,v,o,fc1,fc2.
    LayerNorm gamma=1, beta=0; no linear biases (reference CPU bias_add raises)."""
    rng = np.random.default_rng(seed)
    H, I, V, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["max_position_embeddings"]

    def W(*s):
        return rng.standard_normal(s, dtype=np.float32) * np.float32(std)

    out = {"wte": W(V, H), "wpe": W(P, H), "layers": []}
    for _ in range(cfg["num_layers"]):
        out["layers"].append(dict(q=W(H, H), k=W(H, H), v=W(H, H), o=W(H, H), fc1=W(I, H), fc2=W(H, I)))
    return out


def build_gpt2_ref(cfg: dict, weights: dict) -> RefModel:
    H, eps = cfg["hidden_size"], cfg["norm_eps"]
    one, zero = np.ones(H, np.float32), np.zeros(H, np.float32)
    blocks = []
    for lw in weights["layers"]:
        attn = RefAttention(lw["q"], lw["k"], lw["v"], lw["o"], num_heads=cfg["num_heads"],
                            num_kv_heads=cfg["num_kv_heads"], head_dim=cfg["head_dim"], use_rope=False)
        mlp = RefMLP("gelu", fc1=lw["fc1"], fc2=lw["fc2"])
        blocks.append(RefBlock(RefNorm(one, zero, "layernorm", eps), attn,
                               RefNorm(one, zero, "layernorm", eps), mlp))
    return RefModel(weights["wte"], blocks, RefNorm(one, zero, "layernorm", eps),
                    position_embed=weights["wpe"])
