"""Attention layer (reference: src/pygpukit/llm/layers/attention.py:43-555): MHA / GQA, RoPE, optional
QK-norm (Qwen3), dynamic past-KV prefill path and fixed-cache decode paths.

MI355X-first differences (same inputs/outputs):
  * q/k/v projection weights are row views of ONE fused [q+k+v, hidden] weight (the reference holds the
    three weights plus a fused copy, attention.py:98-107);
  * activations stay in the projection's native [S, H, D] layout: the attention kernels take head/row
    strides and do GQA by indexing, so there is no transpose_3d_021 / repeat_interleave traffic
    (attention.py:261-275);
  * the fixed KV cache is UN-EXPANDED, [num_kv_heads, max_seq, D] (the reference expands it to
    num_heads, attention.py:135): half the bytes per decode step for Qwen3;
  * RoPE tables live on the device in fp32 and a position's row is a zero-copy view, not a per-call upload
    (attention.py:342-343).
"""

from __future__ import annotations

from typing import TYPE_CHECKING

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import float32
from pygpukit_amd.core.factory import from_numpy, zeros
from pygpukit_amd.ops.basic import (bias_add_inplace, concat_axis0, kv_cache_prefill_gqa, kv_cache_update_gqa, rmsnorm,
                                   rope_inplace, rope_inplace_f32table, sdpa_causal_fixed_cache, sdpa_causal_strided,
                                   split_qkv_batch)

from .linear import LinearBF16, LinearFP8
from .norm import Norm
from .rope import precompute_freqs_cis

if TYPE_CHECKING:
    from pygpukit_amd.llm.config import TransformerConfig


class Attention:
    def __init__(self, q_proj, k_proj, v_proj, o_proj, config: "TransformerConfig", q_bias: GPUArray | None = None,
                 k_bias: GPUArray | None = None, v_bias: GPUArray | None = None, o_bias: GPUArray | None = None,
                 q_norm: Norm | None = None, k_norm: Norm | None = None):
        self.config = config
        self.head_dim = config.head_dim
        self.num_heads = config.num_heads
        self.num_kv_heads: int = config.num_kv_heads
        self.num_kv_groups = config.num_kv_groups
        self.q_dim = self.num_heads * self.head_dim
        self.k_dim = self.v_dim = self.num_kv_heads * self.head_dim
        self.q_norm, self.k_norm = q_norm, k_norm

        def wrap(p, b):
            return p if isinstance(p, (LinearBF16, LinearFP8)) else LinearBF16(p, b)

        self.o_proj = wrap(o_proj, o_bias)
        self.qkv_proj: LinearBF16 | None = None
        if isinstance(q_proj, LinearFP8) or isinstance(k_proj, LinearFP8) or isinstance(v_proj, LinearFP8):
            self.q_proj, self.k_proj, self.v_proj = wrap(q_proj, q_bias), wrap(k_proj, k_bias), wrap(v_proj, v_bias)
        else:
            qw = q_proj.weight if isinstance(q_proj, LinearBF16) else q_proj
            kw = k_proj.weight if isinstance(k_proj, LinearBF16) else k_proj
            vw = v_proj.weight if isinstance(v_proj, LinearBF16) else v_proj
            qb = q_proj.bias if isinstance(q_proj, LinearBF16) else q_bias
            kb = k_proj.bias if isinstance(k_proj, LinearBF16) else k_bias
            vb = v_proj.bias if isinstance(v_proj, LinearBF16) else v_bias
            fused = concat_axis0(concat_axis0(qw, kw), vw)
            hidden = fused.shape[1]
            self.qkv_proj = LinearBF16(fused, None)
            self.q_proj = LinearBF16(fused._view(0, (self.q_dim, hidden)), qb)
            self.k_proj = LinearBF16(fused._view(self.q_dim * hidden, (self.k_dim, hidden)), kb)
            self.v_proj = LinearBF16(fused._view((self.q_dim + self.k_dim) * hidden, (self.v_dim, hidden)), vb)

        self._cos: np.ndarray | None = None
        self._sin: np.ndarray | None = None
        self._cos_gpu: GPUArray | None = None
        self._sin_gpu: GPUArray | None = None
        if config.use_rope:
            self._cos, self._sin = precompute_freqs_cis(self.head_dim, config.max_position_embeddings, config.rope_theta)

        self._k_cache: GPUArray | None = None
        self._v_cache: GPUArray | None = None
        self._max_cache_len = 0
        self._confirmed_pos = 0
        self._logical_pos = 0

    # ------------------------------------------------------------------ caches / rope tables
    def init_fixed_cache(self, max_seq_len: int, dtype: str = "float16") -> None:
        """Fixed-length KV cache [num_kv_heads, max_seq_len, head_dim] (attention.py:128-146, un-expanded here)."""
        shape = (self.num_kv_heads, max_seq_len, self.head_dim)
        self._k_cache = zeros(shape, dtype)
        self._v_cache = zeros(shape, dtype)
        self._max_cache_len = max_seq_len
        self._confirmed_pos = self._logical_pos = 0

    def _rope_rows(self, start: int, count: int) -> tuple[GPUArray, GPUArray]:
        """fp32 [count, D] views of the device-resident tables for positions start..start+count-1."""
        if self._cos_gpu is None:
            self._cos_gpu = from_numpy(self._cos.astype(np.float32))
            self._sin_gpu = from_numpy(self._sin.astype(np.float32))
        if start + count > self._cos_gpu.shape[0]:
            raise ValueError(f"position {start + count - 1} beyond max_position_embeddings {self._cos_gpu.shape[0]}")
        D = self.head_dim
        return self._cos_gpu._view(start * D, (count, D)), self._sin_gpu._view(start * D, (count, D))

    def _apply_rope(self, q3: GPUArray, k3: GPUArray, cos: GPUArray, sin: GPUArray) -> None:
        if q3.dtype == float32:
            rope_inplace(q3, k3, cos, sin)
        else:
            rope_inplace_f32table(q3, k3, cos, sin)

    # lookahead bookkeeping kept for API compatibility (attention.py:152-171)
    def set_confirmed_pos(self, pos: int) -> None:
        assert 0 <= pos <= self._max_cache_len, f"Invalid pos {pos}"
        self._confirmed_pos = self._logical_pos = pos

    def reset_lookahead(self) -> None:
        self._logical_pos = self._confirmed_pos

    def commit_lookahead(self, n_accepted: int) -> None:
        new_pos = self._confirmed_pos + n_accepted
        assert new_pos <= self._max_cache_len, f"Commit exceeds cache: {new_pos}"
        self._confirmed_pos = self._logical_pos = new_pos

    def get_confirmed_pos(self) -> int:
        return self._confirmed_pos

    # ------------------------------------------------------------------ projections
    def _project_qkv(self, x: GPUArray) -> tuple[GPUArray, GPUArray, GPUArray]:
        """q [S, q_dim], k [S, k_dim], v [S, v_dim] as separate contiguous buffers (biases applied)."""
        S = x.shape[0]
        if self.qkv_proj is not None:
            qkv = self.qkv_proj(x)
            if S == 1:
                q, k, v = qkv.narrow(0, self.q_dim), qkv.narrow(self.q_dim, self.k_dim), qkv.narrow(self.q_dim + self.k_dim, self.v_dim)
            else:
                q, k, v = (GPUArray((S, n), x.dtype) for n in (self.q_dim, self.k_dim, self.v_dim))
                split_qkv_batch(qkv, q, k, v, self.q_dim, self.k_dim, self.v_dim)
            for t, lin in ((q, self.q_proj), (k, self.k_proj), (v, self.v_proj)):
                if lin.bias is not None:
                    bias_add_inplace(t, lin.bias)
            return q, k, v
        return self.q_proj(x), self.k_proj(x), self.v_proj(x)

    def _qk_norm(self, q: GPUArray, k: GPUArray, S: int) -> None:
        D = self.head_dim
        if self.q_norm is not None:
            qf = q.view((S * self.num_heads, D))
            rmsnorm(qf, self.q_norm.weight, self.q_norm.eps, out=qf)
        if self.k_norm is not None:
            kf = k.view((S * self.num_kv_heads, D))
            rmsnorm(kf, self.k_norm.weight, self.k_norm.eps, out=kf)

    # ------------------------------------------------------------------ prefill / dynamic cache
    def __call__(self, x: GPUArray, position_ids: list[int] | None = None, past_kv: tuple | None = None,
                 use_cache: bool = False) -> tuple[GPUArray, tuple | None]:
        """x [S, hidden] -> (out [S, hidden], present_kv); past/present K,V are [kv_len, Hkv, D] (attention.py:173-277)."""
        S = x.shape[0]
        if position_ids is None:
            position_ids = list(range(S))
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_dim
        q, k, v = self._project_qkv(x)
        self._qk_norm(q, k, S)
        q3, k3, v3 = q.view((S, Hq, D)), k.view((S, Hkv, D)), v.view((S, Hkv, D))
        if self.config.use_rope:
            p0 = position_ids[0]
            if list(position_ids) == list(range(p0, p0 + S)):
                cos, sin = self._rope_rows(p0, S)
            else:
                cos = from_numpy(self._cos[position_ids].astype(np.float32))
                sin = from_numpy(self._sin[position_ids].astype(np.float32))
            self._apply_rope(q3, k3, cos, sin)
        if past_kv is not None:
            past_k, past_v = past_kv
            if not isinstance(past_k, GPUArray):
                past_k, past_v = from_numpy(past_k), from_numpy(past_v)
            k3, v3 = concat_axis0(past_k, k3), concat_axis0(past_v, v3)
        present_kv = (k3, v3) if use_cache else None
        kv_len = k3.shape[0]
        attn = GPUArray((S, Hq * D), x.dtype)
        sdpa_causal_strided(q3, k3, v3, attn, Hq, Hkv, S, kv_len, D, (D, Hq * D), (D, Hkv * D), (D, Hq * D))
        return self.o_proj(attn), present_kv

    # ------------------------------------------------------------------ fixed cache decode
    def forward_fixed_cache(self, x: GPUArray, position: int, context_len: int, *, out: GPUArray | None = None) -> GPUArray:
        """Single token x [1, hidden] against the fixed cache (attention.py:279-370); `out` = attention
        output buffer [num_heads, 1, head_dim]."""
        assert self._k_cache is not None, "Call init_fixed_cache first"
        assert x.shape[0] == 1, "forward_fixed_cache expects single token"
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_dim
        q, k, v = self._project_qkv(x)
        self._qk_norm(q, k, 1)
        q3, k3, v3 = q.view((1, Hq, D)), k.view((1, Hkv, D)), v.view((1, Hkv, D))
        if self.config.use_rope:
            cos, sin = self._rope_rows(position, 1)
            self._apply_rope(q3, k3, cos, sin)
        kv_cache_update_gqa(k3, self._k_cache, Hq, position)
        kv_cache_update_gqa(v3, self._v_cache, Hq, position)
        attn_out = out if out is not None else GPUArray((Hq, 1, D), x.dtype)
        sdpa_causal_fixed_cache(q.view((Hq, 1, D)), self._k_cache, self._v_cache, attn_out, context_len)
        return self.o_proj(attn_out.view((1, Hq * D)))

    def forward_fixed_cache_batch(self, x: GPUArray, start_position: int, context_len: int) -> GPUArray:
        """M consecutive tokens of ONE sequence (speculative verify, attention.py:372-461)."""
        assert self._k_cache is not None, "Call init_fixed_cache first"
        S = x.shape[0]
        if S == 1:
            return self.forward_fixed_cache(x, start_position, context_len)
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_dim
        q, k, v = self._project_qkv(x)
        self._qk_norm(q, k, S)
        q3, k3, v3 = q.view((S, Hq, D)), k.view((S, Hkv, D)), v.view((S, Hkv, D))
        if self.config.use_rope:
            cos, sin = self._rope_rows(start_position, S)
            self._apply_rope(q3, k3, cos, sin)
        kv_cache_prefill_gqa(k3, self._k_cache, Hq, start_position)
        kv_cache_prefill_gqa(v3, self._v_cache, Hq, start_position)
        attn = GPUArray((S, Hq * D), x.dtype)
        M = self._max_cache_len
        sdpa_causal_strided(q3, self._k_cache, self._v_cache, attn, Hq, Hkv, S, context_len, D, (D, Hq * D), (M * D, D), (D, Hq * D))
        return self.o_proj(attn)

    def forward_fixed_cache_batch_zero_alloc(self, x: GPUArray, start_position: int, context_len: int, buffers=None,
                                             rope_cos_gpu=None, rope_sin_gpu=None, start_pos_buf=None) -> GPUArray:
        """Signature-compatible with attention.py:463-555; the pooled allocator makes the plain batch path
        allocation-free in steady state, so it is used directly."""
        return self.forward_fixed_cache_batch(x, start_position, context_len)


__all__ = ["Attention"]
