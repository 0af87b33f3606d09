"""Pre-allocated decode / prefill buffers (reference: src/pygpukit/llm/buffers.py:25-621).  Field names
follow the reference so strategy code reads the same; MoE fields are omitted (out of scope)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import TYPE_CHECKING

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.factory import zeros

if TYPE_CHECKING:
    from pygpukit_amd.llm.config import TransformerConfig


@dataclass
class DecodeBuffers:
    hidden: GPUArray           # [1, hidden]
    q: GPUArray                # [1, Hq, D]
    k: GPUArray                # [1, Hkv, D]
    v: GPUArray                # [1, Hkv, D]
    attn_out: GPUArray         # [Hq, 1, D]
    mlp_gate: GPUArray         # [1, I]
    mlp_up: GPUArray           # [1, I]
    mlp_down: GPUArray         # [1, hidden]
    q_proj_out: GPUArray       # [1, Hq*D]
    k_proj_out: GPUArray       # [1, Hkv*D]
    v_proj_out: GPUArray       # [1, Hkv*D]
    o_proj_out: GPUArray       # [1, hidden]
    q_t: GPUArray              # [Hq, 1, D]
    cos: GPUArray              # [1, D]
    sin: GPUArray              # [1, D]
    embed_out: GPUArray        # [1, hidden]
    residual: GPUArray         # [1, hidden]
    norm_out: GPUArray         # [1, hidden]
    q_2d: GPUArray | None = None
    k_2d: GPUArray | None = None
    q_flat: GPUArray | None = None
    k_flat: GPUArray | None = None
    position_buf: GPUArray | None = None     # [1] int32
    qkv_proj_out: GPUArray | None = None     # [1, q+k+v]
    gate_up_out: GPUArray | None = None      # [1, 2I]
    q_view: GPUArray | None = None
    k_view: GPUArray | None = None
    v_view: GPUArray | None = None
    gate_view: GPUArray | None = None
    up_view: GPUArray | None = None
    logits: GPUArray | None = None           # [1, vocab]
    sampled_token: GPUArray | None = None    # [1] int32
    random_val: GPUArray | None = None       # [1] float32
    token_id_buf: GPUArray | None = None     # [1] int32
    context_len_buf: GPUArray | None = None  # [1] int32
    max_batch_size: int = 0
    hidden_batch: GPUArray | None = None
    logits_batch: GPUArray | None = None
    token_ids_batch_buf: GPUArray | None = None
    start_position_batch_buf: GPUArray | None = None

    @classmethod
    def allocate(cls, config: "TransformerConfig", dtype: str = "float16", use_qk_norm: bool = False,
                 vocab_size: int | None = None, max_batch_size: int = 0, moe_config: dict | None = None) -> "DecodeBuffers":
        if moe_config is not None:
            raise NotImplementedError("MoE decode buffers are out of scope")
        H, Hq, Hkv, D, I = config.hidden_size, config.num_heads, config.num_kv_heads, config.head_dim, config.intermediate_size
        qd, kd = Hq * D, Hkv * D
        z = lambda *s: zeros(s, dtype)  # noqa: E731
        qkv = z(1, qd + 2 * kd)
        gate_up = z(1, 2 * I)
        b = cls(hidden=z(1, H), q=z(1, Hq, D), k=z(1, Hkv, D), v=z(1, Hkv, D), attn_out=z(Hq, 1, D), mlp_gate=z(1, I),
                mlp_up=z(1, I), mlp_down=z(1, H), q_proj_out=z(1, qd), k_proj_out=z(1, kd), v_proj_out=z(1, kd),
                o_proj_out=z(1, H), q_t=z(Hq, 1, D), cos=z(1, D), sin=z(1, D), embed_out=z(1, H), residual=z(1, H),
                norm_out=z(1, H), position_buf=zeros((1,), "int32"), qkv_proj_out=qkv, gate_up_out=gate_up,
                q_view=qkv.narrow(0, qd), k_view=qkv.narrow(qd, kd), v_view=qkv.narrow(qd + kd, kd),
                gate_view=gate_up.narrow(0, I), up_view=gate_up.narrow(I, I), max_batch_size=max_batch_size)
        if use_qk_norm:
            b.q_2d, b.k_2d, b.q_flat, b.k_flat = z(Hq, D), z(Hkv, D), z(Hq, D), z(Hkv, D)
        if vocab_size is not None:
            b.logits = z(1, vocab_size)
            b.sampled_token = zeros((1,), "int32")
            b.random_val = zeros((1,), "float32")
            b.token_id_buf = zeros((1,), "int32")
            b.context_len_buf = zeros((1,), "int32")
        if max_batch_size > 0:
            b.hidden_batch = z(max_batch_size, H)
            b.token_ids_batch_buf = zeros((max_batch_size,), "int32")
            b.start_position_batch_buf = zeros((1,), "int32")
            if vocab_size is not None:
                b.logits_batch = z(max_batch_size, vocab_size)
        return b


@dataclass
class PrefillBuffers:
    """Kept for API compatibility (buffers.py:476-621): the pooled allocator serves the prefill
    temporaries without driver calls after the first pass, so only the result buffers are pinned here."""

    max_seq_len: int
    hidden: GPUArray
    logits: GPUArray | None = None

    @classmethod
    def allocate(cls, config: "TransformerConfig", max_seq_len: int, dtype: str = "float16", use_qk_norm: bool = False) -> "PrefillBuffers":
        return cls(max_seq_len=max_seq_len, hidden=zeros((max_seq_len, config.hidden_size), dtype))
