"""Pre-allocated decode / prefill buffers (reference: src/pygpukit/llm/buffers.py:25-621).  Field names
follow the reference so strategy code reads the same; MoE buffers are never allocated (out of scope)."""

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
    # batch decode (buffers.py:299-336): M tokens per step
    max_batch_size: int = 0
    hidden_batch: GPUArray | None = None          # [M, hidden]
    residual_batch: GPUArray | None = None        # [M, hidden]
    norm_out_batch: GPUArray | None = None        # [M, hidden]
    qkv_proj_out_batch: GPUArray | None = None    # [M, q+k+v]
    q_batch: GPUArray | None = None               # [M, Hq, D]
    k_batch: GPUArray | None = None               # [M, Hkv, D]
    v_batch: GPUArray | None = None               # [M, Hkv, D]
    q_t_batch: GPUArray | None = None             # [Hq, M, D]
    attn_out_batch: GPUArray | None = None        # [Hq, M, D]
    attn_out_t_batch: GPUArray | None = None      # [M, Hq, D]
    o_proj_out_batch: GPUArray | None = None      # [M, hidden]
    gate_up_out_batch: GPUArray | None = None     # [M, 2I]
    mlp_down_batch: GPUArray | None = None        # [M, hidden]
    cos_batch: GPUArray | None = None             # [M, D]
    sin_batch: GPUArray | None = None             # [M, D]
    logits_batch: GPUArray | None = None          # [M, vocab]
    q_flat_batch: GPUArray | None = None          # [M*Hq, D]
    k_flat_batch: GPUArray | None = None          # [M*Hkv, D]
    token_ids_batch_buf: GPUArray | None = None   # [M] int32
    start_position_batch_buf: GPUArray | None = None  # [1] int32
    # MoE (buffers.py:338-378): not on this path; the fields exist so attribute access does not fail
    moe_num_experts: int = 0
    moe_num_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    moe_router_logits: GPUArray | None = None
    moe_router_weights: GPUArray | None = None
    moe_expert_indices: GPUArray | None = None
    moe_expert_counts: GPUArray | None = None
    moe_expert_offsets: GPUArray | None = None
    moe_permute_indices: GPUArray | None = None
    moe_reverse_perm: GPUArray | None = None
    moe_row_expert_ids: GPUArray | None = None
    moe_gathered: GPUArray | None = None
    moe_gate_out: GPUArray | None = None
    moe_up_out: GPUArray | None = None
    moe_intermediate: GPUArray | None = None
    moe_expert_outputs: GPUArray | None = None
    moe_output: GPUArray | None = None

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
            M = max_batch_size
            b.hidden_batch, b.residual_batch, b.norm_out_batch = z(M, H), z(M, H), z(M, H)
            b.qkv_proj_out_batch = z(M, qd + 2 * kd)
            b.q_batch, b.k_batch, b.v_batch = z(M, Hq, D), z(M, Hkv, D), z(M, Hkv, D)
            b.q_t_batch, b.attn_out_batch, b.attn_out_t_batch = z(Hq, M, D), z(Hq, M, D), z(M, Hq, D)
            b.o_proj_out_batch, b.gate_up_out_batch, b.mlp_down_batch = z(M, H), z(M, 2 * I), z(M, H)
            b.cos_batch, b.sin_batch = z(M, D), z(M, D)
            b.token_ids_batch_buf = zeros((M,), "int32")
            b.start_position_batch_buf = zeros((1,), "int32")
            if vocab_size is not None:
                b.logits_batch = z(M, vocab_size)
            if use_qk_norm:
                b.q_flat_batch, b.k_flat_batch = z(M * Hq, D), z(M * Hkv, D)
        return b


@dataclass
class PrefillBuffers:
    """Named prefill activations with the reference's fields and shapes (buffers.py:476-621).  The engine's prefill
    keeps its own workspace; these serve strategy code that addresses buffers by name."""

    max_seq_len: int
    hidden: GPUArray           # [S, hidden]
    q: GPUArray                # [S, Hq, D]
    k: GPUArray                # [S, Hkv, D]
    v: GPUArray                # [S, Hkv, D]
    q_proj_out: GPUArray       # [S, Hq*D]
    k_proj_out: GPUArray       # [S, Hkv*D]
    v_proj_out: GPUArray       # [S, Hkv*D]
    o_proj_out: GPUArray       # [S, hidden]
    q_t: GPUArray              # [Hq, S, D]
    k_t: GPUArray              # [Hq, S, D]  (GQA-expanded in the reference)
    v_t: GPUArray              # [Hq, S, D]
    attn_out: GPUArray         # [Hq, S, D]
    attn_out_t: GPUArray       # [S, Hq, D]
    attn_out_2d: GPUArray      # [S, Hq*D]
    mlp_gate: GPUArray         # [S, I]
    mlp_up: GPUArray           # [S, I]
    mlp_down: GPUArray         # [S, hidden]
    cos: GPUArray              # [S, D]
    sin: GPUArray              # [S, D]
    residual: GPUArray         # [S, hidden]
    norm_out: GPUArray         # [S, hidden]
    q_2d: GPUArray | None = None   # [S*Hq, D]
    k_2d: GPUArray | None = None   # [S*Hkv, D]
    logits: GPUArray | None = None

    @classmethod
    def allocate(cls, config: "TransformerConfig", max_seq_len: int, dtype: str = "float16", use_qk_norm: bool = False) -> "PrefillBuffers":
        S, H, Hq, Hkv, D, I = max_seq_len, config.hidden_size, config.num_heads, config.num_kv_heads, config.head_dim, config.intermediate_size
        z = lambda *s: zeros(s, dtype)  # noqa: E731
        b = cls(max_seq_len=S, hidden=z(S, H), q=z(S, Hq, D), k=z(S, Hkv, D), v=z(S, Hkv, D), q_proj_out=z(S, Hq * D),
                k_proj_out=z(S, Hkv * D), v_proj_out=z(S, Hkv * D), o_proj_out=z(S, H), q_t=z(Hq, S, D), k_t=z(Hq, S, D),
                v_t=z(Hq, S, D), attn_out=z(Hq, S, D), attn_out_t=z(S, Hq, D), attn_out_2d=z(S, Hq * D), mlp_gate=z(S, I),
                mlp_up=z(S, I), mlp_down=z(S, H), cos=z(S, D), sin=z(S, D), residual=z(S, H), norm_out=z(S, H))
        if use_qk_norm:
            b.q_2d, b.k_2d = z(S * Hq, D), z(S * Hkv, D)
        return b
