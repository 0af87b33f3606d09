"""Transformer block (reference: src/pygpukit/llm/layers/block.py:18-57): pre-norm attention and MLP with
residual adds."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops.basic import add

from .attention import Attention
from .mlp import MLP
from .norm import Norm


class TransformerBlock:
    def __init__(self, attn_norm: Norm, attn: Attention, mlp_norm: Norm, mlp: MLP):
        self.attn_norm, self.attn, self.mlp_norm, self.mlp = attn_norm, attn, mlp_norm, mlp

    def __call__(self, x: GPUArray, position_ids: list[int] | None = None, past_kv: tuple | None = None,
                 use_cache: bool = False) -> tuple[GPUArray, tuple | None]:
        attn_out, present_kv = self.attn(self.attn_norm(x), position_ids, past_kv, use_cache)
        x = add(x, attn_out)
        x = add(x, self.mlp(self.mlp_norm(x)))
        return x, present_kv


__all__ = ["TransformerBlock"]
