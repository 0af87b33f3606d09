"""Norm layer (reference: src/pygpukit/llm/layers/norm.py:18-39)."""

from __future__ import annotations

from typing import Literal

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops.basic import layernorm, rmsnorm


class Norm:
    def __init__(self, weight: GPUArray, bias: GPUArray | None = None,
                 norm_type: Literal["rmsnorm", "layernorm"] = "rmsnorm", eps: float = 1e-5):
        self.weight, self.bias, self.norm_type, self.eps = weight, bias, norm_type, eps

    def __call__(self, x: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
        if self.norm_type == "rmsnorm":
            return rmsnorm(x, self.weight, self.eps, out=out)
        if self.bias is None:
            raise ValueError("LayerNorm requires bias")
        return layernorm(x, self.weight, self.bias, self.eps, out=out)


__all__ = ["Norm"]
