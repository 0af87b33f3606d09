"""MLP layer (reference: src/pygpukit/llm/layers/mlp.py:25-98): GELU (fc1 -> gelu -> fc2) or SwiGLU
(silu(gate) * up -> down).  With raw bf16 weights gate and up are fused into ONE [2I, H] weight
(mlp.py:84-86); here gate_proj / up_proj are zero-copy row views of that fused weight, so the bytes exist
once, and the forward is one projection + one packed GLU kernel + the down projection."""

from __future__ import annotations

from typing import TYPE_CHECKING

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops.basic import concat_axis0, gelu, glu_packed, swiglu

from .linear import LinearBF16, LinearFP8

if TYPE_CHECKING:
    from pygpukit_amd.llm.config import TransformerConfig


def _wrap(proj, bias=None):
    if proj is None or isinstance(proj, (LinearBF16, LinearFP8)):
        return proj
    return LinearBF16(proj, bias)


class MLP:
    def __init__(self, config: "TransformerConfig", fc1_weight=None, fc1_bias: GPUArray | None = None, fc2_weight=None,
                 fc2_bias: GPUArray | None = None, gate_proj=None, up_proj=None, down_proj=None):
        self.config = config
        self.activation = config.activation
        if config.activation == "gelu":
            if fc1_weight is None or fc2_weight is None:
                raise ValueError("GELU MLP requires fc1_weight and fc2_weight")
            self.fc1, self.fc2 = _wrap(fc1_weight, fc1_bias), _wrap(fc2_weight, fc2_bias)
            return
        if gate_proj is None or up_proj is None or down_proj is None:
            raise ValueError("SwiGLU MLP requires gate_proj, up_proj, down_proj")
        self.down_proj = _wrap(down_proj)
        if isinstance(gate_proj, GPUArray) and isinstance(up_proj, GPUArray):
            inter, hidden = gate_proj.shape
            fused = concat_axis0(gate_proj, up_proj)
            self.gate_up_proj: LinearBF16 | None = LinearBF16(fused, None)
            self.gate_proj = LinearBF16(fused._view(0, (inter, hidden)))
            self.up_proj = LinearBF16(fused._view(inter * hidden, (inter, hidden)))
            self.intermediate_size = inter
        else:
            self.gate_up_proj = None
            self.gate_proj, self.up_proj = _wrap(gate_proj), _wrap(up_proj)
            self.intermediate_size = self.gate_proj.out_features

    def __call__(self, x: GPUArray) -> GPUArray:
        if self.activation == "gelu":
            h = self.fc1(x)
            return self.fc2(gelu(h, out=h))
        if self.gate_up_proj is not None:
            act = glu_packed(self.gate_up_proj(x), self.intermediate_size)
        else:
            act = swiglu(self.gate_proj(x), self.up_proj(x))
        return self.down_proj(act)


__all__ = ["MLP"]
