"""Fused ops (reference: src/pygpukit/ops/nn/fused.py:16-179 -> ops.cuh:199-212)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, check_out, validate_float, validate_same_dtype, validate_same_shape
from pygpukit_amd.ops.nn.norm import _check


def rmsnorm_residual(input: GPUArray, residual: GPUArray, gamma: GPUArray, eps: float = 1e-5, *,
                     out: GPUArray | None = None) -> GPUArray:
    """out = rmsnorm(input + residual) * gamma."""
    _check(input, gamma, "rmsnorm_residual")
    validate_same_shape(input, residual, "rmsnorm_residual")
    validate_same_dtype(input, residual, "rmsnorm_residual")
    o = check_out(out, input.shape, input.dtype, "rmsnorm_residual")
    call("pgk_rmsnorm_residual", input._p, residual._p, gamma._p, o._p, input.shape[0], input.shape[1], eps, input.dtype.code, None)
    return o


def _glu(gate: GPUArray, up: GPUArray, act: int, name: str, out: GPUArray | None) -> GPUArray:
    validate_float(gate, name)
    validate_same_shape(gate, up, name)
    validate_same_dtype(gate, up, name)
    o = check_out(out, gate.shape, gate.dtype, name)
    call("pgk_glu", gate._p, up._p, o._p, gate.size, act, gate.dtype.code, None)
    return o


def swiglu(gate_proj: GPUArray, up_proj: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """silu(gate) * up."""
    return _glu(gate_proj, up_proj, 0, "swiglu", out)


def geglu(gate_proj: GPUArray, up_proj: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """gelu(gate) * up."""
    return _glu(gate_proj, up_proj, 1, "geglu", out)


def glu_packed(gate_up: GPUArray, inter: int, *, activation: str = "silu", out: GPUArray | None = None) -> GPUArray:
    """gate_up [rows, 2*inter] (fused gate_up projection) -> act(gate) * up as [rows, inter]."""
    validate_float(gate_up, "glu_packed")
    if gate_up.ndim != 2 or gate_up.shape[1] != 2 * inter:
        raise ValueError(f"glu_packed: expected [rows, {2 * inter}], got {gate_up.shape}")
    o = check_out(out, (gate_up.shape[0], inter), gate_up.dtype, "glu_packed")
    call("pgk_glu_packed", gate_up._p, o._p, gate_up.shape[0], inter, 0 if activation == "silu" else 1, gate_up.dtype.code, None)
    return o
