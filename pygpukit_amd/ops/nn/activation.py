"""Activations (reference: src/pygpukit/ops/nn/activation.py:16-225 -> ops.cuh:136,176-191)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, check_out, validate_float

_ACT = {"silu": 0, "gelu": 1, "sigmoid": 2, "tanh": 3, "relu2": 4}


def _act(a: GPUArray, name: str, out: GPUArray | None) -> GPUArray:
    validate_float(a, name)
    o = check_out(out, a.shape, a.dtype, name)
    call("pgk_activation", a._p, o._p, a.size, _ACT[name], a.dtype.code, None)
    return o


def silu(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """x / (1 + exp(-x)); `out` may alias the input."""
    return _act(a, "silu", out)


def gelu(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """tanh GELU with the reference's constants 0.7978845608 / 0.044715."""
    return _act(a, "gelu", out)


def sigmoid(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _act(a, "sigmoid", out)


def tanh(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _act(a, "tanh", out)


def relu2(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _act(a, "relu2", out)
