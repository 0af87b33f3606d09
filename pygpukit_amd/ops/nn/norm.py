"""Row normalisations (reference: src/pygpukit/ops/nn/norm.py:16-218 -> ops.cuh:143-155)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, check_out, validate_float, validate_same_dtype


def _check(input: GPUArray, gamma: GPUArray, name: str) -> None:
    validate_float(input, name)
    if input.ndim != 2:
        raise ValueError(f"{name} expects 2D input [batch, features], got {input.ndim}D")
    if gamma.ndim != 1:
        raise ValueError(f"{name} expects 1D gamma")
    validate_same_dtype(input, gamma, name)
    if input.shape[1] != gamma.shape[0]:
        raise ValueError(f"{name}: input features ({input.shape[1]}) must match gamma size ({gamma.shape[0]})")


def rmsnorm(input: GPUArray, gamma: GPUArray, eps: float = 1e-5, *, out: GPUArray | None = None) -> GPUArray:
    _check(input, gamma, "rmsnorm")
    o = check_out(out, input.shape, input.dtype, "rmsnorm")
    call("pgk_rmsnorm", input._p, gamma._p, o._p, input.shape[0], input.shape[1], eps, input.dtype.code, None)
    return o


def layernorm(input: GPUArray, gamma: GPUArray, beta: GPUArray, eps: float = 1e-5, *, out: GPUArray | None = None) -> GPUArray:
    _check(input, gamma, "layernorm")
    if beta.shape != gamma.shape or beta.dtype != gamma.dtype:
        raise ValueError("layernorm: beta must match gamma")
    o = check_out(out, input.shape, input.dtype, "layernorm")
    call("pgk_layernorm", input._p, gamma._p, beta._p, o._p, input.shape[0], input.shape[1], eps, input.dtype.code, None)
    return o
