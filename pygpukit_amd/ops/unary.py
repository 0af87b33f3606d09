"""Unary math (reference: src/pygpukit/ops/unary.py:16-260 -> native/ops/unary, ops.cuh:60-101).  fp32 math on
float32 / float16 / bfloat16 storage; one launch through pgk_activation's op codes."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, validate_float

_CODES = {"exp": 5, "log": 6, "relu": 7, "sin": 8, "cos": 9, "sqrt": 10, "rsqrt": 11, "abs": 12, "neg": 13}


def _unary(a: GPUArray, name: str) -> GPUArray:
    validate_float(a, name)
    o = GPUArray(a.shape, a.dtype)
    call("pgk_activation", a._p, o._p, a.size, _CODES[name], a.dtype.code, None)
    return o


def exp(a: GPUArray) -> GPUArray:
    return _unary(a, "exp")


def log(a: GPUArray) -> GPUArray:
    return _unary(a, "log")


def relu(a: GPUArray) -> GPUArray:
    return _unary(a, "relu")


def sin(a: GPUArray) -> GPUArray:
    return _unary(a, "sin")


def cos(a: GPUArray) -> GPUArray:
    return _unary(a, "cos")


def sqrt(a: GPUArray) -> GPUArray:
    return _unary(a, "sqrt")


def rsqrt(a: GPUArray) -> GPUArray:
    return _unary(a, "rsqrt")


def abs(a: GPUArray) -> GPUArray:  # noqa: A001 - the reference's name
    return _unary(a, "abs")


def neg(a: GPUArray) -> GPUArray:
    return _unary(a, "neg")


__all__ = ["exp", "log", "relu", "sin", "cos", "sqrt", "rsqrt", "abs", "neg"]
