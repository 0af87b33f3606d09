"""Reductions on the hot path (reference: src/pygpukit/ops/reduction.py -> ops.cuh:104)."""

from __future__ import annotations

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import int32, int64
from pygpukit_amd.ops._common import call, validate_float


def argmax_rows(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """Index of the maximum of each row of a 1-D or 2-D array -> int32 [rows]; ties -> lowest index."""
    validate_float(a, "argmax")
    rows, n = (1, a.shape[0]) if a.ndim == 1 else (a.shape[0], a.size // a.shape[0])
    o = out if out is not None else GPUArray((rows,), int32)
    call("pgk_argmax", a._p, rows, n, a.dtype.code, o._p, None)
    return o


def argmax(a: GPUArray) -> GPUArray:
    """Flat index of the maximum -> int64 [1] on the device (reduction.py:249-268; np.argmax semantics, ties -> lowest)."""
    validate_float(a, "argmax")
    o = GPUArray((1,), int32)
    call("pgk_argmax", a._p, 1, a.size, a.dtype.code, o._p, None)
    o64 = GPUArray((1,), int64)
    call("pgk_widen_i32_i64", o._p, o64._p, 1, None)
    return o64


def argmax_int(a: GPUArray) -> int:
    """argmax as a Python int (one 8-byte D2H copy)."""
    return int(np.asarray(argmax(a).to_numpy())[0])


_RED = {"sum": 0, "mean": 1, "max": 2, "min": 3}


def _reduce(a: GPUArray, name: str) -> GPUArray:
    validate_float(a, name)
    if a.size == 0:
        if name == "sum":               # np.sum of nothing is 0 (the reference's CPU branch); max / min / mean have no value
            from pygpukit_amd.core.factory import zeros
            return zeros((1,), a.dtype)
        raise ValueError(f"{name} of an empty array")
    o = GPUArray((1,), a.dtype)
    call("pgk_reduce", a._p, o._p, a.size, _RED[name], a.dtype.code, None)
    return o


def sum(a: GPUArray) -> GPUArray:  # noqa: A001 - the reference's names
    """Sum of all elements -> shape [1] in the input dtype (reduction.py:16-52); fp32 accumulation, fixed tree."""
    return _reduce(a, "sum")


def mean(a: GPUArray) -> GPUArray:
    return _reduce(a, "mean")


def max(a: GPUArray) -> GPUArray:  # noqa: A001
    return _reduce(a, "max")


def min(a: GPUArray) -> GPUArray:  # noqa: A001
    return _reduce(a, "min")


def softmax(input: GPUArray, axis: int = -1) -> GPUArray:  # noqa: A002
    """Softmax over the last axis of a 2-D..4-D array (reduction.py:133-224): leading axes are flattened to rows."""
    validate_float(input, "softmax")
    if input.ndim < 2:
        raise ValueError(f"softmax expects at least 2D input, got {input.ndim}D")
    if input.ndim > 4:
        raise ValueError(f"softmax supports up to 4D input, got {input.ndim}D")
    if axis < 0:
        axis = input.ndim + axis
    if axis != input.ndim - 1:
        raise ValueError(f"softmax currently only supports axis=-1 (last axis), got axis={axis}")
    n = input.shape[-1]
    o = GPUArray(input.shape, input.dtype)
    call("pgk_softmax_rows", input._p, o._p, input.size // n if n else 0, n, input.dtype.code, None)
    return o


def sum_axis(a: GPUArray, axis: int) -> GPUArray:
    """2-D [M, N]: axis 0 -> [N], axis 1 -> [M] (reduction.py:271-300)."""
    validate_float(a, "sum_axis")
    if a.ndim != 2:
        raise ValueError(f"sum_axis requires 2D input, got {a.ndim}D")
    if axis not in (0, 1):
        raise ValueError(f"sum_axis: axis must be 0 or 1, got {axis}")
    m, n = a.shape
    o = GPUArray((n if axis == 0 else m,), a.dtype)
    call("pgk_sum_axis", a._p, o._p, m, n, axis, a.dtype.code, None)
    return o
