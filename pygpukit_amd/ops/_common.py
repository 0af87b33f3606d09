"""Validation helpers shared by the op wrappers (reference: src/pygpukit/ops/_common.py).
Python-side checks raise ValueError before anything is launched, like the reference's wrappers;
failures inside the native library raise RuntimeError (pygpukit_amd._hip.PgkError)."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import FLOAT_DTYPES, DataType

call = _hip.call
vp = C.c_void_p


def validate_float(a: GPUArray, name: str) -> None:
    if a.dtype not in FLOAT_DTYPES:
        raise ValueError(f"{name} requires float32/float16/bfloat16, got {a.dtype}")


def validate_same_shape(a: GPUArray, b: GPUArray, name: str) -> None:
    if a.shape != b.shape:
        raise ValueError(f"{name} requires arrays of same shape, got {a.shape} and {b.shape}")


def validate_same_dtype(a: GPUArray, b: GPUArray, name: str) -> None:
    if a.dtype != b.dtype:
        raise ValueError(f"{name} requires arrays of same dtype, got {a.dtype} and {b.dtype}")


def check_out(out: GPUArray | None, shape, dtype: DataType, name: str) -> GPUArray:
    """Return `out` after validating it, or a fresh array from the pool."""
    shape = tuple(shape)
    if out is None:
        return GPUArray(shape, dtype)
    if out.shape != shape:
        raise ValueError(f"{name}: out shape {out.shape} does not match expected {shape}")
    if out.dtype != dtype:
        raise ValueError(f"{name}: out dtype {out.dtype} does not match {dtype}")
    return out
