"""Elementwise ops (reference: src/pygpukit/ops/elementwise.py:18-251 -> native/ops/ops.cuh:24-37,413-419)."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, check_out, validate_float, validate_same_dtype, validate_same_shape

_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3}


def _binary(a: GPUArray, b: GPUArray, name: str, out: GPUArray | None = None) -> GPUArray:
    validate_same_shape(a, b, name)
    validate_same_dtype(a, b, name)
    validate_float(a, name)
    c = check_out(out, a.shape, a.dtype, name)
    call("pgk_binary", a._p, b._p, c._p, a.size, _OPS[name], a.dtype.code, None)
    return c


def add(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _binary(a, b, "add", out)


def sub(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _binary(a, b, "sub", out)


def mul(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _binary(a, b, "mul", out)


def div(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _binary(a, b, "div", out)


def add_inplace(a: GPUArray, b: GPUArray) -> None:
    """a += b (elementwise.py:203-217)."""
    _binary(a, b, "add", a)


def mul_inplace(a: GPUArray, b: GPUArray) -> None:
    """a *= b (elementwise.py:220-234)."""
    _binary(a, b, "mul", a)


def copy_to(src: GPUArray, dst: GPUArray) -> None:
    """Device-to-device copy into a pre-allocated array (elementwise.py:237-251)."""
    if src.nbytes != dst.nbytes:
        raise ValueError(f"copy_to: size mismatch, {src.shape}/{src.dtype} -> {dst.shape}/{dst.dtype}")
    if src.dtype != dst.dtype:
        raise ValueError(f"copy_to: dtype mismatch {src.dtype} vs {dst.dtype}")
    call("pgk_memcpy_d2d", dst._p, src._p, src.nbytes, None)


def clamp(a: GPUArray, min_val: float, max_val: float) -> GPUArray:
    """Values limited to [min_val, max_val] (elementwise.py:254-276)."""
    validate_float(a, "clamp")
    o = GPUArray(a.shape, a.dtype)
    call("pgk_clamp", a._p, o._p, a.size, C.c_float(min_val), C.c_float(max_val), a.dtype.code, None)
    return o


def where(cond: GPUArray, a: GPUArray, b: GPUArray) -> GPUArray:
    """cond != 0 ? a : b; cond is uint8 / int8, one byte per element (elementwise.py:279-308)."""
    validate_same_shape(a, b, "where")
    validate_same_dtype(a, b, "where")
    validate_float(a, "where")
    if cond.dtype.itemsize != 1 or cond.size != a.size:
        raise ValueError(f"where: cond must be uint8/int8 with {a.size} elements, got {cond.dtype} x {cond.size}")
    o = GPUArray(a.shape, a.dtype)
    call("pgk_where", cond._p, a._p, b._p, o._p, a.size, a.dtype.code, None)
    return o
