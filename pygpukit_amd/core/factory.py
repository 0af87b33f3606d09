"""Array constructors (reference: src/pygpukit/core/factory.py:17-203: zeros / ones / empty /
from_numpy; a uint16 ndarray means bfloat16)."""

from __future__ import annotations

import ctypes as C

import numpy as np

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import DataType, as_dtype


def _shape(shape) -> tuple[int, ...]:
    return (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(d) for d in shape)


def empty(shape, dtype: "str | DataType" = "float32") -> GPUArray:
    return GPUArray(_shape(shape), as_dtype(dtype))


def zeros(shape, dtype: "str | DataType" = "float32") -> GPUArray:
    a = GPUArray(_shape(shape), as_dtype(dtype))
    a.fill_zeros()
    return a


def ones(shape, dtype: "str | DataType" = "float32") -> GPUArray:
    a = GPUArray(_shape(shape), as_dtype(dtype))
    if a.size:
        _hip.call("pgk_fill", a._p, 1.0, a.size, a.dtype.code, None)
    return a


def from_numpy(array: np.ndarray) -> GPUArray:
    array = np.ascontiguousarray(array)
    a = GPUArray(array.shape, DataType.from_numpy_dtype(array.dtype))
    if array.nbytes:
        _hip.call("pgk_memcpy_h2d", a._p, array.ctypes.data_as(C.c_void_p), array.nbytes, None)
    return a
