"""Reductions on the hot path (reference: src/pygpukit/ops/reduction.py -> ops.cuh:104)."""

from __future__ import annotations

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import int32
from pygpukit_amd.ops._common import call, validate_float


def argmax_rows(a: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """Index of the maximum of each row of a 1-D or 2-D array -> int32 [rows]; ties -> lowest index."""
    validate_float(a, "argmax")
    rows, n = (1, a.shape[0]) if a.ndim == 1 else (a.shape[0], a.size // a.shape[0])
    o = out if out is not None else GPUArray((rows,), int32)
    call("pgk_argmax", a._p, rows, n, a.dtype.code, o._p, None)
    return o


def argmax(a: GPUArray) -> int:
    """Flat argmax as a Python int (np.argmax semantics)."""
    validate_float(a, "argmax")
    o = GPUArray((1,), int32)
    call("pgk_argmax", a._p, 1, a.size, a.dtype.code, o._p, None)
    return int(np.asarray(o.to_numpy())[0])
