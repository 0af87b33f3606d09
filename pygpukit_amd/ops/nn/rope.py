"""Rotary position embedding, in place (reference: src/pygpukit/ops/nn/rope.py:16-133 -> ops.cuh:218-224).
q [S,Hq,D], k [S,Hk,D], cos/sin [S,D]; rotate-half, only table columns d < D/2 are read."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import float32
from pygpukit_amd.ops._common import call, validate_float


def _rope(q: GPUArray, k: GPUArray, cos: GPUArray, sin: GPUArray, f32_table: bool, name: str) -> None:
    validate_float(q, name)
    if q.ndim != 3 or k.ndim != 3:
        raise ValueError(f"{name} expects 3D q, k [seq_len, n_heads, head_dim]")
    if cos.ndim != 2 or sin.ndim != 2:
        raise ValueError(f"{name} expects 2D cos, sin [seq_len, head_dim]")
    S, Hq, D = q.shape
    if k.shape[0] != S or k.shape[2] != D or cos.shape != (S, D) or sin.shape != (S, D):
        raise ValueError(f"{name}: shape mismatch q{q.shape} k{k.shape} cos{cos.shape} sin{sin.shape}")
    if k.dtype != q.dtype:
        raise ValueError(f"{name}: q and k dtypes differ")
    want = float32 if f32_table else q.dtype
    if cos.dtype != want or sin.dtype != want:
        raise ValueError(f"{name}: cos/sin must be {want}")
    call("pgk_rope_inplace", q._p, k._p, cos._p, sin._p, S, Hq, k.shape[1], D, q.dtype.code, int(f32_table), None)


def rope_inplace(q: GPUArray, k: GPUArray, cos: GPUArray, sin: GPUArray) -> None:
    _rope(q, k, cos, sin, False, "rope_inplace")


def rope_inplace_f32table(q: GPUArray, k: GPUArray, cos: GPUArray, sin: GPUArray) -> None:
    """bf16/f16 q,k with fp32 tables (no table rounding)."""
    _rope(q, k, cos, sin, True, "rope_inplace_f32table")
