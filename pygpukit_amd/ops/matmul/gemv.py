"""GEMV family (reference: src/pygpukit/ops/matmul/gemv.py:15-205 -> gemv_bf16_bf16_sm120 /
gemv_fp8_bf16_sm120 / gemv_fp8_bf16_batched_sm120)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, uint8
from pygpukit_amd.ops._common import call, check_out


def gemv_bf16(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """C[N] = A[K] @ B[N,K]^T, bf16 in/out, fp32 accumulate."""
    if a.ndim != 1:
        raise ValueError(f"gemv_bf16 requires 1D input vector, got {a.ndim}D")
    if b.ndim != 2:
        raise ValueError(f"gemv_bf16 requires 2D weight matrix, got {b.ndim}D")
    if a.dtype != bfloat16 or b.dtype != bfloat16:
        raise ValueError("gemv_bf16 requires bfloat16 inputs")
    K, N = a.shape[0], b.shape[0]
    if b.shape[1] != K:
        raise ValueError(f"gemv_bf16 dimension mismatch: A[{K}] vs B[{N}, {b.shape[1]}]")
    c = check_out(out, (N,), bfloat16, "gemv_bf16")
    call("pgk_gemv", a._p, b._p, c._p, K, N, bfloat16.code, None)
    return c


def gemv_bf16_opt_available() -> bool:
    """The reference gates its optimised kernel on SM >= 80; the wave64 kernel is always on."""
    return True


def _check_fp8(a: GPUArray, b_nk: GPUArray, b_scale: GPUArray, K: int, name: str) -> int:
    if b_nk.ndim != 2 or b_nk.dtype != uint8:
        raise ValueError(f"{name} requires uint8 weight [N, K]")
    if a.dtype != bfloat16 or b_scale.dtype != bfloat16:
        raise ValueError(f"{name} requires bfloat16 activations and scales")
    N = b_nk.shape[0]
    if b_nk.shape[1] != K:
        raise ValueError(f"{name} dimension mismatch: K={K} vs B[{N}, {b_nk.shape[1]}]")
    if K % 128 or N % 128 or b_scale.shape != (N // 128, K // 128):
        raise ValueError(f"{name}: scale must be [N/128, K/128] = [{N // 128}, {K // 128}], got {b_scale.shape}")
    return N


def gemv_fp8_bf16(a: GPUArray, b_nk: GPUArray, b_scale: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """C[N] = A[K] @ dequant(B_fp8[N,K], scale[N/128,K/128])^T."""
    if a.ndim != 1:
        raise ValueError(f"gemv_fp8_bf16 requires 1D input vector, got {a.ndim}D")
    K = a.shape[0]
    N = _check_fp8(a, b_nk, b_scale, K, "gemv_fp8_bf16")
    c = check_out(out, (N,), bfloat16, "gemv_fp8_bf16")
    call("pgk_gemv_fp8_bf16", a._p, b_nk._p, b_scale._p, c._p, 1, K, N, None)
    return c


def gemv_fp8_bf16_batched(a: GPUArray, b_nk: GPUArray, b_scale: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """C[M,N] = A[M,K] @ dequant(B_fp8[N,K])^T."""
    if a.ndim != 2:
        raise ValueError(f"gemv_fp8_bf16_batched requires 2D input [M, K], got {a.ndim}D")
    M, K = a.shape
    N = _check_fp8(a, b_nk, b_scale, K, "gemv_fp8_bf16_batched")
    c = check_out(out, (M, N), bfloat16, "gemv_fp8_bf16_batched")
    call("pgk_gemv_fp8_bf16", a._p, b_nk._p, b_scale._p, c._p, M, K, N, None)
    return c
