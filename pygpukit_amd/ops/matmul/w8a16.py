"""W8A16 GEMM: bf16 activations x fp8-e4m3 weights with 128x128 block scales
(reference: src/pygpukit/ops/matmul/w8a16.py:38-117 -> pygpukit_w8a16_gemm_sm120).  The reference's
name is kept so LinearFP8 callers run unchanged; there is no SM120 here."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, uint8
from pygpukit_amd.ops._common import call, check_out


def w8a16_gemm_sm120(a: GPUArray, b_fp8: GPUArray, b_scale: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """C[M,N] = A[M,K] @ dequant(B_fp8[K,N], scale[K/128,N/128])  - note the [K,N] weight layout."""
    if a.ndim != 2 or b_fp8.ndim != 2:
        raise ValueError("w8a16_gemm requires 2D inputs")
    if a.dtype != bfloat16 or b_scale.dtype != bfloat16 or b_fp8.dtype != uint8:
        raise ValueError("w8a16_gemm requires bf16 activations/scales and uint8 weights")
    M, K = a.shape
    if b_fp8.shape[0] != K:
        raise ValueError(f"w8a16_gemm dimension mismatch: A[{M},{K}] vs B[{b_fp8.shape[0]},{b_fp8.shape[1]}]")
    N = b_fp8.shape[1]
    if K % 128 or N % 128 or b_scale.shape != (K // 128, N // 128):
        raise ValueError(f"w8a16_gemm: scale must be [K/128, N/128], got {b_scale.shape}")
    c = check_out(out, (M, N), bfloat16, "w8a16_gemm")
    call("pgk_w8a16_gemm_kn", a._p, b_fp8._p, b_scale._p, c._p, M, N, K, None)
    return c


w8a16_gemm = w8a16_gemm_sm120


def w8a16_gemm_nk(a: GPUArray, w_fp8: GPUArray, w_scale: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """C[M,N] = A[M,K] @ dequant(W_fp8[N,K], scale[N/128,K/128])^T on the stored layout (no transposed copies)."""
    if a.ndim != 2 or w_fp8.ndim != 2 or a.shape[1] != w_fp8.shape[1]:
        raise ValueError(f"w8a16_gemm_nk dimension mismatch: {a.shape} @ {w_fp8.shape}^T")
    if a.dtype != bfloat16 or w_scale.dtype != bfloat16 or w_fp8.dtype != uint8:
        raise ValueError("w8a16_gemm_nk requires bf16 activations/scales and uint8 weights")
    M, K = a.shape
    N = w_fp8.shape[0]
    if K % 128 or N % 128 or w_scale.shape != (N // 128, K // 128):
        raise ValueError(f"w8a16_gemm_nk: scale must be [N/128, K/128], got {w_scale.shape}")
    c = check_out(out, (M, N), bfloat16, "w8a16_gemm_nk")
    call("pgk_w8a16_gemm_nk", a._p, w_fp8._p, w_scale._p, c._p, M, N, K, None)
    return c


def gemm_w8a16_init_lut() -> None:
    """The reference uploads a 256-entry LUT to constant memory; gfx950 converts e4m3 in hardware."""
