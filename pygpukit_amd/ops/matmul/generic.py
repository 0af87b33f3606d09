"""Dense matmul (reference: src/pygpukit/ops/matmul/generic.py:18-162 -> native/ops/matmul/matmul.cu:43-354).
MFMA kernels for bf16/f16, LDS-tiled FMA for fp32; fp32 accumulation everywhere."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops._common import call, check_out, validate_float, validate_same_dtype


def matmul(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None, use_tf32: bool | None = None) -> GPUArray:
    """C[M,N] = A[M,K] @ B[K,N].  `use_tf32` is accepted for signature compatibility; gfx950 has no
    TF32/xf32 path and fp32 inputs use exact fp32 FMAs."""
    if a.ndim != 2:
        raise ValueError(f"matmul requires 2D arrays, got {a.ndim}D for first argument")
    if b.ndim != 2:
        raise ValueError(f"matmul requires 2D arrays, got {b.ndim}D for second argument")
    if a.shape[1] != b.shape[0]:
        raise ValueError(f"matmul dimension mismatch: {a.shape} @ {b.shape} (inner dimensions {a.shape[1]} and {b.shape[0]} must match)")
    validate_same_dtype(a, b, "matmul")
    validate_float(a, "matmul")
    M, K = a.shape
    N = b.shape[1]
    c = check_out(out, (M, N), a.dtype, "matmul")
    call("pgk_gemm_nn", a._p, b._p, c._p, M, N, K, a.dtype.code, None)
    return c


def matmul_nt(a: GPUArray, w: GPUArray, bias: GPUArray | None = None, *, out: GPUArray | None = None) -> GPUArray:
    """C[M,N] = A[M,K] @ W[N,K]^T (+ bias[N]): the Linear layer on the PyTorch-layout weight, no transposed copy."""
    if a.ndim != 2 or w.ndim != 2 or a.shape[1] != w.shape[1]:
        raise ValueError(f"matmul_nt dimension mismatch: {a.shape} @ {w.shape}^T")
    validate_same_dtype(a, w, "matmul_nt")
    validate_float(a, "matmul_nt")
    if bias is not None and (bias.shape != (w.shape[0],) or bias.dtype != a.dtype):
        raise ValueError("matmul_nt: bias must be [N] of the input dtype")
    M, K = a.shape
    N = w.shape[0]
    c = check_out(out, (M, N), a.dtype, "matmul_nt")
    call("pgk_gemm_nt", a._p, w._p, bias._p if bias is not None else None, c._p, M, N, K, a.dtype.code, None)
    return c


def transpose(a: GPUArray) -> GPUArray:
    """2-D transpose (generic.py:122-162); any element size (also used on uint8 fp8 weights)."""
    if a.ndim != 2:
        raise ValueError(f"transpose requires 2D array, got {a.ndim}D")
    out = GPUArray((a.shape[1], a.shape[0]), a.dtype)
    call("pgk_transpose_2d", a._p, out._p, a.shape[0], a.shape[1], a.itemsize, None)
    return out


def batched_matmul(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """[B,M,K] @ [B,K,N]: one GEMM launch per batch entry."""
    if a.ndim != 3 or b.ndim != 3 or a.shape[0] != b.shape[0] or a.shape[2] != b.shape[1]:
        raise ValueError(f"batched_matmul dimension mismatch: {a.shape} @ {b.shape}")
    validate_same_dtype(a, b, "batched_matmul")
    validate_float(a, "batched_matmul")
    B, M, K = a.shape
    N = b.shape[2]
    c = check_out(out, (B, M, N), a.dtype, "batched_matmul")
    isz = a.itemsize
    for i in range(B):
        call("pgk_gemm_nn", a.data_ptr() + i * M * K * isz, b.data_ptr() + i * K * N * isz, c.data_ptr() + i * M * N * isz,
             M, N, K, a.dtype.code, None)
    return c


def linear_bias_gelu(input: GPUArray, weight: GPUArray, bias: GPUArray) -> GPUArray:
    """gelu(input @ weight^T + bias), weight [out,in] (generic.py:165-230)."""
    from pygpukit_amd.ops.nn.activation import gelu

    y = matmul_nt(input, weight, bias)
    return gelu(y, out=y)
