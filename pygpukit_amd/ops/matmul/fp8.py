"""fp8 (OCP e4m3) x fp8 GEMM with 128-wide block scales, and the quantisers that feed it
(reference: src/pygpukit/ops/matmul/fp8.py:20-363 - matmul_fp8 / matmul_fp8_fp8_blockwise_sm120; its native
side is CUTLASS and absent from the checkout, so the numerics are the formula in include/pgk_hip.h and
oracle.cpu_ref.gemm_fp8_blockwise).  The reference's per-architecture names are kept as aliases."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, float16, float32, uint8
from pygpukit_amd.ops._common import call, check_out


def quantize_fp8_rows(x: GPUArray) -> tuple[GPUArray, GPUArray]:
    """Per (row, 128-k block) e4m3 quantisation of a 2-D float array: returns (codes uint8 [M,K],
    scale float32 [M, K/128]) with scale = absmax/448 (1 for an all-zero block)."""
    if x.ndim != 2 or x.dtype not in (bfloat16, float16, float32):
        raise ValueError(f"quantize_fp8_rows requires a 2D float array, got {x.ndim}D {x.dtype}")
    M, K = x.shape
    if K % 128:
        raise ValueError(f"quantize_fp8_rows: K={K} must be a multiple of 128")
    codes, scale = GPUArray((M, K), uint8), GPUArray((M, K // 128), float32)
    call("pgk_quantize_fp8_rows", x._p, codes._p, scale._p, M, K, x.dtype.code, None)
    return codes, scale


def quantize_fp8_blocks(w: GPUArray) -> tuple[GPUArray, GPUArray]:
    """Per 128x128-block e4m3 quantisation of a bf16 weight [N,K] into the LinearFP8 storage format:
    returns (codes uint8 [N,K], scale_inv bfloat16 [ceil(N/128), K/128])."""
    if w.ndim != 2 or w.dtype != bfloat16:
        raise ValueError(f"quantize_fp8_blocks requires a 2D bfloat16 array, got {w.ndim}D {w.dtype}")
    N, K = w.shape
    if K % 128:
        raise ValueError(f"quantize_fp8_blocks: K={K} must be a multiple of 128")
    codes, scale = GPUArray((N, K), uint8), GPUArray(((N + 127) // 128, K // 128), bfloat16)
    call("pgk_quantize_fp8_blocks", w._p, codes._p, scale._p, N, K, None)
    return codes, scale


def gemm_fp8_fp8_blockwise_nt(a_fp8: GPUArray, w_fp8: GPUArray, scale_a: GPUArray, scale_w: GPUArray, *,
                              out: GPUArray | None = None) -> GPUArray:
    """C[M,N] (bf16) = blockwise-scaled A_fp8[M,K] @ W_fp8[N,K]^T; scale_a float32 [M,K/128], scale_w bf16 [N/128,K/128]."""
    if a_fp8.ndim != 2 or w_fp8.ndim != 2:
        raise ValueError("gemm_fp8_fp8_blockwise_nt requires 2D arrays")
    if a_fp8.dtype != uint8 or w_fp8.dtype != uint8:
        raise ValueError("gemm_fp8_fp8_blockwise_nt requires uint8 inputs (FP8)")
    M, K = a_fp8.shape
    N = w_fp8.shape[0]
    if w_fp8.shape[1] != K:
        raise ValueError(f"gemm_fp8_fp8_blockwise_nt dimension mismatch: {a_fp8.shape} @ {w_fp8.shape}^T")
    if K % 128:
        raise ValueError(f"gemm_fp8_fp8_blockwise_nt: K={K} must be a multiple of 128")
    if scale_a.dtype != float32 or scale_a.shape != (M, K // 128):
        raise ValueError(f"gemm_fp8_fp8_blockwise_nt: scale_a must be float32 [M, K/128], got {scale_a.dtype} {scale_a.shape}")
    if scale_w.dtype != bfloat16 or scale_w.shape != ((N + 127) // 128, K // 128):
        raise ValueError(f"gemm_fp8_fp8_blockwise_nt: scale_w must be bfloat16 [N/128, K/128], got {scale_w.dtype} {scale_w.shape}")
    c = check_out(out, (M, N), bfloat16, "gemm_fp8_fp8_blockwise_nt")
    call("pgk_gemm_fp8_nt", a_fp8._p, scale_a._p, w_fp8._p, scale_w._p, c._p, M, N, K, None)
    return c


def matmul_fp8(a: GPUArray, b: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """float32 A[M,K] @ B[K,N] through fp8: both operands are quantised on the device (A per row-block, B per
    128x128 block of its transpose), multiplied on the fp8 MFMA path and returned as float32 (fp8.py:20-70)."""
    from pygpukit_amd.ops.matmul.generic import transpose

    if a.ndim != 2:
        raise ValueError(f"matmul_fp8 requires 2D arrays, got {a.ndim}D for first argument")
    if b.ndim != 2:
        raise ValueError(f"matmul_fp8 requires 2D arrays, got {b.ndim}D for second argument")
    if a.shape[1] != b.shape[0]:
        raise ValueError(f"matmul_fp8 dimension mismatch: {a.shape} @ {b.shape} "
                         f"(inner dimensions {a.shape[1]} and {b.shape[0]} must match)")
    if a.dtype != float32 or b.dtype != float32:
        raise ValueError("matmul_fp8 requires float32 inputs")
    M, K = a.shape
    N = b.shape[1]
    if K % 128:
        raise ValueError(f"matmul_fp8: K={K} must be a multiple of 128 on this backend")
    a8, sa = quantize_fp8_rows(a)
    w8, sw = quantize_fp8_blocks(transpose(b).astype(bfloat16))
    c = gemm_fp8_fp8_blockwise_nt(a8, w8, sa, sw)
    c32 = c.astype(float32)
    if out is None:
        return c32
    check_out(out, (M, N), float32, "matmul_fp8")
    from pygpukit_amd.ops.elementwise import copy_to

    copy_to(c32, out)
    return out


# the reference's per-architecture entry points all land on the one gfx950 kernel
matmul_fp8_sm90 = matmul_fp8_sm100 = matmul_fp8_sm120 = matmul_fp8
gemm_fp8_f32_sm90 = gemm_fp8_f32_sm100 = gemm_fp8_f32_sm120 = matmul_fp8


def fp8_available() -> bool:
    return True


fp8_sm90_available = fp8_sm100_available = fp8_sm120_available = fp8_fp8_sm120_available = fp8_available
gemm_fp8_available = gemm_fp8_f32_sm90_available = gemm_fp8_f32_sm100_available = gemm_fp8_f32_sm120_available = fp8_available


def fp8_init_lut() -> None:
    """The reference uploads an e4m3 LUT; gfx950 converts in hardware."""


__all__ = ["matmul_fp8", "matmul_fp8_sm90", "matmul_fp8_sm100", "matmul_fp8_sm120", "gemm_fp8_f32_sm90", "gemm_fp8_f32_sm100",
           "gemm_fp8_f32_sm120", "gemm_fp8_fp8_blockwise_nt", "quantize_fp8_rows", "quantize_fp8_blocks", "fp8_available",
           "fp8_sm90_available", "fp8_sm100_available", "fp8_sm120_available", "fp8_fp8_sm120_available", "fp8_init_lut",
           "gemm_fp8_available", "gemm_fp8_f32_sm90_available", "gemm_fp8_f32_sm100_available", "gemm_fp8_f32_sm120_available"]

