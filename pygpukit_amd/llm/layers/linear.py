"""Linear layers (reference: src/pygpukit/llm/layers/linear.py:25-260).

LinearBF16: y = x W^T + b with W stored [out, in].  One kernel family reads W in that layout for
every batch size (weight-streaming GEMV for <= 8 rows, MFMA GEMM above), so no transposed copy of W
exists (the reference keeps W and W^T, linear.py:59-60) and `out=` works for every path, including
graph capture (the reference's GEMV path allocates and is skipped when out is given, linear.py:63-69).
LinearFP8: fp8-e4m3 weights [out, in] + bf16 128x128 block scales, dequantised in the kernels.
"""

from __future__ import annotations

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16
from pygpukit_amd.core.factory import from_numpy
from pygpukit_amd.ops.basic import bias_add_inplace, gemv_fp8_bf16_batched, matmul_nt, w8a16_gemm_nk


class LinearBF16:
    _use_gemv: bool = True  # kept for API compatibility; kernel selection is by row count

    def __init__(self, weight: GPUArray, bias: GPUArray | None = None):
        if weight.ndim != 2:
            raise ValueError(f"weight must be 2D, got {weight.ndim}D")
        self.weight = weight
        self.bias = bias
        self.out_features = weight.shape[0]
        self.in_features = weight.shape[1]

    def __call__(self, x: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
        if x.ndim != 2:
            raise ValueError(f"input must be 2D [batch, in_features], got {x.ndim}D")
        if x.shape[1] != self.in_features:
            raise ValueError(f"input features {x.shape[1]} != weight {self.in_features}")
        return matmul_nt(x, self.weight, self.bias, out=out)


Linear = LinearBF16


class LinearFP8:
    _use_gemv: bool = True
    _FP8_TABLE: np.ndarray | None = None

    @classmethod
    def _get_fp8_table(cls) -> np.ndarray:
        """E4M3 (OCP) -> fp32, 0x7F/0xFF = NaN (linear.py:123-144)."""
        if cls._FP8_TABLE is None:
            i = np.arange(256)
            sign = np.where(i & 0x80, -1.0, 1.0)
            exp, mant = (i >> 3) & 0xF, i & 0x7
            val = np.where(exp == 0, (mant / 8.0) * 2.0**-6, (1.0 + mant / 8.0) * np.exp2(exp.astype(np.float64) - 7))
            val = np.where((exp == 0xF) & (mant == 0x7), np.nan, sign * val)
            cls._FP8_TABLE = val.astype(np.float32)
        return cls._FP8_TABLE

    def __init__(self, weight_fp8: GPUArray, scale_inv: GPUArray, bias: GPUArray | None = None,
                 block_size: tuple[int, int] = (128, 128)):
        if weight_fp8.ndim != 2:
            raise ValueError(f"weight must be 2D, got {weight_fp8.ndim}D")
        if tuple(block_size) != (128, 128):
            raise ValueError("LinearFP8 supports 128x128 scale blocks only")
        self.weight_fp8 = weight_fp8
        self.scale_inv = scale_inv
        self.bias = bias
        self.block_size = block_size
        self.out_features = weight_fp8.shape[0]
        self.in_features = weight_fp8.shape[1]

    def _dequantize_cpu(self) -> np.ndarray:
        """fp32 dequantised weight on the host (linear.py:181-211); diagnostic helper, not on the forward path."""
        codes = self.weight_fp8.to_numpy().view(np.uint8)
        f32 = self._get_fp8_table()[codes]
        s = (self.scale_inv.to_numpy().astype(np.uint32) << 16).view(np.float32)
        H, W = f32.shape
        return (f32.reshape(H // 128, 128, W // 128, 128) * s[:, None, :, None]).reshape(H, W)

    def __call__(self, x: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
        if x.ndim != 2:
            raise ValueError(f"input must be 2D [batch, in_features], got {x.ndim}D")
        if x.shape[1] != self.in_features:
            raise ValueError(f"input features {x.shape[1]} != weight {self.in_features}")
        if x.dtype != bfloat16:
            raise ValueError("LinearFP8 requires bfloat16 activations")
        if x.shape[0] <= 8:
            y = gemv_fp8_bf16_batched(x, self.weight_fp8, self.scale_inv, out=out)
        else:
            y = w8a16_gemm_nk(x, self.weight_fp8, self.scale_inv, out=out)
        if self.bias is not None:
            bias_add_inplace(y, self.bias)
        return y


def quantize_linear_fp8(weight_f32: np.ndarray) -> tuple[GPUArray, GPUArray]:
    """Device arrays for quantize_fp8_host(weight): (codes uint8 [out,in], scale bf16 [out/128,in/128])."""
    codes, sbits = quantize_fp8_host(weight_f32)
    return from_numpy(codes), from_numpy(sbits)


def quantize_fp8_host(weight_f32: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Host helper for synthetic/benchmark weights: per-128x128-block absmax/448 scale (rounded to bf16),
    nearest-even E4M3 codes, never 0x7F/0xFF.  Returns (codes uint8 [out,in], scale bf16-bits [out/128,in/128])."""
    H, W = weight_f32.shape
    blocks = weight_f32.reshape(H // 128, 128, W // 128, 128).astype(np.float32)
    absmax = np.abs(blocks).max(axis=(1, 3))
    scale = np.where(absmax > 0, absmax / 448.0, 1.0).astype(np.float32)
    u = scale.view(np.uint32)
    sbits = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    scale = (sbits.astype(np.uint32) << 16).view(np.float32)
    x = blocks / scale[:, None, :, None]
    table = LinearFP8._get_fp8_table()[:0x7F]
    mag = np.minimum(np.abs(x), 448.0)
    hi = np.clip(np.searchsorted(table, mag.ravel(), side="left").reshape(mag.shape), 0, 0x7E)
    lo = np.clip(hi - 1, 0, 0x7E)
    dlo, dhi = np.abs(table[lo] - mag), np.abs(table[hi] - mag)
    pick_lo = np.where(dlo == dhi, (lo % 2) == 0, dlo < dhi)
    code = np.where(pick_lo, lo, hi).astype(np.uint8)
    code = np.where((x < 0) & (code != 0), code | 0x80, code).astype(np.uint8)
    return code.reshape(H, W), sbits


__all__ = ["LinearBF16", "LinearFP8", "Linear", "quantize_linear_fp8", "quantize_fp8_host"]
