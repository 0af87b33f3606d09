"""Layout shuffles and casts (reference: src/pygpukit/ops/tensor.py:20-552 -> native/ops/ops.cuh:345-436)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, float16, float32
from pygpukit_amd.ops._common import call, check_out, validate_same_dtype


def concat_axis0(a: GPUArray, b: GPUArray) -> GPUArray:
    if a.ndim != b.ndim or a.shape[1:] != b.shape[1:]:
        raise ValueError(f"concat_axis0: trailing dimensions differ, {a.shape} vs {b.shape}")
    validate_same_dtype(a, b, "concat_axis0")
    out = GPUArray((a.shape[0] + b.shape[0],) + a.shape[1:], a.dtype)
    call("pgk_memcpy_d2d", out._p, a._p, a.nbytes, None)
    call("pgk_memcpy_d2d", out.data_ptr() + a.nbytes, b._p, b.nbytes, None)
    return out


def repeat_interleave_axis1(input: GPUArray, repeats: int) -> GPUArray:
    if input.ndim != 3:
        raise ValueError(f"repeat_interleave_axis1 expects 3D input, got {input.ndim}D")
    d0, d1, d2 = input.shape
    out = GPUArray((d0, d1 * repeats, d2), input.dtype)
    call("pgk_repeat_interleave_axis1", input._p, out._p, d0, d1, d2, repeats, input.itemsize, None)
    return out


def transpose_3d_021(input: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """[d0,d1,d2] -> [d1,d0,d2]."""
    if input.ndim != 3:
        raise ValueError(f"transpose_3d_021 expects 3D input, got {input.ndim}D")
    d0, d1, d2 = input.shape
    o = check_out(out, (d1, d0, d2), input.dtype, "transpose_3d_021")
    call("pgk_transpose_3d_021", input._p, o._p, d0, d1, d2, input.itemsize, None)
    return o


def transpose_3d_012(input: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """[d0,d1,d2] -> [d0,d2,d1] (tensor.py:256-318): the last two axes swapped per slab."""
    if input.ndim != 3:
        raise ValueError(f"transpose_3d_012 expects 3D input, got {input.ndim}D")
    d0, d1, d2 = input.shape
    o = check_out(out, (d0, d2, d1), input.dtype, "transpose_3d_012")
    call("pgk_transpose_batched", input._p, o._p, d0, d1, d2, input.itemsize, None)
    return o


def transpose_4d_0132(input: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """[d0,d1,d2,d3] -> [d0,d1,d3,d2] (tensor.py:320-380): K^T per (batch, head)."""
    if input.ndim != 4:
        raise ValueError(f"transpose_4d_0132 expects 4D input, got {input.ndim}D")
    d0, d1, d2, d3 = input.shape
    o = check_out(out, (d0, d1, d3, d2), input.dtype, "transpose_4d_0132")
    call("pgk_transpose_batched", input._p, o._p, d0 * d1, d2, d3, input.itemsize, None)
    return o


def transpose_4d_0213(input: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    """[d0,d1,d2,d3] -> [d0,d2,d1,d3] (tensor.py:191-254): [batch, seq, heads, dim] <-> [batch, heads, seq, dim]."""
    if input.ndim != 4:
        raise ValueError(f"transpose_4d_0213 expects 4D input, got {input.ndim}D")
    d0, d1, d2, d3 = input.shape
    o = check_out(out, (d0, d2, d1, d3), input.dtype, "transpose_4d_0213")
    call("pgk_transpose_4d_0213", input._p, o._p, d0, d1, d2, d3, input.itemsize, None)
    return o


def reshape_copy(input: GPUArray, new_shape=None, *, out: GPUArray | None = None) -> GPUArray:
    """Copy into a new shape (tensor.py:395-477); `out` fixes the shape when given."""
    if out is None:
        if new_shape is None:
            raise ValueError("reshape_copy: new_shape or out is required")
        out = GPUArray(tuple(new_shape), input.dtype)
    if out.size != input.size or out.dtype != input.dtype:
        raise ValueError(f"reshape_copy: cannot reshape {input.shape}/{input.dtype} into {out.shape}/{out.dtype}")
    call("pgk_memcpy_d2d", out._p, input._p, input.nbytes, None)
    return out


def _cast(src: GPUArray, need, dst_dtype, name: str, out: GPUArray | None = None) -> GPUArray:
    if src.dtype != need:
        raise ValueError(f"{name}: input must be {need}, got {src.dtype}")
    o = check_out(out, src.shape, dst_dtype, name)
    call("pgk_cast", src._p, src.dtype.code, o._p, dst_dtype.code, src.size, None)
    return o


def cast_f32_to_bf16(src: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _cast(src, float32, bfloat16, "cast_f32_to_bf16", out)


def cast_f32_to_f16(src: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _cast(src, float32, float16, "cast_f32_to_f16", out)


def cast_bf16_to_f32(src: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _cast(src, bfloat16, float32, "cast_bf16_to_f32", out)


def cast_f16_to_f32(src: GPUArray, *, out: GPUArray | None = None) -> GPUArray:
    return _cast(src, float16, float32, "cast_f16_to_f32", out)
