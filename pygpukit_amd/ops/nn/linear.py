"""Linear-layer helpers (reference: src/pygpukit/ops/nn/linear.py:14-159 -> ops.cuh:139,271,410)."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import int32
from pygpukit_amd.ops._common import call, validate_float, validate_same_dtype


def bias_add_inplace(output: GPUArray, bias: GPUArray) -> None:
    """output[batch, features] += bias[features]."""
    validate_float(output, "bias_add_inplace")
    if output.ndim != 2 or bias.ndim != 1 or bias.shape[0] != output.shape[1]:
        raise ValueError(f"bias_add_inplace: output {output.shape} / bias {bias.shape} mismatch")
    validate_same_dtype(output, bias, "bias_add_inplace")
    call("pgk_bias_add_inplace", output._p, bias._p, output.shape[0], output.shape[1], output.dtype.code, None)


def split_qkv_batch(qkv: GPUArray, q_out: GPUArray, k_out: GPUArray, v_out: GPUArray, q_dim: int, k_dim: int, v_dim: int) -> None:
    """qkv [rows, q+k+v] -> q_out [rows, q_dim...], k_out, v_out (pre-allocated, any trailing shape)."""
    if qkv.ndim != 2 or qkv.shape[1] != q_dim + k_dim + v_dim:
        raise ValueError(f"split_qkv_batch: qkv {qkv.shape} != [rows, {q_dim + k_dim + v_dim}]")
    rows = qkv.shape[0]
    for o, d, n in ((q_out, q_dim, "q_out"), (k_out, k_dim, "k_out"), (v_out, v_dim, "v_out")):
        if o.size != rows * d or o.dtype != qkv.dtype:
            raise ValueError(f"split_qkv_batch: {n} {o.shape}/{o.dtype} does not hold [{rows}, {d}] of {qkv.dtype}")
    call("pgk_split_qkv_batch", qkv._p, q_out._p, k_out._p, v_out._p, rows, q_dim, k_dim, v_dim, qkv.itemsize, None)


def slice_rows_range_ptr(table: GPUArray, out: GPUArray, start_pos_buf: GPUArray, count: int) -> None:
    """out[0:count, :] = table[start:start+count, :], start read from a device int32."""
    if table.ndim != 2 or out.ndim != 2 or out.shape[1] != table.shape[1] or out.shape[0] < count:
        raise ValueError(f"slice_rows_range_ptr: table {table.shape} / out {out.shape} / count {count} mismatch")
    validate_same_dtype(table, out, "slice_rows_range_ptr")
    if start_pos_buf.dtype != int32:
        raise ValueError("slice_rows_range_ptr: start_pos_buf must be int32")
    call("pgk_slice_rows_range_ptr", table._p, out._p, start_pos_buf._p, count, table.shape[1], table.itemsize, None)
