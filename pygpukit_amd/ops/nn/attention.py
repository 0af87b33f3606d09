"""Causal scaled-dot-product attention (reference: src/pygpukit/ops/nn/attention.py:16-235 ->
ops.cuh:287-300).  K/V may carry fewer heads than Q (un-expanded GQA): kv head = q head // (Hq/Hkv)."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import float32, int32
from pygpukit_amd.ops._common import call, check_out, validate_float

_ws_cache: dict[tuple[int, int, int], GPUArray] = {}


def _check_qkv(Q: GPUArray, K: GPUArray, V: GPUArray, name: str):
    validate_float(Q, name)
    if Q.ndim != 3 or K.ndim != 3 or V.ndim != 3:
        raise ValueError(f"{name} expects 3D inputs [n_heads, seq_len, head_dim]")
    if Q.dtype != K.dtype or Q.dtype != V.dtype:
        raise ValueError(f"{name}: Q, K, V must have same dtype")
    hq, q_len, d = Q.shape
    if K.shape[0] != V.shape[0] or hq % K.shape[0] != 0:
        raise ValueError(f"{name}: n_heads mismatch")
    if K.shape[2] != d or V.shape[2] != d:
        raise ValueError(f"{name}: head_dim mismatch")
    if K.shape[1] != V.shape[1]:
        raise ValueError(f"{name}: K and V seq_len mismatch")
    return hq, K.shape[0], q_len, K.shape[1], d


def sdpa_causal(Q: GPUArray, K: GPUArray, V: GPUArray, scale: float = 0.0, *, out: GPUArray | None = None) -> GPUArray:
    """softmax(Q K^T * scale + causal mask) V; scale <= 0 -> 1/sqrt(head_dim); the mask lets query i see
    kv positions <= (kv_len - q_len) + i."""
    hq, hkv, q_len, kv_len, d = _check_qkv(Q, K, V, "sdpa_causal")
    o = check_out(out, (hq, q_len, d), Q.dtype, "sdpa_causal")
    call("pgk_sdpa_causal", Q._p, K._p, V._p, o._p, hq, hkv, q_len, kv_len, d, float(scale), q_len * d, d, kv_len * d, d,
         q_len * d, d, Q.dtype.code, None)
    return o


def sdpa_causal_strided(q: GPUArray, k: GPUArray, v: GPUArray, out: GPUArray, hq: int, hkv: int, q_len: int, kv_len: int,
                        d: int, q_strides, kv_strides, o_strides, scale: float = 0.0) -> None:
    """Same op on [S,H,D]-layout (or any head/row-strided) buffers: strides are (head, row) in elements."""
    call("pgk_sdpa_causal", q._p, k._p, v._p, out._p, hq, hkv, q_len, kv_len, d, float(scale), q_strides[0], q_strides[1],
         kv_strides[0], kv_strides[1], o_strides[0], o_strides[1], q.dtype.code, None)


def _workspace(hq: int, d: int, max_seq: int) -> GPUArray:
    key = (hq, d, max_seq)
    ws = _ws_cache.get(key)
    if ws is None:
        nbytes = _hip.load().pgk_sdpa_decode_workspace_bytes(hq, d, max_seq)
        ws = GPUArray(((nbytes + 3) // 4,), float32)
        _ws_cache[key] = ws
    return ws


def sdpa_causal_fixed_cache(Q: GPUArray, K: GPUArray, V: GPUArray, out: GPUArray, context_len: int, scale: float = 0.0) -> None:
    """Attention of Q [Hq,q_len,D] over the first context_len rows of the fixed caches K,V [Hc,max_seq,D]."""
    hq, hc, q_len, max_seq, d = _check_qkv(Q, K, V, "sdpa_causal_fixed_cache")
    if out.shape != (hq, q_len, d) or out.dtype != Q.dtype:
        raise ValueError("sdpa_causal_fixed_cache: output shape/dtype mismatch")
    if context_len <= 0 or context_len > max_seq:
        raise ValueError(f"sdpa_causal_fixed_cache: invalid context_len {context_len}")
    ws = _workspace(hq, d, max_seq) if q_len == 1 else None
    call("pgk_sdpa_fixed_cache", Q._p, K._p, V._p, out._p, hq, hc, q_len, max_seq, d, float(scale), context_len, None,
         ws._p if ws is not None else None, Q.dtype.code, None)


def sdpa_causal_fixed_cache_ptr(Q: GPUArray, K: GPUArray, V: GPUArray, out: GPUArray, context_len_buf: GPUArray,
                                max_kv_len: int, scale: float = 0.0) -> None:
    """As above with context_len read from a device int32 (graph replay)."""
    hq, hc, q_len, max_seq, d = _check_qkv(Q, K, V, "sdpa_causal_fixed_cache_ptr")
    if context_len_buf.dtype != int32:
        raise ValueError("sdpa_causal_fixed_cache_ptr: context_len_buf must be int32")
    if q_len != 1:
        raise ValueError("sdpa_causal_fixed_cache_ptr: q_len must be 1")
    ws = _workspace(hq, d, max_seq)
    call("pgk_sdpa_fixed_cache", Q._p, K._p, V._p, out._p, hq, hc, q_len, max_seq, d, float(scale), 0, context_len_buf._p,
         ws._p, Q.dtype.code, None)
