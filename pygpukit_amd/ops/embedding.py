"""Embedding gathers and KV-cache scatter (reference: src/pygpukit/ops/embedding.py:15-190 ->
native/ops/ops.cuh:397-410).  Cache heads may be Hq (the reference's GQA-expanded layout) or Hkv
(un-expanded, half the bytes for Qwen3): the op reads the layout off cache.shape[0]."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import int32
from pygpukit_amd.ops._common import call, validate_same_dtype


def _check_table(table: GPUArray, out: GPUArray, name: str) -> None:
    if table.ndim != 2 or out.ndim != 2 or out.shape[1] != table.shape[1]:
        raise ValueError(f"{name}: table {table.shape} / out {out.shape} mismatch")
    validate_same_dtype(table, out, name)


def embedding_lookup(embed_matrix: GPUArray, out: GPUArray, token_id: int) -> None:
    _check_table(embed_matrix, out, "embedding_lookup")
    if not 0 <= token_id < embed_matrix.shape[0]:
        raise ValueError(f"embedding_lookup: token_id {token_id} outside [0, {embed_matrix.shape[0]})")
    call("pgk_embedding_lookup", embed_matrix._p, out._p, embed_matrix.shape[1], embed_matrix.itemsize, token_id, None, 1, None)


def embedding_lookup_ptr(embed_matrix: GPUArray, out: GPUArray, token_id_buf: GPUArray) -> None:
    _check_table(embed_matrix, out, "embedding_lookup_ptr")
    if token_id_buf.dtype != int32:
        raise ValueError("embedding_lookup_ptr: token_id_buf must be int32")
    call("pgk_embedding_lookup", embed_matrix._p, out._p, embed_matrix.shape[1], embed_matrix.itemsize, 0, token_id_buf._p, 1, None)


def embedding_lookup_batch(embed_matrix: GPUArray, out: GPUArray, token_ids_buf: GPUArray, batch_size: int) -> None:
    _check_table(embed_matrix, out, "embedding_lookup_batch")
    if token_ids_buf.dtype != int32 or token_ids_buf.size < batch_size or out.shape[0] < batch_size:
        raise ValueError("embedding_lookup_batch: need int32 ids and room for batch_size rows")
    call("pgk_embedding_lookup", embed_matrix._p, out._p, embed_matrix.shape[1], embed_matrix.itemsize, 0, token_ids_buf._p,
         batch_size, None)


def _check_kv(new_kv: GPUArray, cache: GPUArray, num_heads: int, name: str):
    if new_kv.ndim != 3 or cache.ndim != 3 or cache.shape[2] != new_kv.shape[2]:
        raise ValueError(f"{name}: new_kv {new_kv.shape} / cache {cache.shape} mismatch")
    validate_same_dtype(new_kv, cache, name)
    hc, hkv = cache.shape[0], new_kv.shape[1]
    if hc % hkv != 0 or (hc != num_heads and hc != hkv):
        raise ValueError(f"{name}: cache heads {hc} must equal num_heads {num_heads} or num_kv_heads {hkv}")
    return hkv, hc


def kv_cache_update_gqa(new_kv: GPUArray, cache: GPUArray, num_heads: int, position: int) -> None:
    hkv, hc = _check_kv(new_kv, cache, num_heads, "kv_cache_update_gqa")
    if new_kv.shape[0] != 1:
        raise ValueError("kv_cache_update_gqa: new_kv must be [1, num_kv_heads, head_dim]")
    if not 0 <= position < cache.shape[1]:
        raise ValueError(f"kv_cache_update_gqa: position {position} outside cache of {cache.shape[1]} rows")
    call("pgk_kv_cache_write", new_kv._p, cache._p, 1, hkv, hc, cache.shape[1], cache.shape[2], cache.itemsize, position, None, None)


def kv_cache_update_gqa_ptr(new_kv: GPUArray, cache: GPUArray, num_heads: int, position_buf: GPUArray) -> None:
    hkv, hc = _check_kv(new_kv, cache, num_heads, "kv_cache_update_gqa_ptr")
    if position_buf.dtype != int32:
        raise ValueError("kv_cache_update_gqa_ptr: position_buf must be int32")
    call("pgk_kv_cache_write", new_kv._p, cache._p, 1, hkv, hc, cache.shape[1], cache.shape[2], cache.itemsize, 0,
         position_buf._p, None)


def kv_cache_prefill_gqa(new_kv: GPUArray, cache: GPUArray, num_heads: int, start_pos: int = 0) -> None:
    hkv, hc = _check_kv(new_kv, cache, num_heads, "kv_cache_prefill_gqa")
    if start_pos < 0 or start_pos + new_kv.shape[0] > cache.shape[1]:
        raise ValueError(f"kv_cache_prefill_gqa: rows {start_pos}..{start_pos + new_kv.shape[0]} outside cache")
    call("pgk_kv_cache_write", new_kv._p, cache._p, new_kv.shape[0], hkv, hc, cache.shape[1], cache.shape[2], cache.itemsize,
         start_pos, None, None)


def kv_cache_update(new_kv: GPUArray, cache: GPUArray, position: int) -> None:
    """new_kv [1, Hkv, D] -> cache[position] of a SEQUENCE-major cache [max_seq, Hkv, D] (embedding.py:78-100;
    the un-expanded layout, as opposed to kv_cache_update_gqa's [Hq, max_seq, D])."""
    if new_kv.ndim != 3 or cache.ndim != 3 or new_kv.shape[0] != 1 or new_kv.shape[1:] != cache.shape[1:] or new_kv.dtype != cache.dtype:
        raise ValueError(f"kv_cache_update: new_kv {new_kv.shape}/{new_kv.dtype} does not fit cache {cache.shape}/{cache.dtype}")
    if not 0 <= position < cache.shape[0]:
        raise ValueError(f"kv_cache_update: position {position} outside cache of {cache.shape[0]}")
    row = new_kv.nbytes
    call("pgk_memcpy_d2d", C.c_void_p(cache.data_ptr() + position * row), new_kv._p, row, None)


def kv_cache_prefill(new_kv: GPUArray, cache: GPUArray, start_pos: int = 0) -> None:
    """new_kv [S, Hkv, D] -> cache[start_pos : start_pos + S] of a sequence-major cache (embedding.py:103-125)."""
    if new_kv.ndim != 3 or cache.ndim != 3 or new_kv.shape[1:] != cache.shape[1:] or new_kv.dtype != cache.dtype:
        raise ValueError(f"kv_cache_prefill: new_kv {new_kv.shape}/{new_kv.dtype} does not fit cache {cache.shape}/{cache.dtype}")
    if start_pos < 0 or start_pos + new_kv.shape[0] > cache.shape[0]:
        raise ValueError(f"kv_cache_prefill: rows {start_pos}..{start_pos + new_kv.shape[0]} outside cache of {cache.shape[0]}")
    row = new_kv.nbytes // max(new_kv.shape[0], 1)
    call("pgk_memcpy_d2d", C.c_void_p(cache.data_ptr() + start_pos * row), new_kv._p, new_kv.nbytes, None)
