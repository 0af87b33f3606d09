"""Paged KV cache and continuous-batching helpers (reference: native/ops/ops.cuh:466-563, bound in
native/bindings/nn/*; the reference ships no Python wrappers for them, so the names are the native ones)."""

from __future__ import annotations

import ctypes as C

import numpy as np

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, float16, int32
from pygpukit_amd.core.factory import from_numpy, zeros
from pygpukit_amd.ops._common import call, check_out


def allocate_kv_cache(num_blocks: int, num_kv_heads: int, block_size: int, head_dim: int, dtype=float16) -> GPUArray:
    """[num_blocks, num_kv_heads, block_size, head_dim], zero-filled (the reference allocates FP16)."""
    return zeros((num_blocks, num_kv_heads, block_size, head_dim), dtype)


def _check_cache(k_cache: GPUArray, v_cache: GPUArray, name: str):
    if k_cache.ndim != 4 or v_cache.shape != k_cache.shape or k_cache.dtype != v_cache.dtype:
        raise ValueError(f"{name}: K/V caches must be matching [num_blocks, num_kv_heads, block_size, head_dim] arrays")
    if k_cache.dtype not in (float16, bfloat16):
        raise ValueError(f"{name}: cache dtype must be float16 or bfloat16, got {k_cache.dtype}")


def paged_attention_v1(q: GPUArray, k_cache: GPUArray, v_cache: GPUArray, block_tables: GPUArray, context_lens: GPUArray,
                       scale: float = 0.0, *, out: GPUArray | None = None, max_context: int | None = None) -> GPUArray:
    """Single-query attention over a paged cache.  q [num_seqs, num_heads, head_dim] -> same shape.
    max_context bounds context_lens (default: every page of the table); it only sizes the KV split."""
    _check_cache(k_cache, v_cache, "paged_attention_v1")
    if q.ndim != 3 or q.dtype != k_cache.dtype or q.shape[2] != k_cache.shape[3]:
        raise ValueError(f"paged_attention_v1: q {q.shape} {q.dtype} does not match the cache {k_cache.shape} {k_cache.dtype}")
    num_seqs, num_heads, head_dim = q.shape
    _, num_kv_heads, block_size, _ = k_cache.shape
    if block_tables.dtype != int32 or block_tables.ndim != 2 or block_tables.shape[0] != num_seqs:
        raise ValueError("paged_attention_v1: block_tables must be int32 [num_seqs, max_blocks_per_seq]")
    if context_lens.dtype != int32 or context_lens.size != num_seqs:
        raise ValueError("paged_attention_v1: context_lens must be int32 [num_seqs]")
    if num_heads % num_kv_heads:
        raise ValueError("paged_attention_v1: num_heads must be a multiple of num_kv_heads")
    max_blocks = block_tables.shape[1]
    max_context = max_blocks * block_size if max_context is None else max_context
    o = check_out(out, q.shape, q.dtype, "paged_attention_v1")
    ws_bytes = _hip.load().pgk_paged_attention_workspace_bytes(num_seqs, num_heads, head_dim, max_context)
    ws = GPUArray((max(ws_bytes // 4, 1),), _f32()) if ws_bytes else None
    call("pgk_paged_attention_v1", q._p, k_cache._p, v_cache._p, block_tables._p, context_lens._p, o._p, num_seqs, num_heads,
         num_kv_heads, head_dim, block_size, max_blocks, max_context, C.c_float(scale), ws._p if ws is not None else None,
         q.dtype.code, None)
    return o


def _f32():
    from pygpukit_amd.core.dtypes import float32

    return float32


def _cache_write(k: GPUArray, v: GPUArray, k_cache: GPUArray, v_cache: GPUArray, slot_mapping: GPUArray, name: str) -> None:
    _check_cache(k_cache, v_cache, name)
    if k.ndim != 3 or v.shape != k.shape or k.dtype != k_cache.dtype or v.dtype != k_cache.dtype:
        raise ValueError(f"{name}: K/V must be matching [tokens, num_kv_heads, head_dim] arrays of the cache dtype")
    if k.shape[1] != k_cache.shape[1] or k.shape[2] != k_cache.shape[3]:
        raise ValueError(f"{name}: K {k.shape} does not match the cache {k_cache.shape}")
    if slot_mapping.dtype != int32 or slot_mapping.size != k.shape[0]:
        raise ValueError(f"{name}: slot_mapping must be int32 [tokens]")
    call("pgk_paged_cache_write", k._p, v._p, k_cache._p, v_cache._p, slot_mapping._p, k.shape[0], k.shape[1], k_cache.shape[2],
         k.shape[2], k.itemsize, None)


def copy_to_paged_cache(k_new: GPUArray, v_new: GPUArray, k_cache: GPUArray, v_cache: GPUArray, slot_mapping: GPUArray) -> None:
    """Decode phase: one new K/V row per sequence into its physical slot (block * block_size + offset)."""
    _cache_write(k_new, v_new, k_cache, v_cache, slot_mapping, "copy_to_paged_cache")


def reshape_and_cache(k: GPUArray, v: GPUArray, k_cache: GPUArray, v_cache: GPUArray, slot_mapping: GPUArray) -> None:
    """Prefill phase: [batch * seq_len, num_kv_heads, head_dim] rows into their slots."""
    _cache_write(k, v, k_cache, v_cache, slot_mapping, "reshape_and_cache")


def gather_embeddings(token_ids: GPUArray, embeddings: GPUArray, total_tokens: int) -> GPUArray:
    if token_ids.dtype != int32 or token_ids.size < total_tokens or embeddings.ndim != 2:
        raise ValueError("gather_embeddings: need int32 ids [total_tokens] and a 2D table")
    out = GPUArray((total_tokens, embeddings.shape[1]), embeddings.dtype)
    call("pgk_embedding_lookup", embeddings._p, out._p, embeddings.shape[1], embeddings.itemsize, 0, token_ids._p, total_tokens, None)
    return out


def scatter_last_token_logits(logits: GPUArray, seq_start_positions: GPUArray, seq_lens: GPUArray, batch_size: int,
                              vocab_size: int) -> GPUArray:
    if logits.ndim != 2 or logits.shape[1] != vocab_size:
        raise ValueError(f"scatter_last_token_logits: logits must be [batch_tokens, {vocab_size}]")
    if seq_start_positions.dtype != int32 or seq_lens.dtype != int32:
        raise ValueError("scatter_last_token_logits: positions / lengths must be int32")
    out = GPUArray((batch_size, vocab_size), logits.dtype)
    call("pgk_scatter_last_token_logits", logits._p, out._p, seq_start_positions._p, seq_lens._p, batch_size, vocab_size,
         logits.itemsize, None)
    return out


def prepare_position_ids(seq_start_positions: GPUArray, seq_context_lens: GPUArray, is_prefill: GPUArray, input_lens: GPUArray,
                         batch_size: int, total_tokens: int) -> GPUArray:
    for a in (seq_start_positions, seq_context_lens, is_prefill, input_lens):
        if a.dtype != int32 or a.size < batch_size:
            raise ValueError("prepare_position_ids: all inputs must be int32 [batch_size]")
    out = zeros((total_tokens,), int32)
    call("pgk_prepare_position_ids", seq_start_positions._p, seq_context_lens._p, is_prefill._p, input_lens._p, out._p, batch_size, None)
    return out


def argmax_sample(logits: GPUArray, batch_size: int, vocab_size: int) -> GPUArray:
    """Greedy token per row -> int32 [batch_size]; lowest index on ties (np.argmax)."""
    if logits.ndim != 2 or logits.shape != (batch_size, vocab_size):
        raise ValueError(f"argmax_sample: logits must be [{batch_size}, {vocab_size}]")
    out = GPUArray((batch_size,), int32)
    call("pgk_argmax", logits._p, batch_size, vocab_size, logits.dtype.code, out._p, None)
    return out


def check_eos(tokens: GPUArray, eos_token_id: int) -> GPUArray:
    if tokens.dtype != int32:
        raise ValueError("check_eos: tokens must be int32")
    out = GPUArray((tokens.size,), int32)
    call("pgk_check_eos", tokens._p, out._p, tokens.size, int(eos_token_id), None)
    return out


def compute_cumsum(input: GPUArray) -> GPUArray:  # noqa: A002  (reference argument name)
    """Exclusive prefix sum of an int32 vector (sequence start positions from sequence lengths)."""
    if input.dtype != int32 or input.ndim != 1:
        raise ValueError("compute_cumsum: expected a 1D int32 array")
    out = GPUArray(input.shape, int32)
    call("pgk_exclusive_cumsum_i32", input._p, out._p, input.size, None)
    return out


def prepare_batch_inputs(token_lists: list[list[int]]) -> tuple[GPUArray, int]:
    """Flatten per-sequence token lists into one int32 device vector; returns (token_ids, total_tokens)."""
    flat = np.fromiter((t for seq in token_lists for t in seq), dtype=np.int32)
    if flat.size == 0:
        raise ValueError("prepare_batch_inputs: no tokens")
    return from_numpy(flat), int(flat.size)


__all__ = ["allocate_kv_cache", "paged_attention_v1", "copy_to_paged_cache", "reshape_and_cache", "gather_embeddings",
           "scatter_last_token_logits", "prepare_position_ids", "argmax_sample", "check_eos", "compute_cumsum", "prepare_batch_inputs"]
