"""Sampling ops (reference: src/pygpukit/ops/sampling.py:11-141 -> native/ops/sampling/sampling.cu).  Everything runs on
the device and only the sampled id comes back.  Stochastic sampling is a deterministic function of (logits,
temperature, top_k, top_p, u) defined in csrc/ops_sampling.hip; u is drawn on the host from a seedable generator
(the reference draws it from a thread-local std::mt19937, sampling.cu:16-17) or read from a device buffer."""

from __future__ import annotations

import ctypes as C

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import FLOAT_DTYPES, float32, int32
from pygpukit_amd.ops._common import call
from pygpukit_amd.ops.reduction import argmax_int

_rng = np.random.default_rng()


def set_sampling_seed(seed: int) -> None:
    """Seed the host generator that supplies u (reference: native set_sampling_seed)."""
    global _rng
    _rng = np.random.default_rng(seed)


def _rows_vocab(logits: GPUArray, name: str) -> tuple[int, int]:
    if logits.ndim not in (1, 2):
        raise ValueError(f"{name}: expected 1D or 2D logits, got {logits.ndim}D")
    if logits.dtype not in FLOAT_DTYPES:
        raise ValueError(f"{name}: unsupported dtype {logits.dtype}")
    return (1, logits.shape[0]) if logits.ndim == 1 else (logits.shape[0], logits.shape[1])


def _sample(logits: GPUArray, temperature: float, top_k: int, top_p: float, name: str, u: float | None = None) -> int:
    rows, vocab = _rows_vocab(logits, name)
    if rows != 1:
        raise ValueError(f"{name}: expected [vocab] or [1, vocab] logits, got {logits.shape}")
    if temperature <= 0:
        raise ValueError(f"{name}: temperature must be > 0")
    if u is None:
        u = float(_rng.random(dtype=np.float32))
    out = GPUArray((1,), int32)
    call("pgk_sample_token", logits._p, 1, vocab, logits.dtype.code, C.c_float(temperature), int(top_k), C.c_float(top_p),
         C.c_float(u), None, out._p, None)
    return int(out.to_numpy()[0])


def sample_greedy(logits: GPUArray) -> int:
    """argmax over [vocab] or [1, vocab]; lowest index wins ties (np.argmax), unlike the reference's
    CUDA kernel whose tie-break depends on the thread layout (sampling_kernels.cuh:55-198)."""
    return argmax_int(logits)


def sample_multinomial(logits: GPUArray, temperature: float, *, u: float | None = None) -> int:
    return _sample(logits, temperature, 0, 1.0, "sample_multinomial", u)


def sample_topk(logits: GPUArray, top_k: int, temperature: float, *, u: float | None = None) -> int:
    if top_k <= 0:
        raise ValueError("sample_topk: top_k must be > 0")
    return _sample(logits, temperature, top_k, 1.0, "sample_topk", u)


def sample_topp(logits: GPUArray, top_p: float, temperature: float, *, u: float | None = None) -> int:
    if not 0 < top_p <= 1:
        raise ValueError("sample_topp: top_p must be in (0, 1]")
    return _sample(logits, temperature, 0, top_p, "sample_topp", u)


def sample_token_gpu(logits: GPUArray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, *,
                     u: float | None = None) -> int:
    """temperature=0: greedy; otherwise top-k (if > 0) then top-p (if < 1) then the draw - the order the reference's
    host sampler applies them in (src/pygpukit/llm/sampling.py:36-55)."""
    if temperature == 0:
        return sample_greedy(logits)
    return _sample(logits, temperature, top_k, top_p, "sample_token_gpu", u)


def sample_topk_to_buf_ptr(logits: GPUArray, result_buf: GPUArray, random_val_buf: GPUArray, top_k: int, temperature: float) -> None:
    """Top-k sampling whose random number is READ FROM DEVICE MEMORY and whose result stays there: capturable in a
    graph and replayable after updating random_val_buf (reference: sampling.cu:238-290; any float dtype here)."""
    rows, vocab = _rows_vocab(logits, "sample_topk_to_buf_ptr")
    if result_buf.dtype != int32 or result_buf.size < rows:
        raise ValueError("sample_topk_to_buf_ptr: result_buf must be int32 with one slot per row")
    if random_val_buf.dtype != float32:
        raise ValueError("sample_topk_to_buf_ptr: random_val_buf must be float32")
    if temperature <= 0 or top_k <= 0:
        raise ValueError("sample_topk_to_buf_ptr: temperature and top_k must be > 0")
    call("pgk_sample_token", logits._p, rows, vocab, logits.dtype.code, C.c_float(temperature), int(top_k), C.c_float(1.0),
         C.c_float(0.0), random_val_buf._p, result_buf._p, None)


__all__ = ["sample_greedy", "sample_multinomial", "sample_topk", "sample_topp", "sample_token_gpu", "sample_topk_to_buf_ptr",
           "set_sampling_seed"]
