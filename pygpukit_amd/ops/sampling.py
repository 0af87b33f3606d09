"""Sampling ops (reference: src/pygpukit/ops/sampling.py:11-141 -> ops.cuh:572-628).  Greedy is on the
device; temperature / top-k / top-p sampling ("next") raises until it is built."""

from __future__ import annotations

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.ops.reduction import argmax


def sample_greedy(logits: GPUArray) -> int:
    """argmax over [vocab] or [1, vocab]; lowest index wins ties (np.argmax), unlike the reference's
    CUDA kernel whose tie-break depends on the thread layout (sampling_kernels.cuh:55-198)."""
    return argmax(logits)


def sample_token_gpu(logits: GPUArray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> int:
    if temperature == 0:
        return sample_greedy(logits)
    raise NotImplementedError("sample_token_gpu: only greedy (temperature=0) is implemented on the device; "
                              "use pygpukit_amd.llm.sampling.sample_token on host logits for stochastic sampling")
