"""Host-side token sampling (reference: src/pygpukit/llm/sampling.py:12-63): temperature, top-k,
top-p; temperature == 0 is argmax of the (masked) softmax, first maximum wins."""

from __future__ import annotations

import numpy as np


def _keep_only(probs: np.ndarray, idx: np.ndarray) -> np.ndarray:
    kept = np.zeros_like(probs)
    kept[idx] = probs[idx]
    return kept / kept.sum()


def sample_token(logits: np.ndarray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> int:
    if temperature != 1.0 and temperature > 0:
        logits = logits / temperature
    e = np.exp(logits - logits.max())
    probs = e / e.sum()
    if 0 < top_k < len(probs):
        probs = _keep_only(probs, np.argsort(probs)[-top_k:])
    if top_p < 1.0:
        order = np.argsort(probs)[::-1]
        cut = min(int(np.searchsorted(np.cumsum(probs[order]), top_p)) + 1, len(order))
        probs = _keep_only(probs, order[:cut])
    if temperature == 0:
        return int(np.argmax(probs))
    return int(np.random.choice(len(probs), p=probs))
