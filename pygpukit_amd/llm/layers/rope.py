"""RoPE tables (reference: src/pygpukit/llm/layers/rope.py:13-24): fp32 [max_seq, head_dim] with the
half-table duplicated."""

from __future__ import annotations

import numpy as np


def precompute_freqs_cis(head_dim: int, max_seq_len: int, theta: float = 10000.0) -> tuple[np.ndarray, np.ndarray]:
    inv_freq = 1.0 / (theta ** (np.arange(0, head_dim, 2, dtype=np.float32) / head_dim))
    ang = np.outer(np.arange(max_seq_len, dtype=np.float32), inv_freq)
    cos, sin = np.cos(ang), np.sin(ang)
    return np.concatenate([cos, cos], axis=-1), np.concatenate([sin, sin], axis=-1)


def apply_rotary_pos_emb_numpy(q: np.ndarray, k: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Host-side rotate-half RoPE (llm/layers/rope.py:27-43): q [S, Hq, D], k [S, Hkv, D], cos / sin [S, D] (the
    [cos, cos] / [sin, sin] tables of precompute_freqs_cis), broadcast over the head axis."""
    def rot(x):
        half = x.shape[-1] // 2
        return np.concatenate([-x[..., half:], x[..., :half]], axis=-1)

    c, s_ = cos[:, None, :], sin[:, None, :]
    return q * c + rot(q) * s_, k * c + rot(k) * s_
