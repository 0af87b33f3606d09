"""Configs shared by the golden generator and the tests (must match tests/golden/gen_golden.py)."""
import numpy as np

TINY = dict(vocab_size=1024, hidden_size=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
            intermediate_size=512, rope_theta=1e6, norm_eps=1e-6)


def cfg_checksum(weights) -> float:
    """Same traversal as gen_golden.checksum: float64 sum of every array."""
    acc = 0.0
    stack = [weights]
    while stack:
        w = stack.pop()
        if isinstance(w, dict):
            stack.extend(w[k] for k in sorted(w))
        elif isinstance(w, list):
            stack.extend(w)
        else:
            acc += float(np.sum(w.astype(np.float64)))
    return acc
