"""Shared pytest configuration.

Markers:
  gpu  - needs a real MI355X (run by `pytest -m gpu` on the GPU box, through the C-ABI).
Everything else must pass on a CPU-only container (`pytest -m "not gpu"`).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP path through the C-ABI)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """||a-b||_2 / ||b||_2, the reference's own metric (tests/test_gemv_correctness.py:144-149)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
