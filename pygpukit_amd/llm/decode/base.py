"""Decode strategy base (reference: src/pygpukit/llm/decode/base.py:19-87)."""

from __future__ import annotations

from abc import ABC, abstractmethod


class DecodeStrategy(ABC):
    def __init__(self) -> None:
        self._model = None

    def bind(self, model) -> None:
        self._model = model

    @property
    def model(self):
        if self._model is None:
            raise RuntimeError("Strategy not bound to a model. Call bind() first.")
        return self._model

    @abstractmethod
    def step(self, token_id: int, position: int, context_len: int, buffers):
        """One decode step -> logits."""

    def init_graph(self, max_seq_len: int = 512) -> None:  # noqa: B027
        """Strategies with graph support override this."""

    def has_graph(self) -> bool:
        return False
