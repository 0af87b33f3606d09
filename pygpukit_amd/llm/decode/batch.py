"""Batch decode (reference: src/pygpukit/llm/decode/batch.py:28-363).

Two things are called "batch" on this path:
  * the reference's DecodeBatch = M consecutive tokens of ONE sequence (speculative verify):
    `step_batch(token_ids, start_position, context_len, buffers)` -> logits [M, vocab];
  * BASELINE config 4 = B INDEPENDENT sequences decoded together (one KV cache each), which the
    reference has no implementation of: `init_graph` / `step_graph_independent` / `run_greedy` drive the
    native engine with batch B; each sequence follows exactly the single-sequence semantics.
The reference's captured batch graph replays baked scalars for the KV write / attention
(batch.py:207-209) - not reproduced: positions are read from device memory."""

from __future__ import annotations

import numpy as np

from pygpukit_amd.llm.decode.base import DecodeStrategy
from pygpukit_amd.ops.basic import matmul_nt


class DecodeBatch(DecodeStrategy):
    def __init__(self, batch_size: int = 8) -> None:
        super().__init__()
        self.batch_size = batch_size
        self._engine = None
        self._graph_ready = False

    def step(self, token_id, position, context_len, buffers):
        return self.step_batch([token_id], position, context_len, buffers)

    def step_batch(self, token_ids: list[int], start_position: int, context_len: int, buffers=None):
        """Verify-style batch: M consecutive tokens of one sequence against the model's fixed caches."""
        model = self.model
        hidden = model._decode_step_fixed_cache_batch(list(token_ids), start_position, context_len)
        head = model._lm_head if model._lm_head is not None else model.embed_tokens
        out = buffers.logits_batch.slice_rows(len(token_ids)) if buffers is not None and buffers.logits_batch is not None else None
        return matmul_nt(hidden, head, out=out)

    def step_graph(self, token_ids: list[int], start_position: int, context_len: int):
        """batch.py:308-363: the verify step for exactly `batch_size` tokens after init_graph() -> logits
        [batch_size, vocab].  Same result as step_batch; nothing is replayed from baked scalars (the reference's
        captured graph freezes the KV-write position): positions are read from device memory."""
        assert self._graph_ready, "Call init_graph() first"
        if len(token_ids) != self.batch_size:
            raise ValueError(f"token_ids length ({len(token_ids)}) must match batch_size ({self.batch_size})")
        return self.step_batch(token_ids, start_position, context_len, None)

    # ---- independent sequences on the native engine ----
    def init_graph(self, max_seq_len: int = 512) -> None:
        self._engine = self.model.build_engine(max_seq_len=max_seq_len, max_batch=self.batch_size)
        self._engine.capture(self.batch_size)
        self._graph_ready = True

    def has_graph(self) -> bool:
        return self._graph_ready

    @property
    def engine(self):
        return self._engine

    def prefill(self, prompts: list[list[int]]) -> np.ndarray:
        """Prefill sequence slot b with prompts[b]; returns the first greedy token of each sequence."""
        assert self._graph_ready, "Call init_graph() first"
        first = [int(np.argmax(self._engine.prefill(p, seq=b, start_pos=0))) for b, p in enumerate(prompts)]
        return np.asarray(first, np.int32)

    def step_graph_independent(self, token_ids, positions):
        """One step for B independent sequences -> fp32 logits [B, vocab]."""
        assert self._graph_ready, "Call init_graph() first"
        self._engine.set_state(token_ids, positions)
        self._engine.replay(1)
        return self._engine.logits(self.batch_size)

    def run_greedy(self, first_tokens, positions, n_steps: int) -> np.ndarray:
        """-> int32 [n_steps, B] greedy tokens."""
        assert self._graph_ready, "Call init_graph() first"
        self._engine.set_state(first_tokens, positions)
        self._engine.reset_log()
        self._engine.replay(n_steps)
        self._engine.synchronize()
        return self._engine.read_tokens(self.batch_size, n_steps)
