"""Graph-captured single-token decode (reference: src/pygpukit/llm/decode/m1_graph.py:44-594).

The reference captures 2L+2 graphs and runs KV update + SDPA eagerly between them with a device sync
after each (m1_graph.py:463-589).  Here the WHOLE step - embedding, every layer including the KV write
and attention, lm_head and greedy argmax - is one hipGraph built by the native engine; token id and
position live in device memory, so `step_graph` is a 8-byte state upload + one graph launch, and
`run_greedy(n)` queues n steps with no host round trip at all."""

from __future__ import annotations

import numpy as np

from pygpukit_amd.llm.decode.base import DecodeStrategy


class DecodeM1Graph(DecodeStrategy):
    def __init__(self) -> None:
        super().__init__()
        self._engine = None
        self._graph_ready = False
        self._graph_max_seq_len = 0
        self._decode_buffers = None

    def step(self, token_id, position, context_len, buffers):
        raise NotImplementedError("DecodeM1Graph does not support non-graph decode. Use DecodeM1, or call "
                                  "init_graph() and step_graph().")

    def init_graph(self, max_seq_len: int = 512) -> None:
        self._engine = self.model.build_engine(max_seq_len=max_seq_len, max_batch=1)
        self._engine.capture(1)
        self._graph_max_seq_len = max_seq_len
        self._decode_buffers = None
        self._graph_ready = True

    def has_graph(self) -> bool:
        return self._graph_ready

    @property
    def engine(self):
        return self._engine

    @property
    def buffers(self):
        """The strategy's DecodeBuffers (reference: m1_graph.py:67,248-262 allocates them in init_graph and its graphs
        run on them).  Here the whole step runs inside the native engine on the engine's own fp32 intermediates, so
        the object exists for code that addresses buffers by name: `logits`, `token_id_buf` and `position_buf` ARE the
        engine's live device state (the step's fp32 logits; the token / position the next replay consumes, updated
        by every replay); the per-layer activation fields have the reference's names and shapes but no step writes
        them."""
        if not self._graph_ready:
            return None
        if self._decode_buffers is None:
            from pygpukit_amd.llm.buffers import DecodeBuffers

            spec = getattr(self.model, "spec", None)
            b = DecodeBuffers.allocate(self.model.config, dtype="bfloat16", use_qk_norm=bool(spec is not None and spec.use_qk_norm))
            b.logits = self._engine.logits(1)
            b.token_id_buf, b.position_buf = self._engine.state_arrays(1)
            self._decode_buffers = b
        return self._decode_buffers

    def prefill(self, input_ids: list[int]) -> np.ndarray:
        """Fill the engine's KV cache from the prompt; returns the last row's fp32 logits."""
        assert self._graph_ready, "Call init_graph() first"
        return self._engine.prefill(input_ids, seq=0, start_pos=0)

    def load_kv_from_model(self, context_len: int) -> None:
        """Copy the model's fixed caches ([Hkv, max_seq, D] per layer, filled by prefill_fixed_cache) into
        the engine's caches - the reference's prefill -> fixed-cache hand-off."""
        from pygpukit_amd.ops.basic import copy_to

        for i, block in enumerate(self.model.blocks):
            if block.attn._k_cache is None or block.attn._max_cache_len != self._graph_max_seq_len:
                raise ValueError("load_kv_from_model: model fixed caches must exist with the graph's max_seq_len")
            k, v = self._engine.kv_cache(i)
            copy_to(block.attn._k_cache, k._view(0, block.attn._k_cache.shape))
            copy_to(block.attn._v_cache, v._view(0, block.attn._v_cache.shape))

    def step_graph(self, token_id: int, position: int, context_len: int):
        """One graph replay for (token_id at `position`; attends over position+1 == context_len rows)
        -> fp32 logits [1, vocab] (engine-owned buffer, valid until the next step)."""
        assert self._graph_ready, "Call init_graph() first"
        if context_len != position + 1:
            raise ValueError(f"step_graph: context_len {context_len} must equal position + 1 ({position + 1})")
        self._engine.set_state([token_id], [position])
        self._engine.replay(1)
        return self._engine.logits(1)

    def run_greedy(self, first_token: int, position: int, n_steps: int) -> list[int]:
        """n_steps device-resident greedy steps starting from first_token at `position`."""
        assert self._graph_ready, "Call init_graph() first"
        self._engine.set_state([first_token], [position])
        self._engine.reset_log()
        self._engine.replay(n_steps)
        self._engine.synchronize()
        return [int(t) for t in self._engine.read_tokens(1, n_steps)[:, 0]]
