"""CausalTransformerModel (reference: src/pygpukit/llm/models/causal.py:79-997): the single runtime
model for GPT-2 / Llama / Qwen families; `__call__` -> (hidden, present_kv), `get_logits`, `generate`,
`generate_stream`, plus the fixed-cache decode helpers the decode strategies use.

Differences that keep results identical but fit the hardware: token (and position) embeddings are
gathered on the device (the reference gathers on the host and uploads, causal.py:135-146); logits use
the [vocab, hidden] head weight directly (no transposed copy, causal.py:167-177)."""

from __future__ import annotations

from collections.abc import Generator

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16
from pygpukit_amd.core.factory import from_numpy
from pygpukit_amd.llm.buffers import DecodeBuffers
from pygpukit_amd.llm.config import ModelSpec, TransformerConfig
from pygpukit_amd.llm.layers import MLP, Attention, Norm, TransformerBlock
from pygpukit_amd.llm.sampling import sample_token
from pygpukit_amd.ops.basic import (add, add_inplace, embedding_lookup, embedding_lookup_batch, glu_packed, matmul_nt,
                                   rmsnorm, sample_token_gpu)


def _to_float32_logits(logits_np: np.ndarray) -> np.ndarray:
    """bf16 logits travel as uint16 (causal.py:62-71)."""
    if logits_np.dtype == np.uint16:
        return (logits_np.astype(np.uint32) << 16).view(np.float32)
    return logits_np.astype(np.float32)


class CausalTransformerModel:
    def __init__(self, config: TransformerConfig, embed_tokens: GPUArray, blocks: list[TransformerBlock], final_norm: Norm,
                 lm_head: GPUArray | None = None, position_embed: GPUArray | None = None, spec: ModelSpec | None = None):
        self.config = config
        self.embed_tokens = embed_tokens
        self.blocks = blocks
        self.final_norm = final_norm
        self._lm_head = lm_head
        self.position_embed = position_embed
        self.spec = spec

    @property
    def lm_head(self) -> GPUArray | None:
        return self._lm_head

    # ------------------------------------------------------------------ forward
    def _embed(self, input_ids, position_ids) -> GPUArray:
        V = self.embed_tokens.shape[0]
        ids = np.asarray(input_ids, dtype=np.int32)
        if ids.size and (ids.min() < 0 or ids.max() >= V):
            raise ValueError(f"token id outside [0, {V})")
        hidden = GPUArray((len(ids), self.embed_tokens.shape[1]), self.embed_tokens.dtype)
        embedding_lookup_batch(self.embed_tokens, hidden, from_numpy(ids), len(ids))
        if self.position_embed is not None:
            pos = GPUArray(hidden.shape, hidden.dtype)
            embedding_lookup_batch(self.position_embed, pos, from_numpy(np.asarray(position_ids, dtype=np.int32)), len(ids))
            add_inplace(hidden, pos)
        return hidden

    def __call__(self, input_ids: list[int], position_ids: list[int] | None = None,
                 past_key_values: list[tuple | None] | None = None, use_cache: bool = False):
        seq_len = len(input_ids)
        if position_ids is None:
            past_len = 0
            if past_key_values is not None and past_key_values[0] is not None:
                past_len = past_key_values[0][0].shape[0]
            position_ids = list(range(past_len, past_len + seq_len))
        hidden = self._embed(input_ids, position_ids)
        present = []
        for i, block in enumerate(self.blocks):
            past_kv = past_key_values[i] if past_key_values else None
            hidden, kv = block(hidden, position_ids, past_kv, use_cache)
            present.append(kv)
        hidden = self.final_norm(hidden)
        return (hidden, present) if use_cache else (hidden, None)

    forward = __call__  # the "QwenModel.forward()" spelling of the docs

    def get_logits(self, hidden: GPUArray) -> GPUArray:
        """[seq, hidden] -> [seq, vocab]: hidden @ lm_head^T (tied to embed_tokens when lm_head is None)."""
        head = self._lm_head if self._lm_head is not None else self.embed_tokens
        return matmul_nt(hidden, head)

    # ------------------------------------------------------------------ generation
    def _sample(self, logits: GPUArray, temperature, top_k, top_p, gpu_sampling) -> int:
        last = logits._view((logits.shape[0] - 1) * logits.shape[1], (logits.shape[1],))
        if gpu_sampling:
            return sample_token_gpu(last, temperature, top_k, top_p)
        return sample_token(_to_float32_logits(last.to_numpy()), temperature, top_k, top_p)

    def generate(self, input_ids: list[int], max_new_tokens: int = 20, temperature: float = 1.0, top_k: int = 50,
                 top_p: float = 0.9, eos_token_id: int | None = None, use_cache: bool = True,
                 gpu_sampling: bool = False) -> list[int]:
        """causal.py:179-255: prefill, sample from the last row, then one-token steps with past_key_values."""
        tokens = list(input_ids)
        if use_cache:
            hidden, past = self(tokens, use_cache=True)
            nxt = self._sample(self.get_logits(hidden), temperature, top_k, top_p, gpu_sampling)
            tokens.append(nxt)
            if eos_token_id is not None and nxt == eos_token_id:
                return tokens
            for _ in range(max_new_tokens - 1):
                hidden, past = self([nxt], past_key_values=past, use_cache=True)
                nxt = self._sample(self.get_logits(hidden), temperature, top_k, top_p, gpu_sampling)
                tokens.append(nxt)
                if eos_token_id is not None and nxt == eos_token_id:
                    break
        else:
            for _ in range(max_new_tokens):
                hidden, _ = self(tokens, use_cache=False)
                nxt = self._sample(self.get_logits(hidden), temperature, top_k, top_p, gpu_sampling)
                tokens.append(nxt)
                if eos_token_id is not None and nxt == eos_token_id:
                    break
        return tokens

    def generate_stream(self, input_ids: list[int], max_new_tokens: int = 20, temperature: float = 1.0, top_k: int = 50,
                        top_p: float = 0.9, eos_token_id: int | None = None,
                        gpu_sampling: bool = False) -> Generator[int, None, None]:
        hidden, past = self(list(input_ids), use_cache=True)
        nxt = self._sample(self.get_logits(hidden), temperature, top_k, top_p, gpu_sampling)
        yield nxt
        if eos_token_id is not None and nxt == eos_token_id:
            return
        for _ in range(max_new_tokens - 1):
            hidden, past = self([nxt], past_key_values=past, use_cache=True)
            nxt = self._sample(self.get_logits(hidden), temperature, top_k, top_p, gpu_sampling)
            yield nxt
            if eos_token_id is not None and nxt == eos_token_id:
                return

    # ------------------------------------------------------------------ fixed-cache decode (eager ops path)
    def init_fixed_cache(self, max_seq_len: int, dtype: str | None = None) -> None:
        dt = dtype or str(self.embed_tokens.dtype)
        for block in self.blocks:
            block.attn.init_fixed_cache(max_seq_len, dt)

    def _mlp_forward_zero_alloc(self, mlp: MLP, x: GPUArray, buffers: DecodeBuffers) -> None:
        """MLP of one token into buffers.hidden using the pre-allocated buffers (causal.py:489-514)."""
        if mlp.activation == "silu" and mlp.gate_up_proj is not None and buffers.gate_up_out is not None:
            mlp.gate_up_proj(x, out=buffers.gate_up_out)
            glu_packed(buffers.gate_up_out, mlp.intermediate_size, out=buffers.mlp_gate)
            mlp.down_proj(buffers.mlp_gate, out=buffers.hidden)
        else:
            from pygpukit_amd.ops.basic import copy_to

            copy_to(mlp(x), buffers.hidden)

    def _decode_step_fixed_cache(self, token_id: int, position: int, context_len: int) -> GPUArray:
        """One token through every block against the fixed caches (causal.py:799-840) -> hidden [1, H]."""
        hidden = GPUArray((1, self.embed_tokens.shape[1]), self.embed_tokens.dtype)
        embedding_lookup(self.embed_tokens, hidden, token_id)
        for block in self.blocks:
            attn_out = block.attn.forward_fixed_cache(block.attn_norm(hidden), position, context_len)
            hidden = add(hidden, attn_out)
            hidden = add(hidden, block.mlp(block.mlp_norm(hidden)))
        return self.final_norm(hidden)

    def _decode_step_fixed_cache_batch(self, token_ids: list[int], start_position: int, context_len: int) -> GPUArray:
        """M consecutive tokens of one sequence (speculative verify, causal.py:842-891) -> hidden [M, H]."""
        hidden = self._embed(token_ids, list(range(start_position, start_position + len(token_ids))))
        for block in self.blocks:
            attn_out = block.attn.forward_fixed_cache_batch(block.attn_norm(hidden), start_position, context_len)
            hidden = add(hidden, attn_out)
            hidden = add(hidden, block.mlp(block.mlp_norm(hidden)))
        return self.final_norm(hidden)

    def prefill_fixed_cache(self, input_ids: list[int]) -> GPUArray:
        """Prefill that leaves K/V in the fixed caches (the chat flow's prefill + kv_cache_prefill_gqa,
        examples/chat/chat_cli.py:531-534) -> logits [S, V]."""
        from pygpukit_amd.ops.basic import kv_cache_prefill_gqa

        hidden, present = self(list(input_ids), use_cache=True)
        for block, (k, v) in zip(self.blocks, present):
            kv_cache_prefill_gqa(k, block.attn._k_cache, block.attn.num_heads, 0)
            kv_cache_prefill_gqa(v, block.attn._v_cache, block.attn.num_heads, 0)
        return self.get_logits(hidden)

    def snapshot_kv_cache(self) -> list[tuple[np.ndarray, np.ndarray]]:
        return [(b.attn._k_cache.to_numpy(), b.attn._v_cache.to_numpy()) for b in self.blocks]

    def restore_kv_cache(self, snapshot: list[tuple[np.ndarray, np.ndarray]]) -> None:
        for b, (k, v) in zip(self.blocks, snapshot):
            b.attn._k_cache.copy_from_numpy(k)
            b.attn._v_cache.copy_from_numpy(v)

    # ------------------------------------------------------------------ native engine
    def build_engine(self, max_seq_len: int = 512, max_batch: int = 1):
        """Hand this model's weights (zero copy) to the native decode/prefill engine.  Requires bf16 or fp8
        linears, RMSNorm, SwiGLU, RoPE and no biases (Llama / Qwen3 families)."""
        from pygpukit_amd.llm.engine import Engine
        from pygpukit_amd.llm.layers.linear import LinearFP8

        c = self.config
        if c.norm_type != "rmsnorm" or c.activation != "silu" or not c.use_rope or self.position_embed is not None:
            raise NotImplementedError("the native engine covers RMSNorm + SwiGLU + RoPE models (Llama / Qwen families)")
        if self.embed_tokens.dtype != bfloat16:
            raise NotImplementedError("the native engine needs bfloat16 weights")
        layers, fp8 = [], False
        for b in self.blocks:
            a, m = b.attn, b.mlp
            if any(l.bias is not None for l in (a.q_proj, a.k_proj, a.v_proj, a.o_proj)):
                raise NotImplementedError("the native engine does not take projection biases")
            if a.qkv_proj is None or m.gate_up_proj is None:
                fp8 = True
                raise NotImplementedError("build_engine from LinearFP8 layers: use llm.synthetic.build_engine_from_weights")
            layers.append(dict(attn_norm=b.attn_norm.weight, w_qkv=a.qkv_proj.weight,
                               q_norm=a.q_norm.weight if a.q_norm else None, k_norm=a.k_norm.weight if a.k_norm else None,
                               w_o=a.o_proj.weight, mlp_norm=b.mlp_norm.weight, w_gate_up=m.gate_up_proj.weight,
                               w_down=m.down_proj.weight))
        eps = self.blocks[0].attn_norm.eps
        cfg = dict(vocab_size=self.embed_tokens.shape[0], hidden_size=c.hidden_size, num_layers=len(self.blocks),
                   num_heads=c.num_heads, num_kv_heads=c.num_kv_heads, head_dim=c.head_dim,
                   intermediate_size=self.blocks[0].mlp.intermediate_size, norm_eps=eps, rope_theta=c.rope_theta)
        return Engine(cfg, self.embed_tokens, layers, self.final_norm.weight, self._lm_head, max_seq_len=max_seq_len,
                      max_batch=max_batch, weight_format="fp8" if fp8 else "bf16",
                      use_qk_norm=self.blocks[0].attn.q_norm is not None)


GPT2Model = CausalTransformerModel
LlamaModel = CausalTransformerModel
QwenModel = CausalTransformerModel

__all__ = ["CausalTransformerModel", "GPT2Model", "LlamaModel", "QwenModel"]
