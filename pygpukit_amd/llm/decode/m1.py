"""Single-token decode, eager (reference: src/pygpukit/llm/decode/m1.py:27-121): one op call per
kernel through the operator surface, against the model's fixed KV caches.  Same sequence of ops as
the reference minus the copy_to shuffles its buffer plumbing needs."""

from __future__ import annotations

from pygpukit_amd.llm.decode.base import DecodeStrategy
from pygpukit_amd.ops.basic import add_inplace, embedding_lookup, matmul_nt, rmsnorm


class DecodeM1(DecodeStrategy):
    def step(self, token_id: int, position: int, context_len: int, buffers):
        """-> logits [1, vocab] (buffers.logits)."""
        model = self.model
        embedding_lookup(model.embed_tokens, buffers.hidden, token_id)
        for block in model.blocks:
            rmsnorm(buffers.hidden, block.attn_norm.weight, block.attn_norm.eps, out=buffers.norm_out)
            attn_out = block.attn.forward_fixed_cache(buffers.norm_out, position, context_len, out=buffers.attn_out)
            add_inplace(buffers.hidden, attn_out)
            rmsnorm(buffers.hidden, block.mlp_norm.weight, block.mlp_norm.eps, out=buffers.norm_out)
            buffers.residual, buffers.hidden = buffers.hidden, buffers.residual  # MLP writes buffers.hidden
            model._mlp_forward_zero_alloc(block.mlp, buffers.norm_out, buffers)
            add_inplace(buffers.hidden, buffers.residual)
        rmsnorm(buffers.hidden, model.final_norm.weight, model.final_norm.eps, out=buffers.norm_out)
        assert buffers.logits is not None, "logits buffer not allocated"
        head = model._lm_head if model._lm_head is not None else model.embed_tokens
        matmul_nt(buffers.norm_out, head, out=buffers.logits)
        return buffers.logits
