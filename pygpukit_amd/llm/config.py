"""Model configuration (reference: src/pygpukit/llm/config.py:20-507).

TransformerConfig carries the dimensions; ModelSpec is the data-only description of one
architecture family (HF tensor-name patterns + architecture flags + default hyper-parameters).
Only dense families on the hot path are described: GPT-2, Llama, Qwen2, Qwen3 (MoE families are
out of scope)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal


@dataclass(frozen=True)
class ModelSpec:
    name: str
    embed_tokens: str
    position_embed: str | None
    lm_head: str | None
    final_norm: str
    final_norm_bias: str | None
    attn_norm: str
    attn_norm_bias: str | None
    q_proj: str
    k_proj: str
    v_proj: str
    o_proj: str
    q_bias: str | None
    k_bias: str | None
    v_bias: str | None
    o_bias: str | None
    q_norm: str | None
    k_norm: str | None
    mlp_norm: str
    mlp_norm_bias: str | None
    fc1: str | None
    fc1_bias: str | None
    fc2: str | None
    fc2_bias: str | None
    gate_proj: str | None
    up_proj: str | None
    down_proj: str | None
    moe_gate: str | None = None
    expert_gate_proj: str | None = None
    expert_up_proj: str | None = None
    expert_down_proj: str | None = None
    norm_type: Literal["rmsnorm", "layernorm"] = "rmsnorm"
    activation: Literal["gelu", "silu"] = "silu"
    use_rope: bool = True
    use_qk_norm: bool = False
    use_position_embed: bool = False
    qkv_combined: bool = False
    weight_transpose: bool = False
    is_moe: bool = False
    default_norm_eps: float = 1e-5
    default_rope_theta: float = 10000.0
    hf_model_type: str = ""


def _hf_decoder_spec(name: str, *, qk_norm: bool, qkv_bias: bool, eps: float, theta: float) -> ModelSpec:
    """Llama-style HF checkpoint naming: model.layers.{layer}.{self_attn|mlp}.*"""
    L = "model.layers.{layer}."
    A = L + "self_attn."
    return ModelSpec(
        name=name, embed_tokens="model.embed_tokens.weight", position_embed=None, lm_head="lm_head.weight",
        final_norm="model.norm.weight", final_norm_bias=None,
        attn_norm=L + "input_layernorm.weight", attn_norm_bias=None,
        q_proj=A + "q_proj.weight", k_proj=A + "k_proj.weight", v_proj=A + "v_proj.weight", o_proj=A + "o_proj.weight",
        q_bias=A + "q_proj.bias" if qkv_bias else None, k_bias=A + "k_proj.bias" if qkv_bias else None,
        v_bias=A + "v_proj.bias" if qkv_bias else None, o_bias=None,
        q_norm=A + "q_norm.weight" if qk_norm else None, k_norm=A + "k_norm.weight" if qk_norm else None,
        mlp_norm=L + "post_attention_layernorm.weight", mlp_norm_bias=None,
        fc1=None, fc1_bias=None, fc2=None, fc2_bias=None,
        gate_proj=L + "mlp.gate_proj.weight", up_proj=L + "mlp.up_proj.weight", down_proj=L + "mlp.down_proj.weight",
        norm_type="rmsnorm", activation="silu", use_rope=True, use_qk_norm=qk_norm,
        default_norm_eps=eps, default_rope_theta=theta, hf_model_type=name)


LLAMA_SPEC = _hf_decoder_spec("llama", qk_norm=False, qkv_bias=False, eps=1e-5, theta=10000.0)
QWEN2_SPEC = _hf_decoder_spec("qwen2", qk_norm=False, qkv_bias=True, eps=1e-6, theta=1000000.0)
QWEN3_SPEC = _hf_decoder_spec("qwen3", qk_norm=True, qkv_bias=False, eps=1e-6, theta=1000000.0)

_G = "h.{layer}."
GPT2_SPEC = ModelSpec(
    name="gpt2", embed_tokens="wte.weight", position_embed="wpe.weight", lm_head=None,
    final_norm="ln_f.weight", final_norm_bias="ln_f.bias",
    attn_norm=_G + "ln_1.weight", attn_norm_bias=_G + "ln_1.bias",
    q_proj=_G + "attn.c_attn.weight", k_proj=_G + "attn.c_attn.weight", v_proj=_G + "attn.c_attn.weight",
    o_proj=_G + "attn.c_proj.weight",
    q_bias=_G + "attn.c_attn.bias", k_bias=_G + "attn.c_attn.bias", v_bias=_G + "attn.c_attn.bias",
    o_bias=_G + "attn.c_proj.bias", q_norm=None, k_norm=None,
    mlp_norm=_G + "ln_2.weight", mlp_norm_bias=_G + "ln_2.bias",
    fc1=_G + "mlp.c_fc.weight", fc1_bias=_G + "mlp.c_fc.bias", fc2=_G + "mlp.c_proj.weight", fc2_bias=_G + "mlp.c_proj.bias",
    gate_proj=None, up_proj=None, down_proj=None,
    norm_type="layernorm", activation="gelu", use_rope=False, use_qk_norm=False, use_position_embed=True,
    qkv_combined=True, weight_transpose=True, default_norm_eps=1e-5, default_rope_theta=10000.0, hf_model_type="gpt2")

MODEL_SPECS: dict[str, ModelSpec] = {"gpt2": GPT2_SPEC, "llama": LLAMA_SPEC, "qwen3": QWEN3_SPEC, "qwen2": QWEN2_SPEC}


def detect_model_spec(tensor_names: list[str]) -> ModelSpec:
    """Pick the family from checkpoint tensor names (config.py:380-431): QK-norm -> Qwen3, QKV biases ->
    Qwen2, model.embed_tokens -> Llama, wte -> GPT-2.  MoE checkpoints are rejected (out of scope)."""
    names = set(tensor_names)
    if any("block_sparse_moe" in n or "mlp.experts" in n for n in names):
        raise ValueError("MoE checkpoints are not supported by pygpukit_amd")
    if any("q_norm" in n for n in names):
        return QWEN3_SPEC
    if "model.embed_tokens.weight" in names:
        return QWEN2_SPEC if "model.layers.0.self_attn.q_proj.bias" in names else LLAMA_SPEC
    if "wte.weight" in names:
        return GPT2_SPEC
    raise ValueError(f"Cannot detect model type from tensor names. First 10 names: {list(tensor_names)[:10]}")


@dataclass
class TransformerConfig:
    """config.py:440-507.  head_dim defaults to hidden_size // num_heads unless _head_dim is given
    (Qwen3-0.6B: hidden 1024, 16 heads, head_dim 128)."""

    vocab_size: int = 32000
    hidden_size: int = 2048
    num_layers: int = 22
    num_heads: int = 32
    num_kv_heads: int | None = None
    intermediate_size: int | None = None
    _head_dim: int | None = None
    num_experts: int | None = None
    num_experts_per_tok: int = 2
    moe_intermediate_size: int | None = None
    norm_type: Literal["rmsnorm", "layernorm"] = "rmsnorm"
    activation: Literal["gelu", "silu"] = "silu"
    use_rope: bool = True
    causal: bool = True
    max_position_embeddings: int = 2048
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    tie_word_embeddings: bool = True

    def __post_init__(self):
        if self.num_kv_heads is None:
            self.num_kv_heads = self.num_heads
        if self.intermediate_size is None:
            self.intermediate_size = 4 * self.hidden_size
        if self.moe_intermediate_size is None:
            self.moe_intermediate_size = self.intermediate_size

    @property
    def is_moe(self) -> bool:
        return self.num_experts is not None and self.num_experts > 1

    @property
    def head_dim(self) -> int:
        return self._head_dim if self._head_dim is not None else self.hidden_size // self.num_heads

    @property
    def num_kv_groups(self) -> int:
        return self.num_heads // self.num_kv_heads


# ---- legacy per-family configs (config.py:515-620): HF-style field names, converted to TransformerConfig ----------
@dataclass
class GPT2Config:
    vocab_size: int = 50257
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    n_positions: int = 1024
    layer_norm_eps: float = 1e-5

    @property
    def n_inner(self) -> int:
        return 4 * self.n_embd

    def to_transformer_config(self) -> TransformerConfig:
        return TransformerConfig(vocab_size=self.vocab_size, hidden_size=self.n_embd, num_layers=self.n_layer, num_heads=self.n_head,
                                 num_kv_heads=self.n_head, intermediate_size=self.n_inner, norm_type="layernorm", activation="gelu",
                                 use_rope=False, causal=True, max_position_embeddings=self.n_positions, norm_eps=self.layer_norm_eps)


@dataclass
class LlamaConfig:
    vocab_size: int = 32000
    hidden_size: int = 2048
    intermediate_size: int = 5632
    num_hidden_layers: int = 22
    num_attention_heads: int = 32
    num_key_value_heads: int = 4
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def to_transformer_config(self) -> TransformerConfig:
        return TransformerConfig(vocab_size=self.vocab_size, hidden_size=self.hidden_size, num_layers=self.num_hidden_layers,
                                 num_heads=self.num_attention_heads, num_kv_heads=self.num_key_value_heads,
                                 intermediate_size=self.intermediate_size, norm_type="rmsnorm", activation="silu", use_rope=True,
                                 causal=True, max_position_embeddings=self.max_position_embeddings, norm_eps=self.rms_norm_eps,
                                 rope_theta=self.rope_theta)


@dataclass
class Qwen3Config:
    vocab_size: int = 151936
    hidden_size: int = 4096
    intermediate_size: int = 12288
    num_hidden_layers: int = 36
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 128            # not hidden_size // heads: Qwen3-0.6B is 1024 / 16 heads with 128
    max_position_embeddings: int = 40960
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0

    def to_transformer_config(self) -> TransformerConfig:
        """The reference's conversion drops head_dim (config.py:600-616) and so gets 64 for Qwen3-0.6B; it is carried
        through here."""
        return TransformerConfig(vocab_size=self.vocab_size, hidden_size=self.hidden_size, num_layers=self.num_hidden_layers,
                                 num_heads=self.num_attention_heads, num_kv_heads=self.num_key_value_heads,
                                 intermediate_size=self.intermediate_size, _head_dim=self.head_dim, norm_type="rmsnorm",
                                 activation="silu", use_rope=True, causal=True, max_position_embeddings=self.max_position_embeddings,
                                 norm_eps=self.rms_norm_eps, rope_theta=self.rope_theta)
