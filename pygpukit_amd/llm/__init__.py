"""LLM layer of pygpukit_amd: config, layers, CausalTransformerModel, decode strategies, native engine."""

from pygpukit_amd.llm.buffers import DecodeBuffers, PrefillBuffers
from pygpukit_amd.llm.config import (GPT2_SPEC, LLAMA_SPEC, MODEL_SPECS, QWEN2_SPEC, QWEN3_SPEC, GPT2Config, LlamaConfig, ModelSpec,
                                    Qwen3Config, TransformerConfig, detect_model_spec)
from pygpukit_amd.llm.decode import DecodeBatch, DecodeM1, DecodeM1Graph, DecodeStrategy
from pygpukit_amd.llm.engine import Engine
from pygpukit_amd.llm.layers import (MLP, Attention, Linear, LinearBF16, LinearFP8, Norm, TransformerBlock,
                                    apply_rotary_pos_emb_numpy, precompute_freqs_cis)
from pygpukit_amd.llm.models import CausalTransformerModel, GPT2Model, LlamaModel, QwenModel
from pygpukit_amd.llm.loader import (FP8QuantConfig, load_gpt2_from_safetensors, load_llama_from_safetensors,  # noqa: E402
                                     load_model_from_safetensors, load_qwen3_from_safetensors)
from pygpukit_amd.llm.safetensors import Dtype, SafeTensorsFile, ShardedSafeTensorsFile, TensorInfo, load_safetensors  # noqa: E402
from pygpukit_amd.llm.sampling import sample_token

# legacy component names (models/causal.py:1496-1501)
RMSNorm = LayerNorm = Norm
LlamaAttention = CausalSelfAttention = Attention
LlamaMLP = MLP
LlamaBlock = TransformerBlock

__all__ = ["load_model_from_safetensors", "load_qwen3_from_safetensors", "load_llama_from_safetensors", "load_gpt2_from_safetensors",
           "FP8QuantConfig", "SafeTensorsFile", "ShardedSafeTensorsFile", "TensorInfo", "Dtype", "load_safetensors",
           "DecodeBuffers", "PrefillBuffers", "ModelSpec", "TransformerConfig", "GPT2_SPEC", "LLAMA_SPEC", "QWEN2_SPEC",
           "QWEN3_SPEC", "MODEL_SPECS", "detect_model_spec", "DecodeStrategy", "DecodeM1", "DecodeM1Graph", "DecodeBatch",
           "Engine", "MLP", "Attention", "Linear", "LinearBF16", "LinearFP8", "Norm", "TransformerBlock",
           "precompute_freqs_cis", "CausalTransformerModel", "GPT2Model", "LlamaModel", "QwenModel", "sample_token",
           "GPT2Config", "LlamaConfig", "Qwen3Config", "apply_rotary_pos_emb_numpy", "RMSNorm", "LayerNorm", "LlamaAttention",
           "CausalSelfAttention", "LlamaMLP", "LlamaBlock"]
