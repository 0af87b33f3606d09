"""Checkpoint loading (reference: src/pygpukit/llm/loader.py:63-612): a .safetensors file (or a sharded index) ->
CausalTransformerModel, architecture and dimensions inferred from tensor names and shapes exactly as the reference does
(detect_model_spec; head_dim from the QK-norm weight or the 128/64/256 probe; rope_theta / rms_norm_eps and the FP8
`quantization_config` from a sibling config.json).  Tensors whose stored dtype equals the target dtype go from the file
mapping straight to device memory; others are converted on the host first.  MoE checkpoints are out of scope."""

from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, float16, float32, uint8
from pygpukit_amd.core.factory import from_numpy
from pygpukit_amd.llm.config import GPT2_SPEC, LLAMA_SPEC, QWEN3_SPEC, ModelSpec, TransformerConfig, detect_model_spec
from pygpukit_amd.llm.layers import MLP, Attention, Norm, TransformerBlock
from pygpukit_amd.llm.layers.linear import LinearBF16, LinearFP8
from pygpukit_amd.llm.models.causal import CausalTransformerModel
from pygpukit_amd.llm.safetensors import Dtype, load_safetensors


@dataclass
class FP8QuantConfig:
    """`quantization_config` of an FP8 checkpoint's config.json (loader.py:29-60)."""
    quant_method: str
    fmt: str
    weight_block_size: tuple[int, int]
    modules_to_not_convert: list[str]

    @classmethod
    def from_config(cls, config: dict) -> "FP8QuantConfig | None":
        qc = config.get("quantization_config")
        if not qc or qc.get("quant_method") != "fp8":
            return None
        bs = qc.get("weight_block_size", [128, 128])
        return cls("fp8", qc.get("fmt", "e4m3"), (int(bs[0]), int(bs[1])), list(qc.get("modules_to_not_convert", [])))


def _read_config(model_path: str) -> dict:
    p = os.path.join(os.path.dirname(os.path.abspath(model_path)), "config.json")
    if not os.path.exists(p):
        return {}
    try:
        with open(p, encoding="utf-8") as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def load_model_from_safetensors(model_path: str, dtype: str = "bfloat16", spec: ModelSpec | None = None,
                                repack_weights: bool = True) -> CausalTransformerModel:
    """loader.py:132-612.  dtype: "float32" | "float16" | "bfloat16" (the reference defaults to float32; bf16 is the
    native format of the checkpoints this path serves).  `repack_weights` is accepted for signature compatibility: arrays
    come from the device pool already."""
    st = load_safetensors(model_path)
    names = set(st.tensor_names)
    target = {"float32": (float32, Dtype.Float32, np.float32), "float16": (float16, Dtype.Float16, np.float16),
              "bfloat16": (bfloat16, Dtype.BFloat16, np.uint16)}.get(dtype)
    if target is None:
        raise ValueError(f"load_model_from_safetensors: unsupported dtype {dtype}")
    target_dt, target_id, target_np = target
    if spec is None:
        spec = detect_model_spec(st.tensor_names)
    hf_config = _read_config(model_path)
    fp8 = FP8QuantConfig.from_config(hf_config)

    def load_tensor(name: str, do_transpose: bool = False) -> GPUArray:
        info = st.tensor_info(name)
        if info.dtype == target_id and not do_transpose:
            out = GPUArray(tuple(info.shape), target_dt)
            st.upload(name, out)                       # file mapping -> device, no host copy
            return out
        if info.dtype in (Dtype.Float8E4M3, Dtype.Float8E5M2):
            raise ValueError(f"{name}: FP8 tensor without quantization_config / _scale_inv companion")
        arr = st.tensor_as_f32(name)
        if do_transpose and arr.ndim == 2:
            arr = arr.T
        arr = np.ascontiguousarray(arr)
        if target_id == Dtype.BFloat16:
            u = arr.view(np.uint32)
            arr = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        else:
            arr = arr.astype(target_np)
        return from_numpy(arr) if target_id != Dtype.BFloat16 else _bf16_array(arr)

    def try_load(name: str | None, do_transpose: bool = False) -> GPUArray | None:
        return load_tensor(name, do_transpose) if name is not None and name in names else None

    def is_fp8_weight(name: str) -> bool:
        return fp8 is not None and name + "_scale_inv" in names

    def load_linear(weight_name: str, bias_name: str | None = None, do_transpose: bool = False):
        bias = try_load(bias_name)
        if is_fp8_weight(weight_name):
            info = st.tensor_info(weight_name)
            codes = GPUArray(tuple(info.shape), uint8)
            st.upload(weight_name, codes)
            sname = weight_name + "_scale_inv"
            sinfo = st.tensor_info(sname)
            if sinfo.dtype == Dtype.BFloat16:
                scale = GPUArray(tuple(sinfo.shape), bfloat16)
                st.upload(sname, scale)
            else:
                s32 = np.ascontiguousarray(st.tensor_as_f32(sname))
                u = s32.view(np.uint32)
                scale = _bf16_array(((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16))
            return LinearFP8(codes, scale, bias, fp8.weight_block_size)
        return LinearBF16(load_tensor(weight_name, do_transpose), bias)

    def nm(pattern: str | None, layer: int) -> str | None:
        return None if pattern is None else pattern.format(layer=layer)

    embed_info = st.tensor_info(spec.embed_tokens)
    vocab_size, hidden_size = embed_info.shape
    num_layers = 0
    while nm(spec.attn_norm, num_layers) in names:
        num_layers += 1
    if num_layers == 0:
        raise ValueError(f"no layers found for spec '{spec.name}' (looked for {nm(spec.attn_norm, 0)})")

    q_info = st.tensor_info(nm(spec.q_proj, 0))
    if spec.qkv_combined:                       # GPT-2: c_attn is [hidden, 3 hidden] (Conv1D layout)
        q_dim = hidden_size
        head_dim = 64 if hidden_size % 64 == 0 else hidden_size
    else:
        q_dim = q_info.shape[0]
        head_dim = 0
        if spec.use_qk_norm and nm(spec.q_norm, 0) in names:
            head_dim = st.tensor_info(nm(spec.q_norm, 0)).shape[0]
        if not head_dim:
            for hd in (128, 64, 256):
                if q_dim % hd == 0 and hidden_size % hd == 0 and 4 <= q_dim // hd <= 128:
                    head_dim = hd
                    break
        if not head_dim:
            head_dim = hf_config.get("head_dim") or hidden_size // int(hf_config.get("num_attention_heads", 1))
    if spec.qkv_combined and "n_head" in hf_config:
        head_dim = hidden_size // int(hf_config["n_head"])
    num_heads = q_dim // head_dim
    num_kv_heads = num_heads if spec.qkv_combined else st.tensor_info(nm(spec.k_proj, 0)).shape[0] // head_dim
    if spec.activation == "silu":
        intermediate = st.tensor_info(nm(spec.gate_proj, 0)).shape[0]
    else:
        fc1 = st.tensor_info(nm(spec.fc1, 0)).shape
        intermediate = fc1[1] if spec.weight_transpose else fc1[0]
    cfg = TransformerConfig(vocab_size=vocab_size, hidden_size=hidden_size, num_layers=num_layers, num_heads=num_heads,
                            num_kv_heads=num_kv_heads, intermediate_size=intermediate,
                            _head_dim=head_dim if head_dim != hidden_size // num_heads else None, norm_type=spec.norm_type,
                            activation=spec.activation, use_rope=spec.use_rope,
                            max_position_embeddings=int(hf_config.get("max_position_embeddings", hf_config.get("n_positions", 2048))),
                            norm_eps=float(hf_config.get("rms_norm_eps", hf_config.get("layer_norm_epsilon", spec.default_norm_eps))),
                            rope_theta=float(hf_config.get("rope_theta", spec.default_rope_theta)))
    eps = cfg.norm_eps

    blocks = []
    for layer in range(num_layers):
        attn_norm = Norm(load_tensor(nm(spec.attn_norm, layer)), try_load(nm(spec.attn_norm_bias, layer)), spec.norm_type, eps)
        mlp_norm = Norm(load_tensor(nm(spec.mlp_norm, layer)), try_load(nm(spec.mlp_norm_bias, layer)), spec.norm_type, eps)
        qn = kn = None
        if spec.use_qk_norm:
            qw, kw = try_load(nm(spec.q_norm, layer)), try_load(nm(spec.k_norm, layer))
            qn = Norm(qw, None, spec.norm_type, eps) if qw is not None else None
            kn = Norm(kw, None, spec.norm_type, eps) if kw is not None else None
        if spec.qkv_combined:
            w = st.tensor_as_f32(nm(spec.q_proj, layer))
            w = w.T if spec.weight_transpose else w                       # -> [3 hidden, hidden]
            b = st.tensor_as_f32(nm(spec.q_bias, layer)) if nm(spec.q_bias, layer) in names else None
            parts = [_to_device(np.ascontiguousarray(w[i * hidden_size:(i + 1) * hidden_size]), target_id, target_np) for i in range(3)]
            biases = [None] * 3 if b is None else [_to_device(np.ascontiguousarray(b[i * hidden_size:(i + 1) * hidden_size]), target_id, target_np)
                                                   for i in range(3)]
            o = load_linear(nm(spec.o_proj, layer), nm(spec.o_bias, layer), spec.weight_transpose)
            attn = Attention(parts[0], parts[1], parts[2], o, cfg, q_bias=biases[0], k_bias=biases[1], v_bias=biases[2])
        else:
            attn = Attention(load_linear(nm(spec.q_proj, layer), nm(spec.q_bias, layer)), load_linear(nm(spec.k_proj, layer), nm(spec.k_bias, layer)),
                             load_linear(nm(spec.v_proj, layer), nm(spec.v_bias, layer)), load_linear(nm(spec.o_proj, layer), nm(spec.o_bias, layer)),
                             cfg, q_norm=qn, k_norm=kn)
        if spec.activation == "silu":
            g, u_, d = (load_linear(nm(spec.gate_proj, layer)), load_linear(nm(spec.up_proj, layer)), load_linear(nm(spec.down_proj, layer)))
            if isinstance(g, LinearBF16) and isinstance(u_, LinearBF16) and g.bias is None and u_.bias is None:
                g, u_ = g.weight, u_.weight                               # lets MLP fuse gate|up into one matrix
            mlp = MLP(cfg, gate_proj=g, up_proj=u_, down_proj=d)
        else:
            mlp = MLP(cfg, fc1_weight=load_tensor(nm(spec.fc1, layer), spec.weight_transpose), fc1_bias=try_load(nm(spec.fc1_bias, layer)),
                      fc2_weight=load_tensor(nm(spec.fc2, layer), spec.weight_transpose), fc2_bias=try_load(nm(spec.fc2_bias, layer)))
        blocks.append(TransformerBlock(attn_norm, attn, mlp_norm, mlp))

    final_norm = Norm(load_tensor(spec.final_norm), try_load(spec.final_norm_bias), spec.norm_type, eps)
    lm_head = try_load(spec.lm_head) if spec.lm_head else None            # tied embeddings when absent
    position_embed = try_load(spec.position_embed) if spec.use_position_embed else None
    return CausalTransformerModel(cfg, load_tensor(spec.embed_tokens), blocks, final_norm, lm_head, position_embed, spec)


def _bf16_array(bits: np.ndarray) -> GPUArray:
    out = GPUArray(tuple(bits.shape), bfloat16)
    out.copy_from_numpy(np.ascontiguousarray(bits, dtype=np.uint16))
    return out


def _to_device(arr_f32: np.ndarray, target_id: int, target_np) -> GPUArray:
    if target_id == Dtype.BFloat16:
        u = np.ascontiguousarray(arr_f32, dtype=np.float32).view(np.uint32)
        return _bf16_array(((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16))
    return from_numpy(np.ascontiguousarray(arr_f32).astype(target_np))


def load_gpt2_from_safetensors(model_path: str, dtype: str = "float32") -> CausalTransformerModel:
    return load_model_from_safetensors(model_path, dtype=dtype, spec=GPT2_SPEC)


def load_llama_from_safetensors(model_path: str, dtype: str = "bfloat16") -> CausalTransformerModel:
    return load_model_from_safetensors(model_path, dtype=dtype, spec=LLAMA_SPEC)


def load_qwen3_from_safetensors(model_path: str, dtype: str = "bfloat16") -> CausalTransformerModel:
    return load_model_from_safetensors(model_path, dtype=dtype, spec=QWEN3_SPEC)


__all__ = ["load_model_from_safetensors", "load_gpt2_from_safetensors", "load_llama_from_safetensors", "load_qwen3_from_safetensors",
           "FP8QuantConfig"]
