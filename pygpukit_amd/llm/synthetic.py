"""Random-init model builders for the BASELINE shapes (Qwen3-0.6B, Llama-3-8B) and for tests: take the
fp32 (bf16-rounded) weight dict produced by the shared generator (same draws as the CPU oracle uses) and
put it on the GPU, either as an Engine or as a CausalTransformerModel."""

from __future__ import annotations

import numpy as np

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.factory import from_numpy


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _bf16(x: np.ndarray) -> GPUArray:
    return from_numpy(f32_to_bf16_bits(x).reshape(x.shape))


def engine_layer_arrays(lw: dict, weight_format: str = "bf16") -> dict:
    """One layer of the oracle's weight dict (q,k,v,o,gate,up,down,*_norm as fp32 ndarrays) -> engine layout."""
    from pygpukit_amd.ops.matmul.fp8 import quantize_fp8_blocks

    out = {"attn_norm": _bf16(lw["attn_norm"]), "mlp_norm": _bf16(lw["mlp_norm"]),
           "q_norm": _bf16(lw["q_norm"]) if "q_norm" in lw else None, "k_norm": _bf16(lw["k_norm"]) if "k_norm" in lw else None}
    mats = {"w_qkv": np.concatenate([lw["q"], lw["k"], lw["v"]], axis=0), "w_o": lw["o"],
            "w_gate_up": np.concatenate([lw["gate"], lw["up"]], axis=0), "w_down": lw["down"]}
    for name, m in mats.items():
        if weight_format in ("fp8", "fp8a8"):
            # block-quantised on the device (pgk_quantize_fp8_blocks: value-identical to the oracle's host quantiser,
            # tests/test_gpu_ops.py::test_quantize_fp8_blocks_matches_oracle) - seconds instead of minutes at 0.6B+
            out[name], out["s" + name[1:]] = quantize_fp8_blocks(_bf16(m))
        else:
            out[name] = _bf16(m)
    return out


def build_engine_from_weights(cfg: dict, weights: dict, *, max_seq_len: int = 512, max_batch: int = 1,
                              weight_format: str = "bf16"):
    from pygpukit_amd.llm.engine import Engine

    layers = [engine_layer_arrays(lw, weight_format) for lw in weights["layers"]]
    return Engine(cfg, _bf16(weights["embed"]), layers, _bf16(weights["final_norm"]), None, max_seq_len=max_seq_len,
                  max_batch=max_batch, weight_format=weight_format, use_qk_norm="q_norm" in weights["layers"][0])


def build_model_from_weights(cfg: dict, weights: dict, *, dtype: str = "bfloat16", max_pos: int = 2048):
    """CausalTransformerModel (Qwen3-style) on the GPU from the oracle weight dict."""
    from pygpukit_amd.llm.config import QWEN3_SPEC, TransformerConfig
    from pygpukit_amd.llm.layers import MLP, Attention, Norm, TransformerBlock
    from pygpukit_amd.llm.models.causal import CausalTransformerModel

    def W(x):
        return _bf16(x) if dtype == "bfloat16" else from_numpy(np.ascontiguousarray(x, dtype=np.float32))

    c = TransformerConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_layers=cfg["num_layers"],
                          num_heads=cfg["num_heads"], num_kv_heads=cfg["num_kv_heads"], intermediate_size=cfg["intermediate_size"],
                          _head_dim=cfg["head_dim"], norm_type="rmsnorm", activation="silu", use_rope=True,
                          max_position_embeddings=max_pos, norm_eps=cfg["norm_eps"], rope_theta=cfg["rope_theta"])
    eps = cfg["norm_eps"]
    blocks = []
    for lw in weights["layers"]:
        attn = Attention(W(lw["q"]), W(lw["k"]), W(lw["v"]), W(lw["o"]), c, q_norm=Norm(W(lw["q_norm"]), None, "rmsnorm", eps),
                         k_norm=Norm(W(lw["k_norm"]), None, "rmsnorm", eps))
        mlp = MLP(c, gate_proj=W(lw["gate"]), up_proj=W(lw["up"]), down_proj=W(lw["down"]))
        blocks.append(TransformerBlock(Norm(W(lw["attn_norm"]), None, "rmsnorm", eps), attn,
                                       Norm(W(lw["mlp_norm"]), None, "rmsnorm", eps), mlp))
    return CausalTransformerModel(c, W(weights["embed"]), blocks, Norm(W(weights["final_norm"]), None, "rmsnorm", eps), None, None,
                                  QWEN3_SPEC)


def build_gpt2_from_weights(cfg: dict, weights: dict, *, dtype: str = "float32", biases: dict | None = None):
    """CausalTransformerModel (GPT-2 style: learned positions, LayerNorm, GELU MLP, no RoPE) on the GPU from the
    weight dict of BASELINE config 1 ({"wte", "wpe", "layers": [{q,k,v,o,fc1,fc2}]}); LayerNorm gamma=1, beta=0.
    `biases` (optional, {"q","k","v","o","fc1","fc2"} -> per-layer list of vectors) adds Linear biases, which the
    reference's CPU path cannot run but real GPT-2 checkpoints carry."""
    from pygpukit_amd.llm.config import GPT2_SPEC, TransformerConfig
    from pygpukit_amd.llm.layers import MLP, Attention, Norm, TransformerBlock
    from pygpukit_amd.llm.models.causal import CausalTransformerModel

    def W(x):
        return _bf16(x) if dtype == "bfloat16" else from_numpy(np.ascontiguousarray(x, dtype=np.float32))

    H, eps = cfg["hidden_size"], cfg["norm_eps"]
    c = TransformerConfig(vocab_size=cfg["vocab_size"], hidden_size=H, num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
                          num_kv_heads=cfg["num_kv_heads"], intermediate_size=cfg["intermediate_size"], _head_dim=cfg["head_dim"],
                          norm_type="layernorm", activation="gelu", use_rope=False, causal=True,
                          max_position_embeddings=cfg["max_position_embeddings"], norm_eps=eps)
    one, zero = np.ones(H, np.float32), np.zeros(H, np.float32)

    def ln():
        return Norm(W(one), W(zero), "layernorm", eps)

    def B(name, i):
        return W(biases[name][i]) if biases else None

    blocks = []
    for i, lw in enumerate(weights["layers"]):
        attn = Attention(W(lw["q"]), W(lw["k"]), W(lw["v"]), W(lw["o"]), c, q_bias=B("q", i), k_bias=B("k", i), v_bias=B("v", i),
                         o_bias=B("o", i))
        mlp = MLP(c, fc1_weight=W(lw["fc1"]), fc1_bias=B("fc1", i), fc2_weight=W(lw["fc2"]), fc2_bias=B("fc2", i))
        blocks.append(TransformerBlock(ln(), attn, ln(), mlp))
    return CausalTransformerModel(c, W(weights["wte"]), blocks, ln(), None, W(weights["wpe"]), GPT2_SPEC)


def make_qwen3_weights(cfg: dict, seed: int = 0, std: float = 0.02) -> dict:
    """Same generator as oracle.cpu_ref.make_qwen3_weights (kept here so bench.py's GPU leg does not import
    the oracle): N(0, std^2) float32 draws in the order embed, then per layer q,k,v,o,gate,up,down, each
    rounded to bf16 and widened back to fp32; norm gammas are 1."""
    rng = np.random.default_rng(seed)
    H, D, I, V = cfg["hidden_size"], cfg["head_dim"], cfg["intermediate_size"], cfg["vocab_size"]
    Hq, Hkv = cfg["num_heads"], cfg["num_kv_heads"]

    def W(*s):
        w = rng.standard_normal(s, dtype=np.float32) * np.float32(std)
        return (f32_to_bf16_bits(w).astype(np.uint32) << 16).view(np.float32).reshape(s)

    out = {"embed": W(V, H), "layers": []}
    for _ in range(cfg["num_layers"]):
        out["layers"].append(dict(q=W(Hq * D, H), k=W(Hkv * D, H), v=W(Hkv * D, H), o=W(H, Hq * D), gate=W(I, H), up=W(I, H),
                                  down=W(H, I), attn_norm=np.ones(H, np.float32), mlp_norm=np.ones(H, np.float32),
                                  q_norm=np.ones(D, np.float32), k_norm=np.ones(D, np.float32)))
    out["final_norm"] = np.ones(H, np.float32)
    return out


QWEN3_0_6B = dict(vocab_size=151936, hidden_size=1024, num_layers=28, num_heads=16, num_kv_heads=8, head_dim=128,
                  intermediate_size=3072, rope_theta=1e6, norm_eps=1e-6)
LLAMA3_8B = dict(vocab_size=128256, hidden_size=4096, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
                 intermediate_size=14336, rope_theta=5e5, norm_eps=1e-5)


def random_engine_weights(cfg: dict, seed: int = 0, *, std: float = 0.02, fp8: bool = False, keep_bf16: bool = True,
                          use_qk_norm: bool = False, threads: int = 8) -> dict:
    """Random-init weights in ENGINE layout for models too large to stage as fp32 on the host (Llama-3-8B shape):
    each fused matrix (qkv, o, gate_up, down) is drawn N(0, std^2) from its own generator
    (SeedSequence [seed, layer, index]), rounded to bf16, uploaded, and - with fp8=True - block-quantised ON THE
    DEVICE by pgk_quantize_fp8_blocks.  Returns {"embed", "final_norm", "bf16": [layer dicts] | None,
    "fp8": [layer dicts] | None}; the two layer lists describe the same model."""
    from concurrent.futures import ThreadPoolExecutor

    from pygpukit_amd.ops.matmul.fp8 import quantize_fp8_blocks

    H, D, I, V = cfg["hidden_size"], cfg["head_dim"], cfg["intermediate_size"], cfg["vocab_size"]
    Hq, Hkv = cfg["num_heads"], cfg["num_kv_heads"]
    shapes = {"w_qkv": ((Hq + 2 * Hkv) * D, H), "w_o": (H, Hq * D), "w_gate_up": (2 * I, H), "w_down": (H, I)}

    def draw(key, shape):
        rng = np.random.default_rng(np.random.SeedSequence([seed, *key]))
        w = rng.standard_normal(shape, dtype=np.float32)
        w *= np.float32(std)
        return f32_to_bf16_bits(w).reshape(shape)

    ones_h, ones_d = _bf16(np.ones(H, np.float32)), _bf16(np.ones(D, np.float32))
    out = {"embed": from_numpy(draw((1 << 20,), (V, H))), "final_norm": ones_h, "bf16": [] if keep_bf16 or not fp8 else None,
           "fp8": [] if fp8 else None}
    jobs = [(l, i, name) for l in range(cfg["num_layers"]) for i, name in enumerate(shapes)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        results = pool.map(lambda j: draw((j[0], j[1]), shapes[j[2]]), jobs)
        cur16, cur8 = {}, {}
        for (l, i, name), bits in zip(jobs, results):
            dev = from_numpy(bits)
            if out["bf16"] is not None:
                cur16[name] = dev
            if fp8:
                cur8[name], cur8["s" + name[1:]] = quantize_fp8_blocks(dev)
            if i == len(shapes) - 1:
                norms = {"attn_norm": ones_h, "mlp_norm": ones_h, "q_norm": ones_d if use_qk_norm else None,
                         "k_norm": ones_d if use_qk_norm else None}
                if out["bf16"] is not None:
                    out["bf16"].append({**cur16, **norms})
                if fp8:
                    out["fp8"].append({**cur8, **norms})
                cur16, cur8 = {}, {}
    return out
