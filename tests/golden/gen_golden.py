"""Generate golden vectors by IMPORTING the reference (runs only in the build container,
where /root/reference is mounted read-only).  Outputs are small .npz fixtures committed
next to this script; nothing here runs on the GPU box.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src:. \
        python3 tests/golden/gen_golden.py

Weights are NOT stored: they are re-drawn from the seed with oracle.cpu_ref.make_*_weights
(identical generator in the tests); fixtures hold inputs that are not seed-derived,
expected outputs, and a weight checksum.
"""

from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from oracle import cpu_ref as O  # noqa: E402  (weight generators / bf16 helpers only)

import pygpukit  # noqa: E402  (the reference, via PYTHONPATH=/root/reference/src)
from pygpukit.core.factory import from_numpy  # noqa: E402
from pygpukit.llm import (  # noqa: E402
    MLP, Attention, CausalTransformerModel, Norm, TransformerBlock, TransformerConfig,
)
from pygpukit.llm.config import GPT2_SPEC, QWEN3_SPEC  # noqa: E402
from pygpukit.llm.layers.rope import precompute_freqs_cis  # noqa: E402
from pygpukit.llm.quant import dequantize_fp8_e4m3_block  # noqa: E402
from pygpukit.llm.sampling import sample_token  # noqa: E402
from pygpukit.ops import basic as B  # noqa: E402

assert "/root/reference" in pygpukit.__file__, pygpukit.__file__
G = from_numpy


def checksum(weights) -> float:
    acc = 0.0
    stack = [weights]
    while stack:
        w = stack.pop()
        if isinstance(w, dict):
            stack.extend(w[k] for k in sorted(w))
        elif isinstance(w, list):
            stack.extend(w)
        else:
            acc += float(np.sum(w.astype(np.float64)))
    return acc


def gen_ops():
    rng = np.random.default_rng(1234)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    out = {}
    # matmul, incl. odd sizes and M in {1,16,128}
    for tag, (m, k, n) in {"m1": (1, 64, 48), "m16": (16, 96, 80), "m128": (128, 128, 64),
                           "odd": (7, 33, 19)}.items():
        a, b = f(m, k), f(k, n)
        out[f"matmul_{tag}_a"], out[f"matmul_{tag}_b"] = a, b
        out[f"matmul_{tag}_c"] = B.matmul(G(a), G(b)).to_numpy()
    a = f(5, 9)
    out["transpose_in"], out["transpose_out"] = a, B.transpose(G(a)).to_numpy()
    x, g, bt = f(6, 128), f(128), f(128)
    out["norm_x"], out["norm_g"], out["norm_b"] = x, g, bt
    out["rmsnorm_1e6"] = B.rmsnorm(G(x), G(g), 1e-6).to_numpy()
    out["rmsnorm_1e5"] = B.rmsnorm(G(x), G(g), 1e-5).to_numpy()
    out["layernorm_1e5"] = B.layernorm(G(x), G(g), G(bt), 1e-5).to_numpy()
    e = f(4, 257) * 3
    out["act_x"] = e
    out["silu"] = B.silu(G(e)).to_numpy()
    out["gelu"] = B.gelu(G(e)).to_numpy()
    out["add"] = B.add(G(e), G(e[::-1].copy())).to_numpy()
    out["mul"] = B.mul(G(e), G(e[::-1].copy())).to_numpy()
    # rope: D=128 theta=1e6, positions 3.., GQA head counts
    S, Hq, Hk, D = 5, 4, 2, 128
    cos, sin = precompute_freqs_cis(D, 64, 1e6)
    pos = [3, 4, 5, 6, 40]
    q, k = f(S, Hq, D), f(S, Hk, D)
    out["rope_q"], out["rope_k"], out["rope_pos"] = q, k, np.array(pos)
    out["rope_cos_tab"], out["rope_sin_tab"] = cos.astype(np.float32), sin.astype(np.float32)
    qg, kg = G(q.copy()), G(k.copy())
    B.rope_inplace(qg, kg, G(cos[pos].astype(np.float32)), G(sin[pos].astype(np.float32)))
    out["rope_q_out"], out["rope_k_out"] = qg.to_numpy(), kg.to_numpy()
    # sdpa: q_len < kv_len (offset mask), and square
    for tag, (h, ql, kl, d) in {"off": (4, 3, 11, 64), "sq": (2, 9, 9, 128), "dec": (4, 1, 37, 128)}.items():
        Q, K, V = f(h, ql, d), f(h, kl, d), f(h, kl, d)
        out[f"sdpa_{tag}_q"], out[f"sdpa_{tag}_k"], out[f"sdpa_{tag}_v"] = Q, K, V
        out[f"sdpa_{tag}_o"] = B.sdpa_causal(G(Q), G(K), G(V)).to_numpy()
    Q, K, V = f(2, 4, 32), f(2, 4, 32), f(2, 4, 32)
    out["sdpa_scale_q"], out["sdpa_scale_k"], out["sdpa_scale_v"] = Q, K, V
    out["sdpa_scale_o"] = B.sdpa_causal(G(Q), G(K), G(V), 0.25).to_numpy()
    t = f(3, 2, 8)
    out["shuffle_in"] = t
    out["repeat_interleave_3"] = B.repeat_interleave_axis1(G(t), 3).to_numpy()
    out["transpose_3d_021"] = B.transpose_3d_021(G(t)).to_numpy()
    out["concat_axis0"] = B.concat_axis0(G(t), G(t[:2].copy())).to_numpy()
    # astype(bf16) round trip (RNE incl. ties) through the reference GPUArray
    v = np.concatenate([f(1000), np.array([1.00390625, 1.01171875, -1.00390625, 0.0, 3.0e38, 1e-40],
                                          np.float32)])
    from pygpukit.core.dtypes import bfloat16, float32
    bits = G(v).astype(bfloat16)
    out["bf16_in"], out["bf16_bits"] = v, bits.to_numpy()
    out["bf16_back"] = bits.astype(float32).to_numpy()
    # host sampler, temperature 0 (argmax of softmax; first max wins) and top-k/top-p masks
    lg = f(8, 500) * 4
    lg[3, 100] = lg[3, 400] = lg[3].max() + 1.0  # exact tie -> lowest index
    out["sample_logits"] = lg
    out["sample_t0"] = np.array([sample_token(r, 0.0, 0, 1.0) for r in lg])
    out["sample_t0_k50_p09"] = np.array([sample_token(r, 0.0, 50, 0.9) for r in lg])
    np.savez_compressed(os.path.join(HERE, "g1_ops.npz"), **out)
    print("g1_ops", len(out))


def gen_fp8():
    rng = np.random.default_rng(99)
    codes = np.array([c for c in range(256) if c not in (0x7F, 0xFF)], np.uint8)
    H, W = 256, 384
    w = rng.choice(codes, size=(H, W)).astype(np.uint8)
    w[0, :254] = codes  # every finite code appears
    scale = (rng.random((H // 128, W // 128)).astype(np.float32) * 0.01 + 1e-3)
    scale_bits = O.f32_to_bf16_bits(scale)
    deq = dequantize_fp8_e4m3_block(w, scale_bits)
    x = rng.standard_normal((3, W)).astype(np.float32)
    y = B.matmul(G(x), B.transpose(G(deq.astype(np.float32)))).to_numpy()
    np.savez_compressed(os.path.join(HERE, "g2_fp8.npz"), codes=w, scale_bits=scale_bits, deq=deq, x=x, y=y)
    print("g2_fp8")


TINY = dict(vocab_size=1024, hidden_size=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
            intermediate_size=512, rope_theta=1e6, norm_eps=1e-6)


def ref_qwen3(cfg, weights, max_pos):
    c = TransformerConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                          num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
                          num_kv_heads=cfg["num_kv_heads"], intermediate_size=cfg["intermediate_size"],
                          _head_dim=cfg["head_dim"], norm_type="rmsnorm", activation="silu", use_rope=True,
                          max_position_embeddings=max_pos, norm_eps=cfg["norm_eps"], rope_theta=cfg["rope_theta"])
    eps = cfg["norm_eps"]
    blocks = []
    for lw in weights["layers"]:
        attn = Attention(G(lw["q"]), G(lw["k"]), G(lw["v"]), G(lw["o"]), c,
                         q_norm=Norm(G(lw["q_norm"]), None, "rmsnorm", eps),
                         k_norm=Norm(G(lw["k_norm"]), None, "rmsnorm", eps))
        mlp = MLP(c, gate_proj=G(lw["gate"]), up_proj=G(lw["up"]), down_proj=G(lw["down"]))
        blocks.append(TransformerBlock(Norm(G(lw["attn_norm"]), None, "rmsnorm", eps), attn,
                                       Norm(G(lw["mlp_norm"]), None, "rmsnorm", eps), mlp))
    return CausalTransformerModel(c, G(weights["embed"]), blocks,
                                  Norm(G(weights["final_norm"]), None, "rmsnorm", eps), None, None, QWEN3_SPEC)


def greedy_with_logits(model, prompt, n_new):
    """Same call sequence as CausalTransformerModel.generate (causal.py:205-239) but
    keeping each step's last-row logits."""
    tokens = list(prompt)
    logits_steps = []
    hidden, past = model(tokens, use_cache=True)
    logits = model.get_logits(hidden).to_numpy()
    prefill_hidden, prefill_logits = hidden.to_numpy(), logits
    nxt = sample_token(logits[-1].astype(np.float32), 0.0, 0, 1.0)
    logits_steps.append(logits[-1])
    tokens.append(nxt)
    for _ in range(n_new - 1):
        hidden, past = model([nxt], past_key_values=past, use_cache=True)
        logits = model.get_logits(hidden).to_numpy()
        nxt = sample_token(logits[-1].astype(np.float32), 0.0, 0, 1.0)
        logits_steps.append(logits[-1])
        tokens.append(nxt)
    return tokens, np.stack(logits_steps), prefill_hidden, prefill_logits


def gen_tiny_qwen3():
    w = O.make_qwen3_weights(TINY, seed=40, bf16=True)
    model = ref_qwen3(TINY, w, max_pos=128)
    prompt = [int(t) for t in np.random.default_rng(6).integers(0, TINY["vocab_size"], 12)]
    tokens, step_logits, ph, pl = greedy_with_logits(model, prompt, 10)
    ids = model.generate(prompt, max_new_tokens=10, temperature=0.0, top_k=0, top_p=1.0)
    assert ids == tokens, (ids, tokens)
    np.savez_compressed(os.path.join(HERE, "g3_tiny_qwen3.npz"), prompt=np.array(prompt), tokens=np.array(tokens),
                        step_logits=step_logits, prefill_hidden=ph, prefill_logits=pl, seed=40,
                        wsum=checksum(w), cfg=np.array(sorted(TINY.items()), dtype=object), allow_pickle=True)
    print("g3_tiny_qwen3", tokens)


def gen_gpt2_small():
    cfg = O.GPT2_SMALL
    w = O.make_gpt2_weights(cfg, seed=0)
    c = TransformerConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_layers=cfg["num_layers"],
                          num_heads=cfg["num_heads"], num_kv_heads=cfg["num_kv_heads"],
                          intermediate_size=cfg["intermediate_size"], norm_type="layernorm", activation="gelu",
                          use_rope=False, max_position_embeddings=cfg["max_position_embeddings"],
                          norm_eps=cfg["norm_eps"])
    H = cfg["hidden_size"]
    one, zero = np.ones(H, np.float32), np.zeros(H, np.float32)
    blocks = []
    for lw in w["layers"]:
        attn = Attention(G(lw["q"]), G(lw["k"]), G(lw["v"]), G(lw["o"]), c)
        mlp = MLP(c, fc1_weight=G(lw["fc1"]), fc2_weight=G(lw["fc2"]))
        blocks.append(TransformerBlock(Norm(G(one), G(zero), "layernorm", 1e-5), attn,
                                       Norm(G(one), G(zero), "layernorm", 1e-5), mlp))
    model = CausalTransformerModel(c, G(w["wte"]), blocks, Norm(G(one), G(zero), "layernorm", 1e-5),
                                   None, G(w["wpe"]), GPT2_SPEC)
    t0 = time.perf_counter()
    ids = model.generate([1, 2, 3, 4], max_new_tokens=16, temperature=0.0, top_k=0, top_p=1.0)
    dt = time.perf_counter() - t0
    np.savez_compressed(os.path.join(HERE, "g4_gpt2_small.npz"), prompt=np.array([1, 2, 3, 4]), tokens=np.array(ids),
                        seed=0, wsum=checksum(w), ref_wall_s=dt)
    print("g4_gpt2_small", ids, f"{dt:.1f}s")


def gen_qwen3_layer():
    cfg = dict(O.QWEN3_0_6B, num_layers=1, vocab_size=64)
    w = O.make_qwen3_weights(cfg, seed=11, bf16=True)
    model = ref_qwen3(cfg, w, max_pos=64)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, cfg["hidden_size"])).astype(np.float32)
    pos = [0, 1, 2, 3, 4, 5]
    y, kv = model.blocks[0](G(x), pos, None, True)
    # one more token against the cache (decode-shaped call)
    x1 = rng.standard_normal((1, cfg["hidden_size"])).astype(np.float32)
    y1, kv1 = model.blocks[0](G(x1), [6], kv, True)
    np.savez_compressed(os.path.join(HERE, "g5_qwen3_layer.npz"), x=x, y=y.to_numpy(), x1=x1, y1=y1.to_numpy(),
                        k=kv1[0].to_numpy(), v=kv1[1].to_numpy(), seed=11, wsum=checksum(w))
    print("g5_qwen3_layer")


def gen_basic():
    """G6: the rest of ops.basic on the reference's NumPy path - unary math, whole-array reductions, softmax,
    sum_axis, clamp, where, sigmoid / tanh / relu2."""
    from pygpukit.ops.nn.activation import relu2

    rng = np.random.default_rng(4321)
    x = (rng.standard_normal((7, 333)) * 2).astype(np.float32)
    pos = np.abs(x) + np.float32(0.01)
    out = {"x": x, "pos": pos}
    for name in ("exp", "relu", "sin", "cos", "abs", "neg", "sigmoid", "tanh"):
        out[name] = getattr(B, name)(G(x)).to_numpy()
    out["relu2"] = relu2(G(x)).to_numpy()
    for name in ("log", "sqrt", "rsqrt"):
        out[name] = getattr(B, name)(G(pos)).to_numpy()
    for name in ("sum", "mean", "max", "min", "argmax"):
        out["red_" + name] = getattr(B, name)(G(x)).to_numpy()
    out["softmax"] = B.softmax(G(x)).to_numpy()
    x3 = (rng.standard_normal((2, 3, 50)) * 3).astype(np.float32)
    out["x3"], out["softmax3"] = x3, B.softmax(G(x3)).to_numpy()
    out["sum_axis0"], out["sum_axis1"] = B.sum_axis(G(x), 0).to_numpy(), B.sum_axis(G(x), 1).to_numpy()
    out["clamp"] = B.clamp(G(x), -0.5, 1.25).to_numpy()
    cond = (rng.random((7, 333)) < 0.4).astype(np.uint8)
    y = rng.standard_normal((7, 333)).astype(np.float32)
    out["cond"], out["y"], out["where"] = cond, y, B.where(G(cond), G(x), G(y)).to_numpy()
    np.savez_compressed(os.path.join(HERE, "g6_basic_ops.npz"), **out)
    print("g6_basic_ops", len(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "fp8", "tiny", "layer", "gpt2", "basic"]
    if "ops" in which:
        gen_ops()
    if "fp8" in which:
        gen_fp8()
    if "tiny" in which:
        gen_tiny_qwen3()
    if "layer" in which:
        gen_qwen3_layer()
    if "gpt2" in which:
        gen_gpt2_small()
    if "basic" in which:
        gen_basic()
