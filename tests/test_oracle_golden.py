"""Pin the CPU oracle (oracle/cpu_ref.py) against golden vectors produced by importing
the reference itself (tests/golden/gen_golden.py).  CPU only; bit-exact for fp32."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cpu_ref as O
from tests.conftest import load_golden
from tests.golden_cfg import TINY, cfg_checksum

g1 = load_golden("g1_ops.npz")


def eq(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("tag", ["m1", "m16", "m128", "odd"])
def test_matmul(tag):
    eq(O.matmul(g1[f"matmul_{tag}_a"], g1[f"matmul_{tag}_b"]), g1[f"matmul_{tag}_c"])


def test_transpose():
    eq(O.transpose(g1["transpose_in"]), g1["transpose_out"])


def test_norms():
    x, g, b = g1["norm_x"], g1["norm_g"], g1["norm_b"]
    eq(O.rmsnorm(x, g, 1e-6), g1["rmsnorm_1e6"])
    eq(O.rmsnorm(x, g, 1e-5), g1["rmsnorm_1e5"])
    eq(O.layernorm(x, g, b, 1e-5), g1["layernorm_1e5"])


def test_activations_and_elementwise():
    x = g1["act_x"]
    eq(O.silu(x), g1["silu"])
    eq(O.gelu(x), g1["gelu"])
    eq(O.add(x, x[::-1].copy()), g1["add"])
    eq(O.mul(x, x[::-1].copy()), g1["mul"])


def test_rope_table_and_rotation():
    cos, sin = O.precompute_freqs_cis(128, 64, 1e6)
    eq(cos.astype(np.float32), g1["rope_cos_tab"])
    eq(sin.astype(np.float32), g1["rope_sin_tab"])
    pos = g1["rope_pos"]
    q, k = O.rope(g1["rope_q"], g1["rope_k"], cos[pos].astype(np.float32), sin[pos].astype(np.float32))
    eq(q, g1["rope_q_out"])
    eq(k, g1["rope_k_out"])


@pytest.mark.parametrize("tag", ["off", "sq", "dec"])
def test_sdpa(tag):
    eq(O.sdpa_causal(g1[f"sdpa_{tag}_q"], g1[f"sdpa_{tag}_k"], g1[f"sdpa_{tag}_v"]), g1[f"sdpa_{tag}_o"])


def test_sdpa_explicit_scale():
    eq(O.sdpa_causal(g1["sdpa_scale_q"], g1["sdpa_scale_k"], g1["sdpa_scale_v"], 0.25), g1["sdpa_scale_o"])


def test_sdpa_fixed_cache_equals_sliced_sdpa():
    """kernel-defined fixed-cache SDPA == reference SDPA on the valid prefix (expanded and GQA caches)."""
    q, k, v = g1["sdpa_dec_q"], g1["sdpa_dec_k"], g1["sdpa_dec_v"]
    kc = np.zeros((4, 64, 128), np.float32)
    vc = np.full((4, 64, 128), 7.0, np.float32)
    kc[:, :37], vc[:, :37] = k, v
    eq(O.sdpa_causal_fixed_cache(q, kc, vc, 37), g1["sdpa_dec_o"])
    # un-expanded cache (2 kv heads feeding 4 q heads)
    k2, v2 = k[::2], v[::2]
    ref = O.sdpa_causal(q, np.repeat(k2, 2, axis=0), np.repeat(v2, 2, axis=0))
    kc2 = np.zeros((2, 64, 128), np.float32)
    vc2 = np.zeros((2, 64, 128), np.float32)
    kc2[:, :37], vc2[:, :37] = k2, v2
    eq(O.sdpa_causal_fixed_cache(q, kc2, vc2, 37), ref)


def test_shuffles():
    t = g1["shuffle_in"]
    eq(O.repeat_interleave_axis1(t, 3), g1["repeat_interleave_3"])
    eq(O.transpose_3d_021(t), g1["transpose_3d_021"])
    eq(O.concat_axis0(t, t[:2].copy()), g1["concat_axis0"])


def test_bf16_round_trip():
    v = g1["bf16_in"]
    eq(O.f32_to_bf16_bits(v), g1["bf16_bits"])
    eq(O.bf16_bits_to_f32(g1["bf16_bits"]), g1["bf16_back"])


def test_sampler():
    lg = g1["sample_logits"]
    eq([O.sample_token(r, 0.0, 0, 1.0) for r in lg], g1["sample_t0"])
    eq([O.sample_token(r, 0.0, 50, 0.9) for r in lg], g1["sample_t0_k50_p09"])
    assert g1["sample_t0"][3] == 100  # exact tie -> lowest index
    eq([O.argmax_lowest_index(r) for r in lg], g1["sample_t0"])


def test_fp8_table_and_block_dequant():
    g2 = load_golden("g2_fp8.npz")
    deq = O.dequantize_fp8_e4m3_block(g2["codes"], g2["scale_bits"])
    eq(deq, g2["deq"])
    y = O.matmul(g2["x"], O.transpose(deq.astype(np.float32)))
    eq(y, g2["y"])
    t = O.fp8_e4m3_table()
    assert np.isnan(t[0x7F]) and np.isnan(t[0xFF]) and t[0x7E] == 448.0 and t[0x01] == 2.0**-9


def test_fp8_quantiser_round_trip():
    """The synthetic quantiser is the inverse of the reference dequantiser on exactly
    representable inputs and never emits the NaN codes."""
    rng = np.random.default_rng(0)
    codes = np.array([c for c in range(256) if c not in (0x7F, 0xFF, 0x80)], np.uint8)
    w = rng.choice(codes, size=(128, 256)).astype(np.uint8)
    w[0, 0] = 0x7E  # make absmax hit 448 so scale is exactly s
    w[0, 128] = 0x7E
    s = O.f32_to_bf16_bits(np.full((1, 2), 0.0078125, np.float32))
    deq = O.dequantize_fp8_e4m3_block(w, s)
    c2, s2 = O.quantize_fp8_e4m3_block(deq)
    eq(s2, s)
    eq(O.dequantize_fp8_e4m3_block(c2, s2), deq)
    assert not np.any((c2 == 0x7F) | (c2 == 0xFF))
    x = rng.standard_normal((256, 256)).astype(np.float32) * 0.02
    c3, s3 = O.quantize_fp8_e4m3_block(x)
    assert not np.any((c3 == 0x7F) | (c3 == 0xFF))
    err = np.abs(O.dequantize_fp8_e4m3_block(c3, s3) - x).max() / np.abs(x).max()
    assert err < 0.04  # e4m3 has 3 mantissa bits: half-ulp relative error 2^-4 at most


def test_tiny_qwen3_model_bit_exact():
    g3 = load_golden("g3_tiny_qwen3.npz")
    w = O.make_qwen3_weights(TINY, seed=int(g3["seed"]), bf16=True)
    assert cfg_checksum(w) == float(g3["wsum"])
    model = O.build_qwen3_ref(TINY, w, max_pos=128)
    prompt = [int(t) for t in g3["prompt"]]
    hidden, _ = model(prompt, use_cache=True)
    eq(hidden, g3["prefill_hidden"])
    eq(model.get_logits(hidden), g3["prefill_logits"])
    tokens, step_logits = model.generate(prompt, max_new_tokens=10, temperature=0.0, top_k=0, top_p=1.0,
                                         return_logits=True)
    eq(tokens, g3["tokens"])
    eq(np.stack(step_logits), g3["step_logits"])


def test_full_width_qwen3_layer_bit_exact():
    g5 = load_golden("g5_qwen3_layer.npz")
    cfg = dict(O.QWEN3_0_6B, num_layers=1, vocab_size=64)
    w = O.make_qwen3_weights(cfg, seed=int(g5["seed"]), bf16=True)
    assert cfg_checksum(w) == float(g5["wsum"])
    block = O.build_qwen3_ref(cfg, w, max_pos=64).blocks[0]
    y, kv = block(g5["x"], [0, 1, 2, 3, 4, 5], None, True)
    eq(y, g5["y"])
    y1, kv1 = block(g5["x1"], [6], kv, True)
    eq(y1, g5["y1"])
    eq(kv1[0], g5["k"])
    eq(kv1[1], g5["v"])


def test_decode_step_equals_prefill_row():
    """Equivalence used to pin the kernel-defined fixed-cache path: decoding token t with a
    KV cache equals row t of a length-(t+1) prefill (up to fp32 summation order)."""
    w = O.make_qwen3_weights(TINY, seed=3, bf16=True)
    model = O.build_qwen3_ref(TINY, w, max_pos=64)
    ids = [5, 9, 300, 77, 1000, 12]
    full, _ = model(ids, use_cache=False)
    h, past = model(ids[:-1], use_cache=True)
    last, _ = model(ids[-1:], past_key_values=past, use_cache=True)
    np.testing.assert_allclose(last[0], full[-1], rtol=2e-4, atol=2e-5)


def test_gpt2_small_config1_tokens():
    """BASELINE config 1: GPT-2-small random-init, 16-token greedy decode on the CPU path."""
    g4 = load_golden("g4_gpt2_small.npz")
    w = O.make_gpt2_weights(O.GPT2_SMALL, seed=int(g4["seed"]))
    assert cfg_checksum(w) == float(g4["wsum"])
    model = O.build_gpt2_ref(O.GPT2_SMALL, w)
    ids = model.generate([int(t) for t in g4["prompt"]], max_new_tokens=16, temperature=0.0, top_k=0, top_p=1.0)
    eq(ids, g4["tokens"])


# ---- fp8 x fp8 blockwise GEMM restatement (build-defined contract, include/pgk_hip.h) ----------------------
def test_fp8_row_quantiser_round_trip_and_gemm_identity():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((5, 256)) * 3).astype(np.float32)
    x[2, :128] = 0
    codes, scale = O.quantize_fp8_rows(x)
    assert codes.dtype == np.uint8 and scale.shape == (5, 2) and scale[2, 0] == 1.0
    deq = O.fp8_e4m3_table()[codes].reshape(5, 2, 128) * scale[:, :, None]
    # e4m3 keeps 3 mantissa bits: relative step 2^-3, half of it after rounding; absmax maps exactly to 448
    assert np.max(np.abs(deq.reshape(5, 256) - x)) <= np.abs(x).max() * 2.0**-4
    assert np.all(np.abs(O.fp8_e4m3_table()[codes]).max(axis=1) == 448.0)
    # GEMM against an identity-like weight reproduces the dequantised activations
    w = np.zeros((128, 256), np.float32)
    w[np.arange(128), np.arange(128)] = 1.0
    wc, ws = O.quantize_fp8_e4m3_block(w)
    out = O.gemm_fp8_blockwise(codes, scale, wc, ws)
    np.testing.assert_allclose(out, deq.reshape(5, 256)[:, :128] * (O.fp8_e4m3_table()[wc][0, 0] * O.bf16_bits_to_f32(ws)[0, 0]), rtol=1e-6)


def test_sample_token_u_restates_host_sampler_sets():
    """The kept sets of sample_token_u are those of the reference-restating host sampler (sample_token): every token it
    can return under top-k / top-p has non-zero probability there, u = 0 returns the lowest kept index, and the draw
    frequencies follow the restricted softmax."""
    rng = np.random.default_rng(5)
    lg = (rng.standard_normal(300) * 2).astype(np.float32)
    for k, p in ((0, 1.0), (10, 1.0), (0, 0.8), (25, 0.9)):
        z = lg / np.float32(0.9)
        probs = np.exp(z - z.max())
        probs /= probs.sum()
        if 0 < k < len(probs):
            keep = np.argsort(probs)[-k:]
            m = np.zeros_like(probs, bool); m[keep] = True
            probs = np.where(m, probs, 0.0); probs /= probs.sum()
        if p < 1.0:
            si = np.argsort(probs)[::-1]
            cut = min(np.searchsorted(np.cumsum(probs[si]), p) + 1, len(probs))
            m = np.zeros_like(probs, bool); m[si[:cut]] = True
            probs = np.where(m, probs, 0.0); probs /= probs.sum()
        toks = [O.sample_token_u(lg, 0.9, k, p, float(u)) for u in rng.random(3000)]
        assert all(probs[t] > 0 for t in toks)
        assert O.sample_token_u(lg, 0.9, k, p, 0.0) == int(np.nonzero(probs)[0][0])
        freq = np.bincount(toks, minlength=300) / 3000
        assert np.abs(freq - probs).max() < 0.04


def test_paged_attention_oracle_equals_dense_fixed_cache():
    """The paged restatement gathers the same rows the golden-pinned dense fixed-cache attention reads."""
    rng = np.random.default_rng(8)
    hq, hkv, d, bs, ctx = 4, 2, 32, 8, 21
    k, v = rng.standard_normal((ctx, hkv, d)).astype(np.float32), rng.standard_normal((ctx, hkv, d)).astype(np.float32)
    kc, vc = np.zeros((6, hkv, bs, d), np.float32), np.zeros((6, hkv, bs, d), np.float32)
    pages = [4, 1, 3]
    O.paged_cache_write(k, v, kc, vc, [pages[t // bs] * bs + t % bs for t in range(ctx)])
    q = rng.standard_normal((1, hq, d)).astype(np.float32)
    got = O.paged_attention_v1(q, kc, vc, np.array([pages], np.int32), [ctx])
    dense = O.sdpa_causal_fixed_cache(q.transpose(1, 0, 2), np.repeat(k.transpose(1, 0, 2), 2, axis=0),
                                      np.repeat(v.transpose(1, 0, 2), 2, axis=0), ctx)
    np.testing.assert_allclose(got[0], dense[:, 0], rtol=2e-5, atol=2e-6)


def test_basic_ops_oracle_vs_golden():
    """G6: unary math, reductions, softmax, sum_axis, clamp, where - the oracle's restatements against the
    reference's NumPy path (bit-equal: the same NumPy expressions on the same inputs)."""
    g6 = load_golden("g6_basic_ops.npz")
    x, pos = g6["x"], g6["pos"]
    for name in ("exp", "relu", "sin", "cos", "abs", "neg", "sigmoid", "tanh", "relu2"):
        np.testing.assert_allclose(O.unary(name, x), g6[name], rtol=1e-6, atol=1e-7, err_msg=name)
    for name in ("log", "sqrt", "rsqrt"):
        np.testing.assert_allclose(O.unary(name, pos), g6[name], rtol=1e-6, atol=1e-7, err_msg=name)
    for name in ("sum", "mean", "max", "min", "argmax"):
        got = O.reduce_all(name, x)
        assert got.shape == (1,) and got.dtype == g6["red_" + name].dtype
        np.testing.assert_allclose(got, g6["red_" + name], rtol=1e-6)
    np.testing.assert_allclose(O.softmax_last(x), g6["softmax"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(O.softmax_last(g6["x3"]), g6["softmax3"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(O.sum_axis(x, 0), g6["sum_axis0"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.sum_axis(x, 1), g6["sum_axis1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(O.clamp(x, -0.5, 1.25), g6["clamp"])
    np.testing.assert_array_equal(O.where(g6["cond"], x, g6["y"]), g6["where"])
