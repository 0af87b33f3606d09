"""GPU parity tests at model level: CausalTransformerModel (op surface) and the native engine
(fused decode kernels + whole-step hipGraph) vs the CPU oracle / golden vectors.
Bars (BASELINE.json north_star): greedy tokens bit-identical; logits rel L2 err <= 1e-2 (bf16), <= 5e-2 (fp8)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cpu_ref as O
from tests.conftest import load_golden, rel_err
from tests.golden_cfg import TINY

pytestmark = pytest.mark.gpu

pk = pytest.importorskip("pygpukit_amd")
from pygpukit_amd import llm  # noqa: E402
from pygpukit_amd.core import bfloat16, from_numpy  # noqa: E402
from pygpukit_amd.llm import synthetic as S  # noqa: E402

g3 = load_golden("g3_tiny_qwen3.npz")
PROMPT = [int(t) for t in g3["prompt"]]


def f32(a):
    h = a.to_numpy()
    return O.bf16_bits_to_f32(h) if a.dtype == bfloat16 else h.astype(np.float32)


@pytest.fixture(scope="module")
def tiny_weights():
    return O.make_qwen3_weights(TINY, seed=int(g3["seed"]), bf16=True)


def margin(logits: np.ndarray) -> float:
    s = np.sort(logits)
    return float(s[-1] - s[-2])


def test_fixture_margins_are_safe():
    """Token-level parity is only meaningful when the oracle's top-1/top-2 gap exceeds bf16 noise."""
    for row in g3["step_logits"]:
        assert margin(row) > 0.02 * np.abs(row).max()


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-4), ("bfloat16", 1e-2)])
def test_model_forward_and_generate_vs_golden(tiny_weights, dtype, tol):
    model = S.build_model_from_weights(TINY, tiny_weights, dtype=dtype, max_pos=128)
    hidden, kv = model(PROMPT, use_cache=True)
    assert hidden.shape == (len(PROMPT), TINY["hidden_size"]) and len(kv) == TINY["num_layers"]
    assert kv[0][0].shape == (len(PROMPT), TINY["num_kv_heads"], TINY["head_dim"])
    assert rel_err(f32(hidden), g3["prefill_hidden"]) < tol
    assert rel_err(f32(model.get_logits(hidden)), g3["prefill_logits"]) < tol
    ids = model.generate(PROMPT, max_new_tokens=10, temperature=0.0, top_k=0, top_p=1.0)
    np.testing.assert_array_equal(ids, g3["tokens"])
    ids_gpu = model.generate(PROMPT, max_new_tokens=10, temperature=0.0, top_k=0, top_p=1.0, gpu_sampling=True)
    np.testing.assert_array_equal(ids_gpu, g3["tokens"])
    assert list(model.generate_stream(PROMPT, max_new_tokens=4, temperature=0.0, top_k=0, top_p=1.0)) == list(g3["tokens"][12:16])
    assert llm.QwenModel is llm.CausalTransformerModel and model.forward(PROMPT)[0].shape == hidden.shape


def test_full_width_layer_vs_golden():
    g5 = load_golden("g5_qwen3_layer.npz")
    cfg = dict(O.QWEN3_0_6B, num_layers=1, vocab_size=64)
    w = O.make_qwen3_weights(cfg, seed=int(g5["seed"]), bf16=True)
    for dtype, tol in (("float32", 2e-4), ("bfloat16", 1e-2)):
        block = S.build_model_from_weights(cfg, w, dtype=dtype, max_pos=64).blocks[0]
        x = from_numpy(g5["x"]) if dtype == "float32" else from_numpy(O.f32_to_bf16_bits(g5["x"]))
        y, kv = block(x, [0, 1, 2, 3, 4, 5], None, True)
        assert rel_err(f32(y), g5["y"]) < tol
        x1 = from_numpy(g5["x1"]) if dtype == "float32" else from_numpy(O.f32_to_bf16_bits(g5["x1"]))
        y1, kv1 = block(x1, [6], kv, True)
        assert rel_err(f32(y1), g5["y1"]) < tol
        assert rel_err(f32(kv1[0]), g5["k"]) < tol and rel_err(f32(kv1[1]), g5["v"]) < tol


def test_engine_greedy_tokens_and_logits_vs_golden(tiny_weights):
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=128, max_batch=1)
    toks = eng.generate_greedy(PROMPT, max_new_tokens=10)
    np.testing.assert_array_equal(toks, g3["tokens"])
    assert rel_err(eng.last_prefill_logits, g3["step_logits"][0]) < 1e-2
    # eager (un-captured) launches give the same tokens as graph replay
    np.testing.assert_array_equal(eng.generate_greedy(PROMPT, max_new_tokens=10, use_graph=False), g3["tokens"])
    # per-step logits: step t decodes token g3.tokens[12+t-1] at position 12+t-1
    eng.prefill(PROMPT)
    for t in range(1, 10):
        tok, pos = int(g3["tokens"][len(PROMPT) + t - 1]), len(PROMPT) + t - 1
        eng.set_state([tok], [pos])
        eng.replay(1)
        lg = eng.logits(1).to_numpy()[0]
        assert rel_err(lg, g3["step_logits"][t]) < 1e-2
        assert int(np.argmax(lg)) == int(g3["tokens"][len(PROMPT) + t])
    assert eng.launches_per_step() in (4 * TINY["num_layers"] + 2, 6 * TINY["num_layers"] + 2)


def test_engine_kv_cache_matches_oracle(tiny_weights):
    ref = O.build_qwen3_ref(TINY, tiny_weights, max_pos=128)
    _, kv = ref(PROMPT, use_cache=True)
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=128, max_batch=1)
    eng.prefill(PROMPT)
    for layer in range(TINY["num_layers"]):
        k, v = eng.kv_cache(layer)
        kh = O.bf16_bits_to_f32(k.to_numpy())[0, :, : len(PROMPT)]  # [Hkv, S, D]
        vh = O.bf16_bits_to_f32(v.to_numpy())[0, :, : len(PROMPT)]
        assert rel_err(kh, kv[layer][0].transpose(1, 0, 2)) < 1e-2
        assert rel_err(vh, kv[layer][1].transpose(1, 0, 2)) < 1e-2


def test_engine_decode_equals_prefill_row(tiny_weights):
    """Size-independent property: decoding token t against the cache == row t of a (t+1)-long prefill."""
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=128, max_batch=1)
    ids = [5, 9, 300, 77, 1000, 12, 800, 3]
    full = eng.prefill(ids)
    eng.prefill(ids[:-1])
    eng.set_state([ids[-1]], [len(ids) - 1])
    eng.decode_step(1)
    assert rel_err(eng.logits(1).to_numpy()[0], full) < 5e-3


def test_engine_batch_of_independent_sequences(tiny_weights):
    """BASELINE config 4 semantics at small scale: B sequences, own prompts/lengths, one fused step each."""
    ref = O.build_qwen3_ref(TINY, tiny_weights, max_pos=128)
    rng = np.random.default_rng(21)
    B = 5
    prompts, want = [], []
    for n in (3, 12, 7, 1, 20):  # draw prompts until the oracle's top-1/top-2 margins are safely above bf16 noise
        for _ in range(200):
            p = [int(t) for t in rng.integers(0, TINY["vocab_size"], n)]
            toks, lgs = ref.generate(p, max_new_tokens=6, temperature=0.0, top_k=0, top_p=1.0, return_logits=True)
            if min(margin(r) / np.abs(r).max() for r in lgs) > 0.03:
                break
        else:
            raise AssertionError("no safe-margin prompt found")
        prompts.append(p)
        want.append(toks)
    model = S.build_model_from_weights(TINY, tiny_weights, dtype="bfloat16", max_pos=128)
    strat = llm.DecodeBatch(batch_size=B)
    strat.bind(model)
    strat.init_graph(max_seq_len=128)
    first = strat.prefill(prompts)
    toks = strat.run_greedy(first, [len(p) for p in prompts], 5)
    for b in range(B):
        got = prompts[b] + [int(first[b])] + [int(t) for t in toks[:, b]]
        assert got == want[b], f"sequence {b}"


@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
@pytest.mark.parametrize("B", [3, 8, 11, 19])
def test_engine_batched_mfma_path_matches_gemv_path(tiny_weights, B, fmt, monkeypatch):
    """Chunks of 3..16 sequences run the MFMA projections (engine_batched.cuh: 8-row and 16-row LDS images, a 16+3
    split at B = 19); with PGK_BATCHED_MFMA=0 the same batch runs the GEMV kernels (fp32 activations in chunks of
    1-2 sequences, bf16 in chunks of 4 / 8).  The difference is bf16 rounding of the activations and summation order:
    logits agree to 6e-3 (the bf16 bar is 1e-2) and - for rows with a safe top-1 margin - the tokens are identical."""
    rng = np.random.default_rng(60 + B)
    prompts = [[int(t) for t in rng.integers(0, TINY["vocab_size"], int(rng.integers(1, 24)))] for _ in range(B)]
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PGK_BATCHED_MFMA", "2" if mode == "1" else "0")     # 2: MFMA chunks from 3 sequences up
        eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=64, max_batch=B, weight_format=fmt)
        first = [int(np.argmax(eng.prefill(p, seq=b))) for b, p in enumerate(prompts)]
        eng.set_state(first, [len(p) for p in prompts])
        for _ in range(3):
            eng.decode_step(B)
        eng.synchronize()
        outs[mode] = (eng.logits(B).to_numpy().copy(), eng.read_tokens(B, 3).copy())
    lg1, tk1 = outs["1"]
    lg0, tk0 = outs["0"]
    assert rel_err(lg1, lg0) < 6e-3, rel_err(lg1, lg0)
    safe = np.array([margin(r) / np.abs(r).max() > 0.02 for r in lg0])
    assert safe.sum() >= B // 2
    np.testing.assert_array_equal(tk1[-1][safe], tk0[-1][safe])
    np.testing.assert_array_equal(np.argmax(lg1, axis=1)[safe], np.argmax(lg0, axis=1)[safe])


def test_decode_strategies_api(tiny_weights):
    model = S.build_model_from_weights(TINY, tiny_weights, dtype="bfloat16", max_pos=128)
    model.init_fixed_cache(128, "bfloat16")
    assert model.blocks[0].attn._k_cache.shape == (TINY["num_kv_heads"], 128, TINY["head_dim"])
    logits = model.prefill_fixed_cache(PROMPT)
    assert rel_err(f32(logits), g3["prefill_logits"]) < 1e-2
    first = int(np.argmax(f32(logits)[-1]))
    assert first == int(g3["tokens"][len(PROMPT)])
    # eager strategy
    m1 = llm.DecodeM1()
    m1.bind(model)
    bufs = llm.DecodeBuffers.allocate(model.config, dtype="bfloat16", use_qk_norm=True, vocab_size=TINY["vocab_size"])
    lg = f32(m1.step(first, len(PROMPT), len(PROMPT) + 1, bufs))[0]
    assert rel_err(lg, g3["step_logits"][1]) < 1e-2
    # graph strategy, fed from the model's fixed caches
    mg = llm.DecodeM1Graph()
    mg.bind(model)
    with pytest.raises(NotImplementedError):
        mg.step(first, 0, 1, bufs)
    mg.init_graph(max_seq_len=128)
    assert mg.has_graph()
    mg.load_kv_from_model(len(PROMPT))
    lg2 = mg.step_graph(first, len(PROMPT), len(PROMPT) + 1).to_numpy()[0]
    assert rel_err(lg2, g3["step_logits"][1]) < 1e-2
    with pytest.raises(ValueError):
        mg.step_graph(first, 5, 9)
    rest = mg.run_greedy(first, len(PROMPT), 9)
    assert [first] + rest[:-1] == [int(t) for t in g3["tokens"][len(PROMPT):len(PROMPT) + 9]]
    # verify-style batch of consecutive tokens of one sequence
    db = llm.DecodeBatch(batch_size=4)
    db.bind(model)
    model.prefill_fixed_cache(PROMPT)
    nxt = [int(t) for t in g3["tokens"][len(PROMPT):len(PROMPT) + 3]]
    lgb = f32(db.step_batch(nxt, len(PROMPT), len(PROMPT) + 3))
    for i in range(3):
        assert rel_err(lgb[i], g3["step_logits"][i + 1]) < 1e-2


def test_engine_fp8_weights(tiny_weights):
    """w8a16 path (config 3 semantics): fp8-e4m3 linears + bf16 block scales, bf16 embedding / lm_head."""
    wq = {"embed": tiny_weights["embed"], "final_norm": tiny_weights["final_norm"], "layers": []}
    for lw in tiny_weights["layers"]:
        d = dict(lw)
        # quantise the FUSED matrices exactly as the engine builder does, then split back for the oracle
        for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
            fused = np.concatenate([lw[n] for n in names], axis=0)
            codes, sbits = O.quantize_fp8_e4m3_block(fused)
            deq = O.dequantize_fp8_e4m3_block(codes, sbits)
            r = 0
            for n in names:
                d[n] = deq[r:r + lw[n].shape[0]]
                r += lw[n].shape[0]
        wq["layers"].append(d)
    ref = O.build_qwen3_ref(TINY, wq, max_pos=128)
    want, want_logits = ref.generate(PROMPT, max_new_tokens=6, temperature=0.0, top_k=0, top_p=1.0, return_logits=True)
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=128, max_batch=1, weight_format="fp8")
    got = eng.generate_greedy(PROMPT, max_new_tokens=6)
    assert rel_err(eng.last_prefill_logits, want_logits[0]) < 1e-2  # vs the dequantised-weight oracle
    # Against the UNQUANTISED bf16 model the gap is the e4m3 weight-quantisation error itself (3 mantissa
    # bits; the reference quotes ~12 % for its W8A16 GEMV, README.md:455-457): the GPU must add nothing to
    # what the oracle shows for the same codes.
    q_err_oracle = rel_err(want_logits[0], g3["step_logits"][0])
    q_err_gpu = rel_err(eng.last_prefill_logits, g3["step_logits"][0])
    assert q_err_gpu < 1.1 * q_err_oracle + 1e-3, (q_err_gpu, q_err_oracle)
    assert got == want


LLAMA_MINI = dict(vocab_size=2048, hidden_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=128,
                  intermediate_size=1536, norm_eps=1e-5, rope_theta=5e5)


def test_fp8_activation_prefill_llama_shape_vs_oracle():
    """BASELINE config 5 in miniature (Llama-style: no QK-norm, theta 5e5): prefill of S > 128 tokens with
    weight_format "fp8a8" runs every projection on the fp8 x fp8 MFMA GEMM with activations quantised per
    (row, 128 k).  An e4m3 quantiser is chaotic under the bf16-level differences that exist upstream of it (a
    0.3 % input change flips ~3 % of the codes and moves the GEMM output by ~60 % of the quantisation error
    itself, tools/fp8a8_err.py), so two correct implementations cannot agree tightly at model level; exact
    agreement is pinned at op level (test_gpu_ops.py: same codes in -> 3e-3, bit-exact on integers).  Here the
    bar is statistical: measured against the oracle that runs the SAME fp8 weights with unquantised activations
    (the w8a16 semantics of the reference's LinearFP8), the GPU may not be further away than the oracle's own
    fp8-activation model is (x1.25), and the two fp8-activation results may not differ by more than that either.
    On this 2-layer random-init model the oracle's own fp8-activation cost is ~5.2e-2, i.e. e4m3 activations alone
    use up BASELINE's 5e-2 fp8 budget - recorded in DESIGN.md, not hidden by a loose constant here."""
    cfg = LLAMA_MINI
    w = O.make_qwen3_weights(cfg, seed=11, bf16=True)
    for lw in w["layers"]:
        del lw["q_norm"], lw["k_norm"]
    S_ = 200
    prompt = [int(t) for t in np.random.default_rng(12).integers(0, cfg["vocab_size"], S_)]
    eng = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1, weight_format="fp8a8")
    got = eng.prefill(prompt)
    ref = O.build_qwen3_ref_fp8a8(cfg, w, max_pos=256)
    hidden, _ = ref(prompt)
    want = ref.get_logits(hidden)[-1]
    # same fp8 weights, bf16 activations (the w8a16 semantics the reference's LinearFP8 has)
    wq = {"embed": w["embed"], "final_norm": w["final_norm"], "layers": []}
    for lw in w["layers"]:
        d = dict(lw)
        for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
            fused = np.concatenate([lw[n] for n in names], axis=0)
            deq = O.dequantize_fp8_e4m3_block(*O.quantize_fp8_e4m3_block(fused))
            r = 0
            for n in names:
                d[n] = deq[r:r + lw[n].shape[0]]
                r += lw[n].shape[0]
        wq["layers"].append(d)
    ref16 = O.build_qwen3_ref(cfg, wq, max_pos=256)
    h16, _ = ref16(prompt)
    want16 = ref16.get_logits(h16)[-1]
    q_oracle = rel_err(want, want16)          # what fp8 activations cost in the oracle itself
    assert q_oracle < 8e-2, q_oracle
    assert rel_err(got, want16) < 1.25 * q_oracle, (rel_err(got, want16), q_oracle)
    assert rel_err(got, want) < 1.25 * q_oracle, (rel_err(got, want), q_oracle)
    assert int(np.argmax(got)) == int(np.argmax(want16)) or np.sort(want16)[-1] - np.sort(want16)[-2] < 0.1
    # decode after an fp8a8 prefill continues on the w8a16 GEMV path from the cache the prefill wrote
    eng16 = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1, weight_format="fp8")
    l16 = eng16.prefill(prompt)
    assert rel_err(l16, want16) < 1e-2


def test_qwen3_0_6b_full_size_engine_vs_oracle():
    """BASELINE config 2 at full size: random-init Qwen3-0.6B, prefill 128 + 8 greedy tokens."""
    cfg = O.QWEN3_0_6B
    w = O.make_qwen3_weights(cfg, seed=0, bf16=True)
    prompt = [int(t) for t in np.random.default_rng(0).integers(0, cfg["vocab_size"], 128)]
    ref = O.build_qwen3_ref(cfg, w, max_pos=256)
    want, want_logits = ref.generate(prompt, max_new_tokens=8, temperature=0.0, top_k=0, top_p=1.0, return_logits=True)
    eng = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1)
    got = eng.generate_greedy(prompt, max_new_tokens=8)
    err = rel_err(eng.last_prefill_logits, want_logits[0])
    assert err < 1e-2, err
    assert got == want
    # last decode step's logits against the oracle's
    assert rel_err(eng.logits(1).to_numpy()[0], want_logits[-1]) < 1e-2


@pytest.mark.parametrize("fmt", ["bf16", "fp8"])
def test_config3_long_context_decode_is_consistent_with_prefill(fmt):
    """BASELINE config 3 at full size (Qwen3-0.6B shape, context 2048, bf16 and w8a16): the CPU oracle cannot run a
    2048-token prompt inside a test, so parity is carried by a size-independent property - the logits of position
    2048 computed by the DECODE path (split-KV flash-decoding over the 2048 cached rows, GEMV projections) must equal
    those computed by the PREFILL path (MFMA GEMMs, flash-attention) for the same 2049 tokens, to the bf16 bar; and the
    split-KV decode must not depend on how the cache was filled (one prefill vs prefill + 3 decode steps)."""
    cfg = O.QWEN3_0_6B
    w = S.make_qwen3_weights(cfg, seed=5)
    toks = [int(t) for t in np.random.default_rng(7).integers(0, cfg["vocab_size"], 2049)]
    eng = S.build_engine_from_weights(cfg, w, max_seq_len=2112, max_batch=1, weight_format=fmt)
    full = eng.prefill(toks).copy()                          # logits of the last position via the prefill path
    eng.prefill(toks[:2048])                                 # refill rows 0..2047, then feed token 2048 through decode
    eng.set_state([toks[2048]], [2048])
    eng.decode_step(1)
    eng.synchronize()
    dec = eng.logits(1).to_numpy()[0].copy()
    assert rel_err(dec, full) < 1e-2, rel_err(dec, full)
    assert int(np.argmax(dec)) == int(np.argmax(full)) or margin(full) / np.abs(full).max() < 0.02
    # same position reached by prefill(2046) + 3 decode steps over given tokens
    eng.prefill(toks[:2046])
    for i in (2046, 2047, 2048):
        eng.set_state([toks[i]], [i])
        eng.decode_step(1)
    eng.synchronize()
    dec2 = eng.logits(1).to_numpy()[0]
    assert rel_err(dec2, dec) < 1e-2, rel_err(dec2, dec)


@pytest.mark.parametrize("fmt", ["bf16", "fp8a8"])
def test_config5_width_prefill_is_consistent_across_chunkings(fmt):
    """BASELINE config 5 widths (Llama-3-8B: H 4096, I 14336, 32/8 heads) at S = 4096 with 2 layers.  The oracle cannot
    run this inside a test; the property used instead: a 4096-token prefill in ONE pass (M = 4096: the 256-tile LDS-DMA
    GEMMs, flash attention with kv_len == q_len) must give the same last-row logits as the same tokens in TWO passes of
    2048 (other GEMM tile counts, flash attention against 2048 cached rows with a kv offset).  Row-wise fp8 activation
    quantisation is independent of M, so both formats agree to accumulation order (bar 1e-2)."""
    from pygpukit_amd.llm.engine import Engine

    cfg = dict(S.LLAMA3_8B, num_layers=2, vocab_size=8192)
    w = S.random_engine_weights(cfg, seed=9, fp8=(fmt != "bf16"), keep_bf16=(fmt == "bf16"), threads=8)
    layers = w["bf16"] if fmt == "bf16" else w["fp8"]
    toks = [int(t) for t in np.random.default_rng(10).integers(0, cfg["vocab_size"], 4096)]
    eng = Engine(cfg, w["embed"], layers, w["final_norm"], None, max_seq_len=4096, max_batch=1, weight_format=fmt, use_qk_norm=False)
    one = eng.prefill(toks).copy()
    eng.prefill(toks[:2048], want_last_logits=False)
    two = eng.prefill(toks[2048:], start_pos=2048).copy()
    assert np.isfinite(one).all() and rel_err(two, one) < 1e-2, rel_err(two, one)


def test_engine_in_graph_sampling_matches_oracle_sampler(tiny_weights):
    """Stochastic decode inside the captured whole-step graph: every step draws from the step's logits with the queued
    uniform number.  Replayed against the oracle: feed the oracle model the engine's own token history, apply
    sample_token_u with the same u, and require the same token wherever the draw is not within 3e-3 (of the kept mass) of a boundary
    (the engine's bf16 logits differ from the oracle's fp32 ones by ~1e-3 relative)."""
    B, T, k, p, steps = 2, 0.8, 40, 0.95, 6
    prompts = [[5, 100, 7, 900], [33, 2]]
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=64, max_batch=B)
    u = eng.set_sampling(T, top_k=k, top_p=p, n_steps=steps, seed=123)
    first = [int(np.argmax(eng.prefill(pr, seq=b))) for b, pr in enumerate(prompts)]
    eng.set_state(first, [len(pr) for pr in prompts])
    eng.capture(B)
    eng.replay(steps)
    eng.synchronize()
    toks = eng.read_tokens(B, steps)                          # [steps, B]
    ref = O.build_qwen3_ref(TINY, tiny_weights, max_pos=64)
    checked = 0
    for b in range(B):
        hist = prompts[b] + [first[b]]
        for s_ in range(steps):
            hid, _ = ref(hist)
            lg = ref.get_logits(hid)[-1]
            want, margin_ = O.sample_token_u(lg, T, k, p, float(u[s_, b]), return_margin=True)
            if margin_ > 3e-3:
                assert int(toks[s_, b]) == want, (b, s_)
                checked += 1
            hist.append(int(toks[s_, b]))                     # follow the engine's own history
    assert checked >= steps                                    # most draws are away from a boundary
    # greedy again after switching sampling off (re-capture: the sampling node is part of the graph)
    eng.set_sampling(0.0)
    eng.set_state(first, [len(pr) for pr in prompts])
    eng.capture(B)
    eng.replay(2)
    eng.synchronize()
    g = eng.read_tokens(B, steps + 2)[-2:]
    for b in range(B):
        hid, _ = ref(prompts[b] + [first[b]])
        assert int(g[0, b]) == int(np.argmax(ref.get_logits(hid)[-1])) or margin(ref.get_logits(hid)[-1]) < 0.02


def test_engine_generate_with_sampling_is_reproducible_and_diverse(tiny_weights):
    eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=64, max_batch=1)
    a = eng.generate([1, 2, 3], 12, temperature=1.0, top_k=50, top_p=0.9, seed=4)
    b = eng.generate([1, 2, 3], 12, temperature=1.0, top_k=50, top_p=0.9, seed=4)
    c = eng.generate([1, 2, 3], 12, temperature=1.0, top_k=50, top_p=0.9, seed=5)
    g = eng.generate([1, 2, 3], 12, temperature=0.0)
    assert a == b and len(a) == 15 and a[:3] == [1, 2, 3]
    assert a != c                                              # another seed, another continuation
    assert g == eng.generate_greedy([1, 2, 3], 12)             # greedy restored after sampling was switched off


@pytest.mark.parametrize("B", [1, 3])
def test_split_kv_in_launch_merge_is_bit_identical_to_merge_kernel(tiny_weights, B, monkeypatch):
    """Long-context decode merges the split-KV slices inside the attention launch (the workgroup that draws the last
    ticket of a kv head combines them, after an agent-scope release / acquire hand-off).  The arithmetic is the merge
    kernel's, in the same order, so the two paths must agree BIT FOR BIT: 33 slices x 2 layers x 48 steps of hand-offs
    per sequence, any stale read shows up as a differing logit or token."""
    rng = np.random.default_rng(70 + B)
    prompts = [[int(t) for t in rng.integers(0, TINY["vocab_size"], 1500 + 37 * b)] for b in range(B)]
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PGK_ATTN_INKERNEL_MERGE", mode)
        eng = S.build_engine_from_weights(TINY, tiny_weights, max_seq_len=2112, max_batch=B)
        first = [int(np.argmax(eng.prefill(p, seq=b))) for b, p in enumerate(prompts)]
        eng.set_state(first, [len(p) for p in prompts])
        eng.capture(B)
        eng.replay(48)
        eng.synchronize()
        outs[mode] = (eng.read_tokens(B, 48).copy(), eng.logits(B).to_numpy().copy(), eng.launches_per_step())
    np.testing.assert_array_equal(outs["1"][0], outs["0"][0])
    np.testing.assert_array_equal(outs["1"][1], outs["0"][1])
    assert outs["1"][2] == outs["0"][2] - TINY["num_layers"] * (1 if B == 1 else 2)      # one launch fewer per layer and chunk
