"""CPU-only tests (no GPU): the C-ABI library loads and exports every symbol include/pgk_hip.h declares,
host-side logic (dtypes, config, sampler, RoPE tables, fp8 quantiser, sharding) matches the reference's
golden vectors / the oracle, and the product path fails loudly without a device."""

from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import pytest

from oracle import cpu_ref as O
from tests.conftest import ROOT, load_golden

g1 = load_golden("g1_ops.npz")


def header_symbols() -> list[str]:
    text = open(os.path.join(ROOT, "include", "pgk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pygpukit_amd import _hip

    lib = _hip.load()  # raises if libpgk_hip.so was not built
    names = header_symbols()
    assert len(names) > 70
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes prototypes cover the header exactly
    assert sorted(_hip.EXPORTED_SYMBOLS) == names
    assert b"gfx950" in lib.pgk_version()
    raw = ctypes.CDLL(_hip.LIB_PATH)
    assert raw.pgk_last_error is not None


def test_no_device_fails_loudly_not_silently():
    import pygpukit_amd as pk
    from pygpukit_amd import _hip

    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        pk.zeros((4,), "float32")
    with pytest.raises(RuntimeError):
        pk.from_numpy(np.zeros(3, np.float32))
    assert pk.get_backend().is_available() is False


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pygpukit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(dirpath, f)


def test_dtypes_surface():
    from pygpukit_amd.core import dtypes as D

    assert D.DataType.from_numpy_dtype(np.uint16) is D.bfloat16  # bf16 travels as uint16
    assert D.DataType.from_string("float16") is D.float16 and D.bfloat16.itemsize == 2
    assert D.bfloat16.to_numpy_dtype() == np.uint16 and D.int4.to_numpy_dtype() == np.uint8
    assert [d.code for d in (D.float64, D.float32, D.float16, D.bfloat16, D.int64, D.int32, D.int16, D.int8, D.uint8, D.int4)] == list(range(10))
    with pytest.raises(ValueError):
        D.DataType.from_string("complex64")
    with pytest.raises(ValueError):
        D.DataType.from_numpy_dtype(np.complex64)


def test_config_and_specs():
    from pygpukit_amd.llm import GPT2_SPEC, LLAMA_SPEC, MODEL_SPECS, QWEN2_SPEC, QWEN3_SPEC, TransformerConfig, detect_model_spec

    c = TransformerConfig(vocab_size=151936, hidden_size=1024, num_layers=28, num_heads=16, num_kv_heads=8, intermediate_size=3072,
                          _head_dim=128, norm_eps=1e-6, rope_theta=1e6)
    assert c.head_dim == 128 and c.num_kv_groups == 2 and not c.is_moe
    d = TransformerConfig(hidden_size=768, num_heads=12)
    assert d.head_dim == 64 and d.num_kv_heads == 12 and d.intermediate_size == 3072
    assert QWEN3_SPEC.use_qk_norm and QWEN3_SPEC.default_norm_eps == 1e-6 and QWEN3_SPEC.q_norm.format(layer=3).endswith("layers.3.self_attn.q_norm.weight")
    assert GPT2_SPEC.norm_type == "layernorm" and GPT2_SPEC.activation == "gelu" and not GPT2_SPEC.use_rope and GPT2_SPEC.qkv_combined
    assert set(MODEL_SPECS) == {"gpt2", "llama", "qwen3", "qwen2"}
    assert detect_model_spec(["model.embed_tokens.weight", "model.layers.0.self_attn.q_norm.weight"]) is QWEN3_SPEC
    assert detect_model_spec(["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.bias"]) is QWEN2_SPEC
    assert detect_model_spec(["model.embed_tokens.weight"]) is LLAMA_SPEC
    assert detect_model_spec(["wte.weight"]) is GPT2_SPEC
    with pytest.raises(ValueError):
        detect_model_spec(["foo"])


def test_host_sampler_matches_reference_golden():
    from pygpukit_amd.llm import sample_token

    lg = g1["sample_logits"]
    np.testing.assert_array_equal([sample_token(r, 0.0, 0, 1.0) for r in lg], g1["sample_t0"])
    np.testing.assert_array_equal([sample_token(r, 0.0, 50, 0.9) for r in lg], g1["sample_t0_k50_p09"])
    np.random.seed(0)
    assert 0 <= sample_token(lg[0], 0.8, 40, 0.95) < lg.shape[1]


def test_rope_tables_match_reference_golden():
    from pygpukit_amd.llm import precompute_freqs_cis

    cos, sin = precompute_freqs_cis(128, 64, 1e6)
    np.testing.assert_array_equal(cos.astype(np.float32), g1["rope_cos_tab"])
    np.testing.assert_array_equal(sin.astype(np.float32), g1["rope_sin_tab"])


def test_fp8_table_and_quantiser_match_oracle():
    from pygpukit_amd.llm.layers.linear import LinearFP8, quantize_fp8_host

    np.testing.assert_array_equal(LinearFP8._get_fp8_table(), O.fp8_e4m3_table())
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((256, 384)) * 0.02).astype(np.float32)
    codes, sbits = quantize_fp8_host(w)
    oc, os_ = O.quantize_fp8_e4m3_block(w)
    np.testing.assert_array_equal(sbits, os_)
    # identical up to the sign of zero (0x00 vs 0x80), which decodes to the same value
    np.testing.assert_array_equal(O.dequantize_fp8_e4m3_block(codes, sbits), O.dequantize_fp8_e4m3_block(oc, os_))
    assert not np.any((codes == 0x7F) | (codes == 0xFF))


def test_synthetic_weight_generator_is_the_oracles():
    from pygpukit_amd.llm import synthetic as S
    from tests.golden_cfg import TINY, cfg_checksum

    assert cfg_checksum(S.make_qwen3_weights(TINY, seed=40)) == cfg_checksum(O.make_qwen3_weights(TINY, seed=40, bf16=True))
    assert S.QWEN3_0_6B == {k: O.QWEN3_0_6B[k] for k in S.QWEN3_0_6B}
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    np.testing.assert_array_equal(S.f32_to_bf16_bits(x), O.f32_to_bf16_bits(x))


def test_shard_range_partitions_exactly():
    from pygpukit_amd.parallel import shard_range

    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_bench_accounting_formulas():
    """1.192 GB of weights per token, 114 688 B of KV per context row, 113/152.6 GFLOP prefill."""
    import bench

    ab = bench.algorithmic_bytes_per_token(O.QWEN3_0_6B, 128, "bf16")
    assert ab["weights"] + ab["lm_head"] == pytest.approx(596_050_944 * 2 + 311_296 - 155_582_464 * 0, rel=2e-3)
    assert ab["kv_read"] == 114_688 * 128 and ab["kv_write"] == 114_688
    assert bench.prefill_flops(O.QWEN3_0_6B, 128, all_rows=True) == pytest.approx(152.6e9 + 1.88e9, rel=5e-3)
    assert bench.prefill_flops(O.QWEN3_0_6B, 128, all_rows=False) == pytest.approx(113.1e9 + 1.88e9, rel=5e-3)


def test_legacy_configs_aliases_and_numpy_rope():
    """GPT2Config / LlamaConfig / Qwen3Config -> TransformerConfig (config.py:515-620), the legacy component names
    (models/causal.py:1496-1501) and the host RoPE helper (layers/rope.py:27-43) against the oracle's rope."""
    import numpy as np

    from oracle import cpu_ref as O
    from pygpukit_amd import llm

    g = llm.GPT2Config().to_transformer_config()
    assert (g.hidden_size, g.num_layers, g.num_heads, g.num_kv_heads, g.intermediate_size) == (768, 12, 12, 12, 3072)
    assert (g.norm_type, g.activation, g.use_rope, g.max_position_embeddings) == ("layernorm", "gelu", False, 1024)
    l = llm.LlamaConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, intermediate_size=14336,
                        num_hidden_layers=32, rope_theta=5e5).to_transformer_config()
    assert (l.head_dim, l.num_kv_groups, l.norm_type, l.activation, l.rope_theta) == (128, 4, "rmsnorm", "silu", 5e5)
    q = llm.Qwen3Config(hidden_size=1024, num_attention_heads=16, num_key_value_heads=8, intermediate_size=3072,
                        num_hidden_layers=28).to_transformer_config()
    assert (q.head_dim, q.norm_eps, q.rope_theta, q.vocab_size) == (128, 1e-6, 1e6, 151936)
    assert llm.RMSNorm is llm.Norm and llm.LayerNorm is llm.Norm and llm.LlamaAttention is llm.Attention
    assert llm.CausalSelfAttention is llm.Attention and llm.LlamaMLP is llm.MLP and llm.LlamaBlock is llm.TransformerBlock
    rng = np.random.default_rng(8)
    qh, kh = rng.standard_normal((5, 4, 64)).astype(np.float32), rng.standard_normal((5, 2, 64)).astype(np.float32)
    cos, sin = llm.precompute_freqs_cis(64, 32, 1e6)
    pos = [3, 4, 9, 20, 31]
    qe, ke = llm.apply_rotary_pos_emb_numpy(qh, kh, cos[pos], sin[pos])
    qo, ko = O.rope(qh, kh, cos[pos].astype(np.float32), sin[pos].astype(np.float32))
    np.testing.assert_allclose(qe, qo, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ke, ko, rtol=1e-6, atol=1e-6)


def test_backend_surface_without_a_cpu_fallback():
    """core.backend keeps the reference's names (backend.py:198-560): Backend ABC, DeviceProperties, NativeBackend,
    get/set/reset_backend, has_rust_module - and a CPUSimulationBackend that refuses to exist."""
    from pygpukit_amd.core import backend as B

    assert issubclass(B.HipBackend, B.Backend) and B.NativeBackend is B.HipBackend
    b = B.get_backend()
    assert isinstance(b, B.NativeBackend) and B.get_backend() is b
    B.reset_backend()
    assert B.get_backend() is not b
    B.set_backend(b)
    assert B.get_backend() is b
    assert B.has_rust_module() is False and B.get_rust_module() is None
    with pytest.raises(RuntimeError, match="no CPU simulation backend"):
        B.CPUSimulationBackend()
    with pytest.raises(TypeError):
        B.Backend()                      # abstract
    p = B.DeviceProperties(name="x", total_memory=1, arch="gfx950", warp_size=64)
    assert p["wavefront_size"] == 64 and p["arch"] == "gfx950" and p.max_threads_per_block == 1024
