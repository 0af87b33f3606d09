"""safetensors reader and checkpoint loader.  CPU: the C++ mmap reader against files written by the test
(format, dtypes, sharded index, malformed input).  GPU: tiny Qwen3 / GPT-2 / FP8 checkpoints written in Hugging Face
naming, loaded through load_model_from_safetensors and compared with the oracle model built from the same weights."""

from __future__ import annotations

import json
import os

import numpy as np
import pytest

from oracle import cpu_ref as O
from pygpukit_amd.llm import safetensors as ST
from tests.conftest import rel_err
from tests.golden_cfg import TINY


def _bf16(x):
    return O.f32_to_bf16_bits(np.ascontiguousarray(x, dtype=np.float32)).reshape(x.shape)


def test_reader_parses_header_and_serves_bytes(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 5)).astype(np.float32)
    b = _bf16(rng.standard_normal((4, 2, 6)).astype(np.float32))
    c = rng.integers(0, 255, (7,), dtype=np.uint8)
    d = np.array(3, np.int64)                                          # rank-0 tensor
    p = str(tmp_path / "m.safetensors")
    ST.save_safetensors(p, {"a.weight": (a, "F32"), "deep.name.b": (b, "BF16"), "c": (c, "F8_E4M3"), "d": (d, "I64")},
                        {"format": "pt", "note": 'quote " and { brace'})
    f = ST.load_safetensors(p)
    assert isinstance(f, ST.SafeTensorsFile) and len(f) == 4 and f.num_tensors == 4 and f.file_size == os.path.getsize(p)
    assert f.tensor_names == ["a.weight", "deep.name.b", "c", "d"] and "c" in f and "zzz" not in f
    ia, ib, ic, id_ = (f.tensor_info(n) for n in f.tensor_names)
    assert (ia.dtype, ia.shape, ia.size_bytes, ia.numel, ia.dtype_name) == (ST.Dtype.Float32, [3, 5], 60, 15, "float32")
    assert (ib.dtype, ib.shape, ib.size_bytes) == (ST.Dtype.BFloat16, [4, 2, 6], 96)
    assert (ic.dtype, ic.dtype_name, ic.size_bytes) == (ST.Dtype.Float8E4M3, "float8_e4m3", 7)
    assert (id_.dtype, id_.shape, id_.size_bytes) == (ST.Dtype.Int64, [], 8)
    np.testing.assert_array_equal(f.tensor_numpy("a.weight"), a)
    np.testing.assert_array_equal(f.tensor_numpy("deep.name.b"), b)
    np.testing.assert_array_equal(f.tensor_as_f32("deep.name.b"), O.bf16_bits_to_f32(b))
    assert f.tensor_bytes("c") == c.tobytes()
    ptr, n = f.tensor_data_ptr("a.weight")
    assert n == 60 and ptr % 4 == 0 and open(p, "rb").read()[ia.offset:ia.offset + 60] == a.tobytes()
    with pytest.raises(KeyError):
        f.tensor_info("missing")
    with pytest.raises(ValueError):
        f.tensor_as_f32("c")
    assert "num_tensors=4" in repr(f)


def test_reader_rejects_bad_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        ST.SafeTensorsFile(str(tmp_path / "nope.safetensors"))
    bad = tmp_path / "bad.safetensors"
    bad.write_bytes((10**9).to_bytes(8, "little") + b"{}")               # header longer than the file
    with pytest.raises(ValueError, match="header length"):
        ST.SafeTensorsFile(str(bad))
    hj = json.dumps({"t": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    bad.write_bytes(len(hj).to_bytes(8, "little") + hj + b"\0" * 8)      # data shorter than declared
    with pytest.raises(ValueError, match="outside the file"):
        ST.SafeTensorsFile(str(bad))
    hj = json.dumps({"t": {"dtype": "Q4", "shape": [1], "data_offsets": [0, 1]}}).encode()
    bad.write_bytes(len(hj).to_bytes(8, "little") + hj + b"\0")
    with pytest.raises(ValueError, match="unknown dtype"):
        ST.SafeTensorsFile(str(bad))
    # a crafted header must not be able to point outside the mapping: negative or wrapping offsets, offsets past the file
    # after the add, shapes that disagree with the byte range, negative / overflowing / fractional numbers
    for entry, why in (
            ('{"dtype": "U8", "shape": [1], "data_offsets": [0, -1]}', "malformed"),
            ('{"dtype": "U8", "shape": [1], "data_offsets": [0, 18446744073709551615]}', "malformed"),
            ('{"dtype": "U8", "shape": [1], "data_offsets": [0, 4611686018427387904]}', "malformed|outside the file"),
            ('{"dtype": "U8", "shape": [16], "data_offsets": [8, 24]}', "outside the file"),
            ('{"dtype": "F32", "shape": [1000, 1000], "data_offsets": [0, 16]}', "does not match"),
            ('{"dtype": "F32", "shape": [4294967296, 4294967296], "data_offsets": [0, 16]}', "does not match"),
            ('{"dtype": "F32", "shape": [-4], "data_offsets": [0, 16]}', "malformed"),
            ('{"dtype": "F32", "shape": [4.0], "data_offsets": [0, 16]}', "malformed"),
            ('{"dtype": "F32", "shape": [2], "data_offsets": [8, 4]}', "outside the file|does not match"),
            ('{"dtype": "BF16", "shape": [3], "data_offsets": [0, 16]}', "does not match")):
        hj = ('{"t": ' + entry + "}").encode()
        bad.write_bytes(len(hj).to_bytes(8, "little") + hj + b"\0" * 16)
        with pytest.raises(ValueError, match=why):
            ST.SafeTensorsFile(str(bad))
    hj = b'{"t": {"dtype": "F32", "shape": [0, 7], "data_offsets": [4, 4]}}'      # an empty tensor is legal
    bad.write_bytes(len(hj).to_bytes(8, "little") + hj + b"\0" * 16)
    assert ST.SafeTensorsFile(str(bad)).tensor_info("t").size_bytes == 0


def test_sharded_index(tmp_path):
    a, b = np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(4, dtype=np.int32)
    ST.save_safetensors(str(tmp_path / "s1.safetensors"), {"a": (a, "F32")})
    ST.save_safetensors(str(tmp_path / "s2.safetensors"), {"b": (b, "I32")})
    idx = tmp_path / "model.safetensors.index.json"
    idx.write_text(json.dumps({"metadata": {}, "weight_map": {"a": "s1.safetensors", "b": "s2.safetensors"}}))
    f = ST.load_safetensors(str(idx))
    assert isinstance(f, ST.ShardedSafeTensorsFile) and sorted(f.tensor_names) == ["a", "b"] and "b" in f and len(f) == 2
    np.testing.assert_array_equal(f.tensor_numpy("b"), b)
    assert f.tensor_info("a").shape == [2, 3] and f.file_size > 0
    with pytest.raises(KeyError):
        f.tensor_info("c")


def test_fp8_quant_config_detection():
    from pygpukit_amd.llm.loader import FP8QuantConfig

    assert FP8QuantConfig.from_config({}) is None
    assert FP8QuantConfig.from_config({"quantization_config": {"quant_method": "awq"}}) is None
    q = FP8QuantConfig.from_config({"quantization_config": {"quant_method": "fp8", "fmt": "e4m3", "weight_block_size": [128, 128]}})
    assert q.fmt == "e4m3" and q.weight_block_size == (128, 128)


# ----------------------------------------------------------------------------------------------- GPU: whole models
def _write_qwen3(tmp_path, cfg, w, fp8=False):
    t = {"model.embed_tokens.weight": (_bf16(w["embed"]), "BF16"), "model.norm.weight": (_bf16(w["final_norm"]), "BF16")}
    for i, lw in enumerate(w["layers"]):
        L = f"model.layers.{i}."
        t[L + "input_layernorm.weight"] = (_bf16(lw["attn_norm"]), "BF16")
        t[L + "post_attention_layernorm.weight"] = (_bf16(lw["mlp_norm"]), "BF16")
        if "q_norm" in lw:
            t[L + "self_attn.q_norm.weight"] = (_bf16(lw["q_norm"]), "BF16")
            t[L + "self_attn.k_norm.weight"] = (_bf16(lw["k_norm"]), "BF16")
        for key, name in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"),
                          ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"), ("down", "mlp.down_proj")):
            if fp8:
                codes, sbits = O.quantize_fp8_e4m3_block(lw[key])
                t[L + name + ".weight"] = (codes, "F8_E4M3")
                t[L + name + ".weight_scale_inv"] = (sbits, "BF16")
            else:
                t[L + name + ".weight"] = (_bf16(lw[key]), "BF16")
    p = str(tmp_path / "model.safetensors")
    ST.save_safetensors(p, t, {"format": "pt"})
    conf = {"model_type": "qwen3", "rope_theta": cfg["rope_theta"], "rms_norm_eps": cfg["norm_eps"], "max_position_embeddings": 256}
    if fp8:
        conf["quantization_config"] = {"quant_method": "fp8", "fmt": "e4m3", "weight_block_size": [128, 128]}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    return p


@pytest.mark.gpu
def test_load_qwen3_checkpoint_matches_oracle(tmp_path):
    from pygpukit_amd.llm.loader import load_model_from_safetensors

    w = O.make_qwen3_weights(TINY, seed=40, bf16=True)
    model = load_model_from_safetensors(_write_qwen3(tmp_path, TINY, w))
    c = model.config
    assert (c.vocab_size, c.hidden_size, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.intermediate_size) == (
        TINY["vocab_size"], TINY["hidden_size"], TINY["num_layers"], TINY["num_heads"], TINY["num_kv_heads"], TINY["head_dim"],
        TINY["intermediate_size"])
    assert model.spec.name == "qwen3" and c.rope_theta == TINY["rope_theta"] and c.norm_eps == TINY["norm_eps"]
    prompt = [int(t) for t in np.random.default_rng(6).integers(0, TINY["vocab_size"], 12)]
    ref = O.build_qwen3_ref(TINY, w, max_pos=256)
    hid, _ = ref(prompt)
    want = ref.get_logits(hid)
    h, _ = model(prompt)
    got = O.bf16_bits_to_f32(model.get_logits(h).to_numpy())
    assert rel_err(got, want) < 1e-2
    assert model.generate(prompt, max_new_tokens=4, temperature=0.0, top_k=0, top_p=1.0) == ref.generate(
        prompt, max_new_tokens=4, temperature=0.0, top_k=0, top_p=1.0)


@pytest.mark.gpu
def test_load_fp8_checkpoint_uses_linear_fp8(tmp_path):
    from pygpukit_amd.llm.layers.linear import LinearFP8
    from pygpukit_amd.llm.loader import load_model_from_safetensors

    cfg = dict(TINY, hidden_size=256, intermediate_size=512)          # every projection a multiple of 128 in both dims
    w = O.make_qwen3_weights(cfg, seed=41, bf16=True)
    model = load_model_from_safetensors(_write_qwen3(tmp_path, cfg, w, fp8=True))
    assert isinstance(model.blocks[0].attn.q_proj, LinearFP8) and isinstance(model.blocks[1].mlp.down_proj, LinearFP8)
    wq = {"embed": w["embed"], "final_norm": w["final_norm"], "layers": []}
    for lw in w["layers"]:
        d = dict(lw)
        for k in ("q", "k", "v", "o", "gate", "up", "down"):
            d[k] = O.dequantize_fp8_e4m3_block(*O.quantize_fp8_e4m3_block(lw[k]))
        wq["layers"].append(d)
    prompt = [3, 77, 512, 9, 1000, 41]
    ref = O.build_qwen3_ref(cfg, wq, max_pos=256)
    hid, _ = ref(prompt)
    h, _ = model(prompt)
    assert rel_err(O.bf16_bits_to_f32(model.get_logits(h).to_numpy()), ref.get_logits(hid)) < 1e-2


@pytest.mark.gpu
def test_load_gpt2_checkpoint_conv1d_layout(tmp_path):
    """GPT-2 checkpoints store Conv1D weights ([in, out]) and one fused c_attn: the loader transposes and splits."""
    from pygpukit_amd.llm.loader import load_model_from_safetensors

    cfg = dict(vocab_size=512, hidden_size=128, num_layers=2, num_heads=2, num_kv_heads=2, head_dim=64, intermediate_size=512,
               max_position_embeddings=64, norm_eps=1e-5)
    w = O.make_gpt2_weights(cfg, seed=2)
    H = cfg["hidden_size"]
    t = {"wte.weight": (w["wte"], "F32"), "wpe.weight": (w["wpe"], "F32"), "ln_f.weight": (np.ones(H, np.float32), "F32"),
         "ln_f.bias": (np.zeros(H, np.float32), "F32")}
    for i, lw in enumerate(w["layers"]):
        g = f"h.{i}."
        for ln in ("ln_1", "ln_2"):
            t[g + ln + ".weight"] = (np.ones(H, np.float32), "F32")
            t[g + ln + ".bias"] = (np.zeros(H, np.float32), "F32")
        t[g + "attn.c_attn.weight"] = (np.ascontiguousarray(np.concatenate([lw["q"], lw["k"], lw["v"]], axis=0).T), "F32")
        t[g + "attn.c_proj.weight"] = (np.ascontiguousarray(lw["o"].T), "F32")
        t[g + "mlp.c_fc.weight"] = (np.ascontiguousarray(lw["fc1"].T), "F32")
        t[g + "mlp.c_proj.weight"] = (np.ascontiguousarray(lw["fc2"].T), "F32")
    p = str(tmp_path / "model.safetensors")
    ST.save_safetensors(p, t)
    (tmp_path / "config.json").write_text(json.dumps({"model_type": "gpt2", "n_head": 2, "n_positions": 64, "layer_norm_epsilon": 1e-5}))
    model = load_model_from_safetensors(p, dtype="float32")
    assert model.spec.name == "gpt2" and model.config.num_heads == 2 and model.config.intermediate_size == 512
    ref = O.build_gpt2_ref(cfg, w)
    prompt = [1, 2, 3, 4, 200, 17]
    hid, _ = ref(prompt)
    h, _ = model(prompt)
    np.testing.assert_allclose(model.get_logits(h).to_numpy(), ref.get_logits(hid), rtol=2e-4, atol=2e-5)
