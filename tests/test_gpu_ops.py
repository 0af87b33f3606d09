"""GPU parity tests for the op surface: HIP kernels (through the C ABI) vs the CPU oracle and the
golden vectors generated from the reference.  Tolerances: fp32 ops ~1e-5 (summation order only);
bf16 <= 1e-2 relative L2 error (BASELINE.json / reference tests/test_gemv_correctness.py:144-149);
fp8 <= 5e-2; copies and index ops bit-exact."""

from __future__ import annotations

import os

import numpy as np
import pytest

from oracle import cpu_ref as O
from tests.conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

pk = pytest.importorskip("pygpukit_amd")
from pygpukit_amd import ops  # noqa: E402
from pygpukit_amd.core import bfloat16, float16, float32, from_numpy  # noqa: E402

g1 = load_golden("g1_ops.npz")
DT = {"float32": float32, "float16": float16, "bfloat16": bfloat16}
TOL = {"float32": 2e-5, "float16": 2e-3, "bfloat16": 1e-2}


def dev(x: np.ndarray, dt: str = "float32"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dt == "float32":
        return from_numpy(x)
    if dt == "float16":
        return from_numpy(x.astype(np.float16))
    return from_numpy(O.f32_to_bf16_bits(x).reshape(x.shape))


def host(a) -> np.ndarray:
    h = a.to_numpy()
    if a.dtype == bfloat16:
        return O.bf16_bits_to_f32(h)
    return h.astype(np.float32)


def rounded(x: np.ndarray, dt: str) -> np.ndarray:
    """The value the device actually holds for x in dtype dt."""
    if dt == "bfloat16":
        return O.bf16_round(x)
    if dt == "float16":
        return x.astype(np.float16).astype(np.float32)
    return x.astype(np.float32)


def close(a, ref, dt):
    e = rel_err(a, ref)
    assert e <= TOL[dt], f"rel err {e:.3e} > {TOL[dt]} ({dt})"


# ----------------------------------------------------------------------------- runtime
def test_device_info_and_raw_copies():
    """core.device / core.memory with the reference's names (device.py:11-120, memory.py:18-215)."""
    from pygpukit_amd.core import memory as M

    assert pk.is_cuda_available()
    info = pk.get_device_info(0)
    assert isinstance(info, pk.DeviceInfo) and info.warp_size == 64 and info.multiprocessor_count >= 64
    assert info.compute_capability == (9, 5) and info.total_memory > (64 << 30) and info.max_threads_per_block == 1024
    caps = pk.get_device_capabilities(0)
    assert caps.sm_version == 950 and caps.tensorcore_bf16 and caps.async_copy and caps.name == info.name
    free, total = M.get_memory_info()
    assert 0 < free <= total == info.total_memory
    src = np.arange(4096, dtype=np.float32)
    a, b, c = pk.zeros((4096,)), pk.zeros((4096,)), pk.zeros((4096,))
    M.synchronize()          # the zero fills run on the library's stream; the copies below use another one
    M.copy_to_device(a, src.ctypes.data, src.nbytes)
    np.testing.assert_array_equal(a.to_numpy(), src)
    st = pk.Stream()
    M.copy_to_device_async(b, src.ctypes.data, src.nbytes // 2, st)
    M.copy_device_to_device_offset(b, src.nbytes // 2, a, 0, src.nbytes // 2)
    st.synchronize()
    M.synchronize()
    np.testing.assert_array_equal(b.to_numpy(), np.concatenate([src[:2048], src[:2048]]))
    M.copy_device_to_device_async(c, a, st)
    st.synchronize()
    np.testing.assert_array_equal(c.to_numpy(), src)
    with pytest.raises(ValueError):
        M.copy_device_to_device_async(pk.zeros((8,)), a, st)
    with pytest.raises(ValueError):
        M.copy_to_device(pk.zeros((8,)), src.ctypes.data, src.nbytes)


def test_device_is_mi355x_and_pool_works():
    b = pk.get_backend()
    assert b.is_available()
    props = b.get_device_properties(0)
    assert props["wavefront_size"] == 64
    s0 = b.pool_stats()
    a = pk.zeros((1000,), "float32")
    p = a.data_ptr()
    del a
    a2 = pk.zeros((1000,), "float32")  # same size class -> served from the free list
    s1 = b.pool_stats()
    assert a2.data_ptr() == p
    assert s1["n_pool_hit"] >= s0["n_pool_hit"] + 1


def test_roundtrip_and_views():
    x = np.arange(24, dtype=np.float32).reshape(4, 6)
    a = from_numpy(x)
    np.testing.assert_array_equal(a.to_numpy(), x)
    np.testing.assert_array_equal(pk.ones((3, 2), "bfloat16").astype(float32).to_numpy(), np.ones((3, 2), np.float32))
    row = from_numpy(x[:1])
    v = row.narrow(2, 3)
    np.testing.assert_array_equal(v.to_numpy(), x[:1, 2:5])
    np.testing.assert_array_equal(a.view((6, 4)).to_numpy(), x.reshape(6, 4))
    np.testing.assert_array_equal(a.slice_rows(2).to_numpy(), x[:2])
    np.testing.assert_array_equal(a.T.to_numpy(), x.T)
    np.testing.assert_array_equal(a.reshape(-1, 8).to_numpy(), x.reshape(-1, 8))
    np.testing.assert_array_equal(a[1:3, ::2].to_numpy(), x[1:3, ::2])
    np.testing.assert_array_equal(a.clone().to_numpy(), x)
    with pytest.raises(ValueError):
        a.view((5, 5))
    # bf16 RNE on the device == reference formula (golden bits)
    v = g1["bf16_in"][:1000]
    np.testing.assert_array_equal(from_numpy(v).astype(bfloat16).to_numpy(), g1["bf16_bits"][:1000])


# ----------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("dt", ["float32", "float16", "bfloat16"])
def test_binary_and_inplace(dt):
    rng = np.random.default_rng(0)
    for n in (1, 7, 1024, 4099):
        a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32) + 3.0
        ar, br = rounded(a, dt), rounded(b, dt)
        A, B = dev(a, dt), dev(b, dt)
        close(host(ops.add(A, B)), ar + br, dt)
        close(host(ops.sub(A, B)), ar - br, dt)
        close(host(ops.mul(A, B)), ar * br, dt)
        close(host(ops.div(A, B)), ar / br, dt)
        ops.add_inplace(A, B)
        close(host(A), ar + br, dt)
        ops.mul_inplace(B, B)
        close(host(B), br * br, dt)
    with pytest.raises(ValueError):
        ops.add(dev(np.zeros(3)), dev(np.zeros(4)))


def test_elementwise_golden_fp32():
    x = g1["act_x"]
    np.testing.assert_allclose(host(ops.silu(dev(x))), g1["silu"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(ops.gelu(dev(x))), g1["gelu"], rtol=2e-5, atol=2e-6)
    np.testing.assert_array_equal(host(ops.add(dev(x), dev(x[::-1].copy()))), g1["add"])
    np.testing.assert_array_equal(host(ops.mul(dev(x), dev(x[::-1].copy()))), g1["mul"])


def test_basic_ops_golden_fp32():
    """The rest of ops.basic against fixture G6 (outputs of the reference's NumPy path): unary math, whole-array
    reductions (shape [1], input dtype; argmax int64), softmax 2-D / 3-D, sum_axis, clamp, where."""
    g6 = load_golden("g6_basic_ops.npz")
    x, pos = g6["x"], g6["pos"]
    for name in ("exp", "relu", "sin", "cos", "abs", "neg", "sigmoid", "tanh", "relu2"):
        np.testing.assert_allclose(host(getattr(ops, name)(dev(x))), g6[name], rtol=3e-6, atol=1e-6, err_msg=name)
    for name in ("log", "sqrt", "rsqrt"):
        np.testing.assert_allclose(host(getattr(ops, name)(dev(pos))), g6[name], rtol=3e-6, atol=1e-6, err_msg=name)
    for name in ("sum", "mean", "max", "min"):
        got = getattr(ops, name)(dev(x))
        assert got.shape == (1,) and got.dtype == pk.float32
        np.testing.assert_allclose(host(got), g6["red_" + name], rtol=2e-5, atol=2e-4, err_msg=name)   # fp32 sums of 2331 terms, another order
    am = ops.argmax(dev(x))
    assert am.shape == (1,) and am.dtype == pk.int64
    np.testing.assert_array_equal(am.to_numpy(), g6["red_argmax"])
    np.testing.assert_allclose(host(ops.softmax(dev(x))), g6["softmax"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(host(ops.softmax(dev(g6["x3"]))), g6["softmax3"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(host(ops.sum_axis(dev(x), 0)), g6["sum_axis0"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(host(ops.sum_axis(dev(x), 1)), g6["sum_axis1"], rtol=1e-5, atol=1e-4)
    np.testing.assert_array_equal(host(ops.clamp(dev(x), -0.5, 1.25)), g6["clamp"])
    cond = from_numpy(g6["cond"])
    np.testing.assert_array_equal(host(ops.where(cond, dev(x), dev(g6["y"]))), g6["where"])
    with pytest.raises(ValueError):
        ops.softmax(dev(x[0]))
    with pytest.raises(ValueError):
        ops.sum_axis(dev(x), 2)
    with pytest.raises(ValueError):
        ops.where(cond, dev(x), dev(g6["y"][:3].copy()))


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_basic_ops_half_precision_vs_oracle(dt):
    """Same ops on 16-bit storage: fp32 math on the rounded inputs, one rounding on the way out; a 1M-element
    reduction is the same on every run (fixed two-level tree)."""
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((9, 1000)) * 1.5).astype(np.float32)
    xr = rounded(x, dt)
    for name in ("exp", "relu", "sin", "cos", "abs", "neg"):
        close(host(getattr(ops, name)(dev(x, dt))), O.unary(name, xr), dt)
    pr = np.abs(xr) + rounded(np.full_like(xr, 0.25), dt)
    for name in ("log", "sqrt", "rsqrt"):
        close(host(getattr(ops, name)(dev(pr, dt))), O.unary(name, rounded(pr, dt)), dt)
    close(host(ops.softmax(dev(x, dt))), O.softmax_last(xr), dt)
    close(host(ops.sum_axis(dev(x, dt), 1)), O.sum_axis(xr, 1), dt)
    close(host(ops.clamp(dev(x, dt), -1.0, 0.5)), O.clamp(xr, -1.0, 0.5), dt)
    assert abs(float(host(ops.max(dev(x, dt)))[0]) - xr.max()) == 0.0 and abs(float(host(ops.min(dev(x, dt)))[0]) - xr.min()) == 0.0
    big = rng.standard_normal(1 << 20).astype(np.float32)
    d = dev(big)
    first = host(ops.sum(d))[0]
    assert all(host(ops.sum(d))[0] == first for _ in range(3))
    assert abs(first - big.astype(np.float64).sum()) < 1e-3 * np.sqrt(big.size)
    assert abs(host(ops.mean(d))[0] - big.mean(dtype=np.float64)) < 1e-5


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
def test_activations_glu_bias_cast(dt):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((5, 264)) * 2).astype(np.float32)
    u = rng.standard_normal((5, 264)).astype(np.float32)
    xr, ur = rounded(x, dt), rounded(u, dt)
    close(host(ops.silu(dev(x, dt))), O.silu(xr), dt)
    close(host(ops.gelu(dev(x, dt))), O.gelu(xr), dt)
    close(host(ops.swiglu(dev(x, dt), dev(u, dt))), O.swiglu(xr, ur), dt)
    X = dev(x, dt)
    ops.silu(X, out=X)  # in place
    close(host(X), O.silu(xr), dt)
    b = rng.standard_normal(264).astype(np.float32)
    Y = dev(x, dt)
    ops.bias_add_inplace(Y, dev(b, dt))
    close(host(Y), xr + rounded(b, dt), dt)
    if dt == "float32":
        np.testing.assert_array_equal(ops.cast_f32_to_bf16(dev(x)).to_numpy(), O.f32_to_bf16_bits(x))
        np.testing.assert_array_equal(host(ops.cast_bf16_to_f32(dev(x, "bfloat16"))), O.bf16_round(x))
        np.testing.assert_array_equal(ops.cast_f32_to_f16(dev(x)).to_numpy(), x.astype(np.float16))


# ----------------------------------------------------------------------------- norms / rope
def test_norms_golden_fp32():
    x, g, b = g1["norm_x"], g1["norm_g"], g1["norm_b"]
    np.testing.assert_allclose(host(ops.rmsnorm(dev(x), dev(g), 1e-6)), g1["rmsnorm_1e6"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(ops.rmsnorm(dev(x), dev(g), 1e-5)), g1["rmsnorm_1e5"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(host(ops.layernorm(dev(x), dev(g), dev(b), 1e-5)), g1["layernorm_1e5"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
@pytest.mark.parametrize("shape", [(1, 1024), (16, 128), (3, 4096), (2, 5000), (5, 100)])
def test_rmsnorm_shapes(dt, shape):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(shape[1])).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32)
    xr, gr, rr = rounded(x, dt), rounded(g, dt), rounded(r, dt)
    close(host(ops.rmsnorm(dev(x, dt), dev(g, dt), 1e-6)), O.rmsnorm(xr, gr, 1e-6), dt)
    close(host(ops.rmsnorm_residual(dev(x, dt), dev(r, dt), dev(g, dt), 1e-6)), O.rmsnorm_residual(xr, rr, gr, 1e-6), dt)
    out = dev(np.zeros(shape), dt)
    ops.rmsnorm(dev(x, dt), dev(g, dt), 1e-6, out=out)
    close(host(out), O.rmsnorm(xr, gr, 1e-6), dt)


def test_rope_golden_and_bf16():
    pos = g1["rope_pos"]
    cos, sin = g1["rope_cos_tab"][pos], g1["rope_sin_tab"][pos]
    q, k = dev(g1["rope_q"]), dev(g1["rope_k"])
    ops.rope_inplace(q, k, dev(cos), dev(sin))
    np.testing.assert_allclose(host(q), g1["rope_q_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(host(k), g1["rope_k_out"], rtol=1e-6, atol=1e-6)
    qb, kb = dev(g1["rope_q"], "bfloat16"), dev(g1["rope_k"], "bfloat16")
    ops.rope_inplace_f32table(qb, kb, dev(cos), dev(sin))
    qr, kr = O.rope(O.bf16_round(g1["rope_q"]), O.bf16_round(g1["rope_k"]), cos, sin)
    close(host(qb), qr, "bfloat16")
    close(host(kb), kr, "bfloat16")


# ----------------------------------------------------------------------------- byte movers
def test_shuffles_bit_exact():
    t = g1["shuffle_in"]
    np.testing.assert_array_equal(ops.repeat_interleave_axis1(dev(t), 3).to_numpy(), g1["repeat_interleave_3"])
    np.testing.assert_array_equal(ops.transpose_3d_021(dev(t)).to_numpy(), g1["transpose_3d_021"])
    np.testing.assert_array_equal(ops.concat_axis0(dev(t), dev(t[:2].copy())).to_numpy(), g1["concat_axis0"])
    np.testing.assert_array_equal(ops.transpose(dev(g1["transpose_in"])).to_numpy(), g1["transpose_out"])
    rng = np.random.default_rng(3)
    for shape in [(1, 1), (65, 130), (300, 77), (128, 1024)]:
        x = rng.integers(0, 60000, shape).astype(np.uint16)
        np.testing.assert_array_equal(ops.transpose(from_numpy(x)).to_numpy(), x.T)
        xb = rng.integers(0, 255, shape).astype(np.uint8)
        np.testing.assert_array_equal(ops.transpose(from_numpy(xb)).to_numpy(), xb.T)
    x3 = rng.integers(0, 60000, (5, 7, 24)).astype(np.uint16)
    np.testing.assert_array_equal(ops.transpose_3d_021(from_numpy(x3)).to_numpy(), x3.transpose(1, 0, 2))
    np.testing.assert_array_equal(ops.reshape_copy(from_numpy(x3), (35, 24)).to_numpy(), x3.reshape(35, 24))
    qkv = rng.integers(0, 60000, (6, 32 + 16 + 16)).astype(np.uint16)
    q, k, v = pk.empty((6, 32), "bfloat16"), pk.empty((6, 16), "bfloat16"), pk.empty((6, 16), "bfloat16")
    ops.split_qkv_batch(from_numpy(qkv), q, k, v, 32, 16, 16)
    np.testing.assert_array_equal(q.to_numpy(), qkv[:, :32])
    np.testing.assert_array_equal(k.to_numpy(), qkv[:, 32:48])
    np.testing.assert_array_equal(v.to_numpy(), qkv[:, 48:])


def test_embedding_and_kv_cache():
    rng = np.random.default_rng(4)
    E = rng.integers(0, 60000, (50, 96)).astype(np.uint16)
    Ed = from_numpy(E)
    out = pk.empty((1, 96), "bfloat16")
    ops.embedding_lookup(Ed, out, 17)
    np.testing.assert_array_equal(out.to_numpy(), E[17:18])
    ops.embedding_lookup_ptr(Ed, out, from_numpy(np.array([33], np.int32)))
    np.testing.assert_array_equal(out.to_numpy(), E[33:34])
    ids = np.array([3, 49, 0, 7], np.int32)
    outb = pk.empty((4, 96), "bfloat16")
    ops.embedding_lookup_batch(Ed, outb, from_numpy(ids), 4)
    np.testing.assert_array_equal(outb.to_numpy(), E[ids])
    sl = pk.empty((3, 96), "bfloat16")
    ops.slice_rows_range_ptr(Ed, sl, from_numpy(np.array([10], np.int32)), 3)
    np.testing.assert_array_equal(sl.to_numpy(), E[10:13])
    with pytest.raises(ValueError):
        ops.embedding_lookup(Ed, out, 50)
    # KV cache: expanded (reference layout) and un-expanded
    Hq, Hkv, D, MAX = 4, 2, 32, 16
    for hc in (Hq, Hkv):
        cache = np.zeros((hc, MAX, D), np.float32)
        cd = from_numpy(cache)
        new1 = rng.standard_normal((1, Hkv, D)).astype(np.float32)
        ops.kv_cache_update_gqa(from_numpy(new1), cd, Hq, 5)
        O.kv_cache_update_gqa(new1, cache, Hq, 5)
        new2 = rng.standard_normal((1, Hkv, D)).astype(np.float32)
        ops.kv_cache_update_gqa_ptr(from_numpy(new2), cd, Hq, from_numpy(np.array([6], np.int32)))
        O.kv_cache_update_gqa(new2, cache, Hq, 6)
        new3 = rng.standard_normal((4, Hkv, D)).astype(np.float32)
        ops.kv_cache_prefill_gqa(from_numpy(new3), cd, Hq, 8)
        O.kv_cache_prefill_gqa(new3, cache, Hq, 8)
        np.testing.assert_array_equal(cd.to_numpy(), cache)
        with pytest.raises(ValueError):
            ops.kv_cache_update_gqa(from_numpy(new1), cd, Hq, MAX)


def test_argmax_lowest_index_ties():
    lg = g1["sample_logits"].copy()
    for r, want in zip(lg, g1["sample_t0"]):
        assert ops.sample_greedy(dev(r)) == int(want)
        assert ops.sample_greedy(dev(r, "bfloat16")) == int(np.argmax(O.bf16_round(r)))
    big = np.zeros(151936, np.float32)
    big[[150000, 70000, 1234]] = 5.0
    assert int(ops.argmax(dev(big)).to_numpy()[0]) == 1234 and ops.argmax_int(dev(big)) == 1234
    rows = ops.argmax_rows(dev(lg)).to_numpy()
    np.testing.assert_array_equal(rows, np.argmax(lg, axis=1))
    assert ops.sample_token_gpu(dev(lg[0]), 0.0, 0, 1.0) == int(np.argmax(lg[0]))


# ----------------------------------------------------------------------------- GEMV / GEMM
@pytest.mark.parametrize("dt", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("kn", [(1024, 4096), (2048, 1024), (3072, 1024), (1024, 6144), (100, 37), (4096, 130)])
def test_gemv(dt, kn):
    K, N = kn
    rng = np.random.default_rng(5)
    a = rng.standard_normal(K).astype(np.float32)
    b = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = (rounded(b, dt).astype(np.float64) @ rounded(a, dt).astype(np.float64)).astype(np.float32)
    A, B = dev(a, dt), dev(b, dt)
    if dt == "bfloat16":
        c = ops.gemv_bf16(A, B)
        out = pk.empty((N,), "bfloat16")
        ops.gemv_bf16(A, B, out=out)
        np.testing.assert_array_equal(out.to_numpy(), c.to_numpy())
    else:
        c = ops.matmul_nt(A.view((1, K)), B).view((N,))
    close(host(c), ref, dt)


def test_gemv_bf16_vs_unrounded_fp32_reference_bar():
    """The reference's own check: bf16 GEMV vs fp32 matmul of the UNROUNDED inputs, rel err < 1e-2."""
    rng = np.random.default_rng(6)
    K, N = 2048, 8192
    a, b = rng.standard_normal(K).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    c = host(ops.gemv_bf16(dev(a, "bfloat16"), dev(b, "bfloat16")))
    assert rel_err(c, b @ a) < 1e-2


def test_fp8_hardware_decode_matches_reference_table():
    """Every finite E4M3 code through the GEMV kernel == the reference LUT (quant.py:292-320)."""
    table = O.fp8_e4m3_table()
    codes = np.array([c for c in range(256) if c not in (0x7F, 0xFF)], np.uint8)
    K = N = 256
    W = np.zeros((N, K), np.uint8)
    for i, c in enumerate(codes):
        W[i, i] = c  # row i picks code c against a one-hot... use identity activation instead
    scale = O.f32_to_bf16_bits(np.ones((2, 2), np.float32))
    eye = np.eye(K, dtype=np.float32)
    out = host(ops.gemv_fp8_bf16_batched(dev(eye[:8], "bfloat16"), from_numpy(W), from_numpy(scale)))
    for m in range(8):
        np.testing.assert_array_equal(out[m, m], O.bf16_round(np.array([table[codes[m]]], np.float32))[0])
    # all codes: y[n] = sum_k x[k] * W[n,k] with x = ones picks the diagonal code
    y = host(ops.gemv_fp8_bf16(dev(np.ones(K, np.float32), "bfloat16"), from_numpy(W), from_numpy(scale)))
    np.testing.assert_array_equal(y[: len(codes)], O.bf16_round(table[codes]))


@pytest.mark.parametrize("m", [1, 3, 8, 11])
def test_gemv_fp8(m):
    rng = np.random.default_rng(7)
    K, N = 1024, 512
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    codes, sbits = O.quantize_fp8_e4m3_block(w)
    a = rng.standard_normal((m, K)).astype(np.float32)
    ref = O.bf16_round(a) @ O.dequantize_fp8_e4m3_block(codes, sbits).T
    if m == 1:
        c = host(ops.gemv_fp8_bf16(dev(a[0], "bfloat16"), from_numpy(codes), from_numpy(sbits)))[None]
    else:
        c = host(ops.gemv_fp8_bf16_batched(dev(a, "bfloat16"), from_numpy(codes), from_numpy(sbits)))
    assert rel_err(c, ref) < 1e-2  # vs the fp32 dequantised oracle
    assert rel_err(c, O.bf16_round(a) @ w.T) < 5e-2  # vs the unquantised weights (BASELINE fp8 bar)


@pytest.mark.parametrize("dt", ["bfloat16", "float16", "float32"])
@pytest.mark.parametrize("mnk", [(128, 256, 1024), (16, 96, 80), (33, 100, 72), (7, 19, 33), (200, 1024, 512), (64, 3072, 1024)])
def test_gemm_nt_nn(dt, mnk):
    M, N, K = mnk
    rng = np.random.default_rng(8)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ar, wr = rounded(a, dt).astype(np.float64), rounded(w, dt).astype(np.float64)
    ref = (ar @ wr.T).astype(np.float32)
    close(host(ops.matmul_nt(dev(a, dt), dev(w, dt))), ref, dt)
    close(host(ops.matmul_nt(dev(a, dt), dev(w, dt), dev(bias, dt))), ref + rounded(bias, dt), dt)
    close(host(ops.matmul(dev(a, dt), dev(np.ascontiguousarray(w.T), dt))), ref, dt)
    out = pk.empty((M, N), dt)
    ops.matmul(dev(a, dt), dev(np.ascontiguousarray(w.T), dt), out=out)
    close(host(out), ref, dt)


def test_matmul_golden_fp32():
    for tag in ("m1", "m16", "m128", "odd"):
        c = host(ops.matmul(dev(g1[f"matmul_{tag}_a"]), dev(g1[f"matmul_{tag}_b"])))
        np.testing.assert_allclose(c, g1[f"matmul_{tag}_c"], rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.matmul(dev(np.zeros((2, 3))), dev(np.zeros((4, 5))))


@pytest.mark.parametrize("m", [16, 100])
def test_w8a16_gemm_kn(m):
    rng = np.random.default_rng(9)
    K, N = 512, 384
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    codes, sbits = O.quantize_fp8_e4m3_block(w)  # [N,K], scale [N/128,K/128]
    a = rng.standard_normal((m, K)).astype(np.float32)
    ref = O.bf16_round(a) @ O.dequantize_fp8_e4m3_block(codes, sbits).T
    b_kn = from_numpy(np.ascontiguousarray(codes.T))
    s_kn = from_numpy(np.ascontiguousarray(sbits.T))
    c = host(ops.w8a16_gemm_sm120(dev(a, "bfloat16"), b_kn, s_kn))
    assert rel_err(c, ref) < 1e-2
    # layout-equivalence the reference relies on (linear.py:173-179): GEMV on [N,K] == GEMM on [K,N]
    c2 = host(ops.gemv_fp8_bf16_batched(dev(a[:8], "bfloat16"), from_numpy(codes), from_numpy(sbits)))
    assert rel_err(c[:8], c2) < 1e-2


# ----------------------------------------------------------------------------- 256 x 256 LDS-DMA GEMM
@pytest.mark.parametrize("shape", [(256, 256, 64), (512, 768, 256), (300, 520, 192), (1000, 260, 1024), (300, 576, 192), (700, 192, 320)])
def test_gemm256_bf16_matches_oracle_and_128_tile_kernel(shape, monkeypatch):
    """The large-tile kernel (LDS-DMA staging, source-side swizzle) against the oracle, on shapes with ragged M / N
    edges and several K tiles; the 128-tile kernel on the same inputs must agree to bf16 rounding of the same sums.
    N = 768 / 576 / 192 take the 192-column tile variant (a W stage of 12 DMA instructions split 2 + 1 over the wave halves)."""
    M, N, K = shape
    rng = np.random.default_rng(31)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    w[:, 0] += np.arange(N) % 7 * 0.01       # asymmetric: a transposed store cannot pass
    bias = rng.standard_normal(N).astype(np.float32)
    ref = O.bf16_round(a) @ O.bf16_round(w).T + O.bf16_round(bias)
    ad, wd, bd = dev(a, "bfloat16"), dev(w, "bfloat16"), dev(bias, "bfloat16")
    monkeypatch.setenv("PGK_GEMM256", "1")
    c256 = host(ops.matmul_nt(ad, wd, bd))
    monkeypatch.setenv("PGK_GEMM256", "0")
    c128 = host(ops.matmul_nt(ad, wd, bd))
    assert rel_err(c256, ref) < 1e-2 and rel_err(c128, ref) < 1e-2
    assert rel_err(c256, c128) < 3e-3


def test_w8a16_gemm_large_goes_through_dequant_and_gemm256(monkeypatch):
    """Large w8a16 products dequantise the weight once and use the LDS-DMA bf16 kernel; result equals the 128-tile
    in-staging-dequant kernel on the same inputs up to accumulation order."""
    rng = np.random.default_rng(34)
    M, N, K = 512, 768, 512
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    codes, sbits = O.quantize_fp8_e4m3_block(w)
    a = rng.standard_normal((M, K)).astype(np.float32)
    ref = O.bf16_round(a) @ O.dequantize_fp8_e4m3_block(codes, sbits).T
    args = [dev(a, "bfloat16"), from_numpy(codes), from_numpy(sbits)]
    monkeypatch.setenv("PGK_GEMM256", "1")
    c256 = host(ops.w8a16_gemm_nk(*args))
    monkeypatch.setenv("PGK_GEMM256", "0")
    c128 = host(ops.w8a16_gemm_nk(*args))
    assert rel_err(c256, ref) < 1e-2 and rel_err(c128, ref) < 1e-2 and rel_err(c256, c128) < 3e-3


@pytest.mark.parametrize("N", [512, 384])
def test_gemm256_exact_integers(N, monkeypatch):
    """Small integers are exact in bf16 and fp32: the two kernels and the oracle must agree bit for bit (N = 384: 192-column tiles)."""
    rng = np.random.default_rng(32)
    M, K = 512, 128
    a = rng.integers(-3, 4, (M, K)).astype(np.float32)
    w = rng.integers(-3, 4, (N, K)).astype(np.float32)
    monkeypatch.setenv("PGK_GEMM256", "1")
    c = host(ops.matmul_nt(dev(a, "bfloat16"), dev(w, "bfloat16")))
    np.testing.assert_array_equal(c, O.bf16_round(a @ w.T))


# ----------------------------------------------------------------------------- fp8 x fp8 GEMM
def _fp8_vals(codes):
    return O.fp8_e4m3_table()[codes]


@pytest.mark.parametrize("dt", ["bfloat16", "float32"])
def test_quantize_fp8_rows_matches_oracle(dt):
    rng = np.random.default_rng(21)
    M, K = 37, 512
    x = (rng.standard_normal((M, K)) * rng.uniform(0.01, 30.0, (M, 1))).astype(np.float32)
    x[3, 128:256] = 0.0            # an all-zero block: scale 1, codes 0
    x[5, 7] = 1e-9                 # far below the block's smallest subnormal step
    x = rounded(x, dt)
    codes, scale = ops.quantize_fp8_rows(dev(x, dt))
    rc, rs = O.quantize_fp8_rows(x)
    np.testing.assert_array_equal(scale.to_numpy(), rs)          # fp32 absmax/448: bit-exact
    np.testing.assert_array_equal(_fp8_vals(codes.to_numpy()), _fp8_vals(rc))   # same e4m3 value everywhere
    assert np.all(codes.to_numpy()[3, 128:256] == 0) and scale.to_numpy()[3, 1] == 1.0


def test_quantize_fp8_blocks_matches_oracle():
    rng = np.random.default_rng(22)
    N, K = 200, 384               # ragged last row-block
    w = O.bf16_round((rng.standard_normal((N, K)) * 0.02).astype(np.float32))
    codes, scale = ops.quantize_fp8_blocks(dev(w, "bfloat16"))
    wp = np.zeros((256, K), np.float32)
    wp[:N] = w
    rc, rs = O.quantize_fp8_e4m3_block(wp)
    np.testing.assert_array_equal(scale.to_numpy(), rs)
    np.testing.assert_array_equal(_fp8_vals(codes.to_numpy()), _fp8_vals(rc[:N]))


@pytest.mark.parametrize("shape", [(1, 128, 128), (100, 384, 512), (256, 256, 1024), (130, 200, 256)])
def test_gemm_fp8_blockwise(shape):
    """Same codes and scales on both sides: what is left is fp32 accumulation order (bar 1e-3; bf16 output rounding
    alone is 2e-3 relative per element, ~1e-3 in L2)."""
    M, N, K = shape
    rng = np.random.default_rng(23)
    a = (rng.standard_normal((M, K)) * rng.uniform(0.1, 4.0, (M, 1))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    a8, sa = O.quantize_fp8_rows(a)
    npad = (N + 127) // 128 * 128
    wp = np.zeros((npad, K), np.float32)
    wp[:N] = w
    w8, sw = O.quantize_fp8_e4m3_block(wp)
    w8 = np.ascontiguousarray(w8[:N])
    ref = O.gemm_fp8_blockwise(a8, sa, w8, sw)
    c = host(ops.gemm_fp8_fp8_blockwise_nt(from_numpy(a8), from_numpy(w8), from_numpy(sa), from_numpy(sw)))
    assert rel_err(c, ref) < 3e-3
    # and against the unquantised product: the fp8 bar of BASELINE.json
    assert rel_err(c, a @ w.T) < 5e-2


@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 520, 384), (1000, 260, 1024)])
def test_gemm256_fp8_matches_oracle_and_128_tile_kernel(shape, monkeypatch):
    """fp8 x fp8 on the 256-tile LDS-DMA structure (operand tiles AND scales arrive by DMA) vs the oracle and vs the
    128-tile kernel, on ragged M / N (N = 260, 520: a last 128-column scale block that is only partly there)."""
    M, N, K = shape
    rng = np.random.default_rng(33)
    a = (rng.standard_normal((M, K)) * rng.uniform(0.1, 4.0, (M, 1))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.02 * rng.uniform(0.5, 2.0, (N, 1))).astype(np.float32)
    a8, sa = O.quantize_fp8_rows(a)
    npad = (N + 127) // 128 * 128
    wp = np.zeros((npad, K), np.float32)
    wp[:N] = w
    w8, sw = O.quantize_fp8_e4m3_block(wp)
    w8 = np.ascontiguousarray(w8[:N])
    ref = O.gemm_fp8_blockwise(a8, sa, w8, sw)
    args = [from_numpy(a8), from_numpy(w8), from_numpy(sa), from_numpy(sw)]
    monkeypatch.setenv("PGK_GEMM256", "1")
    c256 = host(ops.gemm_fp8_fp8_blockwise_nt(*args))
    monkeypatch.setenv("PGK_GEMM256", "0")
    c128 = host(ops.gemm_fp8_fp8_blockwise_nt(*args))
    assert rel_err(c256, ref) < 3e-3 and rel_err(c128, ref) < 3e-3
    assert rel_err(c256, c128) < 3e-3


CONFIG5_GEMMS = [(4096, 6144, 4096), (4096, 28672, 4096), (4096, 4096, 14336)]     # qkv, gate_up, down of Llama-3-8B at S = 4096


@pytest.mark.parametrize("shape", CONFIG5_GEMMS)
def test_config5_shape_gemms_fp8a8_and_w8a16_vs_oracle(shape, monkeypatch):
    """BASELINE config 5's projection GEMMs at their REAL shapes (M = 4096 rows of Llama-3-8B), both the 256-tile LDS-DMA
    kernels and the 128-tile ones (PGK_GEMM256 = 1 / 0), fp8 x fp8 (activations quantised per row and 128 k) and w8a16:
      * against the oracle evaluated on the SAME codes and scales (what is left is accumulation order and the bf16 store):
        3e-3 for fp8 x fp8, 1e-2 for w8a16 (bf16 activations against fp32 ones in the oracle);
      * against the product of the unquantised operands: BASELINE's 5e-2 fp8 bar, per GEMM.
    Operands are quantised on the device (value-identical to the oracle's quantisers: test_quantize_fp8_* above); the
    oracle is evaluated on every 16th output row (a 4096 x 28672 x 4096 product in float64 is not a unit test)."""
    M, N, K = shape
    rng = np.random.default_rng(41)
    a = (rng.standard_normal((M, K), dtype=np.float32) * rng.uniform(0.2, 3.0, (M, 1)).astype(np.float32))
    w = rng.standard_normal((N, K), dtype=np.float32) * np.float32(0.02)
    a16, w16 = dev(a, "bfloat16"), dev(w, "bfloat16")
    a = O.bf16_round(a)
    rows = np.arange(0, M, 16)
    exact = a[rows] @ O.bf16_round(w).T                                   # the unquantised product (bf16-valued operands)
    del w
    w8, sw = ops.quantize_fp8_blocks(w16)
    a8, sa = ops.quantize_fp8_rows(a16)
    wdq = O.dequantize_fp8_e4m3_block(w8.to_numpy(), sw.to_numpy())     # [N, K] fp32: the values both GEMMs multiply
    tab = O.fp8_e4m3_table()
    adq = tab[a8.to_numpy()[rows]].astype(np.float32) * np.repeat(sa.to_numpy()[rows], 128, axis=1)
    ref_a8 = adq @ wdq.T
    ref_w8 = a[rows] @ wdq.T
    for tile in ("1", "0"):
        monkeypatch.setenv("PGK_GEMM256", tile)
        c = host(ops.gemm_fp8_fp8_blockwise_nt(a8, w8, sa, sw))
        assert np.isfinite(c).all()
        assert rel_err(c[rows], ref_a8) < 3e-3, (tile, rel_err(c[rows], ref_a8))
        assert rel_err(c[rows], exact) < 5e-2, (tile, rel_err(c[rows], exact))
        c = host(ops.w8a16_gemm_nk(a16, w8, sw))
        assert np.isfinite(c).all()
        assert rel_err(c[rows], ref_w8) < 1e-2, (tile, rel_err(c[rows], ref_w8))
        assert rel_err(c[rows], exact) < 5e-2, (tile, rel_err(c[rows], exact))


def test_gemm_fp8_exact_integers_and_asymmetric_operand():
    """Small-integer operands with unit scales are exact in e4m3 and in fp32: the result must be bit-exact, which
    pins the A/B lane->k pairing and the C row/col map (an asymmetric W catches a transposed store)."""
    rng = np.random.default_rng(24)
    M, N, K = 48, 160, 256
    table = O.fp8_e4m3_table()
    ints = {float(v): c for c, v in enumerate(table[:0x7F]) if v == np.floor(v) and v <= 8}
    ai = rng.integers(-4, 5, (M, K)).astype(np.float32)
    wi = rng.integers(-4, 5, (N, K)).astype(np.float32)
    wi[:, 0] = np.arange(N) % 5       # asymmetric
    enc = lambda x: (np.vectorize(lambda v: ints[abs(float(v))])(x).astype(np.uint8) | np.where(x < 0, 0x80, 0).astype(np.uint8))
    a8, w8 = enc(ai), enc(wi)
    sa = np.ones((M, K // 128), np.float32)
    sw = np.full((2, K // 128), 0x3F80, np.uint16)
    for force in ("0", "1"):      # 128-tile kernel, then the 256-tile LDS-DMA kernel
        os.environ["PGK_GEMM256"] = force
        try:
            c = host(ops.gemm_fp8_fp8_blockwise_nt(from_numpy(a8), from_numpy(w8), from_numpy(sa), from_numpy(sw)))
        finally:
            del os.environ["PGK_GEMM256"]
        np.testing.assert_array_equal(c, O.bf16_round(ai @ wi.T))


def test_matmul_fp8_auto_quantise():
    rng = np.random.default_rng(25)
    M, K, N = 64, 256, 192
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    c = ops.matmul_fp8(from_numpy(a), from_numpy(b))
    assert c.dtype == float32 and c.shape == (M, N)
    assert rel_err(c.to_numpy(), a @ b) < 5e-2
    with pytest.raises(ValueError):
        ops.matmul_fp8(from_numpy(a), from_numpy(b[:100]))


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("tag", ["off", "sq", "dec"])
def test_sdpa_golden_fp32(tag):
    o = host(ops.sdpa_causal(dev(g1[f"sdpa_{tag}_q"]), dev(g1[f"sdpa_{tag}_k"]), dev(g1[f"sdpa_{tag}_v"])))
    np.testing.assert_allclose(o, g1[f"sdpa_{tag}_o"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("cfg", [(4, 4, 9, 9, 128), (4, 2, 3, 11, 64), (16, 8, 128, 128, 128), (8, 2, 70, 200, 128),
                                 (2, 2, 130, 130, 64), (4, 1, 1, 37, 128),
                                 # kv_len <= 128 at head_dim 128 (bf16): the one-tile kernel, ragged rows and a kv offset
                                 (8, 2, 70, 100, 128), (2, 1, 33, 128, 128), (4, 4, 17, 17, 128),
                                 # q_len > 128: the transposed-score kernel (ops_flash.hip), incl. ragged tiles and kv offset
                                 (4, 2, 129, 129, 128), (8, 2, 300, 300, 128), (2, 1, 257, 400, 128), (4, 4, 513, 513, 64),
                                 (2, 2, 200, 1000, 64)])
def test_sdpa_flash_prefill(dt, cfg):
    hq, hkv, ql, kl, d = cfg
    rng = np.random.default_rng(10)
    q, k, v = (rng.standard_normal(s).astype(np.float32) for s in ((hq, ql, d), (hkv, kl, d), (hkv, kl, d)))
    qr, kr, vr = rounded(q, dt), rounded(k, dt), rounded(v, dt)
    rep = hq // hkv
    ref = O.sdpa_causal(qr, np.repeat(kr, rep, axis=0), np.repeat(vr, rep, axis=0))
    o = host(ops.sdpa_causal(dev(q, dt), dev(k, dt), dev(v, dt)))
    close(o, ref, dt)
    assert np.abs(o - ref).max() < 5e-2  # reference's FA2-vs-SDPA bar (benchmarks/test_flash_attention.py:169)


@pytest.mark.parametrize("ql", [128, 192])   # 128: first-generation kernel, 192: transposed-score kernel
def test_sdpa_softmax_max_jump(ql):
    """Force the online-softmax rescale: one key far above the rest late in the sequence."""
    rng = np.random.default_rng(11)
    hq, d = 2, 128
    q, k, v = (rng.standard_normal((hq, ql, d)).astype(np.float32) for _ in range(3))
    k[:, ql - 42] = q[:, ql - 32] * 3.0  # huge score for late queries, in the last kv tile
    ref = O.sdpa_causal(O.bf16_round(q), O.bf16_round(k), O.bf16_round(v))
    o = host(ops.sdpa_causal(dev(q, "bfloat16"), dev(k, "bfloat16"), dev(v, "bfloat16")))
    close(o, ref, "bfloat16")


def test_sdpa_flash_long_sequence_properties():
    """S = 2048 (BASELINE config-5 scale per head is 4096): size-independent checks of the long path -
    row 0 attends to key 0 only (out == V[0]); constant V is returned unchanged (softmax weights sum to 1);
    and a strided sample of rows matches the oracle computed for those rows alone."""
    rng = np.random.default_rng(15)
    hq, hkv, S, d = 4, 2, 2048, 128
    q, k = rng.standard_normal((hq, S, d)).astype(np.float32), rng.standard_normal((hkv, S, d)).astype(np.float32)
    v = rng.standard_normal((hkv, S, d)).astype(np.float32)
    qd, kd = dev(q, "bfloat16"), dev(k, "bfloat16")
    o = host(ops.sdpa_causal(qd, kd, dev(v, "bfloat16")))
    vr = O.bf16_round(v)
    np.testing.assert_array_equal(o[:, 0], np.repeat(vr[:, 0], hq // hkv, axis=0))
    oc = host(ops.sdpa_causal(qd, kd, dev(np.full_like(v, 0.75), "bfloat16")))
    np.testing.assert_allclose(oc, 0.75, atol=4e-3)
    rows = np.array([1, 63, 64, 127, 128, 1000, 1023, 1024, 2047])
    qr, kr = O.bf16_round(q), O.bf16_round(k)
    for h in range(hq):
        kh, vh = kr[h // (hq // hkv)], vr[h // (hq // hkv)]
        sc = (qr[h, rows] @ kh.T) / np.sqrt(d)
        sc[np.arange(S)[None, :] > rows[:, None]] = -np.inf
        p = np.exp(sc - sc.max(axis=1, keepdims=True))
        ref = (p / p.sum(axis=1, keepdims=True)) @ vh
        assert rel_err(o[h, rows], ref) < 1e-2


@pytest.mark.parametrize("dt", ["bfloat16", "float32"])
@pytest.mark.parametrize("cfg", [(16, 8, 128, 2112, 2048), (16, 16, 128, 512, 100), (8, 2, 64, 300, 299), (4, 4, 128, 64, 1)])
def test_sdpa_fixed_cache_decode(dt, cfg):
    hq, hc, d, max_seq, ctx = cfg
    rng = np.random.default_rng(12)
    q = rng.standard_normal((hq, 1, d)).astype(np.float32)
    kc = rng.standard_normal((hc, max_seq, d)).astype(np.float32)
    vc = rng.standard_normal((hc, max_seq, d)).astype(np.float32)
    kc[:, ctx:], vc[:, ctx:] = 1e4, 1e4  # garbage beyond the context must never be read into the result
    ref = O.sdpa_causal_fixed_cache(rounded(q, dt), rounded(kc, dt), rounded(vc, dt), ctx)
    out = pk.empty((hq, 1, d), dt)
    ops.sdpa_causal_fixed_cache(dev(q, dt), dev(kc, dt), dev(vc, dt), out, ctx)
    close(host(out), ref, dt)
    assert np.abs(host(out) - ref).max() < 1e-2  # reference's flash-decoding bar (benchmarks/test_flash_decoding.py:76)
    out2 = pk.empty((hq, 1, d), dt)
    ops.sdpa_causal_fixed_cache_ptr(dev(q, dt), dev(kc, dt), dev(vc, dt), out2, from_numpy(np.array([ctx], np.int32)), max_seq)
    np.testing.assert_array_equal(out2.to_numpy(), out.to_numpy())


def test_reference_flash_switches_select_other_kernels_with_the_same_result(monkeypatch):
    """PYGPUKIT_FLASH_ATTENTION / PYGPUKIT_FLASH_DECODING are the reference's A/B switches (sdpa_causal.inl:380-447): with
    them off the one-workgroup-per-row fallback resp. the general SDPA path run instead of the MFMA flash / split-KV
    kernels; results agree to the bf16 bar either way."""
    rng = np.random.default_rng(14)
    hq, hkv, s_len, d = 4, 2, 200, 128
    q = rng.standard_normal((hq, s_len, d)).astype(np.float32)
    k, v = (rng.standard_normal((hkv, s_len, d)).astype(np.float32) for _ in range(2))
    ref = O.sdpa_causal(O.bf16_round(q), np.repeat(O.bf16_round(k), hq // hkv, axis=0), np.repeat(O.bf16_round(v), hq // hkv, axis=0), 0.0)
    outs = {}
    for mode in ("auto", "0", "1"):
        monkeypatch.setenv("PYGPUKIT_FLASH_ATTENTION", mode)
        outs[mode] = host(ops.sdpa_causal(dev(q, "bfloat16"), dev(np.repeat(k, hq // hkv, axis=0), "bfloat16"),
                                          dev(np.repeat(v, hq // hkv, axis=0), "bfloat16")))
        close(outs[mode], ref, "bfloat16")
    np.testing.assert_array_equal(outs["auto"], outs["1"])
    assert not np.array_equal(outs["auto"], outs["0"])          # another kernel, another rounding order
    monkeypatch.delenv("PYGPUKIT_FLASH_ATTENTION")
    max_seq, ctx = 256, 131
    q1 = rng.standard_normal((hq, 1, d)).astype(np.float32)
    kc, vc = (rng.standard_normal((hkv, max_seq, d)).astype(np.float32) for _ in range(2))
    ref1 = O.sdpa_causal_fixed_cache(O.bf16_round(q1), O.bf16_round(kc), O.bf16_round(vc), ctx)
    dec = {}
    for mode in ("-1", "0", "1"):
        monkeypatch.setenv("PYGPUKIT_FLASH_DECODING", mode)
        out = pk.empty((hq, 1, d), "bfloat16")
        ops.sdpa_causal_fixed_cache(dev(q1, "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"), out, ctx)
        dec[mode] = host(out)
        close(dec[mode], ref1, "bfloat16")
    np.testing.assert_array_equal(dec["-1"], dec["1"])
    # a device-resident context length has only the split-KV kernel: "0" must not break it
    out = pk.empty((hq, 1, d), "bfloat16")
    ops.sdpa_causal_fixed_cache_ptr(dev(q1, "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"), out, from_numpy(np.array([ctx], np.int32)), max_seq)
    np.testing.assert_array_equal(host(out), dec["1"])


def test_sdpa_fixed_cache_multi_query():
    rng = np.random.default_rng(13)
    hq, hc, d, max_seq, ctx, ql = 4, 2, 128, 96, 50, 5
    q = rng.standard_normal((hq, ql, d)).astype(np.float32)
    kc, vc = (rng.standard_normal((hc, max_seq, d)).astype(np.float32) for _ in range(2))
    ref = O.sdpa_causal_fixed_cache(O.bf16_round(q), O.bf16_round(kc), O.bf16_round(vc), ctx)
    out = pk.empty((hq, ql, d), "bfloat16")
    ops.sdpa_causal_fixed_cache(dev(q, "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"), out, ctx)
    close(host(out), ref, "bfloat16")
    with pytest.raises(ValueError):
        ops.sdpa_causal_fixed_cache(dev(q, "bfloat16"), dev(kc, "bfloat16"), dev(vc, "bfloat16"), out, max_seq + 1)


# ----------------------------------------------------------------------------- paged KV cache / continuous batching
def _paged_setup(rng, num_seqs, hkv, bs, d, ctxs, dt, num_blocks=None):
    max_blocks = max((c + bs - 1) // bs for c in ctxs) + 1
    num_blocks = num_blocks or num_seqs * max_blocks + 3
    perm = rng.permutation(num_blocks)                      # scattered physical pages
    tables = np.zeros((num_seqs, max_blocks), np.int32)
    nxt = 0
    for s_ in range(num_seqs):
        for b in range((ctxs[s_] + bs - 1) // bs):
            tables[s_, b] = perm[nxt]
            nxt += 1
    kc = rounded(rng.standard_normal((num_blocks, hkv, bs, d)).astype(np.float32), dt)
    vc = rounded(rng.standard_normal((num_blocks, hkv, bs, d)).astype(np.float32), dt)
    return tables, kc, vc


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
@pytest.mark.parametrize("cfg", [(3, 16, 8, 128, 16, [37, 1, 160]), (2, 8, 8, 64, 8, [17, 64]), (1, 4, 1, 128, 32, [700]),
                                 (2, 16, 2, 128, 16, [2048, 1300]), (5, 2, 2, 64, 4, [3, 9, 1, 12, 7])])
def test_paged_attention_v1(dt, cfg):
    """Split-KV flash-decoding over scattered pages vs the oracle; long contexts exercise the multi-slice merge, the
    last page of every sequence is partial, and unused table entries point at page 0 (never read)."""
    num_seqs, hq, hkv, d, bs, ctxs = cfg
    rng = np.random.default_rng(51)
    tables, kc, vc = _paged_setup(rng, num_seqs, hkv, bs, d, ctxs, dt)
    q = rounded(rng.standard_normal((num_seqs, hq, d)).astype(np.float32), dt)
    ref = O.paged_attention_v1(q, kc, vc, tables, np.array(ctxs, np.int32))
    o = host(ops.paged_attention_v1(dev(q, dt), dev(kc, dt), dev(vc, dt), from_numpy(tables), from_numpy(np.array(ctxs, np.int32)),
                                    max_context=max(ctxs)))
    close(o, ref, dt)
    assert np.abs(o - ref).max() < 1e-2


def test_paged_cache_write_and_attention_round_trip():
    """reshape_and_cache a prefill, copy_to_paged_cache one decode row, then attend: equals attention over the dense rows."""
    rng = np.random.default_rng(52)
    hq, hkv, d, bs, n = 8, 4, 128, 16, 45
    num_blocks = 8
    k = O.bf16_round(rng.standard_normal((n + 1, hkv, d)).astype(np.float32))
    v = O.bf16_round(rng.standard_normal((n + 1, hkv, d)).astype(np.float32))
    pages = [5, 2, 7]
    slots = np.array([pages[t // bs] * bs + t % bs for t in range(n + 1)], np.int32)
    kc, vc = ops.allocate_kv_cache(num_blocks, hkv, bs, d, bfloat16), ops.allocate_kv_cache(num_blocks, hkv, bs, d, bfloat16)
    slot_pf = slots[:n].copy()
    slot_pf[10] = -1                                      # a padding token: skipped
    ops.reshape_and_cache(dev(k[:n], "bfloat16"), dev(v[:n], "bfloat16"), kc, vc, from_numpy(slot_pf))
    ops.copy_to_paged_cache(dev(k[n:], "bfloat16"), dev(v[n:], "bfloat16"), kc, vc, from_numpy(slots[n:]))
    rk, rv = np.zeros((num_blocks, hkv, bs, d), np.float32), np.zeros((num_blocks, hkv, bs, d), np.float32)
    O.paged_cache_write(k[:n], v[:n], rk, rv, slot_pf)
    O.paged_cache_write(k[n:], v[n:], rk, rv, slots[n:])
    np.testing.assert_array_equal(host(kc), rk)
    np.testing.assert_array_equal(host(vc), rv)
    q = O.bf16_round(rng.standard_normal((1, hq, d)).astype(np.float32))
    tables = np.array([pages + [0]], np.int32)
    o = host(ops.paged_attention_v1(dev(q, "bfloat16"), kc, vc, from_numpy(tables), from_numpy(np.array([n + 1], np.int32))))
    kd, vd = k.copy(), v.copy()
    kd[10], vd[10] = 0, 0                                 # the skipped slot stayed zero
    dense = O.sdpa_causal_fixed_cache(q.transpose(1, 0, 2), np.repeat(kd.transpose(1, 0, 2), hq // hkv, axis=0),
                                      np.repeat(vd.transpose(1, 0, 2), hq // hkv, axis=0), n + 1)
    close(o[0], dense[:, 0], "bfloat16")


def test_continuous_batching_helpers():
    rng = np.random.default_rng(53)
    lists = [[5, 9, 2], [7], [1, 1, 4, 8, 3]]
    ids, total = ops.prepare_batch_inputs(lists)
    assert total == 9 and ids.to_numpy().tolist() == [5, 9, 2, 7, 1, 1, 4, 8, 3]
    lens = from_numpy(np.array([3, 1, 5], np.int32))
    starts = ops.compute_cumsum(lens)
    assert starts.to_numpy().tolist() == [0, 3, 4]
    E = rng.standard_normal((12, 64)).astype(np.float32)
    g = ops.gather_embeddings(ids, dev(E, "bfloat16"), total)
    np.testing.assert_array_equal(host(g), O.bf16_round(E)[ids.to_numpy()])
    ctx = from_numpy(np.array([0, 17, 0], np.int32))
    pf = from_numpy(np.array([1, 0, 1], np.int32))
    pos = ops.prepare_position_ids(starts, ctx, pf, lens, 3, total)
    np.testing.assert_array_equal(pos.to_numpy(), O.prepare_position_ids([0, 3, 4], [0, 17, 0], [1, 0, 1], [3, 1, 5], total))
    logits = rng.standard_normal((total, 1000)).astype(np.float32)
    last = ops.scatter_last_token_logits(dev(logits, "float16"), starts, lens, 3, 1000)
    np.testing.assert_array_equal(host(last), rounded(logits, "float16")[[2, 3, 8]])
    toks = ops.argmax_sample(last, 3, 1000)
    np.testing.assert_array_equal(toks.to_numpy(), np.argmax(rounded(logits, "float16")[[2, 3, 8]], axis=1))
    fin = ops.check_eos(toks, int(toks.to_numpy()[1]))
    assert fin.to_numpy()[1] == 1 and fin.to_numpy().sum() >= 1


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
def test_transposes_4d_3d_and_sequence_major_kv_cache(dt):
    """transpose_3d_012 / 4d_0132 / 4d_0213 (tensor.py:191-380) and the un-expanded, sequence-major cache writers
    kv_cache_update / kv_cache_prefill (embedding.py:78-125): pure data movement, exact against NumPy."""
    rng = np.random.default_rng(61)
    x3 = rng.standard_normal((3, 70, 45)).astype(np.float32)
    x4 = rng.standard_normal((2, 5, 66, 24)).astype(np.float32)
    np.testing.assert_array_equal(host(ops.transpose_3d_012(dev(x3, dt))), rounded(x3, dt).transpose(0, 2, 1))
    np.testing.assert_array_equal(host(ops.transpose_4d_0132(dev(x4, dt))), rounded(x4, dt).transpose(0, 1, 3, 2))
    np.testing.assert_array_equal(host(ops.transpose_4d_0213(dev(x4, dt))), rounded(x4, dt).transpose(0, 2, 1, 3))
    out = pk.empty((2, 66, 5, 24), dt)
    assert ops.transpose_4d_0213(dev(x4, dt), out=out) is out
    np.testing.assert_array_equal(host(out), rounded(x4, dt).transpose(0, 2, 1, 3))
    with pytest.raises(ValueError):
        ops.transpose_4d_0213(dev(x3, dt))
    cache = pk.zeros((16, 4, 32), dt)
    new = rng.standard_normal((6, 4, 32)).astype(np.float32)
    one = rng.standard_normal((1, 4, 32)).astype(np.float32)
    ops.kv_cache_prefill(dev(new, dt), cache, 3)
    ops.kv_cache_update(dev(one, dt), cache, 9)
    want = np.zeros((16, 4, 32), np.float32)
    want[3:9], want[9] = rounded(new, dt), rounded(one, dt)[0]
    np.testing.assert_array_equal(host(cache), want)
    with pytest.raises(ValueError):
        ops.kv_cache_update(dev(one, dt), cache, 16)
    with pytest.raises(ValueError):
        ops.kv_cache_prefill(dev(new, dt), cache, 11)


# ----------------------------------------------------------------------------- device sampling
def _safe_us(lg, T, k, p, rng, n=6):
    """u values whose decision is at least 1e-4 of the kept mass away from a boundary (expf vs np.exp differ by ulps)."""
    us = []
    while len(us) < n:
        u = float(np.float32(rng.random()))
        _, margin = O.sample_token_u(lg, T, k, p, u, return_margin=True)
        if margin > 1e-4:
            us.append(u)
    return us


@pytest.mark.parametrize("dt", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("params", [(1.0, 0, 1.0), (0.7, 40, 1.0), (1.3, 0, 0.9), (0.8, 50, 0.95), (1.0, 1, 1.0), (2.0, 1000, 0.5)])
def test_sample_token_gpu_matches_oracle(dt, params):
    """Index-exact against the oracle's restatement for given u (multinomial, top-k, nucleus, both), on a peaked
    LLM-like row of Qwen3's vocabulary size."""
    T, k, p = params
    rng = np.random.default_rng(41)
    V = 151936
    lg = rounded((rng.standard_normal(V) * 2.5).astype(np.float32), dt)
    lg[rng.integers(0, V, 20)] += rounded(rng.uniform(4, 9, 20).astype(np.float32), dt)
    lg = rounded(lg, dt)
    d = dev(lg, dt)
    for u in _safe_us(lg, T, k, p, rng) + [0.0]:
        assert ops.sample_token_gpu(d, T, k, p, u=u) == O.sample_token_u(lg, T, k, p, u)


@pytest.mark.parametrize("V", [4095, 4096, 4097, 8192 + 5, 40000])
@pytest.mark.parametrize("k", [1, 7, 1024, 1025])
def test_sample_sliced_topk_boundaries(V, k):
    """1 <= top_k <= 1024 runs the sliced two-stage kernels (4096-token slices), above that the whole-row kernel:
    slice edges, a ragged last slice, heavy ties (a quantised row: ~60 distinct values) and several rows per call
    all give the oracle's token."""
    if k >= V:
        pytest.skip("top_k >= vocab keeps everything")
    rng = np.random.default_rng(V * 31 + k)
    rows = 3
    lg = (np.round(rng.standard_normal((rows, V)) * 8) / 4).astype(np.float32)        # many exact ties
    lg[:, V - 1] = lg.max() + 0.25                                                     # the last token is in every top-k
    d = dev(lg)
    res = pk.empty((rows,), "int32")
    for p_ in (1.0, 0.8):
        for _ in range(3):
            u = float(np.float32(rng.random()))
            ub = from_numpy(np.array([u], np.float32))
            if p_ == 1.0:
                ops.sample_topk_to_buf_ptr(d, res, ub, k, 1.0)      # all rows in one launch, u from device memory
            for r in range(rows):
                tok, margin = O.sample_token_u(lg[r], 1.0, k, p_, u, return_margin=True)
                if margin <= 1e-4:
                    continue
                if p_ == 1.0:
                    assert res.to_numpy()[r] == tok, (V, k, r, u)
                assert ops.sample_token_gpu(dev(lg[r]), 1.0, k, p_, u=u) == tok, (V, k, p_, r, u)


def test_sample_whole_vocabulary_sliced_paths():
    """top_k = 0 on a long row runs sliced: multinomial picks the owning 4096-token slice; a nucleus is cut inside the
    best 256 keys when they carry top_p of the mass (a peaked, LLM-like row) and falls through to the whole-row kernel
    when they do not (a flat row).  Three rows in one call - peaked, flat, heavy ties - so both outcomes of the device-side
    decision happen in the same launch; every token is the oracle's."""
    import ctypes as C

    from pygpukit_amd import _hip

    rng = np.random.default_rng(91)
    V = 50000 + 37
    peaked = (rng.standard_normal(V) * 1.5).astype(np.float32)
    peaked[rng.integers(0, V, 40)] += rng.uniform(9, 14, 40).astype(np.float32)
    flat = (rng.standard_normal(V) * 0.5).astype(np.float32)
    ties = (np.round(rng.standard_normal(V) * 4) / 2).astype(np.float32)
    lg = np.stack([peaked, flat, ties])
    d, res = dev(lg), pk.empty((3,), "int32")
    for T, p_ in ((1.0, 1.0), (0.8, 0.9), (1.0, 0.5), (1.2, 0.999)):
        for _ in range(4):
            u = float(np.float32(rng.random()))
            _hip.call("pgk_sample_token", d._p, 3, V, d.dtype.code, C.c_float(T), 0, C.c_float(p_), C.c_float(u), None, res._p, None)
            got = res.to_numpy()
            for r in range(3):
                tok, margin = O.sample_token_u(lg[r], T, 0, p_, u, return_margin=True)
                assert margin <= 1e-4 or got[r] == tok, (T, p_, r, u, got[r], tok)
    # the peaked row's nucleus really is inside its 256 best tokens, the flat row's is not (both branches ran)
    for row, inside in ((peaked, True), (flat, False)):
        z = np.sort(row.astype(np.float64) / 0.8)[::-1]
        pr = np.exp(z - z[0])
        assert (pr[:256].sum() >= 0.9 * pr.sum()) == inside
    for u in (0.0, 1.0):
        assert ops.sample_multinomial(dev(peaked), 1.0, u=u) == O.sample_token_u(peaked, 1.0, 0, 1.0, u)


def test_sample_ties_and_small_vocab():
    """Ties at the top-k / nucleus boundary are kept lowest-index-first; tiny rows (V < threads) work."""
    lg = np.array([1.0, 3.0, 3.0, 3.0, 0.5, 3.0, -2.0], np.float32)
    d = dev(lg)
    for k, p in ((2, 1.0), (3, 1.0), (0, 0.5), (0, 0.3), (4, 0.6)):
        for u in (0.0, 0.2, 0.45, 0.7, 0.99):
            _, margin = O.sample_token_u(lg, 1.0, k, p, u, return_margin=True)
            if margin > 1e-4:
                assert ops.sample_token_gpu(d, 1.0, k, p, u=u) == O.sample_token_u(lg, 1.0, k, p, u), (k, p, u)
    assert ops.sample_topk(d, 1, 1.0, u=0.77) == 1          # top-1 == lowest-index maximum
    assert ops.sample_token_gpu(d, 0.0) == 1                # temperature 0 -> greedy


def test_sample_distribution_and_seed():
    """With the host generator seeded, draws are reproducible and follow softmax(logits / T) restricted to top-k."""
    lg = np.array([2.0, 1.0, 0.0, -1.0, 3.0, 0.5], np.float32)
    d = dev(lg)
    ops.set_sampling_seed(7)
    a = [ops.sample_topk(d, 3, 1.0) for _ in range(400)]
    ops.set_sampling_seed(7)
    b = [ops.sample_topk(d, 3, 1.0) for _ in range(400)]
    assert a == b and set(a) <= {0, 1, 4}
    pr = np.exp(lg[[0, 1, 4]])
    pr /= pr.sum()
    freq = np.array([a.count(0), a.count(1), a.count(4)]) / 400
    assert np.abs(freq - pr).max() < 0.08


def test_sample_topk_to_buf_ptr_in_graph():
    """The random number is read from device memory at replay time: one captured launch, different u -> different tokens."""
    rng = np.random.default_rng(42)
    lg = (rng.standard_normal((2, 5000)) * 3).astype(np.float32)
    d = dev(lg, "float16")
    res, ub = pk.empty((2,), "int32"), from_numpy(np.array([0.0], np.float32))
    graph = pk.CudaGraph()
    graph.begin_capture()
    ops.sample_topk_to_buf_ptr(d, res, ub, 20, 0.9)
    graph.end_capture()
    lgr = rounded(lg, "float16")
    for u in (0.1, 0.5, 0.93):
        ub.copy_from_numpy(np.array([u], np.float32))
        graph.replay()
        graph.synchronize()
        got = res.to_numpy()
        for r in range(2):
            tok, margin = O.sample_token_u(lgr[r], 0.9, 20, 1.0, u, return_margin=True)
            assert margin < 1e-4 or got[r] == tok


# ----------------------------------------------------------------------------- graph
def test_graph_capture_replay_reads_device_scalars():
    """Capture embedding_lookup_ptr + rmsnorm once; replay with a different token id in the device buffer."""
    rng = np.random.default_rng(14)
    E = rng.standard_normal((20, 256)).astype(np.float32)
    Ed, g = dev(E, "bfloat16"), dev(np.ones(256, np.float32), "bfloat16")
    tok = from_numpy(np.array([3], np.int32))
    x, y = pk.empty((1, 256), "bfloat16"), pk.empty((1, 256), "bfloat16")
    graph = pk.CudaGraph()
    graph.begin_capture()
    ops.embedding_lookup_ptr(Ed, x, tok)
    ops.rmsnorm(x, g, 1e-6, out=y)
    graph.end_capture()
    assert graph.is_ready() and graph.num_nodes >= 2
    for t in (3, 11, 19):
        tok.copy_from_numpy(np.array([t], np.int32))
        graph.replay()
        graph.synchronize()
        close(host(y), O.rmsnorm(O.bf16_round(E[t : t + 1]), np.ones(256, np.float32), 1e-6), "bfloat16")
