"""JIT (hiprtc) front end: the reference's tests/test_jit.py cases (availability, version, jit() surface, error
structure) run on the CPU - hiprtc compiles for gfx950 without a GPU - and the launch path runs under -m gpu."""

from __future__ import annotations

import numpy as np
import pytest

import importlib

J = importlib.import_module("pygpukit_amd.jit")   # `pygpukit_amd.jit` the attribute is the jit() function, as in the reference

SCALE_SRC = '''
extern "C" __global__ void scale(float* x, float factor, int n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) x[idx] *= factor;
}
'''

needs_rtc = pytest.mark.skipif(not J.is_nvrtc_available(), reason="libhiprtc not loadable")


def test_availability_and_version_surface():
    assert isinstance(J.is_nvrtc_available(), bool)
    v = J.get_nvrtc_version()
    if J.is_nvrtc_available():
        assert isinstance(v, tuple) and len(v) == 2 and v[0] > 0
        p = J.get_nvrtc_path()
        assert isinstance(p, str) and "hiprtc" in p and J.get_nvrtc_path() == p
        ok, msg = J.check_driver_compatibility()
        assert ok and "hiprtc" in msg
    else:
        assert v is None and J.get_nvrtc_path() is None
    assert J.is_hiprtc_available is J.is_nvrtc_available and J.HiprtcError is J.NvrtcError
    assert J.get_driver_requirements()["target"] == "gfx950"


@needs_rtc
def test_jit_creates_kernel_with_reference_surface():
    k = J.jit(SCALE_SRC, func="scale")
    assert isinstance(k, J.JITKernel) and k.name == "scale" and k.source == SCALE_SRC
    assert k.is_compiled and callable(k) and k.block_size == 256
    assert "scale" in repr(k) and "compiled" in repr(k)
    assert isinstance(k.ptx, bytes) and k.ptx[:4] == b"\x7fELF"          # a gfx950 code object where the reference has PTX
    k2 = J.jit(SCALE_SRC, func="scale", options=["-O3", "-arch=sm_80"], block_size=128)   # NVRTC arch flags are dropped
    assert k2.options == ["-O3", "-arch=sm_80"] and k2.block_size == 128 and k2.is_compiled
    assert k._compute_cache_key() != k2._compute_cache_key()


@needs_rtc
def test_invalid_function_name_and_compile_error_are_structured():
    with pytest.raises(ValueError, match="not found in source"):
        J.jit(SCALE_SRC, func="nonexistent")
    with pytest.raises(J.NvrtcError) as ei:
        J.jit('extern "C" __global__ void bad(float* x) { x[0] = undefined_symbol; }', func="bad")
    assert ei.value.code == J.NvrtcErrorCode.Compilation
    assert "undefined_symbol" in ei.value.compilation_log and "[Compilation]" in str(ei.value)
    out = J.compile_to_ptx(SCALE_SRC)
    assert out.ptx[:4] == b"\x7fELF" and isinstance(out.log, str)


@needs_rtc
def test_warmup_api():
    assert J.warmup() is True and J.is_warmup_done() and J.get_warmup_error() is None
    hit = []
    assert J.warmup(callback=lambda: hit.append(1)) and hit == [1]


@needs_rtc
@pytest.mark.gpu
def test_jit_kernel_launch_and_results():
    pk = pytest.importorskip("pygpukit_amd")
    from pygpukit_amd.core import from_numpy

    x = np.arange(1000, dtype=np.float32)
    d = from_numpy(x)
    k = J.jit(SCALE_SRC, func="scale")
    k(d, 0.5, 1000)                                  # grid from the first array argument
    np.testing.assert_array_equal(d.to_numpy(), x * 0.5)
    k(d, np.float32(4.0), np.int32(10), grid_size=1, block_size=64)   # only the first 10 elements
    want = x * 0.5
    want[:10] *= 4.0
    np.testing.assert_array_equal(d.to_numpy(), want)
    assert 64 <= k.get_suggested_block_size() <= 1024
    # LDS, a 2-D grid and 64-lane wave intrinsics: row sums of a [rows, 256] matrix
    src = '''
    extern "C" __global__ void row_sum(const float* a, float* out, int cols) {
        __shared__ float part[4];
        const float* row = a + (size_t)blockIdx.y * cols;
        float s = 0.f;
        for (int i = threadIdx.x; i < cols; i += blockDim.x) s += row[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[blockIdx.y] = part[0] + part[1] + part[2] + part[3];
    }
    '''
    a = np.random.default_rng(0).integers(-8, 9, (37, 256)).astype(np.float32)
    o = pk.zeros((37,), "float32")
    J.jit(src, "row_sum")(from_numpy(a), o, 256, grid_size=(1, 37), block_size=256)
    np.testing.assert_array_equal(o.to_numpy(), a.sum(axis=1))
    # a kernel that is not extern "C" has a mangled name: structured FunctionNotFound
    with pytest.raises(J.NvrtcError) as ei:
        J.jit("__global__ void mangled(float* x) { x[0] = 1.f; }", "mangled")(d, grid_size=1)
    assert ei.value.code == J.NvrtcErrorCode.FunctionNotFound
