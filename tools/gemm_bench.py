"""Time the C-ABI GEMMs at prefill shapes: TFLOP/s against the dense MFMA peak.
usage: gemm_bench.py {bf16|w8a16|fp8} M N K [M N K ...]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd import _hip

PGK_BF16 = 3


def dev(nbytes, fill=None):
    p = C.c_void_p()
    _hip.call("pgk_malloc", C.byref(p), nbytes)
    if fill is not None:
        _hip.call("pgk_memcpy_h2d", p, fill.ctypes.data_as(C.c_void_p), fill.nbytes, None)
    return p


def main():
    kind = sys.argv[1]
    dims = [int(x) for x in sys.argv[2:]]
    _hip.require_device()
    rng = np.random.default_rng(0)
    import os
    fill = os.environ.get("FILL", "normal")     # operand statistics change the clock the chip holds (DVFS): normal | uniform | zeros
    gen = rng

    class _R:   # same call sites, different operand statistics
        def standard_normal(self, shape):
            return gen.uniform(-1.0, 1.0, shape) if fill == "uniform" else (np.zeros(shape) if fill == "zeros" else gen.standard_normal(shape))

        def integers(self, *a, **k):
            return gen.integers(*a, **k)

    rng = _R()
    for i in range(0, len(dims), 3):
        M, N, K = dims[i:i + 3]
        a16 = (rng.standard_normal((M, K)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        c = dev(M * N * 2)
        if kind == "bf16":
            w16 = (rng.standard_normal((N, K)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            a, w = dev(a16.nbytes, a16), dev(w16.nbytes, w16)
            run = lambda: _hip.call("pgk_gemm_nt", a, w, None, c, M, N, K, PGK_BF16, None)
        else:
            w8 = rng.integers(0, 0x7E, (N, K), dtype=np.uint8)
            sc = np.full((N // 128, K // 128), 0x3C00, np.uint16)  # bf16 2^-7
            w, s = dev(w8.nbytes, w8), dev(sc.nbytes, sc)
            if kind == "w8a16":
                a = dev(a16.nbytes, a16)
                run = lambda: _hip.call("pgk_w8a16_gemm_nk", a, w, s, c, M, N, K, None)
            else:
                a8 = rng.integers(0, 0x7E, (M, K), dtype=np.uint8)
                asc = np.ones((M, K // 128), np.float32)
                a, sa = dev(a8.nbytes, a8), dev(asc.nbytes, asc)
                run = lambda: _hip.call("pgk_gemm_fp8_nt", a, sa, w, s, c, M, N, K, None)
        e0, e1 = C.c_void_p(), C.c_void_p()
        _hip.call("pgk_event_create", C.byref(e0)); _hip.call("pgk_event_create", C.byref(e1))
        for _ in range(5): run()
        reps = 20
        _hip.call("pgk_event_record", e0, None)
        for _ in range(reps): run()
        _hip.call("pgk_event_record", e1, None)
        _hip.call("pgk_event_sync", e1)
        ms = C.c_float()
        _hip.call("pgk_event_elapsed_ms", e0, e1, C.byref(ms))
        us = ms.value * 1e3 / reps
        print(f"{kind} M={M} N={N} K={K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


main()
