"""Time pgk_sdpa_causal (prefill) : TFLOP/s = 4 S^2 D Hq / 2 per launch.  usage: attn_bench.py Hq Hkv S D [...]"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd import _hip
_hip.require_device()
a = [int(x) for x in sys.argv[1:]]
rng = np.random.default_rng(0)
def dev(arr):
    p = C.c_void_p(); _hip.call("pgk_malloc", C.byref(p), arr.nbytes)
    _hip.call("pgk_memcpy_h2d", p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, None); return p
for i in range(0, len(a), 4):
    hq, hkv, S, D = a[i:i + 4]
    bf = lambda shape: (rng.standard_normal(shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    q, k, v = dev(bf((S, hq, D))), dev(bf((hkv, S, D))), dev(bf((hkv, S, D)))
    o = C.c_void_p(); _hip.call("pgk_malloc", C.byref(o), S * hq * D * 2)
    # q/out in the projection's [S, H, D] layout, K/V in the cache layout [Hkv, S, D]
    run = lambda: _hip.call("pgk_sdpa_causal", q, k, v, o, hq, hkv, S, S, D, C.c_float(0.0), D, hq * D, S * D, D, D, hq * D, 3, None)
    e0, e1 = C.c_void_p(), C.c_void_p()
    _hip.call("pgk_event_create", C.byref(e0)); _hip.call("pgk_event_create", C.byref(e1))
    for _ in range(3): run()
    _hip.call("pgk_event_record", e0, None)
    for _ in range(10): run()
    _hip.call("pgk_event_record", e1, None); _hip.call("pgk_event_sync", e1)
    ms = C.c_float(); _hip.call("pgk_event_elapsed_ms", e0, e1, C.byref(ms))
    us = ms.value * 100
    print(f"Hq={hq} Hkv={hkv} S={S} D={D}: {us:9.1f} us  {4.0 * S * S * D * hq / 2 / us / 1e6:7.1f} TFLOP/s", flush=True)
