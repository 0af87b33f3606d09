"""Time pgk_sample_token on one fp32 logits row of Qwen3's vocabulary."""
import ctypes as C, sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd import _hip
_hip.require_device()
V = 151936
rng = np.random.default_rng(0)
flat = (rng.standard_normal(V) * 2.5).astype(np.float32)            # random-init-like: the 1024 best tokens carry ~half the mass
peaked = flat.copy(); peaked[rng.integers(0, V, 40)] += rng.uniform(10, 16, 40).astype(np.float32)   # LLM-like: a few tokens dominate
o = C.c_void_p(); _hip.call("pgk_malloc", C.byref(o), 64)
e0, e1 = C.c_void_p(), C.c_void_p()
_hip.call("pgk_event_create", C.byref(e0)); _hip.call("pgk_event_create", C.byref(e1))
cases = [("multinomial", flat, 0, 1.0), ("top-k 50", flat, 50, 1.0), ("top-k 50 + top-p 0.9", flat, 50, 0.9),
         ("top-p 0.9, peaked row", peaked, 0, 0.9), ("top-p 0.9, flat row (falls through)", flat, 0, 0.9), ("top-k 2000 (whole-row kernel)", flat, 2000, 1.0)]
p = C.c_void_p(); _hip.call("pgk_malloc", C.byref(p), flat.nbytes)
for name, lg, k, tp in cases:
    _hip.call("pgk_memcpy_h2d", p, lg.ctypes.data_as(C.c_void_p), lg.nbytes, None)
    run = lambda: _hip.call("pgk_sample_token", p, 1, V, 1, C.c_float(0.8), k, C.c_float(tp), C.c_float(0.37), None, o, None)
    for _ in range(3): run()
    _hip.call("pgk_event_record", e0, None)
    for _ in range(20): run()
    _hip.call("pgk_event_record", e1, None); _hip.call("pgk_event_sync", e1)
    ms = C.c_float(); _hip.call("pgk_event_elapsed_ms", e0, e1, C.byref(ms))
    print(f"{name:40s} {ms.value * 50:8.1f} us", flush=True)
