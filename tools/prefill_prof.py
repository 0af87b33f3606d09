"""Run the Qwen3-0.6B prefill (S tokens) a few times: target for rocprofv3 --kernel-trace --stats.
usage: prefill_prof.py [S=128] [reps=10] [bf16|fp8]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd.llm import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(S.QWEN3_0_6B)
w = S.make_qwen3_weights(cfg, seed=0)
fmt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
eng = S.build_engine_from_weights(cfg, w, max_seq_len=max(256, n + 8), max_batch=1, weight_format=fmt)
prompt = [int(t) for t in np.random.default_rng(1).integers(0, cfg['vocab_size'], n)]
import time
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for _ in range(3):
    eng.prefill(prompt, want_last_logits=False)
eng.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.prefill(prompt, want_last_logits=False)
eng.synchronize()
print(f"prefill S={n} {fmt}: {(time.perf_counter() - t0) * 1e3 / reps:.3f} ms")
