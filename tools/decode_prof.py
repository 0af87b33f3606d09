"""Decode-step workload for rocprofv3: tools/decode_prof.py <batch> [steps] [prompt_len] [bf16|fp8] [eager|graph] [max_seq_len]
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/decode_prof.py 64 20"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pygpukit_amd.llm import synthetic as S
B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
fmt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = dict(S.QWEN3_0_6B)
w = S.make_qwen3_weights(cfg, seed=0)
cap = int(sys.argv[6]) if len(sys.argv) > 6 else P + steps + 16      # KV-cache rows per sequence (the launch sequence follows the context, not this)
eng = S.build_engine_from_weights(cfg, w, max_seq_len=cap, max_batch=B, weight_format=fmt)
pr = np.random.default_rng(1).integers(0, cfg["vocab_size"], (B, P))
first = [int(np.argmax(eng.prefill([int(t) for t in pr[b]], seq=b))) for b in range(B)]
eng.set_state(first, [P] * B)
eager = len(sys.argv) > 5 and sys.argv[5] == "eager"   # un-captured steps (per-launch timing without a graph; an earlier form of the packed-weight step crashed rocprofv3 inside hipGraphLaunch)
if eager:
    def run(n):
        for _ in range(n):
            eng.decode_step(B)
else:
    eng.capture(B)
    run = eng.replay
run(4); eng.synchronize()
t0 = time.perf_counter(); run(steps); eng.synchronize(); dt = time.perf_counter() - t0
print(f"batch {B} ctx {P} cache {cap} {fmt}: {dt * 1e3 / steps:.3f} ms/step, {B * steps / dt:.0f} tok/s, {eng.launches_per_step()} launches/step")
