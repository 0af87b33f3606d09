"""Qwen3-0.6B bf16 batch-1 decode with IN-GRAPH sampling: ms per token for greedy and for temperature / top-k / top-p
draws (the sampler's launches are nodes of the step graph; uniforms come from the device ring).
usage: sampled_decode.py [steps=64]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd.llm import synthetic as S
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(S.QWEN3_0_6B)
w = S.make_qwen3_weights(cfg, seed=0)
prompt = [int(t) for t in np.random.default_rng(1).integers(0, cfg['vocab_size'], 128)]
for name, (T, k, p) in {"greedy": (0.0, 0, 1.0), "T=0.8 top-k 50 top-p 0.9": (0.8, 50, 0.9), "T=0.8 top-k 50": (0.8, 50, 1.0),
                        "T=1.0 multinomial": (1.0, 0, 1.0), "T=0.8 top-p 0.9": (0.8, 0, 0.9)}.items():
    eng = S.build_engine_from_weights(cfg, w, max_seq_len=512, max_batch=1)
    logits = eng.prefill(prompt)
    eng.set_state([int(np.argmax(logits))], [len(prompt)])
    if T > 0:
        eng.set_sampling(T, k, p, uniforms=np.random.default_rng(2).random((steps + 8, 1), dtype=np.float32))
    eng.capture(1)
    eng.replay(8); eng.synchronize()
    eng.set_state([int(np.argmax(logits))], [len(prompt)]); eng.reset_log()
    t = time.perf_counter(); eng.replay(steps); eng.synchronize(); dt = time.perf_counter() - t
    toks = eng.read_tokens(1, steps)[:, 0]
    print(f"{name:28s} {dt / steps * 1e3:7.3f} ms/token  {steps / dt:7.0f} tok/s   launches/step {eng.launches_per_step()}  distinct tokens {len(set(toks.tolist()))}", flush=True)
    del eng
