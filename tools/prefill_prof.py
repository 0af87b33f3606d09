"""Run the Qwen3-0.6B prefill (S tokens) a few times: target for rocprofv3 --kernel-trace --stats."""
import sys, numpy as np
sys.path.insert(0, '.')
from pygpukit_amd.llm import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(S.QWEN3_0_6B)
w = S.make_qwen3_weights(cfg, seed=0)
eng = S.build_engine_from_weights(cfg, w, max_seq_len=max(256, n + 8), max_batch=1)
prompt = [int(t) for t in np.random.default_rng(1).integers(0, cfg['vocab_size'], n)]
for _ in range(10):
    eng.prefill(prompt, want_last_logits=False)
eng.synchronize()
