"""Timing ablation of the decode step (diagnostic; results are numerically meaningless).
PGK_DEBUG_SKIP bits: 1 no activation loads, 2 no weight loads, 4 no cross-lane reduction,
8 no prologue block reduction, 16 no attention kernel."""
import os, subprocess, sys, json
code = r'''
import time, numpy as np, sys, os
sys.path.insert(0,'.')
from pygpukit_amd.llm import synthetic as S
cfg=dict(S.QWEN3_0_6B)
w=S.make_qwen3_weights(cfg,seed=0)
eng=S.build_engine_from_weights(cfg,w,max_seq_len=400,max_batch=1)
prompt=[int(t) for t in np.random.default_rng(1).integers(0,cfg['vocab_size'],128)]
eng.prefill(prompt); eng.set_state([5],[128]); eng.capture(1); eng.replay(8); eng.synchronize()
t0=time.perf_counter(); eng.replay(100); eng.synchronize(); t1=time.perf_counter()
print("RESULT", os.environ.get("PGK_DEBUG_SKIP","0"), os.environ.get("PGK_FUSED_ATTN","1"), round(1e6*(t1-t0)/100,1), eng.launches_per_step())
'''
for fused in ("1", "0"):
    for mask in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31):
        env = dict(os.environ, PGK_DEBUG_SKIP=str(mask), PGK_FUSED_ATTN=fused)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print([l for l in out.stdout.splitlines() if l.startswith("RESULT")] or out.stderr[-300:], flush=True)
