"""A/B the staggered vs lockstep 256x256 bf16 GEMM inside the Llama-8B-shape bf16 prefill, interleaved in one process."""
import os, sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd.llm import synthetic as S
from pygpukit_amd.llm.engine import Engine
L, n = 8, 4096
cfg = dict(S.LLAMA3_8B, num_layers=L)
w = S.random_engine_weights(cfg, seed=0, fp8=False, keep_bf16=True, threads=12)
eng = Engine(cfg, w["embed"], w["bf16"], w["final_norm"], None, max_seq_len=n + 8, max_batch=1, weight_format="bf16", use_qk_norm=False)
prompt = [int(t) for t in np.random.default_rng(1).integers(0, cfg["vocab_size"], n)]
eng.prefill(prompt, want_last_logits=False); eng.synchronize()
res = {"0": [], "1": []}
for rep in range(4):
    for mode in ("1", "0"):
        os.environ["PGK_GEMM256S"] = mode
        t = time.perf_counter(); eng.prefill(prompt, want_last_logits=False); eng.synchronize(); res[mode].append((time.perf_counter() - t) * 1e3)
print("staggered ms:", [round(x, 2) for x in res["1"]], " lockstep ms:", [round(x, 2) for x in res["0"]])
