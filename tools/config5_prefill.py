"""BASELINE config 5: Llama-3-8B-shape random-init weights, fp8 (e4m3) MFMA GEMM prefill of S tokens on one GPU.
Reports prefill time and TFLOP/s (projection GEMM FLOPs + causal attention, last-row lm_head) for the
fp8 x fp8 path ("fp8a8"), the w8a16 path ("fp8") and bf16, and the last-row logits differences between them.
usage: config5_prefill.py [S=4096] [layers=32] [reps=3]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd.llm import synthetic as S
from pygpukit_amd.llm.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = dict(S.LLAMA3_8B, num_layers=L)
t0 = time.perf_counter()
w = S.random_engine_weights(cfg, seed=0, fp8=True, keep_bf16=True, threads=12)
print(f"weights: {time.perf_counter() - t0:.1f} s", flush=True)
H, D, I, V = cfg["hidden_size"], cfg["head_dim"], cfg["intermediate_size"], cfg["vocab_size"]
per_layer = H * (cfg["num_heads"] + 2 * cfg["num_kv_heads"]) * D + cfg["num_heads"] * D * H + 3 * H * I
gemm_flop = 2.0 * n * L * per_layer + 2.0 * V * H
attn_flop = 4.0 * n * n * D * cfg["num_heads"] * L / 2
prompt = [int(t) for t in np.random.default_rng(1).integers(0, V, n)]
logits = {}
for fmt in ("fp8a8", "fp8", "bf16"):
    layers = w["bf16"] if fmt == "bf16" else w["fp8"]
    eng = Engine(cfg, w["embed"], layers, w["final_norm"], None, max_seq_len=n + 8, max_batch=1, weight_format=fmt, use_qk_norm=False)
    logits[fmt] = eng.prefill(prompt).copy()
    eng.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); eng.prefill(prompt, want_last_logits=False); eng.synchronize(); ts.append(time.perf_counter() - t)
    ms = min(ts) * 1e3
    print(f"{fmt:6s} S={n} L={L}: {ms:9.2f} ms  {(gemm_flop + attn_flop) / ms / 1e9:8.1f} TFLOP/s (GEMM {gemm_flop / 1e12:.2f} + attention {attn_flop / 1e12:.2f} TFLOP)", flush=True)
    del eng
re = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("logits rel-err  fp8a8 vs bf16: %.4f   fp8(w8a16) vs bf16: %.4f   fp8a8 vs fp8: %.4f" % (
    re(logits["fp8a8"], logits["bf16"]), re(logits["fp8"], logits["bf16"]), re(logits["fp8a8"], logits["fp8"])))
print("argmax", {k: int(np.argmax(v)) for k, v in logits.items()})
