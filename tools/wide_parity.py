"""Llama-3-8B WIDTHS (H=4096, I=14336, 32/8 heads) with 2 layers and a small vocabulary: GPU engine (bf16, w8a16, fp8a8
prefill) against the CPU oracle.  Validates the K = 4096 / 14336 GEMM paths end to end and shows how much of the
fp8-vs-bf16 gap at this width is the quantisation itself (oracle vs oracle)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import cpu_ref as O
from pygpukit_amd.llm import synthetic as S
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cfg = dict(O.LLAMA3_8B, num_layers=L, vocab_size=4096)
re = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
t0 = time.perf_counter()
w = O.make_qwen3_weights(cfg, seed=3, bf16=True)
for lw in w["layers"]:
    del lw["q_norm"], lw["k_norm"]
prompt = [int(t) for t in np.random.default_rng(4).integers(0, cfg["vocab_size"], n)]
print(f"weights {time.perf_counter() - t0:.0f}s", flush=True)
g = {f: S.build_engine_from_weights(cfg, w, max_seq_len=n + 8, max_batch=1, weight_format=f).prefill(prompt).copy() for f in ("bf16", "fp8", "fp8a8")}
print(f"gpu done {time.perf_counter() - t0:.0f}s", flush=True)
rb = O.build_qwen3_ref(cfg, w, max_pos=n + 8); hb, _ = rb(prompt); lb = rb.get_logits(hb)[-1]
print("gpu bf16  vs oracle bf16 :", re(g["bf16"], lb), flush=True)
wq = {"embed": w["embed"], "final_norm": w["final_norm"], "layers": []}
for lw in w["layers"]:
    d = dict(lw)
    for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
        fused = np.concatenate([lw[k] for k in names], axis=0)
        deq = O.dequantize_fp8_e4m3_block(*O.quantize_fp8_e4m3_block(fused)); r = 0
        for k in names:
            d[k] = deq[r:r + lw[k].shape[0]]; r += lw[k].shape[0]
    wq["layers"].append(d)
r16 = O.build_qwen3_ref(cfg, wq, max_pos=n + 8); h16, _ = r16(prompt); l16 = r16.get_logits(h16)[-1]
print("gpu w8a16 vs oracle w8a16:", re(g["fp8"], l16), flush=True)
print("oracle w8a16 vs oracle bf16:", re(l16, lb), "   gpu w8a16 vs gpu bf16:", re(g["fp8"], g["bf16"]), flush=True)
r8 = O.build_qwen3_ref_fp8a8(cfg, w, max_pos=n + 8); h8, _ = r8(prompt); l8 = r8.get_logits(h8)[-1]
print("gpu fp8a8 vs oracle fp8a8:", re(g["fp8a8"], l8), "  oracle fp8a8 vs oracle w8a16:", re(l8, l16), "  gpu fp8a8 vs oracle w8a16:", re(g["fp8a8"], l16), flush=True)
print(f"total {time.perf_counter() - t0:.0f}s")
