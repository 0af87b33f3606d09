"""Per-sequence error distribution of a 64-sequence decode step against the oracle (2 layers at Qwen3-0.6B widths, the
setting of tests/test_gpu_model.py::test_m_tiled_batches_at_qwen3_widths_vs_oracle): tools/err_dist.py [bf16|fp8] [B]
Prints mean / median / p95 / max of the relative L2 error of the logits per sequence, for the prefill's last row and for
the teacher-forced decode step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import cpu_ref as O
from pygpukit_amd.llm import synthetic as S

fmt = sys.argv[1] if len(sys.argv) > 1 else "fp8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = dict(S.QWEN3_0_6B, num_layers=2, vocab_size=4096)
w = S.make_qwen3_weights(cfg, seed=31)
wref = w
if fmt == "fp8":
    wref = {"embed": w["embed"], "final_norm": w["final_norm"], "layers": []}
    for lw in w["layers"]:
        d = dict(lw)
        for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
            fused = np.concatenate([lw[n] for n in names], axis=0)
            deq = (O.dequantize_fp8_e4m3_block_w8a16_gemm if os.environ.get('ORACLE_W8A16_GEMM', '1') == '1' else O.dequantize_fp8_e4m3_block)(*O.quantize_fp8_e4m3_block(fused))
            r = 0
            for n in names:
                d[n] = deq[r:r + lw[n].shape[0]]
                r += lw[n].shape[0]
        wref["layers"].append(d)
ref = O.build_qwen3_ref(cfg, wref, max_pos=64)
rng = np.random.default_rng(32)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
prompts = [[int(t) for t in rng.integers(0, cfg["vocab_size"], int(rng.integers(1, 40)))] for _ in range(B)]
want = [ref.generate(p, max_new_tokens=2, temperature=0.0, top_k=0, top_p=1.0, return_logits=True) for p in prompts]
eng = S.build_engine_from_weights(cfg, w, max_seq_len=64, max_batch=B, weight_format=fmt)
pre = [eng.prefill(p, seq=b).copy() for b, p in enumerate(prompts)]
eng.set_state([wt[0][len(p)] for wt, p in zip(want, prompts)], [len(p) for p in prompts])
eng.decode_step(B)
eng.synchronize()
got = eng.logits(B).to_numpy()
for name, errs in (("prefill last row", [rel(pre[b], want[b][1][0]) for b in range(B)]), ("decode step", [rel(got[b], want[b][1][1]) for b in range(B)])):
    e = np.sort(np.array(errs))
    print(f"{fmt} B={B} {name}: mean {e.mean():.5f} median {np.median(e):.5f} p95 {e[int(0.95 * (B - 1))]:.5f} max {e.max():.5f}  (prompt length of the max: {len(prompts[int(np.argmax(errs))])})")
lens = np.array([len(p) for p in prompts])
errs = np.array([rel(got[b], want[b][1][1]) for b in range(B)])
for lo, hi in ((1, 4), (4, 10), (10, 20), (20, 40)):
    m = (lens >= lo) & (lens < hi)
    if m.any():
        print(f"  prompts of {lo}..{hi - 1} tokens: n={int(m.sum())} mean {errs[m].mean():.5f} max {errs[m].max():.5f}")
