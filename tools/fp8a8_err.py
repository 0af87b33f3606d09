"""Error budget of the fp8-activation prefill (weight_format fp8a8) on a Llama-style mini model."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import cpu_ref as O
from pygpukit_amd.llm import synthetic as S
from tests.conftest import rel_err
from tests.test_gpu_model import LLAMA_MINI as cfg
w = O.make_qwen3_weights(cfg, seed=11, bf16=True)
for lw in w["layers"]:
    del lw["q_norm"], lw["k_norm"]
prompt = [int(t) for t in np.random.default_rng(12).integers(0, cfg["vocab_size"], 200)]
got = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1, weight_format="fp8a8").prefill(prompt)
got16 = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1, weight_format="fp8").prefill(prompt)
ref = O.build_qwen3_ref_fp8a8(cfg, w, max_pos=256)
h, _ = ref(prompt); want = ref.get_logits(h)[-1]
wq = {"embed": w["embed"], "final_norm": w["final_norm"], "layers": []}
for lw in w["layers"]:
    d = dict(lw)
    for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
        fused = np.concatenate([lw[n] for n in names], axis=0)
        deq = O.dequantize_fp8_e4m3_block(*O.quantize_fp8_e4m3_block(fused)); r = 0
        for n in names:
            d[n] = deq[r:r + lw[n].shape[0]]; r += lw[n].shape[0]
    wq["layers"].append(d)
r16 = O.build_qwen3_ref(cfg, wq, max_pos=256); h16, _ = r16(prompt); want16 = r16.get_logits(h16)[-1]
rb = O.build_qwen3_ref(cfg, w, max_pos=256); hb, _ = rb(prompt); wantb = rb.get_logits(hb)[-1]
print("gpu fp8a8 vs oracle fp8a8 :", rel_err(got, want))
print("gpu fp8a8 vs oracle w8a16 :", rel_err(got, want16))
print("oracle fp8a8 vs oracle w8a16:", rel_err(want, want16))
print("gpu w8a16 vs oracle w8a16 :", rel_err(got16, want16))
print("oracle w8a16 vs oracle bf16:", rel_err(want16, wantb))
print("gpu fp8a8 vs oracle bf16  :", rel_err(got, wantb))
