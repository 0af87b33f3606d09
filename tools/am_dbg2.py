import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import cpu_ref as O
from pygpukit_amd.llm import synthetic as S
hkv = int(sys.argv[1])
cfg = dict(S.QWEN3_0_6B, num_layers=2, vocab_size=4096, num_kv_heads=hkv)
w = S.make_qwen3_weights(cfg, seed=61)
print({k: v.shape for k, v in w["layers"][0].items()})
ref = O.build_qwen3_ref(cfg, w, max_pos=512)
toks = [int(t) for t in np.random.default_rng(62).integers(0, cfg["vocab_size"], 40)]
hidden, _ = ref(toks)
want = np.asarray(ref.get_logits(hidden)).reshape(len(toks), -1)
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
eng = S.build_engine_from_weights(cfg, w, max_seq_len=512, max_batch=1)
for p in (1, 17, 40):
    print("prefill", p, rel(eng.prefill(toks[:p]), want[p - 1]))
