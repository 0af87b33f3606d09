import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import cpu_ref as O
from pygpukit_amd.llm import synthetic as S
hkv = int(sys.argv[1]); B = int(sys.argv[2])
cfg = dict(S.QWEN3_0_6B, num_layers=2, vocab_size=4096, num_kv_heads=hkv)
w = S.make_qwen3_weights(cfg, seed=61)
ref = O.build_qwen3_ref(cfg, w, max_pos=512)
toks = [int(t) for t in np.random.default_rng(62).integers(0, cfg["vocab_size"], 200)]
hidden, _ = ref(toks)
want = np.asarray(ref.get_logits(hidden)).reshape(len(toks), -1)
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
eng = S.build_engine_from_weights(cfg, w, max_seq_len=512, max_batch=B)
out = []
for p in (17, 150):
    for b in range(B):
        eng.prefill(toks[:p], seq=b, want_last_logits=False)
    eng.set_state([toks[p]] * B, [p] * B)
    eng.decode_step(B)
    eng.synchronize()
    out.append("%d:%.4f" % (p, max(rel(eng.logits(B).to_numpy()[b], want[p]) for b in range(B))))
print("hkv", hkv, "B", B, " ".join(out), "launches", eng.launches_per_step())
