"""In-kernel phase stamps of the batch-1 decode step (diagnostic build of the library):
    make -C pygpukit_amd/csrc OUT=$PWD/tools/micro/libpgk_stamps.so BUILD=/tmp/build_stamps EXTRA=-DPGK_PHASE_STAMPS
    PGK_LIB=$PWD/tools/micro/libpgk_stamps.so python tools/phase_stamps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pygpukit_amd.llm import synthetic as S
cfg = dict(S.QWEN3_0_6B, num_layers=8)
w = S.make_qwen3_weights(cfg, seed=0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 150
fmt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
eng = S.build_engine_from_weights(cfg, w, max_seq_len=512 if P < 400 else P + 64, max_batch=1, weight_format=fmt)
pr = [int(t) for t in np.random.default_rng(1).integers(0, cfg["vocab_size"], P)]
first = int(np.argmax(eng.prefill(pr)))
eng.set_state([first], [P])
print("context", P)
tl = eng.timeline(1, warm=3)
for t in tl[4:12]:
    print("%-9s wgs %4d  start %7.2f .. %7.2f   end %7.2f .. %7.2f  (span %.2f)" % (t["kernel"], t["workgroups"], t["first_start_us"], t["last_start_us"], t["first_end_us"], t["last_end_us"], t["last_end_us"] - t["first_start_us"]))
