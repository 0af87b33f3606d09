"""Dual-chain step diagnostics on a small Qwen3-width model: tools/dual_debug.py [layers]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pygpukit_amd.llm import synthetic as S
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = dict(S.QWEN3_0_6B, num_layers=L, vocab_size=8192)
w = S.make_qwen3_weights(cfg, seed=3)
eng = S.build_engine_from_weights(cfg, w, max_seq_len=128, max_batch=1)
first = int(np.argmax(eng.prefill([1, 2, 3, 4, 5])))
eng.set_state([first], [5])
print("initial", eng.dep_state())
eng.decode_step(1); print("after eager step", eng.dep_state())
eng.capture(1)
t0 = time.perf_counter(); eng.replay(1); eng.synchronize(); print("replay 1: %.3f ms" % ((time.perf_counter() - t0) * 1e3), eng.dep_state())
t0 = time.perf_counter(); eng.replay(1); eng.synchronize(); print("replay 2: %.3f ms" % ((time.perf_counter() - t0) * 1e3), eng.dep_state())
t0 = time.perf_counter(); eng.replay(50); eng.synchronize(); print("50 replays: %.3f ms each" % ((time.perf_counter() - t0) * 1e3 / 50), eng.dep_state()["epoch"], eng.dep_state()["timed_out"])
for t in eng.timeline(1, warm=1):
    print("%-9s wgs %4d  start %8.2f .. %8.2f   end %8.2f .. %8.2f" % (t["kernel"], t["workgroups"], t["first_start_us"], t["last_start_us"], t["first_end_us"], t["last_end_us"]))
