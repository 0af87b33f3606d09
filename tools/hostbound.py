import time, numpy as np, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpukit_amd.llm import synthetic as S
from pygpukit_amd import _hip
cfg=dict(S.QWEN3_0_6B)
w=S.make_qwen3_weights(cfg,seed=0)
eng=S.build_engine_from_weights(cfg,w,max_seq_len=400,max_batch=1)
prompt=[int(t) for t in np.random.default_rng(1).integers(0,cfg['vocab_size'],128)]
first=int(np.argmax(eng.prefill(prompt)))
eng.set_state([first],[128]); eng.capture(1)
eng.replay(8); eng.synchronize()
for n in (1,64,200):
    t0=time.perf_counter(); eng.replay(n); t1=time.perf_counter(); eng.synchronize(); t2=time.perf_counter()
    print(f"n={n}: enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
# eager
eng.set_state([first],[128])
t0=time.perf_counter()
for _ in range(20): eng.decode_step(1)
t1=time.perf_counter(); eng.synchronize(); t2=time.perf_counter()
print(f"eager: enqueue {1e6*(t1-t0)/20:.1f} us/step total {1e6*(t2-t0)/20:.1f}")
eng.set_state([first],[128]); eng.reset_log(); eng.replay(200); eng.synchronize()
print("shader clock MHz during graph decode:", eng.shader_clock_mhz(200))
