"""In-kernel phase stamps of the packed-weight prefill GEMMs (diagnostic build):
    make -C pygpukit_amd/csrc OUT=$PWD/tools/micro/libpgk_stamps.so BUILD=/tmp/build_stamps EXTRA=-DPGK_PHASE_STAMPS
    PGK_LIB=$PWD/tools/micro/libpgk_stamps.so python tools/pk_stamps.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pygpukit_amd.llm import synthetic as S
from pygpukit_amd import _hip
cfg = dict(S.QWEN3_0_6B, num_layers=4)
w = S.make_qwen3_weights(cfg, seed=0)
eng = S.build_engine_from_weights(cfg, w, max_seq_len=256, max_batch=1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pr = [int(t) for t in np.random.default_rng(1).integers(0, cfg["vocab_size"], n)]
for _ in range(3):
    eng.prefill(pr, want_last_logits=False)
eng.synchronize()
buf = np.zeros((5, 512, 8), np.uint64)
lib = ctypes.CDLL(os.environ["PGK_LIB"])
lib.pgk_debug_pk_stamps(buf.ctypes.data_as(ctypes.c_void_p))
names = {1: "slab (o/down)", 3: "swiglu (gate_up)", 4: "qkv heads"}
for epi, nm in names.items():
    st = buf[epi].astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    if not live.any():
        print(f"{nm}: not launched (the N = hidden projections carry the norm: pkgemm_resid_kernel)")
        continue
    t0 = st[:, 0].min()
    print(f"{nm}: {live.sum()} workgroups; start spread {(st[:, 0].max() - t0) / 100:.2f} us")
    for i, lab in enumerate(["start", "loads issued", "A landed + barrier", "first 16 k-steps", "k loop done", "end"]):
        col = st[:, i]
        if (col > 0).all():
            print(f"   {lab:22s} mean {(col - st[:, 0]).mean() / 100:6.2f} us after own start, last WG at {(col.max() - t0) / 100:6.2f} us")
