"""First execution of csrc/comm.hip on hardware: a world-size-1 RCCL communicator on the one GPU of the test box, driven
through the same RcclComm / ControlPlane objects bench.py uses for N > 1 (weights broadcast from rank 0,
token log all-gathered after the timed steps).  With one rank every collective is the identity, which is exactly what
can be checked against NumPy."""

from __future__ import annotations

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pk = pytest.importorskip("pygpukit_amd")
from pygpukit_amd.core import from_numpy  # noqa: E402
from pygpukit_amd.core.array import GPUArray  # noqa: E402
from pygpukit_amd.core.dtypes import int32  # noqa: E402


def test_world_size_1_rccl_broadcast_all_gather_barrier(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    from pygpukit_amd.parallel import ControlPlane, RcclComm

    cp = ControlPlane()
    assert (cp.rank, cp.world) == (0, 1)
    comm = RcclComm(cp)                                   # ncclGetUniqueId + ncclCommInitRank
    try:
        rng = np.random.default_rng(0)
        # broadcast: the weight arena's path (bf16 bits as uint16, odd byte count, then 64 MiB)
        for n in (1, 12345, 32 * 1024 * 1024):
            host = rng.integers(0, 65535, n, dtype=np.uint16)
            dev = from_numpy(host)
            comm.broadcast(dev, 0)
            comm.barrier()
            np.testing.assert_array_equal(dev.to_numpy(), host)
        # all-gather: the token log's path
        log = rng.integers(0, 151936, (64, 8), dtype=np.int32)
        send = from_numpy(log)
        recv = GPUArray((1,) + log.shape, int32)
        comm.all_gather(send, recv)
        comm.barrier()
        np.testing.assert_array_equal(recv.to_numpy()[0], log)
        # errors surface as exceptions, not as silent no-ops
        with pytest.raises(RuntimeError):
            comm.broadcast(send, 5)                       # root outside the communicator
    finally:
        comm.destroy()
    cp.shutdown()


def test_data_parallel_decoder_on_the_gpu_engine_single_rank():
    """The N = 1 point of config 4 through the harness the N > 1 ranks use: DataParallelDecoder sharding (one shard),
    DecodeBatch on the native engine as the per-shard runner, tokens equal to the engine driven directly."""
    from oracle import cpu_ref as O
    from pygpukit_amd import llm
    from pygpukit_amd.llm import synthetic as S
    from pygpukit_amd.parallel import ControlPlane, DataParallelDecoder
    from tests.golden_cfg import TINY

    w = O.make_qwen3_weights(TINY, seed=40, bf16=True)
    rng = np.random.default_rng(9)
    prompts = [[int(t) for t in rng.integers(0, 1024, n)] for n in (3, 5, 2, 7, 4, 9)]
    model = S.build_model_from_weights(TINY, w, dtype="bfloat16", max_pos=64)

    def runner(ps, n_steps):
        strat = llm.DecodeBatch(batch_size=len(ps))
        strat.bind(model)
        strat.init_graph(max_seq_len=64)
        first = strat.prefill(ps)
        rest = strat.run_greedy(first, [len(p) for p in ps], n_steps - 1)
        return np.concatenate([first[None, :], rest], axis=0)

    toks = DataParallelDecoder(ControlPlane(), runner).decode(prompts, 4)
    ref = O.build_qwen3_ref(TINY, w, max_pos=64)
    for b, p in enumerate(prompts):
        want, lgs = ref.generate(p, max_new_tokens=4, temperature=0.0, top_k=0, top_p=1.0, return_logits=True)
        s = np.sort(np.stack(lgs), axis=1)
        if np.all((s[:, -1] - s[:, -2]) > 0.03 * np.abs(np.stack(lgs)).max(axis=1)):
            assert [int(t) for t in toks[:, b]] == want[len(p):], b
