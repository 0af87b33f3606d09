"""world_size-2 tests of the data-parallel harness on CPU: control plane collectives (the product's TCP hub and the
gloo backend kept beside it), block sharding, and that sharded greedy decode of independent sequences reproduces the
single-process result token for token.  The per-shard runner here is the CPU oracle (test infrastructure standing in
for the GPU engine, which the same harness drives on the GPU box)."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_runner():
    from oracle import cpu_ref as O
    from tests.golden_cfg import TINY

    ref = O.build_qwen3_ref(TINY, O.make_qwen3_weights(TINY, seed=40, bf16=True), max_pos=64)

    def run(prompts, n_steps):
        cols = [ref.generate(p, max_new_tokens=n_steps, temperature=0.0, top_k=0, top_p=1.0)[len(p):] for p in prompts]
        return np.asarray(cols, np.int32).T.reshape(n_steps, len(prompts))

    return run


def _prompts():
    rng = np.random.default_rng(9)
    return [[int(t) for t in rng.integers(0, 1024, n)] for n in (3, 5, 2, 7, 4)]


def _worker(rank: int, world: int, port: int, out_dir: str, backend: str = "gloo") -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PGK_CP_DIR=out_dir)
    from pygpukit_amd.parallel import ControlPlane, DataParallelDecoder, shard_range

    cp = ControlPlane(backend)
    assert (cp.rank, cp.world) == (rank, world)
    cp.barrier()
    assert cp.max_over_ranks(10.0 + rank) == 10.0 + world - 1
    assert cp.sum_over_ranks(1.0) == float(world)
    assert cp.min_over_ranks(10.0 + rank) == 10.0
    assert cp.first_note(None if rank == 0 else f"note from {rank}") == "note from 1"
    assert cp.all_gather_object({"r": rank}) == [{"r": r} for r in range(world)]
    blob = bytes(range(128)) if rank == 0 else None
    assert cp.broadcast_bytes(blob, 128, 0) == bytes(range(128))
    g = cp.gather_int32(np.array([rank, rank * 10, 7], np.int32))
    np.testing.assert_array_equal(g, [[r, r * 10, 7] for r in range(world)])
    prompts = _prompts()
    toks = DataParallelDecoder(cp, _oracle_runner()).decode(prompts, 4)
    lo, hi = shard_range(len(prompts), rank, world)
    np.save(os.path.join(out_dir, f"tokens_{rank}.npy"), toks)
    np.save(os.path.join(out_dir, f"shard_{rank}.npy"), np.array([lo, hi]))
    cp.barrier()
    cp.shutdown()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("socket", 2), ("socket", 3)])
def test_data_parallel_decode_two_ranks(backend, world, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    single = _oracle_runner()(_prompts(), 4)
    shards = [tuple(np.load(tmp_path / f"shard_{r}.npy")) for r in range(world)]
    assert shards == ([(0, 3), (3, 5)] if world == 2 else [(0, 2), (2, 4), (4, 5)])
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"tokens_{r}.npy"), single)


def test_single_process_control_plane_needs_no_rendezvous():
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from pygpukit_amd.parallel import ControlPlane, DataParallelDecoder

    cp = ControlPlane()
    assert (cp.rank, cp.world) == (0, 1) and cp.max_over_ranks(3.5) == 3.5
    cp.barrier()
    toks = DataParallelDecoder(cp, lambda prompts, n: np.arange(n * len(prompts), dtype=np.int32).reshape(n, len(prompts))).decode([[1], [2], [3]], 2)
    np.testing.assert_array_equal(toks, [[0, 1, 2], [3, 4, 5]])


def test_socket_control_plane_under_a_launcher_that_owns_master_port(tmp_path):
    """torch.distributed.run keeps its own store listening on MASTER_PORT, so the hub may not bind it: the ranks must still
    find each other (ephemeral hub port published through the rendezvous file), and a stale file left by an earlier run
    with the same MASTER_PORT must not mislead them."""
    import torch.multiprocessing as mp

    with socket.socket() as occupied:
        occupied.bind(("127.0.0.1", 0))
        occupied.listen(1)
        port = occupied.getsockname()[1]
        stale = tmp_path / f"pgk_cp_{port}_none_{os.getuid()}"
        stale.write_text(f"{_free_port()} deadbeefdeadbeef\n")      # a dead port and a wrong nonce
        mp.spawn(_worker, args=(2, port, str(tmp_path), "socket"), nprocs=2, join=True)
    assert not stale.exists()       # the hub removes its rendezvous file on shutdown
    np.testing.assert_array_equal(np.load(tmp_path / "tokens_0.npy"), np.load(tmp_path / "tokens_1.npy"))
