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


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's N-rank control flow (parallel.open_comm / broadcast_weights / headline_leg / config4_legs) at world 2 and 3
# on CPU: the control plane stands in for RCCL, the CPU oracle for the native engine, NumPy arrays for device arrays - so
# the code that will run on the 8-GPU node has run somewhere.
# ---------------------------------------------------------------------------------------------------------------------
class _PlaneComm:
    """parallel.py's `comm` protocol over the control plane (what RcclComm is on the GPUs)."""

    def __init__(self, cp):
        self.cp, self.destroyed = cp, False

    def broadcast(self, arr, root=0):
        arr[...] = np.frombuffer(self.cp.all_gather_object(arr.tobytes() if self.cp.rank == root else None)[root], dtype=arr.dtype).reshape(arr.shape)

    def all_gather(self, send, recv):
        parts = self.cp.all_gather_object(send.tobytes())
        recv[...] = np.stack([np.frombuffer(p, dtype=send.dtype).reshape(send.shape) for p in parts]).reshape(recv.shape)

    def destroy(self):
        self.destroyed = True


class _HostOps:
    def __init__(self):
        self._t0 = 0.0

    def sync(self):
        pass

    def empty(self, shape, dt):
        return np.empty(shape, dt)

    def from_host(self, a):
        return np.array(a)

    def to_host(self, a):
        return a

    def nbytes(self, a):
        return a.nbytes

    def timer_start(self):
        import time
        self._t0 = time.perf_counter()

    def timer_stop_ms(self):
        import time
        return (time.perf_counter() - self._t0) * 1e3


class _OracleEngine:
    """parallel.py's `engine` protocol on the CPU oracle (greedy, KV-cached), with the native engine's semantics: the state
    (token, position) of every slot lives with the engine, a replay advances every slot by one token and logs it."""

    def __init__(self, ref, max_batch):
        self.ref, self.B = ref, max_batch
        self.past = [None] * max_batch
        self.tok = [0] * max_batch
        self.log = []
        self.captured = 0

    def prefill(self, tokens, seq=0):
        hidden, past = self.ref(list(tokens), use_cache=True)
        self.past[seq] = past
        return self.ref.get_logits(hidden[-1:])[0]

    def set_state(self, tokens, positions):
        self.tok[: len(tokens)] = [int(t) for t in tokens]
        self.log = []

    def capture(self, batch):
        self.captured = batch

    def replay(self, n):
        assert self.captured, "replay before capture"
        for _ in range(n):
            row = []
            for b in range(self.captured):
                hidden, self.past[b] = self.ref([self.tok[b]], past_key_values=self.past[b], use_cache=True)
                self.tok[b] = int(np.argmax(self.ref.get_logits(hidden)[0]))
                row.append(self.tok[b])
            self.log.append(row)

    def read_tokens(self, batch, n):
        return np.asarray(self.log[:n], np.int32).reshape(n, batch)

    def launches_per_step(self):
        return 0


def _bench_worker(rank: int, world: int, port: int, out_dir: str, backend: str, fail_rank: int) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PGK_CP_DIR=out_dir)
    from oracle import cpu_ref as O
    from pygpukit_amd import parallel as DP
    from tests.golden_cfg import TINY

    cp = DP.ControlPlane(backend)
    ops = _HostOps()

    def make_comm(cp_):
        if cp_.rank == fail_rank:
            raise RuntimeError("no RCCL here")
        return _PlaneComm(cp_)

    if fail_rank >= 0:
        # the mandatory-communicator branch: ONE rank fails, EVERY rank raises with that rank's reason and nobody hangs
        with pytest.raises(DP.CommUnavailable, match=f"rank {fail_rank}: RuntimeError: no RCCL here"):
            DP.open_comm(cp, world, make_comm)
        with pytest.raises(DP.CommUnavailable, match="has no GPU of its own"):
            DP.open_comm(cp, world - 1, make_comm)       # the last rank has no device: nobody enters the communicator's rendezvous
        cp.barrier()
        cp.shutdown()
        return
    comm = DP.open_comm(cp, world, make_comm)
    assert comm is not None
    # weights: rank 0 draws them, the others receive them over the communicator
    w = O.make_qwen3_weights(TINY, seed=40, bf16=True)
    names = ["embed"] + [f"{i}.{k}" for i, lw in enumerate(w["layers"]) for k in sorted(lw)]
    flat = [w["embed"]] + [lw[k] for lw in w["layers"] for k in sorted(lw)]
    arrays = [np.array(a) if rank == 0 else np.zeros_like(a) for a in flat]
    bc = DP.broadcast_weights(cp, comm, arrays, ops)
    assert bc["arrays"] == len(names) and bc["GB"] > 0
    for a, b in zip(arrays, flat):
        np.testing.assert_array_equal(a, b)
    ref = O.build_qwen3_ref(TINY, w, max_pos=64)
    B, P, W, K = 2, 5, 1, 3
    prompts = np.random.default_rng(11).integers(0, TINY["vocab_size"], (world * B, P))
    mine = prompts[rank * B:(rank + 1) * B]
    head = DP.headline_leg(cp, comm, _OracleEngine(ref, B), mine, ops, batch=B, prompt_len=P, warm=W, steps=K)
    assert head["gather"]["own_shard_round_trips"] and head["gather"]["all_tokens"].shape == (world, W + K, B)
    assert abs(head["value"] - world * B * K / head["wall_s"]) < 1e-9
    c4 = DP.config4_legs(cp, lambda max_seq, max_batch: _OracleEngine(ref, max_batch),
                         lambda n, plen: np.random.default_rng(12).integers(0, TINY["vocab_size"], (n, plen)), ops,
                         prompt_len=4, steps=2, warm=1, global_batch=6, weak_per_gpu=1, bytes_per_step=lambda b: 1e6 * b)
    np.save(os.path.join(out_dir, f"head_{rank}.npy"), head["gather"]["all_tokens"])
    import json
    with open(os.path.join(out_dir, f"c4_{rank}.json"), "w") as f:
        json.dump(c4, f)
    cp.barrier()
    comm.destroy()
    cp.shutdown()


@pytest.mark.parametrize("backend,world", [("socket", 2), ("gloo", 2), ("socket", 3)])
def test_bench_control_flow_runs_at_world_n_on_cpu(backend, world, tmp_path):
    import json

    import torch.multiprocessing as mp

    from oracle import cpu_ref as O
    from tests.golden_cfg import TINY

    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path), backend, -1), nprocs=world, join=True)
    # every rank holds every rank's token log, and each shard equals the single-process oracle on the same prompt
    ref = O.build_qwen3_ref(TINY, O.make_qwen3_weights(TINY, seed=40, bf16=True), max_pos=64)
    prompts = np.random.default_rng(11).integers(0, TINY["vocab_size"], (world * 2, 5))
    logs = [np.load(tmp_path / f"head_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(logs[r], logs[0])
    for r in range(world):
        for b in range(2):
            p = [int(t) for t in prompts[r * 2 + b]]
            want = ref.generate(p, max_new_tokens=5, temperature=0.0, top_k=0, top_p=1.0)[len(p) + 1:]      # the first token comes from the prefill
            assert [int(t) for t in logs[0][r, :, b]] == want
    c4 = [json.load(open(tmp_path / f"c4_{r}.json")) for r in range(world)]
    for c in c4:
        assert c["weak"]["global_batch"] == world and c["weak"]["batch_per_gpu"] == 1 and c["weak"]["n_gpus"] == world
        assert ("strong" in c) == (6 % world == 0)
        if "strong" in c:
            assert c["strong"]["global_batch"] == 6 and c["strong"]["batch_per_gpu"] == 6 // world
        assert abs(c["weak"]["tokens_per_s"] - c4[0]["weak"]["tokens_per_s"]) < 1e-6      # max-over-ranks time: one number for the job
        assert c["expected"]["weak_expected"] == 1.0


def test_bench_without_a_communicator_on_one_rank_fails_on_every_rank(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_bench_worker, args=(2, _free_port(), str(tmp_path), "socket", 1), nprocs=2, join=True)


class _FakeHip:
    """Stand-in for pygpukit_amd._hip in RcclComm: records the calls, fails the named entry point on the named rank."""

    def __init__(self, rank, fail_rank, fail_call):
        self.rank, self.fail_rank, self.fail_call, self.calls = rank, fail_rank, fail_call, []

    def call(self, name, *args):
        self.calls.append(name)
        if self.rank == self.fail_rank and name == self.fail_call:
            raise RuntimeError(f"{name} failed here")
        if name == "pgk_comm_unique_id":
            args[0].raw = bytes(range(1, 129))
        if name == "pgk_comm_init":
            self.uid = bytes(args[1])
            args[0]._obj.value = 4242


def _bringup_worker(rank: int, world: int, port: int, out_dir: str, fail_rank: int, fail_call: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PGK_CP_DIR=out_dir)
    from pygpukit_amd import parallel as DP

    cp = DP.ControlPlane("socket")
    hip = _FakeHip(rank, fail_rank, fail_call)
    make = lambda cp_: DP.RcclComm(cp_, hip=hip)   # noqa: E731
    if fail_rank < 0:
        comm = DP.open_comm(cp, world, make)
        assert comm._h == 4242 and hip.uid == bytes(range(1, 129))          # every rank initialised with rank 0's id
        assert hip.calls == (["pgk_device_set", "pgk_comm_unique_id", "pgk_comm_init"] if rank == 0 else ["pgk_device_set", "pgk_comm_init"])
    elif fail_call == "pgk_comm_init":
        # past the vote: the failing rank raises inside the rendezvous call, open_comm's own vote reports it everywhere
        with pytest.raises(DP.CommUnavailable, match=f"rank {fail_rank}: RuntimeError: pgk_comm_init failed here"):
            DP.open_comm(cp, world, make)
    else:
        # before the rendezvous: EVERY rank stops with the failing rank's reason and NOBODY calls pgk_comm_init
        # (ncclCommInitRank would block the healthy ranks until the missing one arrives)
        with pytest.raises(DP.CommUnavailable, match=f"rank {fail_rank}: RuntimeError: {fail_call} failed here"):
            DP.open_comm(cp, world, make)
        assert "pgk_comm_init" not in hip.calls
    # ranks that meet in different collectives are told so
    if rank == 0:
        with pytest.raises(RuntimeError, match="different collectives"):
            cp.max_over_ranks(1.0)
    else:
        with pytest.raises(RuntimeError, match="different collectives"):
            cp.min_over_ranks(1.0)
    cp.barrier()
    cp.shutdown()


@pytest.mark.parametrize("fail_rank,fail_call", [(-1, ""), (1, "pgk_device_set"), (0, "pgk_comm_unique_id"), (1, "pgk_comm_init")])
def test_rccl_bring_up_phases_at_world_2(fail_rank, fail_call, tmp_path):
    """RcclComm's bring-up with the C ABI replaced by a recorder, two real processes on the socket control plane: a rank that
    fails BEFORE the communicator's rendezvous (no device, no unique id) must not leave the others in a different collective or
    inside ncclCommInitRank - found by rehearsing `bench.py --gpus 2` on a one-GPU box, where rank 1's device does not exist."""
    import torch.multiprocessing as mp

    mp.spawn(_bringup_worker, args=(2, _free_port(), str(tmp_path), fail_rank, fail_call), nprocs=2, join=True)


def test_expected_strong_scaling_from_the_n1_legs():
    sys.path.insert(0, ROOT)
    from pygpukit_amd.parallel import expected_config4_efficiency

    e = expected_config4_efficiency(9471.0, 55239.0, 1)      # round 2's N = 1 legs: 8 and 64 sequences on one GPU
    assert abs(e["strong_expected_at_8_from_n1_legs"] - 0.1715) < 1e-3 and e["weak_expected"] == 1.0
