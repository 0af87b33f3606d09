"""Data-parallel batch decode over one 8xMI355X node (new functionality: the reference is single-GPU,
docs/scheduler.md:358).

One process per GPU.  Independent sequences never talk to each other, so a decode step has NO collective
on its data path: the global batch is partitioned by rank, each rank runs its own engine on its shard, and
RCCL over xGMI carries only (1) the one-time broadcast of the weights from rank 0 and (2) the gather of
the sampled tokens (4 bytes per sequence per step).  The control plane (rendezvous, barriers, timing
reduction) is torch.distributed's gloo backend over TCP - plumbing, not the product path.

`shard_range` and `ControlPlane` are pure host logic and are covered by world_size-2 gloo tests on CPU."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of n_items over `world` ranks; the first n_items % world ranks get one
    extra item.  Returns [lo, hi)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ControlPlane:
    """Rank/world discovery from the torchrun environment + gloo collectives on small host tensors.
    world == 1 needs neither torch nor a rendezvous."""

    def __init__(self, backend: str = "gloo"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self) -> None:
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def min_over_ranks(self, value: float) -> float:
        return -self.max_over_ranks(-float(value))

    def first_note(self, note: str | None) -> str | None:
        """The first non-empty string any rank holds (rank order) - used to report one rank's error on rank 0."""
        if self._dist is None:
            return note
        outs = [None] * self.world
        self._dist.all_gather_object(outs, note)
        return next((o for o in outs if o), None)

    def all_gather_array(self, local: np.ndarray) -> list[np.ndarray]:
        """Equal-shape int32 arrays from every rank, over the control plane."""
        flat = self.gather_int32(np.ascontiguousarray(local, dtype=np.int32).ravel())
        return [f.reshape(local.shape) for f in flat]

    def sum_over_ranks(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t[0])

    def broadcast_bytes(self, data: bytes | None, n: int, root: int = 0) -> bytes:
        """Broadcast an n-byte blob from root (used for the 128-byte RCCL unique id)."""
        if self._dist is None:
            return bytes(data)
        import torch

        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == root:
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8).clone()
        self._dist.broadcast(t, src=root)
        return bytes(t.numpy().tobytes())

    def gather_int32(self, local: np.ndarray) -> np.ndarray:
        """All-gather equal-length int32 vectors over the control plane (CPU test path / fallback)."""
        local = np.ascontiguousarray(local, dtype=np.int32)
        if self._dist is None:
            return local[None, :]
        import torch

        outs = [torch.zeros(local.size, dtype=torch.int32) for _ in range(self.world)]
        self._dist.all_gather(outs, torch.from_numpy(local.copy()))
        return np.stack([o.numpy() for o in outs])

    def shutdown(self) -> None:
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()


class RcclComm:
    """RCCL communicator over the C ABI (pgk_comm_*): one per process, device = LOCAL_RANK."""

    def __init__(self, cp: ControlPlane):
        from pygpukit_amd import _hip

        self._hip = _hip
        self.cp = cp
        _hip.call("pgk_device_set", cp.local_rank)
        uid, err = None, None
        if cp.rank == 0:
            # a failure here must not skip the broadcast below (the other ranks are already waiting in it)
            try:
                buf = C.create_string_buffer(128)
                _hip.call("pgk_comm_unique_id", buf)
                uid = buf.raw
            except Exception as e:  # noqa: BLE001
                uid, err = bytes(128), e
        uid = cp.broadcast_bytes(uid, 128, 0)
        if uid == bytes(128):
            raise RuntimeError(f"RCCL unique id could not be created on rank 0: {err}")
        h = C.c_void_p()
        _hip.call("pgk_comm_init", C.byref(h), uid, cp.rank, cp.world)
        self._h = h.value

    def broadcast(self, arr, root: int = 0) -> None:
        """In-place broadcast of a GPUArray's bytes from root."""
        self._hip.call("pgk_comm_broadcast", C.c_void_p(self._h), arr._p, arr.nbytes, root, None)

    def all_gather(self, send, recv) -> None:
        """recv [world * send.nbytes] <- every rank's `send`."""
        self._hip.call("pgk_comm_all_gather", C.c_void_p(self._h), send._p, recv._p, send.nbytes, None)

    def barrier(self) -> None:
        self._hip.call("pgk_comm_barrier", C.c_void_p(self._h), None)

    def destroy(self) -> None:
        if self._h:
            self._hip.call("pgk_comm_destroy", C.c_void_p(self._h))
            self._h = 0


class DataParallelDecoder:
    """Greedy batch decode of independent sequences sharded over the ranks (BASELINE config 4).

    `runner(prompts) -> int32 [n_steps, len(prompts)]` decodes one rank's shard (on the GPU path:
    DecodeBatch.prefill + run_greedy on the native engine).  Shards are contiguous blocks of the global
    batch (shard_range); the data path has no collective; tokens are gathered once at the end (or per step
    by the caller) over `gather`, which defaults to the control plane and is RCCL on the GPU path."""

    def __init__(self, cp: ControlPlane, runner, gather=None):
        self.cp, self.runner, self._gather = cp, runner, gather

    def decode(self, prompts: list[list[int]], n_steps: int) -> np.ndarray | None:
        """Every rank passes the SAME global prompt list; returns int32 [n_steps, len(prompts)] on every rank."""
        n = len(prompts)
        lo, hi = shard_range(n, self.cp.rank, self.cp.world)
        local = self.runner(prompts[lo:hi], n_steps) if hi > lo else np.zeros((n_steps, 0), np.int32)
        local = np.ascontiguousarray(local, dtype=np.int32)
        if local.shape != (n_steps, hi - lo):
            raise ValueError(f"runner returned {local.shape}, expected {(n_steps, hi - lo)}")
        # equal-length payloads for the all-gather: pad every shard to the largest shard
        width = -(-n // self.cp.world)
        padded = np.full((n_steps, width), -1, np.int32)
        padded[:, : hi - lo] = local
        gathered = (self._gather or self.cp.gather_int32)(padded.ravel())  # [world, n_steps*width]
        out = np.empty((n_steps, n), np.int32)
        for r in range(self.cp.world):
            rlo, rhi = shard_range(n, r, self.cp.world)
            out[:, rlo:rhi] = np.asarray(gathered[r]).reshape(n_steps, width)[:, : rhi - rlo]
        return out
