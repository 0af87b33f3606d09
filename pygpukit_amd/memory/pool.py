"""MemoryPool: size-class free lists under a byte quota with optional LRU eviction to host memory
(reference: src/pygpukit/memory/pool.py:36-570, Python backend; the Rust twin rust/pygpukit-core/src/memory is the same
bookkeeping and is not reproduced).  Device memory comes from the library's stream-safe caching allocator
(pgk_malloc / pgk_free, csrc/runtime.hip); `device_pool_stats()` exposes that allocator's own counters."""

from __future__ import annotations

import ctypes as C
import threading
import time
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from pygpukit_amd import _hip


@dataclass
class MemoryBlock:
    id: int
    size: int
    device_ptr: Any = None
    host_data: np.ndarray | None = None
    on_gpu: bool = True
    on_host: bool = False
    last_access: float = field(default_factory=time.time)

    def touch(self) -> None:
        self.last_access = time.time()


_default_pool: "MemoryPool | None" = None


def set_default_pool(pool: "MemoryPool | None") -> None:
    global _default_pool
    _default_pool = pool


def get_default_pool() -> "MemoryPool | None":
    return _default_pool


def device_pool_stats() -> dict[str, int]:
    """Counters of the native caching allocator every GPUArray draws from (pgk_pool_stats)."""
    st = _hip.PoolStats()
    _hip.call("pgk_pool_stats", C.byref(st))
    return {name: int(getattr(st, name)) for name, _ in _hip.PoolStats._fields_}


class MemoryPool:
    SIZE_CLASSES = [256, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304, 16777216, 67108864, 268435456]

    def __init__(self, quota: int, enable_eviction: bool = False):
        self._quota, self._enable_eviction = int(quota), bool(enable_eviction)
        self._lock = threading.RLock()
        self._active: dict[int, MemoryBlock] = {}
        self._free_lists: dict[int, list[MemoryBlock]] = {}
        self._lru: OrderedDict[int, MemoryBlock] = OrderedDict()
        self._next_id = 0
        self._used = self._cached = 0
        self._allocation_count = self._reuse_count = self._eviction_count = self._cudamalloc_count = 0

    quota = property(lambda self: self._quota)
    used = property(lambda self: self._used)
    cached = property(lambda self: self._cached)
    available = property(lambda self: self._quota - self._used)

    def _get_size_class(self, size: int) -> int:
        for sc in self.SIZE_CLASSES:
            if size <= sc:
                return sc
        return ((size + 1048575) // 1048576) * 1048576

    # ------------------------------------------------------------------ device memory
    @staticmethod
    def _dev_alloc(nbytes: int) -> int:
        _hip.require_device()
        p = C.c_void_p()
        _hip.call("pgk_malloc", C.byref(p), nbytes)
        return int(p.value)

    @staticmethod
    def _dev_free(ptr: int) -> None:
        _hip.call("pgk_free", C.c_void_p(ptr))

    # ------------------------------------------------------------------ allocate / free
    def allocate(self, size: int) -> MemoryBlock:
        sc = self._get_size_class(int(size))
        with self._lock:
            fl = self._free_lists.get(sc)
            if fl:
                block = fl.pop()
                block.touch()
                self._active[block.id] = self._lru[block.id] = block
                self._lru.move_to_end(block.id)
                self._used += block.size
                self._cached -= block.size
                self._reuse_count += 1
                self._allocation_count += 1
                return block
            if self._used + sc > self._quota:
                if not self._enable_eviction or sc > self._quota:
                    raise MemoryError(f"Memory pool quota exceeded: requested {sc}, used {self._used}, quota {self._quota}")
                self._evict_lru(self._used + sc - self._quota)
                if self._used + sc > self._quota:
                    raise MemoryError(f"Memory pool quota exceeded after eviction: requested {sc}, used {self._used}")
            if self._used + self._cached + sc > self._quota:     # cached blocks of other classes count against the device too
                self._release_cached(self._used + self._cached + sc - self._quota)
            block = MemoryBlock(id=self._next_id, size=sc, device_ptr=self._dev_alloc(sc))
            self._next_id += 1
            self._active[block.id] = self._lru[block.id] = block
            self._used += sc
            self._allocation_count += 1
            self._cudamalloc_count += 1
            return block

    def free(self, block: MemoryBlock) -> None:
        with self._lock:
            if block.id not in self._active:
                return
            del self._active[block.id]
            self._lru.pop(block.id, None)
            if block.on_gpu:
                self._used -= block.size
                self._cached += block.size
                self._free_lists.setdefault(block.size, []).append(block)
            else:                       # evicted: nothing on the device, drop the host copy
                block.host_data = None
                block.on_host = False

    def _release_cached(self, needed: int) -> None:
        freed = 0
        for sc in sorted(self._free_lists, reverse=True):
            fl = self._free_lists[sc]
            while fl and freed < needed:
                b = fl.pop()
                self._dev_free(b.device_ptr)
                b.device_ptr, b.on_gpu = None, False
                self._cached -= b.size
                freed += b.size

    def touch(self, block: MemoryBlock) -> None:
        with self._lock:
            block.touch()
            if block.id in self._lru:
                self._lru.move_to_end(block.id)

    # ------------------------------------------------------------------ eviction
    def _evict_lru(self, needed: int) -> None:
        freed, victims = 0, []
        for b in self._lru.values():
            if freed >= needed:
                break
            if b.on_gpu:
                victims.append(b)
                freed += b.size
        for b in victims:
            self.evict(b)

    def evict(self, block: MemoryBlock) -> None:
        """Move a block's bytes to host memory and release its device memory."""
        if not block.on_gpu:
            return
        with self._lock:
            host = np.empty(block.size, np.uint8)
            _hip.call("pgk_memcpy_d2h", host.ctypes.data_as(C.c_void_p), C.c_void_p(block.device_ptr), block.size, None)
            self._dev_free(block.device_ptr)
            block.host_data, block.device_ptr, block.on_gpu, block.on_host = host, None, False, True
            self._eviction_count += 1
            if block.id in self._active:
                self._used -= block.size

    def restore(self, block: MemoryBlock) -> None:
        if block.on_gpu:
            return
        with self._lock:
            if block.id in self._active and self._used + block.size > self._quota:
                if not self._enable_eviction:
                    raise MemoryError("Memory pool quota exceeded while restoring an evicted block")
                self._evict_lru(self._used + block.size - self._quota)
            ptr = self._dev_alloc(block.size)
            if block.host_data is not None:
                src = np.ascontiguousarray(block.host_data).view(np.uint8).ravel()
                _hip.call("pgk_memcpy_h2d", C.c_void_p(ptr), src.ctypes.data_as(C.c_void_p), min(src.nbytes, block.size), None)
            block.device_ptr, block.on_gpu, block.on_host, block.host_data = ptr, True, False, None
            if block.id in self._active:
                self._used += block.size
                self._lru[block.id] = block
                self._lru.move_to_end(block.id)

    # ------------------------------------------------------------------ data access
    def write(self, block: MemoryBlock, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        if data.nbytes > block.size:
            raise ValueError(f"write of {data.nbytes} bytes into a block of {block.size}")
        if not block.on_gpu:
            self.restore(block)
        _hip.call("pgk_memcpy_h2d", C.c_void_p(block.device_ptr), data.ctypes.data_as(C.c_void_p), data.nbytes, None)
        self.touch(block)

    def read(self, block: MemoryBlock, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        if not block.on_gpu:
            if block.host_data is not None:
                return np.ascontiguousarray(block.host_data).view(np.uint8).ravel()[: block.size // dtype.itemsize * dtype.itemsize].view(dtype)
            return np.zeros(block.size // dtype.itemsize, dtype=dtype)
        out = np.empty(block.size // dtype.itemsize, dtype=dtype)
        _hip.call("pgk_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(block.device_ptr), out.nbytes, None)
        self.touch(block)
        return out

    # ------------------------------------------------------------------ bookkeeping
    def stats(self) -> dict[str, Any]:
        with self._lock:
            return {"quota": self._quota, "used": self._used, "cached": self._cached, "available": self.available,
                    "allocation_count": self._allocation_count, "reuse_count": self._reuse_count,
                    "eviction_count": self._eviction_count, "cudamalloc_count": self._cudamalloc_count,
                    "active_blocks": len(self._active), "free_blocks": sum(len(fl) for fl in self._free_lists.values())}

    def clear(self) -> None:
        with self._lock:
            for b in list(self._active.values()):
                if b.on_gpu and b.device_ptr is not None:
                    self._dev_free(b.device_ptr)
                b.device_ptr, b.on_gpu, b.host_data, b.on_host = None, False, None, False
            for fl in self._free_lists.values():
                for b in fl:
                    if b.device_ptr is not None:
                        self._dev_free(b.device_ptr)
                    b.device_ptr, b.on_gpu = None, False
            self._active.clear()
            self._free_lists.clear()
            self._lru.clear()
            self._used = self._cached = 0

    def __del__(self):
        try:
            self.clear()
        except Exception:  # noqa: BLE001
            pass
