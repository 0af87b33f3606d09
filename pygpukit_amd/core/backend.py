"""Backend selection.  The reference switches between CPUSimulationBackend and NativeBackend
(src/pygpukit/core/backend.py:217-502); this package has ONE backend, HipBackend, backed by
libpgk_hip.so.  The names the reference's callers probe (`get_backend`, `NativeBackend`,
`has_native_module`, `get_native_module`) resolve to it so `isinstance(get_backend(), NativeBackend)`
keeps meaning "the native path is active"; there is no CPU simulation to fall back to."""

from __future__ import annotations

import ctypes as C
import re
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any

import numpy as np

from pygpukit_amd import _hip


@dataclass
class DeviceProperties:
    """backend.py:198-214, with the reference's attribute names (`multiprocessor_count` = CUs, `warp_size` = wavefront
    width, `compute_capability` = the gfx target as (major, minor): gfx950 -> (9, 5)) plus what an AMD part adds.
    Subscripting by the older dict keys keeps working."""

    name: str
    total_memory: int
    compute_capability: tuple[int, int] | None = None
    multiprocessor_count: int = 0
    max_threads_per_block: int = 1024
    warp_size: int = 64
    arch: str = ""
    clock_khz: int = 0
    lds_per_cu: int = 0
    l2_bytes: int = 0

    @property
    def wavefront_size(self) -> int:
        return self.warp_size

    def __getitem__(self, key: str):
        return getattr(self, key)

    def keys(self):
        return ["name", "arch", "total_memory", "multiprocessor_count", "wavefront_size", "clock_khz", "lds_per_cu", "l2_bytes"]


class Backend(ABC):
    """The reference's backend interface (backend.py:217-280).  One implementation exists here."""

    @abstractmethod
    def is_available(self) -> bool: ...

    @abstractmethod
    def get_device_count(self) -> int: ...

    @abstractmethod
    def get_device_properties(self, device_id: int = 0) -> DeviceProperties: ...

    @abstractmethod
    def allocate(self, size_bytes: int) -> Any: ...

    @abstractmethod
    def free(self, ptr: Any) -> None: ...

    @abstractmethod
    def copy_host_to_device(self, host_data: np.ndarray, device_ptr: Any) -> None: ...

    @abstractmethod
    def copy_device_to_host(self, device_ptr: Any, size_bytes: int, dtype) -> np.ndarray: ...

    @abstractmethod
    def memset(self, device_ptr: Any, value: int, size_bytes: int) -> None: ...

    @abstractmethod
    def synchronize(self) -> None: ...

    @abstractmethod
    def create_stream(self, priority: int = 0) -> Any: ...

    @abstractmethod
    def destroy_stream(self, stream: Any) -> None: ...

    @abstractmethod
    def stream_synchronize(self, stream: Any) -> None: ...


class CPUSimulationBackend:
    """The reference's NumPy stand-in for a GPU (backend.py:283-360).  Deliberately absent from this build: every op is
    a HIP kernel, and silently computing on the host would defeat the point.  Constructing it says so."""

    def __init__(self) -> None:
        raise RuntimeError("pygpukit_amd has no CPU simulation backend: ops run on the GPU through libpgk_hip.so or fail")


class HipBackend(Backend):
    """Device management + pooled allocation on top of the C ABI."""

    def get_device_count(self) -> int:
        return _hip.device_count()

    def copy_host_to_device(self, host_data: np.ndarray, device_ptr: int) -> None:
        a = np.ascontiguousarray(host_data)
        _hip.call("pgk_memcpy_h2d", C.c_void_p(device_ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, None)

    def copy_device_to_host(self, device_ptr: int, size_bytes: int, dtype) -> np.ndarray:
        np_dt = dtype.to_numpy_dtype() if hasattr(dtype, "to_numpy_dtype") else np.dtype(dtype)
        out = np.empty(size_bytes // np.dtype(np_dt).itemsize, dtype=np_dt)
        _hip.call("pgk_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(device_ptr), size_bytes, None)
        return out

    def create_stream(self, priority: int = 0) -> int:
        h = C.c_void_p()
        _hip.call("pgk_stream_create", C.byref(h), 1 if priority == 0 else 0)     # StreamPriority.HIGH == 0
        return h.value

    def destroy_stream(self, stream: int) -> None:
        if stream:
            _hip.call("pgk_stream_destroy", C.c_void_p(stream))

    def stream_synchronize(self, stream: int) -> None:
        _hip.call("pgk_stream_sync", C.c_void_p(stream) if stream else None)

    def is_available(self) -> bool:
        try:
            return _hip.device_count() > 0
        except (RuntimeError, OSError):
            return False

    # allocation -----------------------------------------------------------------
    def allocate(self, nbytes: int) -> int:
        _hip.require_device()
        p = C.c_void_p()
        _hip.call("pgk_malloc", C.byref(p), max(int(nbytes), 1))
        return p.value

    def free(self, ptr: int) -> None:
        if ptr:
            _hip.call("pgk_free", C.c_void_p(ptr))

    def memset(self, ptr: int, value: int, nbytes: int) -> None:
        _hip.call("pgk_memset", C.c_void_p(ptr), value, nbytes, None)

    # device ---------------------------------------------------------------------
    def synchronize(self) -> None:
        _hip.call("pgk_device_sync")

    def device_count(self) -> int:
        return _hip.device_count()

    def set_device(self, dev: int) -> None:
        _hip.call("pgk_device_set", dev)

    def get_device_properties(self, dev: int = 0) -> DeviceProperties:
        p = _hip.DeviceProps()
        _hip.call("pgk_device_props", dev, C.byref(p))
        arch = p.arch.decode()
        m = re.match(r"gfx(\d)(\d)", arch)
        return DeviceProperties(name=p.name.decode(), total_memory=p.total_mem, compute_capability=(int(m.group(1)), int(m.group(2))) if m else None,
                                multiprocessor_count=p.cu_count, max_threads_per_block=1024, warp_size=p.wavefront_size, arch=arch,
                                clock_khz=p.clock_khz, lds_per_cu=p.lds_per_cu, l2_bytes=p.l2_bytes)

    def pool_stats(self) -> dict:
        s = _hip.PoolStats()
        _hip.call("pgk_pool_stats", C.byref(s))
        return {f: getattr(s, f) for f, _ in s._fields_}

    def pool_trim(self) -> None:
        _hip.call("pgk_pool_trim")


NativeBackend = HipBackend  # the reference's name for "the GPU backend"
_backend: Backend | None = None


def get_backend() -> Backend:
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(backend: Backend) -> None:
    """backend.py:493-496.  Only the bookkeeping calls (allocation, copies, streams, device queries) go through the
    backend object; the ops themselves always launch HIP kernels."""
    global _backend
    _backend = backend


def reset_backend() -> None:
    global _backend
    _backend = None


def has_rust_module() -> bool:
    """The reference's Rust scheduler / loader extension (backend.py:505-560): replaced by C++ here, never present."""
    return False


def get_rust_module():
    return None


def has_native_module() -> bool:
    try:
        _hip.load()
        return True
    except (RuntimeError, OSError):
        return False


def get_native_module():
    """The loaded ctypes library (reference: the pybind11 module object)."""
    return _hip.load()


def device_synchronize() -> None:
    get_backend().synchronize()
