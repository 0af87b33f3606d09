"""Backend selection.  The reference switches between CPUSimulationBackend and NativeBackend
(src/pygpukit/core/backend.py:217-502); this package has ONE backend, HipBackend, backed by
libpgk_hip.so.  The names the reference's callers probe (`get_backend`, `NativeBackend`,
`has_native_module`, `get_native_module`) resolve to it so `isinstance(get_backend(), NativeBackend)`
keeps meaning "the native path is active"; there is no CPU simulation to fall back to."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd import _hip


class HipBackend:
    """Device management + pooled allocation on top of the C ABI."""

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

    def get_device_properties(self, dev: int = 0) -> dict:
        p = _hip.DeviceProps()
        _hip.call("pgk_device_props", dev, C.byref(p))
        return {"name": p.name.decode(), "arch": p.arch.decode(), "total_memory": p.total_mem,
                "multiprocessor_count": p.cu_count, "wavefront_size": p.wavefront_size,
                "clock_khz": p.clock_khz, "lds_per_cu": p.lds_per_cu, "l2_bytes": p.l2_bytes}

    def pool_stats(self) -> dict:
        s = _hip.PoolStats()
        _hip.call("pgk_pool_stats", C.byref(s))
        return {f: getattr(s, f) for f, _ in s._fields_}

    def pool_trim(self) -> None:
        _hip.call("pgk_pool_trim")


NativeBackend = HipBackend  # the reference's name for "the GPU backend"
_backend: HipBackend | None = None


def get_backend() -> HipBackend:
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


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
