"""Raw copies and memory queries (reference: src/pygpukit/core/memory.py:18-215 -> native memcpy_ptr_to_device*,
memcpy_device_to_device*, core_bindings.cpp:232-361).  `src_ptr` is a HOST address (an int, e.g. from a NumPy buffer or a
file mapping); pin it (pgk_host_alloc) if the asynchronous forms are to overlap with anything."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.stream import Stream


def get_memory_info() -> tuple[int, int]:
    """(free_bytes, total_bytes) of the current device (hipMemGetInfo; the reference returns total twice)."""
    free, total = C.c_size_t(), C.c_size_t()
    _hip.call("pgk_mem_info", C.byref(free), C.byref(total))
    return int(free.value), int(total.value)


def _check(dst: GPUArray, offset: int, size_bytes: int, what: str) -> None:
    if size_bytes < 0 or offset < 0 or offset + size_bytes > dst.nbytes:
        raise ValueError(f"{what}: {size_bytes} bytes at offset {offset} do not fit in {dst.nbytes}")


def copy_to_device_async_raw_stream(dst: GPUArray, src_ptr: int, size_bytes: int, stream_handle: int) -> None:
    _check(dst, 0, size_bytes, "copy_to_device_async")
    _hip.call("pgk_memcpy_h2d_async", dst._p, C.c_void_p(src_ptr), size_bytes, C.c_void_p(stream_handle) if stream_handle else None)


def copy_to_device_async(dst: GPUArray, src_ptr: int, size_bytes: int, stream: Stream) -> None:
    copy_to_device_async_raw_stream(dst, src_ptr, size_bytes, stream.handle)


def copy_to_device(dst: GPUArray, src_ptr: int, size_bytes: int) -> None:
    _check(dst, 0, size_bytes, "copy_to_device")
    _hip.call("pgk_memcpy_h2d", dst._p, C.c_void_p(src_ptr), size_bytes, None)


def copy_device_to_device_async(dst: GPUArray, src: GPUArray, stream: Stream) -> None:
    if dst.nbytes != src.nbytes:
        raise ValueError(f"Size mismatch: dst.nbytes={dst.nbytes}, src.nbytes={src.nbytes}")
    _hip.call("pgk_memcpy_d2d", dst._p, src._p, src.nbytes, C.c_void_p(stream.handle))


def copy_device_to_device_offset(dst: GPUArray, dst_offset_bytes: int, src: GPUArray, src_offset_bytes: int, size_bytes: int) -> None:
    _check(dst, dst_offset_bytes, size_bytes, "copy_device_to_device_offset (dst)")
    _check(src, src_offset_bytes, size_bytes, "copy_device_to_device_offset (src)")
    _hip.call("pgk_memcpy_d2d", C.c_void_p(dst.data_ptr() + dst_offset_bytes), C.c_void_p(src.data_ptr() + src_offset_bytes), size_bytes, None)


def synchronize() -> None:
    _hip.call("pgk_device_sync")
