"""Streams, events and graph capture (reference: native/core/stream.hpp:21-35, event.hpp:13-38,
cuda_graph.hpp:31-88 bound in native/bindings/core_bindings.cpp:208-406, and the Python
Stream wrapper src/pygpukit/core/stream.py).  The reference's class names are kept
(CudaEvent, CudaGraph) so its callers run unchanged; underneath are hipStream/hipEvent/hipGraph."""

from __future__ import annotations

import ctypes as C

from pygpukit_amd import _hip


class Stream:
    def __init__(self, priority: str | int = "low") -> None:
        _hip.require_device()
        h = C.c_void_p()
        high = 1 if priority in ("high", 1, True) else 0
        _hip.call("pgk_stream_create", C.byref(h), high)
        self._h = h.value
        self.priority = "high" if high else "low"

    @property
    def handle(self) -> int:
        return self._h

    def synchronize(self) -> None:
        _hip.call("pgk_stream_sync", C.c_void_p(self._h))

    def make_current(self) -> None:
        _hip.call("pgk_stream_set_current", C.c_void_p(self._h))

    def __enter__(self):
        self.make_current()
        return self

    def __exit__(self, *exc):
        _hip.call("pgk_stream_set_current", None)
        return False

    def __del__(self):
        try:
            if self._h:
                _hip.call("pgk_stream_destroy", C.c_void_p(self._h))
        except Exception:
            pass
        self._h = 0


def default_stream() -> int:
    """Handle of the calling thread's current stream."""
    h = C.c_void_p()
    _hip.call("pgk_stream_get_current", C.byref(h))
    return h.value


def stream_synchronize(handle: int | None = None) -> None:
    _hip.call("pgk_stream_sync", C.c_void_p(handle) if handle else None)


class CudaEvent:
    """hipEvent with the reference's CudaEvent API: record([stream]), synchronize(), query()."""

    def __init__(self, blocking_sync: bool = False) -> None:
        _hip.require_device()
        h = C.c_void_p()
        _hip.call("pgk_event_create", C.byref(h))
        self._h = h.value

    def record(self, stream: Stream | None = None) -> None:
        _hip.call("pgk_event_record", C.c_void_p(self._h), C.c_void_p(stream.handle) if stream else None)

    def synchronize(self) -> None:
        _hip.call("pgk_event_sync", C.c_void_p(self._h))

    def query(self) -> bool:
        done = C.c_int(0)
        _hip.call("pgk_event_query", C.c_void_p(self._h), C.byref(done))
        return bool(done.value)

    def __del__(self):
        try:
            if self._h:
                _hip.call("pgk_event_destroy", C.c_void_p(self._h))
        except Exception:
            pass
        self._h = 0


Event = CudaEvent


def event_elapsed_ms(start: CudaEvent, stop: CudaEvent) -> float:
    ms = C.c_float(0)
    _hip.call("pgk_event_elapsed_ms", C.c_void_p(start._h), C.c_void_p(stop._h), C.byref(ms))
    return ms.value


def event_elapsed_us(start: CudaEvent, stop: CudaEvent) -> float:
    return event_elapsed_ms(start, stop) * 1000.0


class CudaGraph:
    """Stream-capture graph with the reference's method set (cuda_graph.hpp:48-88):
    begin_capture / end_capture / replay / synchronize / reset / is_ready / is_capturing / num_nodes
    / get_stream_handle.  Capture runs on a private stream that becomes the thread's current stream
    for its duration, so every op issued in between lands in the graph."""

    def __init__(self) -> None:
        self._stream = Stream()
        self._g = 0
        self._capturing = False

    def begin_capture(self) -> None:
        if self._capturing:
            raise RuntimeError("CudaGraph: capture already in progress")
        self.reset()
        self._stream.make_current()
        _hip.call("pgk_graph_begin_capture", C.c_void_p(self._stream.handle))
        self._capturing = True

    def end_capture(self) -> None:
        if not self._capturing:
            raise RuntimeError("CudaGraph: end_capture without begin_capture")
        g = C.c_void_p()
        try:
            _hip.call("pgk_graph_end_capture", C.c_void_p(self._stream.handle), C.byref(g))
        finally:
            self._capturing = False
            _hip.call("pgk_stream_set_current", None)
        self._g = g.value

    def replay(self) -> None:
        if not self._g:
            raise RuntimeError("CudaGraph: nothing captured")
        _hip.call("pgk_graph_launch", C.c_void_p(self._g), C.c_void_p(self._stream.handle))

    def synchronize(self) -> None:
        self._stream.synchronize()

    def reset(self) -> None:
        if self._g:
            _hip.call("pgk_graph_destroy", C.c_void_p(self._g))
            self._g = 0

    def is_ready(self) -> bool:
        return bool(self._g)

    def is_capturing(self) -> bool:
        return self._capturing

    @property
    def num_nodes(self) -> int:
        if not self._g:
            return 0
        n = C.c_size_t(0)
        _hip.call("pgk_graph_num_nodes", C.c_void_p(self._g), C.byref(n))
        return n.value

    def get_stream_handle(self) -> int:
        return self._stream.handle

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass


HipGraph = CudaGraph
