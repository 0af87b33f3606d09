"""Streams, events and graph capture (reference: native/core/stream.hpp:21-35, event.hpp:13-38,
cuda_graph.hpp:31-88 bound in native/bindings/core_bindings.cpp:208-406, and the Python
Stream wrapper src/pygpukit/core/stream.py).  The reference's class names are kept
(CudaEvent, CudaGraph) so its callers run unchanged; underneath are hipStream/hipEvent/hipGraph."""

from __future__ import annotations

import ctypes as C
from enum import IntEnum

from pygpukit_amd import _hip


class StreamPriority(IntEnum):
    """core/stream.py:11-15: HIGH is numerically lower, as in the reference."""

    HIGH = 0
    LOW = 1


def _as_priority(priority) -> StreamPriority:
    if isinstance(priority, str):
        return StreamPriority.HIGH if priority.lower() == "high" else StreamPriority.LOW
    if priority is True:
        return StreamPriority.HIGH
    return StreamPriority(int(priority))


class Stream:
    """A hipStream with the reference's Python Stream surface (core/stream.py:18-52): `handle`, `priority`,
    `synchronize()`, repr naming the priority.  `Stream(handle, priority)` wraps an existing stream as the reference's
    constructor does; `Stream()` / `Stream("high")` / `Stream(priority=...)` creates one (the native `Stream(priority)`
    of core_bindings.cpp:208-230).  Used as a context manager it becomes the calling thread's current stream, which every
    op without an explicit stream argument launches on."""

    def __init__(self, stream_handle=None, priority: str | int | StreamPriority = StreamPriority.LOW) -> None:
        if isinstance(stream_handle, (str, StreamPriority)) or stream_handle is True:   # Stream("high")
            stream_handle, priority = None, stream_handle
        self._priority = _as_priority(priority)
        self._owns = stream_handle is None
        if stream_handle is None:
            _hip.require_device()
            h = C.c_void_p()
            _hip.call("pgk_stream_create", C.byref(h), 1 if self._priority == StreamPriority.HIGH else 0)
            stream_handle = h.value
        self._h = stream_handle

    @property
    def handle(self) -> int:
        return self._h

    @property
    def priority(self) -> StreamPriority:
        return self._priority

    def synchronize(self) -> None:
        _hip.call("pgk_stream_sync", C.c_void_p(self._h))

    def make_current(self) -> None:
        _hip.call("pgk_stream_set_current", C.c_void_p(self._h))

    def destroy(self) -> None:
        if self._h and self._owns:
            _hip.call("pgk_stream_destroy", C.c_void_p(self._h))
        self._h = 0

    def __enter__(self):
        self.make_current()
        return self

    def __exit__(self, *exc):
        _hip.call("pgk_stream_set_current", None)
        return False

    def __repr__(self) -> str:
        return f"Stream(priority={'HIGH' if self._priority == StreamPriority.HIGH else 'LOW'})"

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class StreamManager:
    """core/stream.py:55-122: creates, tracks and destroys streams; one lazily created LOW-priority default stream."""

    def __init__(self) -> None:
        self._streams: list[Stream] = []
        self._default_stream: Stream | None = None

    def create_stream(self, priority: str | StreamPriority = "low") -> Stream:
        stream = Stream(None, _as_priority(priority))
        self._streams.append(stream)
        return stream

    def destroy_stream(self, stream: Stream) -> None:
        if stream in self._streams:
            self._streams.remove(stream)
            if stream is self._default_stream:
                self._default_stream = None
            stream.destroy()

    def get_default_stream(self) -> Stream:
        if self._default_stream is None:
            self._default_stream = self.create_stream(StreamPriority.LOW)
        return self._default_stream

    def synchronize_all(self) -> None:
        for stream in self._streams:
            stream.synchronize()

    def __del__(self) -> None:
        for stream in getattr(self, "_streams", []):
            try:
                stream.destroy()
            except Exception:
                pass
        self._streams = []


_stream_manager: StreamManager | None = None


def get_stream_manager() -> StreamManager:
    global _stream_manager
    if _stream_manager is None:
        _stream_manager = StreamManager()
    return _stream_manager


def default_stream() -> Stream:
    """core/stream.py:133-135: the global manager's default stream (a Stream object)."""
    return get_stream_manager().get_default_stream()


def current_stream_handle() -> int:
    """Handle of the stream ops launch on when given none: the innermost `with stream:` of this thread, the graph
    capture stream while capturing, else the library's own non-blocking default stream."""
    h = C.c_void_p()
    _hip.call("pgk_stream_get_current", C.byref(h))
    return h.value or 0


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
        self._order_ev = None

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
        """Launch on the graph's private stream, ordered on both sides with the calling thread's current stream: work
        already queued there (the inputs the graph reads) finishes first, and work queued there afterwards sees the
        graph's results - the stream-ordered behaviour a launch on the current stream itself would have."""
        if not self._g:
            raise RuntimeError("CudaGraph: nothing captured")
        cur = C.c_void_p()
        _hip.call("pgk_stream_get_current", C.byref(cur))
        gs = C.c_void_p(self._stream.handle)
        if cur.value != self._stream.handle:
            if self._order_ev is None:
                ev = C.c_void_p()
                _hip.call("pgk_event_create", C.byref(ev))
                self._order_ev = ev.value
            _hip.call("pgk_event_record", C.c_void_p(self._order_ev), cur)
            _hip.call("pgk_stream_wait_event", gs, C.c_void_p(self._order_ev))
        _hip.call("pgk_graph_launch", C.c_void_p(self._g), gs)
        if cur.value != self._stream.handle:
            _hip.call("pgk_event_record", C.c_void_p(self._order_ev), gs)
            _hip.call("pgk_stream_wait_event", cur, C.c_void_p(self._order_ev))

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
            if self._order_ev:
                _hip.call("pgk_event_destroy", C.c_void_p(self._order_ev))
                self._order_ev = None
        except Exception:
            pass


HipGraph = CudaGraph
