"""Stream manager surface (reference: src/pygpukit/core/stream.py:11-135, pinned by its tests/test_stream.py):
StreamPriority / Stream / StreamManager / default_stream, on hipStreams.  The enum and the host bookkeeping run
without a GPU; everything that creates a stream is a GPU test."""

from __future__ import annotations

import numpy as np
import pytest

from pygpukit_amd.core.stream import Stream, StreamManager, StreamPriority, default_stream, get_stream_manager


def test_priority_enum_orders_high_below_low():
    assert StreamPriority.HIGH == 0 and StreamPriority.LOW == 1
    assert StreamPriority.HIGH < StreamPriority.LOW


def test_wrapping_an_existing_handle_needs_no_device():
    """Stream(handle, priority) is the reference constructor: it wraps, it does not create (and never destroys)."""
    s = Stream(0x1234, StreamPriority.HIGH)
    assert s.handle == 0x1234 and s.priority == StreamPriority.HIGH and "HIGH" in repr(s)
    s.destroy()
    assert s.handle == 0
    m = StreamManager()
    assert m._streams == [] and m._default_stream is None
    m.synchronize_all()          # nothing to wait for


@pytest.mark.gpu
class TestStreamManagerOnDevice:
    def test_create_by_string_and_enum(self):
        m = StreamManager()
        hi, lo, en = m.create_stream(priority="high"), m.create_stream(priority="low"), m.create_stream(priority=StreamPriority.HIGH)
        assert hi.priority == StreamPriority.HIGH and lo.priority == StreamPriority.LOW and en.priority == StreamPriority.HIGH
        assert "HIGH" in repr(hi) and "LOW" in repr(lo)
        assert hi.handle and lo.handle and hi.handle != lo.handle
        hi.synchronize()
        m.synchronize_all()
        assert len(m._streams) == 3

    def test_destroy_removes_the_stream(self):
        m = StreamManager()
        s = m.create_stream()
        m.destroy_stream(s)
        assert s not in m._streams and s.handle == 0
        m.destroy_stream(s)      # a second destroy is a no-op

    def test_default_stream_is_one_low_priority_instance(self):
        m = StreamManager()
        a, b = m.get_default_stream(), m.get_default_stream()
        assert a is b and a.priority == StreamPriority.LOW
        assert isinstance(default_stream(), Stream) and default_stream() is get_stream_manager().get_default_stream()

    def test_many_streams(self):
        m = StreamManager()
        streams = [m.create_stream() for _ in range(5)]
        assert len(streams) == 5 and len(m._streams) == 5 and len({s.handle for s in streams}) == 5

    def test_ops_launch_on_the_current_stream_and_events_order_them(self):
        """Inside `with stream:` every op without an explicit stream goes to that stream; an event recorded there and
        synchronised makes the result visible to the host."""
        import pygpukit_amd as pk
        from pygpukit_amd import ops
        from pygpukit_amd.core import CudaEvent, current_stream_handle, from_numpy

        rng = np.random.default_rng(3)
        a, b = rng.standard_normal((64, 256)).astype(np.float32), rng.standard_normal((64, 256)).astype(np.float32)
        da, db = from_numpy(a), from_numpy(b)
        s1, s2 = Stream("high"), Stream(priority="low")
        base = current_stream_handle()          # the library's own default stream
        assert base not in (s1.handle, s2.handle)
        with s1:
            assert current_stream_handle() == s1.handle
            c1 = ops.add(da, db)
            ev = CudaEvent()
            ev.record(s1)
        with s2:
            assert current_stream_handle() == s2.handle
            c2 = ops.mul(da, db)
        assert current_stream_handle() == base
        ev.synchronize()
        assert ev.query()
        s2.synchronize()
        np.testing.assert_array_equal(c1.to_numpy(), a + b)
        np.testing.assert_array_equal(c2.to_numpy(), a * b)
        pk.device_synchronize()
