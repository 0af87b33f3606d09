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


@pytest.mark.gpu
class TestPoolIsStreamAndGraphAware:
    """The device pool hands a freed block back at once only to the stream that owns it; another stream gets it when the
    owner has drained; blocks allocated while a graph is captured stay out of circulation until the graph is destroyed."""

    def test_a_block_freed_under_pending_work_is_not_given_to_another_stream(self):
        import pygpukit_amd as pk
        from pygpukit_amd import ops
        from pygpukit_amd.core import from_numpy, zeros

        n = 3 * 1024 * 1024 + 17                      # an unusual size class (14 MiB of fp32): nobody else caches it
        big = from_numpy(np.ones(64 * 1024 * 1024, np.float32))
        sa, sb = Stream(), Stream()
        with sa:
            x = zeros((n,), "float32")
            ptr = x.data_ptr()
            for _ in range(40):                       # ~ms of queued work on sa
                ops.add_inplace(big, big)
            del x                                      # freed while sa is still busy
            same_stream = zeros((n,), "float32")      # the owner may reuse it: stream order protects it
            assert same_stream.data_ptr() == ptr
            del same_stream
        with sb:
            other = zeros((n,), "float32")
            assert other.data_ptr() != ptr            # sa has not drained: sb must not get the block
            del other
        sa.synchronize()
        sb.synchronize()
        with sb:
            got = [zeros((n,), "float32") for _ in range(3)]
            assert ptr in [g.data_ptr() for g in got]  # drained: now it may change hands
        pk.device_synchronize()

    def test_blocks_allocated_during_capture_are_pinned_until_the_graph_is_reset(self):
        import pygpukit_amd as pk
        from pygpukit_amd import ops
        from pygpukit_amd.core import CudaGraph, from_numpy, zeros

        n = 5 * 1024 * 1024 + 3
        a = from_numpy(np.arange(n, dtype=np.float32))
        out = zeros((n,), "float32")
        g = CudaGraph()
        g.begin_capture()
        tmp = ops.add(a, a)                            # a temporary allocated INSIDE the capture
        ptr = tmp.data_ptr()
        ops.add(tmp, a, out=out)
        del tmp                                        # freed right away - the graph's nodes still write and read it
        g.end_capture()
        held = [zeros((n,), "float32") for _ in range(6)]
        assert ptr not in [h.data_ptr() for h in held]
        for h in held:
            ops.copy_to(a, h)                          # scribble over everything the pool hands out
        g.replay()
        g.synchronize()
        np.testing.assert_array_equal(out.to_numpy(), 3 * np.arange(n, dtype=np.float32))
        del held
        g.reset()                                      # graph gone: the block returns to the pool
        with Stream(g.get_stream_handle()):
            again = [zeros((n,), "float32") for _ in range(8)]
        assert ptr in [h.data_ptr() for h in again]

    def test_replay_is_ordered_with_the_current_stream(self):
        from pygpukit_amd import ops
        from pygpukit_amd.core import CudaGraph, from_numpy, zeros

        n = 1 << 22
        x = zeros((n,), "float32")
        y = zeros((n,), "float32")
        g = CudaGraph()
        g.begin_capture()
        ops.add(x, x, out=y)
        g.end_capture()
        for k in range(1, 6):
            src = from_numpy(np.full(n, float(k), np.float32))
            ops.copy_to(src, x)                        # current (default) stream
            g.replay()                                 # private stream: must see the copy ...
            np.testing.assert_array_equal(y.to_numpy()[:: 65537], np.full(n, 2.0 * k, np.float32)[:: 65537])   # ... and the read must see the replay
