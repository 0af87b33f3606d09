"""MemoryPool: the reference's tests/test_memory_pool.py cases against real device memory.  Pure
bookkeeping (size classes, quota arithmetic before any allocation) runs on the CPU."""

from __future__ import annotations

import numpy as np
import pytest

from pygpukit_amd.memory import MemoryPool, get_default_pool, set_default_pool

MB = 1024 * 1024


def test_pool_creation_and_size_classes():
    pool = MemoryPool(quota=100 * MB)
    assert pool.quota == 100 * MB and pool.used == 0 and pool.available == pool.quota and pool.cached == 0
    assert pool._get_size_class(1) == 256 and pool._get_size_class(257) == 1024 and pool._get_size_class(MB) == MB
    assert pool._get_size_class(300 * MB) == 300 * MB and pool._get_size_class(300 * MB + 1) == 301 * MB
    with pytest.raises(MemoryError):
        MemoryPool(quota=MB).allocate(2 * MB)               # rejected before touching the device
    set_default_pool(pool)
    assert get_default_pool() is pool
    set_default_pool(None)
    assert set(pool.stats()) >= {"used", "quota", "cached", "allocation_count", "reuse_count", "eviction_count"}


@pytest.mark.gpu
def test_allocate_free_reuse_and_stats():
    pool = MemoryPool(quota=100 * MB)
    b1 = pool.allocate(MB)
    assert b1.on_gpu and b1.size == MB and pool.used == MB
    pool.free(b1)
    assert pool.used == 0 and pool.cached == MB
    b2 = pool.allocate(MB - 5)                              # same size class: reused, no new device allocation
    s = pool.stats()
    assert b2 is b1 and s["reuse_count"] == 1 and s["cudamalloc_count"] == 1 and s["allocation_count"] == 2
    assert s["active_blocks"] == 1 and s["free_blocks"] == 0
    pool.clear()
    assert pool.used == 0 and pool.stats()["active_blocks"] == 0


@pytest.mark.gpu
def test_write_read_evict_restore_round_trip():
    pool = MemoryPool(quota=5 * MB, enable_eviction=True)
    data = np.arange(MB // 4, dtype=np.float32)
    blk = pool.allocate(MB)
    pool.write(blk, data)
    np.testing.assert_array_equal(pool.read(blk, np.float32), data)
    pool.evict(blk)
    assert not blk.on_gpu and blk.on_host and blk.device_ptr is None and pool.used == 0
    np.testing.assert_array_equal(pool.read(blk, np.float32), data)          # served from the host copy
    pool.restore(blk)
    assert blk.on_gpu and not blk.on_host and pool.used == MB
    np.testing.assert_array_equal(pool.read(blk, np.float32), data)
    assert pool.stats()["eviction_count"] == 1


@pytest.mark.gpu
def test_lru_eviction_order_under_quota():
    pool = MemoryPool(quota=10 * MB, enable_eviction=True)
    a, b = pool.allocate(4 * MB), pool.allocate(4 * MB)      # 8 MB of a 10 MB quota
    pool.write(b, np.full(MB, 7, np.int32))
    pool.touch(a)                                            # a is now the most recently used
    c = pool.allocate(4 * MB)                                # does not fit: the least recently used block (b) goes to the host
    assert pool.stats()["eviction_count"] == 1 and pool.used <= pool.quota
    assert a.on_gpu and c.on_gpu and not b.on_gpu and b.on_host
    np.testing.assert_array_equal(pool.read(b, np.int32), 7)                 # its bytes survived the eviction
    small = MemoryPool(quota=4 * MB)                         # eviction disabled: the quota is a hard limit
    small.allocate(4 * MB)
    with pytest.raises(MemoryError):
        small.allocate(256)
