"""Quota'd memory pool (reference: src/pygpukit/memory/__init__.py)."""
from pygpukit_amd.memory.pool import MemoryBlock, MemoryPool, device_pool_stats, get_default_pool, set_default_pool

HAS_RUST_BACKEND = False   # the reference's optional Rust bookkeeping backend has no counterpart: the logic below is the pool

__all__ = ["MemoryBlock", "MemoryPool", "get_default_pool", "set_default_pool", "device_pool_stats", "HAS_RUST_BACKEND"]
