"""Device information (reference: src/pygpukit/core/device.py:11-120; native get_device_properties,
core_bindings.cpp:156-200).  Field names are the reference's; on an AMD part `multiprocessor_count` is the CU count,
`warp_size` the wavefront width (64) and `compute_capability` the gfx target split as (major, minor) - gfx950 -> (9, 5)."""

from __future__ import annotations

import re
from dataclasses import dataclass

from pygpukit_amd.core.backend import get_backend


@dataclass
class DeviceInfo:
    name: str
    total_memory: int
    compute_capability: tuple[int, int] | None
    multiprocessor_count: int
    max_threads_per_block: int
    warp_size: int


def is_cuda_available() -> bool:
    """The reference's name for "is a GPU usable": here, the HIP library loaded and a device is visible."""
    return get_backend().is_available()


is_hip_available = is_cuda_available


def _gfx_capability(arch: str) -> tuple[int, int] | None:
    m = re.match(r"gfx(\d)(\d)", arch or "")
    return (int(m.group(1)), int(m.group(2))) if m else None


def get_device_info(device_id: int = 0) -> DeviceInfo:
    p = get_backend().get_device_properties(device_id)
    return DeviceInfo(name=p.name, total_memory=p.total_memory, compute_capability=p.compute_capability or _gfx_capability(p.arch),
                      multiprocessor_count=p.multiprocessor_count, max_threads_per_block=p.max_threads_per_block, warp_size=p.warp_size)


@dataclass
class FallbackDeviceCapabilities:
    """device.py:66-76 with the reference's field names: `sm_version` carries the gfx number (950), the tensorcore
    flags say which MFMA input types the part has, `async_copy` says global->LDS DMA exists."""

    device_id: int
    name: str
    sm_version: int
    compute_capability: int
    tensorcore: bool
    tensorcore_fp16: bool
    tensorcore_bf16: bool
    async_copy: bool


DeviceCapabilities = FallbackDeviceCapabilities


def get_device_capabilities(device_id: int = 0) -> FallbackDeviceCapabilities:
    p = get_backend().get_device_properties(device_id)
    m = re.match(r"gfx(\d+)", p.arch or "")
    gfx = int(m.group(1)) if m else 0
    cdna = gfx >= 908 and gfx < 1000
    return FallbackDeviceCapabilities(device_id=device_id, name=p.name, sm_version=gfx, compute_capability=gfx, tensorcore=cdna,
                                      tensorcore_fp16=cdna, tensorcore_bf16=gfx >= 910 and cdna, async_copy=gfx >= 942 and cdna)
