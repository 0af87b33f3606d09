"""Core array type, dtypes, factory functions, streams and the backend switch."""

from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.backend import (Backend, CPUSimulationBackend, DeviceProperties, HipBackend, NativeBackend, device_synchronize,
                                      get_backend, get_native_module, get_rust_module, has_native_module, has_rust_module,
                                      reset_backend, set_backend)
from pygpukit_amd.core.dtypes import (DataType, DataTypeKind, bfloat16, float16, float32, float64, int4, int8, int16,
                                     int32, int64, uint8)
from pygpukit_amd.core.device import DeviceInfo, get_device_capabilities, get_device_info, is_cuda_available
from pygpukit_amd.core.factory import empty, from_numpy, ones, zeros
from pygpukit_amd.core.memory import (copy_device_to_device_async, copy_device_to_device_offset, copy_to_device, copy_to_device_async,
                                     get_memory_info, synchronize)
from pygpukit_amd.core.stream import (CudaEvent, CudaGraph, Event, HipGraph, Stream, StreamManager, StreamPriority,
                                     current_stream_handle, default_stream, event_elapsed_ms, event_elapsed_us,
                                     get_stream_manager, stream_synchronize)

__all__ = [
    "GPUArray", "DataType", "DataTypeKind", "float64", "float32", "float16", "bfloat16", "int64", "int32", "int16",
    "int8", "uint8", "int4", "zeros", "ones", "empty", "from_numpy", "get_backend", "HipBackend", "NativeBackend",
    "has_native_module", "get_native_module", "device_synchronize", "Backend", "CPUSimulationBackend", "DeviceProperties",
    "set_backend", "reset_backend", "has_rust_module", "get_rust_module", "Stream", "CudaEvent", "Event", "CudaGraph",
    "HipGraph", "default_stream", "stream_synchronize", "event_elapsed_ms", "event_elapsed_us", "StreamManager", "StreamPriority",
    "get_stream_manager", "current_stream_handle", "DeviceInfo", "get_device_info", "get_device_capabilities", "is_cuda_available",
    "get_memory_info", "copy_to_device", "copy_to_device_async", "copy_device_to_device_async", "copy_device_to_device_offset", "synchronize",
]
