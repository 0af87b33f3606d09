"""pygpukit_amd - MI355X (gfx950) native implementation of PyGPUkit's LLM-inference hot path.

Same operator / model surface as the reference package `pygpukit` for that path (GPUArray, ops.*,
llm.CausalTransformerModel and the decode strategies); every op runs a hand-written HIP kernel
through the C ABI of libpgk_hip.so.  There is no CPU backend: importing works anywhere, using it
without the library or without a GPU raises.
"""

__version__ = "0.1.0"

from pygpukit_amd import core, ops  # noqa: F401
from pygpukit_amd.core import (CudaEvent, CudaGraph, DataType, GPUArray, Stream, bfloat16, device_synchronize, empty,  # noqa: F401
                              event_elapsed_ms, event_elapsed_us, float16, float32, float64, from_numpy, get_backend,
                              has_native_module, int4, int8, int16, int32, int64, ones, uint8, zeros)
from pygpukit_amd.core.device import (DeviceCapabilities, DeviceInfo, FallbackDeviceCapabilities, get_device_capabilities,  # noqa: F401,E402
                                      get_device_info, is_cuda_available)
from pygpukit_amd.core.stream import StreamManager, default_stream  # noqa: F401,E402
from pygpukit_amd.ops.basic import (abs, add, argmax, bias_add_inplace, clamp, cos, div, exp, gelu, layernorm,  # noqa: F401,E402,A004
                                    linear_bias_gelu, log, matmul, max, mean, min, mul, neg, relu, rsqrt, sigmoid, sin, softmax,
                                    sqrt, sub, sum, sum_axis, tanh, transpose, where)
from pygpukit_amd.jit import (JITKernel, NvrtcError, NvrtcErrorCode, get_nvrtc_path, get_nvrtc_version, is_nvrtc_available,  # noqa: F401,E402
                              jit, warmup)
