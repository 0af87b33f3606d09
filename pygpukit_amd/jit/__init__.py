"""Runtime compilation of user kernels (reference: src/pygpukit/jit/__init__.py)."""
from pygpukit_amd.jit.compiler import (JITKernel, NvrtcError, NvrtcErrorCode, check_driver_compatibility, compile_to_code_object,
                                       compile_to_ptx, get_driver_requirements, get_nvrtc_path, get_nvrtc_version, get_warmup_error,
                                       is_nvrtc_available, is_warmup_done, jit, warmup)

HiprtcError, HiprtcErrorCode = NvrtcError, NvrtcErrorCode
is_hiprtc_available, get_hiprtc_version, get_hiprtc_path = is_nvrtc_available, get_nvrtc_version, get_nvrtc_path

__all__ = ["jit", "JITKernel", "NvrtcError", "NvrtcErrorCode", "HiprtcError", "HiprtcErrorCode", "is_nvrtc_available",
           "get_nvrtc_version", "get_nvrtc_path", "is_hiprtc_available", "get_hiprtc_version", "get_hiprtc_path",
           "compile_to_ptx", "compile_to_code_object", "warmup", "is_warmup_done", "get_warmup_error", "get_driver_requirements",
           "check_driver_compatibility"]
