"""JIT compiler front end (reference: src/pygpukit/jit/compiler.py:20-765).  The reference compiles CUDA C++ with NVRTC
to PTX; here HIP C++ (the same kernel-language subset: __global__, blockIdx, __shared__, ...) is compiled with hiprtc to
a gfx950 code object.  The reference's names are kept (`NvrtcError`, `is_nvrtc_available`, ...) so its callers and tests
read unchanged; `Hiprtc*` aliases are exported next to them.  Unlike the reference, whose JITKernel.__call__ is a stub
(compiler.py:570-594), calling a kernel here launches it."""

from __future__ import annotations

import ctypes as C
import hashlib
import re
import threading
from enum import IntEnum
from typing import Any, Callable

import numpy as np

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray


class NvrtcErrorCode(IntEnum):
    """hiprtcResult values (numerically identical to nvrtcResult) plus this layer's 1000+ codes (compiler.py:20-43)."""
    Success = 0
    OutOfMemory = 1
    ProgramCreationFailure = 2
    InvalidInput = 3
    InvalidProgram = 4
    InvalidOption = 5
    Compilation = 6
    BuiltinOperationFailure = 7
    NoNameExpressionsAfterCompilation = 8
    NoLoweredNamesBeforeCompilation = 9
    NameExpressionNotValid = 10
    InternalError = 11
    Linking = 100
    NotLoaded = 1000
    PtxLoadFailed = 1001
    FunctionNotFound = 1002
    LaunchFailed = 1003


class NvrtcError(RuntimeError):
    """Structured compilation / load / launch error (compiler.py:45-84)."""

    def __init__(self, message: str, code: NvrtcErrorCode = NvrtcErrorCode.InternalError, compilation_log: str = "") -> None:
        super().__init__(message)
        self._code, self._log = code, compilation_log

    @property
    def code(self) -> NvrtcErrorCode:
        return self._code

    @property
    def compilation_log(self) -> str:
        return self._log

    def __str__(self) -> str:
        base = super().__str__()
        return f"[{self._code.name}] {base}" + (f"\nCompilation log:\n{self._log}" if self._log else "")


def _code(v: int) -> NvrtcErrorCode:
    try:
        return NvrtcErrorCode(v)
    except ValueError:
        return NvrtcErrorCode.InternalError


def is_nvrtc_available() -> bool:
    """True when the runtime compiler (libhiprtc) can be loaded.  Pre-compiled operators work without it."""
    try:
        return bool(_hip.load().pgk_jit_available())
    except RuntimeError:
        return False


def get_nvrtc_path() -> str | None:
    if not is_nvrtc_available():
        return None
    p = _hip.load().pgk_jit_library_path()
    return p.decode() if p else None


def get_nvrtc_version() -> tuple[int, int] | None:
    if not is_nvrtc_available():
        return None
    major, minor = C.c_int(), C.c_int()
    _hip.call("pgk_jit_version", C.byref(major), C.byref(minor))
    return major.value, minor.value


def get_driver_requirements() -> dict[str, str]:
    v = get_nvrtc_version()
    return {"compiler": "hiprtc", "required": "ROCm >= 7.0 with libhiprtc", "target": "gfx950",
            "hiprtc_version": f"{v[0]}.{v[1]}" if v else "not loaded"}


def check_driver_compatibility() -> tuple[bool, str]:
    if not is_nvrtc_available():
        return False, "libhiprtc could not be loaded: JIT kernels are unavailable (pre-compiled operators are unaffected)"
    v = get_nvrtc_version()
    return True, f"hiprtc {v[0]}.{v[1]} loaded from {get_nvrtc_path()}"


class _Program:
    def __init__(self, source: str, name: str, options: list[str]):
        lib = _hip.load()
        opts = (C.c_char_p * max(len(options), 1))(*[o.encode() for o in options])
        handle, code = C.c_void_p(), C.c_int()
        try:
            _hip.call("pgk_jit_compile", source.encode(), name.encode(), opts, len(options), C.byref(handle), C.byref(code))
        except _hip.PgkError as e:
            log = lib.pgk_jit_program_log(handle).decode(errors="replace") if handle.value else ""
            if handle.value:
                lib.pgk_jit_program_destroy(handle)
            raise NvrtcError(str(e).split("\n")[0][:300], _code(code.value), log) from None
        self.handle = handle
        self.log = lib.pgk_jit_program_log(handle).decode(errors="replace")
        ptr, size = C.c_void_p(), C.c_size_t()
        _hip.call("pgk_jit_program_code", handle, C.byref(ptr), C.byref(size))
        self.code = C.string_at(ptr.value, size.value) if size.value else b""

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                _hip.load().pgk_jit_program_destroy(self.handle)
        except Exception:  # noqa: BLE001
            pass


class CompiledPTX:
    """compile_to_ptx's result (jit_bindings.cpp:65-68): `.ptx` holds the gfx950 code object bytes, `.log` the compiler log."""

    def __init__(self, code: bytes, log: str):
        self.ptx, self.code_object, self.log = code, code, log


def compile_to_ptx(source: str, name: str = "kernel.cu", options: list[str] | None = None) -> CompiledPTX:
    if not is_nvrtc_available():
        raise NvrtcError("runtime compiler not available", NvrtcErrorCode.NotLoaded)
    p = _Program(source, name, list(options or []))
    return CompiledPTX(p.code, p.log)


compile_to_code_object = compile_to_ptx

_SCALARS = {int: C.c_int, float: C.c_float, bool: C.c_bool, np.int32: C.c_int32, np.int64: C.c_int64, np.uint32: C.c_uint32,
            np.uint64: C.c_uint64, np.float32: C.c_float, np.float64: C.c_double, np.int8: C.c_int8, np.uint8: C.c_uint8,
            np.int16: C.c_int16, np.uint16: C.c_uint16}


class JITKernel:
    """A runtime-compiled kernel (compiler.py:270-599).  Arguments: GPUArray -> device pointer; Python int -> int;
    Python float -> float; NumPy scalars keep their width; ctypes values pass through."""

    def __init__(self, source: str, func_name: str, options: list[str] | None = None, block_size: int = 256) -> None:
        self._source, self._name, self._options, self._block_size = source, func_name, list(options or []), block_size
        self._program: _Program | None = None
        self._kernel = None
        self._is_compiled = False
        if not re.search(rf"__global__\s+\w+\s+{re.escape(func_name)}\s*\(", source):
            raise ValueError(f"Function '{func_name}' not found in source code")
        self._compile()

    def _compile(self) -> None:
        if not is_nvrtc_available():
            raise NvrtcError("runtime compiler (libhiprtc) not available", NvrtcErrorCode.NotLoaded)
        self._program = _Program(self._source, self._name + ".hip", self._options)
        self._is_compiled = True   # the code object exists; the module is loaded on first launch (needs a GPU)

    def _load(self) -> None:
        _hip.require_device()
        handle, code = C.c_void_p(), C.c_int()
        try:
            _hip.call("pgk_jit_kernel_create", self._program.handle, self._name.encode(), C.byref(handle), C.byref(code))
        except _hip.PgkError as e:
            raise NvrtcError(str(e), _code(code.value), self._program.log) from None
        self._kernel = handle

    source = property(lambda self: self._source)
    name = property(lambda self: self._name)
    options = property(lambda self: list(self._options))
    block_size = property(lambda self: self._block_size)
    is_compiled = property(lambda self: self._is_compiled)
    ptx = property(lambda self: self._program.code if self._program else None)
    log = property(lambda self: self._program.log if self._program else "")

    def _compute_cache_key(self) -> str:
        return hashlib.sha256((self._source + str(self._options)).encode()).hexdigest()

    def get_suggested_block_size(self, dynamic_smem: int = 0) -> int:
        if self._kernel is None:
            self._load()
        b = C.c_int()
        _hip.call("pgk_jit_suggested_block_size", self._kernel, dynamic_smem, C.byref(b))
        return b.value

    def __call__(self, *args: Any, grid_size: int | tuple | None = None, block_size: int | tuple | None = None,
                 shared_bytes: int = 0, stream=None) -> None:
        if not self._is_compiled:
            raise RuntimeError("Kernel not compiled")
        if self._kernel is None:
            self._load()
        vals, keep = [], []
        first_size = None
        for a in args:
            if isinstance(a, GPUArray):
                vals.append(C.c_void_p(a._p.value if hasattr(a._p, "value") else a._p))
                keep.append(a)
                first_size = a.size if first_size is None else first_size
            elif isinstance(a, C._SimpleCData):
                vals.append(a)
            elif type(a) in _SCALARS:
                vals.append(_SCALARS[type(a)](a))
            elif isinstance(a, np.generic) and a.dtype.type in _SCALARS:
                vals.append(_SCALARS[a.dtype.type](a.item()))
            else:
                raise TypeError(f"JITKernel: unsupported argument type {type(a).__name__}")
        block = block_size if block_size is not None else self._block_size
        block = (block, 1, 1) if isinstance(block, int) else tuple(block) + (1,) * (3 - len(block))
        if grid_size is None:
            if first_size is None:
                raise ValueError("JITKernel: pass grid_size= when no array argument fixes the problem size")
            grid_size = (first_size + block[0] - 1) // block[0]
        grid = (grid_size, 1, 1) if isinstance(grid_size, int) else tuple(grid_size) + (1,) * (3 - len(grid_size))
        argv = (C.c_void_p * max(len(vals), 1))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
        try:
            _hip.call("pgk_jit_launch", self._kernel, *grid, *block, shared_bytes, argv,
                      stream.handle if hasattr(stream, "handle") else stream)
        except _hip.PgkError as e:
            raise NvrtcError(str(e), NvrtcErrorCode.LaunchFailed) from None

    def __del__(self):
        try:
            if self._kernel is not None and self._kernel.value:
                _hip.load().pgk_jit_kernel_destroy(self._kernel)
        except Exception:  # noqa: BLE001
            pass

    def __repr__(self) -> str:
        return f"JITKernel(name={self._name}, {'compiled' if self._is_compiled else 'not compiled'})"


def jit(source: str, func: str, options: list[str] | None = None, block_size: int = 256) -> JITKernel:
    """Compile `func` from HIP/CUDA-style source at run time (compiler.py:601-636)."""
    return JITKernel(source, func, options, block_size)


# ---- warm-up (compiler.py:654-780): the first hiprtc compile loads comgr and takes seconds; do it ahead of time ----
_warmup_lock = threading.Lock()
_warmup_done = False
_warmup_thread: threading.Thread | None = None
_warmup_error: Exception | None = None
_WARMUP_KERNEL_SOURCE = """
extern "C" __global__ void _pygpukit_warmup_kernel(float* x, int n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) x[idx] = x[idx];
}
"""


def _do_warmup(callback: Callable[[], None] | None = None) -> bool:
    global _warmup_done, _warmup_error
    try:
        JITKernel(_WARMUP_KERNEL_SOURCE, "_pygpukit_warmup_kernel")
        ok = True
    except Exception as e:  # noqa: BLE001
        _warmup_error, ok = e, False
    with _warmup_lock:
        _warmup_done = True
    if callback is not None:
        callback()
    return ok


def warmup(background: bool = False, callback: Callable[[], None] | None = None) -> bool:
    global _warmup_thread
    with _warmup_lock:
        if _warmup_done:
            if callback is not None:
                callback()
            return _warmup_error is None
    if background:
        _warmup_thread = threading.Thread(target=_do_warmup, args=(callback,), daemon=True)
        _warmup_thread.start()
        return True
    return _do_warmup(callback)


def is_warmup_done() -> bool:
    return _warmup_done


def get_warmup_error() -> Exception | None:
    return _warmup_error
