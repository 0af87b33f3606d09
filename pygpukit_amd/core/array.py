"""GPUArray: a dense, C-contiguous N-d buffer in MI355X HBM.

Mirrors the public surface of the reference's GPUArray (src/pygpukit/core/array.py:17-742 over
native/core/memory.hpp:65-108): shape/dtype/size/ndim/nbytes/itemsize, to_numpy, astype, clone,
narrow/view/slice_rows zero-copy views (the view keeps its source alive, array.py:451-453),
reshape/transpose/T, arithmetic operators, __getitem__.  Storage is a raw device pointer from the
pooled allocator of libpgk_hip.so; bf16 travels as uint16 on the host (core/dtypes.py:54).
"""

from __future__ import annotations

import ctypes as C
import time
from typing import Any

import numpy as np

from pygpukit_amd import _hip
from pygpukit_amd.core.dtypes import DataType, bfloat16, float16, float32

_FLOATS = (float32, float16, bfloat16)


def _prod(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class GPUArray:
    __slots__ = ("_shape", "_dtype", "_ptr", "_owns_memory", "_source_ref", "_last_access", "__weakref__")

    def __init__(self, shape: tuple[int, ...], dtype: DataType, device_ptr: Any = None,
                 owns_memory: bool = True, _native: Any = None, _source_ref: "GPUArray | None" = None) -> None:
        # `_native` keeps the reference's positional signature (array.py:33); there is no second, native array object
        # here - an array given as `_native` is adopted as a view of its storage
        if _native is not None and device_ptr is None:
            device_ptr, owns_memory, _source_ref = _native.data_ptr(), False, _native
        self._shape = tuple(int(d) for d in shape)
        self._dtype = dtype
        self._source_ref = _source_ref
        self._last_access = time.time()
        if device_ptr is None:
            _hip.require_device()
            p = C.c_void_p()
            _hip.call("pgk_malloc", C.byref(p), max(self.nbytes, 1))
            self._ptr = p.value
            self._owns_memory = True
        else:
            self._ptr = int(device_ptr)
            self._owns_memory = owns_memory

    # ------------------------------------------------------------------ properties
    @property
    def last_access(self) -> float:
        """Timestamp of the last host-visible access (array.py:174-177): construction, device_ptr, to_numpy."""
        return self._last_access

    @property
    def shape(self) -> tuple[int, ...]:
        return self._shape

    @property
    def dtype(self) -> DataType:
        return self._dtype

    @property
    def size(self) -> int:
        return _prod(self._shape)

    @property
    def ndim(self) -> int:
        return len(self._shape)

    @property
    def itemsize(self) -> int:
        return self._dtype.itemsize

    @property
    def nbytes(self) -> int:
        return self.size * self._dtype.itemsize

    @property
    def owns_memory(self) -> bool:
        return self._owns_memory

    @property
    def device_ptr(self) -> int:
        self._last_access = time.time()
        return self._ptr

    @property
    def on_gpu(self) -> bool:
        return True

    def data_ptr(self) -> int:
        """Raw device address (reference: native GPUArray.data_ptr(), core_bindings.cpp:71-154)."""
        return self._ptr

    @property
    def _p(self) -> C.c_void_p:
        return C.c_void_p(self._ptr)

    def _get_native(self) -> "GPUArray":
        """The reference hands a pybind object to native ops; here the array IS the native handle."""
        return self

    # ------------------------------------------------------------------ host <-> device
    def to_numpy(self) -> np.ndarray:
        self._last_access = time.time()
        out = np.empty(self._shape, dtype=self._dtype.to_numpy_dtype())
        if self.nbytes:
            _hip.call("pgk_memcpy_d2h", out.ctypes.data_as(C.c_void_p), self._p, self.nbytes, None)
        return out

    def copy_from_numpy(self, arr: np.ndarray) -> None:
        """Overwrite the contents from a host array of identical byte size
        (reference: native GPUArray.copy_from_numpy)."""
        arr = np.ascontiguousarray(arr)
        if arr.nbytes != self.nbytes:
            raise ValueError(f"copy_from_numpy: {arr.nbytes} bytes into an array of {self.nbytes} bytes")
        if self.nbytes:
            _hip.call("pgk_memcpy_h2d", self._p, arr.ctypes.data_as(C.c_void_p), self.nbytes, None)

    def fill_zeros(self) -> None:
        if self.nbytes:
            _hip.call("pgk_memset", self._p, 0, self.nbytes, None)

    def is_contiguous(self) -> bool:
        return True

    def contiguous(self) -> "GPUArray":
        return self

    def clone(self) -> "GPUArray":
        out = GPUArray(self._shape, self._dtype)
        if self.nbytes:
            _hip.call("pgk_memcpy_d2d", out._p, self._p, self.nbytes, None)
        return out

    def __repr__(self) -> str:
        return f"GPUArray(shape={self._shape}, dtype={self._dtype.name}, backend=hip)"

    __str__ = __repr__

    def __del__(self) -> None:
        try:
            if self._owns_memory and self._ptr:
                _hip.call("pgk_free", C.c_void_p(self._ptr))
        except Exception:
            pass
        self._ptr = 0

    # ------------------------------------------------------------------ dtype conversion
    def astype(self, dtype: DataType) -> "GPUArray":
        """Reference: array.py:354-399 (bf16 RNE).  Float<->float casts run on the device."""
        if self._dtype == dtype:
            return self
        if self._dtype in _FLOATS and dtype in _FLOATS:
            out = GPUArray(self._shape, dtype)
            _hip.call("pgk_cast", self._p, self._dtype.code, out._p, dtype.code, self.size, None)
            return out
        from pygpukit_amd.core.factory import from_numpy

        np_data = self.to_numpy()
        if self._dtype == bfloat16:
            f32 = (np_data.astype(np.uint32) << 16).view(np.float32)
            return from_numpy(f32.astype(dtype.to_numpy_dtype()))
        if dtype == bfloat16:
            u = np.ascontiguousarray(np_data.astype(np.float32)).view(np.uint32)
            bits = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
            return from_numpy(bits)
        return from_numpy(np_data.astype(dtype.to_numpy_dtype()))

    # ------------------------------------------------------------------ zero-copy views
    def _view(self, offset_elements: int, new_shape) -> "GPUArray":
        new_shape = tuple(int(d) for d in new_shape)
        if offset_elements < 0 or offset_elements + _prod(new_shape) > self.size:
            raise ValueError(f"view of shape {new_shape} at offset {offset_elements} exceeds array of size {self.size}")
        root = self._source_ref if self._source_ref is not None else self
        return GPUArray(new_shape, self._dtype, self._ptr + offset_elements * self.itemsize, owns_memory=False,
                        _source_ref=root)

    def narrow(self, offset: int, length: int) -> "GPUArray":
        """View of `length` elements starting at element `offset` (array.py:401-453).  For a 2-D
        [rows, features] source the view is [rows, length] over CONTIGUOUS memory from `offset`, as in
        the reference (native GPUArray::narrow, memory.cpp:189-212): exact for rows == 1."""
        if self.ndim == 2:
            return self._view(offset, (self._shape[0], length))
        if self.ndim == 1:
            return self._view(offset, (length,))
        raise ValueError(f"narrow() only supports 1D or 2D arrays, got {self.ndim}D")

    def view(self, new_shape: tuple[int, ...]) -> "GPUArray":
        if _prod(new_shape) != self.size:
            raise ValueError(f"Cannot view array of size {self.size} as shape {tuple(new_shape)} (size {_prod(new_shape)})")
        return self._view(0, new_shape)

    def slice_rows(self, num_rows: int) -> "GPUArray":
        if self.ndim != 2:
            raise ValueError(f"slice_rows() requires 2D array, got {self.ndim}D")
        if num_rows > self._shape[0]:
            raise ValueError(f"num_rows ({num_rows}) exceeds batch dimension ({self._shape[0]})")
        return self._view(0, (num_rows, self._shape[1]))

    # ------------------------------------------------------------------ shape ops
    def transpose(self, *axes: int) -> "GPUArray":
        """array.py:550-632: device kernels for (1,0) and (1,0,2); other permutations go through the host."""
        from pygpukit_amd.core.factory import from_numpy

        if len(axes) == 0:
            axes = tuple(range(self.ndim - 1, -1, -1))
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if self.ndim == 2 and axes == (1, 0):
            from pygpukit_amd.ops.matmul import transpose as t2d

            return t2d(self)
        if self.ndim == 3 and axes == (1, 0, 2):
            from pygpukit_amd.ops.tensor import transpose_3d_021

            return transpose_3d_021(self)
        return from_numpy(np.ascontiguousarray(self.to_numpy().transpose(*axes)))

    @property
    def T(self) -> "GPUArray":
        return self.transpose()

    def reshape(self, *shape: int) -> "GPUArray":
        """array.py:639-712: a new array with the same elements (device copy)."""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        total = self.size
        neg = [i for i, d in enumerate(shape) if d == -1]
        if len(neg) > 1:
            raise ValueError("reshape: only one dimension can be -1")
        if neg:
            known = _prod(d for d in shape if d != -1)
            if known == 0 or total % known != 0:
                raise ValueError(f"reshape: cannot infer dimension, total size {total} not divisible by {known}")
            shape[neg[0]] = total // known
        if _prod(shape) != total:
            raise ValueError(f"reshape: cannot reshape array of size {total} into shape {tuple(shape)}")
        out = GPUArray(tuple(shape), self._dtype)
        if self.nbytes:
            _hip.call("pgk_memcpy_d2d", out._p, self._p, self.nbytes, None)
        return out

    def __getitem__(self, key) -> "GPUArray":
        """array.py:714-741: NumPy indexing through the host."""
        from pygpukit_amd.core.factory import from_numpy

        result = self.to_numpy()[key]
        if not isinstance(result, np.ndarray):
            result = np.array(result)
        out = from_numpy(np.ascontiguousarray(result))
        if self._dtype == bfloat16:
            out._dtype = bfloat16
        return out

    # ------------------------------------------------------------------ arithmetic
    def _scalar_op(self, scalar, op) -> "GPUArray":
        from pygpukit_amd.core.factory import from_numpy

        np_data = self.to_numpy()
        if self._dtype == bfloat16:
            f = (np_data.astype(np.uint32) << 16).view(np.float32)
            return from_numpy(op(f, scalar).astype(np.float32)).astype(bfloat16)
        return from_numpy(op(np_data, scalar).astype(np_data.dtype))

    def _binary(self, other, name: str, op):
        if isinstance(other, (int, float)):
            return self._scalar_op(other, op)
        if self.shape != other.shape:
            from pygpukit_amd.core.factory import from_numpy

            a, b = self.astype(float32).to_numpy(), other.astype(float32).to_numpy()
            return from_numpy(op(a, b).astype(np.float32)).astype(self._dtype)
        from pygpukit_amd.ops import elementwise

        return getattr(elementwise, name)(self, other)

    def __add__(self, other):
        return self._binary(other, "add", lambda a, b: a + b)

    def __radd__(self, other):
        return self._scalar_op(other, lambda a, b: b + a)

    def __sub__(self, other):
        return self._binary(other, "sub", lambda a, b: a - b)

    def __rsub__(self, other):
        return self._scalar_op(other, lambda a, b: b - a)

    def __mul__(self, other):
        return self._binary(other, "mul", lambda a, b: a * b)

    def __rmul__(self, other):
        return self._scalar_op(other, lambda a, b: b * a)

    def __truediv__(self, other):
        return self._binary(other, "div", lambda a, b: a / b)

    def __rtruediv__(self, other):
        return self._scalar_op(other, lambda a, b: b / a)

    def __matmul__(self, other):
        from pygpukit_amd.ops.matmul import matmul

        return matmul(self, other)
