"""safetensors access (reference: src/pygpukit/llm/safetensors.py:28-400).  The reference wraps a Rust mmap reader; here
the reader is C++ behind the C ABI (csrc/safetensors.hip): header parsed once, tensor bytes served from the mapping,
and `upload` copies file -> device without an intermediate host array.  LazyModelLoader / pool bookkeeping of the
reference (safetensors.py:407-680) is Rust-side state management and out of scope."""

from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from pygpukit_amd import _hip


class Dtype:
    Float32, Float16, BFloat16, Float64, Float8E4M3, Float8E5M2, Int32, Int64, Int16, Int8, UInt8, Bool = range(12)
    _NAMES = {0: "float32", 1: "float16", 2: "bfloat16", 3: "float64", 4: "float8_e4m3", 5: "float8_e5m2", 6: "int32", 7: "int64",
              8: "int16", 9: "int8", 10: "uint8", 11: "bool"}
    _SIZES = {0: 4, 1: 2, 2: 2, 3: 8, 4: 1, 5: 1, 6: 4, 7: 8, 8: 2, 9: 1, 10: 1, 11: 1}

    @classmethod
    def element_size(cls, dtype: int) -> int:
        return cls._SIZES.get(dtype, 0)

    @classmethod
    def name(cls, dtype: int) -> str:
        return cls._NAMES.get(dtype, "unknown")


class TensorInfo:
    def __init__(self, name: str, dtype: int, shape: list[int], offset: int, size_bytes: int):
        self.name, self.dtype, self.shape, self.offset, self.size_bytes = name, dtype, list(shape), offset, size_bytes

    @property
    def numel(self) -> int:
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def dtype_name(self) -> str:
        return Dtype.name(self.dtype)

    def __repr__(self) -> str:
        return f"TensorInfo(name='{self.name}', dtype={self.dtype_name}, shape={self.shape}, size_bytes={self.size_bytes})"


class SafeTensorsFile:
    """Memory-mapped single .safetensors file."""

    def __init__(self, path: str):
        self._path = path
        self._h = C.c_void_p()
        lib = _hip.load()
        try:
            _hip.call("pgk_st_open", os.fsencode(path), C.byref(self._h))
        except _hip.PgkError as e:
            if not os.path.exists(path):
                raise FileNotFoundError(path) from None
            raise ValueError(str(e)) from None
        self._names = [lib.pgk_st_tensor_name(self._h, i).decode() for i in range(lib.pgk_st_num_tensors(self._h))]
        self._name_set = set(self._names)

    @property
    def tensor_names(self) -> list[str]:
        return list(self._names)

    @property
    def file_size(self) -> int:
        return int(_hip.load().pgk_st_file_size(self._h))

    @property
    def num_tensors(self) -> int:
        return len(self._names)

    def tensor_info(self, name: str) -> TensorInfo:
        if name not in self._name_set:
            raise KeyError(f"Tensor '{name}' not found")
        dt, nd, off, nb = C.c_int(), C.c_int(), C.c_uint64(), C.c_uint64()
        shape = (C.c_int64 * 8)()
        _hip.call("pgk_st_tensor_info", self._h, name.encode(), C.byref(dt), C.byref(nd), shape, C.byref(off), C.byref(nb))
        return TensorInfo(name, dt.value, [int(shape[i]) for i in range(nd.value)], off.value, nb.value)

    def tensor_data_ptr(self, name: str) -> tuple[int, int]:
        """(address inside the mapping, size in bytes): valid while this object lives."""
        if name not in self._name_set:
            raise KeyError(f"Tensor '{name}' not found")
        p, n = C.c_void_p(), C.c_uint64()
        _hip.call("pgk_st_tensor_data", self._h, name.encode(), C.byref(p), C.byref(n))
        return int(p.value or 0), int(n.value)

    def tensor_bytes(self, name: str) -> bytes:
        p, n = self.tensor_data_ptr(name)
        return C.string_at(p, n) if n else b""

    def tensor_numpy(self, name: str) -> np.ndarray:
        """Zero-copy read-only view in the stored dtype (bf16 / fp8 as uint16 / uint8)."""
        info = self.tensor_info(name)
        npdt = {0: np.float32, 1: np.float16, 2: np.uint16, 3: np.float64, 4: np.uint8, 5: np.uint8, 6: np.int32, 7: np.int64,
                8: np.int16, 9: np.int8, 10: np.uint8, 11: np.bool_}[info.dtype]
        p, n = self.tensor_data_ptr(name)
        if n == 0:
            return np.zeros(info.shape, npdt)
        buf = (C.c_char * n).from_address(p)
        a = np.frombuffer(buf, dtype=npdt).reshape(info.shape)
        a.flags.writeable = False
        return a          # valid while this SafeTensorsFile is alive (the mapping is not reference-counted by NumPy)

    def tensor_as_f32(self, name: str) -> np.ndarray:
        info = self.tensor_info(name)
        a = self.tensor_numpy(name)
        if info.dtype == Dtype.BFloat16:
            return (a.astype(np.uint32) << 16).view(np.float32)
        if info.dtype in (Dtype.Float8E4M3, Dtype.Float8E5M2):
            raise ValueError(f"Unsupported dtype for float32 conversion: {info.dtype_name}")
        return a.astype(np.float32)

    def upload(self, name: str, dst) -> None:
        """Copy the tensor's bytes from the mapping straight into the GPUArray `dst` (same byte size)."""
        _hip.call("pgk_st_upload", self._h, name.encode(), dst._p, dst.nbytes, None)

    def __len__(self) -> int:
        return self.num_tensors

    def __contains__(self, name: str) -> bool:
        return name in self._name_set

    def __repr__(self) -> str:
        return f"SafeTensorsFile(num_tensors={self.num_tensors}, file_size={self.file_size})"

    def __del__(self):
        try:
            if self._h.value:
                _hip.load().pgk_st_close(self._h)
                self._h = C.c_void_p()
        except Exception:  # noqa: BLE001
            pass


class ShardedSafeTensorsFile:
    """model.safetensors.index.json + shards, opened lazily (safetensors.py:237-380)."""

    def __init__(self, index_json_path: str):
        with open(index_json_path, encoding="utf-8") as f:
            index = json.load(f)
        self._weight_map: dict[str, str] = index["weight_map"]
        self._dir = os.path.dirname(os.path.abspath(index_json_path))
        self._shards: dict[str, SafeTensorsFile] = {}
        self._names = list(self._weight_map)

    def _get_shard(self, shard_file: str) -> SafeTensorsFile:
        if shard_file not in self._shards:
            self._shards[shard_file] = SafeTensorsFile(os.path.join(self._dir, shard_file))
        return self._shards[shard_file]

    def _of(self, name: str) -> SafeTensorsFile:
        if name not in self._weight_map:
            raise KeyError(f"Tensor '{name}' not found")
        return self._get_shard(self._weight_map[name])

    tensor_names = property(lambda self: list(self._names))
    num_tensors = property(lambda self: len(self._names))

    @property
    def file_size(self) -> int:
        return sum(os.path.getsize(os.path.join(self._dir, s)) for s in set(self._weight_map.values()))

    def tensor_info(self, name): return self._of(name).tensor_info(name)          # noqa: E704
    def tensor_bytes(self, name): return self._of(name).tensor_bytes(name)        # noqa: E704
    def tensor_numpy(self, name): return self._of(name).tensor_numpy(name)        # noqa: E704
    def tensor_as_f32(self, name): return self._of(name).tensor_as_f32(name)      # noqa: E704
    def tensor_data_ptr(self, name): return self._of(name).tensor_data_ptr(name)  # noqa: E704
    def upload(self, name, dst): return self._of(name).upload(name, dst)          # noqa: E704
    def __len__(self): return len(self._names)                                    # noqa: E704
    def __contains__(self, name): return name in self._weight_map                 # noqa: E704

    def __repr__(self) -> str:
        return f"ShardedSafeTensorsFile(num_tensors={self.num_tensors}, num_shards={len(set(self._weight_map.values()))})"


def load_safetensors(path: str) -> SafeTensorsFile | ShardedSafeTensorsFile:
    """A .safetensors file or a model.safetensors.index.json (safetensors.py:383-400)."""
    return ShardedSafeTensorsFile(path) if path.endswith(".index.json") else SafeTensorsFile(path)


def save_safetensors(path: str, tensors: dict[str, tuple[np.ndarray, str]], metadata: dict[str, str] | None = None) -> None:
    """Write a .safetensors file from {name: (array, dtype tag)}; bf16 / fp8 data are passed as uint16 / uint8 arrays
    with tag "BF16" / "F8_E4M3".  (Test and export helper: the reference only reads.)"""
    header, blobs, off = {}, [], 0
    if metadata:
        header["__metadata__"] = metadata
    for name, (arr, tag) in tensors.items():
        b = np.ascontiguousarray(arr).tobytes()
        header[name] = {"dtype": tag, "shape": list(arr.shape), "data_offsets": [off, off + len(b)]}
        blobs.append(b)
        off += len(b)
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(len(hj).to_bytes(8, "little"))
        f.write(hj)
        for b in blobs:
            f.write(b)
