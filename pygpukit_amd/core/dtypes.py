"""Element types of a GPUArray.  Same public surface as the reference's
src/pygpukit/core/dtypes.py:10-123 (DataType, DataTypeKind, float32 ... int4, from_string,
from_numpy_dtype with uint16 standing for bfloat16); `code` is the pgk_dtype enum value of
include/pgk_hip.h, which follows the reference's native enum order
(native/bindings/core_bindings.cpp:19-30)."""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Any

import numpy as np


class DataTypeKind(Enum):
    FLOAT64 = "float64"
    FLOAT32 = "float32"
    FLOAT16 = "float16"
    BFLOAT16 = "bfloat16"
    INT64 = "int64"
    INT32 = "int32"
    INT16 = "int16"
    INT8 = "int8"
    UINT8 = "uint8"
    INT4 = "int4"


_NUMPY_OF = {
    DataTypeKind.FLOAT64: np.float64, DataTypeKind.FLOAT32: np.float32, DataTypeKind.FLOAT16: np.float16,
    DataTypeKind.BFLOAT16: np.uint16,  # NumPy has no bfloat16: raw 16-bit words
    DataTypeKind.INT64: np.int64, DataTypeKind.INT32: np.int32, DataTypeKind.INT16: np.int16,
    DataTypeKind.INT8: np.int8, DataTypeKind.UINT8: np.uint8, DataTypeKind.INT4: np.uint8,
}


@dataclass(frozen=True)
class DataType:
    kind: DataTypeKind
    itemsize: int
    name: str
    code: int = -1  # pgk_dtype

    def __str__(self) -> str:
        return self.name

    def __repr__(self) -> str:
        return f"DataType({self.name})"

    def to_numpy_dtype(self) -> Any:
        return np.dtype(_NUMPY_OF[self.kind])

    @staticmethod
    def from_numpy_dtype(dtype: Any) -> "DataType":
        name = np.dtype(dtype).name
        if name == "uint16":  # storage type of bfloat16
            return bfloat16
        if name in _BY_NAME and name not in ("bfloat16", "int4"):
            return _BY_NAME[name]
        raise ValueError(f"Unsupported dtype: {dtype}")

    @staticmethod
    def from_string(name: str) -> "DataType":
        if name not in _BY_NAME:
            raise ValueError(f"Unsupported dtype string: {name}")
        return _BY_NAME[name]


float64 = DataType(DataTypeKind.FLOAT64, 8, "float64", 0)
float32 = DataType(DataTypeKind.FLOAT32, 4, "float32", 1)
float16 = DataType(DataTypeKind.FLOAT16, 2, "float16", 2)
bfloat16 = DataType(DataTypeKind.BFLOAT16, 2, "bfloat16", 3)
int64 = DataType(DataTypeKind.INT64, 8, "int64", 4)
int32 = DataType(DataTypeKind.INT32, 4, "int32", 5)
int16 = DataType(DataTypeKind.INT16, 2, "int16", 6)
int8 = DataType(DataTypeKind.INT8, 1, "int8", 7)
uint8 = DataType(DataTypeKind.UINT8, 1, "uint8", 8)
int4 = DataType(DataTypeKind.INT4, 1, "int4", 9)

_BY_NAME = {d.name: d for d in (float64, float32, float16, bfloat16, int64, int32, int16, int8, uint8, int4)}
FLOAT_DTYPES = (float32, float16, bfloat16)


def as_dtype(dtype: "str | DataType") -> DataType:
    return DataType.from_string(dtype) if isinstance(dtype, str) else dtype
