from .base import DecodeStrategy
from .batch import DecodeBatch
from .m1 import DecodeM1
from .m1_graph import DecodeM1Graph

__all__ = ["DecodeStrategy", "DecodeM1", "DecodeM1Graph", "DecodeBatch"]
