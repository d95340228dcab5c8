"""lz4_amd — MI355X-native LZ4 block/frame hot path behind dart_lz4's API surface.

Host-buffer façade (same names as the reference): :mod:`lz4_amd.api`.
Device-resident batches over torch tensors: :mod:`lz4_amd.device`.
"""
from .api import (lz4BlockDecompressInto, lz4Compress, lz4Decompress, lz4FrameDecode, lz4FrameEncode,
                  lz4FrameEncodeWithOptions, xxh32)
from .exceptions import (Lz4BackendError, Lz4CorruptDataException, Lz4Exception, Lz4FormatException,
                         Lz4OutputLimitException, Lz4UnsupportedFeatureException, RangeError)
from .options import (Lz4CompressionLevel, Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions,
                      Lz4HcOptions)

__all__ = [
    "lz4Compress", "lz4Decompress", "lz4BlockDecompressInto", "lz4FrameEncode", "lz4FrameEncodeWithOptions",
    "lz4FrameDecode", "xxh32", "Lz4FrameOptions", "Lz4FrameBlockSize", "Lz4FrameCompression", "Lz4HcOptions",
    "Lz4CompressionLevel", "Lz4Exception", "Lz4FormatException", "Lz4CorruptDataException",
    "Lz4OutputLimitException", "Lz4UnsupportedFeatureException", "Lz4BackendError", "RangeError",
]
