"""lz4_amd — MI355X-native LZ4 block/frame hot path behind dart_lz4's API surface.

Host-buffer façade (same names as the reference): :mod:`lz4_amd.api`.
Format helpers (frame info, skippable / legacy frames, sized blocks, streaming Xxh32):
:mod:`lz4_amd.formats`.  Chunk-stream transformers: :mod:`lz4_amd.stream`.
Device-resident batches over torch tensors: :mod:`lz4_amd.device`.
"""
from .api import (lz4BlockDecompressInto, lz4Compress, lz4Decompress, lz4FrameDecode, lz4FrameEncode,
                  lz4FrameEncodeWithOptions, xxh32)
from .exceptions import (Lz4BackendError, Lz4CorruptDataException, Lz4Exception, Lz4FormatException,
                         Lz4OutputLimitException, Lz4UnsupportedFeatureException, RangeError)
from .formats import (Lz4FrameInfo, Xxh32, lz4CompressWithSize, lz4DecompressWithSize, lz4FrameInfo,
                      lz4LegacyFrameEncode, lz4SkippableFrameEncode)
from .options import (Lz4CompressionLevel, Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions,
                      Lz4HcOptions)
from .stream import (lz4FrameDecoderTransformer, lz4FrameEncoderTransformer,
                     lz4FrameEncoderTransformerWithOptions)

# façade names (lib/dart_lz4.dart:147-220)
lz4FrameDecoder = lz4FrameDecoderTransformer
lz4FrameEncoder = lz4FrameEncoderTransformer
lz4FrameEncoderWithOptions = lz4FrameEncoderTransformerWithOptions
lz4SkippableEncode = lz4SkippableFrameEncode
lz4LegacyEncode = lz4LegacyFrameEncode

__all__ = [
    "lz4FrameDecoder", "lz4FrameEncoder", "lz4FrameEncoderWithOptions", "lz4SkippableEncode", "lz4LegacyEncode",
    "lz4FrameInfo", "Lz4FrameInfo", "lz4SkippableFrameEncode", "lz4LegacyFrameEncode", "lz4CompressWithSize",
    "lz4DecompressWithSize", "Xxh32", "lz4FrameEncoderTransformer", "lz4FrameEncoderTransformerWithOptions",
    "lz4FrameDecoderTransformer",
    "lz4Compress", "lz4Decompress", "lz4BlockDecompressInto", "lz4FrameEncode", "lz4FrameEncodeWithOptions",
    "lz4FrameDecode", "xxh32", "Lz4FrameOptions", "Lz4FrameBlockSize", "Lz4FrameCompression", "Lz4HcOptions",
    "Lz4CompressionLevel", "Lz4Exception", "Lz4FormatException", "Lz4CorruptDataException",
    "Lz4OutputLimitException", "Lz4UnsupportedFeatureException", "Lz4BackendError", "RangeError",
]
