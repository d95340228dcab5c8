"""Option carriers with the reference's names, defaults and validation.

Lz4FrameOptions / Lz4FrameBlockSize / Lz4FrameCompression:
lib/src/frame/lz4_frame_options.dart:14-129.  Lz4HcOptions:
lib/src/hc/lz4_hc_options.dart:2-17.  Lz4CompressionLevel: lib/dart_lz4.dart:39-45.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Optional

from .exceptions import RangeError


class Lz4CompressionLevel(enum.Enum):
    fast = 0
    hc = 1


class Lz4FrameBlockSize(enum.Enum):
    k64KB = 4
    k256KB = 5
    k1MB = 6
    k4MB = 7

    @property
    def bdId(self) -> int:  # noqa: N802 (reference name)
        return self.value

    @property
    def maxBytes(self) -> int:  # noqa: N802
        return {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}[self.value]


class Lz4FrameCompression(enum.Enum):
    fast = 0
    hc = 1


class Lz4HcOptions:
    def __init__(self, maxSearchDepth: int = 64):  # noqa: N803
        if maxSearchDepth < 1:
            raise RangeError(f"Invalid argument (maxSearchDepth): {maxSearchDepth}")
        self.maxSearchDepth = maxSearchDepth


@dataclass
class Lz4FrameOptions:
    blockSize: Lz4FrameBlockSize = Lz4FrameBlockSize.k4MB  # noqa: N815
    blockIndependence: bool = True  # noqa: N815
    blockChecksum: bool = False  # noqa: N815
    contentChecksum: bool = False  # noqa: N815
    contentSize: Optional[int] = None  # noqa: N815
    dictId: Optional[int] = None  # noqa: N815
    compression: Lz4FrameCompression = Lz4FrameCompression.fast
    acceleration: int = 1
    hcOptions: Optional[Lz4HcOptions] = None  # noqa: N815
    # Not a reference option: blocks follow the lz4 block format's end-of-block rules, so any
    # LZ4 decoder (the lz4 CLI included) accepts the frame; output differs from the reference's.
    specConformantBlocks: bool = False  # noqa: N815

    def __post_init__(self):
        if self.acceleration < 1:
            raise RangeError(f"Invalid argument (acceleration): {self.acceleration}")
        if self.contentSize is not None and self.contentSize < 0:
            raise RangeError(f"Invalid argument (contentSize): {self.contentSize}")
        if self.dictId is not None and (self.dictId < 0 or self.dictId > 0xFFFFFFFF):
            raise RangeError(f"Invalid argument (dictId): {self.dictId}")
