"""Format helpers around the block/frame path, with the reference's names.

lz4FrameInfo (lib/src/frame/lz4_frame_info.dart), lz4SkippableFrameEncode and
lz4LegacyFrameEncode (lib/src/frame/lz4_frame_encoder.dart:27-89), the sized-block pair
(lib/src/block/lz4_sized_block.dart) and the streaming hasher Xxh32
(lib/src/xxhash/xxh32.dart:21-164), all over the C ABI of liblz4mi.so.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass
from typing import Optional

from . import _native as N
from .api import _ctx, lz4Compress, lz4Decompress
from .exceptions import Lz4FormatException, RangeError
from .options import Lz4CompressionLevel


@dataclass(frozen=True)
class Lz4FrameInfo:
    """Lz4FrameInfo (lib/src/frame/lz4_frame_info.dart:13-81)."""
    magic: int
    isSkippable: bool  # noqa: N815
    isLegacy: bool  # noqa: N815
    headerSize: int  # noqa: N815
    skippableSize: Optional[int] = None  # noqa: N815
    blockIndependence: bool = True  # noqa: N815
    blockChecksum: bool = False  # noqa: N815
    contentChecksum: bool = False  # noqa: N815
    contentSize: Optional[int] = None  # noqa: N815
    dictId: Optional[int] = None  # noqa: N815
    blockMaxSize: int = 0  # noqa: N815

    def __str__(self):  # toString :65-80
        if self.isSkippable:
            return f"Lz4FrameInfo(type: skippable, size: {self.skippableSize})"
        if self.isLegacy:
            return "Lz4FrameInfo(type: legacy)"

        def b(v):
            return "true" if v else "false"

        def n(v):
            return "null" if v is None else str(v)
        return ("Lz4FrameInfo(type: standard, "
                f"independent: {b(self.blockIndependence)}, blockChecksum: {b(self.blockChecksum)}, "
                f"contentChecksum: {b(self.contentChecksum)}, contentSize: {n(self.contentSize)}, "
                f"dictId: {n(self.dictId)}, blockMaxSize: {self.blockMaxSize})")


def lz4FrameInfo(src) -> Lz4FrameInfo:  # noqa: N802
    """lz4FrameInfo (lib/src/frame/lz4_frame_info.dart:89-199): header only, no payload."""
    src = bytes(src)
    info = N.FrameInfo()
    st = N.load().lz4mi_frame_info_parse(src, len(src), C.byref(info))
    if st:
        N.raise_status(st)
    if info.is_skippable:
        return Lz4FrameInfo(magic=info.magic, isSkippable=True, isLegacy=False,
                            skippableSize=info.skippable_size, headerSize=info.header_size)
    if info.is_legacy:
        return Lz4FrameInfo(magic=info.magic, isSkippable=False, isLegacy=True, headerSize=info.header_size)
    return Lz4FrameInfo(
        magic=info.magic, isSkippable=False, isLegacy=False, headerSize=info.header_size,
        blockIndependence=bool(info.block_independence), blockChecksum=bool(info.block_checksum),
        contentChecksum=bool(info.content_checksum),
        contentSize=info.content_size if info.has_content_size else None,
        dictId=info.dict_id if info.has_dict_id else None, blockMaxSize=info.block_max_size)


def lz4SkippableFrameEncode(data, *, index: int = 0) -> bytes:  # noqa: N802
    """lz4SkippableFrameEncode (lib/src/frame/lz4_frame_encoder.dart:27-46)."""
    data = bytes(data)
    if index < 0 or index > 15:
        raise RangeError(f"Invalid argument (index): {index}")
    cap = 8 + len(data)
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    st = N.load().lz4mi_skippable_frame_encode(data, len(data), index, dst, cap, C.byref(out))
    if st:
        N.raise_status(st, arg="data.length")
    return C.string_at(dst, out.value)


def lz4LegacyFrameEncode(src, *, acceleration: int = 1, device=None) -> bytes:  # noqa: N802
    """lz4LegacyFrameEncode (lib/src/frame/lz4_frame_encoder.dart:55-89)."""
    src = bytes(src)
    ctx = _ctx(device)
    L = N.load()
    cap = L.lz4mi_legacy_frame_bound(len(src))
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    st = L.lz4mi_legacy_frame_encode(ctx.handle, src, len(src), max(min(int(acceleration), 1 << 40), -1), dst,
                                     cap, C.byref(out))
    if st:
        N.raise_status(st, ctx.handle, arg="acceleration")
    return C.string_at(dst, out.value)


def lz4CompressWithSize(src, *, level: Lz4CompressionLevel = Lz4CompressionLevel.fast,  # noqa: N802
                        acceleration: int = 1, device=None) -> bytes:
    """lz4CompressWithSize (lib/src/block/lz4_sized_block.dart:15-31)."""
    src = bytes(src)
    compressed = lz4Compress(src, level=level, acceleration=acceleration, device=device)
    return struct.pack("<I", len(src) & 0xFFFFFFFF) + compressed


def lz4DecompressWithSize(src, *, device=None) -> bytes:  # noqa: N802
    """lz4DecompressWithSize (lib/src/block/lz4_sized_block.dart:39-55)."""
    src = bytes(src)
    if len(src) < 4:
        raise Lz4FormatException(N.load().lz4mi_status_message(N.FORMAT_SIZE_HEADER).decode())
    (size,) = struct.unpack_from("<I", src, 0)
    return lz4Decompress(src[4:], decompressedSize=size, device=device)


class Xxh32:
    """Xxh32 (lib/src/xxhash/xxh32.dart:21-164): streaming xxHash32; whole stripes run on the GPU."""

    def __init__(self, *, seed: int = 0, device=None):
        self._ctx = _ctx(device)
        self._s = N.XxhState()
        N.load().lz4mi_xxh32_state_init(C.byref(self._s), seed & 0xFFFFFFFF)

    def update(self, input, *, start: int = 0, end: Optional[int] = None):  # noqa: A002
        data = bytes(input)
        e = len(data) if end is None else end
        if start < 0 or start > len(data):
            raise RangeError(f"Invalid argument (start): {start}")
        if e < start or e > len(data):
            raise RangeError(f"Invalid argument (end): {e}")
        part = data[start:e]
        st = N.load().lz4mi_xxh32_state_update(self._ctx.handle, C.byref(self._s), part, len(part))
        if st:
            N.raise_status(st, self._ctx.handle)

    def digest(self) -> int:
        return N.load().lz4mi_xxh32_state_digest(C.byref(self._s))


__all__ = ["Lz4FrameInfo", "lz4FrameInfo", "lz4SkippableFrameEncode", "lz4LegacyFrameEncode",
           "lz4CompressWithSize", "lz4DecompressWithSize", "Xxh32"]
