"""Drop-in façade with the reference's names and argument meaning (lib/dart_lz4.dart).

Every function moves the bytes to the GPU, runs the HIP kernels behind the C ABI
(include/lz4mi.h) and copies the result back; errors are re-raised as the
reference's exception class with the reference's message.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _native as N
from .exceptions import Lz4Exception, RangeError
from .options import (Lz4CompressionLevel, Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions,
                      Lz4HcOptions)

Lz4DictionaryResolver = Callable[[int], Optional[bytes]]  # lib/src/frame/lz4_frame_options.dart:8


def _ctx(device):
    return N.default_context(0 if device is None else device)


def lz4Compress(src, *, level: Lz4CompressionLevel = Lz4CompressionLevel.fast, acceleration: int = 1,  # noqa: N802
                hcOptions: Lz4HcOptions | None = None, dictionary=None, device=None) -> bytes:  # noqa: N803
    """lz4Compress (lib/dart_lz4.dart:60-72).  `dictionary` is the block encoder's optional
    parameter (lib/src/block/lz4_block_encoder.dart:24-28)."""
    src = bytes(src)
    ctx = _ctx(device)
    L = N.load()
    cap = L.lz4mi_compress_bound(len(src)) + 16
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    d = bytes(dictionary) if dictionary is not None else None
    if level == Lz4CompressionLevel.fast:
        if acceleration < 1:
            raise RangeError(f"Invalid argument (acceleration): {acceleration}")
        st = L.lz4mi_block_compress_fast(ctx.handle, src, len(src), d, len(d) if d else 0,
                                         min(int(acceleration), 1 << 40), dst, cap, C.byref(out))
    else:
        depth = (hcOptions or Lz4HcOptions()).maxSearchDepth
        st = L.lz4mi_block_compress_hc(ctx.handle, src, len(src), d, len(d) if d else 0, depth, dst, cap,
                                       C.byref(out))
    if st:
        N.raise_status(st, ctx.handle)
    return C.string_at(dst, out.value)


def lz4Decompress(src, *, decompressedSize: int, device=None) -> bytes:  # noqa: N802,N803
    """lz4Decompress (lib/dart_lz4.dart:80-85)."""
    if decompressedSize < 0:
        raise RangeError(f"Invalid argument (decompressedSize): {decompressedSize}")
    src = bytes(src)
    ctx = _ctx(device)
    dst = C.create_string_buffer(max(decompressedSize, 1))
    st = N.load().lz4mi_block_decompress(ctx.handle, src, len(src), dst, decompressedSize)
    if st:
        N.raise_status(st, ctx.handle)
    return C.string_at(dst, decompressedSize)


def lz4BlockDecompressInto(src, prefix: bytes = b"", maxLength: int | None = None, device=None) -> bytes:  # noqa: N802,N803
    """lz4BlockDecompressInto into ByteWriter(maxLength) pre-filled with `prefix`
    (lib/src/block/lz4_block_decoder.dart:79-119); returns the writer's bytes."""
    src = bytes(src)
    prefix = bytes(prefix)
    if maxLength is None:
        maxLength = len(prefix) + 255 * len(src) + 64
    ctx = _ctx(device)
    buf = C.create_string_buffer(max(maxLength, len(prefix), 1))
    C.memmove(buf, prefix, len(prefix))
    out = C.c_size_t(0)
    st = N.load().lz4mi_block_decompress_into(ctx.handle, src, len(src), buf, len(prefix), maxLength, C.byref(out))
    if st:
        N.raise_status(st, ctx.handle)
    return C.string_at(buf, out.value)


def xxh32(data, seed: int = 0, device=None) -> int:
    """xxh32 (lib/src/xxhash/xxh32.dart:166-240), computed on the GPU."""
    data = bytes(data)
    ctx = _ctx(device)
    out = C.c_uint32(0)
    st = N.load().lz4mi_xxh32(ctx.handle, data, len(data), seed & 0xFFFFFFFF, C.byref(out))
    if st:
        N.raise_status(st, ctx.handle)
    return out.value


def _c_options(options: Lz4FrameOptions) -> N.FrameOptions:
    o = N.FrameOptions()
    N.load().lz4mi_frame_options_default(C.byref(o))
    o.block_size_id = options.blockSize.bdId
    o.block_independence = int(options.blockIndependence)
    o.block_checksum = int(options.blockChecksum)
    o.content_checksum = int(options.contentChecksum)
    o.has_content_size = int(options.contentSize is not None)
    o.content_size = options.contentSize or 0
    o.has_dict_id = int(options.dictId is not None)
    o.dict_id = options.dictId or 0
    o.compression = options.compression.value
    o.acceleration = min(options.acceleration, 0x7FFFFFFF)
    o.hc_max_search_depth = min((options.hcOptions or Lz4HcOptions()).maxSearchDepth, 0x7FFFFFFF)
    o.spec_conformant = int(options.specConformantBlocks)
    return o


def lz4FrameEncodeWithOptions(src, *, options: Lz4FrameOptions, dictionary=None, device=None) -> bytes:  # noqa: N802
    """lz4FrameEncodeWithOptions (lib/dart_lz4.dart:110-117)."""
    src = bytes(src)
    ctx = _ctx(device)
    L = N.load()
    o = _c_options(options)
    cap = L.lz4mi_frame_bound(len(src), C.byref(o))
    dst = np.empty(max(cap, 1), dtype=np.uint8)  # not zero-filled: its pages are first touched by the copy threads
    out = C.c_size_t(0)
    d = bytes(dictionary) if dictionary is not None else None
    st = L.lz4mi_frame_encode(ctx.handle, src, len(src), C.byref(o), d, len(d) if d else 0,
                              C.c_void_p(dst.ctypes.data), cap, C.byref(out))
    if st:
        N.raise_status(st, ctx.handle)
    return dst[:out.value].tobytes()  # one copy of the frame, not of the whole bound


def lz4FrameEncode(src, *, acceleration: int = 1, dictionary=None, device=None) -> bytes:  # noqa: N802
    """lz4FrameEncode (lib/dart_lz4.dart:97-104)."""
    return lz4FrameEncodeWithOptions(src, options=Lz4FrameOptions(acceleration=acceleration),
                                     dictionary=dictionary, device=device)


def lz4FrameDecode(src, *, maxOutputBytes: int | None = None,  # noqa: N802,N803
                   dictionaryResolver: Lz4DictionaryResolver | None = None, device=None) -> bytes:  # noqa: N803
    """lz4FrameDecode (lib/dart_lz4.dart:126-136): all concatenated frames of `src`.

    The loop over frames and the resolver callback run here, as in
    _Lz4FrameDecoder.decodeAll (lib/src/frame/lz4_frame_decoder.dart:45-76)."""
    src = bytes(src)
    ctx = _ctx(device)
    L = N.load()
    addr = C.cast(C.c_char_p(src), C.c_void_p).value or 0  # the bytes object itself, no copy
    parts: list[bytes] = []
    produced_total = 0
    pos = 0
    dict_bytes: bytes | None = None  # _dict, never cleared between frames (:34,235)
    while pos < len(src):
        bound = C.c_size_t(0)
        L.lz4mi_frame_decode_bound(C.c_void_p(addr + pos), len(src) - pos, C.byref(bound))
        remaining = -1 if maxOutputBytes is None else maxOutputBytes - produced_total
        cap = bound.value if remaining < 0 else min(bound.value, remaining + 1)
        dst = np.empty(max(cap, 1), dtype=np.uint8)  # not zero-filled
        consumed, produced, need = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
        state = 0
        frame_dict = dict_bytes
        while True:
            st = L.lz4mi_frame_decode_next(ctx.handle, C.c_void_p(addr + pos), len(src) - pos, C.byref(consumed),
                                           frame_dict, len(frame_dict) if frame_dict else 0, state,
                                           int(dict_bytes is not None), remaining, C.c_void_p(dst.ctypes.data), cap,
                                           C.byref(produced), C.byref(need))
            if st == N.NEED_DICT and state == 0:
                if dictionaryResolver is None:
                    state = 2
                else:
                    d = dictionaryResolver(need.value)
                    if d is None:
                        state = 3
                    else:
                        frame_dict = bytes(d)
                        state = 1
                continue
            break
        if st:
            N.raise_status(st, ctx.handle, dict_id=need.value)
        if state == 1:
            dict_bytes = frame_dict
        parts.append(dst[:produced.value].tobytes())
        produced_total += produced.value
        pos += consumed.value
    return parts[0] if len(parts) == 1 else b"".join(parts)


__all__ = [
    "lz4Compress", "lz4Decompress", "lz4BlockDecompressInto", "lz4FrameEncode", "lz4FrameEncodeWithOptions",
    "lz4FrameDecode", "xxh32", "Lz4FrameOptions", "Lz4FrameBlockSize", "Lz4FrameCompression", "Lz4HcOptions",
    "Lz4CompressionLevel", "Lz4DictionaryResolver", "Lz4Exception",
]
