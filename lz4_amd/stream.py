"""Streaming frame transformers with the reference's names.

lz4FrameEncoderTransformer / lz4FrameEncoderTransformerWithOptions
(lib/src/frame/lz4_frame_stream_encoder.dart:13-236) and lz4FrameDecoderTransformer
(lib/src/frame/lz4_frame_stream_decoder.dart:15-45).  A Dart StreamTransformer maps a stream
of chunks to a stream of chunks; here ``transformer.bind(iterable)`` (or calling the
transformer) returns a generator, and an exception the reference would add to the stream is
raised from the generator at the same point of the output sequence.

The chunk loop is host code; the buffering, the 64 KiB history and every block run behind
the C ABI (lz4mi_frame_encoder_*, lz4mi_frame_decoder_*), several blocks per launch.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator, Optional

from . import _native as N
from .api import Lz4DictionaryResolver, _c_options, _ctx
from .options import Lz4FrameOptions


class _Transformer:
    def __init__(self, fn):
        self._fn = fn

    def bind(self, stream: Iterable[bytes]) -> Iterator[bytes]:
        return self._fn(stream)

    __call__ = bind


def _split_records(buf: bytes, block_checksum: bool):
    """One piece per block record, as the reference yields them (:195,:216)."""
    pos = 0
    while pos < len(buf):
        word = int.from_bytes(buf[pos:pos + 4], "little")
        n = 4 + (word & 0x7FFFFFFF) + (4 if block_checksum else 0)
        yield buf[pos:pos + n]
        pos += n


def lz4FrameEncoderTransformerWithOptions(*, options: Lz4FrameOptions, dictionary=None,  # noqa: N802
                                          device=None) -> _Transformer:
    """lz4FrameEncoderTransformerWithOptions (lib/src/frame/lz4_frame_stream_encoder.dart:25-236)."""
    dict_bytes = bytes(dictionary) if dictionary is not None else None

    def run(stream):
        ctx = _ctx(device)
        L = N.load()
        o = _c_options(options)
        hdr = C.create_string_buffer(32)
        out = C.c_size_t(0)
        st = L.lz4mi_frame_header(C.byref(o), hdr, 32, C.byref(out))
        if st:
            N.raise_status(st, ctx.handle)
        yield C.string_at(hdr, out.value)  # :42-47
        h = C.c_void_p()
        st = L.lz4mi_frame_encoder_create(ctx.handle, C.byref(o), dict_bytes, len(dict_bytes) if dict_bytes else 0,
                                          C.byref(h))
        try:
            if st:
                N.raise_status(st, ctx.handle)
            for chunk in stream:  # :158-201
                data = chunk if isinstance(chunk, bytes) else bytes(chunk)
                if not data:
                    continue
                cap = L.lz4mi_frame_encoder_bound(h, len(data))
                dst = C.create_string_buffer(cap)
                st = L.lz4mi_frame_encoder_update(h, data, len(data), dst, cap, C.byref(out))
                if st:
                    N.raise_status(st, ctx.handle)
                if out.value:
                    yield from _split_records(C.string_at(dst, out.value), options.blockChecksum)
            cap = L.lz4mi_frame_encoder_bound(h, 0)
            dst = C.create_string_buffer(cap)
            st = L.lz4mi_frame_encoder_finish(h, dst, cap, C.byref(out))
            tail = C.string_at(dst, out.value)
            if st:  # the last block is yielded before the contentSize check throws (:203-226)
                yield from _split_records(tail, options.blockChecksum)
                N.raise_status(st, ctx.handle)
            trailer = 4 + (4 if options.contentChecksum else 0)
            yield from _split_records(tail[:len(tail) - trailer], options.blockChecksum)
            yield tail[len(tail) - trailer:len(tail) - trailer + 4]  # end mark :228
            if options.contentChecksum:
                yield tail[-4:]  # :230-234
        finally:
            if h:
                L.lz4mi_frame_encoder_destroy(h)

    return _Transformer(run)


def lz4FrameEncoderTransformer(*, acceleration: int = 1, dictionary=None, device=None) -> _Transformer:  # noqa: N802
    """lz4FrameEncoderTransformer (lib/src/frame/lz4_frame_stream_encoder.dart:13-23)."""
    return lz4FrameEncoderTransformerWithOptions(options=Lz4FrameOptions(acceleration=acceleration),
                                                 dictionary=dictionary, device=device)


def lz4FrameDecoderTransformer(*, maxOutputBytes: Optional[int] = None,  # noqa: N802,N803
                               dictionaryResolver: Optional[Lz4DictionaryResolver] = None,  # noqa: N803
                               device=None, outputCapacity: int = 32 << 20) -> _Transformer:  # noqa: N803
    """lz4FrameDecoderTransformer (lib/src/frame/lz4_frame_stream_decoder.dart:15-45).

    outputCapacity bounds how many blocks one launch may decode (capacity / blockMaxSize)."""

    def run(stream):
        ctx = _ctx(device)
        L = N.load()
        h = C.c_void_p()
        st = L.lz4mi_frame_decoder_create(ctx.handle, -1 if maxOutputBytes is None else maxOutputBytes, C.byref(h))
        if st:
            N.raise_status(st, ctx.handle)
        cap = max(int(outputCapacity), 1)
        dst = C.create_string_buffer(cap)
        out, need = C.c_size_t(0), C.c_uint32(0)

        def drain():  # `while (true) { out = decoder.decodeNext(); ... }` :27-33
            nonlocal cap, dst
            while True:
                st = L.lz4mi_frame_decoder_next(h, dst, cap, C.byref(out), C.byref(need))
                if st == N.OK:
                    yield C.string_at(dst, out.value)
                elif st == N.STREAM_NEED_INPUT:
                    return
                elif st == N.DST_TOO_SMALL:
                    cap = max(out.value, cap * 2)
                    dst = C.create_string_buffer(cap)
                elif st == N.NEED_DICT:  # :237-248
                    if dictionaryResolver is None:
                        L.lz4mi_frame_decoder_set_dictionary(h, None, 0, 2)
                    else:
                        d = dictionaryResolver(need.value)
                        if d is None:
                            L.lz4mi_frame_decoder_set_dictionary(h, None, 0, 3)
                        else:
                            d = bytes(d)
                            L.lz4mi_frame_decoder_set_dictionary(h, d, len(d), 1)
                else:
                    N.raise_status(st, ctx.handle, dict_id=need.value)

        try:
            for chunk in stream:
                data = chunk if isinstance(chunk, bytes) else bytes(chunk)
                st = L.lz4mi_frame_decoder_feed(h, data, len(data))
                if st:
                    N.raise_status(st, ctx.handle)
                yield from drain()
            st = L.lz4mi_frame_decoder_finish(h)  # :36
            if st:
                N.raise_status(st, ctx.handle, dict_id=need.value)
            yield from drain()
        finally:
            L.lz4mi_frame_decoder_destroy(h)

    return _Transformer(run)


__all__ = ["lz4FrameEncoderTransformer", "lz4FrameEncoderTransformerWithOptions", "lz4FrameDecoderTransformer"]
