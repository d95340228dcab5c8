"""ctypes binding of the CPU oracle (liblz4ref.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the lz4_amd package.  The oracle is a
scalar C restatement of dart_lz4's block/frame path (see lz4ref.h for how it is
pinned to the reference's own vectors).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblz4ref.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "lz4ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liblz4ref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Lz4RefError(Exception):
    """Carries the Dart exception class name and message the reference would raise."""

    def __init__(self, status: int, type_name: str, message: str):
        super().__init__(f"{type_name}: {message}")
        self.status = status
        self.type_name = type_name
        self.message = message


class FrameOptions(C.Structure):
    _fields_ = [
        ("block_size_id", C.c_int32),
        ("block_independence", C.c_int32),
        ("block_checksum", C.c_int32),
        ("content_checksum", C.c_int32),
        ("has_content_size", C.c_int32),
        ("has_dict_id", C.c_int32),
        ("compression", C.c_int32),
        ("acceleration", C.c_int32),
        ("hc_max_search_depth", C.c_int32),
        ("dict_id", C.c_uint32),
        ("content_size", C.c_uint64),
        ("spec_conformant", C.c_int32),
        ("reserved", C.c_int32),
    ]


class _XxhState(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32), ("v1", C.c_uint32), ("v2", C.c_uint32), ("v3", C.c_uint32),
        ("v4", C.c_uint32), ("total_len", C.c_uint64), ("mem", C.c_uint8 * 16),
        ("mem_size", C.c_uint32),
    ]


class FrameInfo(C.Structure):
    _fields_ = [
        ("magic", C.c_uint32), ("is_skippable", C.c_int32), ("is_legacy", C.c_int32),
        ("block_independence", C.c_int32), ("block_checksum", C.c_int32), ("content_checksum", C.c_int32),
        ("has_content_size", C.c_int32), ("has_dict_id", C.c_int32), ("skippable_size", C.c_uint32),
        ("dict_id", C.c_uint32), ("block_max_size", C.c_uint32), ("header_size", C.c_uint32),
        ("content_size", C.c_uint64),
    ]


_RESOLVER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.lz4ref_status_type.restype = C.c_char_p
        L.lz4ref_status_message.restype = C.c_char_p
        L.lz4ref_compress_bound.restype = C.c_size_t
        L.lz4ref_compress_bound.argtypes = [C.c_size_t]
        L.lz4ref_block_compress_fast.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                                 C.c_int64, C.c_void_p, C.c_size_t,
                                                 C.POINTER(C.c_size_t)]
        L.lz4ref_block_compress_hc.argtypes = L.lz4ref_block_compress_fast.argtypes
        L.lz4ref_block_compress_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                               C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                                               C.POINTER(C.c_size_t)]
        L.lz4ref_block_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int64]
        L.lz4ref_block_decompress_into.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                   C.c_size_t, C.POINTER(C.c_size_t)]
        L.lz4ref_block_conformant.argtypes = [C.c_char_p, C.c_size_t]
        L.lz4ref_blocks_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                             C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lz4ref_blocks_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.c_void_p]
        L.lz4ref_xxh32.restype = C.c_uint32
        L.lz4ref_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.lz4ref_xxh32_init.argtypes = [C.POINTER(_XxhState), C.c_uint32]
        L.lz4ref_xxh32_update.argtypes = [C.POINTER(_XxhState), C.c_char_p, C.c_size_t]
        L.lz4ref_xxh32_digest.restype = C.c_uint32
        L.lz4ref_xxh32_digest.argtypes = [C.POINTER(_XxhState)]
        L.lz4ref_frame_options_default.argtypes = [C.POINTER(FrameOptions)]
        L.lz4ref_frame_bound.restype = C.c_size_t
        L.lz4ref_frame_bound.argtypes = [C.c_size_t, C.POINTER(FrameOptions)]
        L.lz4ref_frame_encode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FrameOptions),
                                          C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.lz4ref_frame_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int64, _RESOLVER, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_uint32)]
        sz, psz, u8 = C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p
        L.lz4ref_frame_info.argtypes = [u8, sz, C.POINTER(FrameInfo)]
        L.lz4ref_skippable_frame_encode.argtypes = [u8, sz, C.c_int, C.c_void_p, sz, psz]
        L.lz4ref_legacy_frame_bound.restype = sz
        L.lz4ref_legacy_frame_bound.argtypes = [sz]
        L.lz4ref_legacy_frame_encode.argtypes = [u8, sz, C.c_int64, C.c_void_p, sz, psz]
        L.lz4ref_compress_with_size.argtypes = [u8, sz, C.c_int, C.c_int64, C.c_void_p, sz, psz]
        L.lz4ref_decompress_with_size.argtypes = [u8, sz, C.c_void_p, sz, psz]
        L.lz4ref_stream_encoder_new.restype = C.c_void_p
        L.lz4ref_stream_encoder_new.argtypes = [C.POINTER(FrameOptions), u8, sz, C.POINTER(C.c_int)]
        L.lz4ref_stream_encoder_header.argtypes = [C.c_void_p, C.c_void_p, sz, psz]
        L.lz4ref_stream_encoder_add.argtypes = [C.c_void_p, u8, sz, C.c_void_p, sz, psz,
                                                C.POINTER(C.c_uint32), sz, psz]
        L.lz4ref_stream_encoder_close.argtypes = [C.c_void_p, C.c_void_p, sz, psz, C.POINTER(C.c_uint32), sz,
                                                  psz]
        L.lz4ref_stream_encoder_free.argtypes = [C.c_void_p]
        L.lz4ref_stream_encoder_free.restype = None
        L.lz4ref_stream_decoder_new.restype = C.c_void_p
        L.lz4ref_stream_decoder_new.argtypes = [C.c_int64, _RESOLVER, C.c_void_p, C.c_int]
        L.lz4ref_stream_decoder_add.argtypes = [C.c_void_p, u8, sz]
        L.lz4ref_stream_decoder_next.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), psz,
                                                 C.POINTER(C.c_uint32)]
        L.lz4ref_stream_decoder_finish.argtypes = [C.c_void_p]
        L.lz4ref_stream_decoder_free.argtypes = [C.c_void_p]
        L.lz4ref_stream_decoder_free.restype = None
        _lib = L
    return _lib


def _raise(status: int, dict_id: int | None = None):
    L = lib()
    msg = L.lz4ref_status_message(status).decode()
    if dict_id is not None and status == 50:
        msg += str(dict_id)
    raise Lz4RefError(status, L.lz4ref_status_type(status).decode(), msg)


def compress_bound(n: int) -> int:
    return lib().lz4ref_compress_bound(n)


def _compress(fn, src: bytes, dictionary, param: int) -> bytes:
    src = bytes(src)
    cap = compress_bound(len(src)) + 16
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    d = bytes(dictionary) if dictionary is not None else None
    st = fn(src, len(src), d, len(d) if d is not None else 0, param, dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def block_compress_fast(src: bytes, dictionary=None, acceleration: int = 1) -> bytes:
    """lz4BlockCompress (lib/src/block/lz4_block_encoder.dart:24-33)."""
    return _compress(lib().lz4ref_block_compress_fast, src, dictionary, acceleration)


def block_compress_hc(src: bytes, dictionary=None, max_search_depth: int = 64) -> bytes:
    """lz4HcBlockCompress (lib/src/hc/lz4_hc_block_encoder.dart:38-47)."""
    return _compress(lib().lz4ref_block_compress_hc, src, dictionary, max_search_depth)


def block_compress_ex(src: bytes, dictionary=None, level: int = 0, param: int | None = None,
                      conformant: bool = False) -> bytes:
    """Either encoder with the optional lz4-spec end-of-block rules (see lz4ref.h)."""
    src = bytes(src)
    if param is None:
        param = 1 if level == 0 else 64
    cap = compress_bound(len(src)) + 16
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    d = bytes(dictionary) if dictionary is not None else None
    st = lib().lz4ref_block_compress_ex(src, len(src), d, len(d) if d is not None else 0, level, param,
                                        int(conformant), dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def block_decompress(src: bytes, decompressed_size: int) -> bytes:
    """lz4BlockDecompress (lib/src/block/lz4_block_decoder.dart:7-77)."""
    src = bytes(src)
    dst = C.create_string_buffer(max(decompressed_size, 1))
    st = lib().lz4ref_block_decompress(src, len(src), dst, decompressed_size)
    if st:
        _raise(st)
    return dst.raw[:decompressed_size]


def block_decompress_into(src: bytes, prefix: bytes = b"", max_length: int | None = None) -> bytes:
    """lz4BlockDecompressInto (:79-119); returns the whole writer contents (prefix included)."""
    src = bytes(src)
    if max_length is None:
        max_length = len(prefix) + 255 * len(src) + 64
    buf = C.create_string_buffer(max(max_length, len(prefix), 1))
    C.memmove(buf, bytes(prefix), len(prefix))
    out = C.c_size_t(0)
    st = lib().lz4ref_block_decompress_into(src, len(src), buf, len(prefix), max_length, C.byref(out))
    if st:
        _raise(st)
    return buf.raw[: out.value]


def block_conformant(comp: bytes) -> bool:
    """True when a compressed block obeys the LZ4 end-of-block rules (checker only, not a reference function)."""
    comp = bytes(comp)
    return bool(lib().lz4ref_block_conformant(comp, len(comp)))


def xxh32(data: bytes, seed: int = 0) -> int:
    data = bytes(data)
    return lib().lz4ref_xxh32(data, len(data), seed & 0xFFFFFFFF)


class Xxh32:
    """Streaming Xxh32 (lib/src/xxhash/xxh32.dart:21-164)."""

    def __init__(self, seed: int = 0):
        self._s = _XxhState()
        lib().lz4ref_xxh32_init(C.byref(self._s), seed & 0xFFFFFFFF)

    def update(self, data: bytes):
        data = bytes(data)
        lib().lz4ref_xxh32_update(C.byref(self._s), data, len(data))

    def digest(self) -> int:
        return lib().lz4ref_xxh32_digest(C.byref(self._s))


def frame_options(block_size_id=7, block_independence=True, block_checksum=False,
                  content_checksum=False, content_size=None, dict_id=None, compression=0,
                  acceleration=1, hc_max_search_depth=64, spec_conformant=False) -> FrameOptions:
    o = FrameOptions()
    lib().lz4ref_frame_options_default(C.byref(o))
    o.block_size_id = block_size_id
    o.block_independence = int(block_independence)
    o.block_checksum = int(block_checksum)
    o.content_checksum = int(content_checksum)
    o.has_content_size = int(content_size is not None)
    o.content_size = content_size or 0
    o.has_dict_id = int(dict_id is not None)
    o.dict_id = dict_id or 0
    o.compression = compression
    o.acceleration = acceleration
    o.hc_max_search_depth = hc_max_search_depth
    o.spec_conformant = int(spec_conformant)
    return o


def frame_encode(src: bytes, options: FrameOptions | None = None, dictionary=None) -> bytes:
    """lz4FrameEncodeBytesWithOptions (lib/src/frame/lz4_frame_encoder.dart:105-274)."""
    src = bytes(src)
    o = options if options is not None else frame_options()
    cap = lib().lz4ref_frame_bound(len(src), C.byref(o))
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    d = bytes(dictionary) if dictionary is not None else None
    st = lib().lz4ref_frame_encode(src, len(src), C.byref(o), d, len(d) if d is not None else 0,
                                   dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def frame_decode(src: bytes, max_output_bytes: int | None = None, dictionary_resolver=None,
                 capacity: int | None = None) -> bytes:
    """lz4FrameDecodeBytes (lib/src/frame/lz4_frame_decoder.dart:15-50)."""
    src = bytes(src)
    if capacity is None:
        capacity = max(64, len(src) * 255 + 64)
        if max_output_bytes is not None:
            capacity = min(capacity, max_output_bytes + 64)
        capacity = min(capacity, 1 << 31)
    dst = C.create_string_buffer(capacity)
    out = C.c_size_t(0)
    err_id = C.c_uint32(0)
    keep = []

    def _cb(_user, dict_id, pp, plen):
        d = dictionary_resolver(dict_id)
        if d is None:
            return 0
        b = C.create_string_buffer(bytes(d), max(len(d), 1))
        keep.append(b)
        pp[0] = C.cast(b, C.c_void_p).value
        plen[0] = len(d)
        return 1

    cb = _RESOLVER(_cb) if dictionary_resolver is not None else C.cast(None, _RESOLVER)
    st = lib().lz4ref_frame_decode(src, len(src), -1 if max_output_bytes is None else max_output_bytes,
                                   cb, None, dst, capacity, C.byref(out), C.byref(err_id))
    if st:
        _raise(st, err_id.value)
    return dst.raw[: out.value]


# ------------------------------------------------------------ host-only formats

def frame_info(src: bytes) -> dict:
    """lz4FrameInfo (lib/src/frame/lz4_frame_info.dart:89-199) as a dict of the struct fields."""
    src = bytes(src)
    info = FrameInfo()
    st = lib().lz4ref_frame_info(src, len(src), C.byref(info))
    if st:
        _raise(st)
    return {name: getattr(info, name) for name, _ in FrameInfo._fields_}


def skippable_frame_encode(data: bytes, index: int = 0) -> bytes:
    """lz4SkippableFrameEncode (lib/src/frame/lz4_frame_encoder.dart:27-46)."""
    data = bytes(data)
    cap = 8 + len(data)
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    st = lib().lz4ref_skippable_frame_encode(data, len(data), index, dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def legacy_frame_encode(src: bytes, acceleration: int = 1) -> bytes:
    """lz4LegacyFrameEncode (lib/src/frame/lz4_frame_encoder.dart:55-89)."""
    src = bytes(src)
    cap = lib().lz4ref_legacy_frame_bound(len(src))
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    st = lib().lz4ref_legacy_frame_encode(src, len(src), acceleration, dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def compress_with_size(src: bytes, level: int = 0, param: int | None = None) -> bytes:
    """lz4CompressWithSize (lib/src/block/lz4_sized_block.dart:15-31); level 0 fast, 1 hc."""
    src = bytes(src)
    if param is None:
        param = 1 if level == 0 else 64
    cap = compress_bound(len(src)) + 20
    dst = C.create_string_buffer(cap)
    out = C.c_size_t(0)
    st = lib().lz4ref_compress_with_size(src, len(src), level, param, dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


def decompress_with_size(src: bytes) -> bytes:
    """lz4DecompressWithSize (lib/src/block/lz4_sized_block.dart:39-55)."""
    src = bytes(src)
    cap = int.from_bytes(src[:4], "little") if len(src) >= 4 else 0
    dst = C.create_string_buffer(max(cap, 1))
    out = C.c_size_t(0)
    st = lib().lz4ref_decompress_with_size(src, len(src), dst, cap, C.byref(out))
    if st:
        _raise(st)
    return dst.raw[: out.value]


# ------------------------------------------------------------------- streaming

def stream_encode(chunks, options: FrameOptions | None = None, dictionary=None):
    """lz4FrameEncoderTransformerWithOptions (lib/src/frame/lz4_frame_stream_encoder.dart:25-236).

    Returns (pieces, error): the list of yielded chunks and the Lz4RefError that ended the
    stream, or None."""
    L = lib()
    o = options if options is not None else frame_options()
    d = bytes(dictionary) if dictionary is not None else None
    status = C.c_int(0)
    e = L.lz4ref_stream_encoder_new(C.byref(o), d, len(d) if d is not None else 0, C.byref(status))
    if not e:
        _raise(status.value)
    pieces: list[bytes] = []
    bmax = {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}[o.block_size_id]
    try:
        hdr = C.create_string_buffer(32)
        out = C.c_size_t(0)
        st = L.lz4ref_stream_encoder_header(e, hdr, 32, C.byref(out))
        if st:
            _raise(st)
        pieces.append(hdr.raw[: out.value])
        pending = 0

        def collect(dst, n_out, lens, n_pieces):
            pos = 0
            for k in range(n_pieces):
                pieces.append(dst.raw[pos: pos + lens[k]])
                pos += lens[k]
            assert pos == n_out

        for chunk in chunks:
            chunk = bytes(chunk)
            nblk = (pending + len(chunk)) // bmax + 1
            cap = nblk * (bmax + 8) + compress_bound(bmax) + 64
            dst = C.create_string_buffer(cap)
            lens = (C.c_uint32 * (nblk + 4))()
            npieces = C.c_size_t(0)
            st = L.lz4ref_stream_encoder_add(e, chunk, len(chunk), dst, cap, C.byref(out), lens, nblk + 4,
                                             C.byref(npieces))
            collect(dst, out.value, lens, npieces.value)
            if st:
                return pieces, Lz4RefError(st, L.lz4ref_status_type(st).decode(),
                                           L.lz4ref_status_message(st).decode())
            pending = (pending + len(chunk)) % bmax
        cap = bmax + compress_bound(bmax) + 64
        dst = C.create_string_buffer(cap)
        lens = (C.c_uint32 * 8)()
        npieces = C.c_size_t(0)
        st = L.lz4ref_stream_encoder_close(e, dst, cap, C.byref(out), lens, 8, C.byref(npieces))
        collect(dst, out.value, lens, npieces.value)
        if st:
            return pieces, Lz4RefError(st, L.lz4ref_status_type(st).decode(),
                                       L.lz4ref_status_message(st).decode())
        return pieces, None
    finally:
        L.lz4ref_stream_encoder_free(e)


def stream_decode(chunks, max_output_bytes: int | None = None, dictionary_resolver=None):
    """lz4FrameDecoderTransformer (lib/src/frame/lz4_frame_stream_decoder.dart:15-45).

    Returns (pieces, error) like stream_encode."""
    L = lib()
    keep = []

    def _cb(_user, dict_id, pp, plen):
        dd = dictionary_resolver(dict_id)
        if dd is None:
            return 0
        b = C.create_string_buffer(bytes(dd), max(len(dd), 1))
        keep.append(b)
        pp[0] = C.cast(b, C.c_void_p).value
        plen[0] = len(dd)
        return 1

    cb = _RESOLVER(_cb) if dictionary_resolver is not None else C.cast(None, _RESOLVER)
    d = L.lz4ref_stream_decoder_new(-1 if max_output_bytes is None else max_output_bytes, cb, None,
                                    int(dictionary_resolver is not None))
    pieces: list[bytes] = []
    has, ptr, n, err_id = C.c_int(0), C.c_void_p(), C.c_size_t(0), C.c_uint32(0)

    def err(st):
        msg = L.lz4ref_status_message(st).decode()
        if st == 50:
            msg += str(err_id.value)
        return Lz4RefError(st, L.lz4ref_status_type(st).decode(), msg)

    def drain():
        while True:
            st = L.lz4ref_stream_decoder_next(d, C.byref(has), C.byref(ptr), C.byref(n), C.byref(err_id))
            if st:
                return st
            if not has.value:
                return 0
            pieces.append(C.string_at(ptr.value, n.value) if n.value else b"")

    try:
        for chunk in chunks:
            chunk = bytes(chunk)
            L.lz4ref_stream_decoder_add(d, chunk, len(chunk))
            st = drain()
            if st:
                return pieces, err(st)
        st = L.lz4ref_stream_decoder_finish(d)
        if st:
            return pieces, err(st)
        st = drain()
        if st:
            return pieces, err(st)
        return pieces, None
    finally:
        L.lz4ref_stream_decoder_free(d)
