"""ctypes binding of liblz4mi.so (the C ABI declared in include/lz4mi.h).

There is no CPU implementation behind this package: if the HIP library is
missing or no MI355X is usable, calls raise ``Lz4BackendError``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from .exceptions import (Lz4BackendError, Lz4CorruptDataException, Lz4Exception, Lz4FormatException,
                         Lz4OutputLimitException, Lz4UnsupportedFeatureException, RangeError)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LZ4MI_LIB selects another build of the same ABI (e.g. the diagnostic liblz4mi_prof.so)
LIB_PATH = os.environ.get("LZ4MI_LIB") or os.path.join(_HERE, "liblz4mi.so")

OK = 0
DICT_NOT_FOUND = 50
RANGE = 60
BACKEND = 70
DST_TOO_SMALL = 71
NEED_DICT = 72
UNSUPPORTED_HERE = 73
STREAM_NEED_INPUT = 74
FORMAT_CONTENT_SIZE_STREAM = 8
FORMAT_SIZE_HEADER = 9


class Batch(C.Structure):
    _fields_ = [
        ("d_src", C.c_void_p), ("d_src_off", C.c_void_p), ("d_src_len", C.c_void_p),
        ("src_stride", C.c_uint64), ("src_len_uniform", C.c_uint32), ("src_total", C.c_uint64),
        ("d_dst", C.c_void_p), ("d_dst_off", C.c_void_p), ("dst_stride", C.c_uint64),
        ("d_dst_cap", C.c_void_p), ("dst_cap_uniform", C.c_uint32),
        ("d_dict", C.c_void_p), ("dict_len", C.c_uint32), ("nblocks", C.c_uint32),
        ("d_out_len", C.c_void_p), ("d_status", C.c_void_p),
    ]


class FrameOptions(C.Structure):
    _fields_ = [
        ("block_size_id", C.c_int32), ("block_independence", C.c_int32), ("block_checksum", C.c_int32),
        ("content_checksum", C.c_int32), ("has_content_size", C.c_int32), ("has_dict_id", C.c_int32),
        ("compression", C.c_int32), ("acceleration", C.c_int32), ("hc_max_search_depth", C.c_int32),
        ("dict_id", C.c_uint32), ("content_size", C.c_uint64),
        ("spec_conformant", C.c_int32), ("reserved", C.c_int32),
    ]


class XxhState(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32), ("v", C.c_uint32 * 4), ("mem_size", C.c_uint32), ("total_len", C.c_uint64),
        ("mem", C.c_uint8 * 16),
    ]


class FrameInfo(C.Structure):
    _fields_ = [
        ("magic", C.c_uint32), ("is_skippable", C.c_int32), ("is_legacy", C.c_int32),
        ("block_independence", C.c_int32), ("block_checksum", C.c_int32), ("content_checksum", C.c_int32),
        ("has_content_size", C.c_int32), ("has_dict_id", C.c_int32), ("skippable_size", C.c_uint32),
        ("dict_id", C.c_uint32), ("block_max_size", C.c_uint32), ("header_size", C.c_uint32),
        ("content_size", C.c_uint64),
    ]


# every symbol include/lz4mi.h declares
EXPORTS = [
    "lz4mi_status_type", "lz4mi_status_message", "lz4mi_ctx_create", "lz4mi_ctx_destroy",
    "lz4mi_ctx_synchronize", "lz4mi_ctx_device_bytes", "lz4mi_last_backend_error", "lz4mi_version", "lz4mi_compress_bound",
    "lz4mi_block_compress_fast", "lz4mi_block_compress_hc", "lz4mi_block_decompress",
    "lz4mi_block_decompress_into", "lz4mi_xxh32", "lz4mi_compress_fast_batch", "lz4mi_compress_hc_batch",
    "lz4mi_decompress_batch", "lz4mi_decompress_batch_static", "lz4mi_set_decode_form", "lz4mi_ctx_set_decode_form", "lz4mi_xxh32_batch", "lz4mi_xxh32_device", "lz4mi_frame_options_default",
    "lz4mi_frame_bound", "lz4mi_frame_encode", "lz4mi_frame_encode_device", "lz4mi_frame_shard_compress",
    "lz4mi_frame_shard_write", "lz4mi_frame_header", "lz4mi_frame_decode_next", "lz4mi_frame_decode_bound",
    "lz4mi_xxh32_state_init", "lz4mi_xxh32_state_update", "lz4mi_xxh32_state_digest",
    "lz4mi_frame_info_parse", "lz4mi_skippable_frame_encode", "lz4mi_legacy_frame_bound",
    "lz4mi_legacy_frame_encode", "lz4mi_frame_encoder_create", "lz4mi_frame_encoder_bound",
    "lz4mi_frame_encoder_update", "lz4mi_frame_encoder_finish", "lz4mi_frame_encoder_destroy",
    "lz4mi_frame_decoder_create", "lz4mi_frame_decoder_feed", "lz4mi_frame_decoder_next",
    "lz4mi_frame_decoder_set_dictionary", "lz4mi_frame_decoder_finish", "lz4mi_frame_decoder_destroy",
    "lz4mi_frame_block_table", "lz4mi_frame_shard_decode", "lz4mi_blocks_gather", "lz4mi_blocks_conformance",
]

_lib = None
_lock = threading.Lock()


def load():
    """Loads liblz4mi.so; raises Lz4BackendError when it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        try:  # torch bundles its own HIP runtime; it must be the one this process uses
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise Lz4BackendError(
                f"{LIB_PATH} is missing: build it with `make -C lz4_amd/csrc` "
                "(there is no CPU implementation to fall back to)")
        L = C.CDLL(LIB_PATH)
        vp, u8p, sz, i32, i64, u32, u64 = (C.c_void_p, C.c_char_p, C.c_size_t, C.c_int32, C.c_int64,
                                           C.c_uint32, C.c_uint64)
        psz = C.POINTER(C.c_size_t)
        L.lz4mi_status_type.restype = C.c_char_p
        L.lz4mi_status_type.argtypes = [i32]
        L.lz4mi_status_message.restype = C.c_char_p
        L.lz4mi_status_message.argtypes = [i32]
        L.lz4mi_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
        L.lz4mi_ctx_destroy.argtypes = [vp]
        L.lz4mi_ctx_destroy.restype = None
        L.lz4mi_ctx_synchronize.argtypes = [vp]
        if not os.environ.get("LZ4MI_LIB") or hasattr(L, "lz4mi_ctx_device_bytes"):  # an older build under LZ4MI_LIB (A/B timing) may lack it
            L.lz4mi_ctx_device_bytes.restype = u64
            L.lz4mi_ctx_device_bytes.argtypes = [vp]
        L.lz4mi_last_backend_error.restype = C.c_char_p
        L.lz4mi_last_backend_error.argtypes = [vp]
        L.lz4mi_compress_bound.restype = sz
        L.lz4mi_compress_bound.argtypes = [sz]
        L.lz4mi_block_compress_fast.argtypes = [vp, u8p, sz, u8p, sz, i64, vp, sz, psz]
        L.lz4mi_block_compress_hc.argtypes = [vp, u8p, sz, u8p, sz, i64, vp, sz, psz]
        L.lz4mi_block_decompress.argtypes = [vp, u8p, sz, vp, i64]
        L.lz4mi_block_decompress_into.argtypes = [vp, u8p, sz, vp, sz, sz, psz]
        L.lz4mi_xxh32.argtypes = [vp, u8p, sz, u32, C.POINTER(u32)]
        for name in ("lz4mi_compress_fast_batch", "lz4mi_compress_hc_batch"):
            getattr(L, name).argtypes = [vp, C.POINTER(Batch), i64]
        L.lz4mi_decompress_batch.argtypes = [vp, C.POINTER(Batch), i32]
        L.lz4mi_decompress_batch_static.argtypes = [vp, C.POINTER(Batch), i32]
        L.lz4mi_set_decode_form.argtypes = [i32]
        if not os.environ.get("LZ4MI_LIB") or hasattr(L, "lz4mi_ctx_set_decode_form"):
            L.lz4mi_ctx_set_decode_form.argtypes = [C.c_void_p, i32]
        L.lz4mi_xxh32_batch.argtypes = [vp, C.POINTER(Batch), vp]
        L.lz4mi_xxh32_device.argtypes = [vp, vp, u64, u32, vp]
        L.lz4mi_frame_options_default.argtypes = [C.POINTER(FrameOptions)]
        L.lz4mi_frame_options_default.restype = None
        L.lz4mi_frame_bound.restype = sz
        L.lz4mi_frame_bound.argtypes = [sz, C.POINTER(FrameOptions)]
        L.lz4mi_frame_encode.argtypes = [vp, u8p, sz, C.POINTER(FrameOptions), u8p, sz, vp, sz, psz]
        L.lz4mi_frame_encode_device.argtypes = [vp, vp, u64, C.POINTER(FrameOptions), vp, u32, vp, u64,
                                                C.POINTER(u64)]
        L.lz4mi_frame_shard_compress.argtypes = [vp, vp, u64, C.POINTER(FrameOptions), vp, u32, vp,
                                                 C.POINTER(u32)]
        L.lz4mi_frame_shard_write.argtypes = [vp, vp, u64, C.POINTER(FrameOptions), vp, vp, u64]
        L.lz4mi_frame_header.argtypes = [C.POINTER(FrameOptions), vp, sz, psz]
        L.lz4mi_frame_decode_next.argtypes = [vp, vp, sz, psz, u8p, sz, i32, i32, i64, vp, sz, psz,
                                              C.POINTER(u32)]
        L.lz4mi_frame_decode_bound.argtypes = [vp, sz, psz]
        L.lz4mi_frame_block_table.argtypes = [vp, sz, C.POINTER(FrameInfo), vp, sz, psz, psz]
        L.lz4mi_frame_shard_decode.argtypes = [vp, vp, u64, vp, u32, C.POINTER(FrameOptions), vp, u32, vp, vp, vp]
        L.lz4mi_blocks_gather.argtypes = [vp, vp, u64, vp, vp, u32, vp, u64]
        L.lz4mi_blocks_conformance.argtypes = [vp, C.POINTER(Batch), vp]
        L.lz4mi_xxh32_state_init.argtypes = [C.POINTER(XxhState), u32]
        L.lz4mi_xxh32_state_init.restype = None
        L.lz4mi_xxh32_state_update.argtypes = [vp, C.POINTER(XxhState), u8p, sz]
        L.lz4mi_xxh32_state_digest.argtypes = [C.POINTER(XxhState)]
        L.lz4mi_xxh32_state_digest.restype = u32
        L.lz4mi_frame_info_parse.argtypes = [u8p, sz, C.POINTER(FrameInfo)]
        L.lz4mi_skippable_frame_encode.argtypes = [u8p, sz, i32, vp, sz, psz]
        L.lz4mi_legacy_frame_bound.restype = sz
        L.lz4mi_legacy_frame_bound.argtypes = [sz]
        L.lz4mi_legacy_frame_encode.argtypes = [vp, u8p, sz, i64, vp, sz, psz]
        L.lz4mi_frame_encoder_create.argtypes = [vp, C.POINTER(FrameOptions), u8p, sz, C.POINTER(vp)]
        L.lz4mi_frame_encoder_bound.restype = sz
        L.lz4mi_frame_encoder_bound.argtypes = [vp, sz]
        L.lz4mi_frame_encoder_update.argtypes = [vp, u8p, sz, vp, sz, psz]
        L.lz4mi_frame_encoder_finish.argtypes = [vp, vp, sz, psz]
        L.lz4mi_frame_encoder_destroy.argtypes = [vp]
        L.lz4mi_frame_encoder_destroy.restype = None
        L.lz4mi_frame_decoder_create.argtypes = [vp, i64, C.POINTER(vp)]
        L.lz4mi_frame_decoder_feed.argtypes = [vp, u8p, sz]
        L.lz4mi_frame_decoder_next.argtypes = [vp, vp, sz, psz, C.POINTER(u32)]
        L.lz4mi_frame_decoder_set_dictionary.argtypes = [vp, u8p, sz, i32]
        L.lz4mi_frame_decoder_finish.argtypes = [vp]
        L.lz4mi_frame_decoder_destroy.argtypes = [vp]
        L.lz4mi_frame_decoder_destroy.restype = None
        _lib = L
        return _lib


_TYPES = {
    "Lz4FormatException": Lz4FormatException,
    "Lz4CorruptDataException": Lz4CorruptDataException,
    "Lz4OutputLimitException": Lz4OutputLimitException,
    "Lz4UnsupportedFeatureException": Lz4UnsupportedFeatureException,
    "Lz4Exception": Lz4Exception,
}


def raise_status(status: int, ctx=None, dict_id=None, arg: str = ""):
    """Rebuilds the exception the reference throws for this status."""
    L = load()
    if status == RANGE:
        raise RangeError(f"Invalid argument{(' (' + arg + ')') if arg else ''}")
    tname = L.lz4mi_status_type(status).decode()
    msg = L.lz4mi_status_message(status).decode()
    if status == DICT_NOT_FOUND and dict_id is not None:
        msg += str(dict_id)
    cls = _TYPES.get(tname)
    if cls is None:
        detail = L.lz4mi_last_backend_error(ctx).decode() if ctx else ""
        raise Lz4BackendError(f"{msg} [{status}] {detail}".strip())
    raise cls(msg)


class Context:
    """One lz4mi context (HIP stream + device scratch) bound to a device."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = load()
        h = C.c_void_p()
        st = L.lz4mi_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != OK:
            why = L.lz4mi_last_backend_error(None).decode()
            raise Lz4BackendError(
                f"lz4mi_ctx_create(device={device}) failed with status {st}: no usable MI355X / HIP runtime ({why})")
        self.handle = h
        self.device = device
        self._L = L

    def synchronize(self):
        st = self._L.lz4mi_ctx_synchronize(self.handle)
        if st:
            raise_status(st, self.handle)

    def close(self):
        if self.handle:
            self._L.lz4mi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context(device: int = 0) -> Context:
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    if device not in ctxs:
        ctxs[device] = Context(device)
    return ctxs[device]
