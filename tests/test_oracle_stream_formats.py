"""CPU: the oracle's streaming and format-edge restatements against the reference's own tests.

Each case restates a test of the reference (file:line in the docstring) with the same inputs
and expectations, so the restatement is held to what the reference's suite asserts.  Byte
constants come from those tests; everything else they assert is a round trip or an exception
type/message.
"""
import random

import numpy as np
import pytest

from tests.helpers import u32le

MiB = 1 << 20


def chunked(data, sizes):
    """_chunk (test/frame/lz4_frame_stream_decoder_test.dart:7-17)."""
    out, off, i = [], 0, 0
    while off < len(data):
        n = sizes[i % len(sizes)]
        out.append(data[off:off + n])
        off += n
        i += 1
    return out


def random_chunks(data, rng, max_size):
    """_randomChunk (test/frame/lz4_frame_stream_chunk_boundaries_test.dart:7-16)."""
    out, off = [], 0
    while off < len(data):
        n = 1 + rng.randrange(max_size)
        out.append(data[off:off + n])
        off += n
    return out


def dependent_friendly(block_size, copies=3):
    """_makeDependentFriendlyPayload (chunk_boundaries_test.dart:26-46)."""
    block = bytes((i * 31) & 0xFF for i in range(block_size))
    parts = [block]
    for k in range(1, copies):
        parts.append(block[k:] + block[block_size - k:])
    return b"".join(parts)


# ------------------------------------------------------------------ lz4FrameInfo

def test_frame_info_standard(oracle):
    """test/frame/lz4_frame_info_test.dart:8-39."""
    enc = oracle.frame_encode(b"", oracle.frame_options())
    info = oracle.frame_info(enc)
    assert not info["is_skippable"] and not info["is_legacy"]
    assert info["magic"] == 0x184D2204
    assert info["block_independence"] == 1 and info["content_checksum"] == 0
    assert not info["has_content_size"] and not info["has_dict_id"]
    assert info["header_size"] == 7


def test_frame_info_content_size_and_checksum(oracle):
    """lz4_frame_info_test.dart:41-55."""
    enc = oracle.frame_encode(bytes(100), oracle.frame_options(content_size=100, content_checksum=True))
    info = oracle.frame_info(enc)
    assert info["has_content_size"] and info["content_size"] == 100
    assert info["content_checksum"] == 1 and info["header_size"] == 15


def test_frame_info_skippable_legacy_and_errors(oracle):
    """lz4_frame_info_test.dart:57-91."""
    info = oracle.frame_info(bytes([0x50, 0x2A, 0x4D, 0x18, 4, 0, 0, 0, 1, 2, 3, 4]))
    assert info["is_skippable"] and not info["is_legacy"]
    assert info["skippable_size"] == 4 and info["header_size"] == 8
    info = oracle.frame_info(bytes([0x02, 0x21, 0x4C, 0x18]))
    assert info["is_legacy"] and not info["is_skippable"] and info["header_size"] == 4
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.frame_info(bytes(3))
    assert e.value.type_name == "Lz4FormatException"
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.frame_info(bytes(4))
    assert e.value.type_name == "Lz4FormatException" and e.value.message == "Invalid LZ4 frame magic number"


# ------------------------------------------------------------- skippable frames

def test_skippable_encode_bytes(oracle):
    """test/frame/lz4_skippable_frame_test.dart:8-68."""
    assert oracle.skippable_frame_encode(b"") == bytes([0x50, 0x2A, 0x4D, 0x18, 0, 0, 0, 0])
    assert oracle.skippable_frame_encode(bytes([1, 2, 3, 4, 5])) == \
        bytes([0x50, 0x2A, 0x4D, 0x18, 5, 0, 0, 0, 1, 2, 3, 4, 5])
    assert oracle.skippable_frame_encode(bytes([0xAB, 0xCD]), 15) == \
        bytes([0x5F, 0x2A, 0x4D, 0x18, 2, 0, 0, 0, 0xAB, 0xCD])
    for bad in (-1, 16):
        with pytest.raises(oracle.Lz4RefError) as e:
            oracle.skippable_frame_encode(b"", bad)
        assert e.value.type_name == "RangeError"


def test_skippable_then_frame_and_info(oracle):
    """lz4_skippable_frame_test.dart:70-98."""
    payload = b"Hello, World!"
    combined = oracle.skippable_frame_encode(bytes([1, 2, 3]), 5) + oracle.frame_encode(payload)
    assert oracle.frame_decode(combined) == payload
    pieces, err = oracle.stream_decode(chunked(combined, [1, 2, 3]))
    assert err is None and b"".join(pieces) == payload
    info = oracle.frame_info(oracle.skippable_frame_encode(bytes([1, 2, 3, 4]), 7))
    assert info["is_skippable"] and info["skippable_size"] == 4 and info["magic"] == 0x184D2A57


# ---------------------------------------------------------------- legacy frames

def test_legacy_encode_cases(oracle):
    """test/frame/lz4_legacy_frame_encode_test.dart:8-69,89-104."""
    assert oracle.legacy_frame_encode(b"") == bytes([0x02, 0x21, 0x4C, 0x18])
    src = b"Hello, Legacy LZ4 World!"
    frame = oracle.legacy_frame_encode(src)
    assert frame[:4] == bytes([0x02, 0x21, 0x4C, 0x18])
    assert oracle.frame_decode(frame) == src
    src = (np.arange(MiB, dtype=np.uint32) % 256).astype(np.uint8).tobytes()
    assert oracle.frame_decode(oracle.legacy_frame_encode(src)) == src
    info = oracle.frame_info(oracle.legacy_frame_encode(bytes([1, 2, 3, 4, 5])))
    assert info["is_legacy"] and info["magic"] == 0x184C2102 and info["header_size"] == 4
    src = (np.arange(100000, dtype=np.uint32) % 256).astype(np.uint8).tobytes()
    for accel in (1, 10):
        assert oracle.frame_decode(oracle.legacy_frame_encode(src, accel)) == src


def test_legacy_encode_two_blocks(oracle):
    """lz4_legacy_frame_encode_test.dart:48-58: 10 MiB needs two 8 MiB blocks."""
    src = ((np.arange(10 * MiB, dtype=np.uint32) * 7) % 256).astype(np.uint8).tobytes()
    frame = oracle.legacy_frame_encode(src)
    first = int.from_bytes(frame[4:8], "little")
    second = int.from_bytes(frame[8 + first:12 + first], "little")
    assert 12 + first + second == len(frame)
    assert oracle.frame_decode(frame, capacity=len(src) + 64) == src
    pieces, err = oracle.stream_decode(chunked(frame, [MiB + 13]))
    assert err is None and [len(p) for p in pieces] == [8 * MiB, 2 * MiB] and b"".join(pieces) == src


def test_legacy_streaming_decode(oracle):
    """lz4_legacy_frame_encode_test.dart:71-87 and lz4_legacy_frame_decode_test.dart:75-88."""
    src = b"Streaming legacy test data!"
    frame = oracle.legacy_frame_encode(src)
    pieces, err = oracle.stream_decode(chunked(frame, [5]))
    assert err is None and b"".join(pieces) == src
    b0, b1 = b"\x41" * (8 * MiB), b"tail"
    c0, c1 = oracle.block_compress_fast(b0), oracle.block_compress_fast(b1)
    frame = u32le(0x184C2102) + u32le(len(c0)) + c0 + u32le(len(c1)) + c1
    assert oracle.frame_decode(frame, capacity=len(b0) + 64) == b0 + b1
    pieces, err = oracle.stream_decode(chunked(frame, [1, 2, 3, 5, 8, 13, 21]))
    assert err is None and b"".join(pieces) == b0 + b1


def test_legacy_not_full_and_followed_by_frame(oracle):
    """lz4_legacy_frame_decode_test.dart:90-114, sync and streaming."""
    ca, cb = oracle.block_compress_fast(b"\x42" * 1024), oracle.block_compress_fast(b"\x00")
    frame = u32le(0x184C2102) + u32le(len(ca)) + ca + u32le(len(cb)) + cb
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.frame_decode(frame)
    assert e.value.type_name == "Lz4CorruptDataException"
    pieces, err = oracle.stream_decode([frame])
    assert err is not None and err.message == "Legacy block is not full" and pieces == []
    cl = oracle.block_compress_fast(b"legacy")
    combined = u32le(0x184C2102) + u32le(len(cl)) + cl + oracle.frame_encode(b"current")
    assert oracle.frame_decode(combined) == b"legacycurrent"
    pieces, err = oracle.stream_decode(chunked(combined, [3, 1, 4]))
    assert err is None and b"".join(pieces) == b"legacycurrent"


# ----------------------------------------------------------------- sized blocks

def test_sized_block_cases(oracle):
    """test/block/lz4_sized_block_test.dart:8-56."""
    src = bytes(i % 256 for i in range(1000))
    c = oracle.compress_with_size(src)
    assert int.from_bytes(c[:4], "little") == len(src) and len(c) > 4
    assert oracle.decompress_with_size(c) == src
    c = oracle.compress_with_size(b"")
    assert c == bytes(4) and oracle.decompress_with_size(c) == b""
    src = bytes(i % 10 for i in range(1000))
    assert oracle.decompress_with_size(oracle.compress_with_size(src, level=1)) == src
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.decompress_with_size(bytes([1, 2, 3]))
    assert e.value.type_name == "Lz4FormatException" and e.value.message == "Input too short for size header"
    c = oracle.compress_with_size(bytes(range(100)))
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.decompress_with_size(c[:-1])
    assert e.value.type_name == "Lz4FormatException"


# --------------------------------------------------------------- stream decoder

def test_stream_decoder_reference_cases(oracle):
    """test/frame/lz4_frame_stream_decoder_test.dart:49-119."""
    src = b"Hello streaming frame decode"
    pieces, err = oracle.stream_decode(chunked(oracle.frame_encode(src), [1, 2, 3, 1, 5, 8]))
    assert err is None and b"".join(pieces) == src
    combined = oracle.frame_encode(b"A") + oracle.frame_encode(b"B")
    pieces, err = oracle.stream_decode(chunked(combined, [1, 1, 2, 1, 3]))
    assert err is None and b"".join(pieces) == b"AB"
    enc = oracle.frame_encode(b"Hello")
    pieces, err = oracle.stream_decode(chunked(enc, [1, 2, 1, 4]), max_output_bytes=4)
    assert err is not None and err.type_name == "Lz4OutputLimitException"
    pieces, err = oracle.stream_decode(chunked(enc[:-1], [2, 1, 1]))
    assert err is not None and err.type_name == "Lz4FormatException" and err.message == "Unexpected end of input"
    desc = bytes([0x61, 0x40]) + u32le(0x12345678)
    frame = u32le(0x184D2204) + desc + bytes([(oracle.xxh32(desc) >> 8) & 0xFF]) + u32le(0)
    pieces, err = oracle.stream_decode(chunked(frame, [1, 2, 1, 3, 1, 5]))
    assert err is not None and err.type_name == "Lz4UnsupportedFeatureException"
    assert err.message == "Dictionary ID present but no dictionary resolver provided"


# --------------------------------------------------------------- stream encoder

def test_stream_encoder_reference_cases(oracle):
    """test/frame/lz4_frame_stream_encoder_test.dart:28-98."""
    src = b"Hello streaming frame encode"
    pieces, err = oracle.stream_encode(chunked(src, [1, 2, 3, 1, 5, 8]))
    assert err is None and oracle.frame_decode(b"".join(pieces)) == src
    src = b"Round-trip streaming encode/decode"
    pieces, err = oracle.stream_encode(chunked(src, [1, 1, 2, 3, 5, 8]))
    out, err2 = oracle.stream_decode(pieces)
    assert err is None and err2 is None and b"".join(out) == src
    pieces, err = oracle.stream_encode([])
    assert err is None and oracle.frame_decode(b"".join(pieces)) == b""
    with pytest.raises(oracle.Lz4RefError) as e:
        oracle.stream_encode([], oracle.frame_options(acceleration=0))
    assert e.value.type_name == "RangeError"
    src = b"Hello streaming options"
    o = oracle.frame_options(block_size_id=4, block_checksum=True, content_checksum=True, content_size=len(src))
    pieces, err = oracle.stream_encode(chunked(src, [1, 2, 3, 1, 5, 8]), o)
    assert err is None and oracle.frame_decode(b"".join(pieces)) == src
    src = b"\x41" * (128 * 1024)
    o = oracle.frame_options(block_size_id=4, compression=1)
    pieces, err = oracle.stream_encode(chunked(src, [1, 7, 3, 13, 5, 8]), o)
    assert err is None and oracle.frame_decode(b"".join(pieces)) == src


def test_stream_encoder_dependent_blocks_and_content_size(oracle):
    """lz4_frame_stream_encoder_test.dart:100-172."""
    src = dependent_friendly(64 * 1024, copies=2)
    sizes = [1024, 4096, 2048, 8192, 123]
    indep, e1 = oracle.stream_encode(chunked(src, sizes), oracle.frame_options(block_size_id=4))
    dep, e2 = oracle.stream_encode(chunked(src, sizes),
                                   oracle.frame_options(block_size_id=4, block_independence=False))
    assert e1 is None and e2 is None
    assert oracle.frame_decode(b"".join(dep)) == src
    assert len(b"".join(dep)) < len(b"".join(indep))
    pieces, err = oracle.stream_encode(chunked(b"Hello", [1]), oracle.frame_options(content_size=4))
    assert err is not None and err.type_name == "Lz4FormatException"
    assert err.message == "contentSize does not match stream length"
    pieces, err = oracle.stream_encode(chunked(b"Hello", [2, 3]), oracle.frame_options(content_size=6))
    assert err is not None and err.message == "contentSize does not match stream length"
    assert len(pieces) == 2  # header and the block, yielded before the throw


def test_stream_chunk_boundaries(oracle):
    """test/frame/lz4_frame_stream_chunk_boundaries_test.dart:72-134 (8 of its 25 seeds)."""
    src = dependent_friendly(64 * 1024)
    for i in range(8):
        for indep, seed in ((True, 1000 + i), (False, 2000 + i)):
            o = oracle.frame_options(block_size_id=4, block_independence=indep)
            enc, err = oracle.stream_encode(random_chunks(src, random.Random(seed), 8192), o)
            assert err is None
            enc = b"".join(enc)
            assert enc == oracle.frame_encode(src, o)  # same bytes as the one-shot encoder
            out, err = oracle.stream_decode(random_chunks(enc, random.Random(seed + 2000), 1024))
            assert err is None and b"".join(out) == src
    a, b = b"A", b"\x42" * (128 * 1024)
    enc_a, _ = oracle.stream_encode(random_chunks(a, random.Random(5000), 8192))
    enc_b, _ = oracle.stream_encode(random_chunks(b, random.Random(6000), 8192),
                                    oracle.frame_options(block_size_id=4, block_independence=False))
    out, err = oracle.stream_decode(random_chunks(b"".join(enc_a) + b"".join(enc_b), random.Random(7000), 1024))
    assert err is None and b"".join(out) == a + b


def test_stream_decoder_equals_sync_decoder_on_golden_frames(oracle):
    """The seven lz4-CLI golden frames (test/vectors/lz4_frame_reference_vectors_test.dart:7-25)
    through the stream decoder, fed in small chunks."""
    from tests.helpers import golden, golden_frame
    for name in golden()["lz4_cli_frames"]:
        frame, _ = golden_frame(name)
        want = oracle.frame_decode(frame)
        out, err = oracle.stream_decode(chunked(frame, [7, 1, 64, 3]))
        assert err is None and b"".join(out) == want, name
