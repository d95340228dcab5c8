"""GPU: streaming transformers, streaming Xxh32 and the format helpers against the oracle.

Every product call goes through the C ABI (lz4_amd.stream / lz4_amd.formats over
liblz4mi.so); the oracle restates the reference's stream encoder / decoder state machines
and is the checker only.  Bit-exact: yielded pieces, concatenated bytes, and the exception
(class + message) together with everything delivered in front of it.
"""
import random

import numpy as np
import pytest

import lz4_amd
from lz4_amd import (Lz4CompressionLevel, Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions,
                     Lz4HcOptions)
from tests.helpers import corpus_block, golden, golden_frame, u32le
from tests.test_oracle_stream_formats import chunked, dependent_friendly, random_chunks

pytestmark = pytest.mark.gpu

MiB = 1 << 20


def mixed(n, seed):
    """Text-like, LZ-synthetic and random 64 KiB corpus blocks, cut to n bytes."""
    parts, k = [], 0
    while sum(map(len, parts)) < n:
        parts.append(corpus_block((3, 5, 0, 1, 6, 7)[k % 6] + 8 * (seed + k), 65536).tobytes())
        k += 1
    return b"".join(parts)[:n]


def ref_options(oracle, o: Lz4FrameOptions):
    return oracle.frame_options(
        block_size_id=o.blockSize.bdId, block_independence=o.blockIndependence, block_checksum=o.blockChecksum,
        content_checksum=o.contentChecksum, content_size=o.contentSize, dict_id=o.dictId,
        compression=o.compression.value, acceleration=o.acceleration,
        hc_max_search_depth=(o.hcOptions or Lz4HcOptions()).maxSearchDepth)


def run_stream(gen):
    """Collects a transformer's output like Stream.toList(): (pieces, exception or None)."""
    pieces = []
    try:
        for p in gen:
            pieces.append(p)
    except (lz4_amd.Lz4Exception, lz4_amd.RangeError) as e:
        return pieces, e
    return pieces, None


def same_error(got, want):
    if want is None:
        assert got is None, got
        return
    assert got is not None, f"expected {want}"
    assert type(got).__name__ == want.type_name and got.message == want.message, (got, want)


# ------------------------------------------------------------------ Xxh32 state

def test_xxh32_streaming_equals_oracle(oracle):
    rng = random.Random(7)
    data = mixed(3 * MiB + 17, 1)
    for seed in (0, 0x9E3779B1):
        h, r = lz4_amd.Xxh32(seed=seed), oracle.Xxh32(seed)
        pos = 0
        while pos < len(data):
            n = rng.choice([0, 1, 3, 15, 16, 17, 63, 4096, 4111, 100000, 1 << 20])
            h.update(data[pos:pos + n])
            r.update(data[pos:pos + n])
            pos += n
            assert h.digest() == r.digest()
        assert h.digest() == oracle.xxh32(data, seed) == lz4_amd.xxh32(data, seed)
    h = lz4_amd.Xxh32()
    assert h.digest() == oracle.xxh32(b"") == 0x02CC5D05  # test/vectors/xxh32_test.dart:30
    h.update(b"abcdefgh", start=2, end=5)
    assert h.digest() == oracle.xxh32(b"cde")
    with pytest.raises(lz4_amd.RangeError):
        h.update(b"abc", start=4)


# --------------------------------------------------------------------- formats

def test_legacy_encode_bytes_equal_oracle(oracle):
    cases = [b"", b"Hello, Legacy LZ4 World!", mixed(MiB, 3),
             ((np.arange(10 * MiB, dtype=np.uint32) * 7) % 256).astype(np.uint8).tobytes()]
    for src in cases:
        got = lz4_amd.lz4LegacyEncode(src)
        assert got == oracle.legacy_frame_encode(src), len(src)
        assert lz4_amd.lz4FrameDecode(got) == src
    src = mixed(200000, 4)
    assert lz4_amd.lz4LegacyEncode(src, acceleration=10) == oracle.legacy_frame_encode(src, 10)
    assert lz4_amd.lz4LegacyEncode(b"", acceleration=0) == bytes([0x02, 0x21, 0x4C, 0x18])  # encoder never reached
    with pytest.raises(lz4_amd.RangeError):
        lz4_amd.lz4LegacyEncode(b"x", acceleration=0)
    info = lz4_amd.lz4FrameInfo(lz4_amd.lz4LegacyEncode(bytes([1, 2, 3, 4, 5])))
    assert info.isLegacy and info.magic == 0x184C2102 and info.headerSize == 4


def test_sized_block_equals_oracle(oracle):
    for src in (bytes(i % 256 for i in range(1000)), b"", mixed(70000, 5)):
        c = lz4_amd.lz4CompressWithSize(src)
        assert c == oracle.compress_with_size(src)
        assert lz4_amd.lz4DecompressWithSize(c) == src
    src = bytes(i % 10 for i in range(1000))
    c = lz4_amd.lz4CompressWithSize(src, level=Lz4CompressionLevel.hc)
    assert c == oracle.compress_with_size(src, level=1) and lz4_amd.lz4DecompressWithSize(c) == src
    with pytest.raises(lz4_amd.Lz4FormatException) as e:
        lz4_amd.lz4DecompressWithSize(bytes([1, 2, 3]))
    assert e.value.message == "Input too short for size header"
    with pytest.raises(lz4_amd.Lz4FormatException):
        lz4_amd.lz4DecompressWithSize(lz4_amd.lz4CompressWithSize(bytes(range(100)))[:-1])


# -------------------------------------------------------------- stream encoder

ENC_OPTIONS = [
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB),
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False),
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True, contentChecksum=True),
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k256KB, blockIndependence=False, blockChecksum=True,
                    contentChecksum=True, acceleration=3),
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, compression=Lz4FrameCompression.hc,
                    hcOptions=Lz4HcOptions(maxSearchDepth=16)),
    Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False,
                    compression=Lz4FrameCompression.hc, contentChecksum=True),
]


@pytest.mark.parametrize("k", range(len(ENC_OPTIONS)))
def test_stream_encoder_pieces_equal_oracle(oracle, k):
    o = ENC_OPTIONS[k]
    src = mixed(700000 + 1234 * k, 10 + k) + dependent_friendly(64 * 1024, 2)
    for seed, max_size in ((k, 8192), (100 + k, 300000)):
        chunks = random_chunks(src, random.Random(seed), max_size)
        got, err = run_stream(lz4_amd.lz4FrameEncoderWithOptions(options=o).bind(chunks))
        want, werr = oracle.stream_encode(chunks, ref_options(oracle, o))
        assert err is None and werr is None
        assert got == want
        assert b"".join(got) == lz4_amd.lz4FrameEncodeWithOptions(src, options=o)


@pytest.mark.parametrize("indep", [True, False])
def test_stream_encoder_with_dictionary(oracle, indep):
    dictionary = mixed(100000, 30)  # longer than the 64 KiB window
    src = dictionary[-40000:] + mixed(150000, 31)
    o = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=indep, dictId=0x12345678)
    chunks = random_chunks(src, random.Random(5), 50000)
    got, err = run_stream(lz4_amd.lz4FrameEncoderWithOptions(options=o, dictionary=dictionary).bind(chunks))
    want, werr = oracle.stream_encode(chunks, ref_options(oracle, o), dictionary)
    assert err is None and werr is None and got == want
    out, err = run_stream(lz4_amd.lz4FrameDecoder(dictionaryResolver=lambda i: dictionary).bind(got))
    assert err is None and b"".join(out) == src


def test_stream_encoder_reference_cases(oracle):
    """test/frame/lz4_frame_stream_encoder_test.dart:28-172 on the device path."""
    src = b"Hello streaming frame encode"
    got, err = run_stream(lz4_amd.lz4FrameEncoder().bind(chunked(src, [1, 2, 3, 1, 5, 8])))
    assert err is None and lz4_amd.lz4FrameDecode(b"".join(got)) == src
    assert got == oracle.stream_encode(chunked(src, [1, 2, 3, 1, 5, 8]))[0]
    got, err = run_stream(lz4_amd.lz4FrameEncoder().bind([]))
    assert err is None and got == oracle.stream_encode([])[0]
    assert lz4_amd.lz4FrameDecode(b"".join(got)) == b""
    with pytest.raises(lz4_amd.RangeError):
        lz4_amd.lz4FrameEncoder(acceleration=0)
    src = b"\x41" * (128 * 1024)
    o = Lz4FrameOptions(compression=Lz4FrameCompression.hc, blockSize=Lz4FrameBlockSize.k64KB)
    got, err = run_stream(lz4_amd.lz4FrameEncoderWithOptions(options=o).bind(chunked(src, [1, 7, 3, 13, 5, 8] * 40)))
    assert err is None and lz4_amd.lz4FrameDecode(b"".join(got)) == src
    for size, sizes in ((4, [1]), (6, [2, 3])):  # longer / shorter than contentSize
        o = Lz4FrameOptions(contentSize=size)
        got, err = run_stream(lz4_amd.lz4FrameEncoderWithOptions(options=o).bind(chunked(b"Hello", sizes)))
        want, werr = oracle.stream_encode(chunked(b"Hello", sizes), ref_options(oracle, o))
        same_error(err, werr)
        assert isinstance(err, lz4_amd.Lz4FormatException) and got == want


# -------------------------------------------------------------- stream decoder

def check_decode(oracle, frame, sizes=None, rng=None, max_out=None, resolver=None, cap=32 << 20):
    chunks = chunked(frame, sizes) if sizes else random_chunks(frame, rng, 100000)
    got, err = run_stream(lz4_amd.lz4FrameDecoder(maxOutputBytes=max_out, dictionaryResolver=resolver,
                                                  outputCapacity=cap).bind(chunks))
    want, werr = oracle.stream_decode(chunks, max_out, resolver)
    same_error(err, werr)
    assert b"".join(got) == b"".join(want)
    return b"".join(got), err


def test_stream_decoder_golden_frames(oracle):
    for name in golden()["lz4_cli_frames"]:
        frame, _ = golden_frame(name)
        out, err = check_decode(oracle, frame, sizes=[7, 1, 64, 3])
        assert err is None and out == oracle.frame_decode(frame), name


@pytest.mark.parametrize("k", range(len(ENC_OPTIONS)))
def test_stream_decoder_roundtrip_matrix(oracle, k):
    o = ENC_OPTIONS[k]
    src = mixed(900000 + 777 * k, 40 + k)
    frame = oracle.frame_encode(src, ref_options(oracle, o))
    for seed in (k, 50 + k):
        out, err = check_decode(oracle, frame, rng=random.Random(seed))
        assert err is None and out == src
    out, err = check_decode(oracle, frame, sizes=[len(frame)], cap=3 * o.blockSize.maxBytes)  # several launches
    assert err is None and out == src
    out, err = check_decode(oracle, frame, sizes=[len(frame)], cap=1)  # grows to one block
    assert err is None and out == src


def test_stream_decoder_reference_cases(oracle):
    """test/frame/lz4_frame_stream_decoder_test.dart:49-119, lz4_frame_stream_chunk_boundaries_test.dart:109-134."""
    enc = oracle.frame_encode(b"Hello streaming frame decode")
    check_decode(oracle, enc, sizes=[1, 2, 3, 1, 5, 8])
    out, err = check_decode(oracle, oracle.frame_encode(b"A") + oracle.frame_encode(b"B"), sizes=[1, 1, 2, 1, 3])
    assert err is None and out == b"AB"
    enc = oracle.frame_encode(b"Hello")
    _, err = check_decode(oracle, enc, sizes=[1, 2, 1, 4], max_out=4)
    assert isinstance(err, lz4_amd.Lz4OutputLimitException)
    _, err = check_decode(oracle, enc[:-1], sizes=[2, 1, 1])
    assert isinstance(err, lz4_amd.Lz4FormatException)
    desc = bytes([0x61, 0x40]) + u32le(0x12345678)
    frame = u32le(0x184D2204) + desc + bytes([(oracle.xxh32(desc) >> 8) & 0xFF]) + u32le(0)
    _, err = check_decode(oracle, frame, sizes=[1, 2, 1, 3, 1, 5])
    assert isinstance(err, lz4_amd.Lz4UnsupportedFeatureException)
    _, err = check_decode(oracle, frame, sizes=[5], resolver=lambda i: None)
    assert type(err) is lz4_amd.Lz4Exception and err.message == "Dictionary not found for ID: 305419896"
    a, b = b"A", b"\x42" * (128 * 1024)
    combined = oracle.frame_encode(a) + oracle.frame_encode(
        b, oracle.frame_options(block_size_id=4, block_independence=False))
    out, err = check_decode(oracle, combined, rng=random.Random(7000))
    assert err is None and out == a + b


def test_stream_decoder_dictionary_frames(oracle):
    dictionary = mixed(90000, 60)
    src = dictionary[-30000:] + mixed(200000, 61)
    for indep in (True, False):
        o = oracle.frame_options(block_size_id=4, block_independence=indep, dict_id=77, content_checksum=True)
        frame = oracle.frame_encode(src, o, dictionary)
        out, err = check_decode(oracle, frame, rng=random.Random(3), resolver=lambda i: dictionary if i == 77 else None)
        assert err is None and out == src
        # the dictionary does not carry over to the next frame (_resetFrameState)
        plain = oracle.frame_encode(b"after", oracle.frame_options(block_size_id=4))
        out, err = check_decode(oracle, frame + plain, rng=random.Random(4), resolver=lambda i: dictionary)
        assert err is None and out == src + b"after"


def test_stream_decoder_failures_deliver_the_output_in_front(oracle):
    src = mixed(600000, 70)
    o = oracle.frame_options(block_size_id=4, block_checksum=True, content_checksum=True, content_size=len(src))
    frame = bytearray(oracle.frame_encode(src, o))
    # walk to the 5th block record
    pos = 15
    offs = []
    while True:
        w = int.from_bytes(frame[pos:pos + 4], "little")
        if w == 0:
            break
        offs.append(pos)
        pos += 4 + (w & 0x7FFFFFFF) + 4
    end_mark = pos
    cases = []
    bad = bytearray(frame); bad[offs[4] + 10] ^= 0x55; cases.append(bytes(bad))             # block checksum
    bad = bytearray(frame); bad[offs[6]:offs[6] + 4] = u32le(70000); cases.append(bytes(bad))  # size > max
    bad = bytearray(frame); bad[end_mark + 4] ^= 1; cases.append(bytes(bad))                # content checksum
    bad = bytearray(frame); bad[6] ^= 1; cases.append(bytes(bad))                           # content size / HC
    cases.append(bytes(frame[:offs[7] + 100]))                                              # truncated
    cases.append(bytes(frame) + b"\x01\x02\x03")                                            # trailing garbage
    cases.append(bytes(frame) + b"\x01\x02\x03\x04")                                        # bad next magic
    for i, f in enumerate(cases):
        for sizes in ([len(f)], [4097], [100000, 3]):
            out, err = check_decode(oracle, f, sizes=sizes)
            assert err is not None, i
    # corrupt token stream without block checksums: the decoder kernel's own status
    o2 = oracle.frame_options(block_size_id=4)
    f2 = bytearray(oracle.frame_encode(src, o2))
    w = int.from_bytes(f2[7:11], "little") & 0x7FFFFFFF
    second = 7 + 4 + w
    f2[second + 4 + 2000:second + 4 + 2004] = b"\xff\xff\xff\xff"
    for sizes in ([len(f2)], [30000]):
        check_decode(oracle, bytes(f2), sizes=sizes)
    # output limit in the middle of a frame
    out, err = check_decode(oracle, bytes(frame), sizes=[50000], max_out=200000)
    assert isinstance(err, lz4_amd.Lz4OutputLimitException) and len(out) == 196608


def test_stream_decoder_legacy_and_skippable(oracle):
    src = mixed(300000, 80)
    leg = oracle.legacy_frame_encode(src)
    sk = oracle.skippable_frame_encode(b"metadata" * 100, 9)
    std = oracle.frame_encode(b"tail frame", oracle.frame_options(block_size_id=4))
    for f in (leg, sk + leg, leg + std, sk + std + sk + leg + leg, leg + b"\x00\x00"):
        for sizes in ([len(f)], [5], [70001, 13]):
            check_decode(oracle, f, sizes=sizes)
    ca, cb = oracle.block_compress_fast(b"\x42" * 1024), oracle.block_compress_fast(b"\x00")
    not_full = u32le(0x184C2102) + u32le(len(ca)) + ca + u32le(len(cb)) + cb
    _, err = check_decode(oracle, not_full, sizes=[len(not_full)])
    assert isinstance(err, lz4_amd.Lz4CorruptDataException) and err.message == "Legacy block is not full"
    zero = u32le(0x184C2102) + u32le(0) + b"abcd"
    _, err = check_decode(oracle, zero, sizes=[3])
    assert err.message == "Invalid legacy block size"
    check_decode(oracle, u32le(0x184C2102), sizes=[4])  # bare magic: finish() reports EOF
    _, err = check_decode(oracle, leg, sizes=[1000], max_out=1000)
    assert isinstance(err, lz4_amd.Lz4OutputLimitException)


def test_stream_decoder_two_full_legacy_blocks(oracle):
    src = b"\x41" * (8 * MiB) + b"tail"
    c0, c1 = oracle.block_compress_fast(src[:8 * MiB]), oracle.block_compress_fast(b"tail")
    frame = u32le(0x184C2102) + u32le(len(c0)) + c0 + u32le(len(c1)) + c1
    out, err = check_decode(oracle, frame, sizes=[1, 2, 3, 5, 8, 13, 21] * 50 + [4096])
    assert err is None and out == src
    assert lz4_amd.lz4LegacyEncode(src) == oracle.legacy_frame_encode(src)


# ------------------------------------------------------ spec-conformant blocks

def _edge_blocks():
    """Inputs whose reference encoding ends inside a match or searches into the last bytes."""
    rng = np.random.default_rng(5)
    out = [b"", b"a", b"abcd" * 3, b"abcd" * 3 + b"a", b"\x41" * 13, b"\x41" * 65536,
           bytes(i & 0x1F for i in range(70000)), mixed(65536, 90), mixed(65536 + 11, 91)]
    tail = rng.integers(0, 256, size=40, dtype=np.uint8).tobytes()
    for cut in range(0, 16):  # a repeated tail that the unrestricted encoder would match to the end
        body = mixed(3000, 92) + tail + mixed(500, 93) + tail
        out.append(body[:len(body) - cut])
    return out


def test_spec_conformant_frames_equal_oracle_and_decode_with_lz4_cli(oracle, tmp_path):
    import os
    import shutil
    import subprocess
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    have_cli = os.path.exists(cli)
    dictionary = mixed(70000, 94)
    for compression in (Lz4FrameCompression.fast, Lz4FrameCompression.hc):
        for linked in (False, True):
            o = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=not linked,
                                compression=compression, contentChecksum=True, specConformantBlocks=True)
            ro = ref_options(oracle, o)
            ro.spec_conformant = 1
            for k, src in enumerate(_edge_blocks() + [mixed(400000, 95)]):
                got = lz4_amd.lz4FrameEncodeWithOptions(src, options=o)
                assert got == oracle.frame_encode(src, ro), (compression, linked, k)
                assert lz4_amd.lz4FrameDecode(got) == src
                if have_cli and (k % 4 == 0 or len(src) > 100000):
                    p = tmp_path / "f.lz4"
                    p.write_bytes(got)
                    out = subprocess.run([cli, "-d", "-c", str(p)], capture_output=True, check=True).stdout
                    assert out == src
            # with a dictionary (not checked with the CLI: it would need the dictionary file too)
            src = dictionary[-5000:] + mixed(100000, 96)
            od = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=not linked,
                                 compression=compression, dictId=9, specConformantBlocks=True)
            rd = ref_options(oracle, od)
            rd.spec_conformant = 1
            got = lz4_amd.lz4FrameEncodeWithOptions(src, options=od, dictionary=dictionary)
            assert got == oracle.frame_encode(src, rd, dictionary)
            assert lz4_amd.lz4FrameDecode(got, dictionaryResolver=lambda i: dictionary) == src
    # the stream encoder honours the option too
    o = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, specConformantBlocks=True)
    src = mixed(300000, 97)
    pieces, err = run_stream(lz4_amd.lz4FrameEncoderWithOptions(options=o).bind(chunked(src, [70001])))
    assert err is None and b"".join(pieces) == lz4_amd.lz4FrameEncodeWithOptions(src, options=o)


def test_default_mode_unchanged_by_the_conformant_option(oracle):
    """specConformantBlocks=False stays byte-identical to the reference-faithful oracle."""
    for src in _edge_blocks():
        o = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB)
        assert lz4_amd.lz4FrameEncodeWithOptions(src, options=o) == oracle.frame_encode(src, ref_options(oracle, o))
