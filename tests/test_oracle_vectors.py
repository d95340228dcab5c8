"""Pins the CPU oracle to the reference's own offline vectors (no GPU).

Every expected value in this file comes from a file of the reference checkout
(cited per test) or from an independent real implementation present in the
build container (python-xxhash, the lz4 CLI); nothing here was produced by the
oracle itself.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.helpers import GOLDEN_DIR, build_frame, golden, golden_frame, ramp, u32le


def _err(oracle, fn, *a, **k):
    with pytest.raises(oracle.Lz4RefError) as e:
        fn(*a, **k)
    return e.value


# ----------------------------------------------------------------- xxh32

def _xxh_data(g):
    # generator: test/vectors/xxh32_test.dart:22-27
    gen = g["gen_seed"]
    out = bytearray()
    for _ in range(g["data_len"]):
        out.append((gen >> 24) & 0xFF)
        gen = (gen * gen) & 0xFFFFFFFF
    return bytes(out)


def test_xxh32_reference_vectors(oracle):
    g = golden()["xxh32"]
    data = _xxh_data(g)
    for length, seed, expected in g["cases"]:
        assert oracle.xxh32(data[:length], seed) == expected


def test_xxh32_streaming_equals_oneshot(oracle):
    # test/vectors/xxh32_streaming_test.dart:19-59 — several chunkings
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=5000, dtype=np.uint8).tobytes()
    for chunks in ([1] * 40, [3, 5, 7, 11, 13, 17], [16], [15, 1, 16, 17], [4096], [1, 2, 3, 5, 8, 13, 21, 34, 55, 89]):
        for seed in (0, 0x9E3779B1):
            h = oracle.Xxh32(seed)
            pos, k = 0, 0
            while pos < len(data):
                n = chunks[k % len(chunks)]
                h.update(data[pos:pos + n])
                pos += n
                k += 1
            assert h.digest() == oracle.xxh32(data, seed)
    assert oracle.Xxh32(0).digest() == 0x02CC5D05


def test_xxh32_against_python_xxhash(oracle):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(11)
    for n in list(range(0, 70)) + [255, 256, 257, 4095, 65536, 100001]:
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 0x9E3779B1, 0xFFFFFFFF):
            assert oracle.xxh32(d, seed) == xxhash.xxh32(d, seed=seed).intdigest()


# --------------------------------------------------------- block decoder

def test_block_decoder_known_answers(oracle):
    for v in golden()["block_decoder_ok"]:
        assert oracle.block_decompress(bytes(v["src"]), v["size"]) == bytes(v["out"]), v["name"]


def test_block_decoder_malformed(oracle):
    for v in golden()["block_decoder_err"]:
        e = _err(oracle, oracle.block_decompress, bytes(v["src"]), v["size"])
        assert e.type_name == v["type"], v["name"]


def test_block_decoder_error_messages(oracle):
    # messages: lib/src/block/lz4_block_decoder.dart:23,34,44,51,62,73; byte_writer.dart:98
    assert _err(oracle, oracle.block_decompress, b"", 4).message == "Unexpected end of input"
    assert _err(oracle, oracle.block_decompress, bytes([0x50, 1, 2]), 5).message == "Unexpected end of input"
    assert _err(oracle, oracle.block_decompress, bytes([0x50, 1, 2, 3, 4, 5]), 4).message == "Literal length exceeds output size"
    assert _err(oracle, oracle.block_decompress, bytes([0x40, 1, 2, 3, 4, 4, 0]), 5).message == "Match length exceeds output size"
    assert _err(oracle, oracle.block_decompress, bytes([0x10, 1, 2]), 1).message == "Trailing bytes after end of block"
    assert _err(oracle, oracle.block_decompress, bytes([0x10, 1, 5, 0]), 8).message == "Invalid match distance"
    assert _err(oracle, oracle.block_decompress, bytes([0x00]), 0).message == "Trailing bytes after end of block"
    assert _err(oracle, oracle.block_decompress, b"", -1).type_name == "RangeError"
    assert oracle.block_decompress(b"", 0) == b""


def test_block_decoder_lenient_semantics(oracle):
    # lib/src/block/lz4_block_decoder.dart:79-119
    assert oracle.block_decompress_into(b"") == b""
    assert oracle.block_decompress_into(bytes([0x00])) == b""  # lone token accepted
    assert oracle.block_decompress_into(bytes([0x10, 0x41])) == b"A"
    # truncation of literals is reported before the writer cap
    e = _err(oracle, oracle.block_decompress_into, bytes([0x50, 1, 2]), b"", 1)
    assert e.message == "Unexpected end of input"
    e = _err(oracle, oracle.block_decompress_into, bytes([0x50, 1, 2, 3, 4, 5]), b"", 4)
    assert e.type_name == "Lz4OutputLimitException"
    # distance check precedes capacity check (byte_writer.dart:97-104)
    e = _err(oracle, oracle.block_decompress_into, bytes([0x1F, 0x41, 9, 0, 200]), b"", 4)
    assert e.message == "Invalid match distance"
    e = _err(oracle, oracle.block_decompress_into, bytes([0x1F, 0x41, 1, 0, 200]), b"", 4)
    assert e.type_name == "Lz4OutputLimitException"
    # matches may reach into the pre-filled prefix
    assert oracle.block_decompress_into(bytes([0x00, 4, 0]), b"abcd") == b"abcdabcd"
    e = _err(oracle, oracle.block_decompress_into, bytes([0x00, 4]), b"abcd")
    assert e.message == "Unexpected end of input"


# ------------------------------------------------------ golden lz4 frames

@pytest.mark.parametrize("name", ["independent_70k", "dependent_70k", "small_default", "small_no_frame_crc",
                                  "small_block_checksum", "small_content_size", "empty"])
def test_lz4_cli_golden_frames_decode(oracle, name):
    frame, meta = golden_frame(name)
    assert frame[4] == meta["flg"] and frame[5] == meta["bd"]
    assert oracle.frame_decode(frame) == ramp(meta["payload_len"])


def test_skippable_prefix_ignored(oracle):
    # test/vectors/lz4_frame_reference_vectors_test.dart:204-229
    frame, _ = golden_frame("small_default")
    skippable = u32le(0x184D2A50) + u32le(4) + bytes([0xDE, 0xAD, 0xBE, 0xEF])
    assert oracle.frame_decode(skippable + frame) == ramp(1024)
    # concatenated frames (lz4_frame_decoder.dart:45-50)
    assert oracle.frame_decode(frame + frame) == ramp(1024) * 2


def test_golden_frames_with_lz4_cli(oracle, tmp_path):
    """The same frames decode with the real lz4 CLI present in the build image."""
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    if not os.path.exists(cli):
        pytest.skip("lz4 CLI not present")
    for name in ("independent_70k", "dependent_70k", "small_default", "empty"):
        frame, meta = golden_frame(name)
        p = tmp_path / (name + ".lz4")
        p.write_bytes(frame)
        out = subprocess.run([cli, "-d", "-c", str(p)], capture_output=True, check=True).stdout
        assert out == ramp(meta["payload_len"])


# ------------------------------------------------------- frame error matrix

HELLO = b"Hello"


def _bf(oracle, **kw):
    return build_frame(xxh32=oracle.xxh32, **kw)


def test_frame_decoder_unit_cases(oracle):
    # test/frame/lz4_frame_decoder_test.dart:99-190
    raw = [(HELLO, HELLO, True)]
    assert oracle.frame_decode(_bf(oracle, flg=0x60, bd=0x40, blocks=raw)) == HELLO
    assert oracle.frame_decode(_bf(oracle, flg=0x70, bd=0x40, blocks=raw)) == HELLO
    assert oracle.frame_decode(_bf(oracle, flg=0x64, bd=0x40, blocks=raw, add_content_checksum=True)) == HELLO
    dep = [(b"abcd", b"abcd", True), (bytes([0x00, 0x04, 0x00]), b"abcd", False)]
    assert oracle.frame_decode(_bf(oracle, flg=0x40, bd=0x40, blocks=dep)) == b"abcdabcd"


def test_frame_decoder_error_matrix(oracle):
    # test/frame/lz4_frame_decoder_test.dart:192-468
    raw = [(HELLO, HELLO, True)]
    one = [(b"A", b"A", True)]
    good = _bf(oracle, flg=0x60, bd=0x40, blocks=raw)

    bad = bytearray(good); bad[6] ^= 0xFF
    assert _err(oracle, oracle.frame_decode, bytes(bad)).message == "Header checksum mismatch"

    skip = u32le(0x184D2A50) + u32le(4) + bytes([1, 2, 3, 4])
    assert oracle.frame_decode(skip + _bf(oracle, flg=0x60, bd=0x40, blocks=[(b"OK", b"OK", True)])) == b"OK"

    e = _err(oracle, oracle.frame_decode, good, max_output_bytes=4)
    assert e.type_name == "Lz4OutputLimitException" and e.message == "Output limit exceeded"

    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x20, bd=0x40, blocks=one))
    assert e.type_name == "Lz4UnsupportedFeatureException" and e.message == "Unsupported LZ4 frame version"
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x62, bd=0x40, blocks=one))
    assert e.type_name == "Lz4FormatException" and e.message == "Reserved FLG bit is set"
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x60, bd=0x41, blocks=one))
    assert e.type_name == "Lz4FormatException" and e.message == "Reserved BD bits are set"
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x60, bd=0x30, blocks=one))
    assert e.type_name == "Lz4FormatException" and e.message == "Invalid block maximum size"
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x61, bd=0x40, blocks=one, dict_id=0x12345678))
    assert e.type_name == "Lz4UnsupportedFeatureException"
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x68, bd=0x40, blocks=raw, content_size=6))
    assert e.type_name == "Lz4CorruptDataException" and e.message == "Content size mismatch"

    f = bytearray(_bf(oracle, flg=0x70, bd=0x40, blocks=raw)); f[7 + 4 + 5] ^= 0xFF
    assert _err(oracle, oracle.frame_decode, bytes(f)).message == "Block checksum mismatch"
    f = bytearray(_bf(oracle, flg=0x64, bd=0x40, blocks=raw, add_content_checksum=True)); f[-1] ^= 0xFF
    assert _err(oracle, oracle.frame_decode, bytes(f)).message == "Content checksum mismatch"

    big = bytes(65537)
    e = _err(oracle, oracle.frame_decode, _bf(oracle, flg=0x60, bd=0x40, blocks=[(big, big, True)]))
    assert e.type_name == "Lz4CorruptDataException" and e.message == "Block size exceeds maximum"
    e = _err(oracle, oracle.frame_decode, good[:-1])
    assert e.type_name == "Lz4FormatException" and e.message == "Unexpected end of input"
    assert _err(oracle, oracle.frame_decode, b"\x01\x02\x03").message == "Unexpected end of input"
    assert _err(oracle, oracle.frame_decode, b"\x01\x02\x03\x04").message == "Invalid LZ4 frame magic number"


def test_frame_dictionary_resolver_contract(oracle):
    # test/frame/lz4_frame_dict_test.dart:62-125
    for indep in (True, False):
        flg = 0x40 | (0x20 if indep else 0) | 0x01
        f = _bf(oracle, flg=flg, bd=0x40, blocks=[(bytes([1, 2, 3]), bytes([1, 2, 3]), True)], dict_id=123)
        e = _err(oracle, oracle.frame_decode, f)
        assert e.type_name == "Lz4UnsupportedFeatureException"
        e = _err(oracle, oracle.frame_decode, f, dictionary_resolver=lambda i: None)
        assert e.type_name == "Lz4Exception" and e.message == "Dictionary not found for ID: 123"
        assert oracle.frame_decode(f, dictionary_resolver=lambda i: b"" if i == 123 else None) == bytes([1, 2, 3])


# ---------------------------------------------------- encoder known answers

def test_encoder_size_known_answers_repeating(oracle):
    """benchmark/baseline/macos.txt:41,44,47,50 — 1 MiB of i & 0x1f."""
    g = golden()["encoder_sizes_repeating_1MiB"]
    data = ramp(1 << 20)
    assert len(oracle.block_compress_fast(data)) == g["block_fast"]
    assert len(oracle.block_compress_hc(data)) == g["block_hc"]
    assert len(oracle.frame_encode(data, oracle.frame_options(block_size_id=4))) == g["frame64k_fast"]
    assert len(oracle.frame_encode(data, oracle.frame_options(block_size_id=4, compression=1))) == g["frame64k_hc"]


def test_encoder_size_known_answers_incompressible(oracle):
    """benchmark/baseline/macos.txt:20,26 — a 1 MiB input on which no probe hits."""
    g = golden()["encoder_sizes_random_1MiB"]
    data = np.random.default_rng(1234).integers(0, 256, size=1 << 20, dtype=np.uint8).tobytes()
    assert len(oracle.block_compress_fast(data)) == g["block_fast"]
    assert len(oracle.frame_encode(data, oracle.frame_options(block_size_id=4))) == g["frame64k"]


def test_encoder_byte_known_answers(oracle):
    """Hand-derived from lib/src/block/lz4_block_encoder.dart (not executed on Dart);
    'abcdabcd' coincides with the decoder vector test/block/lz4_block_decoder_test.dart:30-34."""
    assert oracle.block_compress_fast(b"") == b""
    assert oracle.block_compress_fast(bytes([1, 2, 3])) == bytes([0x30, 1, 2, 3])
    assert oracle.block_compress_fast(b"abcdabcd") == bytes([0x40, 0x61, 0x62, 0x63, 0x64, 0x04, 0x00])
    assert oracle.block_compress_fast(b"A" * 256) == bytes([0x1F, 0x41, 0x01, 0x00, 0xEC])
    assert oracle.block_compress_hc(b"A" * 256) == bytes([0x1F, 0x41, 0x01, 0x00, 0xEC])
    blk = oracle.block_compress_fast(ramp(65536))
    assert blk == bytes([0xFF, 0x71]) + ramp(128) + bytes([0x80, 0x00]) + b"\xff" * 256 + bytes([0x6D])
    assert _err(oracle, oracle.block_compress_fast, b"abc", None, 0).type_name == "RangeError"
    assert _err(oracle, oracle.block_compress_fast, b"", None, 0).type_name == "RangeError"
    assert _err(oracle, oracle.block_compress_hc, b"abc", None, 0).type_name == "RangeError"


def test_encoder_contracts_from_reference_tests(oracle):
    # test/block/lz4_block_encoder_test.dart:12-65, test/hc/lz4_hc_block_encoder_test.dart:14-69
    for data in (b"", b"a", b"ab", b"abc", b"A" * 256, b"Hello Hello Hello Hello World " * 20):
        for accel in (1, 2, 4):
            c = oracle.block_compress_fast(data, acceleration=accel)
            assert oracle.block_decompress(c, len(data)) == data
        h = oracle.block_compress_hc(data)
        assert oracle.block_decompress(h, len(data)) == data
    a64k = b"A" * 65536
    assert len(oracle.block_compress_hc(a64k)) <= len(oracle.block_compress_fast(a64k))
    # test/frame/lz4_frame_encoder_dict_test.dart: 32 B input matching a 64 B dict encodes small
    d = bytes(range(64))
    f = oracle.frame_encode(d[16:48], oracle.frame_options(block_size_id=4, dict_id=0x12345678), dictionary=d)
    assert len(f) < 30
    assert oracle.frame_decode(f, dictionary_resolver=lambda i: d) == d[16:48]
    # contentSize mismatch is raised after encoding (lz4_frame_encoder.dart:269-271)
    e = _err(oracle, oracle.frame_encode, b"abc", oracle.frame_options(content_size=4))
    assert e.type_name == "Lz4FormatException" and e.message == "contentSize does not match src length"


def test_test_input_txt_roundtrip(oracle):
    """BASELINE configs[0]: test_input.txt block round-trip, fast, acceleration=1."""
    data = open(os.path.join(GOLDEN_DIR, "test_input.txt"), "rb").read()
    assert len(data) == 247
    c = oracle.block_compress_fast(data)
    assert oracle.block_decompress(c, len(data)) == data


def test_frame_roundtrips_all_option_combinations(oracle):
    rng = np.random.default_rng(5)
    text = (b"the quick brown fox jumps over the lazy dog " * 4000)[:150000]
    rnd = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    for data in (b"", b"x", text, rnd, ramp(200000)):
        for indep in (True, False):
            for bck in (False, True):
                for cck in (False, True):
                    for comp in (0, 1):
                        o = oracle.frame_options(block_size_id=4, block_independence=indep, block_checksum=bck,
                                                 content_checksum=cck, compression=comp,
                                                 content_size=len(data) if cck else None,
                                                 hc_max_search_depth=8)
                        f = oracle.frame_encode(data, o)
                        assert oracle.frame_decode(f) == data
    # dependent blocks compress better than independent (test/frame/lz4_frame_encoder_test.dart:79-114)
    fi = oracle.frame_encode(ramp(300000), oracle.frame_options(block_size_id=4))
    fd = oracle.frame_encode(ramp(300000), oracle.frame_options(block_size_id=4, block_independence=False))
    assert len(fd) < len(fi)


def test_oracle_decodes_real_lz4_cli_output(oracle, tmp_path):
    """Frames written by the real lz4 CLI (v1.9.3 in this image) decode identically."""
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    if not os.path.exists(cli):
        pytest.skip("lz4 CLI not present")
    from tests.helpers import corpus
    data = corpus(24).tobytes()
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    for flags in (["-B4"], ["-B4", "-BX", "--content-size"], ["-B7", "-BD"], ["-9", "-B5", "--no-frame-crc"]):
        out = subprocess.run([cli, "-c", "-f"] + flags + [str(src)], capture_output=True, check=True).stdout
        assert oracle.frame_decode(out, capacity=len(data) + 64) == data
