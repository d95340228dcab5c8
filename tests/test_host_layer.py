"""CPU-only checks of the host layer: the C ABI library loads and exports every symbol the
header declares, option validation mirrors the reference, and the product path refuses to
run (loudly) without a GPU instead of falling back to anything."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lz4mi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lz4mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lz4_amd import _native as N
    lib = N.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(N.EXPORTS) == declared


def test_status_strings_match_reference_messages():
    from lz4_amd import _native as N
    L = N.load()
    expect = {
        1: ("Lz4FormatException", "Unexpected end of input"),
        2: ("Lz4FormatException", "Invalid LZ4 frame magic number"),
        3: ("Lz4FormatException", "Reserved FLG bit is set"),
        4: ("Lz4FormatException", "Reserved BD bits are set"),
        5: ("Lz4FormatException", "Invalid block maximum size"),
        6: ("Lz4FormatException", "Integer overflow"),
        7: ("Lz4FormatException", "contentSize does not match src length"),
        8: ("Lz4FormatException", "contentSize does not match stream length"),
        9: ("Lz4FormatException", "Input too short for size header"),
        42: ("Lz4UnsupportedFeatureException", "Content size exceeds supported integer range"),
        10: ("Lz4CorruptDataException", "Literal length exceeds output size"),
        11: ("Lz4CorruptDataException", "Match length exceeds output size"),
        12: ("Lz4CorruptDataException", "Trailing bytes after end of block"),
        13: ("Lz4CorruptDataException", "Invalid match distance"),
        14: ("Lz4CorruptDataException", "Header checksum mismatch"),
        15: ("Lz4CorruptDataException", "Block size exceeds maximum"),
        16: ("Lz4CorruptDataException", "Block checksum mismatch"),
        17: ("Lz4CorruptDataException", "Content checksum mismatch"),
        18: ("Lz4CorruptDataException", "Content size mismatch"),
        30: ("Lz4OutputLimitException", "Output limit exceeded"),
        40: ("Lz4UnsupportedFeatureException", "Unsupported LZ4 frame version"),
        41: ("Lz4UnsupportedFeatureException", "Dictionary ID present but no dictionary resolver provided"),
        50: ("Lz4Exception", "Dictionary not found for ID: "),
    }
    for code, (t, m) in expect.items():
        assert L.lz4mi_status_type(code).decode() == t
        assert L.lz4mi_status_message(code).decode() == m


def test_product_and_oracle_agree_on_status_codes(oracle):
    from lz4_amd import _native as N
    L, R = N.load(), oracle.lib()
    for code in list(range(0, 10)) + list(range(10, 21)) + [30, 40, 41, 42, 50]:
        assert L.lz4mi_status_message(code) == R.lz4ref_status_message(code)


def test_compress_bound_and_frame_bound():
    import ctypes as C
    from lz4_amd import _native as N
    L = N.load()
    assert L.lz4mi_compress_bound(0) == 16
    assert L.lz4mi_compress_bound(65536) == 65536 + 257 + 16
    o = N.FrameOptions()
    L.lz4mi_frame_options_default(C.byref(o))
    assert (o.block_size_id, o.block_independence, o.acceleration, o.hc_max_search_depth) == (7, 1, 1, 64)
    assert L.lz4mi_frame_bound(0, C.byref(o)) >= 15


def test_frame_header_bytes_match_golden_frames():
    """Header construction (host code, no GPU): same bytes as the lz4 CLI wrote."""
    import ctypes as C
    from lz4_amd import _native as N
    from lz4_amd.api import _c_options
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from tests.helpers import golden_frame
    L = N.load()
    cases = {
        "small_default": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, contentChecksum=True),
        "small_no_frame_crc": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB),
        "small_block_checksum": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True, contentChecksum=True),
        "small_content_size": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, contentChecksum=True, contentSize=1024),
        "independent_70k": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True, contentChecksum=True, contentSize=71680),
        "dependent_70k": Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False, blockChecksum=True, contentChecksum=True, contentSize=71680),
    }
    for name, opts in cases.items():
        frame, _ = golden_frame(name)
        o = _c_options(opts)
        buf = C.create_string_buffer(32)
        n = C.c_size_t(0)
        assert L.lz4mi_frame_header(C.byref(o), buf, 32, C.byref(n)) == 0
        assert buf.raw[: n.value] == frame[: n.value], name


def test_frame_decode_bound_walks_blocks():
    import ctypes as C
    from lz4_amd import _native as N
    from tests.helpers import golden_frame
    L = N.load()
    frame, _ = golden_frame("independent_70k")
    b = C.c_size_t(0)
    buf = C.create_string_buffer(frame, len(frame))
    assert L.lz4mi_frame_decode_bound(buf, len(frame), C.byref(b)) == 0
    # two compressed blocks of 299 and 66 payload bytes: min(blockMaxSize, 255 * payload + 32) each
    assert b.value == 65536 + (255 * 66 + 32)


def test_options_validation_mirrors_reference():
    # lib/src/frame/lz4_frame_options.dart:89-99, lib/src/hc/lz4_hc_options.dart:13-15
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions, Lz4HcOptions, RangeError
    o = Lz4FrameOptions()
    assert o.blockSize == Lz4FrameBlockSize.k4MB and o.blockIndependence and not o.blockChecksum
    assert not o.contentChecksum and o.contentSize is None and o.dictId is None and o.acceleration == 1
    assert [b.bdId for b in Lz4FrameBlockSize] == [4, 5, 6, 7]
    assert [b.maxBytes for b in Lz4FrameBlockSize] == [65536, 262144, 1048576, 4194304]
    for kw in (dict(acceleration=0), dict(contentSize=-1), dict(dictId=-1), dict(dictId=1 << 32)):
        with pytest.raises(RangeError):
            Lz4FrameOptions(**kw)
    with pytest.raises(RangeError):
        Lz4HcOptions(maxSearchDepth=0)
    assert Lz4HcOptions().maxSearchDepth == 64


def test_exception_string_form():
    from lz4_amd import Lz4CorruptDataException, Lz4Exception
    e = Lz4CorruptDataException("Invalid match distance")
    assert isinstance(e, Lz4Exception) and str(e) == "Lz4CorruptDataException: Invalid match distance"


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lz4_amd
    with pytest.raises(lz4_amd.Lz4BackendError):
        lz4_amd.lz4Compress(b"hello hello hello")
    with pytest.raises(lz4_amd.Lz4BackendError):
        lz4_amd.lz4FrameDecode(b"\x04\x22\x4d\x18\x60\x40\x82\x00\x00\x00\x00")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lz4_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "lz4ref" not in text and "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_frame_info_and_skippable_encode_match_oracle(oracle):
    """Host-only entry points (no GPU work): lz4mi_frame_info_parse and
    lz4mi_skippable_frame_encode against the oracle, including the reference's own cases
    (test/frame/lz4_frame_info_test.dart:8-91, lz4_skippable_frame_test.dart:8-98)."""
    import lz4_amd
    from tests.helpers import golden, golden_frame
    frames = [golden_frame(n)[0] for n in golden()["lz4_cli_frames"]]
    frames.append(oracle.frame_encode(bytes(100), oracle.frame_options(content_size=100, content_checksum=True)))
    frames.append(oracle.frame_encode(b"x", oracle.frame_options(block_size_id=5, dict_id=0xFFFFFFFF,
                                                                  block_independence=False, block_checksum=True)))
    frames.append(bytes([0x50, 0x2A, 0x4D, 0x18, 4, 0, 0, 0, 1, 2, 3, 4]))
    frames.append(bytes([0x02, 0x21, 0x4C, 0x18]))
    for f in frames:
        want = oracle.frame_info(f)
        got = lz4_amd.lz4FrameInfo(f)
        assert got.magic == want["magic"] and got.headerSize == want["header_size"]
        assert got.isSkippable == bool(want["is_skippable"]) and got.isLegacy == bool(want["is_legacy"])
        assert got.blockIndependence == bool(want["block_independence"])
        assert got.blockChecksum == bool(want["block_checksum"])
        assert got.contentChecksum == bool(want["content_checksum"])
        assert got.contentSize == (want["content_size"] if want["has_content_size"] else None)
        assert got.dictId == (want["dict_id"] if want["has_dict_id"] else None)
        assert got.blockMaxSize == want["block_max_size"]
        if got.isSkippable:
            assert got.skippableSize == want["skippable_size"]
    # every truncation and every single-byte corruption of a full header fails identically
    full = frames[-4][:15]
    variants = [full[:k] for k in range(16)]
    for k in range(4, 15):
        for bit in (0x01, 0x02, 0x40, 0x80):
            v = bytearray(full)
            v[k] ^= bit
            variants.append(bytes(v))
    for v in variants:
        try:
            want = ("ok", oracle.frame_info(v)["header_size"])
        except oracle.Lz4RefError as e:
            want = (e.type_name, e.message)
        try:
            got = ("ok", lz4_amd.lz4FrameInfo(v).headerSize)
        except lz4_amd.Lz4Exception as e:
            got = (type(e).__name__, e.message)
        assert got == want, v.hex()
    assert str(lz4_amd.lz4FrameInfo(frames[-2])) == "Lz4FrameInfo(type: skippable, size: 4)"
    assert str(lz4_amd.lz4FrameInfo(frames[-1])) == "Lz4FrameInfo(type: legacy)"
    for data, index in ((b"", 0), (bytes([1, 2, 3, 4, 5]), 0), (bytes([0xAB, 0xCD]), 15), (bytes(1000), 7)):
        assert lz4_amd.lz4SkippableEncode(data, index=index) == oracle.skippable_frame_encode(data, index)
    assert lz4_amd.lz4SkippableEncode(bytes([0xAB, 0xCD]), index=15) == bytes(
        [0x5F, 0x2A, 0x4D, 0x18, 2, 0, 0, 0, 0xAB, 0xCD])
    import pytest
    for bad in (-1, 16):
        with pytest.raises(lz4_amd.RangeError):
            lz4_amd.lz4SkippableEncode(b"", index=bad)


def test_xxh32_state_digest_of_short_inputs_needs_no_gpu(oracle):
    """Inputs below one stripe never reach the device: init/update/digest are host arithmetic
    on the caller's struct (lib/src/xxhash/xxh32.dart:60-64,127-163)."""
    import ctypes as C
    from lz4_amd import _native as N
    L = N.load()
    for data in (b"", b"a", b"abc", b"0123456789abcde"):
        s = N.XxhState()
        L.lz4mi_xxh32_state_init(C.byref(s), 0)
        for k in range(len(data)):  # one byte at a time: the carry path only
            assert L.lz4mi_xxh32_state_update(None, C.byref(s), data[k:k + 1], 1) == 0
        assert L.lz4mi_xxh32_state_digest(C.byref(s)) == oracle.xxh32(data)


def _split_args(text):
    """Top-level comma split of an argument list."""
    args, depth, cur = [], 0, ""
    for ch in text:
        if ch in "(<":
            depth += 1
        elif ch in ")>":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return [a for a in args if a and a != "void"]


def test_dart_shim_binds_declared_symbols_with_matching_arity():
    """The dart:ffi shim cannot be compiled here (no Dart SDK): hold it to the header mechanically.
    Every symbol it looks up is declared in include/lz4mi.h with the same number of parameters."""
    header = open(os.path.join(ROOT, "include", "lz4mi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decl = {m.group(1): len(_split_args(m.group(2)))
            for m in re.finditer(r"\b(lz4mi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}
    dart = open(os.path.join(ROOT, "lz4_amd", "dart", "dart_lz4_ffi.dart")).read()
    typedefs = {m.group(1): m.group(2) for m in
                re.finditer(r"typedef\s+(\w+)\s*=\s*[\w<>]+\s+Function\((.*?)\);", dart, flags=re.S)}
    seen = 0
    for m in re.finditer(r">\(\s*'(lz4mi_\w+)'\s*\)", dart):
        name = m.group(1)
        sigs = dart[dart.rindex("lookupFunction<", 0, m.start()) + len("lookupFunction<"):m.start()]
        assert name in decl, f"{name} is not declared in lz4mi.h"
        first = sigs.strip()
        fm = re.match(r"[\w<>]+\s+Function\(", first)
        if fm:  # inline native signature: take its balanced argument list
            depth, k = 1, fm.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(first[k], 0)
                k += 1
            nargs = len(_split_args(first[fm.end():k - 1]))
        else:
            alias = re.match(r"(\w+)", first).group(1)
            assert alias in typedefs, alias
            nargs = len(_split_args(typedefs[alias]))
        assert nargs == decl[name], (name, nargs, decl[name])
        seen += 1
    assert seen >= 25
