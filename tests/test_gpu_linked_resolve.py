"""GPU: linked-block frames decoded all blocks at once (size walk, literals + distances, pointer doubling,
gather) against the oracle's decoder, and against the serial chain of the same library.

Frames come from the oracle's frame encoder (lib/src/frame/lz4_frame_encoder.dart:170-206 restated) and from the
library's own; every decode goes through the C ABI (lz4mi_frame_decode_next).  Bit-exact output; for damaged
frames the same exception (class and message) as the oracle's decoder, which is what the chain reports.
"""
import os
import random

import numpy as np
import pytest

import lz4_amd
from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions
from tests.helpers import build_frame, corpus_block, lz4_length as _length

pytestmark = pytest.mark.gpu


def mixed(n, seed):
    parts, k = [], 0
    while sum(map(len, parts)) < n:
        parts.append(corpus_block((3, 5, 0, 1, 6, 7, 2, 4)[k % 8] + 8 * (seed + k), 65536).tobytes())
        k += 1
    return b"".join(parts)[:n]


def decode_both(frame, require=True, **kw):
    """(all-at-once result, chain result); each a bytes object or the exception raised.  With `require` a frame
    that the all-at-once path hands to the chain is an error (LZ4MI_LINK_RESOLVE=2)."""
    out = []
    for flag in ("2" if require else "1", "0"):
        os.environ["LZ4MI_LINK_RESOLVE"] = flag
        try:
            out.append(lz4_amd.lz4FrameDecode(frame, **kw))
        except (lz4_amd.Lz4Exception, lz4_amd.RangeError) as e:
            out.append(e)
        finally:
            os.environ.pop("LZ4MI_LINK_RESOLVE", None)
    return out


def same(a, b):
    if isinstance(a, bytes) or isinstance(b, bytes):
        return a == b
    return type(a) is type(b) and a.message == b.message


DATA = {
    "mixed_3MiB": lambda: mixed(3 * 1024 * 1024 + 12345, 3),
    "zeros_2MiB": lambda: bytes(2 * 1024 * 1024),
    "period3_1MiB": lambda: (b"abc" * 400000)[:1 << 20],
    "period_70001": lambda: (mixed(70001, 9) * 20)[:1 << 20],
    "random_600k": lambda: random.Random(4).randbytes(600000),
    "text_lz_900k": lambda: b"".join(corpus_block(6 + 8 * k, 65536).tobytes() for k in range(14))[:900000],
    "short_tail": lambda: mixed(5 * 65536 + 1, 11),
}


@pytest.mark.parametrize("name", sorted(DATA))
@pytest.mark.parametrize("bs", [Lz4FrameBlockSize.k64KB, Lz4FrameBlockSize.k256KB])
def test_linked_frames_decode_like_the_oracle(oracle, name, bs):
    data = DATA[name]()
    for cc, bc in ((True, False), (False, True)):
        o = oracle.frame_options(block_size_id=bs.bdId, block_independence=False, content_checksum=cc,
                                 block_checksum=bc, compression=1)
        frame = oracle.frame_encode(data, o)
        assert oracle.frame_decode(frame) == data
        fast, chain = decode_both(frame)
        assert fast == data and chain == data


@pytest.mark.parametrize("chunk", [65536, 200000, 1 << 20])
def test_frames_resolved_in_several_chunks(oracle, chunk):
    """References that cross chunk borders: the bytes in front of a chunk are final when it is resolved."""
    dictionary = mixed(50000, 70)
    data = dictionary[-20000:] + mixed(1500000, 71) + bytes(300000) + (b"0123456789abcdef" * 20000) + mixed(400000, 72)
    o = oracle.frame_options(block_size_id=4, block_independence=False, dict_id=3, content_checksum=True)
    frame = oracle.frame_encode(data, o, dictionary)
    os.environ["LZ4MI_LINK_CHUNK"] = str(chunk)
    try:
        fast, chain = decode_both(frame, dictionaryResolver=lambda i: dictionary)
    finally:
        os.environ.pop("LZ4MI_LINK_CHUNK", None)
    assert fast == data and chain == data


def test_linked_frames_of_the_library_round_trip(oracle):
    data = mixed(2 * 1024 * 1024 + 77, 21)
    for comp in (Lz4FrameCompression.fast, Lz4FrameCompression.hc):
        o = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False, contentChecksum=True,
                            compression=comp)
        frame = lz4_amd.lz4FrameEncodeWithOptions(data, options=o)
        assert oracle.frame_decode(frame) == data
        fast, chain = decode_both(frame)
        assert fast == data and chain == data


@pytest.mark.parametrize("dict_len", [100, 40000, 100000])
def test_linked_frames_with_a_dictionary(oracle, dict_len):
    dictionary = mixed(dict_len, 40)
    data = dictionary[-30000:] + mixed(700000, 41) + dictionary[:5000]
    o = oracle.frame_options(block_size_id=4, block_independence=False, dict_id=9, content_checksum=True)
    frame = oracle.frame_encode(data, o, dictionary)
    assert oracle.frame_decode(frame, dictionary_resolver=lambda i: dictionary) == data
    fast, chain = decode_both(frame, dictionaryResolver=lambda i: dictionary)
    assert fast == data and chain == data


def test_linked_frame_with_stored_blocks_between_compressed_ones(oracle):
    rnd = random.Random(8)
    data = mixed(200000, 50) + rnd.randbytes(150000) + mixed(200000, 50) + rnd.randbytes(70000) + mixed(100000, 51)
    o = oracle.frame_options(block_size_id=4, block_independence=False, content_checksum=True)
    frame = oracle.frame_encode(data, o)
    fast, chain = decode_both(frame)
    assert fast == data and chain == data


def test_output_limit_and_damage_report_like_the_chain(oracle):
    data = mixed(1024 * 1024, 60)
    o = oracle.frame_options(block_size_id=4, block_independence=False, content_checksum=True, block_checksum=False)
    frame = bytearray(oracle.frame_encode(data, o))
    # an output limit in the middle of the frame
    fast, chain = decode_both(bytes(frame), require=False, maxOutputBytes=300000)
    assert not isinstance(fast, bytes) and same(fast, chain)
    # damage inside block payloads: whatever the walk makes of it, the report is the chain's (= the reference's)
    rnd = random.Random(61)
    differing = 0
    for trial in range(24):
        bad = bytearray(frame)
        at = rnd.randrange(20, len(bad) - 8)
        bad[at] ^= 1 << rnd.randrange(8)
        fast, chain = decode_both(bytes(bad), require=False)
        assert same(fast, chain), (trial, at, fast, chain)
        try:
            want = oracle.frame_decode(bytes(bad))
        except oracle.Lz4RefError as e:
            want = e
        if isinstance(want, bytes):
            assert fast == want
        else:
            assert not isinstance(fast, bytes) and type(fast).__name__ == want.type_name and fast.message == want.message
            differing += 1
    assert differing > 0


# ------------------------------------------------------------------ hand-built sequences
def synthetic_block(rng, history, target, end_after_match):
    """A compressed block of `target` decoded bytes made of sequences no encoder here writes: literal and match
    lengths with up to four extension bytes, distances from 1 to everything the block may reach, overlapping
    matches, and optionally an end right behind a match (lz4BlockDecompressInto returns at end of input)."""
    out = bytearray()       # decoded bytes of this block
    comp = bytearray()
    hist = bytes(history[-65536:])
    while len(out) < target:
        left = target - len(out)
        ll = min(left, rng.choice([0, 0, 1, 3, 14, 15, 16, 40, 269, 270, 271, 300, 525, 1100, 5000]))
        lits = rng.randbytes(ll)
        reach = len(hist) + len(out) + ll
        want_match = left - ll >= 4 and reach > 0
        if not want_match:
            ll = left  # the block's last sequence: literals only
            lits = rng.randbytes(ll)
            nib, ext = _length(0, ll)
            comp += bytes([nib << 4]) + ext + lits
            out += lits
            break
        ml = min(left - ll, rng.choice([4, 4, 5, 8, 18, 19, 20, 64, 273, 274, 275, 600, 1300, 4000]))
        dist = rng.choice([1, 2, 3, 7, 31, 64, 255, 256, 4095, 65535, rng.randrange(1, 65536), reach])
        dist = max(1, min(dist, reach, 65535))
        lnib, lext = _length(0, ll)
        mnib, mext = _length(4, ml)
        comp += bytes([(lnib << 4) | mnib]) + lext + lits + bytes([dist & 255, dist >> 8]) + mext
        out += lits
        for _ in range(ml):  # byte_writer.dart:116-118
            p = len(out) - dist
            out.append(out[p] if p >= 0 else hist[len(hist) + p])
        if end_after_match and len(out) >= target - 8 and len(out) > 0:
            break
    return bytes(comp), bytes(out)


@pytest.mark.parametrize("seed", range(6))
def test_hand_built_linked_frames_decode_like_the_oracle(oracle, seed):
    rng = random.Random(1000 + seed)
    dictionary = rng.randbytes(rng.choice([0, 0, 100, 70000]))
    history = bytearray(dictionary)
    blocks = []
    for b in range(rng.choice([4, 7, 12])):
        if rng.random() < 0.15:  # a stored block between compressed ones
            raw = rng.randbytes(rng.choice([1, 5000, 65536]))
            blocks.append((raw, raw, True))
            history += raw
            continue
        target = rng.choice([65536, 65536, 65536, 1, 17, 30000])
        comp, dec = synthetic_block(rng, history, target, end_after_match=rng.random() < 0.3)
        assert len(dec) <= 65536
        blocks.append((comp, dec, False))
        history += dec
    flg = 0x40 | 0x04 | (0x01 if dictionary else 0)  # version 01, linked blocks, content checksum
    frame = build_frame(flg, 0x40, blocks, oracle.xxh32, add_content_checksum=True, dict_id=5 if dictionary else None)
    want = bytes(history[len(dictionary):])
    kw = {"dictionaryResolver": (lambda i: dictionary)} if dictionary else {}
    okw = {"dictionary_resolver": (lambda i: dictionary)} if dictionary else {}
    assert oracle.frame_decode(frame, **okw) == want
    fast, chain = decode_both(frame, require=len(blocks) >= 4, **kw)
    assert fast == want and chain == want
