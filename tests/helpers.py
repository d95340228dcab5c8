"""Shared helpers for the test-suite (host-side only, no GPU use)."""
import base64
import json
import os
import struct

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
        return json.load(f)


def golden_frame(name):
    g = golden()["lz4_cli_frames"][name]
    return base64.b64decode(g["b64"]), g


def ramp(n):
    """i & 0x1f payload (benchmark/lz4_benchmark.dart:263-269, vectors test :43-49)."""
    return (np.arange(n, dtype=np.uint32) & 0x1F).astype(np.uint8).tobytes()


def u32le(v):
    return struct.pack("<I", v & 0xFFFFFFFF)


def build_frame(flg, bd, blocks, xxh32, add_content_checksum=False, content_size=None, dict_id=None):
    """Hand-built frame, same construction as test/frame/lz4_frame_decoder_test.dart:37-97.
    blocks: list of (payload: bytes, decoded: bytes, is_uncompressed: bool)."""
    desc = bytes([flg, bd])
    if (flg >> 3) & 1:
        desc += struct.pack("<Q", content_size or 0)
    if flg & 1:
        desc += u32le(dict_id or 0)
    hc = (xxh32(desc) >> 8) & 0xFF
    out = bytearray(u32le(0x184D2204) + desc + bytes([hc]))
    decoded_all = bytearray()
    for payload, decoded, raw in blocks:
        out += u32le((0x80000000 if raw else 0) | len(payload))
        out += payload
        if (flg >> 4) & 1:
            out += u32le(xxh32(payload))
        decoded_all += decoded
    out += u32le(0)
    if add_content_checksum:
        out += u32le(xxh32(bytes(decoded_all)))
    return bytes(out)


# ---------------------------------------------------------------- corpus
from tools.corpus import corpus, corpus_block  # noqa: E402,F401  (C generator, tools/corpus.c)


# ---------------------------------------------------------------- hand-built compressed blocks
def lz4_length(base, n):
    """Token nibble and extension bytes of a length (lib/src/block/lz4_block_decoder.dart:121-130)."""
    if n < base + 15:
        return n - base, b""
    rest = n - base - 15
    return 15, bytes([255] * (rest // 255) + [rest % 255])


def block_of_sequences(seqs, rng, prefix=b""):
    """(compressed block, decoded bytes) for a list of (literal length, match length, distance); a match length of 0
    ends the block with that literal run.  Distances are clamped to what the block may reach (prefix included).
    Sequences no encoder here writes: any number of extension bytes, any distance, overlapping matches."""
    out, comp = bytearray(), bytearray()
    for ll, ml, dist in seqs:
        lits = rng.randbytes(ll)
        lnib, lext = lz4_length(0, ll)
        if ml == 0:
            comp += bytes([lnib << 4]) + lext + lits
            out += lits
            break
        dist = max(1, min(dist, len(prefix) + len(out) + ll, 65535))
        mnib, mext = lz4_length(4, ml)
        comp += bytes([(lnib << 4) | mnib]) + lext + lits + bytes([dist & 255, dist >> 8]) + mext
        out += lits
        for _ in range(ml):  # lib/src/internal/byte_writer.dart:116-118
            p = len(out) - dist
            out.append(out[p] if p >= 0 else prefix[len(prefix) + p])
    return bytes(comp), bytes(out)
