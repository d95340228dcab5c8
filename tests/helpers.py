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
# Synthetic mixed-entropy corpus: block class = blockIndex % 8.  Deterministic,
# every block can be generated independently from its index.

_MASK = (1 << 64) - 1


def _splitmix_stream(seed, n_words):
    """Vectorised splitmix64: returns n_words uint64 outputs for the given seed."""
    idx = np.arange(1, n_words + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _rand_bytes(seed, n):
    return _splitmix_stream(seed, (n + 7) // 8).view(np.uint8)[:n].copy()


_VOCAB = None


def _vocab():
    global _VOCAB
    if _VOCAB is None:
        rng = np.random.default_rng(12345)
        words = []
        for _ in range(4096):
            ln = int(rng.integers(2, 11))
            words.append(bytes(rng.integers(97, 123, size=ln, dtype=np.uint8)))
        ranks = np.arange(1, 4097, dtype=np.float64)
        p = ranks ** -1.1
        p /= p.sum()
        _VOCAB = (words, np.cumsum(p))
    return _VOCAB


def _text_block(seed, n):
    words, cdf = _vocab()
    u = (_splitmix_stream(seed, n // 3 + 16) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    idx = np.searchsorted(cdf, u)
    idx = np.minimum(idx, 4095)
    out = b" ".join(words[i] for i in idx)
    while len(out) < n:
        out += b" " + out
    return np.frombuffer(out[:n], dtype=np.uint8).copy()


def _lz_block(seed, n):
    r = _splitmix_stream(seed, n // 8 + 64)
    rb = r.view(np.uint8)
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    k = 0
    rpos = 0
    while pos < n:
        w = int(r[k]); k += 1
        lit = 1 + int(-24.0 * np.log(1.0 - ((w & 0xFFFF) + 0.5) / 65536.0))
        lit = min(lit, n - pos)
        if rpos + lit > rb.size:
            rpos = 0
        out[pos:pos + lit] = rb[rpos:rpos + lit]
        rpos += lit
        pos += lit
        if pos >= n:
            break
        mlen = 4 + ((w >> 16) & 0xFF)
        mlen = min(mlen, n - pos)
        bits = 1 + ((w >> 24) % 16)
        off = 1 + ((w >> 32) & ((1 << bits) - 1))
        off = min(off, pos)
        if off >= mlen:
            out[pos:pos + mlen] = out[pos - off:pos - off + mlen]
        else:
            for t in range(mlen):
                out[pos + t] = out[pos + t - off]
        pos += mlen
    return out


def corpus_block(index, n=65536, corpus_seed=0):
    """One block of the synthetic mixed-entropy corpus (class = index % 8)."""
    seed = (0x9E3779B97F4A7C15 ^ index ^ (corpus_seed << 32)) & _MASK
    cls = index % 8
    if cls == 0:
        return _rand_bytes(seed, n)
    if cls == 1:
        return (np.arange(n, dtype=np.uint32) & 0x1F).astype(np.uint8)
    if cls == 2:
        return np.full(n, 0x41, dtype=np.uint8)
    if cls in (3, 4):
        return _text_block(seed, n)
    if cls in (5, 6):
        return _lz_block(seed, n)
    a = _rand_bytes(seed, n)
    b = _text_block(seed ^ 0x5555, n)
    seg = (np.arange(n) // 4096) % 2 == 0
    return np.where(seg, a, b).astype(np.uint8)


def corpus(n_blocks, block=65536, corpus_seed=0, first=0):
    return np.concatenate([corpus_block(first + i, block, corpus_seed) for i in range(n_blocks)])
