"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Bit-exact for bytes, exact for exception type+message.

Run with `pytest -m gpu` on an MI355X.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.helpers import (GOLDEN_DIR, block_of_sequences, build_frame, corpus, corpus_block, golden, golden_frame, ramp,
                           u32le)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lz():
    import lz4_amd
    return lz4_amd


@pytest.fixture(scope="module")
def codec():
    import torch
    from lz4_amd.device import DeviceCodec
    assert torch.cuda.is_available()
    return DeviceCodec(0)


def _same_error(lz, oracle, gpu_call, ref_call):
    """Both raise, with the same Dart class name and message."""
    with pytest.raises(oracle.Lz4RefError) as ref:
        ref_call()
    with pytest.raises(Exception) as got:
        gpu_call()
    assert type(got.value).__name__ == ref.value.type_name, (got.value, ref.value)
    assert getattr(got.value, "message", None) == ref.value.message or ref.value.type_name == "RangeError"


# ------------------------------------------------------------------ xxh32

def test_xxh32_reference_vectors(lz):
    g = golden()["xxh32"]
    gen = g["gen_seed"]
    data = bytearray()
    for _ in range(g["data_len"]):
        data.append((gen >> 24) & 0xFF)
        gen = (gen * gen) & 0xFFFFFFFF
    for length, seed, expected in g["cases"]:
        assert lz.xxh32(bytes(data[:length]), seed) == expected


def test_xxh32_lengths_vs_oracle(lz, oracle):
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [63, 64, 65, 255, 256, 257, 1000, 4097, 65536, 300001]:
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for seed in (0, 0x9E3779B1):
            assert lz.xxh32(d, seed) == oracle.xxh32(d, seed), n


def test_xxh32_single_stream_on_both_engines(lz, codec, oracle):
    """One stream of 4 MiB and more is hashed by a host thread (host data where it is, device data from pinned
    pieces), smaller ones and LZ4MI_HASH_ENGINE=gpu by the device kernel: same word either way."""
    import torch
    rng = np.random.default_rng(12)
    for n in (4 * 1024 * 1024 - 1, 5 * 1024 * 1024 + 13, 33 * 1024 * 1024 + 7):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        want = oracle.xxh32(d.tobytes(), 0x1234567)
        t = torch.from_numpy(d).cuda()
        for engine in (None, "gpu", "host"):
            if engine:
                os.environ["LZ4MI_HASH_ENGINE"] = engine
            try:
                assert lz.xxh32(d.tobytes(), 0x1234567) == want, (n, engine)
                assert (int(codec.xxh32(t, 0x1234567)[0]) & 0xFFFFFFFF) == want, (n, engine)
            finally:
                os.environ.pop("LZ4MI_HASH_ENGINE", None)


@pytest.mark.parametrize("repeat", [3, 160])  # 48 blocks: a wavefront per block; 2560: sixteen blocks per wavefront
def test_xxh32_batch_vs_oracle(codec, oracle, repeat):
    import torch
    rng = np.random.default_rng(4)
    lens = [0, 1, 3, 15, 16, 17, 31, 32, 33, 100, 255, 4096, 65536 if repeat == 3 else 4111, 65537 if repeat == 3 else 8193,
            7, 64] * repeat
    offs, blob, pos = [], bytearray(), 0
    for n in lens:
        pos += int(rng.integers(0, 7))  # arbitrary alignment
        blob += bytes(pos - len(blob))
        offs.append(pos)
        blob += rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        pos += n
    t = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
    h = codec.xxh32_blocks(t, torch.tensor(lens, dtype=torch.int32, device="cuda"),
                           offsets=torch.tensor(offs, dtype=torch.int64, device="cuda"))
    got = h.cpu().numpy().view(np.uint32)
    for k, n in enumerate(lens):
        assert int(got[k]) == oracle.xxh32(bytes(blob[offs[k]:offs[k] + n])), (k, n)


# ---------------------------------------------------------- block decoder

def test_block_decoder_known_answers(lz):
    for v in golden()["block_decoder_ok"]:
        assert lz.lz4Decompress(bytes(v["src"]), decompressedSize=v["size"]) == bytes(v["out"]), v["name"]


def test_block_decoder_malformed_matches_reference_types(lz):
    for v in golden()["block_decoder_err"]:
        with pytest.raises(Exception) as e:
            lz.lz4Decompress(bytes(v["src"]), decompressedSize=v["size"])
        assert type(e.value).__name__ == v["type"], v["name"]


STRICT_CASES = [
    (b"", 4), (bytes([0x50, 1, 2]), 5), (bytes([0x50, 1, 2, 3, 4, 5]), 4), (bytes([0x40, 1, 2, 3, 4, 4, 0]), 5),
    (bytes([0x10, 1, 2]), 1), (bytes([0x10, 1, 5, 0]), 8), (bytes([0x00]), 0), (bytes([0xF0]), 20),
    (bytes([0xF0, 255]), 400), (bytes([0x1F, 0x41, 1, 0]), 30), (bytes([0x1F, 0x41, 1, 0, 255]), 600),
    (bytes([0x10, 0x41, 1]), 8), (bytes([0x40, 1, 2, 3, 4, 5, 0]), 12), (bytes([0x00, 0, 0]), 4),
]


def test_block_decoder_strict_error_parity(lz, oracle):
    for src, size in STRICT_CASES:
        _same_error(lz, oracle, lambda: lz.lz4Decompress(src, decompressedSize=size),
                    lambda: oracle.block_decompress(src, size))
    assert lz.lz4Decompress(b"", decompressedSize=0) == b""
    with pytest.raises(Exception) as e:
        lz.lz4Decompress(b"", decompressedSize=-1)
    assert type(e.value).__name__ == "RangeError"


LENIENT_CASES = [
    (bytes([0x50, 1, 2]), b"", 1), (bytes([0x50, 1, 2, 3, 4, 5]), b"", 4), (bytes([0x1F, 0x41, 9, 0, 200]), b"", 4),
    (bytes([0x1F, 0x41, 1, 0, 200]), b"", 4), (bytes([0x00, 4]), b"abcd", None), (bytes([0x00, 5, 0]), b"abcd", None),
    (bytes([0xF0, 255, 255]), b"", None), (bytes([0x0F, 1, 0, 255, 255, 255]), b"x", None),
]


def test_block_decoder_lenient_parity(lz, oracle):
    for src, prefix, maxlen in LENIENT_CASES:
        _same_error(lz, oracle, lambda: lz.lz4BlockDecompressInto(src, prefix, maxlen),
                    lambda: oracle.block_decompress_into(src, prefix, maxlen))
    ok = [(b"", b""), (bytes([0x00]), b""), (bytes([0x10, 0x41]), b""), (bytes([0x00, 4, 0]), b"abcd"),
          (bytes([0x00, 1, 0]), b"abcd"), (bytes([0x2F, 1, 2, 3, 0, 30]), b"abcd")]
    for src, prefix in ok:
        assert lz.lz4BlockDecompressInto(src, prefix) == oracle.block_decompress_into(src, prefix)


def _mutations(rng, blocks, per_block):
    out = []
    for src, size in blocks:
        out.append((src, size))
        for _ in range(per_block):
            b = bytearray(src)
            kind = rng.integers(0, 4)
            if kind == 0 and len(b):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1 and len(b) > 1:
                b = b[: int(rng.integers(0, len(b)))]
            elif kind == 2:
                b += bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
            elif len(b):
                p = int(rng.integers(0, len(b)))
                b[p:p + 2] = bytes(rng.integers(0, 256, size=2, dtype=np.uint8))
            out.append((bytes(b), size + int(rng.integers(-2, 3)) if rng.integers(0, 4) == 0 else size))
    return out


@pytest.fixture(params=[0, 1, 2], ids=["form-by-size", "one-wavefront", "two-wavefronts"])
def decode_form(request, codec):
    """The decoder's two launch forms (one or two wavefronts per block): results never depend on the form.
    The form is a setting of the context; the process default stays untouched."""
    from lz4_amd import _native as N
    L = N.load()
    prev = L.lz4mi_ctx_set_decode_form(codec.ctx.handle, request.param)
    assert prev >= -1
    yield request.param
    assert L.lz4mi_ctx_set_decode_form(codec.ctx.handle, prev) == request.param



def test_block_decoder_fuzz_status_parity(codec, oracle, decode_form):
    """Bit flips / truncations / garbage (the reference's fuzz idea,
    test/block/lz4_block_decoder_fuzz_test.dart:101-195): every block's status and,
    when it decodes, every byte equals the oracle's, strict and lenient."""
    import torch
    from lz4_amd import _native as N
    rng = np.random.default_rng(99)
    base = []
    for i in range(16):
        n = int(rng.integers(1, 3000))
        data = corpus_block(i, 4096).tobytes()[:n]
        base.append((oracle.block_compress_hc(data, max_search_depth=8), n))
        base.append((oracle.block_compress_fast(data), n))
    cases = _mutations(rng, base, 20)
    cases = [(s, max(0, z)) for s, z in cases]
    offs, blob = [], bytearray()
    for s, _ in cases:
        offs.append(len(blob))
        blob += s
    blob += bytes(64)
    comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
    sizes = torch.tensor([len(s) for s, _ in cases], dtype=torch.int32, device="cuda")
    offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
    expected = torch.tensor([z for _, z in cases], dtype=torch.int32, device="cuda")
    slot = 4096
    for strict in (True, False):
        out, out_len, status = codec.decompress_blocks(comp, sizes, slot, comp_offsets=offsets,
                                                       expected=expected, strict=strict)
        out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
        for k, (s, z) in enumerate(cases):
            try:
                ref = oracle.block_decompress(s, z) if strict else oracle.block_decompress_into(s, b"", z)
                ref_status = 0
            except oracle.Lz4RefError as e:
                ref, ref_status = None, e.status
            assert int(status[k]) == ref_status, (strict, k, s[:16].hex(), z)
            if ref is not None:
                assert int(out_len[k]) == len(ref)
                assert out[k, : len(ref)].tobytes() == ref
    assert N.OK == 0


LENGTHS_LIT = [0, 0, 1, 3, 14, 15, 16, 40, 269, 270, 271, 300, 525, 1100, 1799, 1800, 2300, 4347, 4352, 5000]
LENGTHS_MATCH = [4, 4, 5, 8, 18, 19, 20, 64, 273, 274, 275, 528, 529, 600, 1300, 1548, 1549, 4000]
DISTANCES = [1, 2, 3, 7, 31, 64, 200, 255, 256, 4095, 4096, 65535]


def test_block_decoder_hand_built_sequences(codec, oracle, decode_form):
    """Blocks put together from sequences with any number of length-extension bytes, long matches in front of long
    literal runs, overlapping matches and distances up to the window, strict and lenient (with a prefix), every
    byte and every status against the oracle.  The first block is one that a batch once ended wrongly on: the list
    of token starts leaves the real chain behind a match with three extension bytes, and the byte it lands on
    parses as a token that a batch cannot take."""
    import random

    import torch
    rng = random.Random(77)
    cases = [[(300, 4, 1), (0, 274, 200), (40, 600, 200), (16, 273, 64), (0, 1300, 200), (40, 274, 200), (4352, 18, 300)]]
    for _ in range(150):
        n = rng.choice([2, 5, 9, 30])
        seqs = [(rng.choice([5, 40, 300]), 4, 1)]
        seqs += [(rng.choice(LENGTHS_LIT), rng.choice(LENGTHS_MATCH), rng.choice(DISTANCES + [rng.randrange(1, 65536)]))
                 for _ in range(n)]
        if rng.random() < 0.7:
            seqs.append((rng.choice([1, 7, 5000]), 0, 0))  # (an empty last run is trailing bytes for the strict decoder)
        cases.append(seqs)
    for prefix in (b"", rng.randbytes(65536), rng.randbytes(300)):
        built = [block_of_sequences(c, rng, prefix) for c in cases]
        built = [(c, o) for c, o in built if len(o) <= 262144]
        slot = 262144
        offs, blob = [], bytearray()
        for c, _ in built:
            offs.append(len(blob))
            blob += c
        blob += bytes(64)
        comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
        sizes = torch.tensor([len(c) for c, _ in built], dtype=torch.int32, device="cuda")
        offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
        expected = torch.tensor([len(o) for _, o in built], dtype=torch.int32, device="cuda")
        d = torch.tensor(np.frombuffer(prefix, dtype=np.uint8).copy(), device="cuda") if prefix else None
        for strict in ((True, False) if not prefix else (False,)):
            out, out_len, status = codec.decompress_blocks(comp, sizes, slot, comp_offsets=offsets, expected=expected,
                                                           strict=strict, dictionary=d)
            out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
            for k, (c, o) in enumerate(built):
                ref = oracle.block_decompress(c, len(o)) if strict else oracle.block_decompress_into(c, prefix, len(prefix) + len(o))[len(prefix):]
                assert ref == o
                assert int(status[k]) == 0 and int(out_len[k]) == len(o), (len(prefix), strict, k, cases[k][:8])
                assert out[k, : len(o)].tobytes() == o, (len(prefix), strict, k, cases[k][:8])


def test_block_decoder_short_matches_from_the_end_of_a_prefix(codec, oracle, decode_form):
    """Hundreds of short sequences in a row (a token every three to five bytes: the decoder finds such tokens by
    pointer jumping over four chunks at once) whose matches of at most sixteen bytes come from the last bytes of the
    dictionary, far behind the output (one 16-byte load per match; a source within fifteen bytes of the prefix's
    end must not be read past that end), next to matches of the same shape inside the block."""
    import random

    import torch
    rng = random.Random(4242)
    prefix = rng.randbytes(65536)
    cases = []
    for _ in range(40):
        pos, seqs = 0, []
        first = rng.choice([4200, 5000, 9000])
        seqs.append((first, 4, 1))
        pos = first + 4
        for _ in range(rng.choice([150, 400])):
            lit, ml = rng.randrange(0, 3), rng.randrange(4, 17)
            pos += lit
            if rng.random() < 0.7:
                dist = pos + rng.randrange(1, 41)          # the last 40 bytes of the prefix
            else:
                dist = rng.randrange(4100, pos + 1) if pos > 4100 else 1  # far back inside the block
            seqs.append((lit, ml, min(dist, 65535)))
            pos += ml
        seqs.append((rng.choice([1, 7]), 0, 0))
        cases.append(seqs)
    built = [block_of_sequences(c, rng, prefix) for c in cases]
    slot = 65536
    assert all(len(o) <= slot for _, o in built)
    offs, blob = [], bytearray()
    for c, _ in built:
        offs.append(len(blob))
        blob += c
    blob += bytes(64)
    comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
    sizes = torch.tensor([len(c) for c, _ in built], dtype=torch.int32, device="cuda")
    offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
    expected = torch.tensor([len(o) for _, o in built], dtype=torch.int32, device="cuda")
    d = torch.tensor(np.frombuffer(prefix, dtype=np.uint8).copy(), device="cuda")
    out, out_len, status = codec.decompress_blocks(comp, sizes, slot, comp_offsets=offsets, expected=expected,
                                                   strict=False, dictionary=d)
    out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
    for k, (c, o) in enumerate(built):
        assert oracle.block_decompress_into(c, prefix, len(prefix) + len(o))[len(prefix):] == o
        assert int(status[k]) == 0 and int(out_len[k]) == len(o), k
        assert out[k, : len(o)].tobytes() == o, k


def test_block_decoder_short_sequence_blocks_fuzz(codec, oracle, decode_form):
    """Blocks made of thousands of short sequences (text through the HC encoder) take the
    decoder's second kernel, which decodes groups of sequences per step: status and bytes
    against the oracle for intact and damaged blocks, strict and lenient, with and without a
    dictionary prefix, full-size and ragged blocks, reference-style and spec-conformant ends."""
    import torch
    rng = np.random.default_rng(2024)
    dictionary = corpus_block(3 + 8 * 50).tobytes()[:40000]
    for use_dict in (False, True):
        base = []
        for i, (cls, n) in enumerate([(3, 65536), (7, 65536), (3, 70001), (4, 33333), (7, 9000), (3, 700)]):
            data = (corpus_block(cls + 8 * i).tobytes() + corpus_block(cls + 8 * (i + 9)).tobytes())[:n]
            d = dictionary if use_dict else None
            base.append((oracle.block_compress_hc(data, d, 16), n))
            base.append((oracle.block_compress_ex(data, d, level=1, param=4, conformant=True), n))
        cases = [(s, max(0, z)) for s, z in _mutations(rng, base, 10)]
        offs, blob = [], bytearray()
        for s, _ in cases:
            offs.append(len(blob))
            blob += s
        blob += bytes(64)
        comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
        sizes = torch.tensor([len(s) for s, _ in cases], dtype=torch.int32, device="cuda")
        offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
        expected = torch.tensor([z for _, z in cases], dtype=torch.int32, device="cuda")
        dict_t = torch.tensor(np.frombuffer(dictionary, dtype=np.uint8).copy(), device="cuda") if use_dict else None
        slot = 70016
        for strict in ((False,) if use_dict else (True, False)):  # the strict decoder takes no prefix
            out, out_len, status = codec.decompress_blocks(comp, sizes, slot, comp_offsets=offsets, expected=expected,
                                                           strict=strict, dictionary=dict_t)
            out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
            n_ok = 0
            for k, (s, z) in enumerate(cases):
                try:
                    if strict:
                        ref = oracle.block_decompress(s, z)
                    else:
                        pre = dictionary if use_dict else b""
                        ref = oracle.block_decompress_into(s, pre, len(pre) + z)[len(pre):]
                    ref_status = 0
                except oracle.Lz4RefError as e:
                    ref, ref_status = None, e.status
                assert int(status[k]) == ref_status, (use_dict, strict, k, z, len(s))
                if ref is not None:
                    n_ok += 1
                    assert int(out_len[k]) == len(ref)
                    assert out[k, : len(ref)].tobytes() == ref, (use_dict, strict, k)
            assert n_ok >= 12


def _lz4_len(v):
    out = bytearray()
    while v >= 255:
        out.append(255)
        v -= 255
    out.append(v)
    return bytes(out)


def _sequence(lits, dist, ml):
    """One LZ4 sequence: literals, then a match of ml >= 4 bytes at distance dist."""
    m4 = ml - 4
    out = bytearray([(min(len(lits), 15) << 4) | min(m4, 15)])
    if len(lits) >= 15:
        out += _lz4_len(len(lits) - 15)
    out += lits + bytes([dist & 255, dist >> 8])
    if m4 >= 15:
        out += _lz4_len(m4 - 15)
    return bytes(out)


def _last_literals(lits):
    out = bytearray([min(len(lits), 15) << 4])
    if len(lits) >= 15:
        out += _lz4_len(len(lits) - 15)
    return bytes(out + lits)


def test_block_decoder_long_overlapping_matches(codec, oracle, decode_form):
    """Matches longer than their distance (byte-serial copyMatch, lib/src/internal/byte_writer.dart:116-118)
    and longer than the decoder's short-copy path: every distance class (below / at / above a
    wavefront, around 16-byte alignment), lengths up to more than a block, sources that start inside
    a dictionary prefix, followed by a second match that reads the replicated bytes back."""
    import torch
    rng = np.random.default_rng(5)
    dictionary = bytes(rng.integers(0, 256, size=5000, dtype=np.uint8))
    dists = [1, 2, 3, 5, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 1000, 4097]
    lens = [257, 258, 300, 511, 1023, 4101, 8192, 20000, 70000]
    for use_dict in (False, True):
        cases = []
        for d in dists:
            for ml in lens:
                if ml <= d:
                    continue
                nlit = int(rng.integers(0, 40)) if use_dict else d + int(rng.integers(0, 40))
                lits = bytes(rng.integers(0, 256, size=nlit, dtype=np.uint8))
                d2 = int(rng.integers(1, min(nlit + ml, 65535) + 1))
                blk = _sequence(lits, d, ml) + _sequence(b"xy", d2, 19) + _last_literals(b"tail!")
                cases.append((blk, nlit + ml + 2 + 19 + 5))
        offs, blob = [], bytearray()
        for s_, _ in cases:
            offs.append(len(blob))
            blob += s_
        blob += bytes(64)
        comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
        sizes = torch.tensor([len(s_) for s_, _ in cases], dtype=torch.int32, device="cuda")
        offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
        expected = torch.tensor([z for _, z in cases], dtype=torch.int32, device="cuda")
        dict_t = torch.tensor(np.frombuffer(dictionary, dtype=np.uint8).copy(), device="cuda") if use_dict else None
        slot = 75008
        for strict in ((False,) if use_dict else (True, False)):
            out, out_len, status = codec.decompress_blocks(comp, sizes, slot, comp_offsets=offsets, expected=expected,
                                                           strict=strict, dictionary=dict_t)
            out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
            for k, (s_, z) in enumerate(cases):
                pre = dictionary if use_dict else b""
                ref = oracle.block_decompress(s_, z) if strict else oracle.block_decompress_into(s_, pre, len(pre) + z)[len(pre):]
                assert int(status[k]) == 0, (use_dict, strict, k)
                assert int(out_len[k]) == len(ref) == z
                assert out[k, :z].tobytes() == ref, (use_dict, strict, k)


@pytest.mark.parametrize("nb", [24000, 70001])
def test_block_decoder_queue_order_large_ragged_batch(codec, oracle, decode_form, nb):
    """More blocks than resident wavefronts, so the launch goes through the work queue and its
    order (24000 blocks: heavy blocks first, built by one workgroup; 70001: the mixed order of a long
    launch, built by the two-step kernels): blocks of very different sizes and kinds (literal-only,
    run-like, mixed, damaged) all come back with the oracle's bytes and status at their own index."""
    import torch
    rng = np.random.default_rng(77)
    kinds = []
    for i in range(24):
        n = int(rng.integers(1, 1500))
        data = corpus_block(i, 2048).tobytes()[:n]
        kinds.append((oracle.block_compress_fast(data), n))
        kinds.append((oracle.block_compress_hc(data, max_search_depth=4), n))
    kinds.append((_sequence(b"ab", 2, 1200) + _last_literals(b"z"), 1203))
    kinds.append((bytes([0x1F, 0x41, 9, 0, 200]), 300))  # distance beyond the output: corrupt
    pick = rng.integers(0, len(kinds), size=nb)
    offs, blob = [], bytearray()
    for k in pick:
        offs.append(len(blob))
        blob += kinds[int(k)][0]
    blob += bytes(64)
    comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
    sizes = torch.tensor([len(kinds[int(k)][0]) for k in pick], dtype=torch.int32, device="cuda")
    offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
    expected = torch.tensor([kinds[int(k)][1] for k in pick], dtype=torch.int32, device="cuda")
    out, out_len, status = codec.decompress_blocks(comp, sizes, 1536, comp_offsets=offsets, expected=expected, strict=True)
    out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
    refs = []
    for s_, z in kinds:
        try:
            refs.append((oracle.block_decompress(s_, z), 0))
        except oracle.Lz4RefError as e:
            refs.append((None, e.status))
    for i, k in enumerate(pick):
        ref, st = refs[int(k)]
        assert int(status[i]) == st, (i, int(k))
        if ref is not None:
            assert int(out_len[i]) == len(ref)
            assert out[i, : len(ref)].tobytes() == ref, (i, int(k))


# --------------------------------------------------------------- encoder

def test_encoder_byte_known_answers(lz):
    assert lz.lz4Compress(b"") == b""
    assert lz.lz4Compress(bytes([1, 2, 3])) == bytes([0x30, 1, 2, 3])
    assert lz.lz4Compress(b"abcdabcd") == bytes([0x40, 0x61, 0x62, 0x63, 0x64, 0x04, 0x00])
    assert lz.lz4Compress(b"A" * 256) == bytes([0x1F, 0x41, 0x01, 0x00, 0xEC])
    blk = lz.lz4Compress(ramp(65536))
    assert blk == bytes([0xFF, 0x71]) + ramp(128) + bytes([0x80, 0x00]) + b"\xff" * 256 + bytes([0x6D])
    assert len(lz.lz4Compress(ramp(1 << 20))) == golden()["encoder_sizes_repeating_1MiB"]["block_fast"]
    for bad in (b"abc", b""):
        with pytest.raises(Exception) as e:
            lz.lz4Compress(bad, acceleration=0)
        assert type(e.value).__name__ == "RangeError"


def test_encoder_small_inputs_vs_oracle(lz, oracle):
    rng = np.random.default_rng(21)
    text = b"Hello Hello Hello Hello World, hello world; HELLO hello " * 40
    inputs = [text[:n] for n in list(range(0, 40)) + [63, 64, 65, 100, 255, 256, 257, 511, 1000, 2000]]
    inputs += [rng.integers(0, 4, size=n, dtype=np.uint8).tobytes() for n in (5, 17, 64, 300, 5000)]
    inputs += [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in (4, 8, 100, 3000)]
    inputs += [open(os.path.join(GOLDEN_DIR, "test_input.txt"), "rb").read()]
    for data in inputs:
        for accel in (1, 2, 4, 7, 40, 1000):
            got = lz.lz4Compress(data, acceleration=accel)
            assert got == oracle.block_compress_fast(data, acceleration=accel), (len(data), accel)
            assert lz.lz4Decompress(got, decompressedSize=len(data)) == data


def test_encoder_corpus_blocks_vs_oracle_batch(codec, oracle):
    """64 KiB blocks of every corpus class, all in one launch: bytes identical to the oracle,
    then strict decode of those bytes gives the source back."""
    import torch
    nb = 96
    data = corpus(nb)
    src = torch.tensor(data, device="cuda")
    for accel in (1, 3):
        slots, sizes, status = codec.compress_blocks(src, 65536, acceleration=accel)
        assert int(status.abs().sum()) == 0
        h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(nb):
            ref = oracle.block_compress_fast(data[b * 65536:(b + 1) * 65536].tobytes(), acceleration=accel)
            assert int(h_sizes[b]) == len(ref), (b, accel)
            assert h_slots[b, : len(ref)].tobytes() == ref, (b, accel)
        out, out_len, st = codec.decompress_blocks(slots, sizes, 65536, strict=True)
        assert int(st.abs().sum()) == 0
        assert torch.equal(out.reshape(-1), src)


def test_encoder_ragged_last_block_and_sizes(codec, oracle):
    import torch
    for block, total in ((65536, 65536 * 3 + 1234), (4096, 4096 * 5 + 3), (1 << 20, (1 << 20) + 77), (4 << 20, 5 << 20)):
        nb = (total + block - 1) // block
        data = corpus((total + 65535) // 65536)[:total]
        src = torch.tensor(data, device="cuda")
        slots, sizes, status = codec.compress_blocks(src, block)
        assert int(status.abs().sum()) == 0
        h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(nb):
            ref = oracle.block_compress_fast(data[b * block:(b + 1) * block].tobytes())
            assert h_slots[b, : int(h_sizes[b])].tobytes() == ref, (block, b)


def _structured(rng, n):
    """Random bytes over a small alphabet with copies spliced in at all distances: short and long
    literal runs, matches that touch the block end, overlapping (run-length) copies."""
    a = rng.integers(0, int(rng.choice([2, 4, 16, 256])), size=n, dtype=np.uint8)
    for _ in range(int(rng.integers(0, 60))):
        ln = int(rng.integers(4, 600)); s0 = int(rng.integers(0, n)); d0 = int(rng.integers(0, n))
        ln = min(ln, n - s0, n - d0)
        a[d0:d0 + ln] = a[s0:s0 + ln].copy()
    if rng.integers(0, 3) == 0:  # period-k tail (matches running into the end of the block)
        k = int(rng.integers(1, 40)); t = int(rng.integers(1, min(n, 3000)))
        a[n - t:] = np.resize(a[:k], t)
    return a


def test_encoder_structured_fuzz_vs_oracle(codec, oracle):
    """Many blocks of structured random data at several block sizes, accelerations and with a
    shared dictionary: bytes identical to the oracle, then the decoder gives the source back."""
    import torch
    rng = np.random.default_rng(20261112)
    dict_np = _structured(rng, 40000)
    for block, nb, accel, use_dict in ((65536, 40, 1, False), (65536, 24, 2, True), (20000, 48, 1, False),
                                       (4096, 96, 7, True), (777, 128, 1, False), (65536, 16, 40, False)):
        data = np.concatenate([_structured(rng, block) for _ in range(nb)])
        data = data[: data.size - int(rng.integers(0, block - 1))]  # ragged last block
        src = torch.tensor(data, device="cuda")
        d_t = torch.tensor(dict_np, device="cuda") if use_dict else None
        slots, sizes, status = codec.compress_blocks(src, block, acceleration=accel, dictionary=d_t)
        assert int(status.abs().sum()) == 0
        h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(h_sizes.size):
            ref = oracle.block_compress_fast(data[b * block:(b + 1) * block].tobytes(), acceleration=accel,
                                             dictionary=dict_np.tobytes() if use_dict else None)
            assert h_slots[b, : int(h_sizes[b])].tobytes() == ref, (block, b, accel, use_dict)
        if not use_dict:
            out, out_len, st = codec.decompress_blocks(slots, sizes, block, strict=False)
            assert int(st.abs().sum()) == 0
            got = np.concatenate([out.cpu().numpy()[b, : int(out_len[b])] for b in range(h_sizes.size)])
            assert np.array_equal(got, data)


def test_encoder_entry_formats_and_short_periods_vs_oracle(codec, oracle):
    """Both table entry formats at the sizes where they change (positions of 17 bits carry a tag of the hashed
    sequence: block + dictionary up to 131 071 bytes), with data that repeats at short distances all over (several
    probes of one run in the same bucket) and with copies from all distances; bytes identical to the oracle."""
    import torch
    rng = np.random.default_rng(4711)

    def periodic(n):
        out = np.empty(n, dtype=np.uint8)
        p = 0
        while p < n:
            k = int(rng.choice([1, 2, 3, 5, 8, 13, 30, 61, 64, 100]))
            ln = min(n - p, int(rng.integers(8, 400)))
            unit = rng.integers(0, 256, size=k, dtype=np.uint8)
            out[p:p + ln] = np.resize(unit, ln)
            p += ln
            lit = min(n - p, int(rng.integers(0, 90)))
            out[p:p + lit] = rng.integers(0, 256, size=lit, dtype=np.uint8)
            p += lit
        return out

    dict_np = periodic(65535)
    for block, nb, accel, dlen in ((131071, 6, 1, 0), (131072, 6, 1, 0), (65536, 12, 1, 65535), (65537, 6, 2, 65535),
                                   (65536, 24, 1, 0), (65536, 12, 3, 0), (50000, 12, 9, 30000)):
        data = np.concatenate([periodic(block) if b % 2 else _structured(rng, block) for b in range(nb)])
        src = torch.tensor(data, device="cuda")
        d_np = dict_np[:dlen] if dlen else None
        d_t = torch.tensor(d_np, device="cuda") if dlen else None
        slots, sizes, status = codec.compress_blocks(src, block, acceleration=accel, dictionary=d_t)
        assert int(status.abs().sum()) == 0
        h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(nb):
            ref = oracle.block_compress_fast(data[b * block:(b + 1) * block].tobytes(), acceleration=accel,
                                             dictionary=d_np.tobytes() if dlen else None)
            assert int(h_sizes[b]) == len(ref) and h_slots[b, : len(ref)].tobytes() == ref, (block, b, accel, dlen)
        out, out_len, st = codec.decompress_blocks(slots, sizes, block, strict=False, dictionary=d_t)
        assert int(st.abs().sum()) == 0 and torch.equal(out.reshape(-1)[: data.size], src)


def test_encoder_sequence_shapes_vs_oracle(codec, oracle):
    """Inputs built as (random literal run, copy of earlier bytes) pairs whose lengths sit on the
    boundaries of the sequence format (_writeSequence / _writeLength, lib/src/block/lz4_block_encoder.dart:172-223):
    14/15/16-byte runs, 269/270/271 and 524/525 (one, two, three length bytes), runs longer than a
    wavefront and longer than the encoder's input window, copies of 4, 18/19/20, 273/274 and thousands
    of bytes, many more sequences than one write batch holds.  Bytes identical to the oracle."""
    import torch
    rng = np.random.default_rng(31)
    lits = [0, 1, 14, 15, 16, 63, 64, 65, 127, 128, 200, 201, 269, 270, 271, 524, 525, 780, 5000, 9000]
    copies = [4, 5, 18, 19, 20, 64, 273, 274, 528, 529, 3000, 9000]
    blocks = []
    for blk in range(12):
        a = bytearray(rng.integers(0, 256, size=300, dtype=np.uint8).tobytes())
        while len(a) < 200000:
            nl = int(rng.choice(lits)); nc = int(rng.choice(copies))
            a += rng.integers(0, 256, size=nl, dtype=np.uint8).tobytes()
            back = int(rng.integers(1, min(len(a), 65535) + 1))
            for k in range(nc):  # byte-serial so that short distances give overlapping copies
                a.append(a[len(a) - back])
        blocks.append(bytes(a[:196608]))
    for block, accel, d in ((65536, 1, None), (196608, 1, None), (65536, 3, blocks[0][:30000])):
        data = np.frombuffer(b"".join(blocks), dtype=np.uint8)
        src = torch.tensor(data, device="cuda")
        d_t = torch.tensor(np.frombuffer(d, dtype=np.uint8).copy(), device="cuda") if d else None
        slots, sizes, status = codec.compress_blocks(src, block, acceleration=accel, dictionary=d_t)
        assert int(status.abs().sum()) == 0
        h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(h_sizes.size):
            ref = oracle.block_compress_fast(data[b * block:(b + 1) * block].tobytes(), acceleration=accel, dictionary=d)
            assert int(h_sizes[b]) == len(ref), (block, b)
            assert h_slots[b, : len(ref)].tobytes() == ref, (block, b)


def test_single_block_of_20_mib_equals_the_oracle(lz, oracle):
    """lz4Compress takes any length (lib/dart_lz4.dart:60-72): positions past 2^24 use the wide table entries."""
    from lz4_amd import Lz4CompressionLevel
    data = corpus(320).tobytes()  # 20 MiB, every block class
    ref = oracle.block_compress_fast(data)
    got = lz.lz4Compress(data)
    assert got == ref
    assert lz.lz4Decompress(got, decompressedSize=len(data)) == data
    d = corpus_block(3).tobytes()[:40000]
    assert lz.lz4Compress(data[: 17 << 20], dictionary=d) == oracle.block_compress_fast(data[: 17 << 20], dictionary=d)
    # a small block right after on the same context: the slot's wide entries must not leak into it
    small = corpus_block(5).tobytes()
    assert lz.lz4Compress(small) == oracle.block_compress_fast(small)
    # HC past 2^24 positions (run-like data keeps one wavefront quick)
    big = ramp(9 << 20) + bytes(4 << 20) + ramp(4 << 20) + corpus_block(6).tobytes()
    assert lz.lz4Compress(big, level=Lz4CompressionLevel.hc) == oracle.block_compress_hc(big)


def test_encoder_with_dictionary_vs_oracle(lz, oracle):
    text = corpus_block(3).tobytes()
    d32k = text[:32768]
    cases = [(bytes(range(64)), bytes(range(16, 48))), (d32k, text[30000:36000]), (d32k, text[40000:41000]),
             (text[:70000], text[60000:65000]), (b"abcdefgh", b"abcdefghabcdefgh"), (d32k, b"ab"), (b"xy", text[:500])]
    for d, data in cases:
        got = lz.lz4Compress(data, dictionary=d)
        assert got == oracle.block_compress_fast(data, dictionary=d), (len(d), len(data))


# ---------------------------------------------------------------- frames

@pytest.mark.parametrize("name", ["independent_70k", "dependent_70k", "small_default", "small_no_frame_crc",
                                  "small_block_checksum", "small_content_size", "empty"])
def test_lz4_cli_golden_frames_decode(lz, name):
    frame, meta = golden_frame(name)
    assert lz.lz4FrameDecode(frame) == ramp(meta["payload_len"])


def test_skippable_and_concatenated_frames(lz):
    frame, _ = golden_frame("small_default")
    skippable = u32le(0x184D2A50) + u32le(4) + bytes([0xDE, 0xAD, 0xBE, 0xEF])
    assert lz.lz4FrameDecode(skippable + frame) == ramp(1024)
    assert lz.lz4FrameDecode(frame + skippable + frame) == ramp(1024) * 2


def test_frame_decoder_error_matrix_vs_oracle(lz, oracle):
    """Hand-built frames of test/frame/lz4_frame_decoder_test.dart:99-468."""
    bf = lambda **kw: build_frame(xxh32=oracle.xxh32, **kw)  # noqa: E731
    hello = b"Hello"
    raw = [(hello, hello, True)]
    one = [(b"A", b"A", True)]
    assert lz.lz4FrameDecode(bf(flg=0x60, bd=0x40, blocks=raw)) == hello
    assert lz.lz4FrameDecode(bf(flg=0x70, bd=0x40, blocks=raw)) == hello
    assert lz.lz4FrameDecode(bf(flg=0x64, bd=0x40, blocks=raw, add_content_checksum=True)) == hello
    dep = [(b"abcd", b"abcd", True), (bytes([0x00, 0x04, 0x00]), b"abcd", False)]
    assert lz.lz4FrameDecode(bf(flg=0x40, bd=0x40, blocks=dep)) == b"abcdabcd"

    good = bf(flg=0x60, bd=0x40, blocks=raw)
    bad_hc = bytearray(good); bad_hc[6] ^= 0xFF
    bad_bck = bytearray(bf(flg=0x70, bd=0x40, blocks=raw)); bad_bck[7 + 4 + 5] ^= 0xFF
    bad_cck = bytearray(bf(flg=0x64, bd=0x40, blocks=raw, add_content_checksum=True)); bad_cck[-1] ^= 0xFF
    big = bytes(65537)
    frames = [bytes(bad_hc), bf(flg=0x20, bd=0x40, blocks=one), bf(flg=0x62, bd=0x40, blocks=one),
              bf(flg=0x60, bd=0x41, blocks=one), bf(flg=0x60, bd=0x30, blocks=one),
              bf(flg=0x61, bd=0x40, blocks=one, dict_id=0x12345678),
              bf(flg=0x68, bd=0x40, blocks=raw, content_size=6), bytes(bad_bck), bytes(bad_cck),
              bf(flg=0x60, bd=0x40, blocks=[(big, big, True)]), good[:-1], good[:9], b"\x01\x02\x03", b"\x01\x02\x03\x04",
              # a corrupt compressed block followed by a structural error: the earlier block decides
              bf(flg=0x60, bd=0x40, blocks=[(bytes([0x00, 0, 0]), b"", False)])[:-2],
              bf(flg=0x60, bd=0x40, blocks=[(bytes([0x1F, 0x41, 1, 0]) + b"\xff" * 300 + b"\x00", b"", False)])]
    for f in frames:
        _same_error(lz, oracle, lambda: lz.lz4FrameDecode(f), lambda: oracle.frame_decode(f))
    _same_error(lz, oracle, lambda: lz.lz4FrameDecode(good, maxOutputBytes=4),
                lambda: oracle.frame_decode(good, max_output_bytes=4))
    assert lz.lz4FrameDecode(good, maxOutputBytes=5) == hello


def test_frame_dictionary_resolver_contract(lz, oracle):
    for indep in (True, False):
        flg = 0x40 | (0x20 if indep else 0) | 0x01
        f = build_frame(flg=flg, bd=0x40, blocks=[(bytes([1, 2, 3]), bytes([1, 2, 3]), True)], dict_id=123,
                        xxh32=oracle.xxh32)
        with pytest.raises(lz.Lz4UnsupportedFeatureException):
            lz.lz4FrameDecode(f)
        with pytest.raises(lz.Lz4Exception) as e:
            lz.lz4FrameDecode(f, dictionaryResolver=lambda i: None)
        assert type(e.value) is lz.Lz4Exception and e.value.message == "Dictionary not found for ID: 123"
        assert lz.lz4FrameDecode(f, dictionaryResolver=lambda i: b"" if i == 123 else None) == bytes([1, 2, 3])


def test_frame_encode_bytes_equal_oracle(lz, oracle):
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    rng = np.random.default_rng(8)
    datas = [b"", b"x", ramp(200000), corpus(5).tobytes()[:300001], rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()]
    for data in datas:
        for bs in (Lz4FrameBlockSize.k64KB, Lz4FrameBlockSize.k256KB):
            for bck in (False, True):
                for cck in (False, True):
                    opts = Lz4FrameOptions(blockSize=bs, blockChecksum=bck, contentChecksum=cck,
                                           contentSize=len(data) if cck else None, acceleration=1 + int(bck))
                    ref = oracle.frame_encode(data, oracle.frame_options(
                        block_size_id=bs.bdId, block_checksum=bck, content_checksum=cck,
                        content_size=len(data) if cck else None, acceleration=1 + int(bck)))
                    got = lz.lz4FrameEncodeWithOptions(data, options=opts)
                    assert got == ref, (len(data), bs, bck, cck)
                    assert lz.lz4FrameDecode(got) == data
    with pytest.raises(lz.Lz4FormatException) as e:
        lz.lz4FrameEncodeWithOptions(b"abc", options=Lz4FrameOptions(contentSize=4))
    assert e.value.message == "contentSize does not match src length"
    assert len(lz.lz4FrameEncodeWithOptions(ramp(1 << 20), options=Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB))) \
        == golden()["encoder_sizes_repeating_1MiB"]["frame64k_fast"]


def test_frame_4mib_blocks_with_checksums(lz, oracle):
    """BASELINE configs[2] shape at test size: 4 MiB blocks + block/content checksums."""
    from lz4_amd import Lz4FrameOptions
    data = corpus(160).tobytes()  # 10 MiB -> 3 blocks
    opts = Lz4FrameOptions(blockChecksum=True, contentChecksum=True)
    got = lz.lz4FrameEncodeWithOptions(data, options=opts)
    assert got == oracle.frame_encode(data, oracle.frame_options(block_checksum=True, content_checksum=True))
    assert lz.lz4FrameDecode(got) == data


def test_frame_dictionary_roundtrip_vs_oracle(lz, oracle):
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    text = corpus_block(3).tobytes()
    d = text[:32768]
    data = corpus(6, first=3).tobytes()
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, dictId=0x12345678)
    ref = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, dict_id=0x12345678), dictionary=d)
    got = lz.lz4FrameEncodeWithOptions(data, options=opts, dictionary=d)
    assert got == ref
    assert lz.lz4FrameDecode(got, dictionaryResolver=lambda i: d if i == 0x12345678 else None) == data
    # HC-compressed, dictionary-primed frame from the oracle (BASELINE configs[4] decode side)
    hc = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, dict_id=7, compression=1,
                                                        hc_max_search_depth=64), dictionary=d)
    assert lz.lz4FrameDecode(hc, dictionaryResolver=lambda i: d) == data
    # dependent-block frames from the oracle decode too
    dep = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, block_independence=False, compression=1,
                                                         hc_max_search_depth=16))
    assert lz.lz4FrameDecode(dep) == data


def test_decode_real_lz4_cli_frames(lz, tmp_path):
    """Frames written by the lz4 CLI of the image (v1.9.3): independent, linked, HC, 4 MiB blocks."""
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    if not os.path.exists(cli):
        pytest.skip("lz4 CLI not present")
    data = corpus(80).tobytes()
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    for flags in (["-B4"], ["-B4", "-BX", "--content-size"], ["-B7", "-BD"], ["-9", "-B5", "--no-frame-crc"],
                  ["-B7", "-BX", "--content-size"]):
        frame = subprocess.run([cli, "-c", "-f"] + flags + [str(src)], capture_output=True, check=True).stdout
        assert lz.lz4FrameDecode(frame) == data, flags


def test_decode_committed_lz4_cli_fixture(lz):
    path = os.path.join(GOLDEN_DIR, "lz4cli_v193_corpus8_B4_BX.lz4")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    assert lz.lz4FrameDecode(open(path, "rb").read()) == corpus(8).tobytes()


def test_our_frames_through_lz4_cli(lz, tmp_path):
    """Interop check of test/interop/lz4_cli_decode_our_frames_test.dart: frames whose blocks obey
    the LZ4 end-of-block rules (stored-raw blocks here) decode with the real CLI."""
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    if not os.path.exists(cli):
        pytest.skip("lz4 CLI not present")
    data = np.random.default_rng(77).integers(0, 256, size=200000, dtype=np.uint8).tobytes()
    for opts in (Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB),
                 Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True, contentChecksum=True,
                                 contentSize=len(data))):
        p = tmp_path / "f.lz4"
        p.write_bytes(lz.lz4FrameEncodeWithOptions(data, options=opts))
        out = subprocess.run([cli, "-d", "-c", str(p)], capture_output=True, check=True).stdout
        assert out == data


# ------------------------------------------------ size-independent properties

def test_large_batch_roundtrip_and_sampled_oracle(codec, oracle, decode_form):
    """256 MiB of corpus (4096 blocks): every block status OK, decode(encode(x)) == x on the
    device, and a strided sample of blocks is byte-identical to the oracle."""
    import torch
    nb = 4096
    data = corpus(nb)
    src = torch.tensor(data, device="cuda")
    slots, sizes, status = codec.compress_blocks(src, 65536)
    assert int(status.abs().sum()) == 0
    out, out_len, st = codec.decompress_blocks(slots, sizes, 65536, strict=True)
    assert int(st.abs().sum()) == 0
    assert torch.equal(out.reshape(-1), src)
    h_sizes = sizes.cpu().numpy()
    for b in range(0, nb, 37):
        ref = oracle.block_compress_fast(data[b * 65536:(b + 1) * 65536].tobytes())
        assert int(h_sizes[b]) == len(ref)
        assert slots[b, : len(ref)].cpu().numpy().tobytes() == ref
    # second pass over the same scratch (table generations are reused, not cleared)
    slots2, sizes2, _ = codec.compress_blocks(src, 65536)
    assert torch.equal(sizes2, sizes)
    assert torch.equal(slots2[7, : int(h_sizes[7])], slots[7, : int(h_sizes[7])])


def test_full_size_corpus_every_block_equals_the_oracle(codec, oracle):
    """BASELINE configs[1] at full size: all 16 384 blocks of the 1 GiB corpus (seed 0) through the fast
    encoder, every block compared with the oracle's bytes (oracle on a thread pool), then strict decode."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import torch
    nb = 16384
    data = corpus(nb)
    src = torch.from_numpy(data).cuda()
    slots, sizes, status = codec.compress_blocks(src, 65536)
    assert int(status.abs().sum()) == 0
    out, out_len, st = codec.decompress_blocks(slots, sizes, 65536, strict=True)
    assert int(st.abs().sum()) == 0 and torch.equal(out.reshape(-1), src)
    del out
    h_slots = slots.cpu().numpy()
    h_sizes = sizes.cpu().numpy()
    L = oracle.lib()
    bound = oracle.compress_bound(65536)
    base = data.ctypes.data

    def check(span):
        buf = C.create_string_buffer(bound)
        n = C.c_size_t(0)
        bad = []
        for b in range(*span):
            L.lz4ref_block_compress_fast(C.c_char_p(base + b * 65536), 65536, None, 0, 1, buf, bound, C.byref(n))
            if n.value != h_sizes[b] or h_slots[b, : n.value].tobytes() != buf.raw[: n.value]:
                bad.append(b)
        return bad

    spans = [(lo, min(nb, lo + 128)) for lo in range(0, nb, 128)]
    with ThreadPoolExecutor(16) as ex:
        bad = [b for part in ex.map(check, spans) for b in part]
    assert not bad, bad[:10]


def test_device_frame_equals_host_frame(codec, lz, oracle):
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    data = corpus(40)
    src = torch.tensor(data, device="cuda")
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True, contentChecksum=True)
    frame = codec.frame_encode(src, opts)
    ref = oracle.frame_encode(data.tobytes(), oracle.frame_options(block_size_id=4, block_checksum=True,
                                                                   content_checksum=True))
    assert frame.cpu().numpy().tobytes() == ref


def test_shard_assembly_equals_single_frame(codec, oracle):
    """Two shards compressed separately + offsets from the gathered record sizes == one frame."""
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    data = corpus(32)
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True)
    ref = oracle.frame_encode(data.tobytes(), oracle.frame_options(block_size_id=4, block_checksum=True))
    header = codec.frame_header(opts)
    half = 16 * 65536
    parts, recs = [], []
    for r in range(2):
        src = torch.tensor(data[r * half:(r + 1) * half], device="cuda")
        rec = codec.shard_compress(src, opts).clone()
        recs.append(rec)
        parts.append(src)
    allrec = torch.cat(recs).to(torch.int64)
    offs = torch.cumsum(allrec, 0) - allrec
    total = int(allrec.sum())
    body = torch.zeros(total, dtype=torch.uint8, device="cuda")
    for r in range(2):
        codec.shard_compress(parts[r], opts)
        base = int(offs[16 * r])
        codec.shard_write(parts[r], opts, (offs[16 * r:16 * (r + 1)] - base).contiguous(), body[base:])
    frame = header + body.cpu().numpy().tobytes() + u32le(0)
    assert frame == ref


# ------------------------------------------------------------------ HC encoder

def test_hc_encoder_known_answers_and_contracts(lz, oracle):
    from lz4_amd import Lz4CompressionLevel, Lz4HcOptions
    hc = Lz4CompressionLevel.hc
    assert lz.lz4Compress(b"", level=hc) == b""
    assert lz.lz4Compress(b"A" * 256, level=hc) == bytes([0x1F, 0x41, 0x01, 0x00, 0xEC])
    assert len(lz.lz4Compress(ramp(1 << 20), level=hc)) == golden()["encoder_sizes_repeating_1MiB"]["block_hc"]
    a64k = b"A" * 65536
    assert len(lz.lz4Compress(a64k, level=hc)) <= len(lz.lz4Compress(a64k))  # test/hc/...:33-42
    rng = np.random.default_rng(31)
    text = b"Hello Hello Hello Hello World, hello world; HELLO hello " * 60
    inputs = [text[:n] for n in (1, 3, 4, 5, 17, 100, 1000, 3000)]
    inputs += [rng.integers(0, 4, size=n, dtype=np.uint8).tobytes() for n in (64, 700, 5000)]
    inputs += [bytes(rng.integers(0, 256, size=2000, dtype=np.uint8)), ramp(5000), b"ab" * 3000]
    for data in inputs:
        for depth in (1, 4, 64):
            got = lz.lz4Compress(data, level=hc, hcOptions=Lz4HcOptions(maxSearchDepth=depth))
            assert got == oracle.block_compress_hc(data, max_search_depth=depth), (len(data), depth)
            assert lz.lz4Decompress(got, decompressedSize=len(data)) == data


def test_hc_encoder_corpus_blocks_vs_oracle(codec, oracle):
    import torch
    nb = 16
    data = corpus(nb)
    src = torch.tensor(data, device="cuda")
    slots, sizes, status = codec.compress_blocks(src, 65536, hc_depth=64)
    assert int(status.abs().sum()) == 0
    h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
    for b in range(nb):
        ref = oracle.block_compress_hc(data[b * 65536:(b + 1) * 65536].tobytes(), max_search_depth=64)
        assert h_slots[b, : int(h_sizes[b])].tobytes() == ref, b
    out, out_len, st = codec.decompress_blocks(slots, sizes, 65536, strict=True)
    assert int(st.abs().sum()) == 0 and torch.equal(out.reshape(-1), src)


def test_hc_frame_with_dictionary_vs_oracle(lz, oracle):
    """BASELINE configs[4] at test size: LZ4HC (depth 64) frame, 64 KiB blocks, dictId + 32 KiB dictionary."""
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions, Lz4HcOptions
    text = corpus_block(3).tobytes()
    d = text[:32768]
    data = corpus(6, first=2).tobytes()
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, dictId=0x12345678, compression=Lz4FrameCompression.hc,
                           hcOptions=Lz4HcOptions(maxSearchDepth=64))
    ref = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, dict_id=0x12345678, compression=1,
                                                         hc_max_search_depth=64), dictionary=d)
    got = lz.lz4FrameEncodeWithOptions(data, options=opts, dictionary=d)
    assert got == ref
    assert lz.lz4FrameDecode(got, dictionaryResolver=lambda i: d) == data
    assert len(lz.lz4FrameEncodeWithOptions(ramp(1 << 20), options=Lz4FrameOptions(
        blockSize=Lz4FrameBlockSize.k64KB, compression=Lz4FrameCompression.hc))) \
        == golden()["encoder_sizes_repeating_1MiB"]["frame64k_hc"]


def test_hc_encoder_chain_quirks_fuzz_vs_oracle(lz, oracle):
    """Inputs built to hit the corners of the reference's chain: matches extended backwards
    (positions inserted twice), runs of never-inserted positions, far candidates that only
    consume depth, and depth limits smaller than the bucket population."""
    from lz4_amd import Lz4CompressionLevel, Lz4HcOptions
    rng = np.random.default_rng(20261111)
    cases = []
    for k in range(40):
        n = int(rng.integers(200, 24000))
        alpha = int(rng.choice([2, 3, 4, 8, 16, 64]))
        a = rng.integers(0, alpha, size=n, dtype=np.uint8)
        if k % 3 == 0:  # splice in copies at random offsets so that long matches start mid-run
            for _ in range(int(rng.integers(1, 30))):
                ln = int(rng.integers(4, 400)); s0 = int(rng.integers(0, n - 1)); d0 = int(rng.integers(0, n - 1))
                ln = min(ln, n - s0, n - d0)
                a[d0:d0 + ln] = a[s0:s0 + ln].copy()
        cases.append(a.tobytes())
    far = bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
    cases.append(far + bytes(rng.integers(0, 256, size=70000, dtype=np.uint8)) + far + far[:100])  # > 65535 back
    cases.append((b"abcdefgh" * 40 + bytes(rng.integers(0, 256, size=50, dtype=np.uint8))) * 60)
    for idx, data in enumerate(cases):
        for depth in ((1, 3, 64) if idx % 2 else (2, 7, 1000)):
            got = lz.lz4Compress(data, level=Lz4CompressionLevel.hc, hcOptions=Lz4HcOptions(maxSearchDepth=depth))
            assert got == oracle.block_compress_hc(data, max_search_depth=depth), (idx, len(data), depth)


# ------------------------------------------------------------------ linked (dependent) block frames: encode

@pytest.mark.parametrize("hc", [False, True])
def test_linked_frame_encode_bytes_equal_oracle(lz, oracle, hc):
    """blockIndependence=false (lz4_frame_encoder.dart:170-206): block 0 uses the external
    dictionary, every later block the 64 KiB in front of it.  Bytes must equal the oracle's, with
    and without a dictionary, for a block-aligned and a ragged length, and decode back."""
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions, Lz4HcOptions
    base = corpus(7, first=1).tobytes()          # ramp, run, text, text, lz, lz, splice
    rep = (corpus_block(3).tobytes()[:30000]) * 9  # text repeated across block borders: matches into the history
    d = corpus_block(4).tobytes()[:20000]
    for data in (base, rep[:200001], base[:65536 * 2 + 777]):
        for dictionary in (None, d):
            for bs_id, bs in ((4, Lz4FrameBlockSize.k64KB), (5, Lz4FrameBlockSize.k256KB)):
                if hc and bs_id == 5 and dictionary is None:
                    continue  # keep the CPU oracle's HC time down
                opts = Lz4FrameOptions(blockSize=bs, blockIndependence=False, blockChecksum=True,
                                       dictId=77 if dictionary else None,
                                       compression=Lz4FrameCompression.hc if hc else Lz4FrameCompression.fast,
                                       hcOptions=Lz4HcOptions(maxSearchDepth=16))
                ref = oracle.frame_encode(data, oracle.frame_options(
                    block_size_id=bs_id, block_independence=False, block_checksum=True,
                    dict_id=77 if dictionary else None, compression=1 if hc else 0, hc_max_search_depth=16),
                    dictionary=dictionary)
                got = lz.lz4FrameEncodeWithOptions(data, options=opts, dictionary=dictionary)
                assert got == ref, (len(data), dictionary is not None, bs_id)
                assert lz.lz4FrameDecode(got, dictionaryResolver=(lambda i: d) if dictionary else None) == data
