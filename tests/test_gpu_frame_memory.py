"""Frame decode whose device and host memory follows the data, not blocks x blockMaxSize
(lib/src/frame/lz4_frame_decoder.dart:243-296 appends whatever each block yields), and the
argument checks of the two shard calls.  Bit-exact against the oracle; errors by class + message.
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests.helpers import build_frame, corpus, ramp, u32le

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


def _same(lz, oracle, frame, **kw):
    """Same bytes, or the same exception class and message."""
    okw = {}
    if "maxOutputBytes" in kw:
        okw["max_output_bytes"] = kw["maxOutputBytes"]
    try:
        ref = oracle.frame_decode(frame, **okw)
    except oracle.Lz4RefError as e:
        with pytest.raises(lz.Lz4Exception) as got:
            lz.lz4FrameDecode(frame, **kw)
        assert type(got.value).__name__ == e.type_name and got.value.message == e.message
        return None
    got = lz.lz4FrameDecode(frame, **kw)
    assert got == ref
    return got


def _tiny_blocks(oracle, count, seed):
    """(payload, decoded, raw) records of at most 200 payload bytes: stored, compressed, and empty."""
    rng = np.random.default_rng(seed)
    blocks = []
    for k in range(count):
        kind = k % 4
        if kind == 0:
            d = rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8).tobytes()
            blocks.append((d, d, True))
        elif kind == 1:
            d = ramp(int(rng.integers(40, 3000)))
            blocks.append((oracle.block_compress_fast(d), d, False))
        elif kind == 2:
            d = bytes([int(rng.integers(0, 256))]) * int(rng.integers(5, 20000))
            blocks.append((oracle.block_compress_fast(d), d, False))
        else:
            blocks.append((b"\x00", b"", False))  # a lone token: decodes to nothing
    assert all(len(p) <= 200 for p, _, _ in blocks)
    return blocks


def test_thousands_of_tiny_blocks_under_4mib_block_size(lz, oracle):
    import torch
    blocks = _tiny_blocks(oracle, 6000, 1)
    total = sum(len(d) for _, d, _ in blocks)
    free0 = torch.cuda.mem_get_info()[0]
    for flg in (0x60, 0x70, 0x64):  # plain, block checksums, content checksum
        frame = build_frame(flg=flg, bd=0x70, blocks=blocks, xxh32=oracle.xxh32, add_content_checksum=flg == 0x64)
        assert len(_same(lz, oracle, frame)) == total
        assert _same(lz, oracle, frame, maxOutputBytes=total) is not None
        assert _same(lz, oracle, frame, maxOutputBytes=total - 1) is None
        assert _same(lz, oracle, frame, maxOutputBytes=1000) is None
    # 6000 blocks x 4 MiB would be 23 GiB of slots; what is held afterwards follows the data
    assert free0 - torch.cuda.mem_get_info()[0] < (1 << 30)


def test_two_hundred_thousand_empty_blocks(lz, oracle):
    """5-byte records that each decode to nothing (the reference decodes this to an empty list)."""
    import torch
    free0 = torch.cuda.mem_get_info()[0]
    frame = build_frame(flg=0x60, bd=0x70, blocks=[(b"\x00", b"", False)] * 200000, xxh32=oracle.xxh32)
    assert lz.lz4FrameDecode(frame) == b""
    assert lz.lz4FrameDecode(frame, maxOutputBytes=0) == b""
    assert free0 - torch.cuda.mem_get_info()[0] < (1 << 30)
    # and the same with one block that fails in the middle
    bad = [(b"\x00", b"", False)] * 1000 + [(bytes([0x00, 0, 0]), b"", False)] + [(b"\x00", b"", False)] * 10
    _same(lz, oracle, build_frame(flg=0x60, bd=0x70, blocks=bad, xxh32=oracle.xxh32))


def test_decode_in_several_launches(lz, oracle, monkeypatch):
    """A slot budget of 1 MiB forces one launch per handful of blocks; bytes and errors stay the same."""
    monkeypatch.setenv("LZ4MI_DECODE_SCRATCH_MB", "1")
    data = corpus(40).tobytes()
    for indep in (True, False):
        frame = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, block_checksum=True,
                                                               content_checksum=True, block_independence=indep))
        assert _same(lz, oracle, frame) == data
        assert _same(lz, oracle, frame, maxOutputBytes=len(data) - 1) is None
        assert _same(lz, oracle, frame, maxOutputBytes=700000) is None
        broken = bytearray(frame)
        broken[len(frame) * 3 // 4] ^= 0x10  # a block in a later launch
        _same(lz, oracle, bytes(broken))
    # short blocks in between full ones: compaction across launches
    blocks = []
    for k in range(60):
        d = data[k * 65536:(k + 1) * 65536] if k % 3 else data[k * 65536:k * 65536 + 1000 + k]
        c = oracle.block_compress_fast(d)
        blocks.append((c, d, False) if len(c) < len(d) else (d, d, True))
    frame = build_frame(flg=0x64, bd=0x40, blocks=blocks, xxh32=oracle.xxh32, add_content_checksum=True)
    assert _same(lz, oracle, frame) == b"".join(d for _, d, _ in blocks)
    d = data[:32768]
    dframe = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, dict_id=9), dictionary=d)
    assert lz.lz4FrameDecode(dframe, dictionaryResolver=lambda i: d) == data


def test_decode_bound_follows_the_payloads(lz, oracle):
    from lz4_amd import _native as N
    L = N.load()
    blocks = _tiny_blocks(oracle, 400, 2)
    frame = build_frame(flg=0x60, bd=0x70, blocks=blocks, xxh32=oracle.xxh32)
    bound = C.c_size_t(0)
    assert L.lz4mi_frame_decode_bound(frame, len(frame), C.byref(bound)) == 0
    want = sum(len(p) if raw else min(4 << 20, 255 * len(p) + 32) for p, _, raw in blocks)
    assert bound.value == want and bound.value >= sum(len(d) for _, d, _ in blocks)


def test_shard_write_checks_its_arguments(codec, oracle):
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd import _native as N
    from lz4_amd.api import _c_options
    data = corpus(16)
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB)
    o = _c_options(opts)
    src = torch.tensor(data, device="cuda")
    rec = codec.shard_compress(src, opts).to(torch.int64)
    offs = (torch.cumsum(rec, 0) - rec).contiguous()
    total = int(rec.sum())
    body = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    h = codec.ctx.handle

    def write(src_t, n, off_t, cap):
        return codec.L.lz4mi_frame_shard_write(h, C.c_void_p(src_t.data_ptr()), n, C.byref(o),
                                               C.c_void_p(off_t.data_ptr()), C.c_void_p(body.data_ptr()), cap)
    # a window that is too small: refused before anything is written, and the shard stays usable
    assert write(src, src.numel(), offs, total - 1) == N.DST_TOO_SMALL
    shifted = (offs + 100).contiguous()
    assert write(src, src.numel(), shifted, total + 64) == N.DST_TOO_SMALL
    codec.synchronize()
    assert int(body.sum()) == 0
    assert write(src, src.numel(), offs, total) == 0
    codec.synchronize()
    ref = oracle.frame_encode(data.tobytes(), oracle.frame_options(block_size_id=4))
    assert body[:total].cpu().numpy().tobytes() == ref[7:-4]
    # not step 1's source, or another call in between: refused
    other = src.clone()
    assert write(other, other.numel(), offs, total) == N.RANGE
    codec.shard_compress(src, opts)
    codec.xxh32(src[:100])
    assert write(src, src.numel(), offs, total) == N.RANGE
    assert b"shard_compress" in codec.L.lz4mi_last_backend_error(h)


def test_device_frame_encode_reports_content_size_mismatch_after_the_work(codec, oracle):
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd.api import _c_options
    data = corpus(3)
    src = torch.tensor(data, device="cuda")
    o = _c_options(Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, contentSize=5))
    cap = codec.L.lz4mi_frame_bound(src.numel(), C.byref(o))
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    n = C.c_uint64(0)
    st = codec.L.lz4mi_frame_encode_device(codec.ctx.handle, C.c_void_p(src.data_ptr()), src.numel(), C.byref(o), None,
                                           0, C.c_void_p(out.data_ptr()), cap, C.byref(n))
    codec.synchronize()
    assert st == 7  # LZ4MI_FORMAT_CONTENT_SIZE_ARG
    ref = oracle.frame_encode(data.tobytes(), oracle.frame_options(block_size_id=4, content_size=len(data)))
    got = out[: n.value].cpu().numpy().tobytes()
    assert got[:6] == ref[:6] and got[15:] == ref[15:]  # same frame but for the size field and its header checksum


def test_block_decompress_into_with_a_long_prefix(lz, oracle):
    rng = np.random.default_rng(5)
    prefix = rng.integers(0, 256, size=200000, dtype=np.uint8).tobytes()
    data = prefix[-70000:-4000] + ramp(5000)
    comp = oracle.block_compress_fast(data, dictionary=prefix)
    assert lz.lz4BlockDecompressInto(comp, prefix) == oracle.block_decompress_into(comp, prefix) == prefix + data
