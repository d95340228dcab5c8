"""Sharded decode of one frame (lib/src/frame/lz4_frame_decoder.dart:238-304 split by block range),
two ranks played one after the other on one GPU, and the end-of-block conformance count.
Bit-exact against the oracle; per-block status words against what the reference's loop would throw."""
import numpy as np
import pytest

from tests.helpers import build_frame, corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    import torch
    from lz4_amd.device import DeviceCodec
    assert torch.cuda.is_available()
    return DeviceCodec(0)


def _short_block_frame(oracle, data, flg):
    blocks, a, k = [], 0, 0
    while a < len(data):
        chunk = data[a:a + (65536, 4464, 65536, 1, 65536, 33391)[k % 6]]
        a += len(chunk)
        k += 1
        c = oracle.block_compress_fast(chunk)
        blocks.append((c, chunk, False) if len(c) < len(chunk) else (chunk, chunk, True))
    return build_frame(flg=flg, bd=0x40, blocks=blocks, xxh32=oracle.xxh32)


@pytest.mark.parametrize("kind", ["full", "short", "short_no_checksum"])
def test_two_shards_decode_to_the_frame_content(codec, oracle, kind):
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd.sharding import frame_block_table, shard_range
    data = corpus(40).tobytes()
    if kind == "full":
        frame = oracle.frame_encode(data, oracle.frame_options(block_size_id=4, block_checksum=True))
    else:
        frame = _short_block_frame(oracle, data, 0x70 if kind == "short" else 0x60)
    info, offs, end = frame_block_table(frame)
    assert end == len(frame)
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=info["block_checksum"])
    world = 2
    parts = []
    for rank in range(world):
        lo, hi = shard_range(len(offs), rank, world)
        first = int(offs[lo])
        last = int(offs[hi]) if hi < len(offs) else info["body_end"]
        window = torch.tensor(np.frombuffer(frame[first:last], dtype=np.uint8), device="cuda")
        rel = torch.tensor(offs[lo:hi] - first, dtype=torch.int64, device="cuda")
        slots, out_len, status = codec.shard_decode(window, rel, opts)
        assert int(status.abs().sum()) == 0
        parts.append((slots, out_len))
    # the size exchange, here a concatenation: every rank's decoded sizes -> output offsets
    all_len = torch.cat([p[1] for p in parts]).to(torch.int64)
    starts = torch.cumsum(all_len, 0) - all_len
    total = int(all_len.sum())
    assert total == len(data)
    out = torch.zeros(total, dtype=torch.uint8, device="cuda")
    k = 0
    for slots, out_len in parts:
        n = out_len.numel()
        codec.gather_blocks(slots, out_len, starts[k:k + n].contiguous(), out)
        k += n
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == data == oracle.frame_decode(frame)
    # a window that cannot hold the blocks is refused before anything is written
    small = torch.zeros(total - 1, dtype=torch.uint8, device="cuda")
    from lz4_amd import Lz4BackendError
    with pytest.raises(Lz4BackendError):
        codec.gather_blocks(parts[-1][0], parts[-1][1], starts[-parts[-1][1].numel():].contiguous(), small)
    assert int(small.sum()) == 0


def test_shard_decode_status_words(codec, oracle):
    """Per record: what lz4_frame_decoder.dart:251-296 throws for it, in its order."""
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd.sharding import frame_block_table
    data = corpus(8).tobytes()
    frame = bytearray(oracle.frame_encode(data, oracle.frame_options(block_size_id=4, block_checksum=True)))
    info, offs, _ = frame_block_table(bytes(frame))
    sizes = [int.from_bytes(frame[o:o + 4], "little") & 0x7FFFFFFF for o in offs]
    # block 1: payload damaged (checksum); block 2: checksum word itself damaged; block 5/6 (LZ data): a payload
    # byte damaged AND the checksum recomputed, so the decoder's own status shows
    frame[offs[1] + 4 + 10] ^= 1
    frame[offs[2] + 4 + sizes[2] + 1] ^= 0x80
    expect = {1: 16, 2: 16}
    for b in (5, 6):
        o = int(offs[b]) + 4
        payload = bytearray(frame[o:o + sizes[b]])
        payload[3] = 0xFF
        payload[4] = 0xFF  # a huge literal length
        frame[o:o + sizes[b]] = payload
        frame[o + sizes[b]:o + sizes[b] + 4] = oracle.xxh32(bytes(payload)).to_bytes(4, "little")
        try:
            oracle.block_decompress_into(bytes(payload), b"", 65536)
            expect[b] = 0
        except oracle.Lz4RefError as e:
            expect[b] = e.status
    window = torch.tensor(np.frombuffer(bytes(frame[offs[0]:info["body_end"]]), dtype=np.uint8), device="cuda")
    rel = torch.tensor(offs - offs[0], dtype=torch.int64, device="cuda")
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockChecksum=True)
    slots, out_len, status = codec.shard_decode(window, rel, opts)
    st = status.cpu().numpy()
    for b in range(len(offs)):
        assert st[b] == expect.get(b, 0), (b, st[b])
    ok = [b for b in range(len(offs)) if b not in expect]
    for b in ok:
        assert slots[b, : int(out_len[b])].cpu().numpy().tobytes() == data[b * 65536:(b + 1) * 65536]
    # a size word over blockMaxSize, and a window cut inside the last record
    bad = bytearray(frame)
    bad[offs[3]:offs[3] + 4] = (70000).to_bytes(4, "little")
    window = torch.tensor(np.frombuffer(bytes(bad[offs[0]:info["body_end"] - 3]), dtype=np.uint8), device="cuda")
    _, _, status = codec.shard_decode(window, rel, opts)
    st = status.cpu().numpy()
    assert st[3] == 15 and st[len(offs) - 1] == 1


def test_conformance_flags_equal_the_oracle(codec, oracle):
    import torch
    nb = 256
    data = corpus(nb)
    src = torch.tensor(data, device="cuda")
    slots, sizes, status = codec.compress_blocks(src, 65536)
    flags = codec.block_conformance(slots, sizes).cpu().numpy()
    h = sizes.cpu().numpy()
    comp = slots.cpu().numpy()
    want = np.array([0 if oracle.block_conformant(comp[b, : h[b]].tobytes()) else 1 for b in range(nb)])
    assert np.array_equal(flags, want)
    assert flags.sum() > 0  # the period-32 ramp and the byte run end inside a match
    # blocks encoded under the end-of-block rules have none
    for b in (1, 2, 5, 6, 13):
        c = oracle.block_compress_ex(data[b * 65536:(b + 1) * 65536].tobytes(), conformant=True)
        t = torch.tensor(np.frombuffer(c, dtype=np.uint8), device="cuda").reshape(1, -1)
        n = torch.tensor([len(c)], dtype=torch.int32, device="cuda")
        assert int(codec.block_conformance(t, n)[0]) == 0
