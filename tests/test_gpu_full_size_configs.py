"""BASELINE configs[2] and configs[4] at their full size, every block against the oracle.

configs[2]: lz4FrameEncodeWithOptions over the 1 GiB corpus with 4 MiB blocks + block and content checksums
            (lib/src/frame/lz4_frame_encoder.dart:105-274), and decode of the frame the lz4 CLI of the image
            writes for the same bytes.
configs[4]: LZ4HC (maxSearchDepth 64) with a 32 KiB dictionary and dictId, 16 384 blocks of 64 KiB
            (lib/src/hc/lz4_hc_block_encoder.dart:49-197, test/frame/lz4_frame_encoder_dict_test.dart:8-124), and the
            dictionary-primed decode of those blocks (lib/src/frame/lz4_frame_decoder.dart:273-290).

The oracle runs on a thread pool, one C call per span of blocks.  Bit-exact (byte work).
"""
import ctypes as C
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tests.helpers import corpus, corpus_block, u32le

pytestmark = pytest.mark.gpu

GIB_BLOCKS = 16384  # 64 KiB blocks of the 1 GiB corpus


@pytest.fixture(scope="module")
def oracle():
    from oracle import lz4ref
    lz4ref.build()
    return lz4ref


@pytest.fixture(scope="module")
def codec():
    import torch
    from lz4_amd.device import DeviceCodec
    assert torch.cuda.is_available()
    return DeviceCodec(0)


def _oracle_blocks(oracle, data, block, level, param, dictionary=None, threads=16):
    """Every block of `data` through the oracle's encoder: (comp[nb, bound], sizes[nb])."""
    L = oracle.lib()
    nb = (data.size + block - 1) // block
    assert data.size == nb * block
    bound = oracle.compress_bound(block)
    comp = np.empty((nb, bound), dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.int64)
    d = np.frombuffer(dictionary, dtype=np.uint8).copy() if dictionary else None
    per = max(1, nb // (threads * 8))

    def work(lo):
        hi = min(nb, lo + per)
        st = L.lz4ref_blocks_compress(data.ctypes.data + lo * block, hi - lo, block, d.ctypes.data if d is not None else None,
                                      d.size if d is not None else 0, level, param, comp.ctypes.data + lo * bound, bound,
                                      sizes.ctypes.data + 8 * lo)
        assert st == 0, st

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(0, nb, per)))
    return comp, sizes


def _expected_records(oracle, data, block, comp, sizes, block_checksum):
    """The frame records the reference's dispatcher writes for these blocks (lz4_frame_encoder.dart:235-257)."""
    out = []
    nb = sizes.size
    for b in range(nb):
        c = int(sizes[b])
        if c < block:  # stored compressed iff strictly smaller (:235)
            word, payload = c, comp[b, :c].tobytes()
        else:
            word, payload = 0x80000000 | block, data[b * block:(b + 1) * block].tobytes()
        rec = u32le(word) + payload
        if block_checksum:
            rec += u32le(oracle.xxh32(payload))
        out.append(rec)
    return out


def test_config2_full_size_frame_every_record_equals_the_oracle(codec, oracle, tmp_path):
    import torch
    from lz4_amd import Lz4FrameOptions, lz4FrameDecode
    block = 4 << 20
    data = corpus(GIB_BLOCKS)
    src = torch.from_numpy(data).cuda()
    opts = Lz4FrameOptions(blockChecksum=True, contentChecksum=True)  # default block size: 4 MiB
    frame = codec.frame_encode(src, opts).cpu().numpy()
    del src
    torch.cuda.empty_cache()
    comp, sizes = _oracle_blocks(oracle, data, block, 0, 1)
    header = codec.frame_header(opts)
    assert frame[: len(header)].tobytes() == header
    off = len(header)
    for b, rec in enumerate(_expected_records(oracle, data, block, comp, sizes, True)):
        assert frame[off: off + len(rec)].tobytes() == rec, f"record {b} of {sizes.size}"
        off += len(rec)
    want_tail = u32le(0) + u32le(oracle.xxh32(data.tobytes()))  # end mark + xxh32 of all content (:263-267)
    assert frame[off:].tobytes() == want_tail
    # our own frame back through the host-buffer decoder (block + content checksums verified on the way)
    assert lz4FrameDecode(frame.tobytes()) == data.tobytes()
    # the frame the lz4 CLI writes for the same bytes
    cli = shutil.which("lz4") or "/opt/conda/bin/lz4"
    if not os.path.exists(cli):
        pytest.skip("lz4 CLI not present: our frame checked, CLI frame not")
    pin, pout = tmp_path / "in.bin", tmp_path / "out.lz4"
    data.tofile(pin)
    subprocess.run([cli, "-f", "-q", "-B7", "-BX", "--content-size", str(pin), str(pout)], check=True)
    os.unlink(pin)
    cli_frame = pout.read_bytes()
    os.unlink(pout)
    assert lz4FrameDecode(cli_frame) == data.tobytes()


def test_config4_full_size_hc_dictionary_every_block_equals_the_oracle(codec, oracle):
    import torch
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions, Lz4HcOptions
    block = 65536
    data = corpus(GIB_BLOCKS)
    dictionary = corpus_block(3)[:32768].tobytes()  # first 32 KiB of a text-like block
    src = torch.from_numpy(data).cuda()
    d_dict = torch.from_numpy(np.frombuffer(dictionary, dtype=np.uint8).copy()).cuda()
    slots, sizes, status = codec.compress_blocks(src, block, hc_depth=64, dictionary=d_dict)
    assert int(status.abs().sum()) == 0
    out, out_len, st = codec.decompress_blocks(slots, sizes, block, strict=False, dictionary=d_dict)
    assert int(st.abs().sum()) == 0 and bool((out_len == block).all()) and torch.equal(out.reshape(-1), src)
    del out
    comp, want = _oracle_blocks(oracle, data, block, 1, 64, dictionary)
    h_sizes = sizes.cpu().numpy().astype(np.int64)
    assert np.array_equal(h_sizes, want), np.nonzero(h_sizes != want)[0][:10]
    h_slots = slots.cpu().numpy()
    del slots
    bad = [b for b in range(GIB_BLOCKS) if not np.array_equal(h_slots[b, : want[b]], comp[b, : want[b]])]
    assert not bad, bad[:10]
    del h_slots
    # the same blocks as one frame with dictId (lz4_frame_encoder.dart:135-138,167-169): header, every record, end mark
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, compression=Lz4FrameCompression.hc,
                           hcOptions=Lz4HcOptions(maxSearchDepth=64), dictId=0x12345678)
    frame = codec.frame_encode(src, opts, dictionary=d_dict).cpu().numpy()
    header = codec.frame_header(opts)
    expect = header + b"".join(_expected_records(oracle, data, block, comp, want, False)) + u32le(0)
    assert frame.size == len(expect)
    assert frame.tobytes() == expect
