"""lz4mi_decompress_batch_static (the entry point that allocates nothing) against
lz4mi_decompress_batch, through the C ABI: same bytes, lengths and status words, on two calls in
a row, the second with a smaller input; then a launch small enough to need no queue, and the
size limit.  Bit-exact (byte work).
"""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import corpus

pytestmark = pytest.mark.gpu

BLOCK = 8192  # small blocks: a few thousand of them make a queued launch (order list in use)


def _batch(N, comp, sizes, out, out_len, status, strict):
    b = N.Batch()
    b.d_src = comp.data_ptr()
    b.src_stride = comp.stride(0)
    b.d_src_len = sizes.data_ptr()
    b.d_dst = out.data_ptr()
    b.dst_stride = BLOCK
    b.dst_cap_uniform = BLOCK
    b.nblocks = sizes.numel()
    b.d_out_len = out_len.data_ptr()
    b.d_status = status.data_ptr()
    return b


def _decode_both(codec, N, torch, comp, sizes, strict=True):
    nb = sizes.numel()
    res = []
    for fn in (codec.L.lz4mi_decompress_batch, codec.L.lz4mi_decompress_batch_static):
        out = torch.zeros((nb, BLOCK), dtype=torch.uint8, device="cuda")
        out_len = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
        status = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
        b = _batch(N, comp, sizes, out, out_len, status, strict)
        codec.stream.wait_stream(torch.cuda.current_stream())
        st = fn(codec.ctx.handle, C.byref(b), int(strict))
        assert st == 0, codec.L.lz4mi_last_backend_error(codec.ctx.handle)
        codec.synchronize()
        res.append((out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()))
    return res


def test_static_decode_equals_allocating_decode():
    import torch
    from lz4_amd import _native as N
    from lz4_amd.device import DeviceCodec
    assert torch.cuda.is_available()
    codec = DeviceCodec(0)
    for nblocks, seed in ((9000, 0), (5000, 1), (100, 2)):  # queued, queued and smaller, direct
        raw = corpus(nblocks, block=BLOCK, seed=seed)
        src = torch.from_numpy(raw.copy()).cuda()
        comp, sizes, status = codec.compress_blocks(src, BLOCK)
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0
        (o0, l0, s0), (o1, l1, s1) = _decode_both(codec, N, torch, comp, sizes)
        assert np.array_equal(s0, s1) and not s1.any()
        assert np.array_equal(l0, l1) and (l1 == BLOCK).all()
        assert np.array_equal(o0, o1)
        assert np.array_equal(o1.reshape(-1), raw)
        # damaged input: same status words and lengths from both entry points
        bad = comp.clone()
        bad[::3, 5] ^= 0x5A
        (o0, l0, s0), (o1, l1, s1) = _decode_both(codec, N, torch, bad, sizes)
        assert np.array_equal(s0, s1) and np.array_equal(l0, l1)
        ok = s0 == 0
        assert np.array_equal(o0[ok], o1[ok])


def test_static_decode_rejects_more_blocks_than_its_scratch_holds():
    import torch
    from lz4_amd import _native as N
    from lz4_amd.device import DeviceCodec
    codec = DeviceCodec(0)
    b = N.Batch()
    b.nblocks = (1 << 20) + 1
    st = codec.L.lz4mi_decompress_batch_static(codec.ctx.handle, C.byref(b), 1)
    assert st == N.UNSUPPORTED_HERE
    assert b"static scratch" in codec.L.lz4mi_last_backend_error(codec.ctx.handle)
    b.nblocks = 0
    assert codec.L.lz4mi_decompress_batch_static(codec.ctx.handle, C.byref(b), 1) == 0
