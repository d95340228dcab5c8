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


def test_static_decode_first_call_on_a_fresh_context_allocates_nothing():
    """The static entry point as the FIRST decode call of a context (its order buffer has never been grown), with
    the largest batch of this file: neither the context's buffers nor the device's free memory change across the
    call, and the result equals the allocating entry point's, which is called afterwards and does grow them."""
    import torch
    from lz4_amd import _native as N
    from lz4_amd.device import DeviceCodec
    nblocks = 9000
    raw = corpus(nblocks, block=BLOCK, seed=7)
    src = torch.from_numpy(raw.copy()).cuda()
    comp, sizes, status = DeviceCodec(0).compress_blocks(src, BLOCK)  # another context does the compressing
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    codec = DeviceCodec(0)  # fresh: has decoded nothing
    L = codec.L
    outs = [(torch.zeros((nblocks, BLOCK), dtype=torch.uint8, device="cuda"),
             torch.full((nblocks,), -1, dtype=torch.int32, device="cuda"),
             torch.full((nblocks,), -1, dtype=torch.int32, device="cuda")) for _ in range(2)]
    torch.cuda.synchronize()
    held0, free0 = L.lz4mi_ctx_device_bytes(codec.ctx.handle), torch.cuda.mem_get_info()[0]
    b = _batch(N, comp, sizes, *outs[0], True)
    assert L.lz4mi_decompress_batch_static(codec.ctx.handle, C.byref(b), 1) == 0
    codec.synchronize()
    assert L.lz4mi_ctx_device_bytes(codec.ctx.handle) == held0, "the static entry point grew a context buffer"
    assert torch.cuda.mem_get_info()[0] == free0, "device memory was allocated during the static call"
    b = _batch(N, comp, sizes, *outs[1], True)
    assert L.lz4mi_decompress_batch(codec.ctx.handle, C.byref(b), 1) == 0
    codec.synchronize()
    assert L.lz4mi_ctx_device_bytes(codec.ctx.handle) >= held0 + 2 * 4 * nblocks  # its order list: two words per block
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    assert np.array_equal(outs[0][0].cpu().numpy().reshape(-1), raw)


def test_static_decode_from_two_contexts_takes_turns():
    """Two contexts with a stream each share the GPU's scratch array: calls issued back to back without a
    synchronisation in between are ordered by the library (event of the last user), so both come out right."""
    import torch
    from lz4_amd import _native as N
    from lz4_amd.device import DeviceCodec
    a, b_ = DeviceCodec(0), DeviceCodec(0)
    jobs = []
    for k, (codec, nblocks) in enumerate(((a, 6000), (b_, 7000), (a, 3000), (b_, 6500))):
        raw = corpus(nblocks, block=BLOCK, seed=20 + k)
        src = torch.from_numpy(raw.copy()).cuda()
        comp, sizes, status = codec.compress_blocks(src, BLOCK)
        out = torch.zeros((nblocks, BLOCK), dtype=torch.uint8, device="cuda")
        out_len = torch.full((nblocks,), -1, dtype=torch.int32, device="cuda")
        st = torch.full((nblocks,), -1, dtype=torch.int32, device="cuda")
        jobs.append((codec, raw, comp, sizes, out, out_len, st))
    torch.cuda.synchronize()
    for codec, raw, comp, sizes, out, out_len, st in jobs:  # all four enqueued before anything is waited for
        bt = _batch(N, comp, sizes, out, out_len, st, True)
        assert codec.L.lz4mi_decompress_batch_static(codec.ctx.handle, C.byref(bt), 1) == 0
    a.synchronize()
    b_.synchronize()
    for codec, raw, comp, sizes, out, out_len, st in jobs:
        assert not st.cpu().numpy().any()
        assert (out_len.cpu().numpy() == BLOCK).all()
        assert np.array_equal(out.cpu().numpy().reshape(-1), raw)


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
