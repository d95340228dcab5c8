"""Device-resident batch API: torch tensors in HBM in, torch tensors in HBM out.

torch is used only for device memory, streams and (by callers) torch.distributed;
all arithmetic happens in the HIP kernels behind the C ABI (include/lz4mi.h).
The codec owns one torch stream; its
work is fenced against torch's current stream on entry and exit of every call.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .api import _c_options
from .options import Lz4FrameOptions


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceCodec:
    def __init__(self, device: int | None = None, priority: int = 0):
        if not torch.cuda.is_available():
            raise N.Lz4BackendError("no GPU visible to torch: the device codec needs an MI355X")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.torch_device = torch.device("cuda", self.device)
        # the codec's own stream; calls are fenced against torch's current stream on entry and exit
        # (priority: torch's stream priority, -1 = high; a caller that overlaps launches of two codecs gives the
        # one whose launch should get the wavefront slots first the high one)
        self.stream = torch.cuda.Stream(self.torch_device, priority=priority)
        self.ctx = N.Context(self.device, self.stream.cuda_stream)
        self.L = N.load()
        self.fence = True  # set False to manage stream ordering yourself (benchmarks)

    def _enter(self):
        if self.fence:
            self.stream.wait_stream(torch.cuda.current_stream(self.torch_device))

    def _exit(self):
        if self.fence:
            torch.cuda.current_stream(self.torch_device).wait_stream(self.stream)

    # ------------------------------------------------------------------ blocks
    def compress_bound(self, n: int) -> int:
        return self.L.lz4mi_compress_bound(n)

    def slot_stride(self, block_size: int) -> int:
        return (self.compress_bound(block_size) + 63) // 64 * 64

    def compress_blocks(self, src: torch.Tensor, block_size: int, *, acceleration: int = 1,
                        dictionary: torch.Tensor | None = None, hc_depth: int | None = None,
                        out: torch.Tensor | None = None):
        """Per-block lz4BlockCompressToWriter over consecutive `block_size` chunks of `src`.
        Returns (slots [nblocks, stride] uint8, sizes int32 [nblocks], status int32 [nblocks])."""
        assert src.dtype == torch.uint8 and src.is_cuda and src.is_contiguous()
        n = src.numel()
        nb = (n + block_size - 1) // block_size
        stride = self.slot_stride(block_size)
        if out is None:
            out = torch.empty((max(nb, 1), stride), dtype=torch.uint8, device=src.device)
        sizes = torch.empty(max(nb, 1), dtype=torch.int32, device=src.device)
        status = torch.empty(max(nb, 1), dtype=torch.int32, device=src.device)
        b = N.Batch()
        b.d_src = src.data_ptr()
        b.src_stride = block_size
        b.src_len_uniform = block_size
        b.src_total = n
        b.d_dst = out.data_ptr()
        b.dst_stride = stride
        b.dst_cap_uniform = stride
        if dictionary is not None and dictionary.numel():
            b.d_dict = dictionary.data_ptr()
            b.dict_len = dictionary.numel()
        b.nblocks = nb
        b.d_out_len = sizes.data_ptr()
        b.d_status = status.data_ptr()
        if hc_depth is None:
            self._enter()
            st = self.L.lz4mi_compress_fast_batch(self.ctx.handle, C.byref(b), acceleration)
            self._exit()
        else:
            self._enter()
            st = self.L.lz4mi_compress_hc_batch(self.ctx.handle, C.byref(b), hc_depth)
            self._exit()
        if st:
            N.raise_status(st, self.ctx.handle, arg="acceleration" if hc_depth is None else "maxSearchDepth")
        return out[:nb], sizes[:nb], status[:nb]

    def decompress_blocks(self, comp: torch.Tensor, sizes: torch.Tensor, block_size: int, *,
                          comp_offsets: torch.Tensor | None = None, comp_stride: int | None = None,
                          expected: torch.Tensor | None = None, strict: bool = True,
                          dictionary: torch.Tensor | None = None, out: torch.Tensor | None = None):
        """Per-block lz4BlockDecompress (strict) / lz4BlockDecompressInto (lenient).
        Block b reads sizes[b] bytes at comp_offsets[b] (or b * comp_stride) and writes slot b of
        `out` ([nblocks, block_size]).  expected: per-block decompressedSize (strict) or cap."""
        nb = sizes.numel()
        if out is None:
            out = torch.empty((max(nb, 1), block_size), dtype=torch.uint8, device=comp.device)
        out_len = torch.empty(max(nb, 1), dtype=torch.int32, device=comp.device)
        status = torch.empty(max(nb, 1), dtype=torch.int32, device=comp.device)
        b = N.Batch()
        b.d_src = comp.data_ptr()
        if comp_offsets is not None:
            assert comp_offsets.dtype == torch.int64
            b.d_src_off = comp_offsets.data_ptr()
        else:
            b.src_stride = comp_stride if comp_stride is not None else comp.stride(0)
        b.d_src_len = sizes.data_ptr()
        b.d_dst = out.data_ptr()
        b.dst_stride = block_size
        if expected is not None:
            b.d_dst_cap = expected.data_ptr()
        else:
            b.dst_cap_uniform = block_size
        if dictionary is not None and dictionary.numel():
            b.d_dict = dictionary.data_ptr()
            b.dict_len = dictionary.numel()
        b.nblocks = nb
        b.d_out_len = out_len.data_ptr()
        b.d_status = status.data_ptr()
        self._enter()
        st = self.L.lz4mi_decompress_batch(self.ctx.handle, C.byref(b), int(strict))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return out[:nb], out_len[:nb], status[:nb]

    def xxh32_blocks(self, data: torch.Tensor, lengths: torch.Tensor, *, offsets: torch.Tensor | None = None,
                     stride: int | None = None) -> torch.Tensor:
        nb = lengths.numel()
        h = torch.empty(max(nb, 1), dtype=torch.int32, device=data.device)
        b = N.Batch()
        b.d_src = data.data_ptr()
        if offsets is not None:
            b.d_src_off = offsets.data_ptr()
        else:
            b.src_stride = stride if stride is not None else data.stride(0)
        b.d_src_len = lengths.data_ptr()
        b.nblocks = nb
        self._enter()
        st = self.L.lz4mi_xxh32_batch(self.ctx.handle, C.byref(b), _ptr(h))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return h[:nb]

    def xxh32(self, data: torch.Tensor, seed: int = 0) -> torch.Tensor:
        h = torch.empty(1, dtype=torch.int32, device=data.device)
        self._enter()
        st = self.L.lz4mi_xxh32_device(self.ctx.handle, _ptr(data), data.numel(), seed & 0xFFFFFFFF, _ptr(h))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return h

    # ------------------------------------------------------------------ frames
    def frame_bound(self, n: int, options: Lz4FrameOptions) -> int:
        o = _c_options(options)
        return self.L.lz4mi_frame_bound(n, C.byref(o))

    def frame_encode(self, src: torch.Tensor, options: Lz4FrameOptions, dictionary: torch.Tensor | None = None,
                     out: torch.Tensor | None = None) -> torch.Tensor:
        """lz4FrameEncodeWithOptions on device-resident bytes; returns the frame (a view of `out`)."""
        o = _c_options(options)
        n = src.numel()
        cap = self.L.lz4mi_frame_bound(n, C.byref(o))
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=src.device)
        total = C.c_uint64(0)
        self._enter()
        st = self.L.lz4mi_frame_encode_device(self.ctx.handle, _ptr(src), n, C.byref(o), _ptr(dictionary),
                                              dictionary.numel() if dictionary is not None else 0, _ptr(out),
                                              out.numel(), C.byref(total))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        if options.contentSize is not None and options.contentSize != n:
            N.raise_status(7, self.ctx.handle)
        return out[: total.value]

    def shard_compress(self, src: torch.Tensor, options: Lz4FrameOptions,
                       dictionary: torch.Tensor | None = None) -> torch.Tensor:
        """Step 1 of multi-GPU frame assembly: compress this rank's blocks, return the size of
        each block's frame record (int32 [nblocks], device)."""
        o = _c_options(options)
        n = src.numel()
        nb_max = (n + options.blockSize.maxBytes - 1) // options.blockSize.maxBytes
        rec = torch.empty(max(nb_max, 1), dtype=torch.int32, device=src.device)
        nb = C.c_uint32(0)
        self._enter()
        st = self.L.lz4mi_frame_shard_compress(self.ctx.handle, _ptr(src), n, C.byref(o), _ptr(dictionary),
                                               dictionary.numel() if dictionary is not None else 0, _ptr(rec),
                                               C.byref(nb))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return rec[: nb.value]

    def shard_write(self, src: torch.Tensor, options: Lz4FrameOptions, record_offsets: torch.Tensor,
                    out: torch.Tensor):
        """Step 2: write this rank's records at `record_offsets` (int64, relative to `out`)."""
        o = _c_options(options)
        assert record_offsets.dtype == torch.int64
        self._enter()
        st = self.L.lz4mi_frame_shard_write(self.ctx.handle, _ptr(src), src.numel(), C.byref(o),
                                            _ptr(record_offsets), _ptr(out), out.numel())
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)

    def shard_decode(self, records: torch.Tensor, record_offsets: torch.Tensor, options: Lz4FrameOptions,
                     dictionary: torch.Tensor | None = None, out: torch.Tensor | None = None):
        """Decode side of the sharded frame: this rank's block records (`records`, uint8, device) at
        `record_offsets` (int64, relative to `records`).  Returns (slots [nblocks, blockMaxSize],
        out_len int32 [nblocks], status int32 [nblocks]); block b's bytes are slots[b, :out_len[b]]."""
        o = _c_options(options)
        assert record_offsets.dtype == torch.int64 and records.dtype == torch.uint8
        nb = record_offsets.numel()
        bmax = options.blockSize.maxBytes
        if out is None:
            out = torch.empty((max(nb, 1), bmax), dtype=torch.uint8, device=records.device)
        out_len = torch.empty(max(nb, 1), dtype=torch.int32, device=records.device)
        status = torch.empty(max(nb, 1), dtype=torch.int32, device=records.device)
        self._enter()
        st = self.L.lz4mi_frame_shard_decode(self.ctx.handle, _ptr(records), records.numel(), _ptr(record_offsets), nb,
                                             C.byref(o), _ptr(dictionary),
                                             dictionary.numel() if dictionary is not None else 0, _ptr(out),
                                             _ptr(out_len), _ptr(status))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return out[:nb], out_len[:nb], status[:nb]

    def gather_blocks(self, slots: torch.Tensor, lengths: torch.Tensor, dst_offsets: torch.Tensor,
                      out: torch.Tensor):
        """Packs slots[b, :lengths[b]] to out[dst_offsets[b]:] (the decoder appends whatever each block yields)."""
        assert dst_offsets.dtype == torch.int64 and lengths.dtype == torch.int32
        self._enter()
        st = self.L.lz4mi_blocks_gather(self.ctx.handle, _ptr(slots), slots.stride(0), _ptr(lengths), _ptr(dst_offsets),
                                        lengths.numel(), _ptr(out), out.numel())
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)

    def block_conformance(self, comp: torch.Tensor, sizes: torch.Tensor, comp_stride: int | None = None) -> torch.Tensor:
        """int32 [nblocks]: 1 where a compressed block breaks the LZ4 end-of-block rules (the reference's
        encoders end blocks inside matches, lib/src/block/lz4_block_encoder.dart:95,118)."""
        nb = sizes.numel()
        flags = torch.empty(max(nb, 1), dtype=torch.int32, device=comp.device)
        b = N.Batch()
        b.d_src = comp.data_ptr()
        b.src_stride = comp_stride if comp_stride is not None else comp.stride(0)
        b.d_src_len = sizes.data_ptr()
        b.nblocks = nb
        self._enter()
        st = self.L.lz4mi_blocks_conformance(self.ctx.handle, C.byref(b), _ptr(flags))
        self._exit()
        if st:
            N.raise_status(st, self.ctx.handle)
        return flags[:nb]

    def frame_header(self, options: Lz4FrameOptions) -> bytes:
        o = _c_options(options)
        buf = C.create_string_buffer(32)
        n = C.c_size_t(0)
        st = self.L.lz4mi_frame_header(C.byref(o), buf, 32, C.byref(n))
        if st:
            N.raise_status(st, self.ctx.handle)
        return buf.raw[: n.value]

    def synchronize(self):
        self.ctx.synchronize()
