"""Chunk-stream frame encode/decode through the C ABI (lz4mi_frame_encoder_*, lz4mi_frame_decoder_*):
256 MiB of the corpus fed in fixed-size chunks from pageable host memory, preallocated outputs."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
from lz4_amd import _native as N, api
from lz4_amd.options import Lz4FrameOptions, Lz4FrameBlockSize
from tools.corpus import corpus
L = N.load()
ctx = N.default_context(0)
nbytes = 256 << 20
data = corpus(nbytes // 65536, 65536)
vp = C.c_void_p
L.lz4mi_frame_encoder_update.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
L.lz4mi_frame_decoder_feed.argtypes = [vp, vp, C.c_size_t]
for linked in (False, True):
    o = api._c_options(Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=not linked))
    for chunk in (1 << 20, 16 << 20, 64 << 20):
        frame = np.empty(L.lz4mi_frame_bound(nbytes, C.byref(o)) + 64, dtype=np.uint8)
        out = C.c_size_t(0)
        best_e = best_d = 1e9
        for it in range(3):
            h = vp()
            assert L.lz4mi_frame_encoder_create(ctx.handle, C.byref(o), None, 0, C.byref(h)) == 0
            t0 = time.perf_counter()
            L.lz4mi_frame_header(C.byref(o), vp(frame.ctypes.data), 32, C.byref(out))
            w = out.value
            for pos in range(0, nbytes, chunk):
                n = min(chunk, nbytes - pos)
                st = L.lz4mi_frame_encoder_update(h, vp(data.ctypes.data + pos), n, vp(frame.ctypes.data + w),
                                                  frame.size - w, C.byref(out))
                assert st == 0, st
                w += out.value
            st = L.lz4mi_frame_encoder_finish(h, vp(frame.ctypes.data + w), frame.size - w, C.byref(out))
            assert st == 0
            w += out.value
            best_e = min(best_e, time.perf_counter() - t0)
            L.lz4mi_frame_encoder_destroy(h)
        back = np.empty(nbytes + 64, dtype=np.uint8)
        need = C.c_uint32(0)
        for it in range(3):
            h = vp()
            assert L.lz4mi_frame_decoder_create(ctx.handle, -1, C.byref(h)) == 0
            t0 = time.perf_counter()
            r = 0
            for pos in range(0, w, chunk):
                n = min(chunk, w - pos)
                assert L.lz4mi_frame_decoder_feed(h, vp(frame.ctypes.data + pos), n) == 0
                while True:
                    st = L.lz4mi_frame_decoder_next(h, vp(back.ctypes.data + r), back.size - r, C.byref(out), C.byref(need))
                    if st == 0:
                        r += out.value
                        continue
                    assert st == N.STREAM_NEED_INPUT, st
                    break
            assert L.lz4mi_frame_decoder_finish(h) == 0
            best_d = min(best_d, time.perf_counter() - t0)
            L.lz4mi_frame_decoder_destroy(h)
            assert r == nbytes
        assert np.array_equal(back[:nbytes], data)
        print(f"linked={int(linked)} chunk={chunk >> 20:3d} MiB: stream encode {nbytes / 1e9 / best_e:6.2f} GB/s  "
              f"stream decode {nbytes / 1e9 / best_d:6.2f} GB/s  (frame {w} bytes)", flush=True)
