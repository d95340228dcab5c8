"""One-shot decode of a linked-block frame (host-buffer call): the chain kernel's rate."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from lz4_amd import Lz4FrameOptions, Lz4FrameBlockSize, api
from tools.corpus import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
data = corpus(nb, 65536).tobytes()
for bs, name in ((Lz4FrameBlockSize.k64KB, '64KiB'), (Lz4FrameBlockSize.k4MB, '4MiB')):
    f = api.lz4FrameEncodeWithOptions(data, options=Lz4FrameOptions(blockSize=bs, blockIndependence=False))
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter(); out = api.lz4FrameDecode(f); best = min(best, time.perf_counter() - t0)
    assert out == data
    print(f"linked frame, {name} blocks, {len(data) >> 20} MiB: decode {best * 1e3:.1f} ms ({len(data) / 1e9 / best:.2f} GB/s, {best * 1e6 / nb:.0f} us per 64 KiB)", flush=True)
if len(sys.argv) > 2:  # diagnostic build: phase stamps of the two wavefronts over the last decode
    import ctypes as C
    from lz4_amd import _native as N
    L = N.load()
    buf = (C.c_ulonglong * 32)()
    L.lz4mi_debug_dec_phases(buf, 1)
    f = api.lz4FrameEncodeWithOptions(data, options=Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False))
    api.lz4FrameDecode(f)
    L.lz4mi_debug_dec_phases(buf, 1)
    v = list(buf)
    for role in (0, 1):
        print('role', role, {i: round(v[16 * role + i] / nb / 1000, 1) for i in range(16) if v[16 * role + i]}, 'kcycles per block')
