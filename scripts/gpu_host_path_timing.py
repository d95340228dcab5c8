"""Host-buffer frame encode/decode through the C ABI with preallocated buffers: pageable vs registered memory."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N, api
from lz4_amd.options import Lz4FrameOptions, Lz4FrameBlockSize
from tools.corpus import corpus
L = N.load()
ctx = N.default_context(0)
nbytes = 256 << 20
data = corpus(nbytes // 65536, 65536)
with_ck = "ck" in sys.argv[1:]  # `ck`: contentChecksum on (hashed on a host thread beside the GPU work)
linked = "linked" in sys.argv[1:]  # `linked`: blockIndependence off (decode: all blocks at once, or LZ4MI_LINK_RESOLVE=0 the chain)
o = api._c_options(Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, contentChecksum=with_ck, blockIndependence=not linked))
print("contentChecksum:", with_ck, "linked:", linked, flush=True)
cap = L.lz4mi_frame_bound(nbytes, C.byref(o))
dst = np.empty(cap, dtype=np.uint8)
back = np.empty(nbytes + 64, dtype=np.uint8)
rt = torch.cuda.cudart()
def run(tag):
    out = C.c_size_t(0)
    for it in range(3):
        t0 = time.perf_counter()
        st = L.lz4mi_frame_encode(ctx.handle, C.c_char_p(data.ctypes.data), nbytes, C.byref(o), None, 0,
                                  C.c_void_p(dst.ctypes.data), cap, C.byref(out))
        t1 = time.perf_counter()
        if st != 0:
            L.lz4mi_last_backend_error.restype = C.c_char_p
            raise SystemExit(f"{tag}: lz4mi_frame_encode status {st}: {L.lz4mi_last_backend_error(ctx.handle)}")
    consumed, produced, need = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
    for it in range(3):
        t2 = time.perf_counter()
        st = L.lz4mi_frame_decode_next(ctx.handle, C.c_void_p(dst.ctypes.data), out.value, C.byref(consumed), None, 0, 0, 0,
                                       -1, C.c_void_p(back.ctypes.data), back.size, C.byref(produced), C.byref(need))
        t3 = time.perf_counter()
        assert st == 0 and produced.value == nbytes
    assert np.array_equal(back[:nbytes], data)
    print(f"{tag}: encode {1e3*(t1-t0):.1f} ms ({nbytes/1e9/(t1-t0):.2f} GB/s)  decode {1e3*(t3-t2):.1f} ms ({nbytes/1e9/(t3-t2):.2f} GB/s)", flush=True)
run("pageable host buffers")
t0 = time.perf_counter()
ok = all(int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) == 0 for a in (data, dst, back))
if ok:
    print(f"hipHostRegister of {(data.nbytes + dst.nbytes + back.nbytes) >> 20} MiB: {1e3*(time.perf_counter()-t0):.1f} ms")
    run("registered host buffers")
else:
    print("hipHostRegister refused on this machine: registered-buffer timing skipped")
