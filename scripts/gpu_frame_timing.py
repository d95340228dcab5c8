"""Frame-level timing: device-resident frame encode (1 GiB, 64 KiB blocks) and host-buffer encode/decode (256 MiB)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from lz4_amd import Lz4FrameOptions, Lz4FrameBlockSize, api
from tools.corpus import corpus
codec = DeviceCodec(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
data = corpus(nb, 65536)
src = torch.from_numpy(data).cuda()
for bs, name in ((Lz4FrameBlockSize.k64KB, '64KiB'), (Lz4FrameBlockSize.k4MB, '4MiB')):
    for bc, cc in ((False, False), (True, False), (True, True)):
        opts = Lz4FrameOptions(blockSize=bs, blockChecksum=bc, contentChecksum=cc)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            frame = codec.frame_encode(src, opts)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"device frame_encode {name} blockChecksum={bc} contentChecksum={cc}: {best*1e3:.2f} ms ({data.size/1e9/best:.1f} GB/s), {frame.numel()} bytes", flush=True)
host = data[:256 << 20].tobytes()
for bs, name in ((Lz4FrameBlockSize.k64KB, '64KiB'), (Lz4FrameBlockSize.k4MB, '4MiB')):
    opts = Lz4FrameOptions(blockSize=bs)
    t0 = time.perf_counter(); f = api.lz4FrameEncodeWithOptions(host, options=opts); t1 = time.perf_counter()
    t0 = time.perf_counter(); f = api.lz4FrameEncodeWithOptions(host, options=opts); t1 = time.perf_counter()
    out = api.lz4FrameDecode(f); t2 = time.perf_counter()
    assert out == host
    print(f"host-buffer {name}: encode {1e3*(t1-t0):.1f} ms ({len(host)/1e9/(t1-t0):.2f} GB/s), decode {1e3*(t2-t1):.1f} ms ({len(host)/1e9/(t2-t1):.2f} GB/s)", flush=True)
