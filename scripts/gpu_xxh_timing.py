"""Single-stream xxh32 rate on device memory (content checksum), checked against python-xxhash."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, xxhash
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
data = corpus(4096)
src = torch.from_numpy(data).cuda()
for n in (1 << 20, 256 << 20):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); h = codec.xxh32(src[:n]); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    ok = (int(h[0]) & 0xFFFFFFFF) == xxhash.xxh32(data[:n].tobytes()).intdigest()
    print(f"{n >> 20} MiB: {best:.2f} ms, {n / best / 1e6:.2f} GB/s, matches python-xxhash: {ok}")
