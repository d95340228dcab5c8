import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
src = torch.from_numpy(corpus(16384)).cuda()
slots = None
best = 1e9
for it in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); slots, sizes, st = codec.compress_blocks(src, 65536, out=slots); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"encode 16384 blocks: {best:.3f} ms, compressed {int(sizes.sum())}")
