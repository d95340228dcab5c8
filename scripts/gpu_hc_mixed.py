import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
for NB in (2048, 8192):
    src = torch.from_numpy(corpus(NB)).cuda()
    res = []
    for it in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        slots, sizes, st = codec.compress_blocks(src, 65536, hc_depth=64)
        e[1].record()
        torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]))
    print(f"mixed corpus NB {NB}: hc enc {min(res):.1f} ms ({NB * 65536 / 1e6 / min(res):.2f} GB/s) ratio {float(sizes.sum()) / (NB * 65536):.3f}", flush=True)
