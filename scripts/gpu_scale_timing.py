"""Encode/decode time of one corpus class at several batch sizes (latency- vs throughput-bound check)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
cls = int(sys.argv[1])
for NB in [int(x) for x in sys.argv[2:]]:
    blocks = [corpus_block(cls + 8 * k) for k in range(64)]
    data = np.concatenate([blocks[k % 64] for k in range(NB)])
    src = torch.from_numpy(data).cuda()
    slots = out = None
    res = []
    for it in range(4):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        slots, sizes, st = codec.compress_blocks(src, 65536, out=slots)
        e[1].record()
        out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True, out=out)
        e[2].record()
        torch.cuda.synchronize()
        res.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    enc, dec = min(r[0] for r in res), min(r[1] for r in res)
    print(f"class {cls} NB {NB}: enc {enc:.3f} ms  dec {dec:.3f} ms", flush=True)
