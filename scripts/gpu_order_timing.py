"""Encode/decode time of the bench corpus with its blocks in different orders (start-skew check)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
data = corpus(NB).reshape(NB, 65536)
cls = np.arange(NB) % 8
heavy = (cls == 5) | (cls == 6)
orders = {'natural': np.arange(NB), 'heavy first': np.concatenate([np.where(heavy)[0], np.where(~heavy)[0]]),
          'heavy last': np.concatenate([np.where(~heavy)[0], np.where(heavy)[0]])}
for name, perm in orders.items():
    src = torch.from_numpy(np.ascontiguousarray(data[perm]).reshape(-1)).cuda()
    slots = out = None
    res = []
    for it in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        slots, sizes, st = codec.compress_blocks(src, 65536, out=slots)
        e[1].record()
        out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True, out=out)
        e[2].record()
        torch.cuda.synchronize()
        res.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    assert torch.equal(out.reshape(-1)[:src.numel()], src.reshape(-1)), (out.shape, src.shape)
    print(f"{name:12s}: enc {min(r[0] for r in res):.3f} ms  dec {min(r[1] for r in res):.3f} ms", flush=True)
