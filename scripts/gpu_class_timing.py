"""Per-class kernel timing: 2048 blocks of one corpus class per launch (diagnostic)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for cls in range(8):
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
    gb = NB * 65536 / 1e9
    print(f"class {cls}: ratio {float(sizes.sum()) / (NB * 65536):.3f}  enc {enc:.3f} ms ({gb / enc * 1e3:.0f} GB/s)  dec {dec:.3f} ms ({gb / dec * 1e3:.0f} GB/s)", flush=True)
