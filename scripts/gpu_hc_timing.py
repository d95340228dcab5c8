"""HC encoder timing per corpus class (diagnostic)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for cls in range(8):
    blocks = [corpus_block(cls + 8 * k) for k in range(32)]
    src = torch.from_numpy(np.concatenate([blocks[k % 32] for k in range(NB)])).cuda()
    res = []
    for it in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        slots, sizes, st = codec.compress_blocks(src, 65536, hc_depth=64)
        e[1].record()
        torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]))
    print(f"class {cls}: ratio {float(sizes.sum()) / (NB * 65536):.3f}  hc enc {min(res):.1f} ms for {NB} blocks ({NB * 65536 / 1e6 / min(res):.2f} GB/s)", flush=True)
