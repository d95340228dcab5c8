"""Decode time of blocks with many short sequences (corpus text classes compressed by the HC encoder)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for cls in (3, 7, 5):
    blocks = [corpus_block(cls + 8 * k) for k in range(32)]
    src = torch.from_numpy(np.concatenate([blocks[k % 32] for k in range(NB)])).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536, hc_depth=16)
    res = []
    out = None
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    assert torch.equal(out.reshape(-1), src)
    print(f"class {cls} (HC depth 16, ratio {float(sizes.sum()) / (NB * 65536):.3f}): decode {min(res):.3f} ms for {NB} blocks "
          f"({NB * 65536 / 1e6 / min(res):.1f} GB/s)", flush=True)
