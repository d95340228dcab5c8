"""Decode launch in which every workgroup holds one LZ block and one literal-only block: the wavefront with the
literal-only block is done at once and helps with the other one (LZ4MI_LIB selects the library)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
for NB in (4096, 3000):
    blocks = [corpus_block((6 if k % 2 == 0 else 0) + 8 * (k // 2 % 64)) for k in range(128)]
    src = torch.from_numpy(np.concatenate([blocks[k % 128] for k in range(NB)])).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536)
    out = None
    best = 1e9
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert torch.equal(out.view(-1)[: src.numel()], src) and int(st2.abs().sum()) == 0
    print(f"alternating LZ / literal-only, {NB} blocks: decode {best:.3f} ms", flush=True)
