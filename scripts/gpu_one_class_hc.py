"""Runs the HC encoder over NB blocks of one corpus class (for rocprofv3 counter passes)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
cls = int(sys.argv[1]); NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
codec = DeviceCodec(0)
blocks = [corpus_block(cls + 8 * k) for k in range(64)]
src = torch.from_numpy(np.concatenate([blocks[k % 64] for k in range(NB)])).cuda()
for _ in range(2):
    slots, sizes, st = codec.compress_blocks(src, 65536, hc_depth=64)
torch.cuda.synchronize()
print('ok', int(sizes.sum()))
