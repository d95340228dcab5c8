"""Decoder phase breakdown with the diagnostic build (LZ4MI_LIB=lz4_amd/liblz4mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
L = N.load()
NB = int(__import__('os').environ.get('NB', '1024'))
names = ['0 token chain', '1 lane fields, scan, checks', '2 out-of-ring matches', '3 literals', '4 matches', '5 flush',
         '6 general loop', '7 other', '8 #batches', '9 #batched sequences', '10 #out-of-ring matches', '11 #general steps',
         '12 window load', '13 #window loads', '14 buffer slide', '15 total cycles']
for arg in sys.argv[1:] or ['5']:
    idx = [int(x) for x in arg.split(',')]
    blocks = [corpus_block(i) for i in idx]
    src = torch.from_numpy(np.concatenate([blocks[k % len(blocks)] for k in range(NB)])).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536)
    codec.decompress_blocks(slots, sizes, 65536); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    L.lz4mi_debug_dec_phases(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); codec.decompress_blocks(slots, sizes, 65536); e1.record(); torch.cuda.synchronize()
    L.lz4mi_debug_dec_phases(buf, 1)
    v = list(buf)
    print(f'blocks {idx}: {e0.elapsed_time(e1):.3f} ms for {NB} blocks', flush=True)
    for i, nm in enumerate(names):
        if 8 <= i <= 11 or i == 13:
            print(f'   {nm:28s} {v[i] / NB:10.1f} per block')
        elif i == 15:
            print(f'   {nm:28s} {v[i] / NB / 1000:10.1f} kticks per block')
        elif v[i]:
            print(f'   {nm:28s} {v[i] / NB / 1000:10.1f} kticks per block')
