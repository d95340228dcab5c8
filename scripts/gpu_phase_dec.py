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
HC = __import__('os').environ.get('HC')
# phase slots of the two wavefronts of a block (parsing wavefront first)
from scripts.phase_names import names
for arg in sys.argv[1:] or ['5']:
    idx = [int(x) for x in arg.split(',')]
    blocks = [corpus_block(i) for i in idx]
    src = torch.from_numpy(np.concatenate([blocks[k % len(blocks)] for k in range(NB)])).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536, **({'hc_depth': int(HC)} if HC else {}))  # HC=<depth>: the HC encoder's output (short sequences)
    codec.decompress_blocks(slots, sizes, 65536); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    L.lz4mi_debug_dec_phases(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); codec.decompress_blocks(slots, sizes, 65536); e1.record(); torch.cuda.synchronize()
    L.lz4mi_debug_dec_phases(buf, 1)
    v = list(buf)
    print(f'blocks {idx}: {e0.elapsed_time(e1):.3f} ms for {NB} blocks', flush=True)
    for role, title in ((0, 'parsing wavefront'), (1, 'copying wavefront')):
        print(f'  {title}')
        for i, nm in enumerate(names):
            x = v[32 * role + i]  # PH_SLOTS per wavefront
            if x:
                print(f'   {nm:34s} {x / NB:10.0f} per block' if 'count' in nm else f'   {nm:34s} {x / NB / 1000:10.1f} kcycles per block')
