"""Encoder phase breakdown with the diagnostic build (LZ4MI_LIB=lz4_amd/liblz4mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
L = N.load()
NB = int(__import__('os').environ.get('NB', '1024'))
names = ['0 sched+valid', '1 ld32+table gather', '2 collision loop', '3 ref check+tbl write', '4 backward', '5 forward',
         '6 emit', '7 rehash', '8 #chunks', '9 #matches', '10 tail literals', '11 #ref from memory', '12 realtime 100MHz ticks', '13 max wave lifetime x NB (100MHz ticks)', '14 later chunks', '15 total cycles']
for cls in [int(a) for a in sys.argv[1:]] or [5, 3, 1]:
    blocks = [corpus_block(cls + 8 * k) for k in range(64)]
    src = torch.from_numpy(np.concatenate([blocks[k % 64] for k in range(NB)])).cuda()
    codec.compress_blocks(src, 65536); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    L.lz4mi_debug_enc_phases(buf, 1)
    fx = (C.c_ulonglong * 64)()
    if hasattr(L, 'lz4mi_debug_fast_exits'):
        L.lz4mi_debug_fast_exits(fx, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); codec.compress_blocks(src, 65536); e1.record(); torch.cuda.synchronize()
    L.lz4mi_debug_enc_phases(buf, 1)
    v = list(buf)
    if hasattr(L, 'lz4mi_debug_fast_exits'):
        L.lz4mi_debug_fast_exits(fx, 1)
        print('   short path exits per block (0 = done):', [round(x / NB, 1) for x in fx])
    print(f'class {cls}: {e0.elapsed_time(e1):.3f} ms for {NB} blocks', flush=True)
    fnames = {16: 'F loop+entry', 17: 'F probe read+hash+gather issue', 18: 'F gather wait', 19: 'F cand/ref check', 20: 'F extension',
              21: 'F note+probe insert', 22: 'F rehash', 23: 'F gave up (cost)'}
    for i, nm in fnames.items():
        print(f'   {nm:32s} {v[i] / NB / 100:10.1f} x100 cycles per block')
    for i, nm in enumerate(names):
        if i in (8, 9, 11, 12, 13):
            print(f'   {nm:28s} {v[i] / NB:10.1f} per block')
        else:
            print(f'   {nm:28s} {v[i] / NB / 100:10.1f} us per block (100 MHz ticks)')
