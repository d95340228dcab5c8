"""HC encoder: index build against parse, per corpus class (diagnostic build: LZ4MI_LIB=lz4_amd/liblz4mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
L = N.load()
NB = int(__import__('os').environ.get('NB', '256'))
for cls in [int(a) for a in sys.argv[1:]] or [0, 3, 5]:
    blocks = [corpus_block(cls + 8 * k) for k in range(32)]
    src = torch.from_numpy(np.concatenate([blocks[k % 32] for k in range(NB)])).cuda()
    codec.compress_blocks(src, 65536, hc_depth=64); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    L.lz4mi_debug_hc_phases(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); codec.compress_blocks(src, 65536, hc_depth=64); e1.record(); torch.cuda.synchronize()
    L.lz4mi_debug_hc_phases(buf, 1)
    v = list(buf)
    print(f'class {cls}: {e0.elapsed_time(e1):.2f} ms for {NB} blocks; per block: index build {v[0] / NB / 1e3:.0f} kcycles, '
          f'parse {v[1] / NB / 1e3:.0f} kcycles, total {v[15] / NB / 1e3:.0f} kcycles', flush=True)
