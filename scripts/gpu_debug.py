import sys, time, faulthandler
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, '.')
def P(*a):
    print(*a, flush=True)
import torch
P('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
import lz4_amd
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
from oracle import lz4ref
codec = DeviceCodec(0)
P('codec ok')
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
data = corpus(8)
src = torch.tensor(data, device='cuda:0')
torch.cuda.synchronize()
if which in ('all', 'xxh'):
    h = codec.xxh32(src[:100000]); torch.cuda.synchronize()
    P('xxh32', hex(int(h.cpu().numpy().view('uint32')[0])), hex(lz4ref.xxh32(data[:100000].tobytes())))
if which in ('all', 'dec'):
    import numpy as np
    comp = [lz4ref.block_compress_hc(data[b*65536:(b+1)*65536].tobytes(), max_search_depth=8) for b in range(8)]
    stride = 66048
    buf = np.zeros((8, stride), dtype=np.uint8)
    for b, c in enumerate(comp): buf[b, :len(c)] = np.frombuffer(c, dtype=np.uint8)
    tcomp = torch.tensor(buf, device='cuda:0')
    sizes = torch.tensor([len(c) for c in comp], dtype=torch.int32, device='cuda:0')
    for nb in (1, 2, 8):
        P('decode', nb)
        out, ol, st = codec.decompress_blocks(tcomp[:nb], sizes[:nb], 65536, strict=True)
        torch.cuda.synchronize()
        P(' status', st.cpu().tolist(), ol.cpu().tolist(), bool(torch.equal(out.reshape(-1), src[:nb*65536])))
if which in ('all', 'enc'):
    for nb in (1, 2, 8):
        P('encode', nb)
        slots, sizes, status = codec.compress_blocks(src[:nb*65536], 65536)
        torch.cuda.synchronize()
        P(' status', status.cpu().tolist(), sizes.cpu().tolist())
        hs = sizes.cpu().numpy()
        for b in range(nb):
            ref = lz4ref.block_compress_fast(data[b*65536:(b+1)*65536].tobytes())
            got = slots[b, :int(hs[b])].cpu().numpy().tobytes()
            P('  block', b, len(ref), len(got), got == ref)
P('done')
