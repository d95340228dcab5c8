"""Round trip of the bench corpus, steps in sequence against steps software-pipelined over two contexts: the
encode of step k+1 is launched behind the encode of step k, the decode of step k on a second, lower-priority
stream behind its own encode (it then runs in the tail of the next encode launch, where half of the chip idles
behind the longest blocks)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = 20
DELAY = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device('cuda', 0)
src = torch.from_numpy(corpus(NB)).to(dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
PRI = [int(x) for x in sys.argv[2:4]] if len(sys.argv) > 3 else [0, 0]
enc = DeviceCodec(0, priority=PRI[0]); enc.fence = False
dec = DeviceCodec(0, priority=PRI[1]); dec.fence = False
print('priorities enc/dec', PRI, flush=True)
stride = enc.slot_stride(65536)
slots = [torch.empty((NB, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
outs = [torch.empty((NB, 65536), dtype=torch.uint8, device=dev) for _ in range(2)]
torch.cuda.synchronize()

def sequential(steps):
    for _ in range(steps):
        with torch.cuda.stream(enc.stream):
            s, sizes, st = enc.compress_blocks(src, 65536, out=slots[0])
            o, ol, sd = enc.decompress_blocks(s, sizes, 65536, comp_stride=stride, strict=True, out=outs[0])
    torch.cuda.synchronize()

def pipelined(steps, dstream_codec):
    done = [None, None]
    hold = []
    for k in range(steps):
        b = k & 1
        with torch.cuda.stream(enc.stream):
            if done[b] is not None:
                enc.stream.wait_event(done[b])
            s, sizes, st = enc.compress_blocks(src, 65536, out=slots[b])
            e = torch.cuda.Event(); e.record(enc.stream)
        with torch.cuda.stream(dstream_codec.stream):
            dstream_codec.stream.wait_event(e)
            if DELAY:
                torch.cuda._sleep(DELAY)  # lets the next encode launch place its workgroups first
            o, ol, sd = dstream_codec.decompress_blocks(s, sizes, 65536, comp_stride=stride, strict=True, out=outs[b])
            d = torch.cuda.Event(); d.record(dstream_codec.stream)
        done[b] = d
        hold.append((sizes, st, ol, sd))
    torch.cuda.synchronize()
    return hold[-1], hold[-2]

for name, fn in (('sequential', lambda n: sequential(n)), ('pipelined', lambda n: pipelined(n, dec))):
    fn(3)
    t0 = time.perf_counter(); r = fn(K); dt = time.perf_counter() - t0
    print(f'{name}: {dt / K * 1e3:.3f} ms per step, {NB * 65536 * K / dt / 1e9:.1f} GB/s', flush=True)
for b in range(2):
    assert torch.equal(outs[b].reshape(-1), src)
print('round trips equal the source')
