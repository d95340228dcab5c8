"""Start and end of every block inside one decode launch of the bench corpus (spans build, LZ4MI_LIB=lz4_amd/liblz4mi_spans.so),
kept with the block sizes in gpurun_out/dec_spans.npz and printed as a timeline: wavefronts on heavy / streamed blocks and the
bytes (compressed + output) of the blocks that end in each interval."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
L = N.load()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
src = torch.from_numpy(corpus(NB)).cuda()
ev = (C.c_ulonglong * 8)()
for _ in range(3):
    slots, sizes, st = codec.compress_blocks(src, 65536)
    torch.cuda.synchronize()
    if hasattr(L, 'lz4mi_debug_dec_events'): L.lz4mi_debug_dec_events(ev, 1)
    out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True)
torch.cuda.synchronize()
if hasattr(L, 'lz4mi_debug_dec_events'):
    L.lz4mi_debug_dec_events(ev, 1)
    v = list(ev)
    print('meetings: early offers %d (waited %.0f us each), final offers %d (%.0f us), accepts %d (%.0f us), met a standing offer %d (%.0f us)' % (
        v[0], v[4] / 100 / max(v[0], 1), v[1], v[5] / 100 / max(v[1], 1), v[2], v[6] / 100 / max(v[2], 1), v[3], v[7] / 100 / max(v[3], 1)))
nb = min(NB, 65536)
buf = np.zeros(2 * nb, dtype=np.uint64)
L.lz4mi_debug_dec_spans(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nb)
c = sizes.cpu().numpy().astype(np.int64)[:nb]
t0 = buf[0::2].astype(np.float64); t1 = buf[1::2].astype(np.float64)
base = t0.min()
s = (t0 - base) / 100.0; e = (t1 - base) / 100.0
np.savez('gpurun_out/dec_spans.npz', s=s, e=e, c=c)
n = 65536
heavy = (c * 16 < n * 15) & (c * 32 > n)
d = e - s
print(f"launch span {e.max():.0f} us; heavy {heavy.sum()} blocks: mean {d[heavy].mean():.0f} max {d[heavy].max():.0f} last end {e[heavy].max():.0f}; "
      f"streamed {(~heavy).sum()}: mean {d[~heavy].mean():.0f} p99 {np.percentile(d[~heavy], 99):.0f} max {d[~heavy].max():.0f}")
step = 25.0
for t in np.arange(0, e.max(), step):
    run = (s < t + step) & (e > t)
    ends = (e >= t) & (e < t + step)
    gb = (c[ends] + n).sum() / 1e9
    print(f"  t={t:4.0f}: heavy {int((run & heavy).sum()):5d} streamed {int((run & ~heavy).sum()):5d}  ended {int(ends.sum()):5d} blocks, {gb / (step * 1e-6) / 1e3:5.2f} TB/s;"
          f" streamed started here: {int(((s >= t) & (s < t + step) & ~heavy).sum()):5d} mean dur {d[(s >= t) & (s < t + step) & ~heavy].mean() if ((s >= t) & (s < t + step) & ~heavy).any() else 0:4.0f}")
