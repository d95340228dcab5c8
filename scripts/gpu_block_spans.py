"""Per-block start/end times inside one encode and one decode launch of the bench corpus
(diagnostic build: make -C lz4_amd/csrc prof, LZ4MI_LIB=lz4_amd/liblz4mi_prof.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus
codec = DeviceCodec(0)
L = N.load()
NB = int(__import__('os').environ.get('NB', '16384'))  # at most 65536 (size of the stamp arrays)
src = torch.from_numpy(corpus(NB)).cuda()
for _ in range(2):
    slots, sizes, st = codec.compress_blocks(src, 65536)
    out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True)
torch.cuda.synchronize()
heavy = np.isin(np.arange(NB) % 8, (5, 6))
which = sys.argv[1:] or ['enc', 'dec']
for name, fn in [(w, getattr(L, 'lz4mi_debug_%s_spans' % w)) for w in which]:
    buf = np.zeros(2 * NB, dtype=np.uint64)
    fn(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), NB)
    t0 = buf[0::2].astype(np.float64); t1 = buf[1::2].astype(np.float64)
    base = t0.min()
    s = (t0 - base) / 100.0; e = (t1 - base) / 100.0  # microseconds
    d = e - s
    print(f"{name}: launch span {e.max():.0f} us")
    for lab, m in (('heavy', heavy), ('other', ~heavy)):
        print(f"  {lab}: start p50 {np.percentile(s[m],50):.0f} p90 {np.percentile(s[m],90):.0f} max {s[m].max():.0f} | "
              f"duration mean {d[m].mean():.0f} p90 {np.percentile(d[m],90):.0f} max {d[m].max():.0f} | end max {e[m].max():.0f}")
    k = int(np.argmax(e)); print(f"  last block {k} (class {k%8}): start {s[k]:.0f} dur {d[k]:.0f}")
    top = np.argsort(-d)[:5]
    print("  longest:", [(int(i), int(i) % 8, round(float(s[i])), round(float(d[i]))) for i in top])
    # how many blocks are running over time
    for t in (50, 100, 200, 300, 400, 500, 600, 800, 1000, 1500, 2000, 2500):
        print(f"    t={t}: running {(int(((s<=t)&(e>t)).sum()))} heavy running {int(((s<=t)&(e>t)&heavy).sum())} not started {int((s>t).sum())}")
    cls = np.arange(NB) % 8
    print("  per class mean/max duration us:", {int(c): (round(float(d[cls == c].mean())), round(float(d[cls == c].max()))) for c in range(8)})
    if name == 'dec':  # the blocks that end last: index, class, compressed/output size, start, duration
        hs = sizes.cpu().numpy()
        late = np.argsort(-e)[:12]
        print("  ending last:", [(int(i), int(i) % 8, round(float(hs[i]) / 65536, 3), round(float(s[i])), round(float(d[i]))) for i in late])
        for c in range(8):
            m = cls == c
            slow = m & (d > 150)
            print(f"    class {c}: size ratio {hs[m].min() / 65536:.3f}-{hs[m].max() / 65536:.3f}, blocks over 150 us: {int(slow.sum())}, "
                  f"their start p50 {np.percentile(s[slow], 50) if slow.any() else 0:.0f} max {s[slow].max() if slow.any() else 0:.0f}")
