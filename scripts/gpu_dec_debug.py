import sys, random
sys.path.insert(0, '.')
import numpy as np, torch
import lz4_amd
from lz4_amd.device import DeviceCodec
from oracle import lz4ref
from tools.corpus import corpus_block
def mixed(n, seed):
    parts, k = [], 0
    while sum(map(len, parts)) < n:
        parts.append(corpus_block((3, 5, 0, 1, 6, 7)[k % 6] + 8 * (seed + k), 65536).tobytes())
        k += 1
    return b"".join(parts)[:n]
codec = DeviceCodec(0)
for seed in range(40, 48):
    src = mixed(900000 + 777 * (seed - 40), seed)
    data = np.frombuffer(src, dtype=np.uint8)
    nb = (len(src) + 65535) // 65536
    # oracle-compressed blocks, decoded on the device at assorted alignments
    comps = [lz4ref.block_compress_fast(src[b * 65536:(b + 1) * 65536]) for b in range(nb)]
    for mis in (0, 5):
        offs, buf = [], bytearray()
        for c in comps:
            buf += b'\0' * ((-len(buf) + mis) % 16)
            offs.append(len(buf)); buf += c
        comp = torch.tensor(np.frombuffer(bytes(buf), dtype=np.uint8), device='cuda')
        sizes = torch.tensor([len(c) for c in comps], dtype=torch.int32, device='cuda')
        offt = torch.tensor(offs, dtype=torch.int64, device='cuda')
        exp = torch.tensor([min(65536, len(src) - b * 65536) for b in range(nb)], dtype=torch.int32, device='cuda')
        out, ol, st = codec.decompress_blocks(comp, sizes, 65536, comp_offsets=offt, expected=exp, strict=True)
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(-1)[:len(src)]
        bad = np.nonzero(o != data)[0]
        print(seed, mis, 'status', st.cpu().numpy().tolist() if int(st.abs().sum()) else 'ok', 'mismatch', (int(bad[0]), int(bad[0]) // 65536, int(bad[0]) % 65536, len(bad)) if len(bad) else None, flush=True)
        if len(bad):
            p = int(bad[0]); print('   got', o[p-4:p+12].tolist(), 'want', data[p-4:p+12].tolist())
