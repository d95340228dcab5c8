"""configs[4]'s decode by corpus class: HC (depth 64) output behind the 32 KiB dictionary, per class the shape of the
sequences (a sample of blocks walked on the host) and the decode time of a launch of NB such blocks, both forms.
usage: [NB]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from tools.corpus import corpus_block
codec = DeviceCodec(0)
L = N.load()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
BLOCK = 65536
d_host = corpus_block(3)[:32768].copy()
d_dict = torch.from_numpy(d_host).cuda()


def walk(c):
    """sequences, literal bytes, match bytes, matches of <= 16 bytes, matches from more than 4 KiB back"""
    i, n = 0, len(c)
    seqs = lit_b = mat_b = short = far = 0
    while i < n:
        t = c[i]; i += 1
        ll = t >> 4
        if ll == 15:
            while True:
                b = c[i]; i += 1; ll += b
                if b != 255: break
        i += ll; lit_b += ll
        seqs += 1
        if i >= n: break
        d = c[i] | (c[i + 1] << 8); i += 2
        ml = (t & 15) + 4
        if ml == 19:
            while True:
                b = c[i]; i += 1; ml += b
                if b != 255: break
        mat_b += ml; short += ml <= 16; far += d > 4096
    return seqs, lit_b, mat_b, short, far


for cls in range(8):
    blocks = [corpus_block(cls + 8 * k) for k in range(32)]
    src = torch.from_numpy(np.concatenate([blocks[k % 32] for k in range(NB)])).cuda()
    slots, sizes, st = codec.compress_blocks(src, BLOCK, hc_depth=64, dictionary=d_dict)
    hs = sizes.cpu().numpy()
    st_ = np.array([walk(bytes(slots[k, :int(hs[k])].cpu().numpy())) for k in range(0, 32, 4)], dtype=np.float64).mean(0)
    res = {}
    for form in (1, 2):
        L.lz4mi_ctx_set_decode_form(codec.ctx.handle, form)
        out, t = None, []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, ol, st2 = codec.decompress_blocks(slots, sizes, BLOCK, strict=False, dictionary=d_dict, out=out)
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        assert torch.equal(out.reshape(-1), src)
        res[form] = min(t)
    L.lz4mi_ctx_set_decode_form(codec.ctx.handle, 0)
    print(f"class {cls}: ratio {float(hs.sum()) / (NB * BLOCK):.3f}  per block {st_[0]:.0f} sequences, literals {st_[1]:.0f} B, "
          f"matches {st_[2]:.0f} B, <=16 B {st_[3]:.0f}, >4 KiB back {st_[4]:.0f}  decode of {NB} blocks: single {res[1]:.3f} ms, "
          f"pair {res[2]:.3f} ms ({res[1] * 2.1e6 * 1024 / (NB * st_[0]):.0f} / {res[2] * 2.1e6 * 1024 / (NB * st_[0]):.0f} SIMD-cycles per sequence)", flush=True)
