"""Decoder sweep: corpora of several seeds, compressed by the fast encoder (every block checked against the oracle)
and by the HC encoder (short sequences), decoded under every launch form and at launch sizes on both sides of the
form switch; every decode must give back the source.  One-off check after decoder changes."""
import sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from oracle import lz4ref
from tools.corpus import corpus
codec = DeviceCodec(0)
L = N.load()
SEEDS = [int(x) for x in sys.argv[1:]] or [0, 7, 11]
names = {0: 'by size', 1: 'one wavefront', 2: 'two wavefronts'}
for seed in SEEDS:
    NB = 16384
    data = corpus(NB, seed=seed)
    src = torch.from_numpy(data).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536)
    assert int(st.abs().sum()) == 0
    h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
    def check(b):
        ref = lz4ref.block_compress_fast(data[b * 65536:(b + 1) * 65536].tobytes())
        return int(h_sizes[b]) == len(ref) and h_slots[b, :len(ref)].tobytes() == ref
    with ThreadPoolExecutor(16) as ex:
        bad = [b for b, o in enumerate(ex.map(check, range(NB))) if not o]
    print(f"seed {seed}: fast encoder, {NB - len(bad)} of {NB} blocks identical to the oracle", flush=True)
    assert not bad
    hslots, hsizes, hst = codec.compress_blocks(src[: 3000 * 65536], 65536, hc_depth=8)
    assert int(hst.abs().sum()) == 0
    for form in (0, 1, 2):
        prev = L.lz4mi_set_decode_form(form)
        for label, s_, z_, nbs in (('fast', slots, sizes, (1, 1000, 2048, 2049, 3000, 5000, 16384)), ('hc8', hslots, hsizes, (777, 2048, 3000))):
            for nb in nbs:
                out, ol, st2 = codec.decompress_blocks(s_[:nb], z_[:nb], 65536, strict=True)
                assert int(st2.abs().sum()) == 0, (seed, form, label, nb)
                assert torch.equal(out.reshape(-1), src[: nb * 65536]), (seed, form, label, nb)
                out, ol, st2 = codec.decompress_blocks(s_[:nb], z_[:nb], 65536, strict=False)
                assert int(st2.abs().sum()) == 0 and int((ol != 65536).sum()) == 0, (seed, form, label, nb)
                assert torch.equal(out.reshape(-1), src[: nb * 65536]), (seed, form, label, nb)
        L.lz4mi_set_decode_form(prev)
        print(f"seed {seed}: decode form '{names[form]}': fast and HC output, strict and lenient, 10 launch sizes: all equal to the source", flush=True)
print("sweep passed")
