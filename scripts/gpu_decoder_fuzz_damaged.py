"""Decoder fuzz on damaged hand-built blocks: bit flips, truncations, appended bytes, wrong sizes; every block's status
and, when it decodes, every byte against the oracle, strict and lenient, forms 1 and 2.  usage: [rounds] [seed]"""
import random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from oracle import lz4ref as oracle
from tests.helpers import block_of_sequences

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
codec = DeviceCodec(0)
L = N.load()
LIT = [0, 0, 1, 3, 14, 15, 16, 40, 269, 270, 300, 525, 1100, 1800, 2300, 4352]
MAT = [4, 4, 5, 8, 18, 19, 20, 64, 273, 274, 528, 529, 600, 1300, 1548, 1549]
DST = [1, 2, 3, 7, 31, 64, 200, 255, 256, 4095, 4096, 65535]
bad = 0
t0 = time.time()
SLOT = 65536
for rnd in range(rounds):
    rng = random.Random(seed0 * 1000 + rnd)
    PRE = rng.randbytes(65536) if rnd % 3 == 2 else b""
    cases = []
    while len(cases) < 400:
        if len(cases) % 16 < 4:  # token-dense blocks (hundreds of short sequences in a row)
            n = rng.choice([60, 150, 400])
            seqs = [(rng.choice([1, 5, 40]), 4, 1)] + [(rng.randrange(0, 4), rng.randrange(4, 21), rng.choice(DST)) for _ in range(n)]
        else:
            n = rng.choice([1, 3, 9, 30])
            seqs = [(rng.choice([1, 5, 40]), 4, 1)] + [(rng.choice(LIT), rng.choice(MAT), rng.choice(DST)) for _ in range(n)]
        if rng.random() < 0.7:
            seqs.append((rng.choice([1, 7, 300]), 0, 0))
        c, o = block_of_sequences(seqs, rng, PRE)
        if not 0 < len(o) <= SLOT - 8:
            continue
        for _ in range(4):
            b = bytearray(c)
            kind = rng.randrange(5)
            if kind == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif kind == 1:
                b = b[:rng.randrange(len(b))]
            elif kind == 2:
                b += rng.randbytes(rng.randrange(1, 6))
            elif kind == 3:
                p = rng.randrange(len(b)); b[p:p + 2] = rng.randbytes(2)
            size = len(o) + (rng.randrange(-3, 4) if rng.random() < 0.3 else 0)
            cases.append((bytes(b), max(0, min(size, SLOT))))
    offs, blob = [], bytearray()
    for c, _ in cases:
        offs.append(len(blob)); blob += c
    blob += bytes(64)
    comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
    sizes = torch.tensor([len(c) for c, _ in cases], dtype=torch.int32, device="cuda")
    offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
    expected = torch.tensor([z for _, z in cases], dtype=torch.int32, device="cuda")
    pre = PRE  # every third round: lenient decode behind a prefix that the matches reach into
    d = torch.tensor(np.frombuffer(pre, dtype=np.uint8).copy(), device="cuda") if pre else None
    for strict in ((True, False) if not pre else (False,)):
        def ref_of(k):
            c, z = cases[k]
            try:
                return (0, oracle.block_decompress(c, z) if strict else oracle.block_decompress_into(c, pre, len(pre) + z)[len(pre):])
            except oracle.Lz4RefError as e:
                return (e.status, None)
        with ThreadPoolExecutor(16) as ex:
            refs = list(ex.map(ref_of, range(len(cases))))
        for form in (1, 2):
            L.lz4mi_ctx_set_decode_form(codec.ctx.handle, form)
            out, out_len, status = codec.decompress_blocks(comp, sizes, SLOT, comp_offsets=offsets, expected=expected, strict=strict, dictionary=d)
            out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
            for k, (rs, rb) in enumerate(refs):
                ok = int(status[k]) == rs and (rb is None or (int(out_len[k]) == len(rb) and out[k, :len(rb)].tobytes() == rb))
                if not ok:
                    bad += 1
                    if bad <= 10:
                        print('MISMATCH round', rnd, 'strict', strict, 'form', form, 'block', k, 'status', int(status[k]), 'want', rs,
                              'out_len', int(out_len[k]), 'want', None if rb is None else len(rb), cases[k][0][:24].hex(), cases[k][1], flush=True)
    print(f'round {rnd}: {len(cases)} damaged blocks, mismatches so far {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
