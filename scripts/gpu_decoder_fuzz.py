"""Decoder fuzz with hand-built sequences (tests/helpers.py: block_of_sequences): many blocks per launch, every decode
form, strict and lenient with prefixes, bytes and statuses against the oracle.  usage: gpu_decoder_fuzz.py [rounds] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd import _native as N
from lz4_amd.device import DeviceCodec
from oracle import lz4ref as oracle
from tests.helpers import block_of_sequences

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
codec = DeviceCodec(0)
L = N.load()
LIT = [0, 0, 0, 1, 2, 3, 7, 14, 15, 16, 17, 40, 63, 64, 65, 255, 269, 270, 271, 300, 524, 525, 526, 1100, 1799, 1800, 1801, 2040, 2300, 4347, 4352, 5000, 20000]
MAT = [4, 4, 4, 5, 8, 16, 17, 18, 19, 20, 64, 65, 272, 273, 274, 275, 528, 529, 530, 600, 1023, 1024, 1300, 1548, 1549, 1550, 4000, 30000]
DST = [1, 1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63, 64, 65, 200, 255, 256, 1000, 2047, 2048, 2049, 4095, 4096, 4097, 65535]
bad = 0
t_start = time.time()
for rnd in range(rounds):
    rng = random.Random(seed0 * 1000 + rnd)
    style = rnd % 4
    cases = []
    for _ in range(200):
        n = rng.choice([1, 2, 3, 5, 9, 30, 80]) if style != 3 else rng.choice([100, 300])
        seqs = [(rng.choice([1, 5, 40, 300]), 4, 1)]
        for _ in range(n):
            if style == 0:
                seqs.append((rng.choice(LIT), rng.choice(MAT), rng.choice(DST + [rng.randrange(1, 65536)])))
            elif style == 1:   # match-rich, short
                seqs.append((rng.randrange(0, 6), rng.randrange(4, 24), rng.choice(DST)))
            elif style == 2:   # long matches in a row, odd extension byte counts
                seqs.append((rng.choice([0, 0, 1, 16, 40]), rng.choice([273, 274, 528, 529, 530, 784, 1300, 1548, 1549, 3000]), rng.choice(DST)))
            else:              # text-like
                seqs.append((rng.randrange(0, 20), rng.randrange(4, 40), rng.randrange(1, 3000)))
        if rng.random() < 0.7:
            seqs.append((rng.choice([1, 7, 13, 64, 5000]), 0, 0))
        cases.append(seqs)
    for prefix in (b"", rng.randbytes(rng.choice([1, 300, 65536, 70000]))):
        pre = prefix[-65536:]
        built = [block_of_sequences(c, rng, pre) for c in cases]
        built = [(c, o) for c, o in built if 0 < len(o) <= 262144]
        offs, blob = [], bytearray()
        for c, _ in built:
            offs.append(len(blob)); blob += c
        blob += bytes(64)
        comp = torch.tensor(np.frombuffer(bytes(blob), dtype=np.uint8).copy(), device="cuda")
        sizes = torch.tensor([len(c) for c, _ in built], dtype=torch.int32, device="cuda")
        offsets = torch.tensor(offs, dtype=torch.int64, device="cuda")
        expected = torch.tensor([len(o) for _, o in built], dtype=torch.int32, device="cuda")
        d = torch.tensor(np.frombuffer(pre, dtype=np.uint8).copy(), device="cuda") if pre else None
        for form in (1, 2):
            L.lz4mi_ctx_set_decode_form(codec.ctx.handle, form)
            for strict in ((True, False) if not pre else (False,)):
                out, out_len, status = codec.decompress_blocks(comp, sizes, 262144, comp_offsets=offsets, expected=expected,
                                                               strict=strict, dictionary=d)
                out, out_len, status = out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
                for k, (c, o) in enumerate(built):
                    if int(status[k]) != 0 or int(out_len[k]) != len(o) or out[k, :len(o)].tobytes() != o:
                        bad += 1
                        if bad <= 10:
                            print('MISMATCH round', rnd, 'prefix', len(pre), 'form', form, 'strict', strict, 'block', k, 'status', int(status[k]),
                                  'out_len', int(out_len[k]), 'want', len(o), 'seqs', cases[k][:12], flush=True)
    print(f'round {rnd} (style {style}): {len(built)} blocks, mismatches so far {bad}, {time.time() - t_start:.0f} s', flush=True)
print('mismatches:', bad)
