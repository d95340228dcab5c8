"""Frame decode fuzz: frames of the oracle's encoder with random options, damaged (bit flips, truncation, inserted
bytes), through lz4FrameDecode against the oracle's decoder: the bytes, or the exception's class and message.
usage: gpu_frame_fuzz_damaged.py [cases] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
import lz4_amd
from oracle import lz4ref as oracle
from tests.helpers import corpus_block

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed0)
pool = [corpus_block(k, 65536).tobytes() for k in range(48)]
bad = 0
t0 = time.time()
frames = []
for _ in range(24):
    n = rng.choice([70000, 300000, 600000, 1500000])
    data = b"".join(rng.choice(pool) for _ in range(n // 65536 + 1))[:n]
    o = oracle.frame_options(block_size_id=rng.choice([4, 4, 5]), block_independence=rng.random() < 0.4,
                             block_checksum=rng.random() < 0.5, content_checksum=rng.random() < 0.5,
                             content_size=len(data) if rng.random() < 0.3 else None, compression=rng.choice([0, 0, 1]))
    frames.append((oracle.frame_encode(data, o), data))
for k in range(cases):
    frame, data = rng.choice(frames)
    b = bytearray(frame)
    kind = rng.randrange(6)
    if kind == 0:
        for _ in range(rng.choice([1, 1, 2, 3])):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif kind == 1:
        b = b[:rng.randrange(len(b))]
    elif kind == 2:
        p = rng.randrange(len(b)); b[p:p] = rng.randbytes(rng.randrange(1, 5))
    elif kind == 3:
        p = rng.randrange(min(len(b), 40)); b[p] = rng.randrange(256)   # header area
    elif kind == 4:
        b += frame[:rng.randrange(len(frame))]                           # a second, cut frame behind it
    kw, okw = {}, {}
    if rng.random() < 0.3:
        lim = rng.randrange(0, len(data) + 2)
        kw, okw = {"maxOutputBytes": lim}, {"max_output_bytes": lim}
    try:
        want = oracle.frame_decode(bytes(b), **okw)
    except oracle.Lz4RefError as e:
        want = e
    try:
        got = lz4_amd.lz4FrameDecode(bytes(b), **kw)
    except (lz4_amd.Lz4Exception, lz4_amd.RangeError) as e:
        got = e
    ok = (got == want) if isinstance(want, bytes) else (not isinstance(got, bytes) and type(got).__name__ == want.type_name and got.message == want.message)
    if not ok:
        bad += 1
        if bad <= 10:
            print('MISMATCH case', k, 'kind', kind, kw, 'want', repr(want)[:90], 'got', repr(got)[:90], flush=True)
    if k % 100 == 99:
        print(f'{k + 1} cases, mismatches {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
