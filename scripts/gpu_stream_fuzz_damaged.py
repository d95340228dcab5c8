"""Stream decoder fuzz: damaged frames in random chunks through lz4FrameDecoder against the oracle's stream decoder:
what is delivered in front of the failure, and the failure.  usage: gpu_stream_fuzz_damaged.py [cases] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
import lz4_amd
from oracle import lz4ref as oracle
from tests.helpers import corpus_block
from tests.test_oracle_stream_formats import random_chunks

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed0)
pool = [corpus_block(k, 65536).tobytes() for k in range(48)]
frames = []
for _ in range(16):
    n = rng.choice([70000, 300000, 700000])
    data = b"".join(rng.choice(pool) for _ in range(n // 65536 + 1))[:n]
    o = oracle.frame_options(block_size_id=4, block_independence=rng.random() < 0.4, block_checksum=rng.random() < 0.5,
                             content_checksum=rng.random() < 0.5, content_size=len(data) if rng.random() < 0.3 else None)
    frames.append((oracle.frame_encode(data, o), data))
bad = 0
t0 = time.time()
for k in range(cases):
    frame, data = rng.choice(frames)
    b = bytearray(frame)
    kind = rng.randrange(4)
    if kind == 0:
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif kind == 1:
        b = b[:rng.randrange(len(b))]
    elif kind == 2:
        p = rng.randrange(len(b)); b[p:p] = rng.randbytes(rng.randrange(1, 5))
    chunks = random_chunks(bytes(b), random.Random(k), rng.choice([4096, 100000, 400000]))
    lim = rng.randrange(0, len(data) + 2) if rng.random() < 0.25 else None
    want, werr = oracle.stream_decode(chunks, max_output_bytes=lim)
    got, gerr = [], None
    try:
        for p in lz4_amd.lz4FrameDecoder(maxOutputBytes=lim).bind(chunks):
            got.append(p)
    except (lz4_amd.Lz4Exception, lz4_amd.RangeError) as e:
        gerr = e
    same_err = (werr is None and gerr is None) or (werr is not None and gerr is not None and type(gerr).__name__ == werr.type_name and gerr.message == werr.message)
    if b"".join(got) != b"".join(want) or not same_err:  # (pieces are launches here, blocks in the reference)
        bad += 1
        if bad <= 10:
            print('MISMATCH case', k, 'kind', kind, 'limit', lim, 'pieces', len(got), len(want), 'bytes', sum(map(len, got)), sum(map(len, want)), repr(gerr)[:70], repr(werr)[:70], flush=True)
    if k % 100 == 99:
        print(f'{k + 1} cases, mismatches {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
