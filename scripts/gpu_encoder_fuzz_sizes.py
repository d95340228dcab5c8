"""Encoder fuzz at other block sizes: frames of 256 KiB, 1 MiB and 4 MiB blocks (and odd last blocks) of data with the
match structure of hand-built sequences, fast (several accelerations) and HC, independent and linked, against the
oracle's frames.  usage: [cases] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
import lz4_amd
from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions, Lz4HcOptions
from oracle import lz4ref as oracle
from tests.helpers import block_of_sequences

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
t0 = time.time()
for k in range(cases):
    rng = random.Random(seed0 * 1000 + k)
    n = rng.choice([300000, 1 << 20, (1 << 20) + 77, 3 << 20, (9 << 20) + 12345])
    style = rng.randrange(4)
    data = bytearray()
    while len(data) < n:
        seqs = []
        for _ in range(rng.choice([10, 40, 200])):
            if style == 0:
                seqs.append((rng.choice([0, 1, 5, 20, 300]), rng.choice([4, 5, 8, 20, 70, 300, 2000]), rng.choice([1, 2, 3, 4, 7, 8, 16, 64, 255, 256, 4096, 65535])))
            elif style == 1:
                seqs.append((rng.randrange(0, 8), rng.randrange(4, 30), rng.randrange(1, 2000)))
            elif style == 2:
                seqs.append((rng.randrange(20, 200), rng.randrange(4, 10), rng.randrange(1, 65536)))
            else:
                seqs.append((rng.choice([0, 0, 2000, 70000]), rng.choice([4, 30000, 200000]), rng.choice([1, 3, 17, 1000, 65535])))
        _, out = block_of_sequences([(rng.choice([4, 16, 64]), 4, 1)] + seqs, rng, bytes(data[-65536:]))
        data += out
    data = bytes(data[:n])
    hc = rng.random() < 0.3
    bs = rng.choice([Lz4FrameBlockSize.k256KB, Lz4FrameBlockSize.k1MB, Lz4FrameBlockSize.k4MB])
    o = Lz4FrameOptions(blockSize=bs, blockIndependence=rng.random() < 0.6, blockChecksum=rng.random() < 0.5,
                        contentChecksum=rng.random() < 0.5, compression=Lz4FrameCompression.hc if hc else Lz4FrameCompression.fast,
                        hcOptions=Lz4HcOptions(maxSearchDepth=rng.choice([4, 16])) if hc else None, acceleration=rng.choice([1, 1, 2, 9, 40]))
    ro = oracle.frame_options(block_size_id=bs.bdId, block_independence=o.blockIndependence, block_checksum=o.blockChecksum,
                              content_checksum=o.contentChecksum, compression=o.compression.value, acceleration=o.acceleration,
                              hc_max_search_depth=o.hcOptions.maxSearchDepth if hc else 64)
    got = lz4_amd.lz4FrameEncodeWithOptions(data, options=o)
    want = oracle.frame_encode(data, ro)
    back = lz4_amd.lz4FrameDecode(got)
    if got != want or back != data:
        bad += 1
        print('MISMATCH case', k, n, style, o, 'frame equal', got == want, 'round trip', back == data, flush=True)
    print(f'case {k}: {n} bytes, {bs.name}, hc={hc}, linked={not o.blockIndependence}, frame {len(got)}; mismatches {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
