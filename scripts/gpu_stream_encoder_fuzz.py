"""Stream encoder fuzz: random options, random chunkings (also chunks above the parallel-copy and host-hash thresholds),
pieces against the oracle's stream encoder, and the frame against the one-shot encoder.  usage: [cases] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
import lz4_amd
from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions, Lz4HcOptions
from oracle import lz4ref as oracle
from tests.helpers import corpus_block
from tests.test_oracle_stream_formats import random_chunks

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed0)
pool = [corpus_block(k, 65536).tobytes() for k in range(64)]
bad = 0
t0 = time.time()
for k in range(cases):
    n = rng.choice([1, 70000, 300000, 1500000, 12 << 20, 40 << 20])
    data = b"".join(rng.choice(pool) for _ in range(n // 65536 + 1))[:n]
    hc = rng.random() < 0.2 and n <= 1500000
    use_dict = rng.random() < 0.3
    dictionary = b"".join(rng.choice(pool) for _ in range(2))[:rng.choice([100, 40000, 100000])] if use_dict else None
    o = Lz4FrameOptions(blockSize=rng.choice([Lz4FrameBlockSize.k64KB, Lz4FrameBlockSize.k256KB, Lz4FrameBlockSize.k1MB]),
                        blockIndependence=rng.random() < 0.6, blockChecksum=rng.random() < 0.5,
                        contentChecksum=rng.random() < 0.6, contentSize=len(data) if rng.random() < 0.3 else None,
                        dictId=7 if use_dict else None,
                        compression=Lz4FrameCompression.hc if hc else Lz4FrameCompression.fast,
                        hcOptions=Lz4HcOptions(maxSearchDepth=8) if hc else None, acceleration=rng.choice([1, 1, 3]))
    ro = oracle.frame_options(block_size_id=o.blockSize.bdId, block_independence=o.blockIndependence, block_checksum=o.blockChecksum,
                              content_checksum=o.contentChecksum, content_size=o.contentSize, dict_id=o.dictId,
                              compression=o.compression.value, acceleration=o.acceleration, hc_max_search_depth=8 if hc else 64)
    chunks = random_chunks(data, random.Random(k), rng.choice([8192, 300000, 9 << 20, 33 << 20]))
    got = list(lz4_amd.lz4FrameEncoderWithOptions(options=o, dictionary=dictionary).bind(chunks))
    want, werr = oracle.stream_encode(chunks, ro, dictionary)
    one = lz4_amd.lz4FrameEncodeWithOptions(data, options=o, dictionary=dictionary)
    back = lz4_amd.lz4FrameDecode(one, dictionaryResolver=(lambda i: dictionary) if use_dict else None)
    if werr is not None or got != want or b"".join(got) != one or back != data:
        bad += 1
        print('MISMATCH case', k, n, o, 'pieces', len(got), len(want), flush=True)
    if k % 10 == 9:
        print(f'{k + 1} cases, mismatches {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
