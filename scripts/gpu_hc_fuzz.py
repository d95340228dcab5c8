"""HC encoder against the oracle on chain-corner inputs (as tests/test_gpu_parity.py builds them) for more seeds
than the test suite runs, single blocks through lz4Compress and batches of structured blocks through the batch call.
One-off check after HC changes.  usage: gpu_hc_fuzz.py [seed ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import lz4_amd as lz
from lz4_amd import Lz4CompressionLevel, Lz4HcOptions
from lz4_amd.device import DeviceCodec
from oracle import lz4ref
from tools.corpus import corpus_block

codec = DeviceCodec(0)
for seed in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(40):
        n = int(rng.integers(200, 24000))
        alpha = int(rng.choice([2, 3, 4, 8, 16, 64]))
        a = rng.integers(0, alpha, size=n, dtype=np.uint8)
        if k % 3 == 0:
            for _ in range(int(rng.integers(1, 30))):
                ln = int(rng.integers(4, 400)); s0 = int(rng.integers(0, n - 1)); d0 = int(rng.integers(0, n - 1))
                ln = min(ln, n - s0, n - d0)
                a[d0:d0 + ln] = a[s0:s0 + ln].copy()
        cases.append(a.tobytes())
    bad = 0
    for idx, data in enumerate(cases):
        for depth in ((1, 3, 64) if idx % 2 else (2, 7, 1000)):
            got = lz.lz4Compress(data, level=Lz4CompressionLevel.hc, hcOptions=Lz4HcOptions(maxSearchDepth=depth))
            if got != lz4ref.block_compress_hc(data, max_search_depth=depth):
                bad += 1
                print("MISMATCH", seed, idx, len(data), depth, flush=True)
    # a batch of corpus blocks of every class at three block sizes and depths
    for block, depth in ((65536, 64), (20000, 16), (4096, 4)):
        nb = 48
        data = np.concatenate([corpus_block(seed * 1000 + k, block) for k in range(nb)])
        src = torch.from_numpy(data).cuda()
        slots, sizes, st = codec.compress_blocks(src, block, hc_depth=depth)
        assert int(st.abs().sum()) == 0
        hs, hz = slots.cpu().numpy(), sizes.cpu().numpy()
        for b in range(nb):
            ref = lz4ref.block_compress_hc(data[b * block:(b + 1) * block].tobytes(), max_search_depth=depth)
            if hs[b, :int(hz[b])].tobytes() != ref:
                bad += 1
                print("MISMATCH batch", seed, block, depth, b, flush=True)
    print(f"seed {seed}: {len(cases) * 3 + 3 * 48} encodes, {bad} mismatches", flush=True)
    assert bad == 0
print("hc fuzz passed")
