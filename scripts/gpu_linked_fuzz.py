"""Linked-frame decode fuzz: hand-built frames (tests/test_gpu_linked_resolve.py: synthetic_block) through the
all-at-once path and the serial chain, both against the oracle.  usage: gpu_linked_fuzz.py [frames] [seed]"""
import os, random, sys, time
sys.path.insert(0, '.')
import lz4_amd
from oracle import lz4ref as oracle
from tests.helpers import build_frame
import tests.test_gpu_linked_resolve as T

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
t0 = time.time()
for f in range(frames):
    rng = random.Random(seed0 * 100000 + f)
    bs_id, bmax = rng.choice([(4, 65536), (4, 65536), (5, 262144)])
    dictionary = rng.randbytes(rng.choice([0, 0, 1, 100, 40000, 65536, 70000]))
    history = bytearray(dictionary)
    blocks = []
    for b in range(rng.choice([4, 5, 8, 16, 40])):
        if rng.random() < 0.1:
            raw = rng.randbytes(rng.choice([1, 5000, bmax]))
            blocks.append((raw, raw, True)); history += raw
            continue
        target = rng.choice([bmax, bmax, bmax, 1, 17, 30000, bmax - 1])
        comp, dec = T.synthetic_block(rng, history, target, end_after_match=rng.random() < 0.3)
        if len(comp) > bmax:  # a size word above blockMaxSize is refused by the reference: such a block is stored
            blocks.append((dec, dec, True)); history += dec
            continue
        blocks.append((comp, dec, False)); history += dec
    flg = 0x40 | 0x04 | (0x01 if dictionary else 0) | (0x10 if rng.random() < 0.3 else 0)
    frame = build_frame(flg, bs_id << 4, blocks, oracle.xxh32, add_content_checksum=True, dict_id=5 if dictionary else None)
    want = bytes(history[len(dictionary):])
    kw = {"dictionaryResolver": (lambda i: dictionary)} if dictionary else {}
    okw = {"dictionary_resolver": (lambda i: dictionary)} if dictionary else {}
    assert oracle.frame_decode(frame, **okw) == want
    if rng.random() < 0.5:
        os.environ["LZ4MI_LINK_CHUNK"] = str(rng.choice([65536, 300000, 1 << 20]))
    fast, chain = T.decode_both(frame, require=True, **kw)
    os.environ.pop("LZ4MI_LINK_CHUNK", None)
    if fast != want or chain != want:
        bad += 1
        print('MISMATCH frame', f, 'seed', seed0, 'fast ok', fast == want, 'chain ok', chain == want,
              repr(fast)[:80] if not isinstance(fast, bytes) else '', repr(chain)[:80] if not isinstance(chain, bytes) else '', flush=True)
    if f % 20 == 19:
        print(f'{f + 1} frames, mismatches {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
