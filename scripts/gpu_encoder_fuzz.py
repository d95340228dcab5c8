"""Encoder fuzz: 64 KiB blocks of data with the match structure of hand-built sequences (long and overlapping matches,
short periods, far distances, long literal runs) through the fast and HC encoders, every block's bytes against the
oracle.  usage: gpu_encoder_fuzz.py [rounds] [seed]"""
import random, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from oracle import lz4ref as oracle
from tests.helpers import block_of_sequences

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
codec = DeviceCodec(0)
NB = 256
bad = 0
t0 = time.time()
for rnd in range(rounds):
    rng = random.Random(seed0 * 1000 + rnd)
    blocks = []
    for b in range(NB):
        style = rng.randrange(5)
        data = bytearray()
        while len(data) < 65536:
            seqs = []
            for _ in range(rng.choice([3, 10, 40])):
                if style == 0:
                    seqs.append((rng.choice([0, 1, 5, 20, 300]), rng.choice([4, 5, 8, 20, 70, 300, 2000]), rng.choice([1, 2, 3, 4, 7, 8, 16, 64, 255, 256, 4096, 65535])))
                elif style == 1:
                    seqs.append((rng.randrange(0, 8), rng.randrange(4, 30), rng.randrange(1, 2000)))
                elif style == 2:
                    seqs.append((rng.randrange(0, 4), rng.randrange(4, 12), rng.choice([4, 5, 6, 7, 8, 9, 12, 13])))
                elif style == 3:
                    seqs.append((rng.randrange(20, 200), rng.randrange(4, 10), rng.randrange(1, 65536)))
                else:
                    seqs.append((rng.choice([0, 0, 2000]), rng.choice([4, 30000]), rng.choice([1, 3, 17, 1000])))
            _, out = block_of_sequences([(rng.choice([4, 16, 64]), 4, 1)] + seqs, rng, bytes(data[-65536:]))
            data += out
        blocks.append(bytes(data[:65536]))
    src_np = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
    src = torch.from_numpy(src_np).cuda()
    dic = blocks[rng.randrange(NB)][-rng.choice([100, 4096, 32768, 65536]):]
    d_dic = torch.from_numpy(np.frombuffer(dic, dtype=np.uint8).copy()).cuda()
    for name, kw, ref in (("fast a=1", dict(acceleration=1), lambda d: oracle.block_compress_fast(d, acceleration=1)),
                          ("fast a=4", dict(acceleration=4), lambda d: oracle.block_compress_fast(d, acceleration=4)),
                          ("hc 16", dict(hc_depth=16), lambda d: oracle.block_compress_hc(d, max_search_depth=16)),
                          ("fast dict", dict(acceleration=1, dictionary=d_dic), lambda d: oracle.block_compress_fast(d, dic, 1)),
                          ("hc 8 dict", dict(hc_depth=8, dictionary=d_dic), lambda d: oracle.block_compress_hc(d, dic, 8))):
        slots, sizes, st = codec.compress_blocks(src, 65536, **kw)
        assert int(st.abs().sum()) == 0
        out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict="dict" not in name, dictionary=kw.get("dictionary"))
        assert int(st2.abs().sum()) == 0 and torch.equal(out.reshape(-1), src)
        hs, hz = slots.cpu().numpy(), sizes.cpu().numpy()
        def check(b):
            r = ref(blocks[b])
            return int(hz[b]) == len(r) and hs[b, :len(r)].tobytes() == r
        with ThreadPoolExecutor(16) as ex:
            ok = list(ex.map(check, range(NB)))
        nbad = NB - sum(ok)
        bad += nbad
        if nbad:
            print('MISMATCH round', rnd, name, 'blocks', [b for b, o in enumerate(ok) if not o][:8], flush=True)
    # a linked frame of sixteen of the blocks, fast and HC, against the oracle's frame; and back
    import lz4_amd
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameCompression, Lz4FrameOptions
    fdata = b"".join(blocks[:16])
    for comp_kind, cval in ((Lz4FrameCompression.fast, 0), (Lz4FrameCompression.hc, 1)):
        fo = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, blockIndependence=False, contentChecksum=True, compression=comp_kind)
        got = lz4_amd.lz4FrameEncodeWithOptions(fdata, options=fo)
        want = oracle.frame_encode(fdata, oracle.frame_options(block_size_id=4, block_independence=False, content_checksum=True, compression=cval))
        if got != want or lz4_amd.lz4FrameDecode(got) != fdata:
            bad += 1
            print('MISMATCH round', rnd, 'linked frame', comp_kind, flush=True)
    print(f'round {rnd}: {NB} blocks x 5 encoders + linked frames, mismatches so far {bad}, {time.time() - t0:.0f} s', flush=True)
print('mismatches:', bad)
