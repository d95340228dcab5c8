"""Every block of the bench corpus (and of a second corpus seed): GPU fast-encoder bytes against the
oracle's, byte for byte, then GPU decode back to the source.  One-off check, ~30 s of host time."""
import sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np, torch
from lz4_amd.device import DeviceCodec
from oracle import lz4ref
from tools.corpus import corpus
codec = DeviceCodec(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SEEDS = [int(x) for x in sys.argv[2:]] or [0, 7]
for seed in SEEDS:
    data = corpus(NB, seed=seed)
    src = torch.from_numpy(data).cuda()
    slots, sizes, st = codec.compress_blocks(src, 65536)
    assert int(st.abs().sum()) == 0
    out, ol, st2 = codec.decompress_blocks(slots, sizes, 65536, strict=True)
    assert int(st2.abs().sum()) == 0 and torch.equal(out.reshape(-1)[:src.numel()], src)
    h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
    def check(b):
        ref = lz4ref.block_compress_fast(data[b * 65536:(b + 1) * 65536].tobytes())
        return int(h_sizes[b]) == len(ref) and h_slots[b, :len(ref)].tobytes() == ref
    with ThreadPoolExecutor(16) as ex:
        ok = list(ex.map(check, range(NB)))
    bad = [b for b, o in enumerate(ok) if not o]
    print(f"seed {seed}: {NB - len(bad)} of {NB} blocks identical to the oracle; first mismatches {bad[:8]}", flush=True)
    assert not bad
# the HC encoder on a slice of the corpus (the oracle's HC encoder is slow)
NH = min(NB, 2048)
data = corpus(NH, seed=3)
src = torch.from_numpy(data).cuda()
slots, sizes, st = codec.compress_blocks(src, 65536, hc_depth=64)
assert int(st.abs().sum()) == 0
h_slots, h_sizes = slots.cpu().numpy(), sizes.cpu().numpy()
def check_hc(b):
    ref = lz4ref.block_compress_hc(data[b * 65536:(b + 1) * 65536].tobytes(), max_search_depth=64)
    return int(h_sizes[b]) == len(ref) and h_slots[b, :len(ref)].tobytes() == ref
with ThreadPoolExecutor(16) as ex:
    ok = list(ex.map(check_hc, range(NH)))
bad = [b for b, o in enumerate(ok) if not o]
print(f"HC depth 64: {NH - len(bad)} of {NH} blocks identical to the oracle; first mismatches {bad[:8]}", flush=True)
assert not bad
