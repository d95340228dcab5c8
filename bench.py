#!/usr/bin/env python3
"""Headline benchmark: fast compress + decompress of 64 KiB independent blocks.

N = 1 (BASELINE.json configs[1]): a 1 GiB synthetic mixed-entropy corpus (tools/corpus.c), resident in
HBM before the timed region.  One step = one pass of the hot path over the corpus: the batched fast
block encoder, then the batched strict block decoder on the bytes it produced.

N > 1 (configs[3]): one frame sharded by contiguous block range, one process per GPU.  One step per
rank = compress its blocks and plan their frame records -> RCCL all-gather of the record sizes ->
exclusive prefix sum -> write the records into the rank's window of the frame -> decode that window
(size words, stored blocks, compressed blocks) -> all-gather of the decoded sizes + prefix sum (the
format allows short blocks) -> pack the blocks if any was short.  Default: weak scaling, 1 GiB per
rank; --corpus-gib G: strong scaling, G GiB in total (64 = configs[3] as written).

--config 2 / --config 4: the other single-GPU configurations of BASELINE.json as bench lines of
their own (4 MiB blocks + block/content checksums; LZ4HC depth 64 + dictionary-primed decode).

value = uncompressed bytes through the round trip per second, summed over ranks:
        total corpus bytes * steps / wall, wall = max over ranks between two barrier+sync pairs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
METRIC = "compress/decompress GB/s per GPU & node, 64KiB indep. blocks; ratio parity"

# What the reference publishes for this path (benchmark/baseline/macos.txt:20-58): Dart VM 3.10.4, one isolate,
# MacBookPro11,1, Intel i7-4558U @ 2.8 GHz.  MiB/s of uncompressed bytes; other hardware, printed unscaled.
REFERENCE_PUBLISHED = {
    "source": "benchmark/baseline/macos.txt:20-58", "hardware": "Intel i7-4558U 2.8 GHz, Dart VM 3.10.4, 1 isolate",
    "unit": "MiB/s",
    "random_1MiB": {"block_fast_compress": 347.4, "block_fast_decompress": 378.4, "block_hc_compress": 1.9,
                    "block_hc_decompress": 305.1, "frame_fast_encode": 85.6, "frame_fast_decode": 111.5,
                    "frame_hc_encode": 18.7, "frame_hc_decode": 115.8},
    "repeating_1MiB": {"block_fast_compress": 305.8, "block_fast_decompress": 223.1, "block_hc_compress": 144.3,
                       "block_hc_decompress": 234.3, "frame_fast_encode": 114.7, "frame_fast_decode": 95.9,
                       "frame_hc_encode": 163.5, "frame_hc_decode": 89.9},
}
CPU_NOTE = ("C port of the reference algorithm (bit-identical output): an upper bound on what the Dart VM reaches "
            "on the same cores; the Dart SDK is absent on this host, so benchmark/lz4_benchmark.dart itself cannot run")


def _spans(n_items, threads):
    step = (n_items + threads * 8 - 1) // (threads * 8)
    return [(lo, min(n_items, lo + step)) for lo in range(0, n_items, step)]


def reference_throughput(fn, bytes_processed, warmup_iters=8, target_ms=500.0):
    """The reference harness's timing rule (benchmark/lz4_benchmark.dart:154-183): `warmup_iters` untimed calls,
    then batches of 1, 2, 4, ... calls until one timed batch lasts `target_ms`; MiB/s of that batch."""
    for _ in range(warmup_iters):
        fn()
    iters = 1
    while True:
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        dt = time.perf_counter() - t0
        if dt * 1e3 >= target_ms:
            return bytes_processed * iters / (1024.0 * 1024.0) / dt, iters
        iters *= 2


def cpu_baseline(data: np.ndarray, threads: int, hc_depth: int | None = None, dictionary: bytes | None = None,
                 label: str = "fast"):
    """The scalar C oracle (bit-identical output) on the host cores: same blocks, same round trip, timed by the
    reference harness's rule, once on `threads` threads (one block per task) and once on one thread (a prefix of
    the blocks, so that the whole leg stays within seconds).  Prints the reference's `[label] compress: X MiB/s`
    lines (benchmark/lz4_benchmark.dart:92-93) to stderr."""
    from oracle import lz4ref
    L = lz4ref.lib()
    import ctypes as C
    nb = data.size // BLOCK
    bound = lz4ref.compress_bound(BLOCK)
    comp = np.empty((nb, bound), dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.int64)
    conf = np.zeros(nb, dtype=np.int64)
    dlen = len(dictionary) if dictionary else 0
    dbuf = np.frombuffer(dictionary, dtype=np.uint8).copy() if dlen else None
    # decode goes through the lenient decoder behind a dictionary prefix when there is one (what the frame decoder
    # does per block, lib/src/frame/lz4_frame_decoder.dart:273-290): one prefix + output area per worker span
    out = np.empty(nb * BLOCK, dtype=np.uint8)
    src_p, comp_p, out_p = data.ctypes.data, comp.ctypes.data, out.ctypes.data

    level, param = (0, 1) if hc_depth is None else (1, hc_depth)
    sizes_p = sizes.ctypes.data
    dict_addr = dbuf.ctypes.data if dlen else None

    def enc(lo, hi):  # one C call per span of blocks: the GIL is released for all of it
        st = L.lz4ref_blocks_compress(src_p + lo * BLOCK, hi - lo, BLOCK, dict_addr, dlen, level, param,
                                      comp_p + lo * bound, bound, sizes_p + 8 * lo)
        assert st == 0, st

    def dec(lo, hi):
        tmp = None
        if dlen:
            tmp = np.empty(dlen + BLOCK, dtype=np.uint8)
            tmp[:dlen] = dbuf
        st = L.lz4ref_blocks_decompress(comp_p + lo * bound, bound, sizes_p + 8 * lo, hi - lo, BLOCK,
                                        tmp.ctypes.data if dlen else None, dlen, out_p + lo * BLOCK)
        assert st == 0, st

    def run(fn, n_items, nthreads, ex):
        spans = _spans(n_items, nthreads)
        if ex is None:
            return lambda: [fn(lo, hi) for lo, hi in spans]
        return lambda: list(ex.map(lambda s: fn(*s), spans))

    one = min(nb, 2048 if hc_depth is None else 64)  # the one-thread leg works on a prefix
    mib = {}
    iters = {}
    with ThreadPoolExecutor(threads) as ex:
        run(enc, nb, threads, ex)()  # sizes[] of every block before anything is timed
        for name, fn in (("compress", enc), ("decompress", dec)):
            warm, target = (8, 500.0) if hc_depth is None else (1, 500.0)  # an HC pass takes seconds by itself
            mib[name], iters[name] = reference_throughput(run(fn, nb, threads, ex), nb * BLOCK, warm, target)
    assert np.array_equal(out, data)
    for name, fn in (("compress", enc), ("decompress", dec)):
        warm = 8 if hc_depth is None else 1
        mib[name + "_1"], iters[name + "_1"] = reference_throughput(run(fn, one, 1, None), one * BLOCK, warm, 500.0)
    for b in range(nb):  # not timed: blocks that break the LZ4 end-of-block rules
        conf[b] = 0 if L.lz4ref_block_conformant(C.c_char_p(comp_p + b * bound), int(sizes[b])) else 1
    total_c = int(sizes.sum())
    print(f"--- synthetic mixed-entropy {data.size >> 20}MiB, 64KiB blocks: C port of the reference on host cores ---",
          file=sys.stderr)
    print(f"input: {data.size} bytes", file=sys.stderr)
    print(f"[{label}] compressed: {total_c} bytes (ratio: {total_c / data.size:.3f})", file=sys.stderr)
    print(f"[{label}] compress: {mib['compress']:.1f} MiB/s", file=sys.stderr)
    print(f"[{label}] decompress: {mib['decompress']:.1f} MiB/s", file=sys.stderr)
    print(f"[{label}-1thread] compress: {mib['compress_1']:.1f} MiB/s", file=sys.stderr)
    print(f"[{label}-1thread] decompress: {mib['decompress_1']:.1f} MiB/s", file=sys.stderr)
    to_gb = 1024.0 * 1024.0 / 1e9
    c, d, c1, d1 = (mib[k] * to_gb for k in ("compress", "decompress", "compress_1", "decompress_1"))
    return {
        "value": 1.0 / (1.0 / c + 1.0 / d), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"{data.size >> 20} MiB of the corpus, {nb} blocks, compress and decompress each timed by the "
                  f"reference harness's rule (8 warm-up passes, then 1, 2, 4, ... passes until a timed batch lasts "
                  f"500 ms; benchmark/lz4_benchmark.dart:154-183), {threads} threads (one block per task); value = "
                  f"round-trip rate 1 / (1/compress + 1/decompress)"
                  + (f"; dictionary of {dlen} bytes on both sides" if dlen else ""),
        "note": CPU_NOTE,
        "host_cpu_count": os.cpu_count(),
        "compress_GBps": c, "decompress_GBps": d,
        "compress_MiBps": mib["compress"], "decompress_MiBps": mib["decompress"],
        "timed_passes": {k: iters[k] for k in ("compress", "decompress")},
        "value_1thread": 1.0 / (1.0 / c1 + 1.0 / d1),
        "compress_GBps_1thread": c1, "decompress_GBps_1thread": d1,
        "sample_1thread": f"first {one} blocks, same rule",
        "compressed_bytes": total_c,
        "blocks": nb,
        "nonconformant_blocks": int(conf.sum()),
    }


def load_corpus(dev, first_block, nblocks):
    """The rank's blocks in HBM, generated and uploaded 1 GiB at a time (host memory stays bounded)."""
    from tools.corpus import corpus
    src = torch.empty(nblocks * BLOCK, dtype=torch.uint8, device=dev)
    host0 = None
    for lo in range(0, nblocks, 16384):
        n = min(16384, nblocks - lo)
        part = corpus(n, BLOCK, first=first_block + lo)
        if host0 is None:
            host0 = part
        src[lo * BLOCK:(lo + n) * BLOCK].copy_(torch.from_numpy(part))
    torch.cuda.synchronize()
    return src, host0


CLASS_NAMES = ("random", "period-32 ramp", "byte run", "text-like", "text-like", "LZ-synthetic", "LZ-synthetic",
               "random/text splice")


def by_class(codec, src, nb, stride, acceleration):
    """The two kernels on each corpus class by itself (block index mod 8), one launch of that class's blocks: the
    headline launch mixes literal-only blocks, which are plain copies, with match-rich ones, and its fraction of the
    roofline says little about either.  Not part of the timed region."""
    out = {}
    blocks = src.view(nb, BLOCK)
    for cls in range(8):
        part = blocks[cls::8].contiguous().reshape(-1)
        n = part.numel() // BLOCK
        if n == 0:
            continue
        slots = dec = None
        best = [1e9, 1e9]
        with torch.cuda.stream(codec.stream):
            for _ in range(3):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(codec.stream)
                slots, sizes, st_c = codec.compress_blocks(part, BLOCK, acceleration=acceleration, out=slots)
                e[1].record(codec.stream)
                dec, out_len, st_d = codec.decompress_blocks(slots, sizes, BLOCK, comp_stride=stride, strict=True, out=dec)
                e[2].record(codec.stream)
                codec.synchronize()
                best = [min(best[0], e[0].elapsed_time(e[1])), min(best[1], e[1].elapsed_time(e[2]))]
        assert int(st_c.abs().sum()) == 0 and int(st_d.abs().sum()) == 0 and torch.equal(dec.reshape(-1), part)
        c = int(sizes.to(torch.int64).sum())
        algo = (part.numel() + c) / 1e9
        out[f"{cls} {CLASS_NAMES[cls]}"] = {
            "blocks": n, "ratio": c / part.numel(),
            "encode_ms": best[0], "decode_ms": best[1],
            "encode_frac": algo / (best[0] / 1e3) / HBM_PEAK_GBPS, "decode_frac": algo / (best[1] / 1e3) / HBM_PEAK_GBPS,
        }
    return out


def pipelined_round_trip(codec, src, nb, stride, acceleration, slots, out, steps):
    """The same round trip with the steps software-pipelined over two contexts (not the headline: `value` times the
    steps one after the other).  The encode of step k+1 is launched behind the encode of step k, the decode of
    step k on a second context's stream behind its own encode, into double-buffered slots and outputs: the decode
    then runs beside the next encode launch, whose second half leaves wavefront slots free behind its longest
    blocks.  What a caller that streams batches through the codec gets."""
    from lz4_amd.device import DeviceCodec
    dec = DeviceCodec(codec.device)
    dec.fence = False
    dev = src.device
    slots2 = [slots, torch.empty_like(slots)]
    out2 = [out, torch.empty_like(out)]

    def run(n):
        done, hold = [None, None], []
        for k in range(n):
            b = k & 1
            with torch.cuda.stream(codec.stream):
                if done[b] is not None:
                    codec.stream.wait_event(done[b])  # the decode that read these slots two steps ago
                s, sizes, st_c = codec.compress_blocks(src, BLOCK, acceleration=acceleration, out=slots2[b])
                e = torch.cuda.Event()
                e.record(codec.stream)
            with torch.cuda.stream(dec.stream):
                dec.stream.wait_event(e)
                o, out_len, st_d = dec.decompress_blocks(s, sizes, BLOCK, comp_stride=stride, strict=True, out=out2[b])
                d = torch.cuda.Event()
                d.record(dec.stream)
            done[b] = d
            hold.append((sizes, st_c, out_len, st_d))  # kept until the launches that use them have run
        torch.cuda.synchronize()
        return hold

    run(3)
    t0 = time.perf_counter()
    hold = run(steps)
    wall = time.perf_counter() - t0
    for sizes, st_c, out_len, st_d in hold[-2:]:
        assert int(st_c.abs().sum()) == 0 and int(st_d.abs().sum()) == 0
    for o in out2:
        flat = o.reshape(-1)
        for lo in range(0, src.numel(), 1 << 30):
            assert torch.equal(flat[lo:lo + (1 << 30)], src[lo:lo + (1 << 30)]), "pipelined decode(encode(x)) != x"
    del dec
    return {"value": nb * BLOCK * steps / wall / 1e9, "unit": "GB/s", "ms_per_step": wall / steps * 1e3, "steps": steps,
            "note": "steps overlapped over two contexts (decode of step k beside the encode of step k+1); "
                    "not the headline, which runs the steps one after the other"}


def traffic_of(kernel, blocks_per_launch, phase):
    """HBM bytes per launch from the PMC passes, or None when they profiled another launch shape than this run's
    (another number of blocks per launch, or the sharded step's window kernels)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(tpath)).get(kernel, {})
    except Exception:
        return None
    if phase != kernel or rec.get("blocks_per_launch", 16384) != blocks_per_launch:
        return None
    return rec.get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=16384, help="64 KiB blocks per GPU (16384 = 1 GiB)")
    ap.add_argument("--corpus-gib", type=float, default=0.0,
                    help="strong scaling: total corpus in GiB, split over the ranks (64 = configs[3])")
    ap.add_argument("--acceleration", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs index: 1 block round trip (3 with --gpus N), 2 = 4 MiB-block frames "
                         "with checksums, 4 = LZ4HC + dictionary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-by-class", action="store_true",
                    help="leave out the per-class launches behind the timed region (profiling passes: every launch "
                         "of a kernel in the profile is then one of the timed shape)")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: what libraries print there (RCCL's version banner at
    # communicator set-up) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.config in (2, 4):
        from scripts.bench_configs import run_config
        json_out.write(json.dumps(run_config(args)) + "\n")
        json_out.flush()
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LZ4MI_BENCH_DIST=1 takes the multi-rank code path (RCCL init, size exchange, frame assembly) with one
    # rank, to rehearse it on a one-GPU box
    multi = world > 1 or os.environ.get("LZ4MI_BENCH_DIST") == "1" or args.config == 3 or args.corpus_gib > 0
    # LZ4MI_BENCH_ONE_GPU=1 puts every rank on GPU 0 and exchanges through gloo (RCCL takes one rank per device):
    # a rehearsal of the rank > 0 code path on a one-GPU box, not a measurement
    one_gpu = os.environ.get("LZ4MI_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group("gloo")
            # gloo exchanges host tensors: stage the two collectives the step uses through the host
            _ag, _ar = dist.all_gather_into_tensor, dist.all_reduce

            def _ag_host(out, inp, group=None):
                torch.cuda.current_stream().synchronize()
                o = torch.empty(out.shape, dtype=out.dtype)
                _ag(o, inp.cpu(), group=group)
                out.copy_(o)

            def _ar_host(t, op=dist.ReduceOp.SUM, group=None):
                c = t.cpu()
                _ar(c, op=op, group=group)
                t.copy_(c)
            dist.all_gather_into_tensor, dist.all_reduce = _ag_host, _ar_host
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd.device import DeviceCodec
    from lz4_amd.sharding import decoded_offsets, frame_offsets_uniform

    strong = args.corpus_gib > 0
    nb = int(args.corpus_gib * 16384) // world if strong else args.blocks
    src, data0 = load_corpus(dev, rank * nb, nb)
    codec = DeviceCodec(local_rank)
    codec.fence = False  # everything below is issued on the codec's stream
    stream = codec.stream
    stride = codec.slot_stride(BLOCK)
    out = torch.empty((nb, BLOCK), dtype=torch.uint8, device=dev)
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k64KB, acceleration=args.acceleration)
    header = codec.frame_header(opts)
    if multi:
        window = torch.empty(nb * (BLOCK + 8), dtype=torch.uint8, device=dev)  # this rank's part of the frame
    else:
        slots = torch.empty((nb, stride), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    phases = ("shard_compress", "shard_write", "size_exchange", "shard_decode", "decoded_size_exchange") if multi \
        else ("encode_fast_kernel", "decode_kernel")
    ev = {k: [] for k in phases}  # per phase: one list of event pairs per timed step
    keep = {}
    # A rank's shard goes through the encoder in sub-launches of at most this many blocks (8 GiB), so that the
    # library's scratch for compressed payloads stays bounded whatever the shard size (64 GiB on one GPU included).
    # A rank's window of the frame needs only the rank's own record sizes (offsets relative to its first record):
    # it is written sub-launch by sub-launch; the all-gather of the sizes places the windows in the frame.
    SUB = 131072
    rec_all = torch.empty(nb, dtype=torch.int32, device=dev) if multi else None
    rel_all = torch.empty(nb, dtype=torch.int64, device=dev) if multi else None

    def step(record):
        with torch.cuda.stream(stream):
            pairs = {k: [] for k in phases}
            last = [None]

            def mark(phase=None):
                if not record:
                    return
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                if phase is not None:
                    pairs[phase].append((last[0], e))
                last[0] = e
            mark()
            if not multi:
                _, sizes, st_c = codec.compress_blocks(src, BLOCK, acceleration=args.acceleration, out=slots)
                mark("encode_fast_kernel")
                _, out_len, st_d = codec.decompress_blocks(slots, sizes, BLOCK, comp_stride=stride, strict=True, out=out)
                mark("decode_kernel")
                keep.update(sizes=sizes, st_c=st_c, st_d=st_d, out_len=out_len)
            else:
                base = torch.zeros(1, dtype=torch.int64, device=dev)  # where the next sub-launch's records start
                for lo in range(0, nb, SUB):
                    hi = min(nb, lo + SUB)
                    part = src[lo * BLOCK:hi * BLOCK]
                    rec = codec.shard_compress(part, opts)     # compress + record plan (device)
                    mark("shard_compress")
                    r64 = rec.to(torch.int64)
                    ends = torch.cumsum(r64, 0) + base
                    rel = ends - r64                           # offsets in this rank's window of the frame
                    rec_all[lo:hi] = rec
                    rel_all[lo:hi] = rel
                    codec.shard_write(part, opts, rel, window)  # records into the window
                    base = ends[-1:]
                    mark("shard_write")
                mine, body_end = frame_offsets_uniform(rec_all, len(header))  # RCCL all-gather + prefix sum
                mark("size_exchange")
                _, out_len, st_d = codec.shard_decode(window, rel_all, opts, out=out)
                mark("shard_decode")
                doff, total = decoded_offsets(out_len, uniform=True)  # the format allows short blocks
                if not bool((out_len == BLOCK).all()):         # none here: blocks already sit at b * 64 KiB
                    raise AssertionError("short block in the bench corpus")
                mark("decoded_size_exchange")
                keep.update(rec=rec_all, mine=mine, body_end=body_end, rel=rel_all, out_len=out_len, st_d=st_d, total=total)
            if record:
                for k in phases:
                    ev[k].append(pairs[k])

    for _ in range(args.warmup):
        step(False)
        if one_gpu:
            print(f"rank {rank}: warm-up step done", file=sys.stderr, flush=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    wall = time.perf_counter() - t0
    if multi:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- correctness of what was timed (outside the timed region)
    codec.fence = True  # from here on the codec's stream is ordered against torch's
    assert int(keep["st_d"].abs().sum()) == 0, "block status != OK"
    flat = out.reshape(-1)
    for lo in range(0, src.numel(), 1 << 30):  # piecewise: a whole-shard compare would allocate a shard of booleans
        assert torch.equal(flat[lo:lo + (1 << 30)], src[lo:lo + (1 << 30)]), "decode(encode(x)) != x"
    extra = {}
    if not multi:
        assert int(keep["st_c"].abs().sum()) == 0
        total_c = int(keep["sizes"].to(torch.int64).sum())
        # blocks that break the LZ4 end-of-block rules (the reference's encoder ends blocks inside matches,
        # lib/src/block/lz4_block_encoder.dart:95,118): reported, not "fixed"
        extra["nonconformant_blocks"] = int(codec.block_conformance(slots, keep["sizes"], stride).sum())
    else:
        rec = keep["rec"].to(torch.int64)
        total_c = int(rec.sum()) - 4 * nb  # payload bytes of this rank's records
        # every rank's records, once more, through a second all-gather: RCCL saw all ranks and the offsets
        # this rank used are the prefix sums of everybody's sizes
        allrec = torch.empty(world * nb, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allrec, rec)
        starts = torch.cumsum(allrec, 0) - allrec + len(header)
        assert torch.equal(starts[rank * nb:(rank + 1) * nb], keep["mine"]), "record offsets != prefix sums"
        assert torch.equal(keep["mine"] - keep["mine"][0], keep["rel"]), "window offsets != the rank's own prefix sums"
        assert int(keep["body_end"]) == int(starts[-1] + allrec[-1])
        assert int(keep["total"]) == world * nb * BLOCK
        extra["ranks_seen_by_rccl"] = dist.get_world_size()
        if one_gpu:
            extra["rehearsal"] = "every rank on GPU 0, exchange through gloo: not a measurement"
        extra["frame_bytes"] = int(keep["body_end"]) + 4
        if rank == 0:  # this rank's window against the oracle's frame of the same blocks, byte for byte
            from oracle import lz4ref
            chk = min(nb, 4096)
            ref = lz4ref.frame_encode(data0[: chk * BLOCK].tobytes(), lz4ref.frame_options(block_size_id=4))
            body = ref[len(header):-4]
            got = window[: len(body)].cpu().numpy().tobytes()
            assert got == body, "frame window differs from the oracle's frame"
            extra["window_checked_against_oracle_blocks"] = chk

    if rank == 0:
        ms = {k: float(np.mean([sum(a.elapsed_time(b) for a, b in pairs) for pairs in v])) for k, v in ev.items()}
        n_bytes = nb * BLOCK
        gb = n_bytes / 1e9
        algo = (n_bytes + total_c) / 1e9  # algorithmic bytes of one launch: read n write c / read c write n
        algo_of = {"encode_fast_kernel": algo, "decode_kernel": algo, "shard_compress": algo, "shard_decode": algo,
                   "shard_write": 2 * (total_c + 4 * nb) / 1e9}
        roof = {k: {"achieved": algo_of[k] / (ms[k] / 1e3), "frac": algo_of[k] / (ms[k] / 1e3) / HBM_PEAK_GBPS,
                    "avg_launch_ms": ms[k]} for k in ms if k in algo_of}
        dom = max(roof, key=lambda k: roof[k]["avg_launch_ms"])
        dom_kernel = {"shard_compress": "encode_fast_kernel", "shard_decode": "decode_kernel"}.get(dom, dom)
        enc_k, dec_k = ("shard_compress", "shard_decode") if multi else ("encode_fast_kernel", "decode_kernel")
        if strong:
            workload = ("configs[3]: %g GiB synthetic mixed-entropy corpus sharded over %d GPU(s) by block range, one frame "
                        "of 64 KiB independent blocks: fast compress, RCCL all-gather of record sizes, prefix sum, frame "
                        "assembly, decode" % (args.corpus_gib, world))
        elif multi:
            workload = ("configs[3] at 1 GiB per GPU: synthetic mixed-entropy corpus sharded over %d GPU(s) by block range, "
                        "one frame of 64 KiB independent blocks: fast compress, RCCL all-gather of record sizes, prefix "
                        "sum, frame assembly, decode" % world)
        else:
            workload = ("configs[1]: 1 GiB synthetic mixed-entropy corpus per GPU, 64 KiB independent blocks, "
                        "fast compress (acceleration=%d) + strict decompress round trip" % args.acceleration)
        result = {
            "metric": METRIC,
            "value": world * gb * args.steps / wall,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "blocks_per_gpu": nb, "block_bytes": BLOCK, "corpus_bytes_per_gpu": n_bytes,
                "compressed_bytes_rank0": total_c, "ratio": total_c / n_bytes,
                "exchange": "RCCL all-gather of int32 record sizes + prefix sum; all-gather of decoded sizes" if multi
                            else "none",
                **extra,
            },
            "compress_GBps_per_gpu": gb / (ms[enc_k] / 1e3),
            "decompress_GBps_per_gpu": gb / (ms[dec_k] / 1e3),
            "phase_ms": ms,
            "peak_device_bytes": {"torch_allocator": int(torch.cuda.max_memory_allocated(dev)),
                                  "library_scratch": int(codec.L.lz4mi_ctx_device_bytes(codec.ctx.handle)),
                                  "sub_launch_blocks": SUB if multi else nb},
            "roofline": {
                "kernel": dom_kernel, "bound": "hbm", "achieved": roof[dom]["achieved"], "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": roof[dom]["frac"], "traffic": traffic_of(dom_kernel, nb, dom),
                "traffic_source": "profiles/traffic.json (PMC passes of scripts/collect_profiles.sh on this workload)",
                "algorithmic_bytes_per_launch": n_bytes + total_c, "avg_launch_ms": roof[dom]["avg_launch_ms"],
            },
            "roofline_all": roof,
            "reference_published": REFERENCE_PUBLISHED,
        }
        if not multi and not args.no_by_class and nb <= 32768:
            result["pipelined_round_trip"] = pipelined_round_trip(codec, src, nb, stride, args.acceleration, slots, out,
                                                                  args.steps)
        if not multi and not args.no_by_class:
            result["roofline_by_class"] = by_class(codec, src, nb, stride, args.acceleration)
        if world == 1 and not multi and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cb = cpu_baseline(data0, threads)
            result["cpu_baseline"] = cb
            # the oracle works on the first cb["blocks"] blocks (all of them up to 1 GiB per launch)
            gpu_c = int(keep["sizes"][: cb["blocks"]].to(torch.int64).sum())
            result["config"]["ratio_parity"] = (cb["compressed_bytes"] == gpu_c)
            result["config"]["ratio_parity_blocks"] = cb["blocks"]
            result["config"]["nonconformant_blocks_oracle"] = cb["nonconformant_blocks"]
        else:
            result["cpu_baseline"] = None
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
