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


def _pool(fn, n_items, threads):
    step = (n_items + threads * 8 - 1) // (threads * 8)
    spans = [(lo, min(n_items, lo + step)) for lo in range(0, n_items, step)]
    t0 = time.perf_counter()
    if threads == 1:
        for lo, hi in spans:
            fn(lo, hi)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda s: fn(*s), spans))
    return time.perf_counter() - t0


def cpu_baseline(data: np.ndarray, threads: int, hc_depth: int | None = None):
    """The scalar C oracle (bit-identical output) on the host cores: same blocks, same round trip."""
    from oracle import lz4ref
    L = lz4ref.lib()
    import ctypes as C
    nb = data.size // BLOCK
    bound = lz4ref.compress_bound(BLOCK)
    comp = np.empty((nb, bound), dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.int64)
    conf = np.zeros(nb, dtype=np.int64)
    out = np.empty(nb * BLOCK, dtype=np.uint8)
    src_p, comp_p, out_p = data.ctypes.data, comp.ctypes.data, out.ctypes.data

    def enc(lo, hi):
        n = C.c_size_t(0)
        for b in range(lo, hi):
            if hc_depth is None:
                L.lz4ref_block_compress_fast(C.c_char_p(src_p + b * BLOCK), BLOCK, None, 0, 1,
                                             C.c_void_p(comp_p + b * bound), bound, C.byref(n))
            else:
                L.lz4ref_block_compress_hc(C.c_char_p(src_p + b * BLOCK), BLOCK, None, 0, hc_depth,
                                           C.c_void_p(comp_p + b * bound), bound, C.byref(n))
            sizes[b] = n.value

    def dec(lo, hi):
        for b in range(lo, hi):
            L.lz4ref_block_decompress(C.c_char_p(comp_p + b * bound), int(sizes[b]),
                                      C.c_void_p(out_p + b * BLOCK), BLOCK)

    t_enc = _pool(enc, nb, threads)
    t_dec = _pool(dec, nb, threads)
    assert np.array_equal(out, data)
    for b in range(nb):  # not timed: blocks that break the LZ4 end-of-block rules
        conf[b] = 0 if L.lz4ref_block_conformant(C.c_char_p(comp_p + b * bound), int(sizes[b])) else 1
    one = min(nb, 2048 if hc_depth is None else 64)  # single-thread figure on a prefix
    t1 = 0.0
    for fn in (enc, dec):
        t0 = time.perf_counter()
        fn(0, one)
        t1 += time.perf_counter() - t0
    gb = data.size / 1e9
    return {
        "value": gb / (t_enc + t_dec), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"{data.size >> 20} MiB of the corpus, {nb} blocks, compress+decompress round trip, "
                  f"{threads} threads (one block per task)",
        "note": CPU_NOTE,
        "compress_GBps": gb / t_enc, "decompress_GBps": gb / t_dec,
        "value_1thread": one * BLOCK / 1e9 / t1,
        "compressed_bytes": int(sizes.sum()),
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


def traffic_of(kernel):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tpath)).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


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
    multi = world > 1 or os.environ.get("LZ4MI_BENCH_DIST") == "1" or args.config == 3
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

    phases = ("shard_compress", "size_exchange", "shard_write", "shard_decode", "decoded_size_exchange") if multi \
        else ("encode_fast_kernel", "decode_kernel")
    ev = {k: [] for k in phases}
    keep = {}

    def step(record):
        with torch.cuda.stream(stream):
            marks = []

            def mark():
                if record:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    marks.append(e)
            mark()
            if not multi:
                _, sizes, st_c = codec.compress_blocks(src, BLOCK, acceleration=args.acceleration, out=slots)
                mark()
                _, out_len, st_d = codec.decompress_blocks(slots, sizes, BLOCK, comp_stride=stride, strict=True, out=out)
                mark()
                keep.update(sizes=sizes, st_c=st_c, st_d=st_d, out_len=out_len)
            else:
                rec = codec.shard_compress(src, opts)          # compress + record plan (device)
                mark()
                mine, body_end = frame_offsets_uniform(rec, len(header))  # RCCL all-gather + prefix sum
                rel = (mine - mine[0]).contiguous()
                mark()
                codec.shard_write(src, opts, rel, window)      # records into this rank's window of the frame
                mark()
                _, out_len, st_d = codec.shard_decode(window, rel, opts, out=out)
                mark()
                doff, total = decoded_offsets(out_len, uniform=True)  # the format allows short blocks
                if not bool((out_len == BLOCK).all()):         # none here: blocks already sit at b * 64 KiB
                    raise AssertionError("short block in the bench corpus")
                mark()
                keep.update(rec=rec, mine=mine, body_end=body_end, rel=rel, out_len=out_len, st_d=st_d, total=total)
            if record:
                for k, (a, b) in zip(phases, zip(marks, marks[1:])):
                    ev[k].append((a, b))

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
    assert torch.equal(out.reshape(-1), src), "decode(encode(x)) != x"
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
        ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
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
            "roofline": {
                "kernel": dom_kernel, "bound": "hbm", "achieved": roof[dom]["achieved"], "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": roof[dom]["frac"], "traffic": traffic_of(dom_kernel),
                "traffic_source": "profiles/traffic.json (PMC passes of scripts/collect_profiles.sh on this workload)",
                "algorithmic_bytes_per_launch": n_bytes + total_c, "avg_launch_ms": roof[dom]["avg_launch_ms"],
            },
            "roofline_all": roof,
            "reference_published": REFERENCE_PUBLISHED,
        }
        if world == 1 and not multi and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cb = cpu_baseline(data0, threads)
            result["cpu_baseline"] = cb
            result["config"]["ratio_parity"] = (cb["compressed_bytes"] == total_c)
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
