#!/usr/bin/env python3
"""Headline benchmark: fast compress + decompress of 64 KiB independent blocks.

Workload (BASELINE.json configs[1]): a 1 GiB synthetic mixed-entropy corpus per GPU
(tools/corpus.c), resident in HBM before the timed region.  One step = one pass of the hot
path over the corpus: the batched fast block encoder, then the batched strict block decoder
on the bytes it produced.  With N > 1 ranks every rank holds its own 1 GiB shard (weak
scaling) and the step additionally all-gathers the per-block frame record sizes over RCCL
and prefix-sums them (the one exchange the frame format needs).

value = uncompressed bytes through the round trip per second, summed over ranks:
        N * corpus_bytes * steps / wall, wall = max over ranks between two barrier+sync pairs.
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


def cpu_baseline(data: np.ndarray, threads: int):
    """The scalar C oracle (bit-identical output) on the host cores: same corpus, same round trip."""
    from oracle import lz4ref
    L = lz4ref.lib()
    import ctypes as C
    nb = data.size // BLOCK
    bound = lz4ref.compress_bound(BLOCK)
    comp = np.empty((nb, bound), dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.int64)
    out = np.empty(nb * BLOCK, dtype=np.uint8)
    src_p, comp_p, out_p = data.ctypes.data, comp.ctypes.data, out.ctypes.data

    def enc(lo, hi):
        n = C.c_size_t(0)
        for b in range(lo, hi):
            L.lz4ref_block_compress_fast(C.c_char_p(src_p + b * BLOCK), BLOCK, None, 0, 1,
                                         C.c_void_p(comp_p + b * bound), bound, C.byref(n))
            sizes[b] = n.value

    def dec(lo, hi):
        for b in range(lo, hi):
            L.lz4ref_block_decompress(C.c_char_p(comp_p + b * bound), int(sizes[b]),
                                      C.c_void_p(out_p + b * BLOCK), BLOCK)

    def run(fn, nthreads):
        step = (nb + nthreads * 8 - 1) // (nthreads * 8)
        spans = [(lo, min(nb, lo + step)) for lo in range(0, nb, step)]
        t0 = time.perf_counter()
        if nthreads == 1:
            for lo, hi in spans:
                fn(lo, hi)
        else:
            with ThreadPoolExecutor(nthreads) as ex:
                list(ex.map(lambda s: fn(*s), spans))
        return time.perf_counter() - t0

    t_enc = run(enc, threads)
    t_dec = run(dec, threads)
    assert np.array_equal(out, data)
    one = min(nb, 2048)  # single-thread figure on a 128 MiB prefix
    t1 = 0.0
    for fn in (enc, dec):
        t0 = time.perf_counter()
        fn(0, one)
        t1 += time.perf_counter() - t0
    gb = data.size / 1e9
    return {
        "value": gb / (t_enc + t_dec), "unit": "GB/s", "cores": threads, "kind": "port",
        "sample": f"full {data.size >> 20} MiB corpus, {nb} blocks, compress+decompress round trip, "
                  f"{threads} threads (one block per task)",
        "compress_GBps": gb / t_enc, "decompress_GBps": gb / t_dec,
        "value_1thread": one * BLOCK / 1e9 / t1,
        "compressed_bytes": int(sizes.sum()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=16384, help="64 KiB blocks per GPU (16384 = 1 GiB)")
    ap.add_argument("--acceleration", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LZ4MI_BENCH_DIST=1 takes the multi-rank code path (RCCL init, size exchange) with one rank,
    # to rehearse it on a one-GPU box under torch.distributed.run
    multi = world > 1 or os.environ.get("LZ4MI_BENCH_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    from lz4_amd.device import DeviceCodec
    from lz4_amd.sharding import frame_offsets_uniform
    from tools.corpus import corpus

    nb = args.blocks
    data = corpus(nb, BLOCK, first=rank * nb)
    src = torch.from_numpy(data).to(dev)
    codec = DeviceCodec(local_rank)
    codec.fence = False  # everything below is issued on the codec's stream
    stream = codec.stream
    stride = codec.slot_stride(BLOCK)
    slots = torch.empty((nb, stride), dtype=torch.uint8, device=dev)
    out = torch.empty((nb, BLOCK), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    ev = {k: [] for k in ("enc", "dec")}

    def step(record):
        with torch.cuda.stream(stream):
            if record:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(stream)
            _, sizes, st_c = codec.compress_blocks(src, BLOCK, acceleration=args.acceleration, out=slots)
            if record:
                e1.record(stream)
            if multi:  # record sizes of every rank -> absolute frame offsets of this rank's blocks
                rec = torch.where(sizes < BLOCK, sizes, torch.full_like(sizes, BLOCK)) + 4
                frame_offsets_uniform(rec, 7)
            _, out_len, st_d = codec.decompress_blocks(slots, sizes, BLOCK, comp_stride=stride, strict=True, out=out)
            if record:
                e2.record(stream)
                ev["enc"].append((e0, e1))
                ev["dec"].append((e1, e2))
        return sizes, st_c, st_d

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sizes, st_c, st_d = step(True)
    barrier()
    wall = time.perf_counter() - t0
    if multi:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- correctness of what was timed (outside the timed region)
    assert int(st_c.abs().sum()) == 0 and int(st_d.abs().sum()) == 0, "block status != OK"
    assert torch.equal(out.reshape(-1), src), "decode(encode(x)) != x"
    total_c = int(sizes.to(torch.int64).sum())

    if rank == 0:
        enc_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["enc"]]))
        dec_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["dec"]]))
        n_bytes = nb * BLOCK
        gb = n_bytes / 1e9
        algo = (n_bytes + total_c) / 1e9  # algorithmic bytes of one launch: read n write c / read c write n
        kernels = {
            "encode_fast_kernel": {"ms": enc_ms, "GBps_uncompressed": gb / (enc_ms / 1e3),
                                   "achieved_GBps": algo / (enc_ms / 1e3)},
            "decode_kernel": {"ms": dec_ms, "GBps_uncompressed": gb / (dec_ms / 1e3),
                              "achieved_GBps": algo / (dec_ms / 1e3)},
        }
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "compress/decompress GB/s per GPU & node, 64KiB indep. blocks; ratio parity",
            "value": world * gb * args.steps / wall,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 1 GiB synthetic mixed-entropy corpus per GPU, 64 KiB independent blocks, "
                            "fast compress (acceleration=%d) + strict decompress round trip" % args.acceleration,
                "blocks_per_gpu": nb, "block_bytes": BLOCK, "corpus_bytes_per_gpu": n_bytes,
                "compressed_bytes_rank0": total_c, "ratio": total_c / n_bytes,
                "exchange": "RCCL all-gather of int32 record sizes + prefix sum" if multi else "none",
            },
            "compress_GBps_per_gpu": kernels["encode_fast_kernel"]["GBps_uncompressed"],
            "decompress_GBps_per_gpu": kernels["decode_kernel"]["GBps_uncompressed"],
            "roofline": {
                "kernel": dom, "bound": "hbm", "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": kernels[dom]["achieved_GBps"] / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": n_bytes + total_c, "avg_launch_ms": kernels[dom]["ms"],
            },
            "roofline_all": {k: {"achieved": v["achieved_GBps"], "frac": v["achieved_GBps"] / HBM_PEAK_GBPS,
                                 "avg_launch_ms": v["ms"]} for k, v in kernels.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cb = cpu_baseline(data, threads)
            result["cpu_baseline"] = cb
            result["config"]["ratio_parity"] = (cb["compressed_bytes"] == total_c)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
