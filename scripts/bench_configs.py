"""bench.py --config 2 / --config 4: the single-GPU configurations of BASELINE.json besides the headline.

config 2: lz4FrameEncodeWithOptions with 4 MiB blocks + block and content xxHash32 checksums on the 1 GiB
          corpus (device resident), and decode of a frame the lz4 CLI wrote for the same corpus
          (`lz4 -B7 -BX --content-size`, the CLI of the image; our own frame when no CLI is present): host walk of
          the size words, then block checksums + block decode + content checksum on the device.
config 4: LZ4HC (maxSearchDepth 64) block encode with a 32 KiB dictionary + dictionary-primed decode,
          64 KiB blocks.

Same JSON contract as the headline line.  The checks against the oracle run outside the timed region.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile
import time

import numpy as np
import torch

BLOCK = 65536
HBM_PEAK_GBPS = 8000.0


def _events(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


def _mean_ms(pairs):
    return float(np.mean([a.elapsed_time(b) for a, b in pairs]))


def _cli():
    p = shutil.which("lz4") or "/opt/conda/bin/lz4"
    return p if os.path.exists(p) else None


def run_config(args):
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    steps = args.steps
    warmup = args.warmup
    return (config2 if args.config == 2 else config4)(bench, args, dev, steps, warmup)


def config2(bench, args, dev, steps, warmup):
    from lz4_amd import Lz4FrameBlockSize, Lz4FrameOptions
    from lz4_amd.device import DeviceCodec
    from lz4_amd.sharding import frame_block_table
    from oracle import lz4ref
    nb64 = args.blocks
    src, data0 = bench.load_corpus(dev, 0, nb64)
    n = src.numel()
    codec = DeviceCodec(0)
    codec.fence = False
    stream = codec.stream
    opts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k4MB, blockChecksum=True, contentChecksum=True)
    frame_buf = torch.empty(codec.frame_bound(n, opts), dtype=torch.uint8, device=dev)

    # ---- the frame to decode: written by the lz4 CLI for the same bytes
    cli = _cli()
    source = "own frame (no lz4 CLI on this host)"
    cli_frame = None
    if cli:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            pin, pout = os.path.join(td, "in.bin"), os.path.join(td, "out.lz4")
            data0.tofile(pin)
            r = subprocess.run([cli, "-f", "-q", "-B7", "-BX", "--content-size", pin, pout], capture_output=True)
            if r.returncode == 0:
                cli_frame = np.fromfile(pout, dtype=np.uint8)
                ver = subprocess.run([cli, "--version"], capture_output=True, text=True).stdout.strip()
                source = f"`lz4 -B7 -BX --content-size` ({ver})"
    with torch.cuda.stream(stream):
        own = codec.frame_encode(src, opts, out=frame_buf)
    codec.synchronize()
    if cli_frame is None:
        cli_frame = own.cpu().numpy()
    fbytes = cli_frame.tobytes()
    info, offs, end = frame_block_table(fbytes)  # the host walk over the size words: once, before the launches
    assert end == len(fbytes) and info["block_max_size"] == 4 << 20
    dopts = Lz4FrameOptions(blockSize=Lz4FrameBlockSize.k4MB, blockChecksum=info["block_checksum"])
    first = int(offs[0])
    d_records = torch.from_numpy(cli_frame[first:info["body_end"]]).to(dev)
    rel = torch.from_numpy(offs - first).to(dev)
    want_ck = int.from_bytes(fbytes[info["body_end"] + 4:info["body_end"] + 8], "little") if info["content_checksum"] else None
    slots = torch.empty((len(offs), 4 << 20), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    ev = {k: [] for k in ("frame_encode", "frame_decode_blocks", "content_checksum")}
    keep = {}
    # xxHash32 of a whole frame is ONE serial chain (no combine operator): a step holds two of them, the encoder's
    # over the source and the decoder's over its output.  The library hashes device memory of this size on a host
    # thread, from pinned pieces that come over on a side stream (a host core runs the chain several times faster
    # than one GPU lane-chain).  The encoder's runs beside its block launches; the decoder's is called from a
    # worker thread on a second context, so that it runs beside the NEXT step's encode (the caller's pipelining).
    # The decode launch of step k+1 waits for step k's hash to have read the slots it is about to overwrite.
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(1)
    verify = DeviceCodec(0)
    verify.fence = False
    vstream = verify.stream
    pending = None

    def hash_output(e2, record):
        with torch.cuda.stream(vstream):
            vstream.wait_event(e2)
            h0 = _events(vstream)
            ck = verify.xxh32(slots.reshape(-1)[:n])  # xxh32(frameOut)
            h1 = _events(vstream)
        if record:
            ev["content_checksum"].append((h0, h1))
        keep.update(ck=ck)

    def step(record):
        nonlocal pending
        with torch.cuda.stream(stream):
            e0 = _events(stream)
            frame = codec.frame_encode(src, opts, out=frame_buf)  # blocks, block checksums, content checksum, assembly
            e1 = _events(stream)
            if pending is not None:
                pending.result()  # the previous step's hash has read the slots
                pending = None
            _, out_len, status = codec.shard_decode(d_records, rel, dopts, out=slots)  # + block checksum verify
            e2 = _events(stream)
        if want_ck is not None:
            pending = pool.submit(hash_output, e2, record)
        if record:
            ev["frame_encode"].append((e0, e1))
            ev["frame_decode_blocks"].append((e1, e2))
        keep.update(frame=frame, out_len=out_len, status=status)

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    if pending is not None:
        pending.result()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pool.shutdown()

    # ---- checks (not timed)
    assert int(keep["status"].abs().sum()) == 0
    assert int(keep["out_len"].to(torch.int64).sum()) == n  # all blocks full: slots are the output, in place
    assert torch.equal(slots.reshape(-1)[:n], src), "decoded CLI frame != corpus"
    if want_ck is not None:
        assert (int(keep["ck"][0]) & 0xFFFFFFFF) == want_ck, "content checksum of the decoded frame"
    chk = 12 << 20  # three blocks of our frame against the oracle, byte for byte
    ref = lz4ref.frame_encode(data0[:chk].tobytes(), lz4ref.frame_options(block_checksum=True, content_checksum=True))
    got = keep["frame"].cpu().numpy()
    body_ref = ref[7:-8]
    assert got[7:7 + len(body_ref)].tobytes() == body_ref, "frame records differ from the oracle"
    ms = {k: _mean_ms(v) for k, v in ev.items()}
    frame_len = int(keep["frame"].numel())
    gb = n / 1e9
    # the dominant kernel is the single-stream content checksum (xxh32 has no combine operator): n bytes read
    algo = {"frame_encode": (2 * n + frame_len) / 1e9, "frame_decode_blocks": (len(fbytes) + n) / 1e9,
            "content_checksum": n / 1e9}
    roof = {k: {"achieved": algo[k] / (ms[k] / 1e3), "frac": algo[k] / (ms[k] / 1e3) / HBM_PEAK_GBPS,
                "avg_launch_ms": ms[k]} for k in ms}
    result = {
        "metric": bench.METRIC, "value": gb * steps / wall, "unit": "GB/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": "configs[2]: lz4FrameEncodeWithOptions of a %d MiB synthetic mixed-entropy corpus with 4 MiB blocks + "
                        "block and content xxHash32 checksums, and decode of the %s frame of the same corpus; device resident; "
                        "both content checksums on host threads from pinned pieces (one serial xxh32 chain each), the "
                        "decoder's of step k beside step k+1's encode"
                        % (n >> 20, source),
            "blocks": len(offs), "block_bytes": 4 << 20, "frame_bytes": frame_len, "cli_frame_bytes": len(fbytes),
        },
        "encode_GBps": gb / (ms["frame_encode"] / 1e3),
        "decode_GBps": gb / ((ms["frame_decode_blocks"] + ms.get("content_checksum", 0.0)) / 1e3),
        "decode_blocks_only_GBps": gb / (ms["frame_decode_blocks"] / 1e3),
        "phase_ms": ms,
        "roofline": {"kernel": "decode_kernel + xxh32_batch_kernel (256 blocks of 4 MiB, one or two wavefronts each)",
                     "bound": "hbm", "achieved": roof["frame_decode_blocks"]["achieved"],
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": roof["frame_decode_blocks"]["frac"], "traffic": None,
                     "algorithmic_bytes_per_launch": len(fbytes) + n, "avg_launch_ms": ms["frame_decode_blocks"],
                     "note": "the step is bound by the two content checksums (one serial xxh32 stream each, hashed by "
                             "host threads beside the GPU work; phase_ms.content_checksum is the decoder's); of the GPU "
                             "kernels the decode of 256 blocks of 4 MiB is the longest"},
        "roofline_all": roof,
        "reference_published": bench.REFERENCE_PUBLISHED,
    }
    if not args.no_cpu_baseline:
        sample = data0[: 64 << 20].tobytes()
        t0 = time.perf_counter()
        f = lz4ref.frame_encode(sample, lz4ref.frame_options(block_checksum=True, content_checksum=True))
        t1 = time.perf_counter()
        back = lz4ref.frame_decode(f, capacity=len(sample) + 64)
        t2 = time.perf_counter()
        assert back == sample
        result["cpu_baseline"] = {"value": len(sample) / 1e9 / (t2 - t0), "unit": "GB/s", "cores": 1, "kind": "port",
                                  "sample": "first 64 MiB of the corpus, frame encode + decode, 4 MiB blocks, both checksums",
                                  "note": bench.CPU_NOTE, "encode_GBps": len(sample) / 1e9 / (t1 - t0),
                                  "decode_GBps": len(sample) / 1e9 / (t2 - t1)}
    else:
        result["cpu_baseline"] = None
    return result


def config4(bench, args, dev, steps, warmup):
    from lz4_amd.device import DeviceCodec
    from oracle import lz4ref
    from tools.corpus import corpus_block
    nb = args.blocks
    src, data0 = bench.load_corpus(dev, 0, nb)
    n = src.numel()
    dict_host = corpus_block(3)[:32768].copy()  # first 32 KiB of a text-like block; dictId 0x12345678 in a frame
    d_dict = torch.from_numpy(dict_host).to(dev)
    codec = DeviceCodec(0)
    codec.fence = False
    stream = codec.stream
    stride = codec.slot_stride(BLOCK)
    slots = torch.empty((nb, stride), dtype=torch.uint8, device=dev)
    out = torch.empty((nb, BLOCK), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ev = {"encode_hc_kernel": [], "decode_kernel": []}
    keep = {}

    def step(record):
        with torch.cuda.stream(stream):
            e0 = _events(stream)
            _, sizes, st_c = codec.compress_blocks(src, BLOCK, hc_depth=64, dictionary=d_dict, out=slots)
            e1 = _events(stream)
            # lz4BlockDecompressInto behind the dictionary prefix (lz4_frame_decoder.dart:273-290)
            _, out_len, st_d = codec.decompress_blocks(slots, sizes, BLOCK, comp_stride=stride, strict=False,
                                                       dictionary=d_dict, out=out)
            e2 = _events(stream)
            if record:
                ev["encode_hc_kernel"].append((e0, e1))
                ev["decode_kernel"].append((e1, e2))
            keep.update(sizes=sizes, st_c=st_c, st_d=st_d, out_len=out_len)

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    assert int(keep["st_c"].abs().sum()) == 0 and int(keep["st_d"].abs().sum()) == 0
    assert torch.equal(out.reshape(-1), src), "decode(encode(x)) != x"
    h_sizes = keep["sizes"].cpu().numpy()
    db = dict_host.tobytes()
    for b in range(0, min(nb, 256), 5):  # a sample of blocks, byte for byte against the oracle
        ref = lz4ref.block_compress_hc(data0[b * BLOCK:(b + 1) * BLOCK].tobytes(), dictionary=db, max_search_depth=64)
        assert int(h_sizes[b]) == len(ref) and slots[b, : len(ref)].cpu().numpy().tobytes() == ref, b
    total_c = int(h_sizes.astype(np.int64).sum())
    ms = {k: _mean_ms(v) for k, v in ev.items()}
    gb = n / 1e9
    algo = (n + total_c) / 1e9
    roof = {k: {"achieved": algo / (ms[k] / 1e3), "frac": algo / (ms[k] / 1e3) / HBM_PEAK_GBPS, "avg_launch_ms": ms[k]}
            for k in ms}
    dom = max(roof, key=lambda k: roof[k]["avg_launch_ms"])
    result = {
        "metric": bench.METRIC, "value": gb * steps / wall, "unit": "GB/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": "configs[4] on one GPU: LZ4HC (maxSearchDepth=64) block encode with a 32 KiB dictionary + "
                        "dictionary-primed decode, %d MiB synthetic mixed-entropy corpus, 64 KiB blocks" % (n >> 20),
            "blocks_per_gpu": nb, "block_bytes": BLOCK, "compressed_bytes": total_c, "ratio": total_c / n,
            "dictionary_bytes": 32768,
        },
        "compress_GBps_per_gpu": gb / (ms["encode_hc_kernel"] / 1e3),
        "decompress_GBps_per_gpu": gb / (ms["decode_kernel"] / 1e3),
        "phase_ms": ms,
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": roof[dom]["achieved"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": roof[dom]["frac"], "traffic": None,
                     "algorithmic_bytes_per_launch": n + total_c, "avg_launch_ms": roof[dom]["avg_launch_ms"]},
        "roofline_all": roof,
        "reference_published": bench.REFERENCE_PUBLISHED,
    }
    if not args.no_cpu_baseline:
        threads = min(16, os.cpu_count() or 1)
        cb = bench.cpu_baseline(data0[: 512 * BLOCK], threads, hc_depth=64, dictionary=db, label="hc")
        cb["sample"] = "first 32 MiB of the corpus (512 blocks), the same 32 KiB dictionary; " + cb["sample"]
        result["cpu_baseline"] = cb
    else:
        result["cpu_baseline"] = None
    return result
