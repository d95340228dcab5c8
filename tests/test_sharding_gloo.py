"""world_size-2 gloo test of the multi-rank frame assembly logic (CPU, no GPU).

Each rank plays the per-rank compress step with the CPU oracle (test infrastructure standing
in for the device), then runs the product's exchange + prefix-sum code; the records written at
the computed offsets must reproduce the single-process frame byte for byte."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(oracle, data, block, checksum):
    recs = []
    for o in range(0, len(data), block):
        chunk = data[o:o + block]
        c = oracle.block_compress_fast(chunk)
        if len(c) < len(chunk):
            word, payload = len(c), c
        else:
            word, payload = 0x80000000 | len(chunk), chunk
        r = word.to_bytes(4, "little") + payload
        if checksum:
            r += oracle.xxh32(payload).to_bytes(4, "little")
        recs.append(r)
    return recs


def _worker(rank, world, port, uneven, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import lz4ref
        from lz4_amd.sharding import frame_offsets, frame_offsets_uniform, shard_range
        from tools.corpus import corpus
        block = 4096
        n_blocks = 23 if uneven else 24
        data = corpus(n_blocks, block).tobytes()
        header = lz4ref.frame_encode(b"", lz4ref.frame_options(block_size_id=4, block_checksum=True))[:7]
        lo, hi = shard_range(n_blocks, rank, world)
        recs = _records(lz4ref, data[lo * block:hi * block], block, True)
        sizes = torch.tensor([len(r) for r in recs], dtype=torch.int32)
        mine, body_end, all_sizes = frame_offsets(sizes, len(header))
        if not uneven:  # the sync-free form used by the weak-scaling bench must agree
            mine_u, end_u = frame_offsets_uniform(sizes, len(header))
            assert torch.equal(mine_u, mine) and int(end_u) == body_end
        buf = torch.zeros(body_end + 4, dtype=torch.uint8)
        for off, r in zip(mine.tolist(), recs):
            buf[off:off + len(r)] = torch.frombuffer(bytearray(r), dtype=torch.uint8)
        if rank == 0:
            buf[: len(header)] = torch.frombuffer(bytearray(header), dtype=torch.uint8)
        dist.all_reduce(buf)  # disjoint windows: summing == concatenating
        if rank == 0:
            ok = bytes(buf.numpy()) == _single_frame(lz4ref, data, block)
            q.put((ok, int(all_sizes.numel())))
    finally:
        dist.destroy_process_group()


def _single_frame(oracle, data, block):
    # 64 KiB is the smallest real frame block size; emulate smaller blocks with the same rules
    header = oracle.frame_encode(b"", oracle.frame_options(block_size_id=4, block_checksum=True))[:7]
    return header + b"".join(_records(oracle, data, block, True)) + bytes(4)


@pytest.mark.parametrize("uneven", [False, True])
def test_two_rank_frame_assembly_gloo(uneven):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, uneven, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, n = q.get(timeout=10)
    assert ok and n == (23 if uneven else 24)


def test_shard_range_partitions_everything():
    from lz4_amd.sharding import shard_range
    for n in (0, 1, 7, 16384, 1048576):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
