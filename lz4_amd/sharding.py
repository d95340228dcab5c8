"""Block-range sharding of one frame over the ranks of a node.

Independent blocks (blockIndependence, lib/src/frame/lz4_frame_options.dart:80) share nothing,
so rank r encodes the contiguous block range [r*B/G, (r+1)*B/G).  The only exchange the
frame format needs is each block's record size (4 + stored length [+4],
lib/src/frame/lz4_frame_encoder.dart:209-258): one all-gather of int32 words, then every rank
takes an exclusive prefix sum and knows the absolute offset of its records.
Works with any torch.distributed backend (RCCL on GPUs, gloo on CPU tensors).

Decode (lib/src/frame/lz4_frame_decoder.dart:238-304): the host walks the size words once
(`frame_block_table`), rank r decodes the records of its block range into blockMaxSize slots, and
because the format allows short non-final blocks (the reference appends whatever each block yields,
:291-296) the decoded sizes are all-gathered and prefix-summed into output offsets
(`decoded_offsets`) before each rank packs its blocks to their final place.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_blocks: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block range of `rank`."""
    lo = rank * n_blocks // world
    hi = (rank + 1) * n_blocks // world
    return lo, hi


def _gather_counted(local_sizes: torch.Tensor, group=None):
    """(all ranks' sizes in block order, per-rank block counts).  One count exchange, one size exchange."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_sizes, [local_sizes.numel()]
    world = dist.get_world_size(group)
    cnt = torch.empty(world, dtype=torch.int64, device=local_sizes.device)
    dist.all_gather_into_tensor(cnt, torch.tensor([local_sizes.numel()], dtype=torch.int64, device=local_sizes.device),
                                group=group)
    counts = cnt.tolist()  # one read-back: ranks may hold different block counts
    if len(set(counts)) == 1:
        out = torch.empty(world * counts[0], dtype=local_sizes.dtype, device=local_sizes.device)
        dist.all_gather_into_tensor(out, local_sizes.contiguous(), group=group)
        return out, counts
    width = max(counts)
    padded = torch.zeros(width, dtype=local_sizes.dtype, device=local_sizes.device)
    padded[: local_sizes.numel()] = local_sizes
    parts = torch.empty(world * width, dtype=local_sizes.dtype, device=local_sizes.device)
    dist.all_gather_into_tensor(parts, padded, group=group)
    return torch.cat([parts[r * width:r * width + c] for r, c in enumerate(counts)]), counts


def gather_record_sizes(local_sizes: torch.Tensor, group=None) -> torch.Tensor:
    """All ranks' record sizes in block order.  Ranks may hold different counts."""
    return _gather_counted(local_sizes, group)[0]


def frame_offsets(local_sizes: torch.Tensor, header_len: int, group=None):
    """Returns (offsets of this rank's records in the frame [int64], frame body end, all sizes).

    offsets are absolute frame positions; body end is where the end mark goes."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    all_sizes, counts = _gather_counted(local_sizes, group)
    all_sizes = all_sizes.to(torch.int64)
    ends = torch.cumsum(all_sizes, 0) + header_len
    starts = ends - all_sizes
    first = int(sum(counts[:rank]))
    mine = starts[first:first + local_sizes.numel()].contiguous()
    body_end = int(ends[-1].item()) if all_sizes.numel() else header_len
    return mine, body_end, all_sizes


def frame_offsets_uniform(local_sizes: torch.Tensor, header_len: int, group=None):
    """frame_offsets for the case where every rank holds the same number of blocks (the weak-scaling
    layout): one all-gather and one prefix sum, nothing read back to the host.  Returns
    (offsets of this rank's records [int64 tensor], frame body end [0-d int64 tensor])."""
    n = local_sizes.numel()
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        all_sizes, rank = local_sizes.to(torch.int64), 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        gathered = torch.empty(world * n, dtype=local_sizes.dtype, device=local_sizes.device)
        dist.all_gather_into_tensor(gathered, local_sizes.contiguous(), group=group)
        all_sizes = gathered.to(torch.int64)
    ends = torch.cumsum(all_sizes, 0) + header_len
    mine = (ends - all_sizes)[rank * n:(rank + 1) * n]
    return mine, ends[-1] if all_sizes.numel() else torch.tensor(header_len, dtype=torch.int64)


def frame_block_table(frame) -> tuple[dict, np.ndarray, int]:
    """Host walk of one standard frame (no GPU work): (header fields, record offsets [int64], end offset).
    Raises the reference's exception for a frame that is structurally broken."""
    from . import _native as N
    L = N.load()
    buf = bytes(frame) if not isinstance(frame, bytes) else frame
    info = N.FrameInfo()
    n, end = C.c_size_t(0), C.c_size_t(0)
    st = L.lz4mi_frame_block_table(buf, len(buf), C.byref(info), None, 0, C.byref(n), C.byref(end))
    if st:
        N.raise_status(st)
    offs = np.zeros(max(n.value, 1), dtype=np.int64)
    st = L.lz4mi_frame_block_table(buf, len(buf), C.byref(info), offs.ctypes.data_as(C.c_void_p), n.value, C.byref(n),
                                   C.byref(end))
    if st:
        N.raise_status(st)
    fields = {"block_max_size": info.block_max_size, "block_checksum": bool(info.block_checksum),
              "block_independence": bool(info.block_independence), "content_checksum": bool(info.content_checksum),
              "content_size": info.content_size if info.has_content_size else None,
              "dict_id": info.dict_id if info.has_dict_id else None, "header_size": info.header_size,
              # where the last record stops: the end mark, in front of the optional content checksum
              "body_end": end.value - 4 - (4 if info.content_checksum else 0)}
    return fields, offs[: n.value], end.value


def shard_records(record_offsets: np.ndarray, body_end: int, rank: int, world: int):
    """This rank's share of a frame's records: (first block, last block + 1, first byte, last byte + 1).
    `body_end` = offset of the end mark (where the last record stops)."""
    n = len(record_offsets)
    lo, hi = shard_range(n, rank, world)
    first = int(record_offsets[lo]) if lo < n else body_end
    last = int(record_offsets[hi]) if hi < n else body_end
    return lo, hi, first, last


def decoded_offsets(local_out_len: torch.Tensor, group=None, uniform: bool = False):
    """Output offsets of this rank's blocks from every rank's decoded sizes (all-gather + exclusive prefix sum).
    uniform=True: every rank holds the same number of blocks (nothing is read back to the host).
    Returns (offsets [int64], total decoded bytes)."""
    if uniform:
        return frame_offsets_uniform(local_out_len, 0, group)
    mine, total, _ = frame_offsets(local_out_len, 0, group)
    return mine, total
