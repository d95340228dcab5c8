"""Block-range sharding of one frame over the ranks of a node.

Independent blocks (blockIndependence, lib/src/frame/lz4_frame_options.dart:80) share nothing,
so rank r encodes the contiguous block range [r*B/G, (r+1)*B/G).  The only exchange the
frame format needs is each block's record size (4 + stored length [+4],
lib/src/frame/lz4_frame_encoder.dart:209-258): one all-gather of int32 words, then every rank
takes an exclusive prefix sum and knows the absolute offset of its records.
Works with any torch.distributed backend (RCCL on GPUs, gloo on CPU tensors).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_blocks: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block range of `rank`."""
    lo = rank * n_blocks // world
    hi = (rank + 1) * n_blocks // world
    return lo, hi


def gather_record_sizes(local_sizes: torch.Tensor, group=None) -> torch.Tensor:
    """All ranks' record sizes in block order.  Ranks may hold different counts."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_sizes
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=local_sizes.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local_sizes.numel()], dtype=torch.int64, device=local_sizes.device),
                    group=group)
    counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:
        out = torch.empty(world * counts[0], dtype=local_sizes.dtype, device=local_sizes.device)
        dist.all_gather_into_tensor(out, local_sizes.contiguous(), group=group)
        return out
    width = max(counts)
    padded = torch.zeros(width, dtype=local_sizes.dtype, device=local_sizes.device)
    padded[: local_sizes.numel()] = local_sizes
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def frame_offsets(local_sizes: torch.Tensor, header_len: int, group=None):
    """Returns (offsets of this rank's records in the frame [int64], frame body end, all sizes).

    offsets are absolute frame positions; body end is where the end mark goes."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    all_sizes = gather_record_sizes(local_sizes, group).to(torch.int64)
    ends = torch.cumsum(all_sizes, 0) + header_len
    starts = ends - all_sizes
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        counts = [torch.zeros(1, dtype=torch.int64, device=local_sizes.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([local_sizes.numel()], dtype=torch.int64, device=local_sizes.device),
                        group=group)
        first = int(sum(int(c.item()) for c in counts[:rank]))
    else:
        first = 0
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
