"""Block sharding of one logical input over the GPUs of a node.

Blocks are independent (src/encoder.c:288-374 resets all state between blocks), so rank r of
G simply owns a contiguous range of blocks and no data-path collective is needed.  What the
ranks do exchange is tiny: the compressed size of every shard, which places each rank's
stream inside the job's single libhuffman stream (stream of rank r starts at the sum of the
sizes of ranks < r).  Moving whole shards (scatter of the input from a root, gatherv of the
compressed shards to a root) is offered for callers whose data starts or ends on one GPU; RCCL
has no gatherv, so it is grouped send/recv of exactly-sized buffers.

Everything here works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, blocksize: int, rank: int, world: int) -> Tuple[int, int]:
    """Byte range [lo, hi) of the input owned by `rank`: ceil(nblocks/world) whole blocks per
    rank, the last non-empty rank takes the short tail block."""
    if n_total <= 0:
        return 0, 0
    bs = blocksize if blocksize else n_total
    nblocks = (n_total + bs - 1) // bs
    per = (nblocks + world - 1) // world
    b0 = min(rank * per, nblocks)
    b1 = min(b0 + per, nblocks)
    return min(b0 * bs, n_total), min(b1 * bs, n_total)


def exchange_stream_offsets(local_len: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the shard sizes (a 1-element int64 tensor per rank, on the compute device).

    Returns (sizes[world], starts[world]) with starts = exclusive prefix sum, both on the same
    device as `local_len`; no host synchronisation."""
    world = dist.get_world_size(group)
    sizes = torch.empty(world, dtype=torch.int64, device=local_len.device)
    dist.all_gather_into_tensor(sizes, local_len.reshape(1).to(torch.int64), group=group)
    starts = torch.cumsum(sizes, 0) - sizes
    return sizes, starts


def gather_stream(local: torch.Tensor, local_len: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Variable-size gather of the compressed shards to `dst` (rank order = stream order)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes, starts = exchange_stream_offsets(torch.tensor([local_len], dtype=torch.int64, device=local.device), group)
    sizes_h, starts_h = sizes.tolist(), starts.tolist()
    if rank == dst:
        out = torch.empty(int(sum(sizes_h)), dtype=torch.uint8, device=local.device)
        out[starts_h[dst]: starts_h[dst] + sizes_h[dst]] = local[:local_len]
        reqs = []
        for r in range(world):
            if r != dst and sizes_h[r]:
                reqs.append(dist.irecv(out[starts_h[r]: starts_h[r] + sizes_h[r]], src=r, group=group))
        for q in reqs:
            q.wait()
        return out
    if local_len:
        dist.send(local[:local_len].contiguous(), dst=dst, group=group)
    return None


def scatter_input(data: Optional[torch.Tensor], n_total: int, blocksize: int, src: int = 0,
                  device=None, group=None) -> torch.Tensor:
    """Root `src` holds the whole input; every rank receives its block range."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_total, blocksize, rank, world)
    if rank == src:
        reqs = []
        for r in range(world):
            rlo, rhi = shard_range(n_total, blocksize, r, world)
            if r != src and rhi > rlo:
                reqs.append(dist.isend(data[rlo:rhi].contiguous(), dst=r, group=group))
        mine = data[lo:hi].clone()
        for q in reqs:
            q.wait()
        return mine
    mine = torch.empty(hi - lo, dtype=torch.uint8, device=device)
    if hi > lo:
        dist.recv(mine, src=src, group=group)
    return mine


def shard_plan(n_total: int, blocksize: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_total, blocksize, r, world) for r in range(world)]
