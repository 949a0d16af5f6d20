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

from datetime import timedelta
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def _all_to_all(out, inp, out_split, in_split, group, timeout: Optional[float]):
    """all_to_all_single; with `timeout` (seconds) the call is asynchronous and waited for that long -
    a peer that never arrives raises here instead of holding the caller for ever."""
    if timeout is None:
        dist.all_to_all_single(out, inp, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        return
    work = dist.all_to_all_single(out, inp, output_split_sizes=out_split, input_split_sizes=in_split, group=group,
                                  async_op=True)
    try:
        work.wait(timeout=timedelta(seconds=timeout))          # (the backends raise when the time is up; they do not return False)
    except RuntimeError as e:                                  # torch.distributed.DistBackendError is a RuntimeError
        text = str(e).lower()
        if "timeout" in text or "timed out" in text:
            raise TimeoutError("variable-size all-to-all did not complete in %.0f s: %s" % (timeout, e)) from e
        raise                                                  # (an invalid split, a peer's abort, out of memory: as the backend says it)


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


# ---- whole shards between a root and the ranks: ONE variable-size all-to-all per movement ----------
# RCCL has neither scatterv nor gatherv; an all-to-all with split sizes is the grouped send/recv it
# would take to write one, issued as one collective (no per-peer ordering to get wrong, one launch).
# The split sizes are host integers, so every movement has one host synchronisation in front of it
# (the sizes of compressed shards come from the size all-gather).

def scatter_from_root(full: Optional[torch.Tensor], sizes: List[int], mine: torch.Tensor, src: int = 0,
                      group=None, timeout: Optional[float] = None) -> torch.Tensor:
    """Rank `src` holds the concatenation of all shards (`sizes` bytes each, rank order); every
    rank receives its shard into `mine` (sizes[rank] bytes)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert len(sizes) == world and mine.numel() == sizes[rank]
    if rank == src:
        send, in_split = full[: sum(sizes)], list(sizes)
    else:
        send, in_split = mine.new_empty(0), [0] * world
    out_split = [sizes[rank] if r == src else 0 for r in range(world)]
    _all_to_all(mine, send, out_split, in_split, group, timeout)
    return mine


def gather_to_root(local: torch.Tensor, sizes: List[int], out: Optional[torch.Tensor], dst: int = 0,
                   group=None, timeout: Optional[float] = None) -> Optional[torch.Tensor]:
    """Inverse of scatter_from_root with sizes known everywhere: rank r's `local` (sizes[r] bytes)
    lands at offset sum(sizes[:r]) of `out` on rank `dst`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert len(sizes) == world and local.numel() == sizes[rank]
    in_split = [sizes[rank] if r == dst else 0 for r in range(world)]
    if rank == dst:
        recv, out_split = out[: sum(sizes)], list(sizes)
    else:
        recv, out_split = local.new_empty(0), [0] * world
    _all_to_all(recv, local, out_split, in_split, group, timeout)
    return out if rank == dst else None


def gatherv_to_root(local: torch.Tensor, local_len: int, dst: int = 0, group=None, timeout: Optional[float] = None):
    """Variable-size gather of the compressed shards: sizes are exchanged first (all-gather of one
    int64 per rank), then the shards travel.  Returns (stream on `dst` or None, sizes list)."""
    sizes, _ = exchange_stream_offsets(torch.tensor([local_len], dtype=torch.int64, device=local.device), group)
    sizes_h = [int(x) for x in sizes.tolist()]
    rank = dist.get_rank(group)
    out = torch.empty(sum(sizes_h), dtype=torch.uint8, device=local.device) if rank == dst else None
    gather_to_root(local[:local_len], sizes_h, out, dst, group, timeout)
    return out, sizes_h


def plan_decode_ranges(block_offsets, world: int) -> List[Tuple[int, int]]:
    """Contiguous block ranges [b0, b1) per rank for DECODING a gathered stream, balanced by
    compressed bytes: `block_offsets` = the nblocks+1 header offsets of the job's block
    index (last = stream length).  Rank r gets the blocks whose header offset lies in its 1/world
    share of the stream; every block goes to exactly one rank, ranks may be empty."""
    offs = [int(x) for x in block_offsets]
    nblocks = len(offs) - 1
    if nblocks <= 0:
        return [(0, 0)] * world
    total = offs[-1]
    import bisect
    cuts = [0]
    for r in range(1, world):
        # first block whose header lies at or behind the r-th share boundary
        cuts.append(max(cuts[-1], bisect.bisect_left(offs, (total * r + world - 1) // world, 0, nblocks)))
    cuts.append(nblocks)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]
