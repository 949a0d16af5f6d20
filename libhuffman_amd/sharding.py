"""Block sharding of one logical input over the GPUs of a node.

Blocks are independent (src/encoder.c:288-374 resets all state between blocks), so rank r of
G simply owns a contiguous range of blocks and no data-path collective is needed.  What the
ranks do exchange is tiny: the compressed size of every shard, which places each rank's
stream inside the job's single libhuffman stream (stream of rank r starts at the sum of the
sizes of ranks < r).  Moving whole shards (scatter of the input from a root, gatherv of the
compressed shards to a root) is offered for callers whose data starts or ends on one GPU; RCCL
has no gatherv, so it is grouped send/recv of exactly-sized buffers.

The functions of this module work with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.  On the GPUs the movements themselves are the C library's
(`ShardGroup` = hufgpu_encode_sharded / hufgpu_decode_sharded, include/huffman_gpu.h: grouped
ncclSend / ncclRecv to computed offsets and the size all-gather, on a communicator of its own); the
range and plan arithmetic here and there is the same (`shard_range` = hufgpu_shard_range,
`plan_decode_ranges` = hufgpu_shard_plan_decode; tests/test_sharding.py holds them against each other).
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


SHARD_INDEX, SHARD_OWN_LAYOUT = 0x100, 0x200          # include/huffman_gpu.h


class ShardGroup:
    """The ranks of one node as the C library sees them: a communicator of the library's own (its
    unique id travels through the torch.distributed group that is there anyway, as 128 bytes) and
    this rank's scratch shards.  `codec` = this rank's GpuCodec; every rank of `group` makes one."""

    def __init__(self, codec, group=None, id_bytes: Optional[bytes] = None, nranks: Optional[int] = None,
                 rank: Optional[int] = None):
        import ctypes as C
        self._C, self.codec, self.lib = C, codec, codec.lib
        self._sh = C.c_void_p()
        if id_bytes is None:                              # through torch.distributed: rank 0 makes the id
            rank, nranks = dist.get_rank(group), dist.get_world_size(group)
            box = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                err = self.lib.hufgpu_shard_unique_id(buf)
                box[0] = (err, buf.raw, self.lib.hufgpu_shard_last_error(None).decode())
            # (src is a GLOBAL rank: the group's first member, which need not be rank 0 of the world)
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            err, id_bytes, why = box[0]
            if err:
                raise RuntimeError("RCCL is not available to the C library: %s" % why)
        self.nranks, self.rank = int(nranks), int(rank)
        err = self.lib.hufgpu_shard_create(C.byref(self._sh), codec._ctx, None, id_bytes, self.nranks, self.rank)
        if err:
            raise RuntimeError("hufgpu_shard_create failed (%d): %s" % (err, self.lib.hufgpu_shard_last_error(None).decode()))

    def set_timeout(self, ms: int):
        """deadline of every sharded call in milliseconds (0 = none; default HUF_GPU_SHARD_TIMEOUT_MS or 120 000)"""
        self._check(self.lib.hufgpu_shard_set_timeout(self._sh, int(ms)), "hufgpu_shard_set_timeout")

    def close(self):
        if self._sh:
            self.lib.hufgpu_shard_destroy(self._sh)
            self._sh = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, err: int, what: str):
        if err:
            from .codec import HuffmanGpuError
            raise HuffmanGpuError(err, what, self.lib.hufgpu_shard_last_error(self._sh).decode())

    @staticmethod
    def _ptr(t):
        return None if t is None else t.data_ptr()

    def encode(self, data: Optional[torch.Tensor], n_total: int, blocksize: int, stream: Optional[torch.Tensor],
               root: int = 0, index: Optional[torch.Tensor] = None, with_index: bool = False, legs: bool = False):
        """data / stream / index: the root's tensors (None elsewhere); with_index (the same on every rank): the
        root's `index` (int64, block count + 1) receives the block index of the whole stream.  Returns (stream
        length, shard lengths[, legs in ms]) on every rank."""
        C = self._C
        total = C.c_uint64()
        lens = (C.c_uint64 * self.nranks)()
        ms = (C.c_double * 4)() if legs else None
        torch.cuda.current_stream(self.codec.tdev).synchronize()         # (the call works on its own stream)
        err = self.lib.hufgpu_encode_sharded(self._sh, root, self._ptr(data), n_total, blocksize,
                                             SHARD_INDEX if with_index else 0,
                                             self._ptr(stream), stream.numel() if stream is not None else 0,
                                             self._ptr(index), C.byref(total), lens, ms)
        self._check(err, "hufgpu_encode_sharded")
        return (int(total.value), [int(x) for x in lens]) + ((list(ms),) if legs else ())

    def decode(self, stream: Optional[torch.Tensor], stream_len: int, n_total: int, blocksize: int,
               out: Optional[torch.Tensor], root: int = 0, index: Optional[torch.Tensor] = None,
               own_layout: bool = False, relaxed: bool = False, legs: bool = False):
        """own_layout (the same on every rank): the stream is the one this object's last encode made - every rank
        kept its block index and sub-index; else `index` = the root's block index of the stream.  Returns bytes
        decoded[, legs in ms]."""
        C = self._C
        raw = C.c_uint64()
        ms = (C.c_double * 4)() if legs else None
        flags = (1 if relaxed else 0) | (SHARD_OWN_LAYOUT if own_layout else 0)
        torch.cuda.current_stream(self.codec.tdev).synchronize()
        err = self.lib.hufgpu_decode_sharded(self._sh, root, self._ptr(stream), stream_len, self._ptr(index), n_total,
                                             blocksize, flags, self._ptr(out), out.numel() if out is not None else 0,
                                             C.byref(raw), ms)
        self._check(err, "hufgpu_decode_sharded")
        return (int(raw.value), list(ms)) if legs else int(raw.value)


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
