"""Multi-rank logic on CPU: torch.distributed with the gloo backend, world_size 2 (and 3).

The codec itself needs a GPU, so the ranks here exchange stand-in "compressed shards" of known,
different sizes: what is under test is the block-range partition, the size all-gather that
places every rank's stream, the variable-size gather and the input scatter - exactly the code
bench.py and multi-GPU callers run over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libhuffman_amd.sharding import (exchange_stream_offsets, gather_stream, scatter_input,
                                     shard_plan, shard_range)


def test_shard_plan_partitions_whole_blocks():
    for n, bs, world in ((16 << 30, 65536, 8), (1 << 30, 65536, 1), (1000, 64, 4), (65536 * 3 + 5, 65536, 8),
                         (10, 0, 2), (0, 65536, 4), (7 * 4096, 4096, 3)):
        plan = shard_plan(n, bs, world)
        assert plan[0][0] == 0 and plan[-1][1] == n
        for (lo, hi), (lo2, _) in zip(plan, plan[1:]):
            assert hi == lo2 and lo <= hi
        step = bs if bs else max(n, 1)
        per = ((n + step - 1) // step + world - 1) // world * step
        nonempty = [(lo, hi) for lo, hi in plan if hi > lo]
        for lo, hi in nonempty:
            assert lo % step == 0                       # shards start on block boundaries
        for lo, hi in nonempty[:-1]:
            assert hi - lo == per                       # equal block counts, short tail on the last


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, bs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)
        full = torch.from_numpy(rng.integers(0, 256, size=n_total, dtype=np.uint8))
        # 1. scatter of the input from rank 0
        mine = scatter_input(full if rank == 0 else None, n_total, bs, src=0)
        lo, hi = shard_range(n_total, bs, rank, world)
        assert torch.equal(mine, full[lo:hi])
        # 2. a stand-in "compressed shard": deterministic function of the shard, size differs per rank
        comp_len = (hi - lo) // (2 + rank) + 13 * rank + 5
        comp = (torch.arange(comp_len, dtype=torch.int64) * 7 + rank).to(torch.uint8)
        padded = torch.cat([comp, torch.zeros(100, dtype=torch.uint8)])     # capacity > length, like encode_bound
        sizes, starts = exchange_stream_offsets(torch.tensor([comp_len]))
        want_sizes = [(shard_range(n_total, bs, r, world)[1] - shard_range(n_total, bs, r, world)[0]) // (2 + r) + 13 * r + 5
                      for r in range(world)]
        assert sizes.tolist() == want_sizes
        assert starts.tolist() == list(np.cumsum([0] + want_sizes[:-1]))
        # 3. gatherv to rank 0 in rank order
        whole = gather_stream(padded, comp_len, dst=0)
        if rank == 0:
            parts = [(torch.arange(s, dtype=torch.int64) * 7 + r).to(torch.uint8) for r, s in enumerate(want_sizes)]
            assert torch.equal(whole, torch.cat(parts))
        else:
            assert whole is None
        # 4. the same movements as ONE variable-size all-to-all each (what bench.py's root placement times)
        from libhuffman_amd.sharding import scatter_from_root, gather_to_root, gatherv_to_root, shard_plan
        in_sizes = [h - l for l, h in shard_plan(n_total, bs, world)]
        mine2 = torch.empty(hi - lo, dtype=torch.uint8)
        scatter_from_root(full if rank == 0 else None, in_sizes, mine2, 0, timeout=60.0)   # (the waited form bench.py uses)
        assert torch.equal(mine2, full[lo:hi])
        whole2, sizes2 = gatherv_to_root(padded, comp_len, 0)
        assert sizes2 == want_sizes
        if rank == 0:
            assert torch.equal(whole2, torch.cat(parts))
        else:
            assert whole2 is None
        back_shard = torch.empty(comp_len, dtype=torch.uint8)
        scatter_from_root(whole2, sizes2, back_shard, 0)                    # compressed shards out again
        assert torch.equal(back_shard, comp)
        result = torch.empty(n_total, dtype=torch.uint8) if rank == 0 else None
        gather_to_root(mine2, in_sizes, result, 0)
        if rank == 0:
            assert torch.equal(result, full)
        q.put((rank, "ok"))
    except Exception as e:            # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,bs", [(2, 10 * 4096 + 77, 4096), (3, 5 * 1000, 1000),
                                              (8, 3 * 512 + 9, 512)])      # 8 ranks, 4 blocks: half of the shards are EMPTY
def test_exchange_over_gloo(world, n_total, bs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_decode_ranges_balanced_by_compressed_bytes():
    from libhuffman_amd.sharding import plan_decode_ranges
    # 8 blocks: the first four compress 4x better than the last four
    sizes = [100, 100, 100, 100, 400, 400, 400, 400]
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    plan = plan_decode_ranges(offs, 2)
    assert plan == [(0, 6), (6, 8)]                    # by bytes: 1000 | 1000, not 4 blocks | 4 blocks
    for world in (1, 2, 3, 5, 8, 16):
        plan = plan_decode_ranges(offs, world)
        assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == 8
        assert all(plan[r][1] == plan[r + 1][0] for r in range(world - 1))      # every block exactly once, in order
        share = [offs[b1] - offs[b0] for b0, b1 in plan]
        assert max(share) <= offs[-1] / world + max(sizes)
    assert plan_decode_ranges([0], 4) == [(0, 0)] * 4
