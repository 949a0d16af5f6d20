"""One rank of tests/test_gpu_sharded.py's multi-rank run: hufgpu_encode_sharded / hufgpu_decode_sharded through
libhuffman_amd.sharding.ShardGroup, the ranks of one run sharing the box's GPU (the RCCL entry points are
tests/mock_rccl's, HUF_GPU_RCCL_LIB; RCCL itself refuses two ranks on one device).
usage: sharded_worker.py <rank> <nranks> <directory>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec, HuffmanGpuError
from libhuffman_amd.sharding import ShardGroup, shard_range

rank, nranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
codec = GpuCodec(0)
id_path = os.path.join(d, "id")
if rank == 0:
    import ctypes as C
    buf = C.create_string_buffer(128)
    assert codec.lib.hufgpu_shard_unique_id(buf) == 0, codec.lib.hufgpu_shard_last_error(None)
    with open(id_path + ".tmp", "wb") as f:
        f.write(buf.raw)
    os.replace(id_path + ".tmp", id_path)
t0 = time.time()
while not os.path.exists(id_path):
    assert time.time() - t0 < 60, "rank 0 never wrote the id"
    time.sleep(0.01)
group = ShardGroup(codec, id_bytes=open(id_path, "rb").read(), nranks=nranks, rank=rank)

if len(sys.argv) > 4 and sys.argv[4] == "absent":
    # The last rank never makes the call.  Every other rank must come back with HUF_ERROR_FATAL within the deadline (and
    # a little: the helper's host call is inside the transport, the abort comes from the calling thread), the group is
    # broken afterwards - the next call fails at once - and can still be destroyed.
    if rank == nranks - 1:
        time.sleep(8.0)
        print("DONE (never arrived)")
        sys.exit(0)
    group.set_timeout(1500)
    n_total, bs = 3 * (1 << 20), 65536
    data = stream = None
    if rank == 0:
        data = torch.from_numpy(datagen.zipf255(n_total)).cuda()
        stream = torch.zeros(codec.encode_bound(n_total, bs) + 8, dtype=torch.uint8, device="cuda")
    t0 = time.time()
    try:
        group.encode(data, n_total, bs, stream)
        raise SystemExit("the call came back although a rank never arrived")
    except HuffmanGpuError as e:
        took = time.time() - t0
        assert e.err == 4, e                              # HUF_ERROR_FATAL
        assert 1.4 <= took <= 6.0, took
        assert "timed out" in str(e) or "broken" in str(e), e
    t0 = time.time()
    for call in (lambda: group.encode(data, n_total, bs, stream), lambda: group.decode(stream, 10, n_total, bs, None, own_layout=True)):
        try:
            call()
            raise SystemExit("a broken group took a call")
        except HuffmanGpuError as e:
            assert e.err == 4 and "broken" in str(e), e
    assert time.time() - t0 < 0.5
    group.close()
    # the codec itself is untouched by all this
    if rank == 0:
        out, offs, length = codec.encode(data, bs)
        back = torch.empty(n_total, dtype=torch.uint8, device="cuda")
        assert codec.decode(out, length, offs, codec.block_count(n_total, bs), back) == n_total and torch.equal(back, data)
    print("DONE (returned HUF_ERROR_FATAL after %.2f s)" % took)
    sys.exit(0)

CASES = [  # n_total, blocksize, root, workload
    (5 * (1 << 20) + 17, 65536, 0, "zipf255"),
    (5 * (1 << 20) + 17, 65536, 1, "uniform256"),
    (100, 65536, 0, "zipf255"),                 # one short block: two ranks have nothing
    (0, 65536, 0, "zipf255"),
    (3 * 65536, 65536, nranks - 1, "zipf255"),  # a block a rank exactly (three ranks)
    ((1 << 20) + 5, 0, 0, "zipf255"),           # blocksize 0: one block of everything
    (40 * (1 << 20), 1 << 20, 0, "logtext"),
]
for n_total, bs, root, wl in CASES:
    relaxed = wl == "uniform256"
    me_root = rank == root
    nblocks = codec.block_count(n_total, bs)
    data = stream = index = out = None
    if me_root:
        host = getattr(datagen, wl)(n_total) if n_total else np.zeros(0, dtype=np.uint8)
        data = torch.from_numpy(host).cuda()
        stream = torch.zeros(codec.encode_bound(n_total, bs) + 8, dtype=torch.uint8, device="cuda")
        index = torch.zeros(nblocks + 1, dtype=torch.int64, device="cuda")
        out = torch.zeros(max(n_total, 1), dtype=torch.uint8, device="cuda")
    total, lens, legs = group.encode(data, n_total, bs, stream, root=root, index=index, with_index=True, legs=True)
    assert total == sum(lens) and len(lens) == nranks and len(legs) == 4
    for r in range(nranks):                       # a rank without blocks has an empty shard
        lo, hi = shard_range(n_total, bs, r, nranks)
        assert (lens[r] == 0) == (hi == lo), (r, lens, lo, hi)
    if me_root:
        # the stream and the block index of ONE encode of the whole input on one GPU (that one is held against the oracle
        # by tests/test_gpu_parity.py; small inputs here as well)
        want, woffs, wlen = codec.encode(data, bs) if n_total else (stream[:0], torch.zeros(1, dtype=torch.int64, device="cuda"), 0)
        assert total == wlen, (total, wlen)
        assert torch.equal(stream[:total], want[:wlen]), "the gathered stream differs from one GPU's"
        assert torch.equal(index, woffs[: nblocks + 1].to(torch.int64)), "the gathered block index differs from one GPU's"
        if 0 < n_total <= (6 << 20):
            from oracle.oracle import Oracle
            assert np.array_equal(stream[:total].cpu().numpy(), Oracle().encode(host, bs)), "the gathered stream differs from the oracle's"
    # decode with the layout the encode left behind (every rank's own block index and sub-index) ...
    got = group.decode(stream, total, n_total, bs, out, root=root, own_layout=True, relaxed=relaxed)
    assert got == n_total, (got, n_total)
    if me_root:
        assert torch.equal(out[:n_total], data), "own layout: the output differs"
        out.zero_()
    # ... and as a foreign stream: cut by compressed bytes from the root's block index
    got, legs = group.decode(stream, total, n_total, bs, out, root=root, index=index, relaxed=relaxed, legs=True)
    assert got == n_total and len(legs) == 4
    if me_root:
        assert torch.equal(out[:n_total], data), "foreign stream: the output differs"
    # the foreign decode took the buffers the encode's layout lived in: an own-layout decode now says so on every rank (round 5
    # decoded with a stale block index), and after the next encode it works again
    try:
        group.decode(stream, total, n_total, bs, out, root=root, own_layout=True, relaxed=relaxed)
        raise AssertionError("an own-layout decode after a foreign one went through")
    except HuffmanGpuError as e:
        assert e.err == 2, e                                 # HUF_ERROR_INVALID_ARGUMENT
    total2, lens2 = group.encode(data, n_total, bs, stream, root=root)
    assert (total2, lens2) == (total, lens)
    if me_root:
        out.zero_()
    assert group.decode(stream, total, n_total, bs, out, root=root, own_layout=True, relaxed=relaxed) == n_total
    if me_root:
        assert torch.equal(out[:n_total], data), "own layout after a foreign decode and a new encode: the output differs"
    # a damaged payload: every rank returns the same error
    if n_total >= (1 << 20) and not relaxed:
        if me_root:
            bad = stream.clone()
            pos = int(index[nblocks // 2].item()) + 8          # the tree length of a block in the middle: > 1024
            bad[pos] = 0xff
            bad[pos + 1] = 0x7f
        else:
            bad = None
        try:
            group.decode(bad, total, n_total, bs, out, root=root, index=index)
            raise AssertionError("a damaged stream decoded")
        except HuffmanGpuError as e:
            assert e.err == 5, e                                 # HUF_ERROR_BTREE_OVERFLOW, on every rank
    print("rank", rank, "case", (n_total, bs, root, wl), "ok", flush=True)
group.close()
print("rank", rank, "DONE", flush=True)
