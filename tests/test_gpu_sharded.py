"""hufgpu_encode_sharded / hufgpu_decode_sharded (include/huffman_gpu.h; blocks are independent:
/root/reference src/encoder.c:288-374): the RCCL scatter / gather of block buffers behind the C ABI.
  * one rank, the real RCCL: the entry points, the communicator, empty groups, one-rank all-gathers
  * three ranks on the box's one GPU over tests/mock_rccl (RCCL refuses two ranks on a device): who sends what to whom
    at which offset - the gathered stream is one GPU's (and, small, the oracle's), both decode forms give the input back,
    a damaged stream fails alike on every rank
  * (not gpu) the host arithmetic against libhuffman_amd.sharding's."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_range_and_plan_arithmetic_against_sharding_py():
    import ctypes as C
    from libhuffman_amd import _native
    from libhuffman_amd.sharding import plan_decode_ranges, shard_range
    L = _native.load()
    rng = np.random.default_rng(7)
    for _ in range(300):
        world = int(rng.integers(1, 9))
        bs = int(rng.choice([0, 1, 7, 4096, 65536, 1 << 20]))
        n = int(rng.choice([0, 1, 5, 65535, 65536, 65537, int(rng.integers(0, 1 << 33))]))
        for r in range(world):
            lo, hi = C.c_uint64(), C.c_uint64()
            assert L.hufgpu_shard_range(n, bs, r, world, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == shard_range(n, bs, r, world), (n, bs, r, world)
        nb = int(rng.integers(0, 200))
        sizes = rng.integers(19, 100000, size=nb)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        first = (C.c_uint64 * (world + 1))()
        assert L.hufgpu_shard_plan_decode(offs.ctypes.data_as(C.POINTER(C.c_uint64)), nb, world, first) == 0
        want = plan_decode_ranges(offs.tolist(), world)
        assert [(int(first[r]), int(first[r + 1])) for r in range(world)] == want, (nb, world)
    assert L.hufgpu_shard_range(10, 4, 3, 3, None, None) != 0 and L.hufgpu_shard_create(None, None, None, None, 1, 0) != 0


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.mark.gpu
def test_one_rank_through_the_real_rccl(torch_mod):
    torch = torch_mod
    from libhuffman_amd import datagen
    from libhuffman_amd.codec import GpuCodec
    from libhuffman_amd.sharding import ShardGroup
    from oracle.oracle import Oracle
    import ctypes as C
    codec = GpuCodec(0)
    buf = C.create_string_buffer(128)
    assert codec.lib.hufgpu_shard_unique_id(buf) == 0, codec.lib.hufgpu_shard_last_error(None)
    group = ShardGroup(codec, id_bytes=buf.raw, nranks=1, rank=0)
    for n, bs, wl in ((3 * (1 << 20) + 17, 65536, "zipf255"), (0, 65536, "zipf255"), (10, 65536, "zipf255"),
                      (64 << 20, 65536, "uniform256"), ((1 << 20) + 1, 0, "zipf255")):
        host = getattr(datagen, wl)(n) if n else np.zeros(0, dtype=np.uint8)
        data = torch.from_numpy(host.copy()).cuda()
        nb = codec.block_count(n, bs)
        stream = torch.zeros(codec.encode_bound(n, bs) + 8, dtype=torch.uint8, device="cuda")
        index = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
        total, lens = group.encode(data, n, bs, stream, index=index, with_index=True)
        assert lens == [total]
        if n <= (4 << 20):
            want = Oracle().encode(host, bs)
            assert total == want.size and np.array_equal(stream[:total].cpu().numpy(), want), (n, bs)
        else:
            want, woffs, wlen = codec.encode(data, bs)
            assert total == wlen and torch.equal(stream[:total], want[:wlen]) and torch.equal(index, woffs.to(torch.int64))
        assert int(index[nb].item()) == total
        out = torch.zeros(max(n, 1), dtype=torch.uint8, device="cuda")
        relaxed = wl == "uniform256"
        assert group.decode(stream, total, n, bs, out, own_layout=True, relaxed=relaxed) == n and torch.equal(out[:n], data)
        out.zero_()
        got, legs = group.decode(stream, total, n, bs, out, index=index, relaxed=relaxed, legs=True)
        assert got == n and torch.equal(out[:n], data) and len(legs) == 4
    # the calls say no where they must
    from libhuffman_amd.codec import HuffmanGpuError
    with pytest.raises(HuffmanGpuError):
        group.decode(stream, total, n + 1, bs, out, own_layout=True)          # not the layout of the last encode
    with pytest.raises(HuffmanGpuError):
        group.encode(data, n, bs, stream[:10])                                # no room for the bound
    group.close()


def _run_mock_ranks(nranks, extra_args=(), timeout=300):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory(prefix="mockrccl") as d:
        so = os.path.join(d, "libmock_rccl.so")
        subprocess.check_call([hipcc, "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"),
                               "-o", so, "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        env = dict(os.environ, HUF_GPU_RCCL_LIB=so, MOCK_RCCL_DIR=d)
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "sharded_worker.py"), str(r), str(nranks), d, *extra_args],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
        outs, codes = [], []
        try:
            for p in procs:
                try:
                    o, _ = p.communicate(timeout=timeout)
                except subprocess.TimeoutExpired:
                    o = "(timed out)"
                outs.append(o)
                codes.append(p.returncode)
        finally:
            for p in procs:                      # (a rank that waits for a dead peer)
                if p.poll() is None:
                    p.kill()
                    p.wait()
        return outs, codes


@pytest.mark.gpu
def test_a_rank_that_never_arrives_costs_the_others_a_deadline_not_a_hang(torch_mod):
    """RCCL failures map to HUF_ERROR_FATAL.  Three ranks, the last never makes the call: the other two
    return HUF_ERROR_FATAL within the deadline they were given (1.5 s + the grace of the abort), their group is broken
    (every later call fails at once), hufgpu_shard_destroy still works and the codec behind it is untouched."""
    outs, codes = _run_mock_ranks(3, ("absent",), timeout=120)
    for r in range(3):
        assert codes[r] == 0 and "DONE" in outs[r], "rank %d:\n%s" % (r, outs[r][-3000:])
    assert "returned HUF_ERROR_FATAL" in outs[0] and "returned HUF_ERROR_FATAL" in outs[1] and "never arrived" in outs[2]


@pytest.mark.gpu
def test_three_ranks_on_one_gpu_over_the_mock_transport(torch_mod):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory(prefix="mockrccl") as d:
        so = os.path.join(d, "libmock_rccl.so")
        subprocess.check_call([hipcc, "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"),
                               "-o", so, "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        env = dict(os.environ, HUF_GPU_RCCL_LIB=so, MOCK_RCCL_DIR=d)
        nranks = 3
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "sharded_worker.py"), str(r), str(nranks), d],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
        outs, codes = [], []
        try:
            for p in procs:
                try:
                    o, _ = p.communicate(timeout=300)
                except subprocess.TimeoutExpired:
                    o = "(timed out)"
                outs.append(o)
                codes.append(p.returncode)
        finally:
            for p in procs:                      # (a rank that waits for a dead peer)
                if p.poll() is None:
                    p.kill()
                    p.wait()
        for r in range(nranks):
            assert codes[r] == 0 and "DONE" in outs[r], "rank %d:\n%s" % (r, outs[r][-3000:])
