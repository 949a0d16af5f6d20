"""Raw-stream decode (no block index, no sub-index: what huf_decode() gets, src/decoder.c:205-283) of
blocks of many MiB - the reference's default blocksize = 0 makes the whole input ONE block
(src/encoder.c:163-165).  The library builds a sub-index for such a block on the device
(kernels/spec_index.hpp) and decodes it chunk by chunk; every result is compared with the input, the
error cases with the CPU oracle (error code, bytes delivered, the bytes themselves) and with the in-order
decoder of the same library."""
import time

import numpy as np
import pytest

from libhuffman_amd import datagen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


@pytest.fixture(scope="module")
def codec(torch_mod):
    from libhuffman_amd.codec import GpuCodec
    c = GpuCodec(0)
    yield c
    c.close()


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).cuda()


def fibonacci_bytes(n, k=24):
    """byte i occurs ~fib(i) times: codes up to k - 1 bits (longer than the decoder's 12-bit table)"""
    f = [1, 1]
    while len(f) < k:
        f.append(f[-1] + f[-2])
    rep = max(1, n // sum(f))
    one = np.concatenate([np.full(c, i, np.uint8) for i, c in enumerate(f)])
    rng = np.random.default_rng(5)
    data = np.tile(one, rep + 1)[:n].copy()
    rng.shuffle(data)
    return data


@pytest.mark.parametrize("kind", ["zipf255", "uniform256", "const41", "two", "fib"])
def test_one_block_raw_stream(torch_mod, codec, kind):
    torch = torch_mod
    n = (48 << 20) + 12345
    if kind == "two":
        data = dev(torch, (datagen.uniform256(n) & 1) * 7)
    elif kind == "fib":
        data = dev(torch, fibonacci_bytes(n))
    else:
        data = codec.fill(torch.empty(n, dtype=torch.uint8, device="cuda"), kind)
    stream, offs, length = codec.encode(data, 0)                 # blocksize 0: one block
    assert offs.numel() == 2
    out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    res = codec.decode_stream(stream, length, length, out, relaxed=True)
    assert res == (0, n, length), (kind, res)
    assert torch.equal(out[:n], data), kind
    # again, timed: one workgroup for the whole block would take tens of milliseconds here
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = codec.decode_stream(stream, length, length, out, relaxed=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert res == (0, n, length) and torch.equal(out[:n], data), kind
    print(f"\n  raw stream, one block of {n >> 20} MiB of {kind}: {ms:.2f} ms = {n / ms / 1.074e6:.1f} GiB/s")
    assert ms < 15.0, (kind, ms)


def test_several_big_blocks_then_small(torch_mod, codec):
    """big blocks of different statistics in a row (the lanes behind a short payload decode the next
    block's header as if it were payload - and must not matter), a short last block, and a tail of
    small blocks appended as a second stream"""
    torch = torch_mod
    bs = (6 << 20) + 77
    parts = [datagen.zipf255(bs), (datagen.uniform256(bs) & 1) * 9, datagen.uniform256(bs), np.full(bs, 3, np.uint8),
             datagen.zipf255(bs // 5)]
    data = np.concatenate(parts)
    d = dev(torch, data)
    stream, offs, length = codec.encode(d, bs)
    tail = dev(torch, datagen.zipf255(300000))
    stream2, _, length2 = codec.encode(tail, 65536)
    both = torch.cat([stream[:length], stream2[:length2]])
    out = torch.zeros(data.size + 300000 + 64, dtype=torch.uint8, device="cuda")
    for sequential in (False, True):
        res = codec.decode_stream(both, both.numel(), both.numel(), out.zero_(), relaxed=True, sequential=sequential)
        assert res == (0, data.size + 300000, both.numel()), (sequential, res)
        assert torch.equal(out[:data.size], d) and torch.equal(out[data.size:data.size + 300000], tail), sequential
    # `length` ends inside the second block: blocks that begin before it are decoded whole
    cut = int(offs[1].item()) + 5
    res = codec.decode_stream(both, both.numel(), cut, out.zero_(), relaxed=True)
    ref = codec.decode_stream(both, both.numel(), cut, torch.zeros_like(out), relaxed=True, sequential=True)
    assert res == ref and res[0] == 0 and res[1] == 2 * bs, (res, ref)
    assert torch.equal(out[:2 * bs], d[:2 * bs])


def oracle_agrees(oracle, stream_dev, avail, ln, got, out_dev, relaxed=False):
    """(err, raw, used) and the delivered bytes of a decode_stream() call against oracle.decode() on the same bytes
    (src/decoder.c:34-96, 201-287 restated on the CPU): the error code, the number of bytes delivered, the bytes."""
    host = stream_dev[:avail].cpu().numpy()
    oerr, oout, oused = oracle.decode(host, out_dev.numel(), 1025 if relaxed else 1024, length=ln)
    assert oerr != 1, "the oracle's buffer is as large as the device's"
    assert got[0] == oerr, (got, oerr, oout.size, oused)
    assert got[1] == oout.size, (got, oerr, oout.size, oused)
    assert np.array_equal(out_dev[:got[1]].cpu().numpy(), oout)
    if oerr == 0:
        assert got[2] == oused, (got, oused)


def test_damaged_big_blocks_match_the_in_order_decoder(torch_mod, codec, oracle):
    torch = torch_mod
    bs = 5 << 20
    data = dev(torch, np.concatenate([datagen.zipf255(bs), datagen.zipf255(bs)[::-1], datagen.zipf255(bs // 2)]))
    stream, offs, length = codec.encode(data, bs)
    o = [int(x) for x in offs.cpu().tolist()]
    good = stream[:length].clone()
    out = torch.zeros(data.numel() + 64, dtype=torch.uint8, device="cuda")
    ref_out = torch.zeros_like(out)

    def both(s, avail, ln):
        a = codec.decode_stream(s, avail, ln, out.zero_())
        b = codec.decode_stream(s, avail, ln, ref_out.zero_(), sequential=True)
        assert a == b, (a, b)
        n = a[1]
        assert torch.equal(out[:n], ref_out[:n])
        oracle_agrees(oracle, s, avail, ln, a, out)             # ... and both are the reference's outcome
        return a

    # payload damage in the middle of the second block: zipf255's tree is full, so the bits still
    # decode to SOMETHING - the block's symbol count and the following header decide
    bad = good.clone()
    mid = (o[1] + o[2]) // 2
    bad[mid:mid + 64] ^= 0x5a
    both(bad, bad.numel(), bad.numel())
    # the tree of the second block damaged
    bad = good.clone()
    bad[o[1] + 12] ^= 0xff
    both(bad, bad.numel(), bad.numel())
    # block_len of the first block one too large / far too large
    for delta in (1, 1 << 33):
        bad = good.clone()
        v = int.from_bytes(bytes(bad[:8].cpu().tolist()), "little") + delta
        bad[:8] = torch.tensor(list(v.to_bytes(8, "little")), dtype=torch.uint8, device="cuda")
        both(bad, bad.numel(), bad.numel())
    # truncated inside the first block's payload, inside the second block's tree, right after the second block
    for avail in (o[1] - 1000, o[1] + 40, o[2]):
        both(good, avail, avail)
    # output too small for the second block
    small = torch.zeros(bs + 100, dtype=torch.uint8, device="cuda")
    a = codec.decode_stream(good, good.numel(), good.numel(), small)
    b = codec.decode_stream(good, good.numel(), good.numel(), torch.zeros_like(small), sequential=True)
    assert a == b, (a, b)


def test_c_api_default_blocksize_is_one_block(torch_mod):
    """huf_encode() then huf_decode() through memory streams with blocksize = 0 (README.md:50-52 leaves
    it at the default): ONE block of 96 MiB; the stream equals the oracle's byte for byte."""
    import ctypes as C
    from libhuffman_amd import _native as N
    from oracle.oracle import Oracle
    L = N.load()
    n = (96 << 20) + 321
    data = datagen.zipf255(n)
    want = Oracle().encode(data, 0)
    rin, rout, rback = C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)()
    bin_, bout, bback = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.huf_memopen(C.byref(rin), C.byref(bin_), n) == 0
    assert L.huf_memopen(C.byref(rout), C.byref(bout), n) == 0
    assert L.huf_memopen(C.byref(rback), C.byref(bback), n) == 0
    assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
    cfg = N.Config(n, 0, 0, 0, rin, rout)
    t0 = time.perf_counter()
    assert L.huf_encode(C.byref(cfg)) == 0
    t1 = time.perf_counter()
    m = C.c_size_t()
    L.huf_memlen(rout, C.byref(m))
    enc = np.frombuffer(C.string_at(bout.value, m.value), np.uint8)
    assert enc.size == want.size and np.array_equal(enc, want)
    assert int.from_bytes(enc[:8].tobytes(), "little") == n          # one block
    dcfg = N.Config(m.value, 0, 0, 0, rout, rback)
    t2 = time.perf_counter()
    assert L.huf_decode(C.byref(dcfg)) == 0
    t3 = time.perf_counter()
    L.huf_memlen(rback, C.byref(m))
    assert m.value == n
    assert np.array_equal(np.frombuffer(C.string_at(bback.value, n), np.uint8), data)
    print(f"\n  C API, blocksize 0, {n >> 20} MiB: huf_encode {(t1 - t0) * 1e3:.1f} ms, huf_decode {(t3 - t2) * 1e3:.1f} ms "
          f"(host memory streams, PCIe included)")
    for r in (rin, rout, rback):
        L.huf_memclose(C.byref(r))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for b in (bin_, bout, bback):
        libc.free(b)


@pytest.mark.parametrize("devices,threads,want_live", [("0,0,0", 3, 3), (None, 3, 1)])
def test_concurrent_calls_take_different_sessions(torch_mod, devices, threads, want_live):
    """HUF_GPU_DEVICES lists one session per entry; concurrent huf_encode/huf_decode calls from
    different threads run side by side on different sessions (here: three on the one GPU of the
    box), bit-exact with the oracle; without the variable there is one session and they take turns."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("HUF_GPU_DEVICES", None)
    if devices:
        env["HUF_GPU_DEVICES"] = devices
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sessions_child.py"), root, str(threads)],
                       env=env, capture_output=True, text=True, timeout=600)
    print("\n  " + r.stdout.strip().replace("\n", "\n  "))
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"sessions live={want_live} configured={want_live if devices else 1}" in r.stdout, r.stdout


def test_one_block_of_3_5_gib(torch_mod, codec):
    """A block just below 2^32 bytes (where 32-bit symbol counts end) - blocksize = 0 on a multi-GiB input.  The oracle cannot encode that in test time; parity comes from two properties:
    (1) counts scaled by a common factor give the same tree (every comparison of src/tree.c:292-427
    scales with them), so the tree of P repeated 8 m times is the tree of P repeated 8 times;
    (2) the payload of 8 copies of P is a whole number of bytes, so the payload of 8 m copies is
    that byte string m times.  P x 8 (128 MiB, one block) is what the oracle encodes."""
    torch = torch_mod
    from oracle.oracle import Oracle
    p_len, m = 16 << 20, 28                                   # 8 * 28 * 16 MiB = 3.5 GiB
    P = datagen.zipf255(p_len)
    want = Oracle().encode(np.tile(P, 8), 0)
    tl = int.from_bytes(want[8:10].tobytes(), "little", signed=True)
    hdr = 10 + 2 * tl
    period = want.size - hdr                                  # payload bytes of 8 copies, no padding bits
    data = dev(torch, P).repeat(8 * m)
    n = data.numel()
    assert n == 8 * m * p_len and n < (1 << 32)
    sub = codec.new_sub_index(n, 0)
    stream, offs, length = codec.encode(data, 0, sub_index=sub)
    assert offs.numel() == 2 and length == hdr + m * period
    head = stream[:hdr].cpu().numpy()
    assert int.from_bytes(head[:8].tobytes(), "little") == n
    assert np.array_equal(head[8:], want[8:hdr]), "tree differs from the oracle's tree of the same proportions"
    assert torch.equal(stream[hdr:hdr + period], dev(torch, want[hdr:])), "first period differs from the oracle's payload"
    assert torch.equal(stream[hdr:length - period], stream[hdr + period:length]), "payload is not periodic"
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    # with the encoder's sub-index, then as a raw stream (the sub-index is built on the device)
    assert codec.decode(stream, length, offs, 1, out, sub_index=sub, raw_size=n, blocksize=0) == n
    assert torch.equal(out, data)
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = codec.decode_stream(stream, length, length, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert res == (0, n, length) and torch.equal(out, data)
    print(f"\n  raw stream, one block of {n / 2**30:.2f} GiB: {ms:.1f} ms")
    assert ms < 200.0


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_big_blocks_against_the_in_order_decoder(torch_mod, codec, oracle, seed):
    """random alphabets, skews and block sizes around 4-7 MiB; each stream also with random damage
    (bit flips anywhere, truncation): the parallel path (device-built sub-index) and the in-order
    decoder must agree on the error, the bytes written, the bytes consumed and the output"""
    torch = torch_mod
    rng = np.random.default_rng(1000 + seed)
    for case in range(5):
        k = int(rng.choice([2, 3, 5, 17, 64, 200, 256]))
        skew = float(rng.choice([0.0, 0.5, 1.0, 2.0, 4.0]))
        w = 1.0 / np.arange(1, k + 1) ** skew
        bs = int(rng.integers(4 << 20, 7 << 20))
        nblk = int(rng.integers(1, 4))
        n = bs * (nblk - 1) + int(rng.integers(1, bs + 1))
        syms = rng.permutation(256)[:k].astype(np.uint8)
        data = syms[rng.choice(k, size=n, p=w / w.sum())]
        d = dev(torch, data)
        stream, offs, length = codec.encode(d, bs)
        good = stream[:length].clone()
        out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        ref = torch.zeros_like(out)
        a = codec.decode_stream(good, length, length, out, relaxed=True)
        assert a == (0, n, length) and torch.equal(out[:n], d), (seed, case, k, skew, bs, a)
        for trial in range(4):
            bad = good.clone()
            avail = length
            kind = int(rng.integers(0, 4))
            if kind == 0:                                   # a burst of bit flips somewhere
                pos = int(rng.integers(0, length - 16))
                bad[pos:pos + int(rng.integers(1, 16))] ^= int(rng.integers(1, 256))
            elif kind == 1:                                 # inside a header or tree
                b = int(rng.integers(0, nblk))
                bad[int(offs[b].item()) + int(rng.integers(0, 40))] ^= int(rng.integers(1, 256))
            elif kind == 2:                                 # truncated
                avail = int(rng.integers(1, length))
            else:                                           # a single flipped bit
                bad[int(rng.integers(0, length))] ^= 1 << int(rng.integers(0, 8))
            a = codec.decode_stream(bad, avail, avail, out.zero_(), relaxed=True)
            b = codec.decode_stream(bad, avail, avail, ref.zero_(), relaxed=True, sequential=True)
            assert a == b, (seed, case, trial, kind, k, skew, bs, a, b)
            assert torch.equal(out[:a[1]], ref[:a[1]]), (seed, case, trial, kind)
            if trial < 2:                                   # (the CPU oracle takes a second per stream: two of the four)
                oracle_agrees(oracle, bad, avail, avail, a, out, relaxed=True)


def test_blocks_beyond_4_gib(torch_mod, codec):
    """Blocks of 2^32 bytes and more (the reference's block_len is a uint64_t, src/encoder.c:338-341): a
    4.5 GiB block of Zipf bytes - parity through the two properties of test_one_block_of_3_5_gib - and a
    4.25 GiB block of one byte value, whose count does not fit 32 bits.  With the encoder's sub-index, as
    a raw stream (sub-index built on the device), and cut short (the in-order decoder takes over)."""
    torch = torch_mod
    from oracle.oracle import Oracle
    p_len, m = 16 << 20, 36                                   # 8 * 36 * 16 MiB = 4.5 GiB
    P = datagen.zipf255(p_len)
    want = Oracle().encode(np.tile(P, 8), 0)
    tl = int.from_bytes(want[8:10].tobytes(), "little", signed=True)
    hdr = 10 + 2 * tl
    period = want.size - hdr
    data = dev(torch, P).repeat(8 * m)
    n = data.numel()
    assert n == 8 * m * p_len and n > (1 << 32)
    sub = codec.new_sub_index(n, 0)
    stream, offs, length = codec.encode(data, 0, sub_index=sub)
    assert offs.numel() == 2 and length == hdr + m * period
    head = stream[:hdr].cpu().numpy()
    assert int.from_bytes(head[:8].tobytes(), "little") == n
    assert np.array_equal(head[8:], want[8:hdr]), "tree differs from the oracle's tree of the same proportions"
    assert torch.equal(stream[hdr:hdr + period], dev(torch, want[hdr:])), "first period differs from the oracle's payload"
    assert torch.equal(stream[hdr:length - period], stream[hdr + period:length]), "payload is not periodic"
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, offs, 1, out, sub_index=sub, raw_size=n, blocksize=0) == n
    assert torch.equal(out, data)
    res = codec.decode_stream(stream, length, length, out.zero_())
    assert res == (0, n, length) and torch.equal(out, data)
    # cut short by a few bytes: no parallel path applies, the in-order decoder delivers what is there
    t0 = time.perf_counter()
    err, raw, used = codec.decode_stream(stream, length - 100, length - 100, out.zero_())
    torch.cuda.synchronize()
    print(f"\n  a 4.5 GiB block cut short, in-order decoder: {time.perf_counter() - t0:.1f} s, err {err}, {raw} bytes")
    assert err == 3 and n - 200 <= raw < n and torch.equal(out[:raw], data[:raw])
    del out, sub, stream, data
    torch.cuda.empty_cache()

    n1 = (1 << 32) + (1 << 28) + 5                            # 4.25 GiB of one byte value
    ones = torch.full((n1,), 0x37, dtype=torch.uint8, device="cuda")
    stream, offs, length = codec.encode(ones, 0)
    assert length == 10 + 10 + (n1 + 7) // 8
    hb = stream[:20].cpu().numpy()
    assert int.from_bytes(hb[:8].tobytes(), "little") == n1 and int.from_bytes(hb[8:10].tobytes(), "little") == 5
    assert int(stream[20:length].max().item()) == 0
    back = torch.zeros(n1, dtype=torch.uint8, device="cuda")
    res = codec.decode_stream(stream, length, length, back)
    assert res == (0, n1, length) and torch.equal(back, ones)
    # one byte more than the limit is an argument error, not a wrong stream
    with pytest.raises(Exception):
        codec.encode(torch.zeros(16, dtype=torch.uint8, device="cuda"), (1 << 38) + 1)


@pytest.mark.parametrize("bs", [65536, (1 << 20) + 7])
def test_one_call_fans_out_over_free_sessions(torch_mod, bs):
    """With several sessions configured and free, ONE huf_encode() of a memory stream cuts its input
    into rounds of whole blocks and gives them to the sessions (threads of their own); the stream is
    byte for byte the oracle's.  Three sessions on the one GPU here; rounds of 16 MiB."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HUF_GPU_DEVICES"] = "0,0,0"
    env["HUF_GPU_BATCH_MB"] = "16"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sessions_child.py"), root, "1", "150", str(bs)],
                       env=env, capture_output=True, text=True, timeout=600)
    print("\n  " + r.stdout.strip().replace("\n", "\n  "))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sessions live=3 configured=3" in r.stdout, r.stdout
    # both directions went over the three sessions: 4 encodes and 4 decodes of 150 MiB (1 warm-up + 3 timed)
    import re
    m = re.search(r"fanout_encodes=(\d+) fanout_decodes=(\d+)", r.stdout)
    assert m and int(m.group(1)) == 4 and int(m.group(2)) == 4, r.stdout


def test_huffmanfile_over_several_sessions(torch_mod):
    """BASELINE configs[4]'s route on what one box has: 160 MiB of log text in 1 MiB blocks through
    huffmanfile.compress() / decompress() (the reference's huffmanfile.py:294-342, 385-417) with three device sessions
    configured - ONE call of either is dealt out over them.  Stream = the oracle's, round trip = the input, and
    huf_gpu_fanouts() counts both directions."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HUF_GPU_DEVICES"] = "0,0,0"
    env["HUF_GPU_BATCH_MB"] = "16"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fanout_huffmanfile_child.py"), root, "160"],
                       env=env, capture_output=True, text=True, timeout=900)
    print("\n  " + r.stdout.strip().replace("\n", "\n  "))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sessions live=3 configured=3" in r.stdout, r.stdout
    assert "stream_is_oracles=True roundtrip=True" in r.stdout and "incremental_equals_one_shot=True" in r.stdout, r.stdout
    m = re.search(r"fanout_encodes=(\d+) fanout_decodes=(\d+)", r.stdout)
    assert m and int(m.group(1)) >= 1 and int(m.group(2)) >= 1, r.stdout


@pytest.mark.parametrize("run_kib,limit_ms", [(2, 10.0), (24, 25.0), (300, 200.0), (2048, 4000.0)])
def test_runs_of_one_byte_inside_a_big_block(torch_mod, codec, run_kib, limit_ms):
    """A run of one byte value is a periodic bit string: a lane that starts inside it can lock onto the
    pattern a bit off and never fall into step.  The broken chain is mended one share (512 payload
    bytes) per round (spec_repair_kernel: 0.13 ms per round, stretches in parallel); a run of more
    than 1 024 shares leaves the block to the in-order decoder.  Either way the output is exact."""
    torch = torch_mod
    n = 40 << 20
    data = datagen.zipf255(n).copy()
    rng = np.random.default_rng(run_kib)
    run = run_kib << 10
    for i, value in enumerate((0, 0, 1, 3, 17, 200)):
        at = int(rng.integers(0, n - run - 1))
        data[at:at + run] = value
    d = dev(torch, data)
    stream, offs, length = codec.encode(d, 0)
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert codec.decode_stream(stream, length, length, out) == (0, n, length)
    assert torch.equal(out, d)
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert codec.decode_stream(stream, length, length, out) == (0, n, length)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert torch.equal(out, d)
    print(f"\n  40 MiB block with runs of {run_kib} KiB: {ms:.2f} ms")
    if limit_ms is not None:
        assert ms < limit_ms, ms


@pytest.mark.parametrize("blocksize", [64 << 10, 256 << 10, 1 << 20])
def test_raw_stream_of_blocks_whose_bit_rate_changes(torch_mod, codec, blocksize):
    """The raw-stream probe guesses where a block's payload ends from the bits per symbol of the block so far
    (decode_fast.hpp): blocks that begin with cheap symbols and end with dear ones, and the other way round, and
    blocks that end in a run - the guess fails or overshoots there, and the result must not depend on it."""
    torch = torch_mod
    nb = 96
    n = nb * blocksize - 777
    rng = np.random.default_rng(blocksize)
    data = np.empty(n, np.uint8)
    for b in range(nb):
        lo, hi = b * blocksize, min((b + 1) * blocksize, n)
        m = hi - lo
        cut = int(m * rng.choice([0.3, 0.5, 0.75, 0.9]))
        cheap = np.where(rng.random(m) < 0.93, 7, rng.integers(0, 256, m)).astype(np.uint8)
        dear = rng.integers(0, 255, m).astype(np.uint8)
        kind = b % 4
        if kind == 0:
            blk = np.concatenate([cheap[:cut], dear[cut:]])
        elif kind == 1:
            blk = np.concatenate([dear[:cut], cheap[cut:]])
        elif kind == 2:
            blk = np.concatenate([dear[:cut], np.full(m - cut, 7, np.uint8)])
            blk[::97] = dear[::97]                       # (keep the tree an ordinary one)
        else:
            blk = np.concatenate([cheap[:cut // 2], dear[cut // 2:cut], cheap[cut:]])
        data[lo:hi] = blk
    d = dev(torch, data)
    stream, offs, length = codec.encode(d, blocksize)
    out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    assert codec.decode_stream(stream, length, length, out, relaxed=True) == (0, n, length)
    assert torch.equal(out[:n], d)
    ref = torch.zeros_like(out)
    assert codec.decode_stream(stream, length, length, ref, relaxed=True, sequential=True) == (0, n, length)
    assert torch.equal(ref, out)


@pytest.mark.gpu
def test_small_and_full_size_streams_with_codes_beyond_the_table(torch_mod):
    """round 6b.  (a) huf_decode of small memory streams (hufgpu_decode_small: the in-order chain with decode_regs in front) whose
    one block has codes of more than 12 bits: 9 000 ... 100 000 bytes of skewed bytes with rare ones, stream and round trip against
    the oracle.  (b) 256 MiB of the same kind in 64 KiB blocks with the block index alone and as a raw stream: the round trip, and
    the first and last blocks of the stream against the oracle's."""
    import ctypes as C
    torch = torch_mod
    from libhuffman_amd import _native as N
    from libhuffman_amd.codec import GpuCodec
    from oracle.oracle import Oracle
    L = N.load()
    orc = Oracle()
    rng = np.random.default_rng(66)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    def skewed(n):
        a = (rng.zipf(1.35, size=n) % 180).astype(np.uint8)
        k = max(3, n // 3000)
        a[rng.integers(0, n, size=k)] = rng.integers(180, 255, size=k).astype(np.uint8)
        return a

    for n in (9000, 20000, 40001, 65536, 100000):
        data = skewed(n)
        bs = 65536
        want = orc.encode(data, bs)
        rin, rout, rback = C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)()
        bin_, bout, bback = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for r, b in ((rin, bin_), (rout, bout), (rback, bback)):
            assert L.huf_memopen(C.byref(r), C.byref(b), 64) == 0
        assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
        cfg = N.Config(n, bs, 0, 0, rin, rout)
        assert L.huf_encode(C.byref(cfg)) == 0
        m = C.c_size_t()
        L.huf_memlen(rout, C.byref(m))
        enc = np.frombuffer(C.string_at(bout.value, m.value), np.uint8)
        assert np.array_equal(enc, want), n
        dcfg = N.Config(m.value, bs, 0, 0, rout, rback)
        assert L.huf_decode(C.byref(dcfg)) == 0, n
        L.huf_memlen(rback, C.byref(m))
        assert m.value == n and np.array_equal(np.frombuffer(C.string_at(bback.value, n), np.uint8), data), n
        for r in (rin, rout, rback):
            L.huf_memclose(C.byref(r))
        for b in (bin_, bout, bback):
            libc.free(b)

    c = GpuCodec(0)
    tile, n, bs = 8 << 20, 256 << 20, 65536
    host = skewed(tile)
    data = torch.from_numpy(host).cuda().repeat(n // tile)
    nb = c.block_count(n, bs)
    stream, offs, length = c.encode(data, bs)
    oh = offs.cpu().numpy()
    want_tile = orc.encode(host, bs)                                 # the stream of one tile: the first 128 blocks of the whole
    assert np.array_equal(stream[:want_tile.size].cpu().numpy(), want_tile)
    assert np.array_equal(stream[int(oh[nb - 128]):length].cpu().numpy(), want_tile)      # and the last 128
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert c.decode(stream, length, offs, nb, back) == n and torch.equal(back, data)
    assert c.decode_counters()[0] == 0
    back.zero_()
    res = c.decode_stream(stream, length, length, back)
    assert res[0] == 0 and res[1] == n and torch.equal(back, data)
