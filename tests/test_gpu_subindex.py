"""GPU parity tests of the sub-index decode path (hufgpu_encode_sub / hufgpu_decode_sub).

The sub-index is side information from the encoder; the decoder verifies it.  So two things are
tested: with the encoder's own sub-index the output equals the oracle's decode, and with ANY other
content of the sub-index buffer - garbage, zeros, the sub-index of another input - or a damaged
stream, results and error codes are those of the plain indexed decode and of the oracle.
"""
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
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return torch.from_numpy(a).cuda() if a.size else torch.empty(0, dtype=torch.uint8, device="cuda")


def encode_sub(torch, codec, data, bs):
    d = dev(torch, data)
    sub = codec.new_sub_index(d.numel(), bs)
    sub.fill_(-1)                                   # what the encoder does not write must not matter
    stream, offs, length = codec.encode(d, bs, sub_index=sub)
    return stream, offs, length, sub


def decode_sub(torch, codec, stream, length, offs, n, bs, sub, relaxed=False, cap=None):
    out = torch.zeros(max(cap if cap is not None else n, 1), dtype=torch.uint8, device="cuda")
    nb = codec.block_count(n, bs)
    raw = codec.decode(stream, length, offs, nb, out, relaxed=relaxed, sub_index=sub, raw_size=n, blocksize=bs)
    return raw, out


def random_data(rng, n, k, conc):
    alphabet = rng.choice(256, size=k, replace=False)
    p = rng.dirichlet(np.full(k, conc))
    return alphabet[rng.choice(k, size=n, p=p)].astype(np.uint8)


@pytest.mark.parametrize("kind,n,bs", [
    ("zipf255", 262144, 65536), ("uniform256", 262144, 65536), ("uniform255", 262144, 65536),
    ("const41", 262144, 65536), ("zipf255", 65536 + 1000, 65536), ("zipf255", 2 << 20, 1 << 20),
    ("logtext", 3 << 20, 1 << 20), ("zipf255", 300001, 0), ("uniform256", 1 << 20, 0),
    ("zipf255", 100000, 1000), ("uniform255", 70001, 4097), ("zipf255", 5000, 37), ("const41", 1 << 20, 0),
])
def test_sub_index_roundtrip_equals_oracle(torch_mod, codec, oracle, kind, n, bs):
    torch = torch_mod
    data = datagen.GENERATORS[kind](n)
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    want = oracle.encode(data, bs)
    got = stream.cpu().numpy()
    assert got.size == want.size and np.array_equal(got, want)       # the stream does not change with the sub-index
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, sub, relaxed=True)
    assert raw == n
    assert torch.equal(out[:n], dev(torch, data))


@pytest.mark.parametrize("seed", range(12))
def test_sub_index_random_shapes(torch_mod, codec, oracle, seed):
    torch = torch_mod
    rng = np.random.default_rng(4200 + seed)
    n = int(rng.integers(1, 400000))
    k = int(rng.integers(1, 257))
    data = random_data(rng, n, k, float(rng.choice([0.02, 0.3, 1.0, 10.0])))
    bs = int(rng.choice([0, 31, 32, 33, 256, 1000, 8192, 16384, 16385, 65536, 100000, 131072]))
    if bs and n // bs > 3000:
        data = data[: bs * 3000]
        n = data.size
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    want = oracle.encode(data, bs)
    assert np.array_equal(stream.cpu().numpy(), want)
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, sub, relaxed=True)
    assert raw == n and torch.equal(out[:n], dev(torch, data)), (n, k, bs)


def test_sub_index_deep_codes(torch_mod, codec):
    """Fibonacci-weighted symbols: codes far beyond the 12 table bits, groups of many hundred bits."""
    torch = torch_mod
    for bs, nsym in ((65536, 22), (1 << 20, 28), (8 << 20, 32)):
        fib = [1, 1]
        while len(fib) < nsym:
            fib.append(fib[-1] + fib[-2])
        w = np.array(fib[:nsym], dtype=np.float64)
        counts = np.maximum(1, np.floor(w / w.sum() * bs)).astype(np.int64)
        counts[-1] += bs - counts.sum()
        block = np.repeat(np.arange(nsym, dtype=np.uint8), counts)
        rng = np.random.default_rng(bs)
        # rare (deep) symbols first, so that whole tiles hold long codes only
        data = np.concatenate([block, rng.permutation(block)])
        stream, offs, length, sub = encode_sub(torch, codec, data, bs)
        raw, out = decode_sub(torch, codec, stream, length, offs, data.size, bs, sub)
        assert raw == data.size and torch.equal(out[:data.size], dev(torch, data)), bs


@pytest.mark.parametrize("what", ["garbage", "zeros", "ones", "other_input", "shifted", "one_group_off", "tile_off"])
def test_wrong_sub_index_costs_time_not_correctness(torch_mod, codec, what):
    torch = torch_mod
    n, bs = 5 * 65536 + 777, 65536
    data = datagen.zipf255(n)
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    bad = sub.clone()
    nb = codec.block_count(n, bs)
    tiles = nb * ((bs + 2047) // 2048)               # int64 words of tile starts in front of the group counts
    if what == "garbage":
        bad = torch.randint(-2**62, 2**62, bad.shape, dtype=torch.int64, device="cuda")
    elif what == "zeros":
        bad.zero_()
    elif what == "ones":
        bad.fill_(-1)
    elif what == "other_input":
        _, _, _, bad = encode_sub(torch, codec, datagen.uniform255(n), bs)
    elif what == "shifted":
        bad = torch.roll(sub, 1)
    elif what == "one_group_off":
        v = bad.view(torch.int16)
        g = 4 * tiles + 3 * 2048 + 100               # group 100 of block 3: one bit moves to its neighbour
        v[g] += 1
        v[g + 1] -= 1
    elif what == "tile_off":
        bad[2 * 32 + 5] += 8                         # tile 5 of block 2
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, bad)
    assert raw == n and torch.equal(out[:n], dev(torch, data)), what
    # and the codec context is as good as new afterwards
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, sub)
    assert raw == n and torch.equal(out[:n], dev(torch, data))


def test_damaged_stream_matches_the_oracle_block_by_block(torch_mod, codec, oracle):
    """A stream damaged after the encode, decoded with the block index alone and with the sub-index: against
    the ORACLE.  The index fixes where every block starts, so the reference's behaviour is that of its decoder
    run on each block's record [o0, o1) by itself (src/decoder.c:218-276 with that record as the whole input):
    blocks in front of the damaged one are intact; the damaged one either still decodes - to whatever symbols
    the oracle reads out of it - or fails with the oracle's error code after the oracle's number of delivered
    symbols (src/decoder.c:69-91), which are in the output too."""
    torch = torch_mod
    from libhuffman_amd.codec import HuffmanGpuError
    n, bs = 6 * 65536, 65536
    rng = np.random.default_rng(77)
    seen = set()
    for kind in ("zipf255", "uniform255", "const41"):
        data = datagen.GENERATORS[kind](n)
        stream, offs, length, sub = encode_sub(torch, codec, data, bs)
        offs_h = offs.cpu().numpy()
        nb = codec.block_count(n, bs)
        for trial in range(16):
            s = stream[:length].clone()
            b = int(rng.integers(0, nb))
            lo, hi = int(offs_h[b]), int(offs_h[b + 1])
            mode = trial % 4
            if mode == 0:                             # a flipped payload bit
                pos = int(rng.integers(lo + (hi - lo) // 2, hi))
                s[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:                           # payload bytes of 0xff (walks leave the tree at once)
                pos = int(rng.integers(lo + (hi - lo) // 2, hi - 8))
                s[pos:pos + 8] = 0xff
            elif mode == 2:                           # a damaged tree entry
                pos = lo + 10 + 2 * int(rng.integers(0, 5))
                s[pos] ^= 0x55
            else:                                     # a block length that claims a few symbols more
                s[lo] = (int(s[lo]) + 1 + int(rng.integers(0, 40))) & 0xff
            record = s[lo:hi].cpu().numpy()
            oerr, oout, _ = oracle.decode(record, bs + 4096, 1024, length=1)        # length 1: exactly one block
            seen.add(oerr)
            for kw in ({}, dict(sub_index=sub, raw_size=n, blocksize=bs)):
                out = torch.zeros(n + 65536, dtype=torch.uint8, device="cuda")
                tag = (kind, trial, mode, b, "sub" if kw else "index", oerr, oout.size)
                if oerr == 0 and oout.size == bs:
                    raw = codec.decode(s, length, offs, nb, out, **kw)
                    want = data.copy()
                    want[b * bs:(b + 1) * bs] = oout
                    assert raw == n and np.array_equal(out[:n].cpu().numpy(), want), tag
                elif oerr == 0:
                    # the block now claims another length: the blocks behind it land elsewhere in the output
                    # (the index knows where they start in the STREAM); only the part in front is pinned here
                    try:
                        codec.decode(s, length, offs, nb, out, **kw)
                    except HuffmanGpuError:
                        pass
                    assert np.array_equal(out[:b * bs + min(oout.size, bs)].cpu().numpy()[:b * bs], data[:b * bs]), tag
                else:
                    with pytest.raises(HuffmanGpuError) as ei:
                        codec.decode(s, length, offs, nb, out, **kw)
                    assert ei.value.err == oerr, tag + (ei.value.err,)
                    assert ei.value.raw == b * bs + oout.size, tag + (ei.value.raw,)
                    got = out[:ei.value.raw].cpu().numpy()
                    assert np.array_equal(got[:b * bs], data[:b * bs]), tag
                    assert np.array_equal(got[b * bs:], oout), tag
    assert 6 in seen and 0 in seen, seen


def test_sub_index_output_too_small(torch_mod, codec):
    torch = torch_mod
    from libhuffman_amd.codec import HuffmanGpuError
    n, bs = 4 * 65536, 65536
    data = datagen.zipf255(n)
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    with pytest.raises(HuffmanGpuError) as ei:
        decode_sub(torch, codec, stream, length, offs, n, bs, sub, cap=n - 1)
    assert ei.value.err == 1                          # HUF_ERROR_MEMORY_ALLOCATION, like hufgpu_decode


def test_sub_index_unaligned_output(torch_mod, codec):
    torch = torch_mod
    n, bs = 3 * 65536 + 5, 65536
    data = datagen.zipf255(n)
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    nb = codec.block_count(n, bs)
    for shift in (1, 4, 7, 13):
        big = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        out = big[shift:shift + n]
        raw = codec.decode(stream, length, offs, nb, out, sub_index=sub, raw_size=n, blocksize=bs)
        assert raw == n and torch.equal(out, dev(torch, data)), shift
        assert int(big[:shift].sum()) == 0 and int(big[shift + n:].sum()) == 0


def test_full_size_sub_index_roundtrip(torch_mod, codec):
    """BASELINE configs 2-4 at 1 GiB: encode with the sub-index, decode with it, compare all bytes."""
    torch = torch_mod
    n, bs = 1 << 30, 65536
    data = torch.empty(n, dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(codec.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    sub = codec.new_sub_index(n, bs)
    nb = codec.block_count(n, bs)
    for kind in ("zipf255", "uniform256", "const41"):
        codec.fill(data, kind)
        stream, offs, length = codec.encode(data, bs, out=out, sub_index=sub)
        back.zero_()
        raw = codec.decode(out, length, offs, nb, back, relaxed=True, sub_index=sub, raw_size=n, blocksize=bs)
        assert raw == n and torch.equal(back, data), kind
        # the plain indexed decode of the same stream agrees
        back.zero_()
        raw = codec.decode(out, length, offs, nb, back, relaxed=True)
        assert raw == n and torch.equal(back, data), kind


@pytest.mark.parametrize("kind,n,bs", [
    ("zipf255", 3 * 5000001 + 12345, 5000001),      # chunks that do not divide the block, a short last block
    ("uniform256", (9 << 20) + 77, 4 << 20),        # exactly chunk-aligned blocks, k = 256
    ("const41", 12 << 20, 6 << 20),                 # one-symbol blocks, chunked
    ("logtext", 20 << 20, 0),                       # blocksize 0: ONE block of the whole input (src/encoder.c:163-165)
    ("zipf255", (4 << 20) + 1, 0),
    # 2 MiB .. 4 MiB: chunked like the big ones, but with the wave-per-block tree (rates below 2^23); 1 MiB: one workgroup per block
    ("zipf255", (5 << 20) + 4321, 1 << 20),         # configs[4]'s block size, a short last block
    ("logtext", (7 << 20) + 1, (2 << 20) + 77777),  # chunks that do not divide the block
    ("uniform256", 9 << 20, 3 << 20),               # k = 256
    ("const41", (4 << 20) - 1, 2 << 20),            # one-symbol blocks; the last block one byte short
    ("zipf255", (4 << 20) - 1, 0),                  # ONE block just below the 64-bit tree's threshold
])
def test_big_blocks_are_chunked_bit_exact(torch_mod, codec, oracle, kind, n, bs):
    """Blocks of 2 MiB and more are cut into 256 KiB chunks (one workgroup each) by the encoder and
    into 64 KiB chunks by the sub-index decoder; the stream is the oracle's, byte for byte."""
    torch = torch_mod
    data = datagen.GENERATORS[kind](n)
    if kind == "zipf255":                               # a one-symbol block and a two-symbol block among the others
        data = data.copy()
        blk = bs if bs else n
        data[:min(blk, n) // 3] = 7
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    want, woffs = oracle.encode(data, bs, with_offsets=True)
    got = stream.cpu().numpy()
    assert got.size == want.size and np.array_equal(got, want), (kind, n, bs)
    assert np.array_equal(offs.cpu().numpy().astype(np.uint64), woffs)
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, sub, relaxed=True)
    assert raw == n and torch.equal(out[:n], dev(torch, data))
    # the plain indexed decode (one workgroup per block) agrees
    nb = codec.block_count(n, bs)
    out2 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, offs, nb, out2, relaxed=True) == n and torch.equal(out2, out[:n])
    # a damaged chunk start in the sub-index of a big block: found by the chunk in front of it
    bad = sub.clone()
    bad[1] += 8                                         # tile 1 of block 0: in the middle of a chunk
    tiles_per_chunk = 65536 // 2048
    bad[tiles_per_chunk] += 8                           # first tile of decode chunk 1 of block 0
    raw, out3 = decode_sub(torch, codec, stream, length, offs, n, bs, bad, relaxed=True)
    assert raw == n and torch.equal(out3[:n], dev(torch, data))


def test_one_block_of_256_mib(torch_mod, codec, oracle):
    """256 MiB of zipf255 as ONE block (blocksize = 0).  Bit-exact against the
    oracle, and within 3x of the time the same bytes take as 64 KiB blocks."""
    import time
    torch = torch_mod
    n = 256 << 20
    data = torch.empty(n, dtype=torch.uint8, device="cuda")
    codec.fill(data, "zipf255")
    out = torch.empty(codec.encode_bound(n, 65536), dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    times = {}
    for bs in (65536, 0):
        nb = codec.block_count(n, bs)
        offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
        sub = codec.new_sub_index(n, bs)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stream, _, length = codec.encode(data, bs, out=out, offsets=offs, sub_index=sub)
            raw = codec.decode(out, length, offs, nb, back, sub_index=sub, raw_size=n, blocksize=bs)
            torch.cuda.synchronize()
            times[bs] = time.perf_counter() - t0
        assert raw == n and torch.equal(back, data)
        if bs == 0:
            want = oracle.encode(data.cpu().numpy(), 0)
            got = out[:length].cpu().numpy()
            assert got.size == want.size and np.array_equal(got, want)
    assert times[0] < 3 * times[65536] + 1e-3, times


def test_code_lengths_in_the_sub_index_are_checked_against_the_tree(torch_mod, codec):
    """The sub-index carries the code length of every byte value so that the decoder can build its
    tables from prefix sums; it checks them against the stream's tree first.  Lengths that are wrong
    in ways the cheap sums cannot see (two values swapped: the Kraft sum is unchanged) must be
    caught by the position check, and the decode falls back to walking the tree."""
    torch = torch_mod
    n, bs = 4 * 65536, 65536
    for kind in ("zipf255", "uniform255", "logtext"):
        data = datagen.GENERATORS[kind](n)
        stream, offs, length, sub = encode_sub(torch, codec, data, bs)
        nb = codec.block_count(n, bs)
        tiles = nb * ((bs + 2047) // 2048)
        gpb = ((bs + 31) // 32 + 7) & ~7
        lens0 = 8 * tiles + 2 * nb * gpb                   # byte offset of the lengths
        v = sub.view(torch.uint8)
        lens = v[lens0:lens0 + nb * 256].view(nb, 256).clone()
        assert int(lens.max()) <= 32 and int((lens > 0).sum(1).min()) >= 2
        rng = np.random.default_rng(5)
        for trial in range(6):
            bad = sub.clone()
            bv = bad.view(torch.uint8)[lens0:lens0 + nb * 256].view(nb, 256)
            b = int(rng.integers(0, nb))
            present = torch.nonzero(lens[b] > 0).flatten().cpu().numpy()
            if trial % 3 == 0:                              # swap the lengths of two present values
                i, j = rng.choice(present, 2, replace=False)
                bv[b, i], bv[b, j] = lens[b, j], lens[b, i]
            elif trial % 3 == 1:                            # one length off by one
                i = int(rng.choice(present))
                bv[b, i] = lens[b, i] + 1
            else:                                           # the lengths of another block
                bv[b] = lens[(b + 1) % nb].flip(0)
            raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, bad)
            assert raw == n and torch.equal(out[:n], dev(torch, data)), (kind, trial)


def _long_code_data(kind, n, bs):
    """inputs whose trees have codes beyond the decoder's 12-bit table: 13 to 18 bits (second-level
    table), and beyond (the step-by-step path)"""
    rng = np.random.default_rng(len(kind))
    if kind == "zeros_then_zipf":                   # a quarter of every block is one byte value: the Zipf tail gets 13 bits
        data = datagen.zipf255(n).copy()
        view = data[: n - n % bs].reshape(-1, bs)
        view[:, 1000:1000 + bs // 4] = 0
        return data
    if kind == "geometric":                         # p(i) ~ 2^-i: lengths 2, 3, 4, ... up to 24 bits
        w = 0.5 ** np.arange(1, 25)
        return rng.choice(24, size=n, p=w / w.sum()).astype(np.uint8)
    if kind == "rare_bytes":                        # 40 common bytes and 200 bytes that occur a few times per block
        w = np.concatenate([np.full(40, 1.0), np.full(200, 2e-4)])
        return (rng.choice(240, size=n, p=w / w.sum()) + 7).astype(np.uint8)
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["zeros_then_zipf", "geometric", "rare_bytes"])
@pytest.mark.parametrize("bs", [65536, 1 << 20])
def test_codes_beyond_the_first_table(torch_mod, codec, oracle, kind, bs):
    """streams bit-exact with the oracle, sub-index decode exact, and damaged streams decode to the
    same error and bytes with and without the sub-index"""
    torch = torch_mod
    from libhuffman_amd.codec import HuffmanGpuError
    n = 6 * bs + 12345
    data = _long_code_data(kind, n, bs)
    want = oracle.encode(data, bs)
    stream, offs, length, sub = encode_sub(torch, codec, data, bs)
    got = stream[:length].cpu().numpy()
    assert got.size == want.size and np.array_equal(got, want), kind
    raw, out = decode_sub(torch, codec, stream, length, offs, n, bs, sub)
    assert raw == n and torch.equal(out[:n], dev(torch, data)), kind
    nb = codec.block_count(n, bs)
    offs_h = offs.cpu().numpy()
    rng = np.random.default_rng(5)
    for trial in range(8):
        s = stream[:length].clone()
        b = int(rng.integers(0, nb))
        lo, hi = int(offs_h[b]), int(offs_h[b + 1])
        pos = int(rng.integers(lo + (hi - lo) // 3, hi - 8))
        if trial % 2:
            s[pos] ^= 1 << int(rng.integers(0, 8))
        else:
            s[pos:pos + 4] = 0xff
        res, outs = [], []
        for kw in ({}, dict(sub_index=sub, raw_size=n, blocksize=bs)):
            o = torch.zeros(n + bs, dtype=torch.uint8, device="cuda")
            try:
                res.append((0, codec.decode(s, length, offs, nb, o, **kw)))
            except HuffmanGpuError as e:
                res.append((e.err, None))
            outs.append(o)
        assert res[0] == res[1], (kind, trial, res)
        if res[0][0] == 0:
            assert torch.equal(outs[0], outs[1]), (kind, trial)


def test_index_only_decode_lean_and_exact_kernels_agree(torch_mod, codec, oracle):
    """hufgpu_decode() without a sub-index runs decode_fast_kernel (lean, verified) with the exact decoder behind
    it.  Inputs that the lean kernel cannot vouch for - a run of one byte value of many KiB is a periodic bit
    string on which speculative lanes never fall into step, a block whose codes reach 24 bits - must come out
    exactly like everything else; and HUF_GPU_EXACT_DECODE=1 (the exact kernel for every block) must give the
    same bytes in a process of its own."""
    import os
    import subprocess
    import sys
    torch = torch_mod
    bs = 65536
    rng = np.random.default_rng(123)
    parts = [datagen.zipf255(3 * bs)]
    runs = datagen.zipf255(4 * bs).copy()
    runs[1000:1000 + 40000] = 0                       # 40 000 x the shortest code: ~280 lanes of one phase-ambiguous pattern
    runs[bs + 5:bs + 5 + 30000] = 7
    parts.append(runs)
    w = 0.5 ** np.arange(1, 25)
    parts.append(rng.choice(24, size=2 * bs, p=w / w.sum()).astype(np.uint8))      # codes up to 24 bits
    parts.append(datagen.uniform255(2 * bs + 321))
    data = np.concatenate(parts)
    n = data.size
    d = dev(torch, data)
    stream, offs, length = codec.encode(d, bs)
    assert np.array_equal(stream[:length].cpu().numpy(), oracle.encode(data, bs))
    nb = codec.block_count(n, bs)
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, offs, nb, out) == n and torch.equal(out, d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from libhuffman_amd.codec import GpuCodec\n"
        "c = GpuCodec(0); data = torch.from_numpy(np.load(sys.argv[1])).cuda(); bs = 65536\n"
        "s, o, l = c.encode(data, bs); out = torch.zeros_like(data)\n"
        "assert c.decode(s, l, o, c.block_count(data.numel(), bs), out) == data.numel() and torch.equal(out, data)\n"
        "print('exact kernel ok')\n" % root)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "d.npy"), data)
        r = subprocess.run([sys.executable, "-c", child, os.path.join(tmp, "d.npy")], env=dict(os.environ, HUF_GPU_EXACT_DECODE="1"),
                           capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "exact kernel ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_index_only_decode_of_runs(torch_mod, codec):
    """Runs of one byte value are periodic bit strings: the lean decoder settles them from their beginning
    (dfast_run_jump) instead of one lane per round.  Runs of all kinds of lengths and places - inside a lane, across
    waves and segments, one behind the other, of bytes with long codes, alternating pairs (periodic, but not ONE
    codeword) - must come out like everything else."""
    torch = torch_mod
    bs = 65536
    rng = np.random.default_rng(20261003)
    blocks = []
    for b in range(40):
        blk = datagen.zipf255(bs).copy() if b % 3 else rng.integers(0, 255, bs, dtype=np.uint8)     # (255 values: the reference's limit)
        pos = int(rng.integers(0, 3000))
        while pos < bs - 64:
            kind = int(rng.integers(0, 4))
            length = int(rng.choice([40, 300, 2304, 5000, 20000, 48000]))
            length = min(length, bs - pos)
            if kind == 0:
                blk[pos:pos + length] = 0
            elif kind == 1:
                blk[pos:pos + length] = int(rng.integers(0, 255))          # may be a byte with a long code
            elif kind == 2:
                pair = rng.integers(0, 255, 2, dtype=np.uint8)
                blk[pos:pos + length] = np.resize(pair, length)            # ABAB...
            pos += length + int(rng.integers(1, 9000))
        blocks.append(blk)
    data = np.concatenate(blocks)
    d = dev(torch, data)
    stream, offs, length = codec.encode(d, bs)
    nb = codec.block_count(data.size, bs)
    out = torch.zeros(data.size, dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, offs, nb, out) == data.size
    assert torch.equal(out, d)
