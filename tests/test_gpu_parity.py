"""GPU parity tests (run on the MI355X box: pytest -m gpu).

Every test drives the HIP path through the C ABI of libhuffman_amd/libhuffman.so and compares
with (a) the golden vectors captured from the unmodified reference, (b) the CPU oracle on the
same seeded input.  Bit-exact everywhere - the codec is pure integer/byte work.
"""
import hashlib
import os

import numpy as np
import pytest

from libhuffman_amd import datagen

pytestmark = pytest.mark.gpu


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


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


def to_dev(torch, arr):
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.size == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    return torch.from_numpy(a).cuda()


def first_diff(a: np.ndarray, b: np.ndarray) -> str:
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    if d.size == 0:
        return f"sizes {a.size} vs {b.size}, common prefix equal"
    i = int(d[0])
    return (f"sizes {a.size} vs {b.size}; first diff at byte {i}: "
            f"{a[max(0, i - 4): i + 8].tobytes().hex()} vs {b[max(0, i - 4): i + 8].tobytes().hex()}; "
            f"{d.size} bytes differ")


def gpu_encode(torch, codec, data: np.ndarray, blocksize: int):
    d = to_dev(torch, data)
    if d.numel() == 0:
        return np.empty(0, np.uint8), np.zeros(1, np.uint64)
    out, offs, n = codec.encode(d, blocksize)
    return out.cpu().numpy(), offs.cpu().numpy().astype(np.uint64)


def gpu_decode_indexed(torch, codec, stream: np.ndarray, offs: np.ndarray, raw_cap: int, relaxed=False):
    s = to_dev(torch, stream)
    o = torch.from_numpy(offs.astype(np.int64)).cuda()
    out = torch.empty(max(raw_cap, 1), dtype=torch.uint8, device="cuda")
    n = codec.decode(s, stream.size, o, offs.size - 1, out, relaxed=relaxed)
    return out[:n].cpu().numpy()


# ------------------------------------------------------------------------------------------
def test_fill_matches_numpy_generators(torch_mod, codec):
    torch = torch_mod
    n = 262144 + 24
    for kind in ("const41", "uniform256", "uniform255", "zipf255"):
        buf = torch.empty(n, dtype=torch.uint8, device="cuda")
        codec.fill(buf, kind)
        want = datagen.GENERATORS[kind](n)
        got = buf.cpu().numpy()
        assert np.array_equal(got, want), (kind, first_diff(got, want))
        # a shard that starts in the middle of the global sequence
        first = 65536
        part = torch.empty(1000, dtype=torch.uint8, device="cuda")
        codec.fill(part, kind, first=first)
        assert np.array_equal(part.cpu().numpy(), want[first:first + 1000]), kind


def test_histogram_matches_oracle(torch_mod, codec, oracle):
    torch = torch_mod
    for kind, n, bs in (("zipf255", 262144, 65536), ("uniform256", 200000, 65536),
                        ("const41", 70000, 65536), ("uniform255", 12305, 4096), ("zipf255", 5000, 1237)):
        data = datagen.GENERATORS[kind](n)
        hist = codec.histogram(to_dev(torch, data), bs).cpu().numpy().astype(np.uint64)
        for b in range(hist.shape[0]):
            want = oracle.histogram(data[b * bs:(b + 1) * bs])
            assert np.array_equal(hist[b], want), (kind, b)


def test_encode_small_goldens(torch_mod, codec, golden):
    for vec in golden["encode_small"]:
        data = np.frombuffer(bytes.fromhex(vec["input_hex"]), dtype=np.uint8)
        out, offs = gpu_encode(torch_mod, codec, data, vec["blocksize"])
        want = np.frombuffer(bytes.fromhex(vec["output_hex"]), dtype=np.uint8)
        assert np.array_equal(out, want), (vec["name"], first_diff(out, want))


def test_encode_large_goldens(torch_mod, codec, golden, oracle):
    for vec in golden["encode_large"]:
        data = datagen.GENERATORS[vec["generator"]](vec["n"])
        out, offs = gpu_encode(torch_mod, codec, data, vec["blocksize"])
        if sha(out) != vec["output_sha256"]:
            want = oracle.encode(data, vec["blocksize"])
            pytest.fail(f"{vec['generator']} n={vec['n']} bs={vec['blocksize']}: {first_diff(out, want)}")
        assert out.size == vec["output_len"]


@pytest.mark.parametrize("seed", range(10))
def test_encode_random_vs_oracle(torch_mod, codec, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 300000))
    k = int(rng.integers(1, 257))
    alphabet = rng.choice(256, size=k, replace=False)
    p = rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.3, 1.0, 10.0]))))
    data = alphabet[rng.choice(k, size=n, p=p)].astype(np.uint8)
    bs = int(rng.choice([0, 7, 256, 1000, 4096, 4097, 65536, 100000]))
    if bs and n // bs > 3000:
        data = data[: bs * 3000]
    want, woffs = oracle.encode(data, bs, with_offsets=True)
    out, offs = gpu_encode(torch_mod, codec, data, bs)
    assert np.array_equal(out, want), first_diff(out, want)
    assert np.array_equal(offs, woffs)


@pytest.mark.parametrize("chunk", range(4))
def test_fuzz_small_shapes_vs_oracle(torch_mod, codec, oracle, chunk):
    """Many random shapes: odd block sizes, tiny and ragged inputs, every alphabet size, skewed and
    flat distributions.  Encode must equal the oracle's stream; indexed and raw-stream decode
    must give the input back."""
    torch = torch_mod
    rng = np.random.default_rng(9000 + chunk)
    for case in range(60):
        n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 1000, 4095, 4096, 4097, 20000, 70000]))
        n = max(1, n + int(rng.integers(-3, 4)))
        k = int(rng.choice([1, 2, 3, 5, 16, 64, 200, 256]))
        alphabet = rng.choice(256, size=k, replace=False)
        alpha = float(rng.choice([0.02, 0.2, 1.0, 50.0]))
        data = alphabet[rng.choice(k, size=n, p=rng.dirichlet(np.full(k, alpha)))].astype(np.uint8)
        if rng.random() < 0.3:                                   # long runs
            data = np.repeat(data[: max(1, n // 50)], 50)[:n]
            n = data.size
        bs = int(rng.choice([0, 1, 2, 5, 31, 64, 100, 1023, 4096, 65536, 65537]))
        if bs and n // bs > 600:
            bs = n // 600 + 1
        want, woffs = oracle.encode(data, bs, with_offsets=True)
        out, offs = gpu_encode(torch, codec, data, bs)
        assert np.array_equal(out, want), (chunk, case, n, k, bs, first_diff(out, want))
        assert np.array_equal(offs, woffs), (chunk, case, n, k, bs)
        back = gpu_decode_indexed(torch, codec, out, offs, n, relaxed=True)
        assert np.array_equal(back, data), (chunk, case, n, k, bs, first_diff(back, data))
        o = torch.empty(n + 8, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(to_dev(torch, out), out.size, out.size, o, relaxed=True)
        assert (err, raw, used) == (0, n, out.size), (chunk, case, n, k, bs)
        assert np.array_equal(o[:raw].cpu().numpy(), data), (chunk, case, n, k, bs)


def test_one_symbol_blocks_any_alignment(torch_mod, codec, oracle):
    """Blocks of one distinct byte take dedicated paths in pack (zero payload) and decode (fill);
    exercise them at odd block sizes, odd stream offsets and next to ordinary blocks."""
    rng = np.random.default_rng(77)
    for bs in (1, 3, 7, 8, 33, 255, 333, 4096, 5001, 65536):
        parts = []
        for i in range(12):
            if i % 3 == 2:
                parts.append(rng.integers(0, 7, size=bs, dtype=np.uint8))          # ordinary block
            else:
                parts.append(np.full(bs, int(rng.integers(0, 256)), np.uint8))     # one-symbol block
        parts.append(np.full(max(1, bs // 3), 0x41, np.uint8))                     # short one-symbol tail
        data = np.concatenate(parts)
        want, woffs = oracle.encode(data, bs, with_offsets=True)
        out, offs = gpu_encode(torch_mod, codec, data, bs)
        assert np.array_equal(out, want), (bs, first_diff(out, want))
        back = gpu_decode_indexed(torch_mod, codec, out, offs, data.size)
        assert np.array_equal(back, data), (bs, first_diff(back, data))
        # a 1 bit inside an all-zero payload must be reported as a corrupted tree (decoder.c:69-71)
        bad = want.copy()
        pos = int(woffs[0]) + 20                      # first payload byte of block 0 (header is 20 bytes)
        if pos < int(woffs[1]):
            bad[pos] |= 0x80
            torch = torch_mod
            o = torch.empty(data.size + 8, dtype=torch.uint8, device="cuda")
            err, raw, _ = codec.decode_stream(to_dev(torch, bad), bad.size, bad.size, o)
            oerr, oout, _ = oracle.decode(bad, data.size + 8)
            assert err == oerr == 6 and raw == oout.size == 0


def test_one_symbol_payload_stray_bits(torch_mod, codec, oracle):
    """The payload of a one-symbol block is checked 16 bytes at a time for a set bit: first and last
    needed bit, the pad bits behind them (ignored, decoder.c:89-91), any payload alignment, block
    sizes around the byte and 16-byte edges - same error, byte count and bytes as the reference, by
    the indexed kernel and by both raw-stream decoders."""
    import ctypes as C
    torch = torch_mod
    rng = np.random.default_rng(5)
    for bs in (1, 5, 8, 9, 127, 128, 129, 1000, 4099, 65536):
        for lead in (0, 1, 2, 3, 7, 13):                 # an ordinary block in front shifts the alignment
            parts = [rng.integers(0, 5, size=bs, dtype=np.uint8)] if lead else []
            if lead:
                parts[0][:lead] = 200                     # k and the payload length vary with `lead`
            parts += [np.full(bs, 0x30 + i, np.uint8) for i in range(3)]
            data = np.concatenate(parts)
            good, offs = oracle.encode(data, bs, with_offsets=True)
            b = len(parts) - 2                            # the one-symbol block in the middle
            p0, p1 = int(offs[b]) + 20, int(offs[b + 1])
            nbits = (p1 - p0) * 8
            spots = {0, bs - 1, bs, bs + 1, nbits - 1, bs // 2, max(0, bs - 9), min(nbits - 1, 127), min(nbits - 1, 128)}
            spots |= {int(x) for x in rng.integers(0, nbits, size=3)}
            for bit in sorted(x for x in spots if 0 <= x < nbits):
                bad = good.copy()
                bad[p0 + bit // 8] |= 0x80 >> (bit % 8)
                cap = data.size + 32
                oerr, oout, _ = oracle.decode(bad, cap, 1024)
                assert oerr == (6 if bit < bs else 0), (bs, lead, bit, oerr)
                out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
                raw = C.c_uint64(0)
                d_bad, d_offs = to_dev(torch, bad), torch.from_numpy(offs.astype(np.int64)).cuda()
                err = codec.lib.hufgpu_decode(codec._ctx, d_bad.data_ptr(), bad.size, d_offs.data_ptr(), offs.size - 1,
                                              out.data_ptr(), cap, 0, C.byref(raw), None)
                want_raw = oout.size            # indexed too: the blocks in front AND the failing block's symbols before the failure (decoder.c:69-91)
                assert (err, raw.value) == (oerr, want_raw), (bs, lead, bit, err, oerr, raw.value, want_raw)
                assert np.array_equal(out[:oout.size].cpu().numpy(), oout), (bs, lead, bit)
                for sequential in (False, True):
                    out.zero_()
                    err, n, _ = codec.decode_stream(to_dev(torch, bad), bad.size, bad.size, out, sequential=sequential)
                    assert (err, n) == (oerr, oout.size), (bs, lead, bit, sequential, err, oerr, n, oout.size)
                    assert np.array_equal(out[:n].cpu().numpy(), oout), (bs, lead, bit, sequential)


def test_encode_deep_codes(torch_mod, codec, oracle):
    """Fibonacci-weighted inputs give the longest codes a block can have (22 bits at 64 KiB,
    >32 bits needs > 5.7 MB: exercised with one 8 MiB block)."""
    def fib_data(limit):
        f, sym, parts, total = [1, 1], 0, [], 0
        while total + f[-1] <= limit and sym < 250:
            parts.append(np.full(f[-2], sym, np.uint8))
            total += f[-2]
            f.append(f[-1] + f[-2])
            sym += 1
        return np.concatenate(parts)
    for limit, bs in ((65536, 65536), (1 << 20, 1 << 20), (8 << 20, 8 << 20)):
        data = fib_data(limit)
        rng = np.random.default_rng(7)
        rng.shuffle(data)
        want = oracle.encode(data, bs)
        out, offs = gpu_encode(torch_mod, codec, data, bs)
        assert np.array_equal(out, want), (limit, first_diff(out, want))
        back = gpu_decode_indexed(torch_mod, codec, out, offs, data.size)
        assert np.array_equal(back, data), (limit, first_diff(back, data))


def test_blocks_of_4_mib_and_more(torch_mod, codec, oracle):
    """Blocks >= 4 MiB take the unfused kernels (hist256 -> tree with 64-bit keys -> scan_sizes ->
    pack): several such blocks, a one-symbol block among them, a ragged tail; and the block sizes
    just below the switch and around the 16-bit counter limit of the fused kernel (128 KiB)."""
    torch = torch_mod
    for bs in ((5 << 20) + 3, 4 << 20, (4 << 20) - 1, 131072, 131073):
        nblk = 3 if bs >= (4 << 20) - 1 else 5
        parts = [datagen.zipf255(bs), np.full(bs, 0x5a, np.uint8)]
        parts += [datagen.uniform256(bs) for _ in range(nblk - 2)]
        parts.append(datagen.zipf255(bs // 3 + 7))
        data = np.concatenate(parts)
        want, woffs = oracle.encode(data, bs, with_offsets=True)
        out, offs = gpu_encode(torch, codec, data, bs)
        assert np.array_equal(out, want), (bs, first_diff(out, want))
        assert np.array_equal(offs, woffs), bs
        back = gpu_decode_indexed(torch, codec, out, offs, data.size, relaxed=True)
        assert np.array_equal(back, data), (bs, first_diff(back, data))
        o = torch.empty(data.size + 8, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(to_dev(torch, out), out.size, out.size, o, relaxed=True)
        assert (err, raw, used) == (0, data.size, out.size), bs
        assert torch.equal(o[:raw], to_dev(torch, data)), bs


@pytest.mark.parametrize("kind,n,bs", [("const41", 262144, 65536), ("uniform256", 262144, 65536),
                                       ("uniform255", 262144, 65536), ("zipf255", 262144, 65536),
                                       ("zipf255", 2 << 20, 1 << 20), ("logtext", 2 << 20, 1 << 20),
                                       ("zipf255", 66536, 65536), ("uniform255", 12305, 4096),
                                       ("zipf255", 50000, 1237)])
def test_decode_indexed_roundtrip(torch_mod, codec, oracle, kind, n, bs):
    data = datagen.GENERATORS[kind](n)
    stream, offs = oracle.encode(data, bs, with_offsets=True)
    back = gpu_decode_indexed(torch_mod, codec, stream, offs, n, relaxed=True)
    assert np.array_equal(back, data), first_diff(back, data)
    # encode on the GPU, decode on the GPU
    out, goffs = gpu_encode(torch_mod, codec, data, bs)
    back = gpu_decode_indexed(torch_mod, codec, out, goffs, n, relaxed=True)
    assert np.array_equal(back, data)


def test_decode_strict_rejects_k256(torch_mod, codec, oracle):
    from libhuffman_amd.codec import HuffmanGpuError
    data = datagen.uniform256(65536 * 2, 1)
    stream, offs = oracle.encode(data, 65536, with_offsets=True)
    with pytest.raises(HuffmanGpuError) as ei:
        gpu_decode_indexed(torch_mod, codec, stream, offs, data.size, relaxed=False)
    assert ei.value.err == 5          # src/decoder.c:237-239


@pytest.mark.parametrize("seed", range(6))
def test_decode_stream_vs_oracle(torch_mod, codec, oracle, seed):
    torch = torch_mod
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 120000))
    k = int(rng.integers(1, 256))
    alphabet = rng.choice(256, size=k, replace=False)
    data = alphabet[rng.choice(k, size=n, p=rng.dirichlet(np.full(k, 0.4)))].astype(np.uint8)
    bs = int(rng.choice([0, 100, 4096, 65536]))
    if bs == 100:
        data = data[:20000]
    stream = oracle.encode(data, bs)
    out = torch.empty(data.size + 64, dtype=torch.uint8, device="cuda")
    err, raw, used = codec.decode_stream(to_dev(torch, stream), stream.size, stream.size, out)
    assert (err, raw, used) == (0, data.size, stream.size)
    assert np.array_equal(out[:raw].cpu().numpy(), data)


def test_decode_stream_error_goldens(torch_mod, codec, golden):
    torch = torch_mod
    for vec in golden["decode_errors"] + golden["decode_ok"]:
        stream = np.frombuffer(bytes.fromhex(vec["stream_hex"]), dtype=np.uint8)
        length = vec.get("length") or stream.size
        out = torch.empty(4096, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(to_dev(torch, stream), stream.size, length, out)
        assert err == vec["err"], vec["name"]
        assert out[:raw].cpu().numpy().tobytes().hex() == vec["output_hex"], vec["name"]


@pytest.mark.parametrize("kind,n,bs", [("zipf255", 3 << 20, 65536), ("uniform256", 1 << 20, 65536),
                                       ("const41", 4 << 20, 65536), ("logtext", 3 << 20, 1 << 20),
                                       ("zipf255", 700000, 4097)])
def test_decode_stream_parallel_discovery(torch_mod, codec, oracle, kind, n, bs):
    """Raw streams large enough for the parallel block discovery: same bytes, same consumed count
    as the sequential decoder and the oracle."""
    torch = torch_mod
    data = datagen.GENERATORS[kind](n)
    stream = oracle.encode(data, bs)
    s = to_dev(torch, stream)
    for sequential in (False, True):
        out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(s, stream.size, stream.size, out, relaxed=True, sequential=sequential)
        assert (err, raw, used) == (0, n, stream.size), (kind, sequential)
        assert np.array_equal(out[:raw].cpu().numpy(), data)


def _handmade_block(payload: bytes, leaves: int) -> bytes:
    """A block whose tree is complete with 2 or 4 leaves ('A' = 0, 'B' = 1 / 'A'..'D' = 00..11): its payload is ANY byte
    string, one symbol a bit or two (src/decoder.c:34-96 walks whatever tree the stream brings)."""
    import struct
    tree = [0x0101, 0x41, -1, -1, 0x42, -1, -1] if leaves == 2 else \
           [0x0103, 0x0101, 0x41, -1, -1, 0x42, -1, -1, 0x0102, 0x43, -1, -1, 0x44, -1, -1]
    per_byte = 8 if leaves == 2 else 4
    return struct.pack("<Qh", per_byte * len(payload), len(tree)) + b"".join(struct.pack("<h", v) for v in tree) + payload


@pytest.mark.parametrize("leaves,pay_bytes,at,fake_len,small_blocks",
                         [(2, 8192, 3000, 1, 0), (4, 8192, 3000, 1, 0), (4, 60000, 50000, 1, 0), (4, 60000, 17, 1, 0), (2, 60000, 31000, 1, 0),
                          (4, 8192, 3000, 1000000, 0), (4, 8192, 3000, 1, 8190), (2, 8192, 100, 1, 8191)])
def test_raw_stream_with_a_false_header_inside_a_payload(torch_mod, codec, oracle, leaves, pay_bytes, at, fake_len, small_blocks):
    """The discovery takes every offset that LOOKS like a header for a candidate, and a candidate's probe takes the next
    candidate's offset for the end of its own payload (a guess, given up when the symbols come short).  Here a payload holds
    the twelve bytes of a syntactically valid header - block_len 1, a tree of one marker - in the first and in a later
    segment of its block and right behind the real header; blocks of an ordinary encoder stand in front and behind.  The
    chain of blocks then has a link that jumps over a candidate: round 4 found that walk_kernel never followed such a
    link (it wrote block offsets until its array ended: a memory fault on a VALID stream).  Bytes, error code and consumed
    count are the oracle's, with the parallel discovery and without.  (The tree of seven entries takes the exact decoder
    in the probe, the one of fifteen the lean decoder and its guess.  fake_len = 1 000 000: the candidates' claimed
    lengths no longer fit the output, and the probes only count.  small_blocks: that many 256-byte blocks in front, so
    that the link that jumps lies at the seam of the walk's chunks of 8 192 candidates.)"""
    import struct
    torch = torch_mod
    rng = np.random.default_rng(77 + at)
    pay = bytearray(rng.integers(0, 256, pay_bytes, dtype=np.uint8).tobytes())
    pay[at:at + 12] = struct.pack("<Qh", fake_len, 1) + b"\xff\xff"
    tail = oracle.encode(datagen.zipf255(5 * 65536), 65536)
    if small_blocks:
        head_data = datagen.zipf255(small_blocks * 256)
        head = oracle.encode(head_data, 256)
    else:
        head_data = datagen.uniform256(2 * 65536)
        head = oracle.encode(head_data, 65536)
    stream = np.concatenate([head, np.frombuffer(_handmade_block(bytes(pay), leaves), dtype=np.uint8), tail])
    n = head_data.size + 5 * 65536 + (8 if leaves == 2 else 4) * pay_bytes
    cap = n + 64
    oerr, oout, oused = oracle.decode(stream, cap, 1025)
    assert (oerr, oout.size, oused) == (0, n, stream.size)
    s = to_dev(torch, stream)
    for sequential in (False, True):
        out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(s, stream.size, stream.size, out, relaxed=True, sequential=sequential)
        assert (err, raw, used) == (oerr, oout.size, oused), (sequential, err, raw, used)
        assert np.array_equal(out[:raw].cpu().numpy(), oout), sequential


@pytest.mark.parametrize("seed", range(10))
def test_streams_of_random_trees_vs_oracle(torch_mod, codec, oracle, seed):
    """Blocks no encoder wrote (tests/handmade_streams.py): trees of any shape and depth (codes of 1 to > 100 bits, with
    and without the encoder's one-child root), symbols coded with them, padding of zeros or ones.  The raw stream - parallel
    discovery and in order - and the indexed decode deliver what the oracle delivers."""
    import handmade_streams as hm
    torch = torch_mod
    rng = np.random.default_rng(9000 + seed)
    parts, want, deepest = [], [], 0
    total = 0
    while total < (150000 if seed % 2 == 0 else 20000):
        leaves = int(rng.integers(2, 257))
        skew = float(rng.choice([0.0, 0.3, 0.8, 0.97]))
        nsym = int(rng.choice([1, 7, 300, 5000, 40000, 70000]))
        b, syms, depth = hm.block(rng, leaves, skew, bool(rng.integers(0, 2)), nsym, deep_often=bool(rng.integers(0, 2)),
                                  pad_ones=bool(rng.integers(0, 2)))
        parts.append(np.frombuffer(b, dtype=np.uint8))
        want.append(syms)
        deepest = max(deepest, depth)
        total += len(b)
    stream = np.concatenate(parts)
    data = np.concatenate(want)
    oerr, oout, oused = oracle.decode(stream, data.size + 64, 1025)
    assert (oerr, oused) == (0, stream.size) and np.array_equal(oout, data), "the generator and the oracle disagree"
    s = to_dev(torch, stream)
    for sequential in (False, True):
        out = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
        err, raw, used = codec.decode_stream(s, stream.size, stream.size, out, relaxed=True, sequential=sequential)
        assert (err, raw, used) == (0, data.size, stream.size), (seed, sequential, deepest, err, raw, used)
        assert np.array_equal(out[:raw].cpu().numpy(), data), (seed, sequential, deepest)
    offs = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
    got = gpu_decode_indexed(torch, codec, stream, offs, data.size, relaxed=True)
    assert np.array_equal(got, data), (seed, "indexed", deepest)


@pytest.mark.parametrize("seed", range(6))
def test_damaged_streams_of_random_trees_vs_oracle(torch_mod, codec, oracle, seed):
    """The same hand-made streams with a bit flipped, a byte overwritten, the end cut off or the length given short: error code
    and bytes delivered are the oracle's (src/decoder.c:69-91, 218-276), raw parallel and raw in order."""
    import handmade_streams as hm
    torch = torch_mod
    rng = np.random.default_rng(12000 + seed)
    parts, sizes = [], 0
    while sizes < 120000:
        b, syms, _ = hm.block(rng, int(rng.integers(2, 257)), float(rng.choice([0.0, 0.3, 0.8, 0.97])), bool(rng.integers(0, 2)),
                              int(rng.choice([7, 300, 5000, 40000])), deep_often=bool(rng.integers(0, 2)), pad_ones=bool(rng.integers(0, 2)))
        parts.append(np.frombuffer(b, dtype=np.uint8))
        sizes += len(b)
    good = np.concatenate(parts)
    starts = np.concatenate([[0], np.cumsum([p.size for p in parts])])
    cap = int(sum(int.from_bytes(p[:8].tobytes(), "little") for p in parts)) + 70000
    cases = []
    for _ in range(5):
        b = good.copy(); i = int(rng.integers(0, good.size)); b[i] ^= 1 << int(rng.integers(0, 8)); cases.append(("bit %d" % i, b, None))
    for _ in range(3):
        blk = int(rng.integers(0, len(parts)))
        b = good.copy(); i = int(starts[blk]) + int(rng.integers(0, 14)); b[i] = int(rng.integers(0, 256)); cases.append(("header byte %d" % i, b, None))
    cut = int(rng.integers(good.size // 2, good.size))
    cases.append(("cut at %d" % cut, good[:cut].copy(), None))
    cases.append(("length %d" % cut, good.copy(), cut))
    for name, bad, length in cases:
        oerr, oout, oused = oracle.decode(bad, cap, 1025, length=length)
        if oerr == 1:
            continue                                            # (more than the test's buffer: a damaged block_len)
        s = to_dev(torch, bad)
        ln = bad.size if length is None else length
        for sequential in (False, True):
            out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            err, raw, used = codec.decode_stream(s, bad.size, ln, out, relaxed=True, sequential=sequential)
            assert (err, raw) == (oerr, oout.size), (seed, name, sequential, (err, raw, used), (oerr, oout.size, oused))
            assert np.array_equal(out[:raw].cpu().numpy(), oout), (seed, name, sequential)
            if oerr == 0:                                       # (after an error: what the reader had buffered, not a property of the stream)
                assert used == oused, (seed, name, sequential)


def test_decode_stream_parallel_error_parity(torch_mod, codec, oracle):
    """Corruptions in the middle of a long stream: the parallel path hands the unvalidated rest to
    the in-order decoder, so error code, delivered bytes and consumed count equal the oracle's."""
    torch = torch_mod
    rng = np.random.default_rng(4242)
    data = datagen.zipf255(40 * 8192)
    good, offs = oracle.encode(data, 8192, with_offsets=True)
    cases = []
    for blk in (0, 7, 20, 39):
        o0, o1 = int(offs[blk]), int(offs[blk + 1])
        b = good.copy(); b[o0 + 9] = 0x7f; cases.append(("tree_len_overflow", b, None))       # tree_len -> 0x7fxx
        b = good.copy(); b[o0 + 12] ^= 0xff; b[o0 + 13] ^= 0xff; cases.append(("tree_entry_flip", b, None))
        b = good.copy(); b[(o0 + o1) // 2] ^= 0x10; cases.append(("payload_bit_flip", b, None))
        b = good.copy(); b[o0] ^= 0x01; cases.append(("block_len_changed", b, None))
    cases.append(("truncated", good[: int(offs[33]) + 700].copy(), None))
    cases.append(("trailing_garbage", np.concatenate([good, np.full(5, 0xee, np.uint8)]), None))
    cases.append(("length_short", good.copy(), int(offs[11])))
    cases.append(("length_mid_block", good.copy(), int(offs[11]) + 5))
    for name, bad, length in cases:
        oerr, oout, oused = oracle.decode(bad, data.size + 70000, 1024, length=length)
        if oerr == 1:
            continue                                            # larger than the oracle's test buffer
        out = torch.zeros(data.size + 70000, dtype=torch.uint8, device="cuda")
        ln = bad.size if length is None else length
        for sequential in (False, True):
            err, raw, used = codec.decode_stream(to_dev(torch, bad), bad.size, ln, out, sequential=sequential)
            assert err == oerr, (name, sequential, err, oerr)
            assert raw == oout.size, (name, sequential, raw, oout.size)
            assert np.array_equal(out[:raw].cpu().numpy(), oout), (name, sequential)
            if oerr == 0:
                assert used == oused, (name, sequential)


@pytest.mark.parametrize("nblocks,bs", [(255, 512), (256, 512), (257, 512), (601, 1000), (1024, 300), (5000, 64)])
def test_block_offsets_across_groups(torch_mod, codec, oracle, nblocks, bs):
    """The stream offsets are summed inside the kernels in groups of 256 blocks (two-level
    prefix): group edges, a short last group, a short last block, and repeated calls (the
    ticket counters must be back at zero)."""
    torch = torch_mod
    n = (nblocks - 1) * bs + max(1, bs // 3)
    rng = np.random.default_rng(nblocks * 7 + bs)
    # blocks of very different entropy, so that the sizes differ a lot from block to block
    data = np.where(rng.random(n) < 0.5, rng.integers(0, 4, n), rng.integers(0, 256, n)).astype(np.uint8)
    data[: n // 3] = 65
    want, woffs = oracle.encode(data, bs, with_offsets=True)
    for rep in range(2):
        got, offs = gpu_encode(torch, codec, data, bs)
        assert offs.size == nblocks + 1
        assert np.array_equal(offs, woffs), first_diff(offs.view(np.uint8), woffs.astype(np.uint64).view(np.uint8))
        assert np.array_equal(got, want), first_diff(got, want)
        back = gpu_decode_indexed(torch, codec, got, offs, n, relaxed=True)
        assert np.array_equal(back, data), first_diff(back, data)


def test_block_index_alternating_inputs(torch_mod, codec, oracle):
    """Two inputs of the same shape but different block sizes, encoded alternately: a block size
    left over from the previous launch (the ticket of the in-kernel prefix sum overtaking the
    value it announces - tests/stress/stress_encode.py found 3 such launches in 300 000 before the
    s_waitcnt fix) would show up as a wrong index entry.  A cheap guard, not a proof."""
    torch = torch_mod
    rng = np.random.default_rng(99)
    n, bs = 700 * 64 + 5, 64
    inputs = []
    for k in (2, 40):
        data = rng.integers(0, k, n).astype(np.uint8)
        want, woffs = oracle.encode(data, bs, with_offsets=True)
        inputs.append((to_dev(torch, data), to_dev(torch, want), torch.from_numpy(woffs.astype(np.int64)).cuda()))
    for it in range(1500):
        d, ws, wo = inputs[it & 1]
        got, offs, ln = codec.encode(d, bs)
        assert ln == ws.numel() and torch.equal(offs, wo) and torch.equal(got, ws), it


@pytest.mark.parametrize("kind", ["zipf255", "uniform255"])
def test_payload_walk_leaves_tree_mid_block(torch_mod, codec, oracle, kind):
    """A byte of ones in the payload: some codeword then starts with 1, which leaves an
    encoder-made tree at the wrap root (src/decoder.c:69-71 -> error 6, bytes before it delivered).
    Every speculative lane of the GPU decoder meets such bits all the time, so the real one has to
    be told apart wherever it sits: any lane, any segment, before or after a lane's false alarms."""
    torch = torch_mod
    bs = 65536
    data = datagen.GENERATORS[kind](3 * bs + 1234)
    good, offs = oracle.encode(data, bs, with_offsets=True)
    rng = np.random.default_rng(77)
    spots = []
    for blk in (0, 1, 2, 3):
        o0, o1 = int(offs[blk]) + 10 + 2 * 1021, int(offs[blk + 1])
        span = o1 - o0
        spots += [o0 + 3, o0 + span // 2, o1 - 2, o0 + 16384 - 1, o0 + 16384 + 5]      # incl. around a segment seam
        spots += [o0 + int(x) for x in rng.integers(0, span, size=6)]
    out = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
    seen = set()
    for spot in spots:
        if not (0 <= spot < good.size):
            continue
        bad = good.copy()
        bad[spot] = 0xff
        oerr, oout, oused = oracle.decode(bad, data.size + 64, 1024)
        seen.add(oerr)
        for sequential in (False, True):
            err, raw, used = codec.decode_stream(to_dev(torch, bad), bad.size, bad.size, out, sequential=sequential)
            assert (err, raw) == (oerr, oout.size), (kind, spot, sequential, err, oerr, raw, oout.size)
            assert np.array_equal(out[:raw].cpu().numpy(), oout), (kind, spot, sequential)
    assert 6 in seen


def test_damaged_block_len_fields(torch_mod, codec, oracle):
    """A block_len larger than the block really is (found by tests/stress/soak.py): far beyond what the
    payload can hold -> decode what is there, then error 3 like the reference's reader at the end
    of its input; a little too large -> the walk runs into the next header and fails there (6 or
    3) - also when the claimed length would not fit the caller's output buffer."""
    torch = torch_mod
    data = datagen.zipf255(40000)
    good, offs = oracle.encode(data, 4096, with_offsets=True)
    cases = []
    for blk in (0, 3, 9):
        o0 = int(offs[blk])
        b = good.copy(); b[o0 + 5] = 0xff; cases.append(b)                      # ~2^47 symbols
        b = good.copy(); b[o0 + 3] = 0x01; cases.append(b)                      # +16 Mi symbols
        b = good.copy(); b[o0] = (int(b[o0]) + 70) & 0xff; cases.append(b)      # a few dozen too many / too few
        b = good.copy(); b[o0 + 1] ^= 0x40; cases.append(b)                     # +-16 Ki symbols
    cap = data.size + 64
    for i, bad in enumerate(cases):
        oerr, oout, _ = oracle.decode(bad, cap, 1024)
        for sequential in (False, True):
            out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            err, raw, _ = codec.decode_stream(to_dev(torch, bad), bad.size, bad.size, out, sequential=sequential)
            assert err == oerr, (i, sequential, err, oerr)
            if oerr != 1:                                                       # (1 = the buffer is the limit)
                assert raw == oout.size and np.array_equal(out[:raw].cpu().numpy(), oout), (i, sequential, raw, oout.size)


def test_self_synchronisation_worst_cases(torch_mod, codec, oracle):
    """Inputs on which speculative starts do not re-synchronise by themselves: long runs of
    one symbol whose code is longer than a bit, and fixed-length codes."""
    cases = []
    a = np.zeros(70000, np.uint8); a[::997] = 1; a[5::4999] = 2          # runs of a 2-bit code
    cases.append(a)
    cases.append(np.tile(np.arange(4, dtype=np.uint8), 20000))           # 3-bit fixed length, periodic
    cases.append(np.repeat(np.arange(16, dtype=np.uint8), 4096))         # 5-bit fixed, long runs
    b = np.full(65536, 7, np.uint8); b[-1] = 9
    cases.append(b)
    for data in cases:
        stream, offs = oracle.encode(data, 65536, with_offsets=True)
        back = gpu_decode_indexed(torch_mod, codec, stream, offs, data.size)
        assert np.array_equal(back, data), first_diff(back, data)


def test_c_api_fd_streams(torch_mod, oracle, tmp_path, monkeypatch):
    """huf_fdopen streams (src/io.c:9-63): their read(2)/write(2) run on helper threads next to the
    GPU work, two rounds in flight.  Same bytes as the reference for file -> file, file -> memory,
    memory -> file and a pipe; a short input fails with error 3 after the complete rounds went
    out; a descriptor that cannot be written fails with error 3; decode file -> file restores."""
    import ctypes as C
    import threading
    from libhuffman_amd import _native as N
    L = N.load()
    monkeypatch.setenv("HUF_GPU_BATCH_MB", "1")           # many rounds on a small input
    bs = 65536
    data = datagen.zipf255(5 * (1 << 20) + 12345)
    want = oracle.encode(data, bs)
    src = tmp_path / "in.bin"
    src.write_bytes(data.tobytes())

    def fdopen(fd):
        rw = C.POINTER(N.ReadWriter)()
        assert L.huf_fdopen(C.byref(rw), fd) == 0
        return rw

    def memopen(cap=16):
        rw, buf = C.POINTER(N.ReadWriter)(), C.c_void_p()
        assert L.huf_memopen(C.byref(rw), C.byref(buf), cap) == 0
        return rw, buf

    def memcontent(rw, buf):
        n = C.c_size_t()
        L.huf_memlen(rw, C.byref(n))
        return C.string_at(buf.value, n.value)

    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    # file -> file
    fin, fout = os.open(src, os.O_RDONLY), os.open(tmp_path / "out.hm", os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    rin, rout = fdopen(fin), fdopen(fout)
    assert L.huf_encode(C.byref(N.Config(data.size, bs, 0, 0, rin, rout))) == 0
    assert (tmp_path / "out.hm").read_bytes() == want.tobytes()
    assert os.lseek(fin, 0, os.SEEK_CUR) == data.size
    # decode file -> file
    os.lseek(fout, 0, os.SEEK_SET)
    fback = os.open(tmp_path / "back.bin", os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    rback = fdopen(fback)
    assert L.huf_decode(C.byref(N.Config(want.size, 0, 0, 0, rout, rback))) == 0
    assert (tmp_path / "back.bin").read_bytes() == data.tobytes()
    for rw, fd in ((rin, fin), (rout, fout), (rback, fback)):
        L.huf_fdclose(C.byref(rw)); os.close(fd)

    # decode in rounds (1 MiB of stream each) against the reference on damaged and cut streams:
    # error, delivered bytes; a `length` inside a block finishes that block (decoder.c:218)
    cases = [("cut mid-block", want[: want.size - 777], want.size - 777),
             ("length mid-block", want, int(want.size * 0.6)),
             ("length = 1", want, 1),
             ("trailing garbage", np.concatenate([want, np.frombuffer(b"\x01" * 40, np.uint8)]), want.size + 40)]
    for spot in (100, want.size // 3, want.size - 5000):
        bad = want.copy()
        bad[spot] = 0xff
        cases.append((f"damaged byte {spot}", bad, want.size))
    for name, stream, length in cases:
        (tmp_path / "case.hm").write_bytes(stream.tobytes())
        oerr, oout, _ = oracle.decode(stream, data.size + 64, 1024, length=length)
        fin, fout = os.open(tmp_path / "case.hm", os.O_RDONLY), os.open(tmp_path / "case.out", os.O_CREAT | os.O_RDWR | os.O_TRUNC)
        rin, rout = fdopen(fin), fdopen(fout)
        err = L.huf_decode(C.byref(N.Config(length, 0, 0, 0, rin, rout)))
        got = (tmp_path / "case.out").read_bytes()
        assert err == oerr and got == oout.tobytes(), (name, err, oerr, len(got), oout.size)
        for rw, fd in ((rin, fin), (rout, fout)):
            L.huf_fdclose(C.byref(rw)); os.close(fd)

    # file -> memory, memory -> file
    fin = os.open(src, os.O_RDONLY)
    rin = fdopen(fin)
    rmem, bmem = memopen()
    assert L.huf_encode(C.byref(N.Config(data.size, bs, 0, 0, rin, rmem))) == 0
    assert memcontent(rmem, bmem) == want.tobytes()
    L.huf_fdclose(C.byref(rin)); os.close(fin)
    fin = os.open(tmp_path / "out.hm", os.O_RDONLY)          # decode file -> memory
    rin = fdopen(fin)
    rdec, bdec = memopen()
    assert L.huf_decode(C.byref(N.Config(want.size, 0, 0, 0, rin, rdec))) == 0
    assert memcontent(rdec, bdec) == data.tobytes()
    L.huf_fdclose(C.byref(rin)); os.close(fin)
    L.huf_memclose(C.byref(rdec)); libc.free(bdec)
    rsrc, bsrc = memopen()
    assert rsrc.contents.write(rsrc.contents.stream, data.ctypes.data_as(C.c_void_p), data.size) == 0
    fout = os.open(tmp_path / "out2.hm", os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    rout = fdopen(fout)
    assert L.huf_encode(C.byref(N.Config(data.size, bs, 0, 0, rsrc, rout))) == 0
    assert (tmp_path / "out2.hm").read_bytes() == want.tobytes()
    L.huf_fdclose(C.byref(rout)); os.close(fout)
    for rw, b in ((rmem, bmem), (rsrc, bsrc)):
        L.huf_memclose(C.byref(rw)); libc.free(b)

    # pipe -> memory: reads come back in pieces
    pr, pw = os.pipe()
    feeder = threading.Thread(target=lambda: (os.write(pw, data.tobytes()), os.close(pw)))
    feeder.start()
    rin = fdopen(pr)
    rmem, bmem = memopen()
    assert L.huf_encode(C.byref(N.Config(data.size, bs, 0, 0, rin, rmem))) == 0
    feeder.join()
    assert memcontent(rmem, bmem) == want.tobytes()
    L.huf_fdclose(C.byref(rin)); os.close(pr)
    L.huf_memclose(C.byref(rmem)); libc.free(bmem)

    # the input ends early: error 3, the complete rounds (1 MiB each) are on the descriptor
    fin, fout = os.open(src, os.O_RDONLY), os.open(tmp_path / "short.hm", os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    rin, rout = fdopen(fin), fdopen(fout)
    assert L.huf_encode(C.byref(N.Config(data.size + 1000, bs, 0, 0, rin, rout))) == 3
    assert (tmp_path / "short.hm").read_bytes() == oracle.encode(data[: 5 << 20], bs).tobytes()
    L.huf_fdclose(C.byref(rout)); os.close(fout)
    # a descriptor that takes no writes: error 3
    os.lseek(fin, 0, os.SEEK_SET)
    fro = os.open(tmp_path / "short.hm", os.O_RDONLY)
    rout = fdopen(fro)
    assert L.huf_encode(C.byref(N.Config(data.size, bs, 0, 0, rin, rout))) == 3
    for rw, fd in ((rin, fin), (rout, fro)):
        L.huf_fdclose(C.byref(rw)); os.close(fd)


def test_c_api_roundtrip_readme_example(torch_mod, golden):
    """README.md:37-104 / BASELINE config 1 through the drop-in C API: huf_memopen streams,
    huf_encode then huf_decode with swapped streams."""
    import ctypes as C
    from libhuffman_amd import _native as N
    L = N.load()
    for data, bs, rb, wb in ((b"0123456789", 65536, 0, 0), (b"0123456789", 0, 128, 128),
                             (b"abcabc", 4, 3, 5), (b"a" * 1000, 131072, 0, 0)):
        rin, rout = C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)()
        bin_, bout = C.c_void_p(), C.c_void_p()
        assert L.huf_memopen(C.byref(rin), C.byref(bin_), 16) == 0
        assert L.huf_memopen(C.byref(rout), C.byref(bout), 16) == 0
        assert rin.contents.write(rin.contents.stream, data, len(data)) == 0
        cfg = N.Config(len(data), bs, rb, wb, rin, rout)
        assert L.huf_encode(C.byref(cfg)) == 0
        n = C.c_size_t()
        L.huf_memlen(rout, C.byref(n))
        enc = C.string_at(bout.value, n.value)
        want = next(v for v in golden["encode_small"]
                    if bytes.fromhex(v["input_hex"]) == data and v["blocksize"] == bs)
        assert enc.hex() == want["output_hex"]
        # decode: the encoded stream becomes the reader, a fresh stream the writer
        rback, bback = C.POINTER(N.ReadWriter)(), C.c_void_p()
        assert L.huf_memopen(C.byref(rback), C.byref(bback), 16) == 0
        dcfg = N.Config(n.value, 0, rb, wb, rout, rback)
        assert L.huf_decode(C.byref(dcfg)) == 0
        L.huf_memlen(rback, C.byref(n))
        assert C.string_at(bback.value, n.value) == data
        for r in (rin, rout, rback):
            L.huf_memclose(C.byref(r))
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        for b in (bin_, bout, bback):
            libc.free(b)


def test_full_size_config2_const41(torch_mod, codec):
    """BASELINE config 2 at full size: 1 GiB of 0x41, 64 KiB blocks. Size-independent checks:
    exact stream length 16384 * 8212, every block record identical, round trip."""
    torch = torch_mod
    n, bs = 1 << 30, 65536
    data = torch.full((n,), 0x41, dtype=torch.uint8, device="cuda")
    out, offs, length = codec.encode(data, bs)
    assert length == 16384 * 8212
    rec = out[:length].view(16384, 8212)
    assert bool((rec == rec[0:1]).all())
    head = rec[0, :20].cpu().numpy().tobytes().hex()
    assert head == "0000010000000000" + "0500" + "0001" + "4100" + "ffffffffffff"
    assert int(rec[0, 20:].max()) == 0
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(out, length, offs, 16384, back) == n
    assert torch.equal(back, data)


@pytest.mark.parametrize("kind", ["zipf255", "uniform256"])
def test_full_size_roundtrip_and_block_parity(torch_mod, codec, oracle, kind):
    """BASELINE configs 3/4 at 1 GiB: GPU round trip, plus bit-exact comparison of a sample of
    blocks against the oracle (blocks are independent, so any block can be checked alone)."""
    torch = torch_mod
    n, bs = 1 << 30, 65536
    data = torch.empty(n, dtype=torch.uint8, device="cuda")
    codec.fill(data, kind)
    out, offs, length = codec.encode(data, bs)
    offs_h = offs.cpu().numpy()
    for b in (0, 1, 4095, 8191, 16383):
        block = data[b * bs:(b + 1) * bs].cpu().numpy()
        want = oracle.encode(block, bs)
        got = out[int(offs_h[b]): int(offs_h[b + 1])].cpu().numpy()
        assert np.array_equal(got, want), (b, first_diff(got, want))
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(out, length, offs, 16384, back, relaxed=True) == n
    assert torch.equal(back, data)


# ------------------------------------------------------------------------------------------
# round 2: the parity corners of VERDICT r01
def _tree_depths(stream: np.ndarray):
    """Leaf depths (= code lengths) of the first block's serialized tree, by a plain preorder walk."""
    tl = int(np.frombuffer(stream[8:10].tobytes(), dtype="<i2")[0])
    ent = np.frombuffer(stream[10:10 + 2 * tl].tobytes(), dtype="<i2")
    depths, pos = [], 0
    stack = [0]                                   # depth of the slot the next entry fills
    while stack and pos < tl:
        d = stack.pop()
        v = int(ent[pos]); pos += 1
        if v == -1:
            continue
        left_null = pos < tl and ent[pos] == -1
        # a node: its right slot, then its left slot (left is filled first)
        stack.append(d + 1)
        stack.append(d + 1)
        if left_null and pos + 1 < tl and ent[pos + 1] == -1:
            depths.append(d)
    return depths


def _fib_block(n: int, lead_ones: int) -> np.ndarray:
    """n bytes whose counts are lead_ones x 1, then 2, 3, 5, 8, ... (the last count takes the rest):
    the counts that make a Huffman tree as deep as n bytes allow."""
    counts = [1] * lead_ones
    a, b = 1, 2
    while sum(counts) + b <= n:
        counts.append(b)
        a, b = b, a + b
    counts[-1] += n - sum(counts)
    return np.repeat(np.arange(len(counts), dtype=np.uint8), counts)


@pytest.mark.parametrize("bs,lead", [(121392, 2), (121393, 3), (121393, 2), (121392, 3)])
def test_deepest_codes_at_the_short_pack_switch(torch_mod, codec, oracle, bs, lead):
    """hufgpu_encode takes the 32-bit-code pack kernel up to blocksize 121 392 (F(26) - 1: no code
    longer than 23 + the wrap bit) and the 64-bit one above.  Fibonacci counts that fill exactly
    such a block reach the longest code on either side of the switch."""
    rng = np.random.default_rng(bs + lead)
    block = _fib_block(bs, lead)
    data = np.concatenate([rng.permutation(block), block[::-1], rng.permutation(block)[: bs // 3]])
    want = oracle.encode(data, bs)
    out, offs = gpu_encode(torch_mod, codec, data, bs)
    assert np.array_equal(out, want), (bs, lead, first_diff(out, want))
    deepest = max(_tree_depths(want))
    if bs == 121392:
        assert deepest == 24, deepest                 # the bound the SHORT kernel is built on, reached
    else:
        assert deepest >= 24, deepest
    back = gpu_decode_indexed(torch_mod, codec, out, offs, data.size)
    assert np.array_equal(back, data)
    # and through the sub-index path
    torch = torch_mod
    d = to_dev(torch, data)
    sub = codec.new_sub_index(d.numel(), bs)
    stream, o, length = codec.encode(d, bs, sub_index=sub)
    assert np.array_equal(stream.cpu().numpy(), want)
    res = torch.empty(d.numel(), dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, o, codec.block_count(d.numel(), bs), res, sub_index=sub,
                        raw_size=d.numel(), blocksize=bs) == d.numel()
    assert torch.equal(res, d)


@pytest.mark.parametrize("kind,bs", [("zipf255", 65536), ("uniform256", 65536), ("const41", 65536), ("logtext", 1 << 20)])
def test_full_size_stream_equals_oracle_stream(torch_mod, codec, oracle, kind, bs):
    """BASELINE.md's bit-exactness gate at FULL size: every byte of the 1 GiB GPU stream - as written by the
    encode bench.py times, the one that also writes the sub-index - against the CPU oracle's stream of the
    same input; the plain encode must give the same bytes, both timed decodes the input.  Blocks are independent, so the host cores each encode a
    block-aligned slice (ctypes releases the GIL) and the slices are compared where they lie; the
    sha256 of both whole streams is compared on top."""
    from concurrent.futures import ThreadPoolExecutor
    torch = torch_mod
    n = 1 << 30
    if kind == "logtext":
        tile = torch.from_numpy(datagen.logtext(16 << 20)).cuda()
        data = tile.repeat(n // tile.numel())
    else:
        data = torch.empty(n, dtype=torch.uint8, device="cuda")
        codec.fill(data, kind)
    # the encode bench.py times: WITH the sub-index (pack_kernel's extra stores inside its tile loop) ...
    sub = codec.new_sub_index(n, bs)
    out, offs, length = codec.encode(data, bs, sub_index=sub)
    # ... the plain encode must give the same bytes and the same block index
    out2, offs2, length2 = codec.encode(data, bs)
    assert length2 == length and torch.equal(out2[:length2], out[:length]) and torch.equal(offs2, offs)
    del out2, offs2
    # ... and the decodes bench.py times: with the sub-index, and with the block index alone
    nblk = codec.block_count(n, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert codec.decode(out, length, offs, nblk, back, relaxed=True, sub_index=sub, raw_size=n, blocksize=bs) == n
    assert torch.equal(back, data)
    back.zero_()
    assert codec.decode(out, length, offs, nblk, back, relaxed=True) == n
    assert torch.equal(back, data)
    del back, sub
    got = out[:length].cpu().numpy()
    offs_h = offs.cpu().numpy()
    host = data.cpu().numpy()
    del data, out
    ncpu = max(1, os.cpu_count() or 1)
    nb = n // bs
    per = max(1, nb // (4 * ncpu))                                  # blocks per slice
    starts = list(range(0, nb, per))

    def one(b0):
        b1 = min(nb, b0 + per)
        want = oracle.encode(host[b0 * bs:b1 * bs], bs)
        lo, hi = int(offs_h[b0]), int(offs_h[b1])
        if want.size != hi - lo or not np.array_equal(got[lo:hi], want):
            return (b0, first_diff(got[lo:hi], want))
        return hashlib.sha256(want.tobytes())

    with ThreadPoolExecutor(ncpu) as ex:
        res = list(ex.map(one, starts))
    bad = [r for r in res if isinstance(r, tuple)]
    assert not bad, bad[:3]
    # whole-stream digests: the oracle's from its slices in order, the GPU's from its buffer
    h = hashlib.sha256()
    for b0 in starts:
        b1 = min(nb, b0 + per)
        h.update(got[int(offs_h[b0]):int(offs_h[b1])].tobytes())
    assert h.hexdigest() == hashlib.sha256(got.tobytes()).hexdigest()
    assert int(offs_h[nb]) == length


def test_bench_runs_over_rccl_with_one_rank(torch_mod):
    """bench.py's distributed path on this one GPU: nccl init, the size all-gather inside the timed
    step, the root-placement legs (variable-size all-to-alls).  N > 1 runs only on the driver's
    8-GPU node; this pins everything but the second rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--bytes-per-gpu", str(64 << 20), "--secondary", "const41", "--no-cpu-baseline", "--no-live-traffic"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["bit_exact_roundtrip"] is True
    assert d["config"]["generator"] == "zipf255" and "const41" in d["secondary"]
    rp = d["root_placement"]
    assert "error" not in rp, rp
    assert rp["bit_exact_roundtrip"] is True and rp["value"] > 0
    assert rp["mover"].startswith("hufgpu_encode_sharded"), rp["mover"]      # the C library's own RCCL calls, not torch's
    assert set(rp["legs_ms_rank0"]) >= {"scatter_in", "encode", "gather_stream", "scatter_stream", "decode", "gather_out"}


def test_bench_configs3_share_over_rccl_with_one_rank(torch_mod):
    """BASELINE configs[3] is 16 GiB of uniform bytes over 8 GPUs = 2 GiB per rank: that share - its allocations (2 GiB
    in, 2.3 GiB of stream bound, the sub-index, 2 GiB out), 32 768 blocks of k = 256 in relaxed-tree mode - over RCCL with
    the one rank this box has, per-rank times and the size all-gather's cost in the record."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "uniform256",
                          "--bytes-per-gpu", str(2 << 30), "--secondary", "none", "--no-cpu-baseline", "--no-live-traffic",
                          "--no-index-free", "--no-python-layer", "--no-other-decode", "--placement", "resident"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["bytes_per_gpu"] == 2 << 30 and d["config"]["blocks_per_gpu"] == 32768
    assert d["config"]["bit_exact_roundtrip"] is True and d["config"]["generator"] == "uniform256"
    assert len(d["per_rank_ms_per_step"]) == 1 and d["size_allgather_ms"] >= 0


def test_bench_counts_the_longest_kernels_traffic_in_its_own_run(torch_mod):
    """roofline.traffic of the default bench line is counted in the run itself (two rocprofv3 --pmc child passes of
    the same command), not taken from a table: between the algorithmic bytes and 1.3 times them for the kernels the
    step is made of."""
    import json
    import shutil
    import subprocess
    import sys
    if not shutil.which("rocprofv3"):
        pytest.skip("no rocprofv3 here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--secondary", "none",
                          "--no-cpu-baseline", "--no-index-free", "--no-python-layer", "--no-other-decode"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    rf = d["roofline"]
    if not (rf.get("traffic_source") or "").startswith("counted in this run"):
        pytest.skip("the counter passes did not run here (the line carries: %s)" % rf.get("traffic_source"))
    assert rf["alg_bytes_per_launch"] <= rf["traffic"] <= 1.3 * rf["alg_bytes_per_launch"], rf


@pytest.mark.parametrize("top", [17, 24])
def test_many_blocks_of_deep_codes(torch_mod, codec, oracle, top):
    """600 blocks whose longest codes have 17..24 bits (the one-code-per-push form of pack_kernel), whole
    stream against the oracle, several launches: a pack_kernel built for 7 waves per SIMD wrote single
    wrong payload bytes here, in blocks 256 and up only and differently from run to run - nothing
    with fewer blocks, shorter codes or one launch showed it."""
    torch = torch_mod
    bs, nb = 65536, 600
    rng = np.random.default_rng(top)
    w = 0.5 ** np.arange(1, top + 1)
    data = rng.choice(top, size=nb * bs, p=w / w.sum()).astype(np.uint8)
    want = oracle.encode(data, bs)
    d = to_dev(torch, data)
    for rep in range(6):
        stream, offs, length = codec.encode(d, bs)
        got = stream[:length].cpu().numpy()
        assert got.size == want.size
        if not np.array_equal(got, want):
            oh = offs.cpu().numpy()
            bad = [b for b in range(nb) if not np.array_equal(got[oh[b]:oh[b + 1]], want[oh[b]:oh[b + 1]])]
            pytest.fail(f"launch {rep}: {len(bad)} blocks differ from the oracle, first {bad[:8]}")
    out = torch.zeros(nb * bs, dtype=torch.uint8, device="cuda")
    assert codec.decode(stream, length, offs, nb, out) == nb * bs and torch.equal(out, d)


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [32768, 65536, 262144])
def test_codes_beyond_the_table_with_the_block_index_alone(torch_mod, codec, oracle, bs):
    """round 6b: decode_regs.hpp takes blocks whose trees have codes of more than 12 bits (its table marks the twelve bits,
    the pass decodes the lane's iteration again codeword by codeword, the long one by a search over the leaves' codes).
    Rare bytes in skewed data, a geometric distribution (a long code every few hundred symbols), Fibonacci weights (codes up
    to 22 bits, one lane's last iteration can run far behind its share): encoded by the oracle, decoded with the block index
    alone and as a raw stream, both against the input."""
    torch = torch_mod
    rng = np.random.default_rng(bs)
    n = 6 * bs + 777
    sets = []
    z = rng.zipf(1.3, size=n)
    a = (z % 200).astype(np.uint8)
    rare = rng.integers(0, n, size=3 * 56)
    a[rare] = np.repeat(np.arange(200, 256, dtype=np.uint8), 3)             # 56 bytes seen three times each
    sets.append(("zipf + rare bytes", a))
    w = 0.5 ** np.arange(1, 21)
    sets.append(("geometric over 20 bytes", rng.choice(20, size=n, p=w / w.sum()).astype(np.uint8)))
    f, parts, total, sym = [1, 1], [], 0, 0
    while total + f[-2] <= n and sym < 250:
        parts.append(np.full(f[-2], sym, np.uint8)); total += f[-2]; f.append(f[-1] + f[-2]); sym += 1
    fib = np.concatenate(parts + [np.full(n - total, sym - 1, np.uint8)])
    rng.shuffle(fib)
    sets.append(("fibonacci weights", fib))
    text = np.frombuffer((b"GET /index.html HTTP/1.1 200 1234 \"Mozilla/5.0\" " * (n // 48 + 1))[:n], dtype=np.uint8).copy()
    text[rng.integers(0, n, size=40)] = rng.integers(128, 256, size=40).astype(np.uint8)
    sets.append(("text with stray bytes", text))
    for name, data in sets:
        want = oracle.encode(data, bs)
        out, offs = gpu_encode(torch, codec, data, bs)
        assert np.array_equal(out, want), (name, first_diff(out, want))
        back = gpu_decode_indexed(torch, codec, out, offs, data.size)
        assert np.array_equal(back, data), (name, "block index alone", first_diff(back, data))
        assert codec.decode_counters()[0] == 0, (name, "blocks handed to the exact decoder", codec.decode_counters())
        s = to_dev(torch, out)
        dst = torch.empty(data.size, dtype=torch.uint8, device="cuda")
        res = codec.decode_stream(s, out.size, out.size, dst)
        assert res[0] == 0 and res[1] == data.size, (name, res)
        assert np.array_equal(dst.cpu().numpy(), data), (name, "raw stream")


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [16384, 65536])
def test_blocks_of_few_bits_a_symbol_with_the_block_index_alone(torch_mod, codec, oracle, bs):
    """round 6b: with the block index decode_regs.hpp takes blocks down to shares of 64 bits - two or four byte values (all codes
    one length: shares of whole codewords), zeros with a few random bytes, geometric bytes (shares of a hundred bits, the segment done
    again when one holds more than 64 codewords).  The oracle's stream, decoded with the block index alone and as a raw stream."""
    torch = torch_mod
    rng = np.random.default_rng(bs + 1)
    n = 5 * bs + 333
    sets = [("two byte values", rng.integers(0, 2, size=n).astype(np.uint8) * 77),
            ("four byte values", (rng.integers(0, 4, size=n) * 50).astype(np.uint8))]
    sp = np.zeros(n, np.uint8)
    k = n // 100
    sp[rng.integers(0, n, size=k)] = rng.integers(1, 256, size=k).astype(np.uint8)
    sets.append(("zeros, 1 % random bytes", sp))
    w = 0.5 ** np.arange(1, 21)
    sets.append(("geometric", rng.choice(20, size=n, p=w / w.sum()).astype(np.uint8)))
    runs = np.repeat(rng.integers(0, 3, size=n // 500 + 1).astype(np.uint8), 500)[:n]
    sets.append(("runs of three byte values", runs))
    for name, data in sets:
        want = oracle.encode(data, bs)
        out, offs = gpu_encode(torch, codec, data, bs)
        assert np.array_equal(out, want), (name, first_diff(out, want))
        back = gpu_decode_indexed(torch, codec, out, offs, data.size, relaxed=True)
        assert np.array_equal(back, data), (name, "block index alone", first_diff(back, data))
        s = to_dev(torch, out)
        dst = torch.empty(data.size, dtype=torch.uint8, device="cuda")
        res = codec.decode_stream(s, out.size, out.size, dst, relaxed=True)
        assert res[0] == 0 and res[1] == data.size, (name, res)
        assert np.array_equal(dst.cpu().numpy(), data), (name, "raw stream")
