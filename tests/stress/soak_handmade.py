"""Randomised soak of the decoders on streams NO ENCODER wrote (tests/handmade_streams.py: random trees of any shape, codes of
1 to > 100 bits, payloads that hold header-like bytes), whole and damaged, against the oracle: raw parallel, raw in order,
indexed, and huf_decode through a memory stream.  Not part of the pytest suite.
usage: python tests/stress/soak_handmade.py [seconds] [seed]"""
import os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np, torch
import handmade_streams as hm
from libhuffman_amd.codec import GpuCodec
from libhuffman_amd import _native as N
from libhuffman_amd import huffmanfile as HF
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4711
rng = np.random.default_rng(seed)
c, o = GpuCodec(0), Oracle()
L = N.load()
L.huf_gpu_set_relaxed_tree(1)
t_end = time.time() + budget
n_cases = n_corrupt = n_fake = 0

def flat_block(leaves, pay):
    tree = [0x0101, 0x41, -1, -1, 0x42, -1, -1] if leaves == 2 else [0x0103, 0x0101, 0x41, -1, -1, 0x42, -1, -1, 0x0102, 0x43, -1, -1, 0x44, -1, -1]
    per = 8 if leaves == 2 else 4
    b = struct.pack("<Qh", per * len(pay), len(tree)) + b"".join(struct.pack("<h", v) for v in tree) + bytes(pay)
    bits = np.unpackbits(np.frombuffer(bytes(pay), dtype=np.uint8))
    syms = (0x41 + bits) if leaves == 2 else (0x41 + 2 * bits[0::2] + bits[1::2])
    return b, syms.astype(np.uint8)

while time.time() < t_end:
    parts, want = [], []
    total, goal = 0, int(rng.choice([3000, 70000, 200000, 900000]))
    while total < goal:
        if rng.random() < 0.15:                                  # a payload of free bytes, sometimes with the bytes of a header in it
            pay = bytearray(rng.integers(0, 256, int(rng.integers(16, 70000)), dtype=np.uint8).tobytes())
            for _ in range(int(rng.integers(0, 3))):
                at = int(rng.integers(0, len(pay) - 12))
                tl = int(rng.choice([1, 3]))
                fake = struct.pack("<Qh", int(rng.choice([1, 5, 70000, 3000000])), tl) + (b"\xff\xff" if tl == 1 else b"\x41\x00\xff\xff\xff\xff")
                pay[at:at + len(fake)] = fake
                n_fake += 1
            b, syms = flat_block(int(rng.choice([2, 4])), pay)
        else:
            b, syms, _ = hm.block(rng, int(rng.integers(2, 257)), float(rng.choice([0.0, 0.3, 0.8, 0.97])), bool(rng.integers(0, 2)),
                                  int(rng.choice([1, 7, 300, 5000, 40000, 70000, 200000])), deep_often=bool(rng.integers(0, 2)),
                                  pad_ones=bool(rng.integers(0, 2)))
        parts.append(np.frombuffer(b, dtype=np.uint8)); want.append(syms); total += len(b)
    stream, data = np.concatenate(parts), np.concatenate(want)
    cap = data.size + 70000
    oerr, oout, oused = o.decode(stream, cap, 1025)
    assert (oerr, oused) == (0, stream.size) and np.array_equal(oout, data), ("generator vs oracle", seed, n_cases)
    s = torch.from_numpy(stream).cuda()
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    for sequential in (False, True):
        out.zero_()
        err, raw, used = c.decode_stream(s, stream.size, stream.size, out, relaxed=True, sequential=sequential)
        assert (err, raw, used) == (0, data.size, stream.size), ("stream", seed, n_cases, sequential, err, raw, used)
        assert np.array_equal(out[:raw].cpu().numpy(), data), ("stream bytes", seed, n_cases, sequential)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)).cuda()
    out.zero_()
    raw = c.decode(s, stream.size, offs, len(parts), out, relaxed=True)
    assert raw == data.size and np.array_equal(out[:raw].cpu().numpy(), data), ("indexed", seed, n_cases)
    if n_cases % 3 == 0:                                         # the drop-in API on the same stream
        src = HF._WrappedBytes(stream.tobytes()); dst = HF._MemStream(16)
        cfg = N.Config(stream.size, 0, int(rng.choice([0, 4096])), int(rng.choice([0, 65536])), src.handle, dst.handle)
        err = L.huf_decode(C.byref(cfg)); dec = dst.getvalue(); src.close(); dst.close()
        assert err == 0 and dec == data.tobytes(), ("huf_decode", seed, n_cases, err, len(dec), data.size)
    n_cases += 1
    for _ in range(3):
        bad = stream.copy()
        how = rng.integers(0, 4)
        if how == 0: bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1: bad[int(rng.integers(0, bad.size))] = int(rng.integers(0, 256))
        elif how == 2: bad = bad[: int(rng.integers(1, bad.size + 1))]
        else: bad = np.concatenate([bad, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
        oerr, oout, oused = o.decode(bad, cap, 1025)
        if oerr == 1: continue
        sb = torch.from_numpy(bad).cuda()
        for sequential in (False, True):
            out.zero_()
            err, raw, used = c.decode_stream(sb, bad.size, bad.size, out, relaxed=True, sequential=sequential)
            assert err == oerr and raw == oout.size, ("corrupt", seed, n_cases, int(how), sequential, err, oerr, raw, oout.size)
            assert np.array_equal(out[:raw].cpu().numpy(), oout), ("corrupt bytes", seed, n_cases, int(how), sequential)
        n_corrupt += 1
print(f"soak_handmade ok: {n_cases} streams ({n_fake} header-like strings planted), {n_corrupt} corruptions, seed {seed}, {budget:.0f} s")
