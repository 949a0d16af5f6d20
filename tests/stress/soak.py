"""Randomised soak of the GPU codec against the oracle (not part of the pytest suite: minutes long).
usage: python tests/stress/soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle

import ctypes as C
from libhuffman_amd import _native as N
from libhuffman_amd import huffmanfile as HF

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
c, o = GpuCodec(0), Oracle()
t_end = time.time() + budget
n_cases = n_corrupt = 0

def make_data():
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 200000), rng.integers(200000, 1500000),
                        rng.integers(1500000, 9000000)]))           # the last: blocks that are cut into chunks (bs 0 / 4 MiB)
    k = int(rng.choice([1, 2, 3, rng.integers(2, 20), rng.integers(20, 257)]))
    syms = rng.choice(256, size=k, replace=False)
    kind = rng.integers(0, 5)
    if kind == 0: p = np.full(k, 1.0 / k)
    elif kind == 1: p = rng.dirichlet(np.full(k, 0.3))
    elif kind == 2: p = 1.0 / np.arange(1, k + 1); p /= p.sum()
    elif kind == 3: p = 0.5 ** np.arange(1, k + 1); p[-1] += 1 - p.sum()          # very deep codes
    else: p = rng.dirichlet(np.full(k, 5.0))
    data = syms[rng.choice(k, size=n, p=p)].astype(np.uint8)
    if rng.random() < 0.3:                                                          # long runs
        i = int(rng.integers(0, n)); data[i:i + int(rng.integers(1, n // 2 + 2))] = syms[0]
    return data

while time.time() < t_end:
    data = make_data()
    n = data.size
    bs = int(rng.choice([0, 64, 257, 4096, 65536, 131072, 1 << 20, 4 << 20, (4 << 20) + 12345]))
    if bs and n // bs > 30000: bs = 4096
    want, woffs = o.encode(data, bs, with_offsets=True)
    d = torch.from_numpy(data).cuda()
    got, offs, _ = c.encode(d, bs)
    gnp = got.cpu().numpy()
    if not np.array_equal(gnp, want):
        m = min(gnp.size, want.size)
        d = int(np.flatnonzero(gnp[:m] != want[:m])[0]) if np.any(gnp[:m] != want[:m]) else m
        blk = int(np.searchsorted(woffs, d, side="right") - 1)
        np.save("gpurun_out/soak_fail_data.npy", data)
        print("ENCODE MISMATCH", dict(seed=seed, case=n_cases, n=n, bs=bs, sizes=(gnp.size, want.size), first_diff=d, block=blk,
                                      block_off=int(woffs[blk]), goffs=offs.cpu().numpy()[blk:blk + 2].tolist(), woffs=woffs[blk:blk + 2].tolist(),
                                      got=gnp[d - 4:d + 12].tolist(), want=want[d - 4:d + 12].tolist(),
                                      block_data=data[blk * max(bs, 1):(blk + 1) * max(bs, 1)].tolist() if bs else None))
        raise SystemExit(1)
    assert np.array_equal(offs.cpu().numpy().astype(np.uint64), woffs), ("offsets", seed, n_cases)
    nb = woffs.size - 1
    back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    raw = c.decode(got, want.size, offs, nb, back, relaxed=True)
    assert raw == n and np.array_equal(back[:n].cpu().numpy(), data), ("decode", seed, n_cases, n, bs)
    for sequential in (False, True):
        back.zero_()
        err, raw, used = c.decode_stream(got, want.size, want.size, back, relaxed=True, sequential=sequential)
        assert (err, raw, used) == (0, n, want.size), ("stream", seed, n_cases, sequential, err, raw, used)
        assert np.array_equal(back[:n].cpu().numpy(), data)
    # the sub-index path: the encoder's own sub-index (the stream must not change), then the same
    # sub-index damaged in a random place (results must not change: it is verified, not trusted)
    sub = c.new_sub_index(n, bs)
    got2, offs2, len2 = c.encode(d if isinstance(d, torch.Tensor) else torch.from_numpy(data).cuda(), bs, sub_index=sub)
    assert len2 == want.size and np.array_equal(got2.cpu().numpy(), want), ("encode_sub", seed, n_cases, n, bs)
    for damage in (False, True):
        s2 = sub
        if damage:
            s2 = sub.clone()
            v = s2.view(torch.int16)
            for _ in range(int(rng.integers(1, 4))):
                v[int(rng.integers(0, v.numel()))] += int(rng.integers(1, 300))
        back.zero_()
        raw = c.decode(got2, len2, offs2, nb, back, relaxed=True, sub_index=s2, raw_size=n, blocksize=bs)
        assert raw == n and np.array_equal(back[:n].cpu().numpy(), data), ("decode_sub", seed, n_cases, n, bs, damage)
    n_cases += 1
    # corruptions: compare error code and delivered bytes with the oracle
    for _ in range(3):
        bad = want.copy()
        how = rng.integers(0, 4)
        if how == 0: bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1: bad[int(rng.integers(0, bad.size))] = 0xff
        elif how == 2: bad = bad[: int(rng.integers(1, bad.size + 1))]
        else: bad = np.concatenate([bad, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
        oerr, oout, oused = o.decode(bad, n + 70000, 1025)
        if oerr == 1: continue
        for sequential in (False, True):
            err, raw, used = c.decode_stream(torch.from_numpy(bad).cuda(), bad.size, bad.size,
                                             torch.empty(n + 70000, dtype=torch.uint8, device="cuda"),
                                             relaxed=True, sequential=sequential)
            assert err == oerr and raw == oout.size, ("corrupt", seed, n_cases, how, sequential, err, oerr, raw, oout.size)
        n_corrupt += 1
    # the drop-in host API on the same case: huf_encode / huf_decode through memstreams, with the
    # device-direct copy and with the read()/write() callbacks
    if n_cases % 4 == 0:
        L = N.load()
        L.huf_gpu_set_relaxed_tree(1)
        for zc in ("1", "0"):
            os.environ["HUF_GPU_ZERO_COPY"] = zc
            os.environ["HUF_GPU_BATCH_MB"] = str(int(rng.choice([1, 256])))
            def run(fn, payload, length, bsz, cap):
                src = HF._MemStream(max(len(payload), 1)) if zc == "0" or rng.random() < 0.5 else HF._WrappedBytes(payload)
                if isinstance(src, HF._MemStream): src.write(payload)
                dst = HF._MemStream(cap)
                wbuf = int(rng.choice([0, 5, 65536]))
                cfg = N.Config(length, bsz, int(rng.choice([0, 7, 4096])), wbuf, src.handle, dst.handle)
                err = fn(C.byref(cfg)); outb = dst.getvalue(); src.close(); dst.close()
                run.wbuf = wbuf
                return err, outb
            err, enc = run(L.huf_encode, data.tobytes(), n, bs, 16)
            assert err == 0 and enc == want.tobytes(), ("huf_encode", seed, n_cases, zc, err)
            err, dec = run(L.huf_decode, enc, len(enc), 0, 16)
            assert err == 0 and dec == data.tobytes(), ("huf_decode", seed, n_cases, zc, err)
            bad = want.copy(); bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
            oerr, oout, _ = o.decode(bad, n + 70000, 1025)
            if oerr != 1:
                err, dec = run(L.huf_decode, bad.tobytes(), bad.size, 0, 16)
                # an unbuffered writer has everything the reference wrote before the error; a buffered
                # one may hold back less than its buffer (no flush on the error path, decoder.c:278-286)
                assert err == oerr and dec == oout.tobytes()[: len(dec)] and len(oout) - len(dec) <= run.wbuf, (
                    "huf_decode corrupt", seed, n_cases, zc, err, oerr, len(dec), len(oout))
        L.huf_gpu_set_relaxed_tree(0)
print(f"soak ok: {n_cases} cases, {n_corrupt} corruptions, seed {seed}, {budget:.0f} s")
