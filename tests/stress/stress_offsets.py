"""Hammers the in-kernel prefix sums (two-level tickets): many tiny blocks, many launches, every
block index compared with the oracle's.  usage: python tests/stress/stress_offsets.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(5)
c, o = GpuCodec(0), Oracle()
bad = 0
for n, bs in ((184069, 64), (1 << 22, 64), (3_000_001, 257), (1 << 24, 4096)):
    data = rng.integers(0, 7, n).astype(np.uint8)
    want, woffs = o.encode(data, bs, with_offsets=True)
    d = torch.from_numpy(data).cuda()
    wo = torch.from_numpy(woffs.astype(np.int64)).cuda()
    ws = torch.from_numpy(want).cuda()
    nb = woffs.size - 1
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for it in range(launches):
        got, offs, ln = c.encode(d, bs)
        if not torch.equal(offs, wo) or ln != want.size or not torch.equal(got, ws):
            bad += 1
            k = int(torch.nonzero(offs != wo)[0]) if not torch.equal(offs, wo) else -1
            print("MISMATCH", dict(n=n, bs=bs, launch=it, first_bad_index=k, blocks=nb))
        raw = c.decode(got, ln, offs, nb, back, relaxed=True)
        if raw != n or not torch.equal(back, d):
            bad += 1
            print("DECODE MISMATCH", dict(n=n, bs=bs, launch=it))
    print("done", n, bs, nb, "blocks x", launches, "launches")
print("stress", "FAILED" if bad else "ok", bad)
