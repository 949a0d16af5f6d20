"""Many encodes of randomly shaped inputs, each compared with the oracle (bytes and block index);
every case is encoded several times in a row to expose races.  usage: stress_encode.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
c, o = GpuCodec(0), Oracle()
t_end = time.time() + budget
cases = launches = bad = 0
while time.time() < t_end:
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 400000)]))
    k = int(rng.choice([1, 2, 3, rng.integers(2, 20), rng.integers(20, 257)]))
    syms = rng.choice(256, size=k, replace=False)
    p = rng.dirichlet(np.full(k, float(rng.choice([0.3, 1.0, 5.0]))))
    data = syms[rng.choice(k, size=n, p=p)].astype(np.uint8)
    bs = int(rng.choice([0, 64, 100, 257, 4096, 65536]))
    want, woffs = o.encode(data, bs, with_offsets=True)
    d = torch.from_numpy(data).cuda(); ws = torch.from_numpy(want).cuda(); wo = torch.from_numpy(woffs.astype(np.int64)).cuda()
    for rep in range(8):
        got, offs, ln = c.encode(d, bs)
        launches += 1
        if ln != want.size or not torch.equal(offs, wo) or not torch.equal(got, ws):
            bad += 1
            g = got.cpu().numpy(); m = min(g.size, want.size)
            diff = np.flatnonzero(g[:m] != want[:m])
            od = torch.nonzero(offs != wo).flatten().cpu().numpy()
            blk = int(np.searchsorted(woffs, diff[0], side="right") - 1) if diff.size else -1
            print("MISMATCH", dict(case=cases, rep=rep, n=n, k=k, bs=bs, nb=woffs.size - 1, len=(ln, want.size), ndiff=int(diff.size),
                                   first=int(diff[0]) if diff.size else -1, block=blk, in_block=int(diff[0] - woffs[blk]) if diff.size else -1,
                                   bad_index_entries=od[:6].tolist(), n_bad_index=int(od.size)), flush=True)
    cases += 1
print("stress_encode", "FAILED" if bad else "ok", dict(cases=cases, launches=launches, bad=bad, seed=seed))
