"""Many launches of many blocks (300..900 per launch: more than the first workgroup per CU) of geometric
byte distributions whose longest codes have 3..24 bits - pack_kernel's triple, pair and one-code-per-push
forms - whole streams against the oracle.  usage: stress_deep_codes.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
c, o = GpuCodec(0), Oracle()
t_end = time.time() + seconds
cases = launches = bad = 0
while time.time() < t_end:
    top = int(rng.integers(3, 25))
    bs = int(rng.choice([65536, 65536, 32768, 100000, 121392]))
    nb = int(rng.integers(300, 900))
    base = float(rng.choice([0.5, 0.55, 0.6]))
    w = base ** np.arange(1, top + 1)
    data = rng.choice(top, size=nb * bs, p=w / w.sum()).astype(np.uint8)
    want = o.encode(data, bs)
    d = torch.from_numpy(data).cuda()
    for rep in range(8):
        stream, offs, length = c.encode(d, bs)
        got = stream[:length].cpu().numpy()
        launches += 1
        if got.size != want.size or not np.array_equal(got, want):
            bad += 1
            print("MISMATCH", dict(top=top, bs=bs, nb=nb, base=base, rep=rep), flush=True)
    cases += 1
print("stress_deep_codes", "ok" if bad == 0 else "FAILED", dict(cases=cases, launches=launches, bad=bad))
sys.exit(1 if bad else 0)
