"""hufgpu_decode_stream on small raw streams: the parallel discovery against the in-order chain (flag), wall clock per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
for n in (1000, 4096, 16384, 65536, 1 << 18, 1 << 20, 4 << 20, 16 << 20):
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(d, "zipf255")
    st, offs, ln = c.encode(d, 65536)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    res = {}
    for seq in (False, True):
        for _ in range(3): r = c.decode_stream(st, ln, ln, back, sequential=seq)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 50 if n <= (1 << 20) else 10
        for _ in range(reps): r = c.decode_stream(st, ln, ln, back, sequential=seq)
        torch.cuda.synchronize()
        res["chain" if seq else "parallel"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        assert r == (0, n, ln) and torch.equal(back, d)
    print(n, res, flush=True)
