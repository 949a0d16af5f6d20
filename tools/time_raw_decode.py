"""Device-resident raw-stream decode (no index) vs indexed decode, 1 GiB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 30, 65536
c = GpuCodec(0)
for wl in sys.argv[1:] or ["zipf255"]:
    data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out, offs, length = c.encode(data, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    res = {}
    for name, fn in (("indexed", lambda: c.decode(out, length, offs, c.block_count(n, bs), back, relaxed=True)),
                     ("raw_parallel", lambda: c.decode_stream(out, length, length, back, relaxed=True)),):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): r = fn()
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    ok = torch.equal(back, data)
    print(wl, res, "ms; raw result", r, "roundtrip", ok)
