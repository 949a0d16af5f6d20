"""Does the encode of 1 GiB go faster as two halves on two streams (two contexts), the trees of one half beside the counting or
the packing of the other?  ms per GiB, one stream against two and four.  usage: time_encode_two_streams.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
n, bs = 1 << 30, 65536
c0 = GpuCodec(0)
d = torch.empty(n, dtype=torch.uint8, device="cuda"); c0.fill(d, wl)
for parts in (1, 2, 4, 8):
    m = n // parts
    ctxs = [GpuCodec(0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    outs = [torch.empty(c0.encode_bound(m, bs), dtype=torch.uint8, device="cuda") for _ in range(parts)]
    offs = [torch.empty(m // bs + 1, dtype=torch.int64, device="cuda") for _ in range(parts)]
    subs = [c0.new_sub_index(m, bs) for _ in range(parts)]
    def step():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                ctxs[i].encode(d[i * m:(i + 1) * m], bs, out=outs[i], offsets=offs[i], sync=False, sub_index=subs[i])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{wl}: {parts} stream(s): encode of 1 GiB {ms:.3f} ms", flush=True)
    del ctxs, outs, offs, subs
