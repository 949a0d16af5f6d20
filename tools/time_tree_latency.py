"""How long ONE block's tree takes with nothing beside it (the price of a count -> tree -> pack
fusion is a tree's latency per block): the encode of 1, 256 and 16 384 blocks of 64 KiB, the stages' times from the
library's events.  usage: time_tree_latency.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
c = GpuCodec(0)
bs = 65536
for nb in (1, 8, 256, 1024, 16384):
    n = nb * bs
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(d, wl)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        c.encode(d, bs, out=out, offsets=offs, sync=True)
    c.set_profiling(True)
    reps = 50 if nb <= 1024 else 10
    for _ in range(reps):
        c.encode(d, bs, out=out, offsets=offs, sync=False)
    torch.cuda.synchronize()
    ms, calls = c.profile("encode")
    c.set_profiling(False)
    print(f"{wl} {nb:6d} blocks: " + "  ".join(f"{k} {v / calls * 1e3:8.1f} us" for k, v in ms.items()), flush=True)
