"""Per-stage encode times (hist / tree / pack) for one library build (HUF_LIB_PATH); usage: [workloads...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 30, 65536
c = GpuCodec(0)
for wl in sys.argv[1:] or ["zipf255", "uniform256", "const41"]:
    data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(c.block_count(n, bs) + 1, dtype=torch.int64, device="cuda")
    for _ in range(3): c.encode(data, bs, out=out, offsets=offs, sync=False)
    torch.cuda.synchronize(); c.set_profiling(True)
    for _ in range(8): c.encode(data, bs, out=out, offsets=offs, sync=False)
    torch.cuda.synchronize()
    e, ec = c.profile("encode")
    c.set_profiling(False)
    print(os.path.basename(os.environ.get("HUF_LIB_PATH", "default")), wl, {k: round(v / ec, 4) for k, v in e.items()},
          "sum %.4f" % (sum(e.values()) / ec), flush=True)
