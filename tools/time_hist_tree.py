import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
n, bs = 1 << 30, 65536
for wl in ("zipf255", "uniform256"):
    data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(c.block_count(n, bs) + 1, dtype=torch.int64, device="cuda")
    for _ in range(2): c.encode(data, bs, out=out, offsets=offs, sync=False)
    torch.cuda.synchronize(); c.set_profiling(True)
    for _ in range(6): c.encode(data, bs, out=out, offsets=offs, sync=False)
    torch.cuda.synchronize()
    e, ec = c.profile("encode")
    print(os.environ.get("HUF_LIB_PATH", "default")[-16:], wl, "hist_tree %.4f" % ((e["hist256"] + e["tree"] + e["scan_sizes"]) / ec))
