"""Times hufgpu_encode kernels for one library build (HUF_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 30, 65536
c = GpuCodec(0)
for wl in sys.argv[1:] or ["zipf255"]:
    data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(c.block_count(n, bs) + 1, dtype=torch.int64, device="cuda")
    for _ in range(2): c.encode(data, bs, out=out, offsets=offs)
    c.set_profiling(True)
    for _ in range(5): c.encode(data, bs, out=out, offsets=offs)
    prof, calls = c.profile("encode")
    c.set_profiling(False)
    print(os.path.basename(os.environ.get("HUF_LIB_PATH", "default")), wl, {k: round(v / calls, 3) for k, v in prof.items()})
