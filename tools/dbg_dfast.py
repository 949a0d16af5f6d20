import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 28, 65536
c = GpuCodec(0)
for wl in sys.argv[1:] or ["zipf255", "uniform256"]:
    data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out, offs, length = c.encode(data, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda"); nb = c.block_count(n, bs)
    arr = (C.c_ulonglong * 16)()
    c.lib.hufgpu_debug_dfast(arr, 1)
    c.decode(out, length, offs, nb, back, relaxed=True)
    c.lib.hufgpu_debug_dfast(arr, 0)
    a = list(arr)
    print(wl, "blocks", nb, "fail: exhausted", a[0], "rounds", a[1], "lane_ok", a[2], "take0", a[3], "| lanes !ok", a[4], "lanes exh", a[5],
          "| scan calls/blk %.1f iters/call %.1f rounds/blk %.2f segments/blk %.2f" % (a[8] / nb, a[9] / max(a[8], 1), a[10] / nb, a[11] / nb), "equal", torch.equal(back, data))
