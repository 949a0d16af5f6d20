"""ms per encode+decode step with the per-kernel profiling events off and on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "const41"
n, bs = 1 << 30, 65536
c = GpuCodec(0)
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
back = torch.empty(n, dtype=torch.uint8, device="cuda")
def step():
    c.encode(data, bs, out=out, offsets=offs, sync=False)
    c.decode(out, out.numel(), offs, nb, back, relaxed=True, sync=False)
for prof in (False, True, False):
    c.set_profiling(prof)
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    t0 = time.perf_counter(); e0.record()
    for _ in range(K): step()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(wl, "profiling", prof, "gpu ms/step", round(e0.elapsed_time(e1) / K, 4), "wall ms/step", round((t1 - t0) * 1e3 / K, 4))
    if prof:
        for kind in ("encode", "decode"):
            p, calls = c.profile(kind); print(kind, {k: round(v / calls, 4) for k, v in p.items()})
c.decode_result()
