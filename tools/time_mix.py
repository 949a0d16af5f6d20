"""Times the decode kernel of const41 after different preceding kernels (cache / TLB state effects)."""
import os, sys
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
c.encode(data, bs, out=out, offsets=offs)
length = int(offs[nb].item())
back = torch.empty(n, dtype=torch.uint8, device="cuda")
other = torch.empty(n, dtype=torch.uint8, device="cuda")

def run(name, pre):
    for _ in range(2):
        pre(); c.decode(out, length, offs, nb, back, relaxed=True)
    c.set_profiling(True)
    for _ in range(6):
        pre(); c.decode(out, length, offs, nb, back, relaxed=True)
    prof, calls = c.profile("decode")
    eprof, ecalls = c.profile("encode")
    c.set_profiling(False)
    print(name, {k: round(v / calls, 3) for k, v in prof.items()}, {k: round(v / max(ecalls, 1), 3) for k, v in eprof.items()})

run("decode only", lambda: None)
run("after encode", lambda: c.encode(data, bs, out=out, offsets=offs, sync=False))
run("after histogram", lambda: c.histogram(data, bs))
run("after torch copy", lambda: other.copy_(data))
run("after torch read", lambda: data.sum())
run("after small kernel", lambda: offs.sum())
