"""Config-5-shaped run: log text, 1 MiB blocks (host-generated tile repeated on the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 30, 1 << 20
tile = 16 << 20
t0 = time.time(); host = datagen.logtext(tile); print("generated", tile >> 20, "MiB in", round(time.time() - t0, 2), "s")
c = GpuCodec(0)
data = torch.from_numpy(host).cuda().repeat(n // tile)
out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
back = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    c.encode(data, bs, out=out, offsets=offs); c.decode(out, out.numel(), offs, nb, back)
c.set_profiling(True)
for _ in range(5):
    c.encode(data, bs, out=out, offsets=offs, sync=False); c.decode(out, out.numel(), offs, nb, back, sync=False)
raw = c.decode_result()
for kind in ("encode", "decode"):
    p, calls = c.profile(kind); print(kind, {k: round(v / calls, 3) for k, v in p.items()})
print("ratio", int(offs[nb]) / n, "roundtrip", bool(torch.equal(back, data)), raw == n)
