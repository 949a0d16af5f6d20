"""Index-only decode (decode_fast_kernel) of data whose trees have codes beyond 12 bits, ms per GiB by block size:
log text, zipf-like bytes with rare ones, a geometric distribution.  HUF_LIB_PATH selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
n = 1 << 30
tile = 16 << 20
c = GpuCodec(0)
rng = np.random.default_rng(5)
sets = {}
sets["logtext"] = datagen.logtext(tile)
z = (rng.zipf(1.3, size=tile) % 200).astype(np.uint8)
z[rng.integers(0, tile, size=tile // 4000)] = rng.integers(200, 255, size=tile // 4000).astype(np.uint8)      # a rare byte every 4 000 (255 values: the strict tree limit)
sets["zipf+rare"] = z
w = 0.5 ** np.arange(1, 21)
sets["geometric20"] = rng.choice(20, size=tile, p=w / w.sum()).astype(np.uint8)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, host in sets.items():
    data = torch.from_numpy(host).cuda().repeat(n // tile)
    for bs in (16384, 65536, 1 << 20):
        nb = c.block_count(n, bs)
        out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
        offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
        c.encode(data, bs, out=out, offsets=offs)
        for _ in range(2): c.decode(out, out.numel(), offs, nb, back, relaxed=True)
        c.set_profiling(True)
        for _ in range(4): c.decode(out, out.numel(), offs, nb, back, relaxed=True)
        p, calls = c.profile("decode")
        c.set_profiling(False)
        fixed = c.decode_counters()
        print(os.environ.get("HUF_LIB_PATH", "default").split("/")[-1], name, bs >> 10, "KiB:", {k: round(v / calls, 3) for k, v in p.items()},
              "exact-decoder blocks", fixed, "ok" if torch.equal(back, data) else "MISMATCH", flush=True)
        del out, offs
