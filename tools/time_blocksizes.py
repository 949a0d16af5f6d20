"""Kernel times of the step (encode with the sub-index, decode with it and with the block index alone) at several
block sizes.  usage: time_blocksizes.py [workload] [blocksize KiB ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
sizes = [int(x) for x in sys.argv[2:]] or [16, 64, 256, 1024, 2048]
c = GpuCodec(0)
n = 1 << 30
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
for kib in sizes:
    bs = kib << 10
    nb = c.block_count(n, bs)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    sub = c.new_sub_index(n, bs)
    res = {}
    for name, kw in (("sub", dict(sub_index=sub, raw_size=n, blocksize=bs)), ("index", {})):
        for _ in range(2):
            c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
            c.decode(out, out.numel(), offs, nb, back, relaxed=True, sync=False, **kw)
        c.decode_result()
        c.set_profiling(True)
        for _ in range(5):
            c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
            c.decode(out, out.numel(), offs, nb, back, relaxed=True, sync=False, **kw)
        c.decode_result()
        e, ec = c.profile("encode"); d, dc = c.profile("decode")
        c.set_profiling(False)
        res[name] = (sum(e.values()) / ec, sum(d.values()) / dc, {k: round(v / ec, 3) for k, v in e.items()})
    assert torch.equal(back, data)
    print(f"{wl} {kib:5d} KiB blocks: encode {res['sub'][0]:.3f} ms {res['sub'][2]}, decode with the sub-index {res['sub'][1]:.3f} ms, with the block index alone {res['index'][1]:.3f} ms")
