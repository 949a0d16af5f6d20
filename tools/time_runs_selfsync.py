"""The self-synchronising decoder (block index only) on blocks that hold a run of one byte value:
where the phases of a periodic bit string cannot be told apart it moves one lane per round.
usage: time_runs_selfsync.py [run bytes per 64 KiB block ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
n, bs = 1 << 30, 65536
nb = n // bs
base = c.fill(torch.empty(n, dtype=torch.uint8, device="cuda"), "zipf255")
for run in [int(x) for x in sys.argv[1:]] or [0, 256, 4096, 16384, 49152]:
    data = base.clone()
    if run:
        data.view(nb, bs)[:, 1000:1000 + run] = 0
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    sub = c.new_sub_index(n, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    c.encode(data, bs, out=out, offsets=offs, sub_index=sub)
    res = {}
    for name, kw in (("self-synchronising", {}), ("sub-index", dict(sub_index=sub, raw_size=n, blocksize=bs))):
        for _ in range(2):
            c.decode(out, out.numel(), offs, nb, back, sync=False, **kw)
        c.decode_result()
        c.set_profiling(True)
        for _ in range(5):
            c.decode(out, out.numel(), offs, nb, back, sync=False, **kw)
        c.decode_result()
        p, k = c.profile("decode")
        c.set_profiling(False)
        res[name] = p["decode"] / k
        assert torch.equal(back, data)
    print(f"run of {run:6d} zero bytes in every 64 KiB block: self-synchronising {res['self-synchronising']:.2f} ms, with the sub-index {res['sub-index']:.2f} ms per GiB")
