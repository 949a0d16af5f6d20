"""What slicing the encode could buy: the encode of a SLICE whose input, codes and output stay in the
Infinity Cache (256 MB) between its launches, against the whole gigabyte.  For every slice size: hist / tree / pack per GiB of
input with profiling events, and the wall time of encoding 1 GiB as a row of slices (one stream, sub-index on, no host round trip
between the slices - their streams are not joined, which is the part a real implementation would add)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 30, 65536
c = GpuCodec(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
for mib in (1024, 256, 128, 96, 64, 32):
    m = mib << 20
    k = (n + m - 1) // m
    outs = [torch.empty(c.encode_bound(m, bs), dtype=torch.uint8, device="cuda") for _ in range(min(k, 4))]
    offs = [torch.empty(c.block_count(m, bs) + 1, dtype=torch.int64, device="cuda") for _ in range(min(k, 4))]
    subs = [c.new_sub_index(m, bs) for _ in range(min(k, 4))]
    def row():
        for i in range(k):
            lo = i * m
            c.encode(data[lo:lo + m], bs, out=outs[i % len(outs)], offsets=offs[i % len(outs)], sub_index=subs[i % len(outs)], sync=False)
    for _ in range(2): row()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): row()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    c.set_profiling(True)
    for _ in range(3): row()
    torch.cuda.synchronize()
    e, ec = c.profile("encode")
    c.set_profiling(False)
    per_gib = {kk: round(v / ec * k, 4) for kk, v in e.items()}
    print("%s slices of %4d MiB: %d per GiB, wall %.3f ms per GiB; stages per GiB %s sum %.4f" % (wl, mib, k, wall, per_gib, sum(per_gib.values())), flush=True)
    del outs, offs, subs
