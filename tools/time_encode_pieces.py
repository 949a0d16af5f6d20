"""Does the Infinity Cache serve pack's second read of the input when the encode goes piece by piece?
usage: time_encode_pieces.py [workload]   (1 GiB as 1, 2, 4, 8, 16 separate encodes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
c = GpuCodec(0)
n, bs = 1 << 30, 65536
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
for pieces in (1, 2, 4, 8, 16, 32):
    m = n // pieces
    outs = [torch.empty(c.encode_bound(m, bs), dtype=torch.uint8, device="cuda") for _ in range(pieces)]
    offs = [torch.empty(c.block_count(m, bs) + 1, dtype=torch.int64, device="cuda") for _ in range(pieces)]
    subs = [c.new_sub_index(m, bs) for _ in range(pieces)]
    def run():
        for i in range(pieces):
            c.encode(data[i * m:(i + 1) * m], bs, out=outs[i], offsets=offs[i], sync=False, sub_index=subs[i])
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{wl}: 1 GiB encoded as {pieces:2d} pieces of {m >> 20:4d} MiB: {e0.elapsed_time(e1) / 10:.3f} ms")
