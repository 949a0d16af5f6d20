"""Average time of the sub-index decode (and of the encode kernels) on one workload.
usage: time_decode_sub.py [workload] [reps]   (HUF_LIB_PATH selects an experimental build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = GpuCodec(0)
n, bs = 1 << 30, 65536
data = torch.empty(n, dtype=torch.uint8, device="cuda")
c.fill(data, wl)
out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
offs = torch.empty(c.block_count(n, bs) + 1, dtype=torch.int64, device="cuda")
sub = c.new_sub_index(n, bs)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
relaxed = wl == "uniform256"
for _ in range(3):
    c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
    c.decode(out, out.numel(), offs, nb, back, relaxed=relaxed, sync=False, sub_index=sub, raw_size=n, blocksize=bs)
torch.cuda.synchronize()
c.set_profiling(True)
for _ in range(reps):
    c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
    c.decode(out, out.numel(), offs, nb, back, relaxed=relaxed, sync=False, sub_index=sub, raw_size=n, blocksize=bs)
try:
    c.decode_result()
except Exception as e:
    print("decode error (expected in ablation builds):", e)
e, ec = c.profile("encode"); d, dc = c.profile("decode")
print(os.environ.get("HUF_LIB_PATH", "default"), wl,
      "hist_tree %.4f pack %.4f prepare %.4f decode %.4f ms" % ((e["hist256"] + e["tree"] + e["scan_sizes"]) / ec, e["pack"] / ec, d["prepare_scan"] / dc, d["decode"] / dc),
      "ok" if torch.equal(back, data) else "MISMATCH")
