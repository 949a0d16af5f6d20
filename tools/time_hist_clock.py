"""hist_lanes_kernel's two speeds: the kernel's time and the clock it ran at, by what runs around it.
Run under the profiler, one mode a run:
    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d <dir> -o p -- python3 tools/time_hist_clock.py <mode> [workload]
modes: encode (hist -> tree -> pack, back to back), step (the bench's step: encode + sub-index decode), stepfast (encode + index-only
decode), gap (encode with the device idle for 2 ms in front of every call).  tools/hist_clock_report.py reads the csv files."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
mode = sys.argv[1] if len(sys.argv) > 1 else "encode"
wl = sys.argv[2] if len(sys.argv) > 2 else "zipf255"
c = GpuCodec(0)
n, bs = 1 << 30, 65536
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
offs = torch.empty(c.block_count(n, bs) + 1, dtype=torch.int64, device="cuda")
sub = c.new_sub_index(n, bs)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
for it in range(40):
    if mode == "gap":
        torch.cuda.synchronize(); time.sleep(0.002)
    c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
    if mode == "step":
        c.decode(out, out.numel(), offs, nb, back, sync=False, sub_index=sub, raw_size=n, blocksize=bs)
    elif mode == "stepfast":
        c.decode(out, out.numel(), offs, nb, back, sync=False)
torch.cuda.synchronize()
print("done", mode, wl)
