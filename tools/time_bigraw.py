"""Raw-stream decode of ONE block (blocksize = 0): warm time per call and the kernels behind it.
usage: time_bigraw.py [MiB] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "zipf255"
c = GpuCodec(0)
n = mib << 20
data = c.fill(torch.empty(n, dtype=torch.uint8, device="cuda"), wl)
stream, offs, length = c.encode(data, 0)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2):
    assert c.decode_stream(stream, length, length, out, relaxed=True) == (0, n, length)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    c.decode_stream(stream, length, length, out, relaxed=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"{wl}, one block of {mib} MiB as a raw stream: {ms:.2f} ms per decode = {n / 2**30 / (ms / 1e3):.1f} GiB/s, ok={torch.equal(out, data)}")
