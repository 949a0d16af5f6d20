import sys, numpy as np, torch
sys.path.insert(0, '.')
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
o = Oracle(); c = GpuCodec(0)
rng = np.random.default_rng(5)
n, k, bs = 28245, 113, 4096
data = rng.integers(0, k, size=n).astype(np.uint8)
stream, offs = o.encode(data, bs, with_offsets=True)
s = torch.from_numpy(stream.copy()).cuda()
out = torch.empty(n + 64, dtype=torch.uint8, device='cuda')
print("sequential", c.decode_stream(s, stream.size, stream.size, out, sequential=True), flush=True)
print("parallel  ", c.decode_stream(s, stream.size, stream.size, out), flush=True)
print(np.array_equal(out[:n].cpu().numpy(), data))
