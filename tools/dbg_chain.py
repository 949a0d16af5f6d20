import sys, numpy as np, torch
sys.path.insert(0, '.')
from libhuffman_amd.codec import GpuCodec
from oracle.oracle import Oracle
o = Oracle(); c = GpuCodec(0)
rng = np.random.default_rng(5)
for n, k, bs in [(5000, 3, 0), (5000, 113, 0), (28245, 113, 0), (28245, 113, 4096), (28245, 40, 65536), (100, 5, 0), (20000, 30, 100)]:
    data = rng.integers(0, k, size=n).astype(np.uint8)
    stream, offs = o.encode(data, bs, with_offsets=True)
    s = torch.from_numpy(stream.copy()).cuda()
    out = torch.empty(n + 64, dtype=torch.uint8, device='cuda')
    err, raw, used = c.decode_stream(s, stream.size, stream.size, out)
    ok = err == 0 and np.array_equal(out[:raw].cpu().numpy(), data)
    out2 = torch.empty(n + 64, dtype=torch.uint8, device='cuda')
    r2 = c.decode(s, stream.size, torch.from_numpy(offs.astype(np.int64)).cuda(), offs.size - 1, out2)
    print(n, k, bs, 'chain:', err, raw, used, stream.size, ok, '| indexed ok:', np.array_equal(out2[:r2].cpu().numpy(), data))
