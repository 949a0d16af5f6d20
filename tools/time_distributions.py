"""The three kernels of a step on byte distributions other than the bench's: where are the cliffs?
usage: time_distributions.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
n, bs = 1 << 30, 65536
nb = n // bs
g = torch.Generator(device="cuda"); g.manual_seed(1)


def sample(weights):
    w = torch.tensor(weights, dtype=torch.float32, device="cuda")
    tile = torch.multinomial(w / w.sum(), 64 << 20, replacement=True, generator=g).to(torch.uint8)
    return tile.repeat(n // tile.numel())


zipf = [1.0 / r for r in range(1, 256)] + [0.0]
cases = {
    "zipf255 (the bench's)": zipf,
    "two byte values, 50/50": [1, 1] + [0] * 254,
    "two byte values, 95/5": [95, 5] + [0] * 254,
    "16 byte values, uniform": [1] * 16 + [0] * 240,
    "40 common + 200 rare bytes (few per block)": [1.0] * 40 + [2e-4] * 200 + [0] * 16,
    "text-like: Zipf over 96 bytes, exponent 1.3": [1.0 / r ** 1.3 for r in range(1, 97)] + [0] * 160,
    "geometric 2^-i over 24 bytes (codes to 24 bits)": [0.5 ** i for i in range(1, 25)] + [0] * 232,
    "uniform over 256": [1] * 256,
}
for name, w in cases.items():
    data = sample(w)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    sub = c.new_sub_index(n, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    kw = dict(relaxed=True, sub_index=sub, raw_size=n, blocksize=bs)
    for _ in range(2):
        c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
        c.decode(out, out.numel(), offs, nb, back, sync=False, **kw)
    c.decode_result()
    c.set_profiling(True)
    for _ in range(5):
        c.encode(data, bs, out=out, offsets=offs, sync=False, sub_index=sub)
        c.decode(out, out.numel(), offs, nb, back, sync=False, **kw)
    c.decode_result()
    e, ec = c.profile("encode"); d, dc = c.profile("decode")
    c.set_profiling(False)
    ht = (e["hist256"] + e["tree"] + e["scan_sizes"]) / ec
    pk, dd = e["pack"] / ec, d["decode"] / dc
    ratio = int(offs[nb].item()) / n
    print(f"{name:52s} ratio {ratio:.3f}  hist_tree {ht:.3f}  pack {pk:.3f}  decode_sub {dd:.3f} ms  = {n / 2**30 / ((ht + pk + dd + 0.01) / 1e3):6.0f} GiB/s  ok={torch.equal(back, data)}")
