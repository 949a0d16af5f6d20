import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
def run(wl, n, bs, lead=0, tail=0):
    data_h = datagen.GENERATORS[wl](n)
    data = torch.from_numpy(data_h).cuda()
    out, offs, length = c.encode(data, bs)
    big = torch.zeros(length + lead + tail + 64, dtype=torch.uint8, device="cuda")
    big[lead:lead + length] = out
    st = big[lead:lead + length + tail]
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    nb = c.block_count(n, bs)
    raw = c.decode(st, length, offs, nb, back, relaxed=True)
    ex, handed = c.decode_counters()
    d = (back != data).cpu().numpy()
    idx = np.flatnonzero(d)
    msg = "ok" if idx.size == 0 and raw == n else "MISMATCH %d bytes, first %d (block %d off %d) last %d" % (idx.size, idx[0], idx[0] // bs, idx[0] % bs, idx[-1])
    print("%s n=%d bs=%d lead=%d: raw %d handed_on %d/%d exact %d %s" % (wl, n, bs, lead, raw, handed, nb, ex, msg), flush=True)
for wl in ("zipf255", "logtext"):
    for lead in (0, 1, 2, 3):
        run(wl, 8 * 65536, 65536, lead)
        run(wl, 65536 + 777, 65536, lead)
