import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
def run(wl, n, bs, pad=0):
    data_h = datagen.GENERATORS[wl](n)
    data = torch.from_numpy(data_h).cuda()
    out, offs, length = c.encode(data, bs)
    if pad:
        big = torch.zeros(length + pad, dtype=torch.uint8, device="cuda"); big[:length] = out; out = big
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    nb = c.block_count(n, bs)
    raw = c.decode(out, length, offs, nb, back, relaxed=True)
    ex, handed = c.decode_counters()
    d = (back != data).cpu().numpy()
    idx = np.flatnonzero(d)
    msg = "ok" if idx.size == 0 and raw == n else "MISMATCH %d bytes, first %d (block %d off %d) last %d" % (idx.size, idx[0], idx[0] // bs, idx[0] % bs, idx[-1])
    if idx.size:
        # runs of mismatching positions
        brk = np.flatnonzero(np.diff(idx) > 64)
        starts = np.concatenate([[idx[0]], idx[brk + 1]]); ends = np.concatenate([idx[brk], [idx[-1]]])
        msg += " ranges " + ", ".join("%d-%d" % (a % bs, b % bs) for a, b in list(zip(starts, ends))[:8])
        a = idx[0]
        msg += "\n   got  " + bytes(back[a - 8:a + 24].cpu().numpy()).hex() + "\n   want " + bytes(data_h[a - 8:a + 24]).hex()
    print("%s n=%d bs=%d pad=%d: raw %d handed_on %d/%d exact %d %s" % (wl, n, bs, pad, raw, handed, nb, ex, msg), flush=True)
for wl in ("logtext", "zipf255"):
    run(wl, 4 * 65536, 65536)
    run(wl, 4 * 65536, 65536, pad=4096)
    run(wl, 3 * 65536 + 5000, 65536)
    run(wl, 65536, 65536)
    run(wl, 20000, 65536)
    run(wl, 3000, 65536)
    run(wl, 1 << 20, 1 << 20)
run("uniform255", 4 * 65536, 65536)
run("uniform256", 4 * 65536, 65536)
