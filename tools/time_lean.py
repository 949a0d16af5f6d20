"""decode_lean_kernel (index-only decode): result checked against the input, blocks handed on, time per GiB.
   python tools/time_lean.py [--mib 1024] [workloads...]     (HUF_GPU_LEAN_DECODE=0: round 3's decoder)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec

args = sys.argv[1:]
mib = 1024
if "--mib" in args:
    i = args.index("--mib"); mib = int(args[i + 1]); del args[i:i + 2]
wls = args or ["zipf255", "uniform256", "uniform255", "logtext", "logtext@1m", "zipf255@16k", "zipf255@1m"]
c = GpuCodec(0)
for spec in wls:
    wl, _, b = spec.partition("@")
    bs = {"": 65536, "16k": 16384, "1m": 1 << 20, "256k": 1 << 18, "4k": 4096}[b]
    n = mib << 20
    data = torch.empty(n, dtype=torch.uint8, device="cuda")
    if wl in ("logtext",):
        data.copy_(torch.from_numpy(datagen.GENERATORS[wl](n)))
    else:
        c.fill(data, wl)
    out, offs, length = c.encode(data, bs)
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    nb = c.block_count(n, bs)
    raw = c.decode(out, length, offs, nb, back, relaxed=True)
    same = raw == n and bool(torch.equal(back, data))
    exact, handed = c.decode_counters()
    msg = ""
    if not same:
        diff = (back != data).nonzero()
        first = int(diff[0]) if diff.numel() else -1
        msg = " FIRST MISMATCH at byte %d (block %d, offset %d), %d bytes differ, raw %d" % (first, first // bs, first % bs, int(diff.numel()), raw)
        if first >= 0:
            b = first // bs
            o = offs.cpu().numpy()
            import ctypes as C
            lb = (C.c_uint32 * 8)()
            c.lib.hufgpu_debug_lean_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
            c.lib.hufgpu_debug_lean_blocks(c._ctx, lb, min(8, max(handed, 1)))
            msg += "\n   handed on: %s" % list(lb)[:handed]
            msg += "\n   got  " + bytes(back[first - 8:first + 24].cpu().numpy()).hex() + "\n   want " + bytes(data[first - 8:first + 24].cpu().numpy()).hex()
            # the block alone
            sub = out[int(o[b]):int(o[b + 1])].clone()
            so = torch.tensor([0, int(o[b + 1] - o[b])], dtype=torch.int64, device="cuda")
            one = torch.zeros(bs, dtype=torch.uint8, device="cuda")
            r1 = c.decode(sub, sub.numel(), so, 1, one, relaxed=True)
            want = data[b * bs:(b + 1) * bs]
            d1 = (one[:want.numel()] != want).nonzero()
            msg += "\n   the block alone: raw %d, %d bytes differ, first %d; counters %s; stream bytes %d" % (r1, int(d1.numel()), int(d1[0]) if d1.numel() else -1, c.decode_counters(), sub.numel())
    c.set_profiling(True)
    for _ in range(5):
        c.decode(out, length, offs, nb, back, relaxed=True)
    prof, calls = c.profile("decode")
    c.set_profiling(False)
    ms = {k: round(v / calls * (1024.0 / mib), 3) for k, v in prof.items()}
    print("%-14s lean=%s ok=%s handed_on=%d/%d exact=%d ms/GiB=%s%s" % (spec, os.environ.get("HUF_GPU_LEAN_DECODE", "0"), same, handed, nb, exact, ms, msg), flush=True)
    del data, back, out
