"""Counters of decode_fast.hpp's lean decoder (scan calls, rounds, segments, blocks handed to the exact decoder, guessed
block ends in the raw-stream probe and how many of them had to be done again).  Needs a library built with the
counters in:  HUF_LIB_PATH=$PWD/tools/_ablate/lib_dfastdbg.so HUF_EXTRA_FLAGS=-DDFAST_DEBUG python -m libhuffman_amd.build
then          HUF_LIB_PATH=$PWD/tools/_ablate/lib_dfastdbg.so python tools/dbg_dfast.py [--raw] [zipf255 uniform256 logtext ratechange]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
n, bs = 1 << 28, 65536
c = GpuCodec(0)
raw = "--raw" in sys.argv          # through the raw-stream probe (decode_stream) instead of the indexed kernel
for wl in [a for a in sys.argv[1:] if a != "--raw"] or ["zipf255", "uniform256"]:
    if wl == "logtext":
        from libhuffman_amd import datagen
        tile = torch.from_numpy(datagen.logtext(16 << 20)).cuda()
        data = tile.repeat(n // tile.numel())[:n].contiguous()
    elif wl == "ratechange":        # every block: cheap symbols, then dear ones (or the other way round)
        import numpy as np
        rng = np.random.default_rng(1)
        cheap = np.where(rng.random(n) < 0.93, 7, rng.integers(0, 256, n)).astype(np.uint8)
        dear = rng.integers(0, 255, n).astype(np.uint8)
        pos = np.arange(n) % bs
        first = (np.arange(n) // bs) % 2 == 0
        data = torch.from_numpy(np.where((pos < bs // 2) == first, cheap, dear)).cuda()
    elif wl == "geometric":         # three bits a symbol; in 64 KiB blocks a block in 256 that the raw-stream probe cannot vouch for
        import numpy as np
        rng = np.random.default_rng(9)
        tile = 16 << 20
        rng.integers(0, 2, size=tile); rng.integers(0, 4, size=tile); rng.integers(0, 16, size=tile); k = tile // 100; rng.integers(0, tile, size=k); rng.integers(1, 256, size=k)
        w = 0.5 ** np.arange(1, 21)
        data = torch.from_numpy(rng.choice(20, size=tile, p=w / w.sum()).astype(np.uint8)).cuda().repeat(n // tile)
    else:
        data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
    out, offs, length = c.encode(data, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda"); nb = c.block_count(n, bs)
    arr = (C.c_ulonglong * 16)()
    c.lib.hufgpu_debug_dfast(arr, 1)
    if raw: c.decode_stream(out, length, length, back, relaxed=True)
    else: c.decode(out, length, offs, nb, back, relaxed=True)
    c.lib.hufgpu_debug_dfast(arr, 0)
    a = list(arr)
    print(wl, "blocks", nb, "fail: exhausted", a[0], "rounds", a[1], "lane_ok", a[2], "take0", a[3], "| lanes !ok", a[4], "lanes exh", a[5],
          "| scan calls/blk %.1f iters/call %.1f rounds/blk %.2f segments/blk %.2f" % (a[8] / nb, a[9] / max(a[8], 1), a[10] / nb, a[11] / nb),
          "| waves that scan again in round 0 / 1 / later, per segment: %.2f %.2f %.2f, run_jump calls %.2f" % tuple(a[i] / max(a[11], 1) for i in (6, 7, 14, 15)),
          "| segments with a guessed end", a[12], "done again", a[13], "equal", torch.equal(back, data))
