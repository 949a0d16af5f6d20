"""Index-only decode (decode_fast_kernel) and raw-stream decode of highly compressible data (under four bits a symbol: decode_fast.hpp's
scans, not decode_regs.hpp), ms per GiB at 64 KiB blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libhuffman_amd.codec import GpuCodec
n, tile = 1 << 30, 16 << 20
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
c = GpuCodec(0)
rng = np.random.default_rng(9)
sets = {}
sets["2 symbols"] = rng.integers(0, 2, size=tile).astype(np.uint8)
sets["4 symbols"] = rng.integers(0, 4, size=tile).astype(np.uint8)
sets["16 symbols"] = rng.integers(0, 16, size=tile).astype(np.uint8)
sp = np.zeros(tile, np.uint8); k = tile // 100; sp[rng.integers(0, tile, size=k)] = rng.integers(1, 256, size=k).astype(np.uint8)
sets["zeros, 1 % random bytes"] = sp
w = 0.5 ** np.arange(1, 21)
sets["geometric20"] = rng.choice(20, size=tile, p=w / w.sum()).astype(np.uint8)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
for name, host in sets.items():
    data = torch.from_numpy(host).cuda().repeat(n // tile)
    out = torch.empty(c.encode_bound(n, bs), dtype=torch.uint8, device="cuda")
    offs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    c.encode(data, bs, out=out, offsets=offs)
    ln = int(offs[nb])
    for _ in range(2): c.decode(out, ln, offs, nb, back, relaxed=True)
    c.set_profiling(True)
    for _ in range(3): c.decode(out, ln, offs, nb, back, relaxed=True)
    p, calls = c.profile("decode"); c.set_profiling(False)
    fixed = c.decode_counters()
    ok = bool(torch.equal(back, data))
    back.zero_(); torch.cuda.synchronize()
    c.decode_stream(out, ln, ln, back, relaxed=True)
    t0 = time.perf_counter(); res = c.decode_stream(out, ln, ln, back, relaxed=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(os.environ.get("HUF_LIB_PATH", "default").split("/")[-1], bs >> 10, "KiB", name, "ratio", round(ln / n, 3), "| index-only decode", round(p["decode"] / calls, 3), "ms, exact-decoder blocks", fixed[0], "ok" if ok else "MISMATCH",
          "| raw stream", round((t1 - t0) * 1e3, 3), "ms", "ok" if (res[0] == 0 and torch.equal(back, data)) else "MISMATCH", flush=True)
    del out, offs, data
