"""Share of the decode_sub_kernel phases in cycles (thread 0 of every workgroup), from a build with
-DDEC_PHASE_PROF (HUF_LIB_PATH).  usage: phase_decode_sub.py [workloads...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
names = {2: "block setup", 3: "fast tables", 9: "tile set-up", 10: "lanes"}
for wl in sys.argv[1:] or ["zipf255"]:
    n, bs = 1 << 30, 65536
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(d, wl)
    sub = c.new_sub_index(n, bs)
    out, offs, ln = c.encode(d, bs, sub_index=sub)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    nb = c.block_count(n, bs)
    cyc = (C.c_ulonglong * 16)()
    for _ in range(2):
        c.decode(out, ln, offs, nb, back, relaxed=True, sub_index=sub, raw_size=n, blocksize=bs)
        c.lib.hufgpu_debug_phase_cycles(c._ctx, cyc, 1)
    tot = cyc[11]                       # thread 0's cycles from the kernel's first instruction to its last
    print(wl, {names[i]: round(cyc[i] / tot, 3) for i in names}, "rest", round(1 - sum(cyc[i] for i in names) / tot, 3),
          "cycles per workgroup", tot // nb, "ok" if torch.equal(back, d) else "MISMATCH")
