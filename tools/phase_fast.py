"""Cycles per block of decode_fast_kernel's phases (thread 0 of every workgroup) from a build with -DDEC_PHASE_PROF."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
names = {6: "tables+all", 1: "stage", 2: "first pass", 3: "rounds", 4: "request+sums", 7: "partial walk", 8: "commit", 9: "stores", 10: "plan+request", 11: "tables", 12: "first commit"}
for wl in sys.argv[1:] or ["zipf255"]:
    n, bs = 1 << 28, 65536
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(d, wl)
    out, offs, ln = c.encode(d, bs)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    nb = c.block_count(n, bs)
    cyc = (C.c_ulonglong * 16)()
    c.lib.hufgpu_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    for _ in range(2):
        c.decode(out, ln, offs, nb, back, relaxed=True)
        c.lib.hufgpu_debug_phase_cycles(c._ctx, cyc, 1)
    tot = sum(cyc[i] for i in names)
    print(wl, {v: cyc[k] // nb for k, v in names.items()}, "sum", tot // nb, {v: round(cyc[k] / tot, 3) for k, v in names.items()}, "ok" if torch.equal(back, d) else "MISMATCH")
