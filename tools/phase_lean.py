"""Cycles per block of decode_lean_kernel's phases (thread 0 of every workgroup) and its counters, from a build with
-DDEC_PHASE_PROF:  HUF_LIB_PATH=$PWD/tools/_ablate/lib_leanprof.so HUF_EXTRA_FLAGS=-DDEC_PHASE_PROF python -m libhuffman_amd.build
usage: HUF_LIB_PATH=... phase_lean.py [workloads...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
names = ["tables", "stage", "run-in", "share", "settle", "scan", "image", "flush"]
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
    tot = sum(cyc[i] for i in range(8))
    print(wl, {names[i]: cyc[i] // nb for i in range(8)}, "cycles per block", tot // nb, "ok" if torch.equal(back, d) else "MISMATCH")
    fail = (C.c_ulonglong * 16)()
    c.lib.hufgpu_debug_lean_fail.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    c.lib.hufgpu_debug_lean_fail(c._ctx, fail, 1)
    why = {1: "input exhausted", 2: "rounds", 3: "too many lanes to decode again", 4: "too many lanes to walk", 5: "payload ends early", 6: "no symbol",
           7: "image too small", 8: "track with bits that are no codeword", 9: "tables"}
    print("   handed on (both runs):", {why[i]: fail[i] for i in why if fail[i]}, "of", 2 * nb)
    print("   settle:", {"or-barrier": cyc[8] // nb, "list+barrier": cyc[11] // nb, "decode again": cyc[13] // nb, "barrier": cyc[14] // nb, "owners": cyc[15] // nb})
    seg = max(cyc[12], 1)
    print("   per segment: %.1f lanes not right at first, %.2f rounds; %.2f segments per block" % (cyc[9] / seg, cyc[10] / seg, cyc[12] / nb))
