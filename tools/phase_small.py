"""Cycles of decode_chain_lean_kernel (hufgpu_decode_small: one workgroup, a small stream) by phase, from a build with -DDEC_PHASE_PROF."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from libhuffman_amd.codec import GpuCodec
from libhuffman_amd import datagen
c = GpuCodec(0)
names = {13: "kernel", 14: "header", 15: "payload", 10: "plan+request", 11: "tables", 12: "first commit", 1: "stage", 2: "first pass", 3: "rounds", 4: "request+sums", 7: "walk", 8: "commit", 9: "stores"}
c.lib.hufgpu_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
for n in (4096, 16384, 65536):
    d = torch.from_numpy(datagen.zipf255(n)).cuda()
    out, offs, ln = c.encode(d, 65536)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    cyc = (C.c_ulonglong * 16)()
    reps = 20
    hin = torch.empty(ln, dtype=torch.uint8).pin_memory(); hin.copy_(out[:ln].cpu())
    hout = torch.empty(8 * ln + 4096, dtype=torch.uint8).pin_memory()
    din = torch.empty(ln + 64, dtype=torch.uint8, device="cuda")
    raw, used = C.c_uint64(0), C.c_uint64(0)
    c.lib.hufgpu_decode_small.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    def call():
        return c.lib.hufgpu_decode_small(c._ctx, hin.data_ptr(), ln, ln, 1, din.data_ptr(), back.data_ptr(), back.numel(), hout.data_ptr(), hout.numel(), C.byref(raw), C.byref(used))
    for _ in range(3): call()
    c.lib.hufgpu_debug_phase_cycles(c._ctx, cyc, 1)
    for _ in range(reps): res = (call(), raw.value, used.value)
    c.lib.hufgpu_debug_phase_cycles(c._ctx, cyc, 1)
    print(n, res, {v: cyc[k] // reps for k, v in names.items()}, "(cycles at 100 MHz? see kernel)", torch.equal(back, d))
