import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import datagen
from libhuffman_amd.codec import GpuCodec
c = GpuCodec(0)
n, bs = 64 << 20, 65536
data = torch.from_numpy(datagen.GENERATORS["logtext"](n)).cuda()
out, offs, length = c.encode(data, bs)
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
arr = (C.c_ulonglong * 80)()
c.lib.hufgpu_debug_lean_fail.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
c.lib.hufgpu_debug_lean_fail(c._ctx, arr, 3)
raw = c.decode(out, length, offs, nb, back, relaxed=True)
c.lib.hufgpu_debug_lean_fail(c._ctx, arr, 3)
d = (back != data).nonzero()
print("mismatches", int(d.numel()), "first", int(d[0]) if d.numel() else -1, "fail", list(arr)[:16])
o = offs.cpu().numpy()
print("last block: offset", int(o[nb - 1]), "end", int(o[nb]), "length", length)
for s in range(min(int(arr[15]), 8)):
    v = arr[16 + 8 * s:16 + 8 * s + 8]
    print("seg %d: true_start %d first %d quick %d sb %d nlive %d need_words %d seg_total %d take %d last_end %d lane0.start %d lane0.cnt %d rounds %d produced %d pay&3 %d pay_rel %d" % (
        s, v[0], v[1] >> 32, v[1] & 1, (v[1] >> 8) & 0xffff, v[2] >> 32, v[2] & 0xffffffff, v[3] >> 32, v[3] & 0xffffffff, v[4] >> 32, v[4] & 0xffffffff, v[5] >> 32, v[5] & 0xffffffff, v[6], v[7] >> 32, v[7] & 0xffffffff))
