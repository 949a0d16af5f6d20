import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libhuffman_amd import _native as N
from libhuffman_amd.codec import GpuCodec
L = N.load()
c = GpuCodec(0)
n = 240 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
libc = C.CDLL(None); libc.calloc.restype = C.c_void_p; libc.calloc.argtypes = [C.c_size_t, C.c_size_t]; libc.free.argtypes=[C.c_void_p]
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for rep in range(2):
    p = libc.calloc(n, 1)
    a = t(lambda: L.hufgpu_memcpy_d2h(c._ctx, p, d.data_ptr(), n))
    b = t(lambda: L.hufgpu_memcpy_d2h(c._ctx, p, d.data_ptr(), n))
    h = t(lambda: L.hufgpu_memcpy_h2d(c._ctx, d.data_ptr(), p, n))
    print(f"calloc'd {n>>20} MiB: first D2H {a:.1f} ms, second D2H {b:.1f} ms, H2D {h:.1f} ms")
    libc.free(p)
    p = libc.calloc(n, 1)
    a = t(lambda: C.memset(p, 1, n))
    b = t(lambda: C.memset(p, 2, n))
    print(f"memset fresh {a:.1f} ms, again {b:.1f} ms")
    libc.free(p)
