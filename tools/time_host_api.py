"""End-to-end rates through the drop-in C API (host memstreams -> H2D -> kernels -> D2H)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libhuffman_amd import _native as N, datagen
L = N.load()
libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
def memopen(cap):
    rw, buf = C.POINTER(N.ReadWriter)(), C.c_void_p()
    assert L.huf_memopen(C.byref(rw), C.byref(buf), cap) == 0
    return rw, buf
for wl, n in (("const41", 256 << 20), ("zipf255", 256 << 20)):
    data = datagen.GENERATORS[wl](n)
    for rep in range(2):
        rin, bin_ = memopen(n); rout, bout = memopen(n + n // 4 + (1 << 20))
        assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
        cfg = N.Config(n, 65536, 0, 0, rin, rout)
        t0 = time.perf_counter(); err = L.huf_encode(C.byref(cfg)); t1 = time.perf_counter()
        assert err == 0, err
        clen = C.c_size_t(); L.huf_memlen(rout, C.byref(clen))
        rback, bback = memopen(n + 4096)
        L.huf_gpu_set_relaxed_tree(1)
        dcfg = N.Config(clen.value, 0, 0, 0, rout, rback)
        t2 = time.perf_counter(); err = L.huf_decode(C.byref(dcfg)); t3 = time.perf_counter()
        assert err == 0, err
        blen = C.c_size_t(); L.huf_memlen(rback, C.byref(blen))
        ok = blen.value == n and C.string_at(bback.value, 64) == data[:64].tobytes()
        for rw, b in ((rin, bin_), (rout, bout), (rback, bback)):
            L.huf_memclose(C.byref(rw)); libc.free(b)
    print(f"{wl}: huf_encode {n / 2**30 / (t1 - t0):.2f} GiB/s, huf_decode (raw stream, chain) {n / 2**30 / (t3 - t2):.3f} GiB/s, ok={ok}")
