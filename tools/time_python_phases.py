"""Where a 1 GiB huffmanfile.compress() spends its time (stream set-up, huf_encode, result), and the warm rates of
compress / decompress.  usage: time_python_phases.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import huffmanfile
from libhuffman_amd import datagen, huffmanfile as hf, _native as N
n = 1 << 30
data = datagen.GENERATORS["logtext"](16 << 20).tobytes() * 64
bs = 1 << 20
huffmanfile.compress(data[: 1 << 20], bs)
for rep in range(3):
    t0 = time.perf_counter()
    src, sink = hf._WrappedBytes(data), hf._BytesSink(n + n // 8 + 2064 * 1024 + 4096)
    t1 = time.perf_counter()
    cfg = N.Config(n, bs, 0, 0, src.handle, sink.handle)
    err = N.load().huf_encode(C.byref(cfg))
    t2 = time.perf_counter()
    comp = sink.finish()
    t3 = time.perf_counter()
    src.close(); sink.close()
    t4 = time.perf_counter()
    print(f"compress err {err}: open {t1-t0:.3f} s, huf_encode {t2-t1:.3f} s ({n/2**30/(t2-t1):.1f} GiB/s), finish {t3-t2:.3f} s, close {t4-t3:.3f} s; total {n/2**30/(t4-t0):.2f} GiB/s")
    t0 = time.perf_counter(); back = huffmanfile.decompress(comp); t1 = time.perf_counter()
    print(f"decompress {n/2**30/(t1-t0):.2f} GiB/s", back == data)
    del comp, back
