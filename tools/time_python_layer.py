"""Throughput of the bz2-style Python layer (huffmanfile.compress / decompress), host bytes in and out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import huffmanfile
from libhuffman_amd import datagen
for wl, bs in (("logtext", 1 << 20), ("zipf255", 65536), ("const41", 65536)):
    n = 256 << 20
    data = datagen.GENERATORS[wl](n).tobytes()
    huffmanfile.compress(data[: 1 << 20], bs)
    t0 = time.perf_counter(); comp = huffmanfile.compress(data, bs); t1 = time.perf_counter()
    back = huffmanfile.decompress(comp); t2 = time.perf_counter()
    assert back == data
    print(f"{wl}: compress {n / 2**30 / (t1 - t0):.2f} GiB/s, decompress {n / 2**30 / (t2 - t1):.2f} GiB/s, ratio {len(comp) / n:.3f}")
