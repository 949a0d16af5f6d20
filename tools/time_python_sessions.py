"""huffmanfile.compress / decompress of 1 GiB of log text (configs[4]'s shape) by number of device sessions on ONE GPU
and round size: HUF_GPU_DEVICES / HUF_GPU_BATCH_MB are read when the library starts, so every setting is a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from libhuffman_amd import datagen, huffmanfile
n, bs = 1 << 30, 1 << 20
tile = datagen.logtext(16 << 20)
data = np.tile(tile, n // tile.size)[:n].tobytes()
comp = huffmanfile.compress(data, bs); back = huffmanfile.decompress(comp)
tc = td = 0.0
for _ in range(2):
    t0 = time.perf_counter(); comp = huffmanfile.compress(data, bs); t1 = time.perf_counter()
    back = huffmanfile.decompress(comp); t2 = time.perf_counter()
    tc += t1 - t0; td += t2 - t1
print("compress %%.2f decompress %%.2f both %%.2f GiB/s ok=%%s" %% (2 * n / 2**30 / tc, 2 * n / 2**30 / td, 2 * n / 2**30 / (tc + td), back == data))
''' % ROOT
for devs in ("0", "0,0", "0,0,0", "0,0,0,0"):
    for mb in ("16", "32", "64"):
        env = dict(os.environ, HUF_GPU_DEVICES=devs, HUF_GPU_BATCH_MB=mb)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        print("sessions %-8s rounds of %s MiB: %s" % (devs, mb, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
