cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, numpy as np
from libhuffman_amd import datagen, huffmanfile
n = 1 << 30
tile = datagen.logtext(16 << 20)
data = np.tile(tile, (n + tile.size - 1) // tile.size)[:n].tobytes()
comp = huffmanfile.compress(data, 1 << 20); back = huffmanfile.decompress(comp)
for rep in range(3):
    t0 = time.perf_counter(); c2 = huffmanfile.compress(data, 1 << 20); t1 = time.perf_counter()
    del comp; t2 = time.perf_counter()
    b2 = huffmanfile.decompress(c2); t3 = time.perf_counter()
    del back; t4 = time.perf_counter()
    comp, back = c2, b2
    print("compress %.1f ms, free the old stream %.1f ms, decompress %.1f ms, free the old bytes %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
PY
