cd $GRAFT_REPO_ROOT
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, numpy as np, re
from libhuffman_amd import datagen, huffmanfile
def huge():
    s = open("/proc/self/smaps_rollup").read()
    return {k: int(re.search(k + r":\s+(\d+) kB", s).group(1)) >> 10 for k in ("Rss", "AnonHugePages")}
n = 1 << 30
tile = datagen.logtext(16 << 20)
data = np.tile(tile, (n + tile.size - 1) // tile.size)[:n].tobytes()
print("before", huge())
comp = huffmanfile.compress(data, 1 << 20); print("after compress", huge(), len(comp) >> 20, "MiB")
back = huffmanfile.decompress(comp); print("after decompress", huge())
t0 = time.perf_counter(); del back; print("free %.1f ms" % ((time.perf_counter() - t0) * 1e3), huge())
PY
