"""Times hufgpu_decode kernels for one library build (HUF_LIB_PATH) without checking results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libhuffman_amd.codec import GpuCodec
wl = sys.argv[1] if len(sys.argv) > 1 else "zipf255"
n, bs = 1 << 30, 65536
c = GpuCodec(0)
data = torch.empty(n, dtype=torch.uint8, device="cuda"); c.fill(data, wl)
# the stream comes from this build's own encoder (encode is not ablated)
out, offs, length = c.encode(data, bs)
back = torch.empty(n, dtype=torch.uint8, device="cuda")
nb = c.block_count(n, bs)
for _ in range(2):
    try: c.decode(out, length, offs, nb, back, relaxed=True)
    except Exception as e: pass
c.set_profiling(True)
for _ in range(5):
    try: c.decode(out, length, offs, nb, back, relaxed=True)
    except Exception as e: pass
prof, calls = c.profile("decode")
print(os.environ.get("HUF_LIB_PATH", "default"), wl, {k: round(v / calls, 3) for k, v in prof.items()})
if hasattr(c.lib, "hufgpu_debug_phase_cycles"):
    import ctypes as C
    arr = (C.c_ulonglong * 16)()
    c.lib.hufgpu_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    c.lib.hufgpu_debug_phase_cycles(c._ctx, arr, 1)
    c.decode(out, length, offs, nb, back, relaxed=True)
    c.lib.hufgpu_debug_phase_cycles(c._ctx, arr, 0)
    names = ["tree", "lut", "stage", "passA", "rounds", "scan", "write"]
    tot = sum(arr[i] for i in range(7))
    print({n: round(arr[i] / nb) for i, n in enumerate(names)}, "cycles per block; total", round(tot / nb))
    if arr[10]:
        print("count pass: %.1f loop iterations per wave call (%d calls/block); rounds: %.1f per call (%.1f calls/block)" % (
            arr[8] / arr[10], arr[10] / nb, arr[9] / max(arr[11], 1), arr[11] / nb))
