"""What the host link and the host's memory give on this box, without the codec: pinned and pageable copies of 1 GiB in both
directions, first-touch of fresh pages (with and without huge pages advised, one thread and several), CPU copies out of pinned
memory.  The Python layer's figure (secondary.logtext_huffmanfile) is priced against these.  usage: time_host_link.py"""
import ctypes as C, mmap, os, sys, threading, time
import numpy as np, torch
n = 1 << 30
libc = C.CDLL(None, use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
MADV_HUGEPAGE, MADV_POPULATE_WRITE = 14, 23
def rate(t): return n / 2**30 / t
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
d = torch.empty(n, dtype=torch.uint8, device="cuda"); d.fill_(7)
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
print("cores the cgroup grants:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", "| threads", os.cpu_count())
print("H2D pinned   %.1f GiB/s" % rate(timed(lambda: d.copy_(pin, non_blocking=True))))
print("D2H pinned   %.1f GiB/s" % rate(timed(lambda: pin.copy_(d, non_blocking=True))))
page = torch.empty(n, dtype=torch.uint8); page.fill_(1)
print("H2D pageable %.1f GiB/s (touched pages)" % rate(timed(lambda: d.copy_(page))))
print("D2H pageable %.1f GiB/s (touched pages)" % rate(timed(lambda: page.copy_(d))))
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
class Region:
    """PRIVATE anonymous memory, as malloc() gets it for a large block (mmap.mmap(-1, n) would be a SHARED mapping: shmem pages)"""
    def __init__(self, size): self.size = size; self.addr = libc.mmap(None, size, 3, 0x22, -1, 0)
    def close(self): libc.munmap(self.addr, self.size)
def fresh(huge):
    m = Region(n + (2 << 20))
    lo = (m.addr + (2 << 20) - 1) & ~((2 << 20) - 1)
    if huge: assert libc.madvise(lo, n, MADV_HUGEPAGE) == 0
    a = np.ctypeslib.as_array((C.c_uint8 * n).from_address(lo))
    return m, a, lo
for huge in (False, True):
    m, a, lo = fresh(huge)
    t0 = time.perf_counter(); a[::4096] = 1; t = time.perf_counter() - t0
    print("first touch of 1 GiB, one thread, huge pages %s: %.1f GiB/s" % (huge, rate(t)))
    del a; m.close()
for threads in (1, 2, 4, 8, 16):
    m, a, lo = fresh(True)
    part = n // threads
    def pop(i): libc.madvise(lo + i * part, part, MADV_POPULATE_WRITE)
    ts = [threading.Thread(target=pop, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; t = time.perf_counter() - t0
    print("MADV_POPULATE_WRITE of 1 GiB (huge pages advised), %2d threads: %.1f GiB/s" % (threads, rate(t)))
    del a; m.close()
for huge in (False, True):
    m, a, lo = fresh(huge)
    libc.madvise(lo, n, MADV_POPULATE_WRITE)
    del a
    t0 = time.perf_counter(); m.close(); t = time.perf_counter() - t0
    print("munmap of 1 GiB that was touched, huge pages %s: %.1f ms" % (huge, t * 1e3))
for hold in (False, True):
    # a bytes object of 1 GiB made and dropped again and again, as huffmanfile's results are: with the one before still alive, and without
    kept, t = None, 0.0
    for rep in range(4):
        t0 = time.perf_counter(); b = bytearray(n); dt = time.perf_counter() - t0
        if rep: t += dt
        if hold: kept = b
        del b
    print("bytearray(1 GiB) (malloc + zero), the one before %s: %.1f GiB/s" % ("still alive" if hold else "freed first", rate(t / 3)))
    del kept
src = pin.numpy()
for threads in (1, 2, 4, 8, 16):
    dst = np.empty(n, dtype=np.uint8); dst[::4096] = 0
    part = n // threads
    def cp(i): np.copyto(dst[i * part:(i + 1) * part], src[i * part:(i + 1) * part])
    ts = [threading.Thread(target=cp, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; t = time.perf_counter() - t0
    print("CPU copy pinned -> touched pageable, %2d threads: %.1f GiB/s" % (threads, rate(t)))
