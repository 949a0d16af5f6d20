"""Child process of test_gpu_bigraw.py::test_concurrent_calls_take_different_sessions: several threads call
huf_encode()/huf_decode() at the same time on disjoint configs (legal and parallel in the reference, which
has no global state: src/encoder.c:379-392).  The device list comes from HUF_GPU_DEVICES of this process."""
import ctypes as C
import sys
import threading
import time

import numpy as np

sys.path.insert(0, sys.argv[1])
from libhuffman_amd import _native as N, datagen  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

L = N.load()
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]
NTHREADS = int(sys.argv[2])
n = ((int(sys.argv[3]) if len(sys.argv) > 3 else 24) << 20) + 17
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
inputs = [np.roll(datagen.zipf255(n), 1000 * i) for i in range(NTHREADS)]
oracle = Oracle()
want = [oracle.encode(x, bs) for x in inputs]
errors = []


def roundtrip(i, reps):
    try:
        data = inputs[i]
        for _ in range(reps):
            rin, rout, rback = C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)(), C.POINTER(N.ReadWriter)()
            bin_, bout, bback = C.c_void_p(), C.c_void_p(), C.c_void_p()
            assert L.huf_memopen(C.byref(rin), C.byref(bin_), n) == 0
            assert L.huf_memopen(C.byref(rout), C.byref(bout), n) == 0
            assert L.huf_memopen(C.byref(rback), C.byref(bback), n) == 0
            assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
            cfg = N.Config(n, bs, 0, 0, rin, rout)
            assert L.huf_encode(C.byref(cfg)) == 0
            m = C.c_size_t()
            L.huf_memlen(rout, C.byref(m))
            enc = np.frombuffer(C.string_at(bout.value, m.value), np.uint8)
            assert enc.size == want[i].size and np.array_equal(enc, want[i]), f"thread {i}: stream differs from the oracle's"
            dcfg = N.Config(m.value, 0, 0, 0, rout, rback)
            assert L.huf_decode(C.byref(dcfg)) == 0
            L.huf_memlen(rback, C.byref(m))
            assert m.value == n and np.array_equal(np.frombuffer(C.string_at(bback.value, n), np.uint8), data), f"thread {i}"
            for r in (rin, rout, rback):
                L.huf_memclose(C.byref(r))
            for b in (bin_, bout, bback):
                libc.free(b)
    except BaseException as e:  # noqa: BLE001
        errors.append(repr(e))


def run(reps):
    ts = [threading.Thread(target=roundtrip, args=(i, reps)) for i in range(NTHREADS)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


run(1)                                    # contexts, staging buffers
dt = run(3)
configured = C.c_int(0)
live = L.huf_gpu_sessions(C.byref(configured))
fe, fd = C.c_int(0), C.c_int(0)
L.huf_gpu_fanouts(C.byref(fe), C.byref(fd))
print(f"sessions live={live} configured={configured.value} threads={NTHREADS} "
      f"seconds={dt:.3f} GiB/s={3 * NTHREADS * 2 * n / dt / 2**30:.2f} fanout_encodes={fe.value} fanout_decodes={fd.value}")
if errors:
    print("ERRORS", errors)
    sys.exit(1)
