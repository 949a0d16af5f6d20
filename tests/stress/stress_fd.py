"""Stress of the huf_fdopen paths (helper-thread I/O): random sizes, block sizes, round sizes, files
and pipes on either side (pipes deliver in odd pieces and block), encode checked against the
oracle, decode against the input.  Usage: python tests/stress/stress_fd.py [seconds] [seed]"""
import ctypes as C, os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libhuffman_amd import _native as N, datagen
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = N.load()
orc = Oracle()
tmp = tempfile.mkdtemp()


def fdopen(fd):
    rw = C.POINTER(N.ReadWriter)()
    assert L.huf_fdopen(C.byref(rw), fd) == 0
    return rw


def run(fn, payload: bytes, length, bs, in_pipe, out_pipe):
    """payload -> fn -> bytes, with a file or a pipe on either side"""
    threads, result = [], {}
    if in_pipe:
        fin, pw = os.pipe()
        def feed():
            view, pos = memoryview(payload), 0
            while pos < len(view):
                step = int(rng.integers(1, 1 << 18))
                pos += os.write(pw, view[pos: pos + step])
            os.close(pw)
        threads.append(threading.Thread(target=feed))
    else:
        path = os.path.join(tmp, "in")
        with open(path, "wb") as f:
            f.write(payload)
        fin = os.open(path, os.O_RDONLY)
    if out_pipe:
        pr, fout = os.pipe()
        def drain():
            parts = []
            while True:
                b = os.read(pr, 1 << 16)
                if not b:
                    break
                parts.append(b)
            os.close(pr)
            result["out"] = b"".join(parts)
        threads.append(threading.Thread(target=drain))
    else:
        opath = os.path.join(tmp, "out")
        fout = os.open(opath, os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    for t in threads:
        t.start()
    rin, rout = fdopen(fin), fdopen(fout)
    err = fn(C.byref(N.Config(length, bs, 0, 0, rin, rout)))
    L.huf_fdclose(C.byref(rin)); L.huf_fdclose(C.byref(rout))
    os.close(fout)
    for t in threads:
        t.join()
    os.close(fin)
    if not out_pipe:
        with open(opath, "rb") as f:
            result["out"] = f.read()
    return err, result["out"]


t0, cases = time.time(), 0
while time.time() - t0 < budget:
    kind = ["zipf255", "uniform255", "const41"][int(rng.integers(0, 3))]
    n = int(rng.integers(1, 6 << 20))
    bs = int(rng.choice([4096, 65536, 131072, 1000, 1 << 20]))
    os.environ["HUF_GPU_BATCH_MB"] = str(int(rng.integers(1, 4)))
    data = datagen.GENERATORS[kind](n)
    if rng.integers(0, 3) == 0:
        data[int(rng.integers(0, n)):] = 7                        # a run of one-symbol blocks behind ordinary ones
    want = orc.encode(data, bs)
    pipes = [bool(rng.integers(0, 2)) for _ in range(4)]
    err, enc = run(L.huf_encode, data.tobytes(), n, bs, pipes[0], pipes[1])
    assert err == 0 and enc == want.tobytes(), ("encode", kind, n, bs, pipes, err, len(enc), want.size)
    err, back = run(L.huf_decode, enc, len(enc), 0, pipes[2], pipes[3])
    assert err == 0 and back == data.tobytes(), ("decode", kind, n, bs, pipes, err, len(back))
    cases += 1
print(f"stress_fd ok: {cases} cases in {time.time() - t0:.0f} s")
