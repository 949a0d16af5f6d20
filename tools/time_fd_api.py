"""huf_encode / huf_decode between files through huf_fdopen streams: helper-thread
I/O next to the GPU work against the plain callback path (HUF_GPU_ZERO_COPY=0), same process."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libhuffman_amd import _native as N, datagen
L = N.load()
L.huf_gpu_set_relaxed_tree(1)
tmp = tempfile.mkdtemp(dir=os.environ.get("HUF_TIME_DIR", "/tmp"))
n = 1 << 30


def fdopen(fd):
    rw = C.POINTER(N.ReadWriter)()
    assert L.huf_fdopen(C.byref(rw), fd) == 0
    return rw


def run(fn, src, dst, length):
    fin, fout = os.open(src, os.O_RDONLY), os.open(dst, os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    rin, rout = fdopen(fin), fdopen(fout)
    t0 = time.perf_counter()
    err = fn(C.byref(N.Config(length, 65536, 0, 0, rin, rout)))
    dt = time.perf_counter() - t0
    assert err == 0, err
    size = os.fstat(fout).st_size
    for rw, fd in ((rin, fin), (rout, fout)):
        L.huf_fdclose(C.byref(rw)); os.close(fd)
    return dt, size


for wl in ("const41", "zipf255"):
    src, enc, back = (os.path.join(tmp, x) for x in ("in.bin", "out.hm", "back.bin"))
    with open(src, "wb") as f:
        for i in range(n >> 28):
            f.write(datagen.GENERATORS[wl](1 << 28).tobytes())      # the same 256 MiB four times
    res = {}
    for mode in ("threads", "callbacks", "threads", "callbacks"):
        os.environ["HUF_GPU_ZERO_COPY"] = "1" if mode == "threads" else "0"
        te, clen = run(L.huf_encode, src, enc, n)
        td, blen = run(L.huf_decode, enc, back, clen)
        assert blen == n
        res[mode] = (n / 2**30 / te, n / 2**30 / td)
    print(wl, {k: (round(v[0], 2), round(v[1], 2)) for k, v in res.items()}, "GiB/s (encode, decode), file -> file in", tmp)
    for x in (src, enc, back):
        os.remove(x)
os.rmdir(tmp)
