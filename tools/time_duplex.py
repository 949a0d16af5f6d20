"""huf_encode / huf_decode between memory streams by input size, with the rounds' transfers overlapped (default) and one
after the other (HUF_GPU_DUPLEX=0 in a child process): milliseconds per call, best of a few, and the result checked
against the oracle's stream and the input.   usage: time_duplex.py [MiB ...]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(sizes):
    import numpy as np
    from libhuffman_amd import _native as N, datagen
    L = N.load()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    L.huf_gpu_set_relaxed_tree(1)

    def memopen(cap):
        rw, buf = C.POINTER(N.ReadWriter)(), C.c_void_p()
        assert L.huf_memopen(C.byref(rw), C.byref(buf), cap) == 0
        return rw, buf
    for mib in sizes:
        n = mib << 20
        data = datagen.zipf255(n)
        want = None
        if n <= (64 << 20):
            from oracle.oracle import Oracle
            want = Oracle().encode(data, 65536)
        best_e, best_d, ok = 1e9, 1e9, True
        for rep in range(4):
            rin, bin_ = memopen(n); rout, bout = memopen(16)
            assert rin.contents.write(rin.contents.stream, data.ctypes.data_as(C.c_void_p), n) == 0
            cfg = N.Config(n, 65536, 0, 0, rin, rout)
            t0 = time.perf_counter(); err = L.huf_encode(C.byref(cfg)); t1 = time.perf_counter()
            assert err == 0, err
            clen = C.c_size_t(); L.huf_memlen(rout, C.byref(clen))
            if want is not None and rep == 0:
                got = np.frombuffer(C.string_at(bout.value, clen.value), dtype=np.uint8)
                ok = ok and got.size == want.size and np.array_equal(got, want)
            rback, bback = memopen(16)
            dcfg = N.Config(clen.value, 0, 0, 0, rout, rback)
            t2 = time.perf_counter(); err = L.huf_decode(C.byref(dcfg)); t3 = time.perf_counter()
            assert err == 0, err
            blen = C.c_size_t(); L.huf_memlen(rback, C.byref(blen))
            if rep == 0:
                back = np.frombuffer(C.string_at(bback.value, blen.value), dtype=np.uint8)
                ok = ok and back.size == n and np.array_equal(back, data)
            for rw, b in ((rin, bin_), (rout, bout), (rback, bback)):
                L.huf_memclose(C.byref(rw)); libc.free(b)
            if rep:
                best_e, best_d = min(best_e, t1 - t0), min(best_d, t3 - t2)
        print(f"duplex={os.environ.get('HUF_GPU_DUPLEX', '1')} {mib:5d} MiB: huf_encode {best_e * 1e3:8.2f} ms ({n / 2**30 / best_e:6.1f} GiB/s)  "
              f"huf_decode {best_d * 1e3:8.2f} ms ({n / 2**30 / best_d:6.1f} GiB/s)  {'ok' if ok else 'MISMATCH'}", flush=True)


if __name__ == "__main__":
    if os.environ.get("TIME_DUPLEX_CHILD") == "1":
        run([int(x) for x in sys.argv[1:]])
    else:
        sizes = sys.argv[1:] or ["32", "64", "256", "1024"]
        for dup in ("1", "0"):
            subprocess.run([sys.executable, os.path.abspath(__file__)] + sizes,
                           env=dict(os.environ, TIME_DUPLEX_CHILD="1", HUF_GPU_DUPLEX=dup))
