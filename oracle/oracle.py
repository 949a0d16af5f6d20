"""ctypes loaders for the test oracle.  TEST INFRASTRUCTURE ONLY.

Two checkers live here:

* ``Oracle``     - oracle/liboracle.so, our CPU restatement (oracle/huf_oracle.c);
* ``Reference``  - oracle/_ref/libhuffman_ref.so, the *unmodified reference* compiled from
                   /root/reference/src by ``make -C oracle ref`` (the .so travels to the GPU
                   box, the sources do not).  Driven through its real C API
                   (huf_memopen / huf_encode / huf_decode), struct layouts from
                   include/huffman/config.h:10-36 and include/huffman/io.h:11-21.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this
module; the product package (libhuffman_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libhuffman_ref.so")

STRICT_TREE = 1024
RELAXED_TREE = 1025


def build(ref: bool = True) -> None:
    """Compile the restatement, and the reference when its sources are present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _as_u8(data) -> np.ndarray:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    arr = np.ascontiguousarray(data)
    assert arr.dtype == np.uint8
    return arr


class Oracle:
    """oracle/liboracle.so - flat-buffer restatement."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        L = self.lib
        L.hufo_encode_bound.restype = C.c_size_t
        L.hufo_encode_bound.argtypes = [C.c_size_t, C.c_size_t]
        L.hufo_encode.restype = C.c_int
        L.hufo_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.c_void_p]
        L.hufo_decode.restype = C.c_int
        L.hufo_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]
        L.hufo_histogram.restype = None
        L.hufo_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_long)]

    def histogram(self, data) -> np.ndarray:
        src = _as_u8(data)
        freq = np.zeros(512, dtype=np.uint64)
        start = C.c_long(-1)
        self.lib.hufo_histogram(src.ctypes.data, src.size, freq.ctypes.data, C.byref(start))
        return freq[:256].copy()

    def encode(self, data, blocksize: int, with_offsets: bool = False):
        src = _as_u8(data)
        n = src.size
        cap = int(self.lib.hufo_encode_bound(n, blocksize)) + 64
        out = np.empty(cap, dtype=np.uint8)
        bs = blocksize if blocksize else max(n, 1)
        nblocks = (n + bs - 1) // bs
        offs = np.zeros(nblocks + 1, dtype=np.uint64)
        out_len = C.c_size_t(0)
        err = self.lib.hufo_encode(src.ctypes.data, n, blocksize, out.ctypes.data, cap,
                                   C.byref(out_len), offs.ctypes.data)
        if err:
            raise RuntimeError(f"oracle encode failed: {err}")
        res = out[: out_len.value].copy()
        return (res, offs) if with_offsets else res

    def decode(self, stream, raw_cap: int, max_tree_len: int = STRICT_TREE, length: int | None = None):
        """Returns (err, output bytes produced, reader bytes consumed)."""
        src = _as_u8(stream)
        out = np.empty(max(raw_cap, 1), dtype=np.uint8)
        out_len = C.c_size_t(0)
        used = C.c_size_t(0)
        err = self.lib.hufo_decode(src.ctypes.data, src.size,
                                   src.size if length is None else length,
                                   out.ctypes.data, raw_cap, C.byref(out_len), C.byref(used),
                                   max_tree_len)
        return err, out[: out_len.value].copy(), used.value


class _RW(C.Structure):
    _fields_ = [("stream", C.c_void_p),
                ("write", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("read", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)))]


class _CFG(C.Structure):
    _fields_ = [("length", C.c_uint64), ("blocksize", C.c_uint64),
                ("reader_buffer_size", C.c_size_t), ("writer_buffer_size", C.c_size_t),
                ("reader", C.POINTER(_RW)), ("writer", C.POINTER(_RW))]


class Reference:
    """The unmodified reference library, driven through huf_memopen/huf_encode/huf_decode."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        L = self.lib
        L.huf_memopen.argtypes = [C.POINTER(C.POINTER(_RW)), C.POINTER(C.c_void_p), C.c_size_t]
        L.huf_memclose.argtypes = [C.POINTER(C.POINTER(_RW))]
        L.huf_memlen.argtypes = [C.POINTER(_RW), C.POINTER(C.c_size_t)]
        L.huf_encode.argtypes = [C.POINTER(_CFG)]
        L.huf_decode.argtypes = [C.POINTER(_CFG)]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def _run(self, fn, data, blocksize, rbuf, wbuf, out_cap_hint, length=None):
        src = _as_u8(data)
        rin, rout = C.POINTER(_RW)(), C.POINTER(_RW)()
        bin_, bout = C.c_void_p(), C.c_void_p()
        # input capacity == len so the (defective) growth path of memwrite is never taken
        assert self.lib.huf_memopen(C.byref(rin), C.byref(bin_), max(src.size, 1)) == 0
        assert self.lib.huf_memopen(C.byref(rout), C.byref(bout), max(out_cap_hint, 16)) == 0
        if src.size:
            assert rin.contents.write(rin.contents.stream, src.ctypes.data, src.size) == 0
        cfg = _CFG(src.size if length is None else length, blocksize, rbuf, wbuf, rin, rout)
        err = fn(C.byref(cfg))
        n = C.c_size_t(0)
        self.lib.huf_memlen(rout, C.byref(n))
        out = np.frombuffer(C.string_at(bout.value, n.value), dtype=np.uint8).copy() if n.value else np.empty(0, np.uint8)
        self.lib.huf_memclose(C.byref(rin))
        self.lib.huf_memclose(C.byref(rout))
        self.libc.free(bin_)
        self.libc.free(bout)
        return err, out

    def encode(self, data, blocksize: int, rbuf: int = 0, wbuf: int = 0) -> np.ndarray:
        n = len(data)
        bs = blocksize if blocksize else max(n, 1)
        # large enough that the output memstream never has to grow (memwrite's growth path
        # under-allocates, src/io.c:79-104)
        hint = 2 * n + 4096 + ((n + bs - 1) // bs) * 2064
        err, out = self._run(self.lib.huf_encode, data, blocksize, rbuf, wbuf, hint)
        if err:
            raise RuntimeError(f"reference encode failed: {err}")
        return out

    def decode(self, stream, raw_hint: int = 1 << 16, rbuf: int = 0, wbuf: int = 0, length=None):
        """Returns (err, output bytes written to the writer)."""
        return self._run(self.lib.huf_decode, stream, 0, rbuf, wbuf, raw_hint, length)
