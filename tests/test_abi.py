"""CPU-side tests of the product library: ABI surface and host plumbing (no GPU needed).

Restates, through ctypes, the assertions of the reference's own C unit tests for the host-side
pieces of the boundary: test/io_test.c, test/histogram_test.c, test/symbol_test.c,
test/tree_test.c, and the no-input case of test/decode_test.c:32-36.  No codec compute happens
here - without a GPU huf_encode/huf_decode must fail loudly (there is no CPU fallback).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from libhuffman_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from libhuffman_amd import build
    build.build()
    return N.load()


def have_gpu(L) -> bool:
    return L.hufgpu_device_count() > 0


def test_exports_every_declared_symbol(L):
    """Every function declared in include/*.h is exported by the shared library."""
    declared = set()
    split = sorted(os.path.join("huffman", f) for f in os.listdir(os.path.join(ROOT, "include", "huffman")))
    for hdr in ["huffman_gpu.h"] + split:
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:huf|hufgpu)_[a-z0-9_]+)\s*\(", text))
    declared -= {"huf_error_t", "huf_config_t"}
    assert len(declared) >= 60
    missing = sorted(s for s in declared if not hasattr(L, s))
    assert not missing, missing
    # the reference's full export list, incl. the non-static stream callbacks
    assert not [s for s in N.HOST_SYMBOLS if not hasattr(L, s)]
    assert len(N.HOST_SYMBOLS) == 48   # nm -D of the reference .so lists 48 (SURVEY App. E text says 47)


REFERENCE_HEADER_LIST = ["huffman/errors.h", "huffman/io.h", "huffman/config.h", "huffman/common.h",
                         "huffman/decoder.h", "huffman/encoder.h", "huffman/bufio.h", "huffman/histogram.h",
                         "huffman/malloc.h", "huffman/symbol.h", "huffman/sys.h", "huffman/tree.h"]

CFFI_PROBE = r'''
import sys, cffi
inc, so = sys.argv[1], sys.argv[2]
headers = sys.argv[3:]
# the region-cutting loop of the reference's setup_ffi.py:8-23, restated
src = ""
for header in headers:
    keep = False
    for line in open(inc + "/" + header):
        if line.startswith("#define CFFI"):
            keep = True
            continue
        if line.startswith("#undef CFFI"):
            keep = False
        if keep:
            src += line
ffi = cffi.FFI()
ffi.cdef(src)
lib = ffi.dlopen(so)
assert ffi.sizeof("huf_config_t") == 48 and ffi.sizeof("huf_read_writer_t") == 24
assert ffi.string(lib.huf_error_string(lib.HUF_ERROR_BTREE_CORRUPTED)).startswith(b"Huffman tree is corrupted")
rw, buf = ffi.new("huf_read_writer_t **"), ffi.new("void **")
assert lib.huf_memopen(rw, buf, 64) == 0
assert rw[0].write(rw[0].stream, b"0123456789", 10) == 0
n = ffi.new("size_t *")
assert lib.huf_memlen(rw[0], n) == 0 and n[0] == 10
cfg = ffi.new("huf_config_t **")
assert lib.huf_config_init(cfg) == 0 and cfg[0].blocksize == 0
assert lib.huf_config_free(cfg) == 0 and lib.huf_memclose(rw) == 0
print("cffi ok", len(src.splitlines()), "lines of cdef")
'''


def test_reference_cffi_build_recipe_binds_this_library(L, tmp_path):
    """The reference's binding (setup_ffi.py:8-43) cuts its cdef() text out of twelve per-file
    headers between "#define CFFI_x" and "#undef CFFI_x".  The same loop over the same header list,
    pointed at THIS include/ directory, must give a cdef cffi accepts and a library that binds."""
    import subprocess
    for h in REFERENCE_HEADER_LIST:
        assert os.path.exists(os.path.join(ROOT, "include", h)), h
    py = "/opt/conda/bin/python3.9"                  # the interpreter of this image that has cffi
    if not os.path.exists(py) or subprocess.run([py, "-c", "import cffi"], capture_output=True).returncode:
        pytest.skip("no interpreter with cffi in this image")
    probe = tmp_path / "probe.py"
    probe.write_text(CFFI_PROBE)
    out = subprocess.run([py, str(probe), os.path.join(ROOT, "include"), N.so_path()] + REFERENCE_HEADER_LIST,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cffi ok" in out.stdout, out.stderr[-2000:]


def test_split_headers_compile_alone_and_together(tmp_path):
    """Every per-file header is self-contained, and <huffman.h> plus all of them compile as C and C++."""
    import subprocess
    body = "".join(f"#include <{h}>\n" for h in REFERENCE_HEADER_LIST) + "#include <huffman.h>\n#include <huffman_gpu.h>\n"
    body += "int main(void) { huf_config_t c; huf_histogram_t h; (void)c; (void)h; return HUF_BTREE_LEN == 1024 ? 0 : 1; }\n"
    for comp, lang, name in (("gcc", "c", "all.c"), ("g++", "c++", "all.cpp")):
        f = tmp_path / name
        f.write_text(body)
        subprocess.check_call([comp, "-fsyntax-only", "-Wall", "-Werror", "-x", lang, "-I", os.path.join(ROOT, "include"), str(f)])
    for h in REFERENCE_HEADER_LIST:
        f = tmp_path / "one.c"
        f.write_text(f"#include <{h}>\nint x;\n")
        subprocess.check_call(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(f)])


def test_struct_layouts_are_the_reference_abi():
    assert C.sizeof(N.Config) == 48 and C.sizeof(N.ReadWriter) == 24
    assert N.Config.reader.offset == 32 and N.Config.writer.offset == 40


def test_error_strings(L):
    want = {0: "Success", 3: "Failed on read/write operation", 4: "Fatal error",
            5: "Block is corrupted, Huffman tree has impossible size",
            6: "Huffman tree is corrupted and cannot be used to decode the block",
            7: "Unknown error", -1: "Unknown error", 8: "Unknown error"}    # src/errors.c:5-33
    for code, text in want.items():
        assert L.huf_error_string(code).decode() == text


def _memopen(L, cap):
    rw, buf = C.POINTER(N.ReadWriter)(), C.c_void_p()
    assert L.huf_memopen(C.byref(rw), C.byref(buf), cap) == 0
    return rw, buf


def _free(buf):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(buf)


def test_memstream_semantics(L):
    """test/io_test.c:11-94."""
    rw, buf = _memopen(L, 2)
    n, cap = C.c_size_t(), C.c_size_t()
    assert rw.contents.write(rw.contents.stream, b"ab", 2) == 0
    L.huf_memlen(rw, C.byref(n)); L.huf_memcap(rw, C.byref(cap))
    assert (n.value, cap.value) == (2, 2)
    before = buf.value
    assert rw.contents.write(rw.contents.stream, b"cdefghij", 8) == 0        # growth 2 -> 16
    L.huf_memlen(rw, C.byref(n)); L.huf_memcap(rw, C.byref(cap))
    assert (n.value, cap.value) == (10, 16) and buf.value != before            # io_test.c:53,61
    assert C.string_at(buf.value, 10) == b"abcdefghij"
    # the growth case the reference under-allocates (cap 16, len 10, count 30 -> needs 40)
    assert rw.contents.write(rw.contents.stream, b"x" * 30, 30) == 0
    L.huf_memlen(rw, C.byref(n)); L.huf_memcap(rw, C.byref(cap))
    assert n.value == 40 and cap.value >= 40
    # short read returns the count, the next read returns 0 (io_test.c:75-90)
    out = C.create_string_buffer(64)
    want = C.c_size_t(64)
    assert rw.contents.read(rw.contents.stream, out, C.byref(want)) == 0 and want.value == 40
    want = C.c_size_t(64)
    assert rw.contents.read(rw.contents.stream, out, C.byref(want)) == 0 and want.value == 0
    assert L.huf_memrewind(rw) == 0                                            # truncate
    L.huf_memlen(rw, C.byref(n))
    assert n.value == 0
    assert L.huf_memclose(C.byref(rw)) == 0 and not rw                        # does not free *buf
    _free(buf)


def test_fd_stream_roundtrip(L, tmp_path):
    path = tmp_path / "f.bin"
    fd = os.open(path, os.O_CREAT | os.O_RDWR)
    rw = C.POINTER(N.ReadWriter)()
    assert L.huf_fdopen(C.byref(rw), fd) == 0
    assert rw.contents.write(rw.contents.stream, b"hello world", 11) == 0
    os.lseek(fd, 0, os.SEEK_SET)
    out, want = C.create_string_buffer(32), C.c_size_t(32)
    assert rw.contents.read(rw.contents.stream, out, C.byref(want)) == 0
    assert want.value == 11 and out.raw[:11] == b"hello world"
    assert L.huf_fdclose(C.byref(rw)) == 0
    os.close(fd)


class _Hist(C.Structure):
    _fields_ = [("frequencies", C.POINTER(C.c_uint64)), ("iota", C.c_size_t),
                ("length", C.c_size_t), ("start", C.c_size_t)]


def test_histogram_host_api(L):
    """test/histogram_test.c: init values, 4-byte-wide elements, accumulation, start, reset."""
    h = C.POINTER(_Hist)()
    assert L.huf_histogram_init(C.byref(h), 4, 10) == 0
    assert h.contents.start == (1 << 64) - 1 and h.contents.length == 10       # :20-22
    arr = np.array([1, 2, 3, 3, 5, 9, 9, 9], dtype="<u4")
    assert L.huf_histogram_populate(h, arr.ctypes.data_as(C.c_void_p), arr.nbytes) == 0
    assert [h.contents.frequencies[i] for i in range(10)] == [0, 1, 1, 2, 0, 1, 0, 0, 0, 3]
    assert h.contents.start == 1
    assert L.huf_histogram_populate(h, arr.ctypes.data_as(C.c_void_p), arr.nbytes) == 0
    assert h.contents.frequencies[9] == 6                                       # accumulates (:36-56)
    assert L.huf_histogram_reset(h) == 0
    assert all(h.contents.frequencies[i] == 0 for i in range(10)) and h.contents.start == (1 << 64) - 1
    assert L.huf_histogram_free(C.byref(h)) == 0 and not h


class _Node(C.Structure):
    pass


_Node._fields_ = [("index", C.c_int16), ("parent", C.POINTER(_Node)), ("left", C.POINTER(_Node)),
                  ("right", C.POINTER(_Node))]


class _Tree(C.Structure):
    _fields_ = [("leaves", C.POINTER(C.POINTER(_Node))), ("root", C.POINTER(_Node))]


def test_tree_host_api_single_symbol_shape(L):
    """test/tree_test.c:12-35: {3,3,3,3} -> root 256, left = leaf 3, right = NULL."""
    h, t = C.POINTER(_Hist)(), C.POINTER(_Tree)()
    assert L.huf_histogram_init(C.byref(h), 1, 512) == 0 and L.huf_tree_init(C.byref(t)) == 0
    data = (C.c_uint8 * 4)(3, 3, 3, 3)
    assert L.huf_histogram_populate(h, data, 4) == 0
    assert L.huf_tree_from_histogram(t, h) == 0
    root = t.contents.root.contents
    assert root.index == 256 and not root.right
    assert t.contents.leaves[3].contents.index == 3
    assert C.addressof(root.left.contents) == C.addressof(t.contents.leaves[3].contents)
    buf, n = (C.c_int16 * 1032)(), C.c_size_t()
    assert L.huf_tree_serialize(t, buf, C.byref(n)) == 0
    assert list(buf[: n.value]) == [256, 3, -1, -1, -1]
    L.huf_tree_free(C.byref(t)); L.huf_histogram_free(C.byref(h))


def test_tree_host_api_matches_oracle_serialization(L, oracle):
    """The host tree builder is the same algorithm as the device kernel: compare its
    serialization with the tree inside the oracle's stream for the README input."""
    h, t = C.POINTER(_Hist)(), C.POINTER(_Tree)()
    L.huf_histogram_init(C.byref(h), 1, 512); L.huf_tree_init(C.byref(t))
    for data in (b"0123456789", b"abracadabra", bytes(range(256)) * 3 + b"zzzz"):
        L.huf_histogram_reset(h); L.huf_tree_reset(t)
        arr = (C.c_uint8 * len(data)).from_buffer_copy(data)
        assert L.huf_histogram_populate(h, arr, len(data)) == 0
        assert L.huf_tree_from_histogram(t, h) == 0
        buf, n = (C.c_int16 * 1032)(), C.c_size_t()
        assert L.huf_tree_serialize(t, buf, C.byref(n)) == 0
        stream = oracle.encode(data, 0)
        tl = int(np.frombuffer(stream[8:10].tobytes(), "<i2")[0])
        want = np.frombuffer(stream[10:10 + 2 * tl].tobytes(), "<i2").tolist()
        assert list(buf[: n.value]) == want
        # deserialize -> serialize is the identity
        t2 = C.POINTER(_Tree)()
        L.huf_tree_init(C.byref(t2))
        assert L.huf_tree_deserialize(t2, buf, n.value) == 0
        buf2, n2 = (C.c_int16 * 1032)(), C.c_size_t()
        assert L.huf_tree_serialize(t2, buf2, C.byref(n2)) == 0
        assert list(buf2[: n2.value]) == want
        L.huf_tree_free(C.byref(t2))
    L.huf_tree_free(C.byref(t)); L.huf_histogram_free(C.byref(h))


class _Elem(C.Structure):
    _fields_ = [("length", C.c_size_t), ("coding", C.POINTER(C.c_uint8))]


def test_symbol_mapping_host_api(L):
    """test/symbol_test.c: insert/get identity, overwrite, reset."""
    m = C.c_void_p()
    assert L.huf_symbol_mapping_init(C.byref(m), 256) == 0
    e1, e2, got = C.POINTER(_Elem)(), C.POINTER(_Elem)(), C.POINTER(_Elem)()
    assert L.huf_symbol_mapping_element_init(C.byref(e1), b"0101", 4) == 0
    assert L.huf_symbol_mapping_element_init(C.byref(e2), b"11", 2) == 0
    L.huf_symbol_mapping_insert.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.huf_symbol_mapping_get.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    assert L.huf_symbol_mapping_insert(m, 65, e1) == 0
    assert L.huf_symbol_mapping_get(m, 65, C.byref(got)) == 0
    assert got.contents.length == 4 and bytes(got.contents.coding[:4]) == b"0101"
    assert L.huf_symbol_mapping_insert(m, 65, e2) == 0          # frees the previous element
    L.huf_symbol_mapping_get(m, 65, C.byref(got))
    assert got.contents.length == 2
    assert L.huf_symbol_mapping_insert(m, 256, e2) == 2         # out of range
    L.huf_symbol_mapping_reset.argtypes = [C.c_void_p]
    assert L.huf_symbol_mapping_reset(m) == 0
    L.huf_symbol_mapping_get(m, 65, C.byref(got))
    assert not got
    assert L.huf_symbol_mapping_free(C.byref(m)) == 0


def test_bufio_and_bit_writer(L):
    """huf_bit_write is MSB first (src/bufio.c:18-23); bufio passes through when capacity is 0
    and counts accepted bytes."""
    class Bit(C.Structure):
        _fields_ = [("bits", C.c_uint8), ("offset", C.c_uint8)]
    b = Bit()
    L.huf_bit_read_writer_reset(C.byref(b))
    assert (b.bits, b.offset) == (0, 8)
    for bit in (1, 0, 1, 1):
        L.huf_bit_write(C.byref(b), bit)
    assert (b.bits, b.offset) == (0b10110000, 4)

    class Bufio(C.Structure):
        _fields_ = [("bytes", C.c_void_p), ("offset", C.c_size_t), ("capacity", C.c_size_t),
                    ("length", C.c_size_t), ("processed", C.c_uint64), ("rw", C.c_void_p)]
    for cap in (0, 5):
        rw, buf = _memopen(L, 4)
        io = C.POINTER(Bufio)()
        assert L.huf_bufio_read_writer_init(C.byref(io), rw, cap) == 0
        assert L.huf_bufio_write(io, b"abc", 3) == 0 and L.huf_bufio_write(io, b"defgh", 5) == 0
        assert L.huf_bufio_write_uint8(io, 0x69) == 0
        assert L.huf_bufio_read_writer_flush(io) == 0
        n = C.c_size_t()
        L.huf_memlen(rw, C.byref(n))
        assert C.string_at(buf.value, n.value) == b"abcdefghi" and io.contents.processed == 9
        # reading back through a second bufio: satisfied request, then a short one -> error 3
        rd = C.POINTER(Bufio)()
        L.huf_bufio_read_writer_init(C.byref(rd), rw, cap)
        out = C.create_string_buffer(16)
        assert L.huf_bufio_read(rd, out, 4) == 0 and out.raw[:4] == b"abcd"
        byte = C.c_uint8()
        assert L.huf_bufio_read_uint8(rd, C.byref(byte)) == 0 and byte.value == ord("e")
        assert L.huf_bufio_read(rd, out, 10) == 3                               # bufio.c:251-253
        assert rd.contents.processed == 5
        L.huf_bufio_read_writer_free(C.byref(io)); L.huf_bufio_read_writer_free(C.byref(rd))
        L.huf_memclose(C.byref(rw)); _free(buf)


def test_argument_checks_instead_of_crashes(L):
    """NULL config / reader / writer return INVALID_ARGUMENT."""
    assert L.huf_encode(None) == 2 and L.huf_decode(None) == 2
    cfg = N.Config(10, 0, 0, 0, None, None)
    assert L.huf_encode(C.byref(cfg)) == 2 and L.huf_decode(C.byref(cfg)) == 2


def test_zero_length_is_success_without_gpu(L):
    """length == 0: nothing read, nothing written (encoder.c:288, decoder.c:218,
    test/decode_test.c:32-36) - needs no device."""
    rin, bin_ = _memopen(L, 16)
    rout, bout = _memopen(L, 16)
    cfg = N.Config(0, 0, 128, 128, rin, rout)
    assert L.huf_decode(C.byref(cfg)) == 0 and L.huf_encode(C.byref(cfg)) == 0
    n = C.c_size_t(1)
    L.huf_memlen(rout, C.byref(n))
    assert n.value == 0
    for rw, b in ((rin, bin_), (rout, bout)):
        L.huf_memclose(C.byref(rw)); _free(b)


def test_no_gpu_is_a_loud_error(L, capfd):
    """Without a gfx950 device the codec must fail with HUF_ERROR_FATAL - never fall back."""
    if have_gpu(L):
        pytest.skip("a GPU is present")
    rin, bin_ = _memopen(L, 16)
    rout, bout = _memopen(L, 16)
    rin.contents.write(rin.contents.stream, b"0123456789", 10)
    cfg = N.Config(10, 65536, 0, 0, rin, rout)
    assert L.huf_encode(C.byref(cfg)) == 4
    err = capfd.readouterr().err
    assert "no CPU fallback" in err
    ctx = C.c_void_p()
    assert L.hufgpu_ctx_create(C.byref(ctx), 0) == 4 and not ctx
    assert b"no CPU fallback" in L.hufgpu_last_error(None)
    for rw, b in ((rin, bin_), (rout, bout)):
        L.huf_memclose(C.byref(rw)); _free(b)


def test_product_does_not_touch_the_oracle():
    """The product package must never import, link or open anything under oracle/."""
    pkg = os.path.join(ROOT, "libhuffman_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "datagen.py", os.path.join(dirpath, f)


def test_copy_out_into_a_fresh_bytes_object(L):
    """huf_gpu_copy_out (the Python layer's way to a `bytes` result without a page-fault-bound copy):
    exact bytes for sizes below and above its threading threshold, odd alignments included."""
    import numpy as np
    from libhuffman_amd import huffmanfile as H
    rng = np.random.default_rng(7)
    for n in (0, 1, 4097, (16 << 20) - 1, (16 << 20) + 3, (37 << 20) + 12345):
        src = rng.integers(0, 256, n + 5, dtype=np.uint8)
        out = H._PyBytes_New(None, n)
        assert L.huf_gpu_copy_out(H._PyBytes_AsString(out), src.ctypes.data + 5, n) == 0
        assert type(out) is bytes and len(out) == n and out == src[5:].tobytes()
    assert L.huf_gpu_copy_out(None, None, 10) == N.HUF_ERROR_INVALID_ARGUMENT


def test_session_pool_follows_the_device_list():
    """HUF_GPU_DEVICES is read once per process, without touching a GPU: one session per listed device
    (a device may be listed twice); unset, empty or "all" on a box without devices = one session."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from libhuffman_amd import _native as N; L = N.load(); "
            "c = C.c_int(0); live = L.huf_gpu_sessions(C.byref(c)); print(live, c.value)" % root)
    for value, want in ((None, 1), ("", 1), ("3,1,1", 3), ("0, 2 ,5,7", 4), ("junk", 1), ("all", None)):
        env = dict(os.environ)
        env.pop("HUF_GPU_DEVICES", None)
        if value is not None:
            env["HUF_GPU_DEVICES"] = value
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        live, configured = (int(x) for x in r.stdout.split())
        assert live == 0                                    # no call was made: no context exists
        if want is None:                                    # "all": the visible devices, at least the fallback
            assert configured >= 1
        else:
            assert configured == want, (value, configured)


def test_memwrap_out_is_a_fixed_capacity_writer():
    """huf_gpu_memwrap_out (extension): a writer over memory the caller provides - what is written lands there,
    what does not fit is refused (error 1), nothing is grown, moved or freed."""
    import ctypes as C
    from libhuffman_amd import _native as N
    L = N.load()
    buf = (C.c_char * 16)()
    rw = C.POINTER(N.ReadWriter)()
    assert L.huf_gpu_memwrap_out(C.byref(rw), C.addressof(buf), 16) == 0
    w = rw.contents
    assert w.write(w.stream, b"0123456789", 10) == 0
    n = C.c_size_t()
    assert L.huf_memlen(rw, C.byref(n)) == 0 and n.value == 10
    assert w.write(w.stream, b"abcdefg", 7) == N.HUF_ERROR_MEMORY_ALLOCATION          # 17 > 16
    assert L.huf_memlen(rw, C.byref(n)) == 0 and n.value == 10
    assert w.write(w.stream, b"abcdef", 6) == 0
    assert bytes(buf) == b"0123456789abcdef"
    got = (C.c_char * 16)()
    cnt = C.c_size_t(16)
    assert w.read(w.stream, got, C.byref(cnt)) == 0 and cnt.value == 16 and bytes(got) == bytes(buf)
    assert L.huf_memclose(C.byref(rw)) == 0
    assert bytes(buf) == b"0123456789abcdef"                                             # the memory is the caller's
    assert L.huf_gpu_memwrap_out(C.byref(rw), None, 4) != 0
