"""The Python surface (libhuffman_amd/huffmanfile.py, importable as `huffmanfile`).

CPU tests cover the API shape; GPU tests restate the reference's four pytest cases
(huffmanfile/huffmanfile_test.py:8-54) and the captures of SURVEY §8c through the real codec.
"""
import hashlib
import io

import numpy as np
import pytest

import huffmanfile
from libhuffman_amd import datagen


def _have_gpu():
    from libhuffman_amd import _native
    return _native.load().hufgpu_device_count() > 0


# ------------------------------------------------------------------ CPU: API shape
def test_public_names_and_defaults():
    assert set(huffmanfile.__all__) >= {"HuffmanError", "HuffmanFile", "HuffmanCompressor",
                                        "HuffmanDecompressor", "compress", "decompress"}
    assert huffmanfile.DEFAULT_BLOCK_SIZE == 131072 and huffmanfile.DEFAULT_MEM_LIMIT == 262144
    assert issubclass(huffmanfile.HuffmanError, Exception)


def test_empty_inputs_need_no_device():
    assert huffmanfile.compress(b"") == b"" and huffmanfile.decompress(b"") == b""
    c = huffmanfile.HuffmanCompressor()
    assert c.compress(b"") == b"" and c.flush() == b"" and c.flush() == b""
    with pytest.raises(ValueError):
        c.compress(b"x")                       # bz2 semantics; the reference raises TypeError here


def test_file_object_argument_checks(tmp_path):
    with pytest.raises(ValueError):
        huffmanfile.HuffmanFile(tmp_path / "x", mode="rw")
    with pytest.raises(TypeError):
        huffmanfile.HuffmanFile(12345)
    with pytest.raises(ValueError):
        huffmanfile.open(tmp_path / "x", "rbt")
    with pytest.raises(ValueError):
        huffmanfile.open(tmp_path / "x", "rb", encoding="utf-8")
    f = huffmanfile.HuffmanFile(io.BytesIO(), "w")
    assert f.writable() and not f.readable() and not f.seekable() and not f.closed
    with pytest.raises(io.UnsupportedOperation):
        f.read()
    f.close()
    f.close()                                   # idempotent
    assert f.closed
    with pytest.raises(ValueError):
        f.writable()


def test_no_gpu_raises_huffman_error_with_reference_format():
    if _have_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(huffmanfile.HuffmanError) as ei:
        huffmanfile.compress(b"abc")
    assert str(ei.value) == "Fatal error. Failed to encode the data"


# ------------------------------------------------------------------ GPU: behaviour
gpu = pytest.mark.gpu


@gpu
def test_compress_decompress():                              # huffmanfile_test.py:8-12
    data = b"a" * 1000
    c = huffmanfile.compress(data)
    assert len(c) == 145 and hashlib.sha256(c).hexdigest().startswith("c4270ae06ed145d8")
    assert huffmanfile.decompress(c) == data


@gpu
def test_decompress_corrupted():                             # huffmanfile_test.py:15-18
    with pytest.raises(huffmanfile.HuffmanError) as ei:
        huffmanfile.decompress(b"\x08\x00\x00\x00\x00\x00\x00\x00\x02\x00")
    assert str(ei.value) == "Failed on read/write operation. Failed to decode the data"


@gpu
def test_compress_incremental():                             # huffmanfile_test.py:21-34
    comp = huffmanfile.HuffmanCompressor()
    out, data = b"", b""
    for _ in range(10):
        part = b"z" * 1000
        out += comp.compress(part)
        data += part
    out += comp.flush()
    assert len(out) == 1270
    assert huffmanfile.decompress(out) == data
    assert out == huffmanfile.compress(data)


@gpu
def test_incremental_equals_one_shot_across_blocks():
    """The reference loses buffered bytes here; incremental must equal one-shot."""
    data = datagen.zipf255(300000).tobytes()
    comp = huffmanfile.HuffmanCompressor(65536)
    out = b"".join(comp.compress(data[i:i + 7001]) for i in range(0, len(data), 7001)) + comp.flush()
    assert out == huffmanfile.compress(data, 65536)
    assert huffmanfile.decompress(out) == data


@gpu
def test_memstream_direct_copy_equals_callback_path(monkeypatch):
    """huf_memopen streams are copied to/from the device directly; HUF_GPU_ZERO_COPY=0 forces the
    read()/write() callbacks that every foreign stream uses.  Same bytes either way, several
    GPU rounds per call (HUF_GPU_BATCH_MB=1), and the same error for a truncated stream."""
    data = datagen.zipf255(3 * (1 << 20) + 12345).tobytes()
    monkeypatch.setenv("HUF_GPU_BATCH_MB", "1")
    outs, backs, errs = [], [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("HUF_GPU_ZERO_COPY", mode)
        comp = huffmanfile.compress(data, 65536)
        outs.append(comp)
        backs.append(huffmanfile.decompress(comp))
        with pytest.raises(huffmanfile.HuffmanError) as ei:
            huffmanfile.decompress(comp[:-100])
        errs.append(str(ei.value))
    assert outs[0] == outs[1] and backs[0] == backs[1] == data
    assert errs[0] == errs[1]


@gpu
def test_write_file_text_mode(tmp_path):                     # huffmanfile_test.py:37-54
    text = "Donec rhoncus quis sapien sit amet molestie.\nhéllo\n" * 40
    name = tmp_path / "archive.hm"
    with huffmanfile.open(name, "wt", encoding="utf-8") as f:
        f.write(text)
    with huffmanfile.open(name, "rt", encoding="utf-8") as f:
        assert f.read() == text


@gpu
def test_file_bytes_equal_compress_and_chunked_reads(tmp_path):
    data = datagen.logtext(400000).tobytes()
    name = tmp_path / "a.hm"
    with huffmanfile.HuffmanFile(name, "w") as f:
        assert f.write(data[:150000]) == 150000
        f.write(memoryview(data)[150000:])
    assert name.read_bytes() == huffmanfile.compress(data)            # default blocksize 131072
    with huffmanfile.HuffmanFile(name, "r") as f:
        got = b""
        while True:
            chunk = f.read(50001)
            if not chunk:
                break
            got += chunk
    assert got == data
    with huffmanfile.open(name) as f:
        assert f.read() == data and f.read() == b""


@gpu
def test_concatenated_streams_and_reuse():
    d = huffmanfile.HuffmanDecompressor()
    c1, c2 = huffmanfile.compress(b"hello hello"), huffmanfile.compress(b"world!", 4)
    assert d.decompress(c1 + c2) == b"hello helloworld!"
    assert d.decompress(c2) == b"world!"                               # reusable


@gpu
def test_golden_streams_through_python_layer(golden):
    for vec in golden["encode_small"]:
        if vec["blocksize"] == 0:
            continue
        data = bytes.fromhex(vec["input_hex"])
        assert huffmanfile.compress(data, vec["blocksize"]).hex() == vec["output_hex"], vec["name"]
    vec = next(v for v in golden["encode_large"] if v["generator"] == "logtext" and v["blocksize"] == 1 << 20)
    data = datagen.logtext(vec["n"]).tobytes()
    out = huffmanfile.compress(data, 1 << 20)                           # config 5 shape
    assert len(out) == vec["output_len"] and hashlib.sha256(out).hexdigest() == vec["output_sha256"]
    assert huffmanfile.decompress(out) == data


@gpu
def test_k256_block_strict_and_relaxed():
    from libhuffman_amd import _native
    data = bytes(range(256)) * 4
    c = huffmanfile.compress(data, 65536)
    with pytest.raises(huffmanfile.HuffmanError) as ei:
        huffmanfile.decompress(c)
    assert str(ei.value).startswith("Block is corrupted, Huffman tree has impossible size.")
    _native.load().huf_gpu_set_relaxed_tree(1)
    try:
        assert huffmanfile.decompress(c) == data
    finally:
        _native.load().huf_gpu_set_relaxed_tree(0)


@gpu
def test_read_is_streaming_in_bounded_rounds(tmp_path, monkeypatch):
    """HuffmanFile.read(size) (huffmanfile.py:152-162): the file is taken in pieces that
    end anywhere inside a block; every round decodes the whole blocks it holds and carries the rest.
    With 7 000-byte pieces over 4 KiB / 64 KiB blocks the carry path runs on nearly every round."""
    import io as _io
    data = datagen.zipf255(700001).tobytes()
    for bs in (4096, 65536):
        name = tmp_path / f"s{bs}.hm"
        with huffmanfile.HuffmanFile(name, "w", blocksize=bs) as f:
            f.write(data)
        monkeypatch.setattr(huffmanfile.HuffmanFile, "READ_PIECE", 7000)
        with huffmanfile.HuffmanFile(name, "r") as f:
            got, peak = bytearray(), 0
            while True:
                chunk = f.read(10007)
                peak = max(peak, len(f._plain), len(f._rest))
                if not chunk:
                    break
                assert len(chunk) <= 10007
                got += chunk
            assert bytes(got) == data
            assert peak < 20 * bs + 8 * 7000          # bounded by rounds, not by the file
        with huffmanfile.HuffmanFile(name, "r") as f:
            assert f.read(1) == data[:1] and f.read() == data[1:] and f.read(5) == b""
        # a file cut inside its last block: the bytes of the whole blocks, then the decoder's error
        raw = name.read_bytes()
        with huffmanfile.HuffmanFile(_io.BytesIO(raw[:-3]), "r") as f:
            with pytest.raises(huffmanfile.HuffmanError) as ei:
                f.read()
            assert "read/write" in str(ei.value)
        monkeypatch.undo()


@gpu
def test_one_block_file_is_not_decoded_piece_by_piece(tmp_path, monkeypatch):
    """blocksize = 0 (the reference's default: the whole file is ONE block, src/encoder.c:163-165) with a block
    far larger than a read round: the header says how many bytes the block must at least have, so no decode is
    attempted on a prefix that cannot hold it, and later reads grow geometrically - a handful of decode calls,
    not one per piece.  A failed read stays failed."""
    import io as _io
    data = datagen.zipf255(3 << 20).tobytes()
    name = tmp_path / "one.hm"
    with huffmanfile.HuffmanFile(name, "w", blocksize=len(data)) as f:    # one block, as the C API's blocksize = 0 writes it
        f.write(data)
    monkeypatch.setattr(huffmanfile.HuffmanFile, "READ_PIECE", 64 << 10)
    calls = []
    real = huffmanfile.HuffmanDecompressor.decompress_blocks

    def counting(self, buf):
        calls.append(len(buf))
        return real(self, buf)
    monkeypatch.setattr(huffmanfile.HuffmanDecompressor, "decompress_blocks", counting)
    with huffmanfile.HuffmanFile(name, "r") as f:
        assert f.read() == data
    assert 1 <= len(calls) <= 6, calls                       # (48 pieces of 64 KiB would be 48 growing decodes)
    raw = name.read_bytes()
    with huffmanfile.HuffmanFile(_io.BytesIO(raw[:len(raw) // 2]), "r") as f:
        for _ in range(2):                                    # the second read raises again instead of returning b""
            with pytest.raises(huffmanfile.HuffmanError) as ei:
                f.read(100)
            assert "read/write" in str(ei.value)


@gpu
def test_decode_blocks_pieces_through_the_c_api():
    """huf_gpu_decode_blocks: whole blocks inside the piece are decoded, *consumed stops in front of a
    cut-off block (no error), a damaged block is still huf_decode's error."""
    data = datagen.zipf255(5 * 65536 + 100).tobytes()
    comp = huffmanfile.compress(data, 65536)
    d = huffmanfile.HuffmanDecompressor()
    plain, used = d.decompress_blocks(comp)
    assert plain == data and used == len(comp)
    cut = len(comp) - 1000                                 # inside the last (short) block or the one before
    plain, used = d.decompress_blocks(comp[:cut])
    assert used < cut and data.startswith(plain) and len(plain) % 65536 == 0 and len(plain) >= 4 * 65536
    rest, used2 = d.decompress_blocks(comp[used:])
    assert plain + rest == data and used + used2 == len(comp)
    plain, used = d.decompress_blocks(comp[:5])            # not even a header
    assert (plain, used) == (b"", 0)
    bad = bytearray(comp)
    bad[8] = 0xff; bad[9] = 0x7f                           # tree_len 32767 in the first header
    with pytest.raises(huffmanfile.HuffmanError):
        d.decompress_blocks(bytes(bad))


@gpu
def test_large_results_take_the_threaded_copy():
    """results of 16 MiB and more are copied out by huf_gpu_copy_out (a few threads, odd part sizes):
    every byte, both directions"""
    for n in ((16 << 20) + 3, (45 << 20) + 12345):
        data = datagen.zipf255(n).tobytes()
        comp = huffmanfile.compress(data, blocksize=65536)
        assert len(comp) > (15 << 20)
        back = huffmanfile.decompress(comp)
        assert type(back) is bytes and len(back) == n and back == data
        # the stream equals the one the small-copy path produces piece by piece
        c = huffmanfile.HuffmanCompressor(blocksize=65536)
        parts = [c.compress(data[i:i + (4 << 20)]) for i in range(0, n, 4 << 20)] + [c.flush()]
        assert b"".join(parts) == comp


@gpu
def test_results_in_place_and_the_growable_stream_behind_them(monkeypatch):
    """compress()/decompress() of 8 MiB and more write straight into the bytes object they return
    (huf_gpu_memwrap_out); a result that does not fit the room that was made - here: a stream that begins like
    any other and then holds 200 MiB of one byte in 25 MiB - goes through the growable stream instead, and a
    stream that BEGINS with one-symbol blocks is given nine times its size at once.  Same bytes either way."""
    from libhuffman_amd import huffmanfile as hf
    assert hf._IN_PLACE
    head = datagen.zipf255(1 << 20).tobytes()
    data = head + bytes(200 << 20)
    comp = huffmanfile.compress(data, blocksize=1 << 20)
    assert (8 << 20) < len(comp) < (40 << 20)                  # in place, and more than four times smaller than the data
    back = huffmanfile.decompress(comp)                        # does not fit 4 x len(comp): the growable stream
    assert type(back) is bytes and len(back) == len(data) and back == data
    zeros = bytes(128 << 20)
    zc = huffmanfile.compress(zeros, blocksize=1 << 20)        # 16 MiB of one-symbol blocks
    assert len(zc) > (8 << 20) and huffmanfile.decompress(zc) == zeros
    # with the in-place path switched off the same bytes come out
    monkeypatch.setattr(hf, "_IN_PLACE", False)
    assert huffmanfile.compress(data, blocksize=1 << 20) == comp
    assert huffmanfile.decompress(comp) == data

