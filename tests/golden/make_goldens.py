#!/usr/bin/env python3
"""Capture known-answer vectors from the UNMODIFIED reference (oracle/_ref/libhuffman_ref.so,
built from /root/reference/src by `make -C oracle ref`) into tests/golden/vectors.json.

Runs only in the authoring container (the reference does not exist on the GPU box); the JSON
it writes is data: inputs (literal bytes or generator parameters) and expected outputs (hex
for small streams, sha256 + length for large ones).  Re-run:  python tests/golden/make_goldens.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libhuffman_amd import datagen  # noqa: E402
from oracle.oracle import Reference, build  # noqa: E402


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def main() -> None:
    build(ref=True)
    ref = Reference()
    vectors = {"_about": "captured from the unmodified reference by tests/golden/make_goldens.py",
               "encode_small": [], "encode_large": [], "decode_errors": [], "decode_ok": []}

    # ---- small streams, stored whole (hex) ------------------------------------------
    small = [
        ("one_symbol_test_encode_nobuffer", b"1", 256, 0, 0),             # test/encode_test.c:12-35
        ("readme_0123456789_bs0_buf128", b"0123456789", 0, 128, 128),     # test/encode_test.c:48-94
        ("readme_0123456789_bs65536", b"0123456789", 65536, 0, 0),        # README.md:37-56
        ("aab", b"aab", 0, 0, 0),
        ("abracadabra", b"abracadabra", 0, 0, 0),
        ("abcabcab_bs4", b"abcabcab", 4, 0, 0),
        ("abcabc_bs4_rbuf3_wbuf5", b"abcabc", 4, 3, 5),
        ("a1000_bs131072", b"a" * 1000, 131072, 0, 0),                    # huffmanfile_test.py:8-12
        ("z10000_bs131072", b"z" * 10000, 131072, 0, 0),                  # huffmanfile_test.py:21-34
        ("two_symbols_skewed", b"a" * 300 + b"b", 0, 0, 0),
        ("all_bytes_once_k256", bytes(range(256)), 0, 0, 0),
        ("fib_weights", b"".join(bytes([65 + i]) * f for i, f in
                                 enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144])), 0, 0, 0),
    ]
    for k in (2, 3, 128, 254, 255, 256):
        small.append((f"range{k}_x4", bytes(range(k)) * 4, 65536, 0, 0))
    for name, data, bs, rb, wb in small:
        out = ref.encode(data, bs, rb, wb)
        err, back = ref.decode(out, raw_hint=len(data) + 64)
        vectors["encode_small"].append({
            "name": name, "input_hex": data.hex(), "blocksize": bs,
            "reader_buffer": rb, "writer_buffer": wb,
            "output_hex": out.tobytes().hex(), "output_len": int(out.size),
            "ref_decode_err": int(err), "ref_roundtrip": bool(err == 0 and back.tobytes() == data),
        })

    # ---- generator-defined inputs, stored as digests ------------------------------------
    large = [
        ("const41", 262144, 65536), ("uniform256", 262144, 65536), ("uniform255", 262144, 65536),
        ("zipf255", 262144, 65536), ("zipf255", 2 * 1048576, 1048576), ("zipf255", 65536 + 1000, 65536),
        ("logtext", 2 * 1048576, 1048576), ("logtext", 262144, 65536),
        ("uniform255", 3 * 4096 + 17, 4096), ("zipf255", 1 << 20, 0),
    ]
    for gen, n, bs in large:
        data = datagen.GENERATORS[gen](n)
        out = ref.encode(data, bs)
        err, back = ref.decode(out, raw_hint=n + 64)
        vectors["encode_large"].append({
            "generator": gen, "n": n, "blocksize": bs, "input_sha256": sha(data),
            "output_len": int(out.size), "output_sha256": sha(out),
            "ref_decode_err": int(err),
            "ref_roundtrip": bool(err == 0 and back.tobytes() == data.tobytes()),
        })

    # ---- decode error / edge vectors (test/decode_test.c:32-74, huffmanfile_test.py:15-18) ----
    one = ref.encode(b"1", 256)
    errs = [
        ("empty_input", b"", None),
        ("ten_0x0a_tree_overflow", bytes([10] * 10), None),
        ("truncated_tree_length10", bytes([8, 0, 0, 0, 0, 0, 0, 0, 8, 0, 10, 10, 10, 10]), 10),
        ("childless_root_corrupted", np.array([8, 0, 0, 0, 3, 0, -1, -1, 1, 2, 3], dtype="<i2").tobytes(), None),
        ("py_corrupted_header", bytes([8, 0, 0, 0, 0, 0, 0, 0, 2, 0]), None),
        ("valid_block_plus_2_trailing", one.tobytes() + b"\x00\x00", None),
        ("zero_length_block_header_only", np.array([0, 0, 0, 0, 5, 256, 65, -1, -1, -1], dtype="<i2").tobytes(), None),
        ("right_edge_walk", np.array([2, 0, 0, 0, 7, 300, 301, 97, -1, -1, -1, 98], dtype="<i2").tobytes() + b"\x40", None),
        ("right_leaf_by_exhaustion", np.array([2, 0, 0, 0, 7, 300, 301, 97, -1, -1, -1, 98], dtype="<i2").tobytes() + b"\x80", None),
        ("truncated_payload", ref.encode(b"abracadabra", 0).tobytes()[:-2], None),
        ("negative_tree_len", np.array([1, 0, 0, 0, -3], dtype="<i2").tobytes(), None),
        ("k256_rejected_strict", ref.encode(bytes(range(256)), 0).tobytes(), None),
    ]
    for name, stream, length in errs:
        err, out = ref.decode(stream, raw_hint=4096, length=length)
        vectors["decode_errors"].append({
            "name": name, "stream_hex": stream.hex(), "length": length,
            "err": int(err), "output_hex": out.tobytes().hex(),
        })

    # concatenated streams decode to the concatenation (huffmanfile.py:388-389)
    c1, c2 = ref.encode(b"hello hello", 0), ref.encode(b"world!", 4)
    err, out = ref.decode(c1.tobytes() + c2.tobytes(), raw_hint=64)
    vectors["decode_ok"].append({"name": "concat", "stream_hex": (c1.tobytes() + c2.tobytes()).hex(),
                                 "err": int(err), "output_hex": out.tobytes().hex()})

    path = os.path.join(ROOT, "tests", "golden", "vectors.json")
    with open(path, "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
