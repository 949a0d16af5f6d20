"""hufgpu_encode_small / hufgpu_decode_small through the C ABI (include/huffman_gpu.h): pinned host memory in, one
synchronisation, pinned host memory out - against the oracle (src/encoder.c:261-388, src/decoder.c:34-96, 201-287 restated):
streams, lengths, error codes, delivered bytes; whole streams, streams of several blocks, truncated and damaged ones."""
import ctypes as C

import numpy as np
import pytest

from libhuffman_amd import datagen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from libhuffman_amd.codec import GpuCodec
    from oracle.oracle import Oracle
    c = GpuCodec(0)
    L = c.lib
    vp, u64 = C.c_void_p, C.c_uint64
    L.hufgpu_encode_small.argtypes = [vp, vp, u64, u64, vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.hufgpu_decode_small.argtypes = [vp, vp, u64, u64, C.c_uint32, vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    return torch, c, L, Oracle()


def small_encode(torch, c, L, data: np.ndarray, bs: int):
    n = data.size
    bound = c.encode_bound(n, bs)
    h_in = torch.from_numpy(data.copy()).pin_memory() if n else torch.zeros(1, dtype=torch.uint8).pin_memory()
    d_in = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
    d_out = torch.empty(bound + 8, dtype=torch.uint8, device="cuda")
    h_out = torch.zeros(((bound + 7) & ~7) + 8, dtype=torch.uint8).pin_memory()
    ln = C.c_uint64()
    err = L.hufgpu_encode_small(c._ctx, h_in.data_ptr(), n, bs, d_in.data_ptr(), d_out.data_ptr(), d_out.numel(),
                                h_out.data_ptr(), h_out.numel(), C.byref(ln))
    return err, h_out[: ln.value].numpy().copy()


def small_decode(torch, c, L, stream: np.ndarray, avail: int, length: int, out_cap: int, relaxed=False):
    h_in = torch.from_numpy(stream[:max(avail, 1)].copy()).pin_memory()
    d_in = torch.empty(max(avail, 1), dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(out_cap, dtype=torch.uint8, device="cuda")
    copy = min(out_cap, 8 * avail + 64)
    h_out = torch.zeros(((copy + 7) & ~7) + 48, dtype=torch.uint8).pin_memory()
    raw, used = C.c_uint64(), C.c_uint64()
    err = L.hufgpu_decode_small(c._ctx, h_in.data_ptr(), avail, length, 1 if relaxed else 0, d_in.data_ptr(), d_out.data_ptr(),
                                out_cap, h_out.data_ptr(), h_out.numel(), C.byref(raw), C.byref(used))
    return err, h_out[: raw.value].numpy().copy(), int(used.value)


def test_small_encode_is_the_oracles_stream(env):
    torch, c, L, oracle = env
    rng = np.random.default_rng(11)
    for n, bs in ((1, 65536), (10, 65536), (255, 64), (1000, 256), (4096, 1024), (32768, 65536), (30000, 7000)):
        for data in (datagen.zipf255(n), rng.integers(0, 256, n, dtype=np.uint8), np.full(n, 0x41, dtype=np.uint8)):
            err, got = small_encode(torch, c, L, data, bs)
            want = oracle.encode(data, bs)
            assert err == 0 and got.size == want.size and np.array_equal(got, want), (n, bs)


def test_small_decode_matches_the_oracle_whole_cut_and_damaged(env):
    torch, c, L, oracle = env
    rng = np.random.default_rng(12)
    for n, bs in ((1, 65536), (10, 65536), (300, 100), (5000, 1024), (32768, 8192)):
        data = datagen.zipf255(n) if n > 1 else np.array([7], dtype=np.uint8)
        stream = oracle.encode(data, bs)
        cases = [(stream.copy(), stream.size, stream.size)]
        cases.append((stream.copy(), stream.size - 1, stream.size))                      # the last byte is missing
        cases.append((stream.copy(), stream.size, max(1, stream.size // 2)))             # `length` ends inside the stream (decoder.c:218)
        for _ in range(6):
            bad = stream.copy()
            pos = int(rng.integers(0, stream.size))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            cases.append((bad, stream.size, stream.size))
        for s, avail, length in cases:
            cap = n + 4096
            err, out, used = small_decode(torch, c, L, s, avail, length, cap)
            oerr, oout, oused = oracle.decode(s[:avail], cap, 1024, length=length)
            if oerr == 1:                        # (the oracle ran out of its buffer: a damaged block_len - the device says the same)
                assert err == 1, (n, bs, err)
                continue
            assert err == oerr and out.size == oout.size and np.array_equal(out, oout), (n, bs, avail, length, err, oerr, out.size, oout.size)
            if oerr == 0:
                assert used == oused            # (a damaged stream that still decodes gives other bytes: the oracle's, checked above)


def test_small_calls_say_no_to_bad_arguments(env):
    torch, c, L, _ = env
    raw, used = C.c_uint64(), C.c_uint64()
    assert L.hufgpu_decode_small(c._ctx, None, 10, 10, 0, None, None, 0, None, 0, C.byref(raw), C.byref(used)) == 2
    h = torch.zeros(64, dtype=torch.uint8).pin_memory()
    d = torch.zeros(64, dtype=torch.uint8, device="cuda")
    # a result buffer that cannot hold output + outcome
    assert L.hufgpu_decode_small(c._ctx, h.data_ptr(), 10, 10, 0, d.data_ptr(), d.data_ptr(), 64, h.data_ptr(), 8, C.byref(raw), C.byref(used)) == 2
