"""Device-resident codec object over the hufgpu_* C ABI (include/huffman_gpu.h).

``GpuCodec`` works on torch uint8 tensors that already live in HBM: torch is used for device
memory and streams only, every byte of codec work happens in the HIP kernels of
``csrc/kernels/*.hpp`` (one translation unit, ``csrc/hufgpu_kernels.hip``) through the C ABI.
There is no eager/CPU path: constructing a codec without a usable MI355X raises
``HuffmanGpuError``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native

FILL_KINDS = {"const41": 0, "uniform256": 1, "uniform255": 2, "zipf255": 3}
FILL_SEEDS = {"const41": 0, "uniform256": 1, "uniform255": 2, "zipf255": 3}


class HuffmanGpuError(RuntimeError):
    def __init__(self, err: int, context: str, detail: str = ""):
        self.err = err
        self.raw = None
        msg = f"{_native.error_string(err)}. {context}"
        if detail:
            msg += f" ({detail})"
        super().__init__(msg)


class GpuCodec:
    """One codec context per device. Not thread-safe (like the reference's objects)."""

    def __init__(self, device: int = 0):
        self.lib = _native.load()
        self.device = device
        self._ctx = C.c_void_p()
        err = self.lib.hufgpu_ctx_create(C.byref(self._ctx), device)
        if err:
            raise HuffmanGpuError(err, "Failed to create the GPU codec context",
                                  self.lib.hufgpu_last_error(None).decode())
        self.tdev = torch.device("cuda", device)
        self._pending_decode = None

    def close(self):
        if self._ctx:
            self.lib.hufgpu_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------
    def _check(self, err: int, what: str, raw: int | None = None):
        if err:
            e = HuffmanGpuError(err, what, self.lib.hufgpu_last_error(self._ctx).decode())
            e.raw = raw                 # decode: bytes delivered before the failure (src/decoder.c:69-91)
            raise e

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

    def block_count(self, n: int, blocksize: int) -> int:
        return int(self.lib.hufgpu_block_count(n, blocksize))

    def encode_bound(self, n: int, blocksize: int) -> int:
        return int(self.lib.hufgpu_encode_bound(n, blocksize))

    def sub_index_bytes(self, n: int, blocksize: int) -> int:
        """Size of the optional sub-index of an encode of n bytes (see include/huffman_gpu.h)."""
        return int(self.lib.hufgpu_sub_index_bytes(n, blocksize))

    def new_sub_index(self, n: int, blocksize: int) -> torch.Tensor:
        return torch.empty(max(1, (self.sub_index_bytes(n, blocksize) + 7) // 8), dtype=torch.int64, device=self.tdev)

    # -- hot path ---------------------------------------------------------------------------
    def histogram(self, data: torch.Tensor, blocksize: int) -> torch.Tensor:
        n = data.numel()
        nb = self.block_count(n, blocksize)
        hist = torch.empty((nb, 256), dtype=torch.int32, device=self.tdev)
        self._check(self.lib.hufgpu_histogram(self._ctx, data.data_ptr(), n, blocksize,
                                              hist.data_ptr(), self._stream()), "histogram failed")
        return hist

    def encode(self, data: torch.Tensor, blocksize: int, out: torch.Tensor | None = None,
               offsets: torch.Tensor | None = None, sync: bool = True, sub_index: torch.Tensor | None = None):
        """Returns (stream tensor view, offsets tensor[nblocks+1], length or None).  `sub_index`
        (from new_sub_index) also receives the encoder's sub-index for decode(..., sub_index=)."""
        assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous()
        n = data.numel()
        nb = self.block_count(n, blocksize)
        if out is None:
            out = torch.empty(self.encode_bound(n, blocksize), dtype=torch.uint8, device=self.tdev)
        if offsets is None:
            offsets = torch.empty(nb + 1, dtype=torch.int64, device=self.tdev)
        out_len = C.c_uint64(0)
        if sub_index is not None:
            assert sub_index.numel() * sub_index.element_size() >= self.sub_index_bytes(n, blocksize)
            err = self.lib.hufgpu_encode_sub(self._ctx, data.data_ptr(), n, blocksize, out.data_ptr(),
                                             out.numel(), offsets.data_ptr(), sub_index.data_ptr(),
                                             C.byref(out_len) if sync else None, self._stream())
        else:
            err = self.lib.hufgpu_encode(self._ctx, data.data_ptr(), n, blocksize, out.data_ptr(),
                                         out.numel(), offsets.data_ptr(),
                                         C.byref(out_len) if sync else None, self._stream())
        self._check(err, "Failed to encode the data")
        if sync:
            return out[: out_len.value], offsets, int(out_len.value)
        return out, offsets, None

    def decode(self, stream: torch.Tensor, stream_len: int, offsets: torch.Tensor, nblocks: int,
               out: torch.Tensor, relaxed: bool = False, sync: bool = True,
               sub_index: torch.Tensor | None = None, raw_size: int = 0, blocksize: int = 0):
        """Indexed decode. Returns bytes written (sync) or None (enqueued only).  With `sub_index`
        (as written by encode of `raw_size` bytes in blocks of `blocksize`) every symbol is decoded
        once; the sub-index is verified on the device, never trusted."""
        raw = C.c_uint64(0)
        flags = _native.RELAXED_TREE if relaxed else _native.STRICT_TREE
        if sub_index is not None:
            assert self.block_count(raw_size, blocksize) == nblocks
            err = self.lib.hufgpu_decode_sub(self._ctx, stream.data_ptr(), stream_len, offsets.data_ptr(),
                                             raw_size, blocksize, sub_index.data_ptr(), out.data_ptr(),
                                             out.numel(), flags, C.byref(raw) if sync else None, self._stream())
        else:
            err = self.lib.hufgpu_decode(self._ctx, stream.data_ptr(), stream_len, offsets.data_ptr(),
                                         nblocks, out.data_ptr(), out.numel(), flags,
                                         C.byref(raw) if sync else None, self._stream())
        # an enqueued decode's buffers are read - and, after a failed block, written - by hufgpu_decode_result():
        # they stay referenced until then (include/huffman_gpu.h)
        self._pending_decode = None if sync else (stream, offsets, out, sub_index)
        self._check(err, "Failed to decode the data", raw=int(raw.value) if sync else None)
        return int(raw.value) if sync else None

    def decode_result(self) -> int:
        raw = C.c_uint64(0)
        try:
            self._check(self.lib.hufgpu_decode_result(self._ctx, C.byref(raw)), "Failed to decode the data", raw=int(raw.value))
        finally:
            self._pending_decode = None
        return int(raw.value)

    CALIB_VARIANTS = 8

    def calib_bandwidth(self, kind: str, variant: int, a: torch.Tensor | None, b: torch.Tensor | None, nbytes: int):
        """One launch of the bandwidth calibration kernel (kind: "copy" a -> b, "read" a, "fill" b)."""
        k = {"copy": 0, "read": 1, "fill": 2}[kind]
        self._check(self.lib.hufgpu_calib_bandwidth(self._ctx, k, variant, a.data_ptr() if a is not None else None,
                                                    b.data_ptr() if b is not None else None, nbytes, self._stream()),
                    "calibration launch failed")

    def decode_counters(self):
        """(blocks the exact decoder took, blocks the one-pass index-only decoder handed on) of the last decode."""
        c = (C.c_uint32 * 2)()
        self._check(self.lib.hufgpu_decode_counters(self._ctx, c), "counter readout failed")
        return int(c[0]), int(c[1])

    def decode_stream(self, stream: torch.Tensor, avail: int, length: int, out: torch.Tensor,
                      relaxed: bool = False, sequential: bool = False):
        """Raw-stream decode (no index). Returns (err, bytes written, bytes consumed).
        sequential=True forces the in-order decoder (same results, for cross-checking)."""
        raw, used = C.c_uint64(0), C.c_uint64(0)
        flags = (_native.RELAXED_TREE if relaxed else _native.STRICT_TREE) | (2 if sequential else 0)
        err = self.lib.hufgpu_decode_stream(self._ctx, stream.data_ptr() if stream.numel() else None,
                                            avail, length, out.data_ptr(), out.numel(), flags,
                                            C.byref(raw), C.byref(used), self._stream())
        return int(err), int(raw.value), int(used.value)

    def fill(self, out: torch.Tensor, kind: str, first: int = 0, seed: int | None = None):
        seed = FILL_SEEDS[kind] if seed is None else seed
        self._check(self.lib.hufgpu_fill(self._ctx, out.data_ptr(), out.numel(), FILL_KINDS[kind],
                                         seed, first, self._stream()), "fill failed")
        return out

    def set_profiling(self, on: bool, resume: bool = False):
        """on: record HIP events around every kernel; resume=True keeps what was recorded so far."""
        self.lib.hufgpu_set_profiling(self._ctx, (2 if resume else 1) if on else 0)

    ENCODE_KERNELS = ("hist256", "tree", "scan_sizes", "pack")
    DECODE_KERNELS = ("prepare_scan", "decode")

    def profile(self, kind: str):
        """Per-kernel milliseconds summed over the profiled calls -> (dict name->ms, calls)."""
        ms = (C.c_float * 8)()
        n, calls = C.c_int(0), C.c_int(0)
        k = 0 if kind == "encode" else 1
        self._check(self.lib.hufgpu_get_profile(self._ctx, k, ms, 8, C.byref(n), C.byref(calls)),
                    "profile readout failed")
        names = self.ENCODE_KERNELS if k == 0 else self.DECODE_KERNELS
        return {names[i]: float(ms[i]) for i in range(min(n.value, len(names)))}, calls.value
