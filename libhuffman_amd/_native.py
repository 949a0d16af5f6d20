"""ctypes binding of libhuffman_amd/libhuffman.so (the product library).

Declares both boundaries of the shared library:
  * the libhuffman drop-in API of include/huffman.h (huf_encode, huf_decode, huf_memopen ...),
  * the device-resident API of include/huffman_gpu.h (hufgpu_*).

The reference binds the same C functions through CFFI (setup_ffi.py:30-65); the system
interpreter here has no cffi, so the binding is ctypes - same symbols, same struct layouts.
Loading fails loudly if the library has not been built (there is no pure-Python fallback).
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own libamdhip64.so.7; libhuffman.so needs the same SONAME.  Importing torch
# FIRST makes the loader reuse torch's copy, so the process holds ONE HIP runtime.  In the other
# order the system runtime is loaded for us, torch then loads its own, and the second runtime
# to initialise sees no device ("no ROCm-capable device is detected").
try:
    import torch as _torch  # noqa: F401
except ImportError:          # pure-C / ctypes users without torch: the system runtime is used
    _torch = None

from . import build as _build

_LIB = None

HUF_ERROR_SUCCESS = 0
HUF_ERROR_MEMORY_ALLOCATION = 1
HUF_ERROR_INVALID_ARGUMENT = 2
HUF_ERROR_READ_WRITE = 3
HUF_ERROR_FATAL = 4
HUF_ERROR_BTREE_OVERFLOW = 5
HUF_ERROR_BTREE_CORRUPTED = 6

STRICT_TREE = 0
RELAXED_TREE = 1

WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t))


class ReadWriter(C.Structure):          # include/huffman.h: huf_read_writer_t (24 bytes)
    _fields_ = [("stream", C.c_void_p), ("write", WRITE_FN), ("read", READ_FN)]


class Config(C.Structure):              # include/huffman.h: huf_config_t (48 bytes)
    _fields_ = [("length", C.c_uint64), ("blocksize", C.c_uint64),
                ("reader_buffer_size", C.c_size_t), ("writer_buffer_size", C.c_size_t),
                ("reader", C.POINTER(ReadWriter)), ("writer", C.POINTER(ReadWriter))]


HOST_SYMBOLS = """fdread fdwrite memread memwrite huf_bit_read_writer_reset huf_bit_write huf_bufio_read
huf_bufio_read_uint8 huf_bufio_read_writer_flush huf_bufio_read_writer_free
huf_bufio_read_writer_init huf_bufio_write huf_bufio_write_uint8 huf_config_free huf_config_init
huf_decode huf_decoder_free huf_decoder_init huf_encode huf_encoder_free huf_encoder_init
huf_error_string huf_fdclose huf_fdopen huf_histogram_free huf_histogram_init
huf_histogram_populate huf_histogram_reset huf_malloc huf_memcap huf_memclose huf_memlen
huf_memopen huf_memrewind huf_node_to_string huf_symbol_mapping_element_free
huf_symbol_mapping_element_init huf_symbol_mapping_free huf_symbol_mapping_get
huf_symbol_mapping_init huf_symbol_mapping_insert huf_symbol_mapping_reset huf_tree_deserialize
huf_tree_free huf_tree_from_histogram huf_tree_init huf_tree_reset huf_tree_serialize""".split()

GPU_SYMBOLS = """hufgpu_device_count hufgpu_ctx_create hufgpu_ctx_destroy hufgpu_last_error
hufgpu_block_count hufgpu_encode_bound hufgpu_histogram hufgpu_encode hufgpu_decode
hufgpu_decode_result hufgpu_decode_stream hufgpu_fill hufgpu_malloc hufgpu_free
hufgpu_memcpy_h2d hufgpu_memcpy_d2h hufgpu_memcpy_d2d hufgpu_synchronize hufgpu_set_profiling
hufgpu_get_profile hufgpu_sub_index_bytes hufgpu_encode_sub hufgpu_decode_sub
hufgpu_decode_stream_complete hufgpu_block_index hufgpu_decode_counters hufgpu_calib_bandwidth hufgpu_encode_small hufgpu_decode_small huf_gpu_set_relaxed_tree huf_gpu_memwrap huf_gpu_memwrap_out huf_gpu_decode_blocks huf_gpu_sessions huf_gpu_fanouts huf_gpu_copy_out
hufgpu_ctx_device hufgpu_shard_unique_id hufgpu_shard_create hufgpu_shard_destroy hufgpu_shard_info hufgpu_shard_last_error
hufgpu_shard_range hufgpu_shard_plan_decode hufgpu_encode_sharded hufgpu_decode_sharded hufgpu_shard_set_timeout""".split()


def so_path() -> str:
    return _build.SO_PATH


def load() -> C.CDLL:
    """Load the library, declaring argument types. Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = so_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m libhuffman_amd.build` "
            "(the codec has no pure-Python or CPU fallback)")
    L = C.CDLL(path)
    u64, vp, i32 = C.c_uint64, C.c_void_p, C.c_int
    L.huf_error_string.restype = C.c_char_p
    L.huf_error_string.argtypes = [i32]
    L.huf_memopen.argtypes = [C.POINTER(C.POINTER(ReadWriter)), C.POINTER(vp), C.c_size_t]
    L.huf_memclose.argtypes = [C.POINTER(C.POINTER(ReadWriter))]
    L.huf_memlen.argtypes = [C.POINTER(ReadWriter), C.POINTER(C.c_size_t)]
    L.huf_memcap.argtypes = [C.POINTER(ReadWriter), C.POINTER(C.c_size_t)]
    L.huf_memrewind.argtypes = [C.POINTER(ReadWriter)]
    L.huf_fdopen.argtypes = [C.POINTER(C.POINTER(ReadWriter)), i32]
    L.huf_fdclose.argtypes = [C.POINTER(C.POINTER(ReadWriter))]
    L.huf_encode.argtypes = [C.POINTER(Config)]
    L.huf_decode.argtypes = [C.POINTER(Config)]
    L.huf_gpu_set_relaxed_tree.argtypes = [i32]
    L.huf_gpu_set_relaxed_tree.restype = None
    L.huf_gpu_memwrap.argtypes = [C.POINTER(C.POINTER(ReadWriter)), vp, C.c_size_t]
    L.huf_gpu_memwrap_out.argtypes = [C.POINTER(C.POINTER(ReadWriter)), vp, C.c_size_t]
    L.huf_gpu_copy_out.argtypes = [vp, vp, C.c_size_t]
    L.huf_gpu_copy_out.restype = i32
    L.huf_gpu_sessions.argtypes = [C.POINTER(C.c_int)]
    L.huf_gpu_sessions.restype = i32
    L.huf_gpu_decode_blocks.argtypes = [C.POINTER(Config), C.POINTER(u64)]
    L.huf_gpu_fanouts.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.huf_gpu_fanouts.restype = i32

    L.hufgpu_device_count.restype = i32
    L.hufgpu_ctx_create.argtypes = [C.POINTER(vp), i32]
    L.hufgpu_ctx_destroy.argtypes = [vp]
    L.hufgpu_last_error.restype = C.c_char_p
    L.hufgpu_last_error.argtypes = [vp]
    L.hufgpu_block_count.restype = u64
    L.hufgpu_block_count.argtypes = [u64, u64]
    L.hufgpu_encode_bound.restype = u64
    L.hufgpu_encode_bound.argtypes = [u64, u64]
    L.hufgpu_histogram.argtypes = [vp, vp, u64, u64, vp, vp]
    L.hufgpu_encode.argtypes = [vp, vp, u64, u64, vp, u64, vp, C.POINTER(u64), vp]
    L.hufgpu_decode.argtypes = [vp, vp, u64, vp, u64, vp, u64, C.c_uint32, C.POINTER(u64), vp]
    L.hufgpu_sub_index_bytes.restype = u64
    L.hufgpu_sub_index_bytes.argtypes = [u64, u64]
    L.hufgpu_encode_sub.argtypes = [vp, vp, u64, u64, vp, u64, vp, vp, C.POINTER(u64), vp]
    L.hufgpu_decode_sub.argtypes = [vp, vp, u64, vp, u64, u64, vp, vp, u64, C.c_uint32, C.POINTER(u64), vp]
    L.hufgpu_decode_result.argtypes = [vp, C.POINTER(u64)]
    L.hufgpu_decode_counters.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.hufgpu_calib_bandwidth.argtypes = [vp, i32, i32, vp, vp, u64, vp]
    L.hufgpu_decode_stream_complete.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.hufgpu_decode_stream.argtypes = [vp, vp, u64, u64, vp, u64, C.c_uint32, C.POINTER(u64),
                                       C.POINTER(u64), vp]
    L.hufgpu_fill.argtypes = [vp, vp, u64, i32, u64, u64, vp]
    L.hufgpu_malloc.argtypes = [vp, C.POINTER(vp), u64]
    L.hufgpu_free.argtypes = [vp, vp]
    L.hufgpu_memcpy_h2d.argtypes = [vp, vp, vp, u64]
    L.hufgpu_memcpy_d2h.argtypes = [vp, vp, vp, u64]
    L.hufgpu_memcpy_d2d.argtypes = [vp, vp, vp, u64]
    L.hufgpu_synchronize.argtypes = [vp]
    L.hufgpu_set_profiling.argtypes = [vp, i32]
    L.hufgpu_get_profile.argtypes = [vp, i32, C.POINTER(C.c_float), i32, C.POINTER(i32), C.POINTER(i32)]
    L.hufgpu_ctx_device.argtypes = [vp]
    L.hufgpu_ctx_device.restype = i32
    L.hufgpu_shard_unique_id.argtypes = [vp]
    L.hufgpu_shard_create.argtypes = [C.POINTER(vp), vp, vp, vp, i32, i32]
    L.hufgpu_shard_destroy.argtypes = [vp]
    L.hufgpu_shard_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.hufgpu_shard_last_error.argtypes = [vp]
    L.hufgpu_shard_last_error.restype = C.c_char_p
    L.hufgpu_shard_set_timeout.argtypes = [vp, C.c_uint32]
    L.hufgpu_shard_range.argtypes = [u64, u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.hufgpu_shard_plan_decode.argtypes = [C.POINTER(u64), u64, i32, C.POINTER(u64)]
    L.hufgpu_encode_sharded.argtypes = [vp, i32, vp, u64, u64, C.c_uint32, vp, u64, vp, C.POINTER(u64), C.POINTER(u64),
                                        C.POINTER(C.c_double)]
    L.hufgpu_decode_sharded.argtypes = [vp, i32, vp, u64, vp, u64, u64, C.c_uint32, vp, u64, C.POINTER(u64),
                                        C.POINTER(C.c_double)]
    _LIB = L
    return L


def error_string(err: int) -> str:
    return load().huf_error_string(err).decode("utf-8")
