/*
 * huffman.h - the libhuffman C API, served by the MI355X-native codec.
 *
 * This header declares, with identical names, argument meaning, struct layout and error
 * numbering, the interface that ybubnov/libhuffman v1.0.3 exports, so that a program (or the
 * huffmanfile CFFI/ctypes binding) written against the reference links against this library
 * unchanged.  Each group below cites the reference header it replaces (paths relative to the
 * reference checkout).
 *
 * What runs where:
 *   - huf_encode() / huf_decode() move whole batches of blocks to the GPU and run the
 *     histogram / tree / bit-pack / table-decode kernels there (csrc/hufgpu_kernels.hip).
 *     There is NO CPU fallback: without a usable gfx950 device they return HUF_ERROR_FATAL and
 *     print the reason on stderr.
 *   - everything else in this header is host-side plumbing (streams, buffers, error strings)
 *     or host-callable building blocks that the reference also exports.
 *
 * The declarations live in include/huffman/<name>.h, one file per reference header and under the
 * reference's file names; the "#define CFFI_x / #undef CFFI_x" markers in them delimit the text a
 * cffi cdef() is cut from, exactly as setup_ffi.py:8-43 of the reference expects.  This file pulls
 * them all in (the reference's include/huffman.h:4-9 pulls in six of them).
 */
#ifndef INCLUDE_huffman_h__
#define INCLUDE_huffman_h__

#include "huffman/common.h"
#include "huffman/errors.h"
#include "huffman/malloc.h"
#include "huffman/io.h"
#include "huffman/config.h"
#include "huffman/encoder.h"
#include "huffman/decoder.h"
#include "huffman/bufio.h"
#include "huffman/histogram.h"
#include "huffman/symbol.h"
#include "huffman/tree.h"

#endif /* INCLUDE_huffman_h__ */
