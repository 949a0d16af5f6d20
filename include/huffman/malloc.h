/*
 * huffman/malloc.h - huf_malloc().
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/malloc.h:10-11 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_malloc_h__
#define INCLUDE_huffman_malloc_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "errors.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_malloc_h__
/* Zero-initialised allocation of num elements of `size` bytes (calloc semantics). */
huf_error_t huf_malloc(void** ptr, size_t size, size_t num);
#undef CFFI_huffman_malloc_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_malloc_h__ */
