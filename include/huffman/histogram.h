/*
 * huffman/histogram.h - huf_histogram_t (host-callable building block).
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/histogram.h:10-49 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_histogram_h__
#define INCLUDE_huffman_histogram_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "errors.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_histogram_h__
typedef struct __huf_histogram {
    uint64_t *frequencies;
    size_t iota;                   /* element width in bytes (the codec uses 1) */
    size_t length;                 /* number of counters */
    size_t start;                  /* smallest element seen, (size_t)-1 when empty */
} huf_histogram_t;

huf_error_t huf_histogram_init(huf_histogram_t **self, size_t iota, size_t length);
huf_error_t huf_histogram_free(huf_histogram_t **self);
huf_error_t huf_histogram_reset(huf_histogram_t *self);
huf_error_t huf_histogram_populate(huf_histogram_t *self, void *buf, size_t len);
#undef CFFI_huffman_histogram_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_histogram_h__ */
