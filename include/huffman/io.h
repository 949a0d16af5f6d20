/*
 * huffman/io.h - huf_read_writer_t, memory and file-descriptor streams.
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/io.h:11-31 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_io_h__
#define INCLUDE_huffman_io_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "errors.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_io_h__
typedef struct __huf_read_writer {
    void *stream;
    /* consume exactly `count` bytes or fail */
    huf_error_t (*write)(void *stream, const void *buf, size_t count);
    /* in: *count bytes wanted; out: *count bytes delivered (fewer is not an error here) */
    huf_error_t (*read)(void *stream, void *buf, size_t *count);
} huf_read_writer_t;

/* Growable in-memory stream. The caller owns *buf (it may be replaced on growth, so re-read
 * the pointer after writes); huf_memclose frees the stream objects only. */
huf_error_t huf_memopen(huf_read_writer_t **self, void **buf, size_t capacity);
huf_error_t huf_memlen(const huf_read_writer_t *self, size_t *len);
huf_error_t huf_memcap(const huf_read_writer_t *self, size_t *cap);
huf_error_t huf_memrewind(huf_read_writer_t *self);   /* truncate: len = off = 0 */
huf_error_t huf_memclose(huf_read_writer_t **self);

/* File-descriptor stream. */
huf_error_t huf_fdopen(huf_read_writer_t **self, int fd);
huf_error_t huf_fdclose(huf_read_writer_t **self);
#undef CFFI_huffman_io_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_io_h__ */
