/*
 * huffman/bufio.h - buffered stream and bit writer (host plumbing).
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/bufio.h:13-93 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_bufio_h__
#define INCLUDE_huffman_bufio_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "io.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_bufio_h__
typedef struct __huf_bufio_read_writer {
    uint8_t *bytes;
    size_t offset;                 /* read position inside bytes */
    size_t capacity;
    size_t length;
    uint64_t have_been_processed;  /* bytes accepted (writer) / delivered (reader) so far */
    huf_read_writer_t *read_writer;
} huf_bufio_read_writer_t;

typedef struct __huf_bit_read_writer {
    uint8_t bits;
    uint8_t offset;                /* 8 = empty byte, 0 = full byte */
} huf_bit_read_writer_t;

void huf_bit_write(huf_bit_read_writer_t *self, uint8_t bit);       /* MSB first */
void huf_bit_read_writer_reset(huf_bit_read_writer_t *self);

huf_error_t huf_bufio_read_writer_init(huf_bufio_read_writer_t **self,
                                       huf_read_writer_t *read_writer, size_t size);
huf_error_t huf_bufio_read_writer_free(huf_bufio_read_writer_t **self);
huf_error_t huf_bufio_read_writer_flush(huf_bufio_read_writer_t *self);
huf_error_t huf_bufio_write(huf_bufio_read_writer_t *self, const void *buf, size_t size);
huf_error_t huf_bufio_read(huf_bufio_read_writer_t *self, void *buf, size_t size);
huf_error_t huf_bufio_read_uint8(huf_bufio_read_writer_t *self, uint8_t *byte);
huf_error_t huf_bufio_write_uint8(huf_bufio_read_writer_t *self, uint8_t byte);
#undef CFFI_huffman_bufio_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_bufio_h__ */
