/*
 * huffman/config.h - huf_config_t - the argument of huf_encode()/huf_decode().
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/config.h:10-46 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_config_h__
#define INCLUDE_huffman_config_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "io.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_config_h__
typedef struct __huf_encoder_config {
    uint64_t length;              /* encode: input bytes; decode: COMPRESSED bytes to consume */
    uint64_t blocksize;           /* encode: bytes per block, 0 => one block of `length` */
    size_t reader_buffer_size;    /* 0 => unbuffered; output is identical either way */
    size_t writer_buffer_size;
    huf_read_writer_t *reader;
    huf_read_writer_t *writer;
} huf_config_t;                   /* 48 bytes on LP64 - part of the ABI */

huf_error_t huf_config_init(huf_config_t **self);
huf_error_t huf_config_free(huf_config_t **self);
#undef CFFI_huffman_config_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_config_h__ */
