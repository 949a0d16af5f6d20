/*
 * huffman/decoder.h - huf_decode(): the block loop of src/decoder.c:201-287, run on the MI355X.
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/decoder.h:11-26 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_decoder_h__
#define INCLUDE_huffman_decoder_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_decoder_h__
typedef struct __huf_decoder huf_decoder_t;
huf_error_t huf_decoder_init(huf_decoder_t **self, const huf_config_t *config);
huf_error_t huf_decoder_free(huf_decoder_t **self);
/* Consume config->length compressed bytes from config->reader, write the original bytes. */
huf_error_t huf_decode(const huf_config_t *config);
#undef CFFI_huffman_decoder_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_decoder_h__ */
