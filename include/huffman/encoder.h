/*
 * huffman/encoder.h - huf_encode(): the block loop of src/encoder.c:261-388, run on the MI355X.
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/encoder.h:11-26 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_encoder_h__
#define INCLUDE_huffman_encoder_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_encoder_h__
typedef struct __huf_encoder huf_encoder_t;
huf_error_t huf_encoder_init(huf_encoder_t **self, const huf_config_t *config);
huf_error_t huf_encoder_free(huf_encoder_t **self);
/* Read config->length bytes from config->reader, write the block stream to config->writer. */
huf_error_t huf_encode(const huf_config_t *config);
#undef CFFI_huffman_encoder_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_encoder_h__ */
