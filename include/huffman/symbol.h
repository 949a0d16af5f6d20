/*
 * huffman/symbol.h - symbol -> coding map (host-callable building block).
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/symbol.h:10-79 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_symbol_h__
#define INCLUDE_huffman_symbol_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "errors.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_symbol_h__
typedef struct __huf_symbol_mapping_element {
    size_t length;
    uint8_t *coding;               /* ASCII '0'/'1', leaf -> root order */
} huf_symbol_mapping_element_t;

typedef struct __huf_symbol_mapping {
    size_t length;
    huf_symbol_mapping_element_t **symbols;
} huf_symbol_mapping_t;

huf_error_t huf_symbol_mapping_element_init(huf_symbol_mapping_element_t **self,
                                            const uint8_t *coding, size_t length);
huf_error_t huf_symbol_mapping_element_free(huf_symbol_mapping_element_t **self);
huf_error_t huf_symbol_mapping_init(huf_symbol_mapping_t **self, size_t length);
huf_error_t huf_symbol_mapping_free(huf_symbol_mapping_t **self);
huf_error_t huf_symbol_mapping_insert(huf_symbol_mapping_t *self, size_t position,
                                      huf_symbol_mapping_element_t *element);
huf_error_t huf_symbol_mapping_get(huf_symbol_mapping_t *self, size_t position,
                                   huf_symbol_mapping_element_t **element);
huf_error_t huf_symbol_mapping_reset(huf_symbol_mapping_t *self);
#undef CFFI_huffman_symbol_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_symbol_h__ */
