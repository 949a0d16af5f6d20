/*
 * huffman/errors.h - huf_error_t and huf_error_string().
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/errors.h:6-31 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_errors_h__
#define INCLUDE_huffman_errors_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFFI_huffman_errors_h__
typedef enum {
    HUF_ERROR_SUCCESS,            /* 0 */
    HUF_ERROR_MEMORY_ALLOCATION,  /* 1 */
    HUF_ERROR_INVALID_ARGUMENT,   /* 2: e.g. a NULL pointer */
    HUF_ERROR_READ_WRITE,         /* 3: stream callback failed or delivered too few bytes */
    HUF_ERROR_FATAL,              /* 4: unrecoverable, incl. "no GPU" and HIP failures */
    HUF_ERROR_BTREE_OVERFLOW,     /* 5: serialized tree length outside [0, 1024] */
    HUF_ERROR_BTREE_CORRUPTED,    /* 6: bit walk left the tree */
} huf_error_t;

const char* huf_error_string(huf_error_t error);
#undef CFFI_huffman_errors_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_errors_h__ */
