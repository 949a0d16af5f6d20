/*
 * huffman/common.h - buffer size constants.
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/common.h:7-17 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_common_h__
#define INCLUDE_huffman_common_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HUF_1KIB_BUFFER   1024
#define HUF_64KIB_BUFFER  65536
#define HUF_128KIB_BUFFER 131072
#define HUF_256KIB_BUFFER 262144
#define HUF_512KIB_BUFFER 524288
#define HUF_1MIB_BUFFER   1048576

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_common_h__ */
