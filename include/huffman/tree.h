/*
 * huffman/tree.h - huf_tree_t (host-callable building block).
 *
 * Same file name, declarations, struct layouts and CFFI markers as the reference's
 * include/huffman/tree.h:10-82 (the text between "#define CFFI_x" and "#undef CFFI_x" is what the
 * reference's setup_ffi.py:8-23 cuts out for cffi's cdef()); served by libhuffman_amd/libhuffman.so.
 */
#ifndef INCLUDE_huffman_tree_h__
#define INCLUDE_huffman_tree_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "histogram.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HUF_ASCII_COUNT   256
#define HUF_BTREE_LEN     1024
#define HUF_HISTOGRAM_LEN 512
#define HUF_LEAF_NODE     -1

#define CFFI_huffman_tree_h__
typedef struct __huf_node {
    int16_t index;                 /* byte value (leaf) or creation index >= 256 */
    struct __huf_node *parent;
    struct __huf_node *left;
    struct __huf_node *right;
} huf_node_t;

typedef struct __huf_tree {
    huf_node_t **leaves;           /* 512 slots */
    huf_node_t *root;
} huf_tree_t;

huf_error_t huf_node_to_string(const huf_node_t *self, uint8_t *buf, size_t *len);
huf_error_t huf_tree_init(huf_tree_t **self);
huf_error_t huf_tree_free(huf_tree_t **self);
huf_error_t huf_tree_reset(huf_tree_t *self);
huf_error_t huf_tree_deserialize(huf_tree_t *self, const int16_t *buf, size_t len);
huf_error_t huf_tree_serialize(huf_tree_t *self, int16_t *buf, size_t *len);
huf_error_t huf_tree_from_histogram(huf_tree_t *self, huf_histogram_t *histogram);
#undef CFFI_huffman_tree_h__

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_tree_h__ */
