/*
 * huffman.h - the libhuffman C API, served by the MI355X-native codec.
 *
 * This header declares, with identical names, argument meaning, struct layout and error
 * numbering, the interface that ybubnov/libhuffman v1.0.3 exports, so that a program (or the
 * huffmanfile CFFI/ctypes binding) written against the reference links against this library
 * unchanged.  Each group below cites the reference header it replaces (paths relative to the
 * reference checkout).
 *
 * What runs where:
 *   - huf_encode() / huf_decode() move whole batches of blocks to the GPU and run the
 *     histogram / tree / bit-pack / table-decode kernels there (csrc/hufgpu_kernels.hip).
 *     There is NO CPU fallback: without a usable gfx950 device they return HUF_ERROR_FATAL and
 *     print the reason on stderr.
 *   - everything else in this header is host-side plumbing (streams, buffers, error strings)
 *     or host-callable building blocks that the reference also exports.
 *
 * The "#define CFFI_x / #undef CFFI_x" markers delimit the text a cffi cdef() is cut from,
 * exactly as setup_ffi.py:8-23 of the reference expects.
 */
#ifndef INCLUDE_huffman_h__
#define INCLUDE_huffman_h__

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- include/huffman/common.h:7-17 ---------------------------------------------------- */
#define HUF_1KIB_BUFFER   1024
#define HUF_64KIB_BUFFER  65536
#define HUF_128KIB_BUFFER 131072
#define HUF_256KIB_BUFFER 262144
#define HUF_512KIB_BUFFER 524288
#define HUF_1MIB_BUFFER   1048576

/* ---- include/huffman/errors.h:6-31 ---------------------------------------------------- */
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

/* ---- include/huffman/malloc.h:10-11 --------------------------------------------------- */
#define CFFI_huffman_malloc_h__
/* Zero-initialised allocation of num elements of `size` bytes (calloc semantics). */
huf_error_t huf_malloc(void** ptr, size_t size, size_t num);
#undef CFFI_huffman_malloc_h__

/* ---- include/huffman/io.h:11-31 -------------------------------------------------------- */
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

/* ---- include/huffman/config.h:10-46 --------------------------------------------------- */
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

/* ---- include/huffman/encoder.h:11-26 -------------------------------------------------- */
#define CFFI_huffman_encoder_h__
typedef struct __huf_encoder huf_encoder_t;
huf_error_t huf_encoder_init(huf_encoder_t **self, const huf_config_t *config);
huf_error_t huf_encoder_free(huf_encoder_t **self);
/* Read config->length bytes from config->reader, write the block stream to config->writer. */
huf_error_t huf_encode(const huf_config_t *config);
#undef CFFI_huffman_encoder_h__

/* ---- include/huffman/decoder.h:11-26 -------------------------------------------------- */
#define CFFI_huffman_decoder_h__
typedef struct __huf_decoder huf_decoder_t;
huf_error_t huf_decoder_init(huf_decoder_t **self, const huf_config_t *config);
huf_error_t huf_decoder_free(huf_decoder_t **self);
/* Consume config->length compressed bytes from config->reader, write the original bytes. */
huf_error_t huf_decode(const huf_config_t *config);
#undef CFFI_huffman_decoder_h__

/* ---- include/huffman/bufio.h:13-93 ----------------------------------------------------- */
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

/* ---- include/huffman/histogram.h:10-49 ------------------------------------------------ */
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

/* ---- include/huffman/symbol.h:10-79 ---------------------------------------------------- */
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

/* ---- include/huffman/tree.h:10-82 ------------------------------------------------------ */
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
#endif /* INCLUDE_huffman_h__ */
