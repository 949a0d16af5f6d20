/*
 * huf_oracle.h - CPU restatement of libhuffman's block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is compiled into, linked with or called
 * by the product (libhuffman_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (a) every known-answer vector the reference's own tests hold for the path
 *       (test/encode_test.c:35, test/decode_test.c:32-74, huffmanfile_test.py:8-18), and
 *   (b) outputs captured from the reference itself, compiled unmodified from
 *       /root/reference/src by oracle/Makefile (tests/golden/vectors.json, made by
 *       tools/make_goldens.py), and, when oracle/_ref/libhuffman_ref.so is present,
 *       live against that library on seeded random inputs.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).
 */
#ifndef HUF_ORACLE_H
#define HUF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error numbering of include/huffman/errors.h:6-27. */
enum {
    HUFO_OK = 0,
    HUFO_ERR_MEMORY = 1,
    HUFO_ERR_INVALID_ARGUMENT = 2,
    HUFO_ERR_READ_WRITE = 3,
    HUFO_ERR_FATAL = 4,
    HUFO_ERR_BTREE_OVERFLOW = 5,
    HUFO_ERR_BTREE_CORRUPTED = 6
};

#define HUFO_SYMBOLS     256   /* include/huffman/tree.h:10  HUF_ASCII_COUNT   */
#define HUFO_SLOTS       512   /* include/huffman/tree.h:16  HUF_HISTOGRAM_LEN */
#define HUFO_TREE_STRICT 1024  /* include/huffman/tree.h:13  HUF_BTREE_LEN     */
#define HUFO_TREE_MAX    1025  /* what the encoder really emits for k = 256    */

/* Flat form of the reference's pointer tree (include/huffman/tree.h:24-53). */
typedef struct {
    int16_t left[HUFO_SLOTS];    /* child slot or -1 */
    int16_t right[HUFO_SLOTS];
    int16_t parent[HUFO_SLOTS];
    int16_t root;                /* slot of the root, -1 when empty */
    int     nodes;               /* number of slots in use (by index) */
} hufo_tree_t;

/* One entry of the symbol -> code map (include/huffman/symbol.h:10-42). The reference
 * keeps an ASCII '0'/'1' string leaf->root; here bits[] is root->leaf, one bit per byte. */
typedef struct {
    uint16_t length;                 /* 0 = symbol absent */
    uint8_t  bits[HUFO_SLOTS];
} hufo_code_t;

/* src/histogram.c:73-103 with iota = 1 (src/encoder.c:181). freq has 512 slots; slots
 * 256..511 are the tree builder's scratch.  *start = smallest byte value seen, or -1. */
void hufo_histogram(const uint8_t *buf, size_t len, uint64_t freq[HUFO_SLOTS], long *start);

/* src/tree.c:292-427.  Consumes freq (rates are zeroed/overwritten like the reference). */
void hufo_tree_from_histogram(uint64_t freq[HUFO_SLOTS], long start, hufo_tree_t *tree);

/* src/tree.c:12-47 + src/encoder.c:40-81: codes for all 256 byte values. */
void hufo_codes(const hufo_tree_t *tree, hufo_code_t codes[HUFO_SYMBOLS]);

/* src/tree.c:233-289. Returns the number of int16 entries written (<= 1025). */
size_t hufo_tree_serialize(const hufo_tree_t *tree, int16_t *out);

/* Decoder-side tree: any shape the wire grammar allows (every entry != -1 is a node, so up
 * to 1025 nodes; src/tree.c:138-208). Node 0 is the root when n > 0. */
#define HUFO_DNODES 1026
typedef struct {
    int16_t left[HUFO_DNODES];   /* node number or -1 */
    int16_t right[HUFO_DNODES];
    int16_t value[HUFO_DNODES];  /* the int16 read from the wire (leaf: byte = (uint8_t)value) */
    int     n;                   /* nodes created; 0 => NULL root */
} hufo_dtree_t;

/* src/tree.c:138-227. Returns the number of entries consumed. */
size_t hufo_tree_deserialize(const int16_t *buf, size_t len, hufo_dtree_t *tree);

/* Upper bound of the encoded size of n bytes split in blocks of `blocksize`. */
size_t hufo_encode_bound(size_t n, size_t blocksize);

/* src/encoder.c:261-388 on flat buffers. blocksize 0 => one block of n bytes (:163-165).
 * If block_offsets != NULL it receives nblocks+1 byte offsets of the block headers. */
int hufo_encode(const uint8_t *in, size_t n, size_t blocksize,
                uint8_t *out, size_t cap, size_t *out_len,
                uint64_t *block_offsets);

/* src/decoder.c:201-287 on flat buffers.
 *   avail  = bytes the reader stream can deliver (memstream length),
 *   length = config.length, the compressed byte count that drives the block loop
 *            (decoder.c:218); the two differ in test/decode_test.c:51-60.
 * max_tree_len: 1024 reproduces the reference (strict); 1025 is the relaxed mode that
 * accepts the encoder's own k = 256 blocks.
 * *out_len = bytes produced before success or the first error; *consumed (optional) =
 * reader bytes successfully processed (bufio.c have_been_processed).
 * Deviation from the reference, by decision: tree_len == 0 with
 * block_len > 0 returns HUFO_ERR_BTREE_CORRUPTED where the reference dereferences NULL. */
int hufo_decode(const uint8_t *in, size_t avail, uint64_t length,
                uint8_t *out, size_t cap, size_t *out_len, size_t *consumed,
                int max_tree_len);

#ifdef __cplusplus
}
#endif
#endif /* HUF_ORACLE_H */
