/*
 * hufgpu_common.h - constants and per-block records shared by the kernels and the host side
 * of the device API.  Internal header (not installed).
 */
#ifndef HUFGPU_COMMON_H
#define HUFGPU_COMMON_H

#include <stdint.h>

#define HUF_WAVE            64
#define HUF_NSYM            256
#define HUF_NSLOT           512          /* leaves 0..255 + internal nodes 256..511 (tree.h:16)  */
#define HUF_TREE_MAX        1025         /* 4k+1 entries for k = 256          */
#define HUF_TREE_STRICT     1024         /* decoder.c:237-239                                      */
#define HUF_TREE_STRIDE     1032         /* int16 slots reserved per block in the tree workspace   */
#define HUF_HEADER_FIXED    10           /* u64 block_len + i16 tree_len (encoder.c:325-332)       */

/* error numbering of include/huffman.h */
#define HUFE_OK        0
#define HUFE_MEMORY    1
#define HUFE_ARGUMENT  2
#define HUFE_RW        3
#define HUFE_FATAL     4
#define HUFE_OVERFLOW  5
#define HUFE_CORRUPTED 6

/* Written by the tree kernel, read by scan + pack. 16 bytes. */
struct HufBlockMeta {
    uint32_t tree_len;       /* int16 entries of the serialized tree                    */
    uint32_t max_len;        /* longest code in bits                                    */
    uint64_t payload_bits;   /* sum over symbols of hist * code length                  */
};

/* Written by decode_prepare, read by decode. 16 bytes. */
struct HufDecodeMeta {
    uint64_t block_len;      /* symbols to restore (0 when status != 0)                 */
    int16_t  tree_len;
    int16_t  leaf;           /* the byte of a [root, leaf, -1, -1, -1] tree, else -1    */
    int32_t  status;         /* HUFE_* found while parsing the header                   */
};

/* A code table entry: (code << 8) | length, code right-aligned, root->leaf, first bit = MSB
 * of the `length`-bit field. length == 0 => symbol absent. Codes are <= 56 bits for any
 * block shorter than F(57) = 3.6e11 bytes (a leaf at depth d needs F(d + 2) symbols; + the
 * wrap-root bit). */
typedef uint64_t hufcode_t;
#define HUF_CODE_MAXBITS 56
#define HUF_MAX_BLOCK_LEN ((uint64_t)1 << 38)   /* = HUFGPU_MAX_BLOCK (include/huffman_gpu.h) */

#endif
