/*
 * huf_oracle.c - CPU restatement of libhuffman's block codec on flat buffers.
 *
 * TEST INFRASTRUCTURE ONLY (see huf_oracle.h). Parity status: PINNED against the reference's
 * own known-answer tests and against outputs of the unmodified reference built by
 * oracle/Makefile (tests/test_oracle.py, tests/golden/vectors.json).
 *
 * The restatement keeps the reference's algorithm, order of evaluation and wire format, but
 * uses flat arrays instead of heap nodes and flat memory instead of the callback streams.
 * All file:line citations are relative to the reference checkout.
 */
#include "huf_oracle.h"

#include <string.h>

/* ----------------------------------------------------------------------------------------
 * Histogram - src/histogram.c:73-103 (iota == 1 as configured by src/encoder.c:181).
 * -------------------------------------------------------------------------------------- */
void hufo_histogram(const uint8_t *buf, size_t len, uint64_t freq[HUFO_SLOTS], long *start)
{
    long first = *start;
    for (size_t i = 0; i < len; i++) {
        unsigned sym = buf[i];
        freq[sym] += 1;                       /* histogram.c:95 */
        if (first == -1 || (long)sym < first) /* histogram.c:97-99 */
            first = (long)sym;
    }
    *start = first;
}

/* ----------------------------------------------------------------------------------------
 * Tree construction - src/tree.c:292-427.
 *
 * Each round scans slots [start, node) in ascending order, skipping zero rates, and keeps
 * the two best entries with exactly the reference's comparison ladder (tree.c:337-351):
 *     first non-zero            -> slot 1
 *     rate <= rate1             -> old slot 1 moves to slot 2, new entry takes slot 1
 *     else !rate2 || rate<=rate2 -> new entry takes slot 2
 * which is "first two under ORDER BY rate ASC, index DESC".  Slot 1 becomes the left child,
 * slot 2 the right child of the new internal node `node`; a round that finds a single
 * entry hangs it under a left-only root and stops (tree.c:410-413).
 * -------------------------------------------------------------------------------------- */
void hufo_tree_from_histogram(uint64_t freq[HUFO_SLOTS], long start, hufo_tree_t *tree)
{
    memset(tree->left, 0xff, sizeof(tree->left));
    memset(tree->right, 0xff, sizeof(tree->right));
    memset(tree->parent, 0xff, sizeof(tree->parent));
    tree->root = -1;
    tree->nodes = 0;

    if (start < 0)
        return;                                /* nothing was populated */

    int node = HUFO_SYMBOLS;                   /* tree.c:303 */
    size_t first = (size_t)start;

    while (first < HUFO_SLOTS) {               /* tree.c:320 */
        int64_t rate1 = 0, rate2 = 0;
        int index1 = -1, index2 = -1;

        while (first < HUFO_SLOTS && !freq[first])   /* tree.c:326-328 (bounded here) */
            first++;

        for (size_t j = first; j < (size_t)node; j++) {   /* tree.c:331-352 */
            int64_t rate = (int64_t)freq[j];
            if (!rate)
                continue;
            if (!rate1) {
                rate1 = rate;
                index1 = (int)j;
            } else if (rate <= rate1) {
                rate2 = rate1;
                index2 = index1;
                rate1 = rate;
                index1 = (int)j;
            } else if (!rate2 || rate <= rate2) {
                rate2 = rate;
                index2 = (int)j;
            }
        }

        if (index1 == -1 && index2 == -1) {    /* tree.c:355-358 */
            tree->root = (int16_t)(node - 1);
            break;
        }

        tree->left[node] = -1;                 /* tree.c:390-404 */
        tree->right[node] = -1;
        if (index1 > -1) {
            tree->parent[index1] = (int16_t)node;
            tree->left[node] = (int16_t)index1;
            freq[index1] = 0;
        }
        if (index2 > -1) {
            tree->parent[index2] = (int16_t)node;
            tree->right[node] = (int16_t)index2;
            freq[index2] = 0;
        }
        freq[node] = (uint64_t)(rate1 + rate2);   /* tree.c:407 */
        node++;

        if (index1 > -1 && index2 == -1) {     /* tree.c:410-413 */
            tree->root = (int16_t)(node - 1);
            break;
        }
    }
    tree->nodes = node;
}

/* ----------------------------------------------------------------------------------------
 * Symbol codes - src/tree.c:12-47 (leaf -> root walk, '0' for a left child) and
 * src/encoder.c:40-81 (one element per leaf); the encoder emits the string back to front
 * (src/encoder.c:106-108), i.e. root -> leaf, which is the order stored in bits[].
 * -------------------------------------------------------------------------------------- */
void hufo_codes(const hufo_tree_t *tree, hufo_code_t codes[HUFO_SYMBOLS])
{
    uint8_t path[HUFO_SLOTS];
    for (int sym = 0; sym < HUFO_SYMBOLS; sym++) {
        codes[sym].length = 0;
        /* A byte is a leaf iff the builder gave it a parent (tree.c:381-387). */
        if (tree->parent[sym] < 0)
            continue;
        unsigned depth = 0;
        int cur = sym;
        while (tree->parent[cur] >= 0) {       /* tree.c:23-41 */
            int up = tree->parent[cur];
            path[depth++] = (tree->left[up] == cur) ? 0 : 1;
            cur = up;
        }
        codes[sym].length = (uint16_t)depth;
        for (unsigned b = 0; b < depth; b++)   /* encoder.c:106-108: reversed on emission */
            codes[sym].bits[b] = path[depth - 1 - b];
    }
}

/* ----------------------------------------------------------------------------------------
 * Preorder serialisation - src/tree.c:233-289: index, left subtree, right subtree; an
 * absent child is written as -1.
 * -------------------------------------------------------------------------------------- */
static size_t serialize_from(const hufo_tree_t *tree, int slot, int16_t *out)
{
    if (slot < 0) {                            /* tree.c:263-265 */
        out[0] = -1;
        return 1;
    }
    out[0] = (int16_t)slot;                    /* tree.c:245 */
    size_t used = 1;
    used += serialize_from(tree, tree->left[slot], out + used);
    used += serialize_from(tree, tree->right[slot], out + used);
    return used;
}

size_t hufo_tree_serialize(const hufo_tree_t *tree, int16_t *out)
{
    return serialize_from(tree, tree->root, out);
}

/* ----------------------------------------------------------------------------------------
 * Preorder de-serialisation - src/tree.c:138-227.  The reference recurses; an explicit
 * stack is used here so that 1025-deep chains cannot exhaust the C stack.  Semantics kept:
 *   remaining length < 1 -> NULL child, 0 entries consumed        (tree.c:152-160)
 *   entry == -1          -> NULL child, 1 entry consumed          (tree.c:165-171)
 *   anything else        -> a node; left subtree, then right subtree (tree.c:173-205)
 * -------------------------------------------------------------------------------------- */
size_t hufo_tree_deserialize(const int16_t *buf, size_t len, hufo_dtree_t *tree)
{
    /* pending[] holds nodes whose right child is still to be read. */
    int16_t pending[HUFO_DNODES];
    int top = 0;
    size_t pos = 0;
    int attach_to = -1;        /* node that receives the next subtree */
    int attach_right = 0;

    tree->n = 0;
    for (;;) {
        int made = -1;
        if (pos < len) {
            int16_t v = buf[pos++];
            if (v != -1) {
                made = tree->n++;
                tree->value[made] = v;
                tree->left[made] = -1;
                tree->right[made] = -1;
            }
        }
        if (made >= 0) {
            if (attach_to >= 0) {
                if (attach_right) tree->right[attach_to] = (int16_t)made;
                else              tree->left[attach_to] = (int16_t)made;
            }
            pending[top++] = (int16_t)made;    /* its right child comes after its left */
            attach_to = made;
            attach_right = 0;
            continue;
        }
        /* A NULL child closed the current position: resume at the innermost node that
         * still waits for its right subtree. */
        if (top == 0)
            break;
        attach_to = pending[--top];
        attach_right = 1;
    }
    return pos;
}

/* ----------------------------------------------------------------------------------------
 * Encoder - src/encoder.c:261-388.
 * -------------------------------------------------------------------------------------- */
size_t hufo_encode_bound(size_t n, size_t blocksize)
{
    if (n == 0)
        return 0;
    if (blocksize == 0)
        blocksize = n;
    size_t nblocks = (n + blocksize - 1) / blocksize;
    /* header 10 + 2*1025; payload: at most 9 bits per byte for encoder-built trees when
     * k <= 256 symbols share <= 2^64 counts is not a safe constant, so stay generous: the
     * deepest code an n-byte block can produce is bounded by 256 bits. Tests use small n. */
    size_t worst_bits = 9;  /* Huffman cost <= 8 bits/symbol, +1 for the wrap root */
    return nblocks * (10 + 2 * HUFO_TREE_MAX) + (n * worst_bits + 7) / 8 + nblocks;
}

/* One bit into the byte being assembled - src/bufio.c:18-23 (MSB first, offset 8 -> 0). */
typedef struct {
    uint8_t *out;
    size_t   cap;
    size_t   len;
    uint8_t  bits;
    uint8_t  offset;
} bitsink_t;

static int sink_byte(bitsink_t *s, uint8_t b)
{
    if (s->len >= s->cap)
        return HUFO_ERR_MEMORY;
    s->out[s->len++] = b;
    return HUFO_OK;
}

static int sink_bytes(bitsink_t *s, const void *p, size_t n)
{
    if (s->cap - s->len < n)
        return HUFO_ERR_MEMORY;
    memcpy(s->out + s->len, p, n);
    s->len += n;
    return HUFO_OK;
}

/* src/encoder.c:85-131 */
static int encode_block(bitsink_t *s, const hufo_code_t *codes, const uint8_t *buf, size_t len)
{
    s->bits = 0;                               /* bufio.c:27-32, encoder.c:345 */
    s->offset = 8;
    for (size_t pos = 0; pos < len; pos++) {
        const hufo_code_t *c = &codes[buf[pos]];
        for (unsigned b = 0; b < c->length; b++) {
            s->offset -= 1;                    /* bufio.c:21-22 */
            s->bits |= (uint8_t)((c->bits[b] & 1u) << s->offset);
            if (s->offset)
                continue;
            int err = sink_byte(s, s->bits);   /* encoder.c:114 */
            if (err)
                return err;
            s->bits = 0;
            s->offset = 8;
        }
    }
    if (s->offset != 8)                        /* encoder.c:123-128: zero padded tail */
        return sink_byte(s, s->bits);
    return HUFO_OK;
}

int hufo_encode(const uint8_t *in, size_t n, size_t blocksize,
                uint8_t *out, size_t cap, size_t *out_len, uint64_t *block_offsets)
{
    static const uint8_t le_probe[2] = {1, 0};
    uint16_t probe;
    memcpy(&probe, le_probe, 2);
    if (probe != 1)
        return HUFO_ERR_FATAL;                 /* wire format is little-endian LP64 only */

    bitsink_t sink = {out, cap, 0, 0, 8};
    hufo_code_t codes[HUFO_SYMBOLS];
    hufo_tree_t tree;
    uint64_t freq[HUFO_SLOTS];
    int16_t head[HUFO_TREE_MAX + 3];           /* the reference's 1024-entry array overflows at
                                                  k = 256 (encoder.c:270); sized correctly here */
    size_t nb = 0;

    if (blocksize == 0)                        /* encoder.c:163-165 */
        blocksize = n;

    size_t left = n;
    const uint8_t *src = in;
    while (left > 0) {                         /* encoder.c:288 */
        size_t take = blocksize < left ? blocksize : left;   /* encoder.c:289-293 */
        if (block_offsets)
            block_offsets[nb] = sink.len;
        nb++;

        memset(freq, 0, sizeof(freq));         /* encoder.c:360-373 resets between blocks */
        long start = -1;
        hufo_histogram(src, take, freq, &start);           /* encoder.c:301 */
        hufo_tree_from_histogram(freq, start, &tree);       /* encoder.c:306 */
        hufo_codes(&tree, codes);                            /* encoder.c:311 */
        size_t tree_len = hufo_tree_serialize(&tree, head); /* encoder.c:317 */

        uint64_t block_len = (uint64_t)take;   /* encoder.c:325, sizeof(size_t) == 8 */
        int16_t tl = (int16_t)tree_len;        /* encoder.c:272,331-332 */
        int err;
        if ((err = sink_bytes(&sink, &block_len, 8)) ||
            (err = sink_bytes(&sink, &tl, 2)) ||
            (err = sink_bytes(&sink, head, tree_len * 2)) ||   /* encoder.c:338-339 */
            (err = encode_block(&sink, codes, src, take)))      /* encoder.c:348 */
            return err;

        src += take;
        left -= take;
    }
    if (block_offsets)
        block_offsets[nb] = sink.len;
    *out_len = sink.len;
    return HUFO_OK;
}

/* ----------------------------------------------------------------------------------------
 * Decoder - src/decoder.c:201-287 with the block walk of src/decoder.c:34-96.
 * -------------------------------------------------------------------------------------- */
int hufo_decode(const uint8_t *in, size_t avail, uint64_t length,
                uint8_t *out, size_t cap, size_t *out_len, size_t *consumed,
                int max_tree_len)
{
    hufo_dtree_t tree;
    int16_t head[HUFO_TREE_MAX];
    size_t rd = 0;        /* reader position == have_been_processed (bufio.c:282-284) */
    size_t wr = 0;
    int err = HUFO_OK;

    while (length > rd) {                      /* decoder.c:218 */
        uint64_t block_len;
        int16_t tree_len;

        if (avail - rd < 8) { err = HUFO_ERR_READ_WRITE; break; }   /* decoder.c:220-224 */
        memcpy(&block_len, in + rd, 8);
        rd += 8;
        if (avail - rd < 2) { err = HUFO_ERR_READ_WRITE; break; }   /* decoder.c:231-234 */
        memcpy(&tree_len, in + rd, 2);
        rd += 2;
        if (tree_len < 0 || tree_len > max_tree_len) {              /* decoder.c:237-239 */
            err = HUFO_ERR_BTREE_OVERFLOW;
            break;
        }
        size_t tree_bytes = (size_t)tree_len * 2;
        if (avail - rd < tree_bytes) { err = HUFO_ERR_READ_WRITE; break; }   /* decoder.c:248-252 */
        memcpy(head, in + rd, tree_bytes);
        rd += tree_bytes;

        hufo_tree_deserialize(head, (size_t)tree_len, &tree);       /* decoder.c:255 */

        /* decoder.c:34-96 */
        uint64_t restored = 0;
        int node = 0;                          /* root */
        if (block_len > 0 && tree.n == 0) {    /* reference: NULL dereference; decision: err 6 */
            err = HUFO_ERR_BTREE_CORRUPTED;
            break;
        }
        while (restored < block_len && !err) {
            if (rd >= avail) { err = HUFO_ERR_READ_WRITE; break; }   /* decoder.c:53-56 */
            uint8_t byte = in[rd++];
            for (int bit = 7; bit >= 0; bit--) {                      /* decoder.c:58 */
                node = ((byte >> bit) & 1) ? tree.right[node] : tree.left[node];
                if (node < 0) {                /* decoder.c:69-71 */
                    err = HUFO_ERR_BTREE_CORRUPTED;
                    break;
                }
                if (tree.left[node] >= 0 || tree.right[node] >= 0)   /* decoder.c:74-76 */
                    continue;
                if (wr >= cap) { err = HUFO_ERR_MEMORY; break; }
                out[wr++] = (uint8_t)tree.value[node];               /* decoder.c:78 */
                restored++;
                node = 0;                      /* decoder.c:86 */
                if (restored >= block_len)     /* decoder.c:89-91: pad bits are dropped */
                    break;
            }
        }
        if (err)
            break;
    }
    *out_len = wr;
    if (consumed)
        *consumed = rd;
    return err;
}
