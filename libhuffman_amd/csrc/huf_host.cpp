/*
 * huf_host.cpp - host half of the drop-in libhuffman API (include/huffman.h).
 *
 *  - stream backends, buffered byte I/O, error strings, config/alloc helpers: host plumbing
 *    with the reference's observable behaviour (src/io.c, src/bufio.c, src/errors.c,
 *    src/config.c, src/malloc.c) minus the defects listed in SURVEY Appendix D;
 *  - host-callable building blocks the reference also exports (histogram, symbol map, pointer
 *    tree): small re-implementations so that programs linking those symbols keep working;
 *  - huf_encode()/huf_decode(): read a batch through the caller's reader, run the block
 *    codec ON THE GPU (hufgpu_encode / hufgpu_decode_stream), hand the result to the writer.
 *    There is no CPU codec behind them: no GPU => HUF_ERROR_FATAL.
 *
 * Environment (huf_config_t's 48-byte layout is ABI, so switches live outside it):
 *   HUF_GPU_DEVICE        device ordinal (default 0)
 *   HUF_GPU_BATCH_MB      MiB per GPU round: input of huf_encode (default 256; 32 when a
 *                         huf_fdopen() descriptor is read or written under the GPU work), stream
 *                         bytes of a huf_decode that reads a huf_fdopen() descriptor (default 32)
 *   HUF_GPU_ZERO_COPY     0 = always go through the streams' read/write callbacks (default 1:
 *                         huf_memopen() streams are copied to/from the device directly and
 *                         huf_fdopen() descriptors are read/written by helper threads)
 *   HUF_GPU_RELAXED_TREE  1 = accept 1025-entry trees on decode
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sched.h>

#include <atomic>
#include <sys/mman.h>

#include <hip/hip_runtime_api.h>

#include "../../include/huffman.h"
#include "../../include/huffman_gpu.h"

#define GUARD(ptr)                                   \
    do {                                             \
        if (!(ptr)) return HUF_ERROR_INVALID_ARGUMENT; \
    } while (0)
#define TRY(expr)                                    \
    do {                                             \
        huf_error_t e__ = (huf_error_t)(expr);       \
        if (e__ != HUF_ERROR_SUCCESS) return e__;    \
    } while (0)

extern "C" {

/* ------------------------------------------------------------------ errors / alloc / config */
const char *huf_error_string(huf_error_t error)   /* src/errors.c:5-33 */
{
    /* callers pass anything (the reference answers "Unknown error" for -1, 7, 8 ...): the bytes are read as
     * an int, a C++ load of an out-of-range enum value would be undefined (UBSan: -fsanitize=enum) */
    int code;
    memcpy(&code, &error, sizeof(code));
    static_assert(sizeof(code) == sizeof(error), "huf_error_t is a 4-byte enum");
    switch (code) {
    case HUF_ERROR_SUCCESS: return "Success";
    case HUF_ERROR_MEMORY_ALLOCATION: return "Failed to allocate the requested memory block";
    case HUF_ERROR_INVALID_ARGUMENT: return "An invalid argument was specified to the function";
    case HUF_ERROR_READ_WRITE: return "Failed on read/write operation";
    case HUF_ERROR_FATAL: return "Fatal error";
    case HUF_ERROR_BTREE_OVERFLOW: return "Block is corrupted, Huffman tree has impossible size";
    case HUF_ERROR_BTREE_CORRUPTED: return "Huffman tree is corrupted and cannot be used to decode the block";
    default: return "Unknown error";
    }
}

huf_error_t huf_malloc(void **ptr, size_t size, size_t num)   /* src/malloc.c:7-19 */
{
    GUARD(ptr);
    *ptr = calloc(num, size);
    return *ptr ? HUF_ERROR_SUCCESS : HUF_ERROR_MEMORY_ALLOCATION;
}

huf_error_t huf_config_init(huf_config_t **self)   /* src/config.c:7-19 */
{
    GUARD(self);
    return huf_malloc((void **)self, sizeof(huf_config_t), 1);
}

huf_error_t huf_config_free(huf_config_t **self)   /* src/config.c:22-33 */
{
    GUARD(self);
    free(*self);
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ memory stream (src/io.c:66-226) */
typedef struct {
    void **buf;     /* caller-owned pointer, replaced on growth */
    size_t off;     /* read cursor */
    size_t len;
    size_t cap;
    void *wrapped;  /* huf_gpu_memwrap[_out](): the caller's bytes (buf points here); never freed */
    int readonly;   /* huf_gpu_memwrap(): never written either */
    int fixed;      /* huf_gpu_memwrap_out(): written up to cap, never grown */
} membuf_t;

/* A stream buffer: zeroed like the reference's calloc (src/io.c:79-104, :181), free()d by the caller
 * like the reference's.  From a few MiB on the kernel is asked to back it with huge pages: the
 * first write into such a buffer is bound by page faults, and a 2 MiB page is one fault instead of
 * 512 (transparent huge pages are in "madvise" mode on the GPU boxes). */
#define HUF_BIG_BUFFER ((size_t)4 << 20)
static void advise_huge_pages(void *p, size_t bytes)
{
    if (p && bytes >= HUF_BIG_BUFFER) {
        const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        const uintptr_t lo = ((uintptr_t)p + page - 1) & ~(page - 1), hi = ((uintptr_t)p + bytes) & ~(page - 1);
        if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);      /* advice only: failure is fine */
    }
}
static void *stream_alloc(size_t bytes)
{
    void *p = calloc(bytes ? bytes : 1, 1);
    advise_huge_pages(p, bytes);
    return p;
}

/* room for `count` more bytes behind the stream's contents */
static huf_error_t mem_reserve(membuf_t *m, size_t count)
{
    if (m->readonly) return HUF_ERROR_INVALID_ARGUMENT;
    if (m->fixed && m->cap - m->len < count) return HUF_ERROR_MEMORY_ALLOCATION;      /* the caller's memory ends here */
    if (m->cap - m->len < count) {
        /* growth policy of src/io.c:79-84 (double, or twice the request), but never smaller
         * than what is needed - the reference under-allocates here */
        size_t want = m->cap * 2;
        if (count > want) want = count * 2;
        if (want < m->len + count) want = m->len + count;
        void *grown = stream_alloc(want);
        if (!grown) return HUF_ERROR_MEMORY_ALLOCATION;
        if (m->len) memcpy(grown, *m->buf, m->len);
        free(*m->buf);
        *m->buf = grown;
        m->cap = want;
    }
    return HUF_ERROR_SUCCESS;
}

huf_error_t memwrite(void *stream, const void *buf, size_t count)
{
    membuf_t *m = (membuf_t *)stream;
    if (!m || (!buf && count)) return HUF_ERROR_INVALID_ARGUMENT;
    TRY(mem_reserve(m, count));
    if (count) memcpy((char *)*m->buf + m->len, buf, count);
    m->len += count;
    return HUF_ERROR_SUCCESS;
}

huf_error_t memread(void *stream, void *buf, size_t *count)
{
    membuf_t *m = (membuf_t *)stream;
    if (!m || !count) return HUF_ERROR_INVALID_ARGUMENT;
    size_t left = m->len - m->off;
    size_t take = *count < left ? *count : left;     /* short reads are not an error here */
    if (take) memcpy(buf, (char *)*m->buf + m->off, take);
    m->off += take;
    *count = take;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_memopen(huf_read_writer_t **self, void **buf, size_t capacity)
{
    GUARD(self);
    GUARD(buf);
    huf_read_writer_t *rw = (huf_read_writer_t *)calloc(1, sizeof(*rw));
    membuf_t *m = (membuf_t *)calloc(1, sizeof(*m));
    void *mem = stream_alloc(capacity);
    if (!rw || !m || !mem) {
        free(rw); free(m); free(mem);
        return HUF_ERROR_MEMORY_ALLOCATION;
    }
    *buf = mem;
    m->buf = buf;
    m->cap = capacity;
    rw->stream = m;
    rw->write = memwrite;
    rw->read = memread;
    *self = rw;
    return HUF_ERROR_SUCCESS;
}

/* Extension (not in the reference): a read-only memory stream over bytes the caller already has,
 * e.g. a Python bytes object - no copy into a huf_memopen() buffer.  Closed with huf_memclose(),
 * which never touches the bytes. */
int huf_gpu_memwrap(huf_read_writer_t **self, const void *data, size_t length)
{
    GUARD(self);
    if (!data && length) return HUF_ERROR_INVALID_ARGUMENT;
    huf_read_writer_t *rw = (huf_read_writer_t *)calloc(1, sizeof(*rw));
    membuf_t *m = (membuf_t *)calloc(1, sizeof(*m));
    if (!rw || !m) {
        free(rw); free(m);
        return HUF_ERROR_MEMORY_ALLOCATION;
    }
    m->wrapped = (void *)data;
    m->buf = &m->wrapped;
    m->len = m->cap = length;
    m->readonly = 1;
    rw->stream = m;
    rw->write = memwrite;
    rw->read = memread;
    *self = rw;
    return HUF_ERROR_SUCCESS;
}

/* Extension: a WRITER over memory the caller provides (`capacity` bytes, e.g. a Python bytes object that is
 * to become the result): what huf_encode()/huf_decode() write goes there directly, a write that does not fit
 * fails with HUF_ERROR_MEMORY_ALLOCATION (the memory is never grown, moved or freed).  huf_memlen() says how
 * much was written; closed with huf_memclose(). */
int huf_gpu_memwrap_out(huf_read_writer_t **self, void *buffer, size_t capacity)
{
    GUARD(self);
    if (!buffer && capacity) return HUF_ERROR_INVALID_ARGUMENT;
    huf_read_writer_t *rw = (huf_read_writer_t *)calloc(1, sizeof(*rw));
    membuf_t *m = (membuf_t *)calloc(1, sizeof(*m));
    if (!rw || !m) {
        free(rw); free(m);
        return HUF_ERROR_MEMORY_ALLOCATION;
    }
    m->wrapped = buffer;
    m->buf = &m->wrapped;
    m->cap = capacity;
    m->fixed = 1;
    advise_huge_pages(buffer, capacity);            /* (a fresh result buffer: its first write is bound by page faults) */
    rw->stream = m;
    rw->write = memwrite;
    rw->read = memread;
    *self = rw;
    return HUF_ERROR_SUCCESS;
}

static membuf_t *as_mem(const huf_read_writer_t *rw) { return rw ? (membuf_t *)rw->stream : NULL; }

huf_error_t huf_memlen(const huf_read_writer_t *self, size_t *len)
{
    GUARD(self); GUARD(len);
    *len = as_mem(self)->len;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_memcap(const huf_read_writer_t *self, size_t *cap)
{
    GUARD(self); GUARD(cap);
    *cap = as_mem(self)->cap;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_memrewind(huf_read_writer_t *self)   /* truncate, src/io.c:160-170 */
{
    GUARD(self);
    if (as_mem(self)->readonly) { as_mem(self)->off = 0; return HUF_ERROR_SUCCESS; }   /* wrapped bytes: start over */
    as_mem(self)->len = 0;
    as_mem(self)->off = 0;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_memclose(huf_read_writer_t **self)   /* leaves *buf to the caller, src/io.c:213-226 */
{
    GUARD(self);
    if (*self) {
        free((*self)->stream);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ fd stream (src/io.c:9-63) */
huf_error_t fdwrite(void *stream, const void *buf, size_t count)
{
    if (!stream) return HUF_ERROR_INVALID_ARGUMENT;
    const int fd = *(int *)stream;
    const char *p = (const char *)buf;
    while (count) {                      /* partial writes and EINTR are retried */
        ssize_t w = write(fd, p, count);
        if (w < 0) {
            if (errno == EINTR) continue;
            return HUF_ERROR_READ_WRITE;
        }
        p += w;
        count -= (size_t)w;
    }
    return HUF_ERROR_SUCCESS;
}

huf_error_t fdread(void *stream, void *buf, size_t *count)
{
    if (!stream || !count) return HUF_ERROR_INVALID_ARGUMENT;
    const int fd = *(int *)stream;
    size_t got = 0;
    while (got < *count) {               /* fill the request unless EOF comes first */
        ssize_t r = read(fd, (char *)buf + got, *count - got);
        if (r < 0) {
            if (errno == EINTR) continue;
            *count = got;
            return HUF_ERROR_READ_WRITE;
        }
        if (r == 0) break;
        got += (size_t)r;
    }
    *count = got;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_fdopen(huf_read_writer_t **self, int fd)
{
    GUARD(self);
    huf_read_writer_t *rw = (huf_read_writer_t *)calloc(1, sizeof(*rw));
    int *slot = (int *)malloc(sizeof(int));   /* the reference keeps the address of its own
                                                  parameter (src/io.c:45); a heap copy here */
    if (!rw || !slot) { free(rw); free(slot); return HUF_ERROR_MEMORY_ALLOCATION; }
    *slot = fd;
    rw->stream = slot;
    rw->read = fdread;
    rw->write = fdwrite;
    *self = rw;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_fdclose(huf_read_writer_t **self)
{
    GUARD(self);
    if (*self) {
        free((*self)->stream);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ bit writer (src/bufio.c:18-32) */
void huf_bit_write(huf_bit_read_writer_t *self, uint8_t bit)
{
    if (self->offset) self->offset--;
    self->bits |= (uint8_t)((bit & 1u) << self->offset);
}

void huf_bit_read_writer_reset(huf_bit_read_writer_t *self)
{
    self->bits = 0;
    self->offset = 8;
}

/* ------------------------------------------------------------------ buffered byte I/O (src/bufio.c:37-320) */
huf_error_t huf_bufio_read_writer_init(huf_bufio_read_writer_t **self, huf_read_writer_t *read_writer, size_t size)
{
    GUARD(self); GUARD(read_writer);
    huf_bufio_read_writer_t *b = (huf_bufio_read_writer_t *)calloc(1, sizeof(*b));
    if (!b) return HUF_ERROR_MEMORY_ALLOCATION;
    if (size) {                          /* 0 => pass-through (src/bufio.c:58-68) */
        b->bytes = (uint8_t *)calloc(size, 1);
        if (!b->bytes) { free(b); return HUF_ERROR_MEMORY_ALLOCATION; }
    }
    b->capacity = size;
    b->read_writer = read_writer;
    *self = b;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_bufio_read_writer_free(huf_bufio_read_writer_t **self)
{
    GUARD(self);
    if (*self) {
        free((*self)->bytes);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_bufio_read_writer_flush(huf_bufio_read_writer_t *self)
{
    GUARD(self);
    if (!self->length) return HUF_ERROR_SUCCESS;
    TRY(self->read_writer->write(self->read_writer->stream, self->bytes, self->length));
    self->length = 0;                    /* bytes were counted when they were accepted */
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_bufio_write(huf_bufio_read_writer_t *self, const void *buf, size_t size)
{
    GUARD(self); GUARD(buf);
    if (self->capacity && self->length >= self->capacity) TRY(huf_bufio_read_writer_flush(self));
    if (self->capacity && size <= self->capacity - self->length) {
        memcpy(self->bytes + self->length, buf, size);
        self->length += size;
        self->have_been_processed += size;
        return HUF_ERROR_SUCCESS;
    }
    if (size) {                          /* too big for the buffer: drain, then write through */
        TRY(huf_bufio_read_writer_flush(self));
        TRY(self->read_writer->write(self->read_writer->stream, buf, size));
        self->have_been_processed += size;
    }
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_bufio_read(huf_bufio_read_writer_t *self, void *buf, size_t size)
{
    GUARD(self); GUARD(buf);
    uint8_t *dst = (uint8_t *)buf;
    size_t want = size;
    size_t have = self->length - self->offset;
    if (have && want) {
        size_t take = have < want ? have : want;
        memcpy(dst, self->bytes + self->offset, take);
        self->offset += take;
        dst += take;
        want -= take;
    }
    if (want) {
        if (want >= self->capacity) {    /* straight into the destination (src/bufio.c:239-257) */
            size_t got = want;
            TRY(self->read_writer->read(self->read_writer->stream, dst, &got));
            self->length = self->offset = 0;
            if (got < want) return HUF_ERROR_READ_WRITE;
        } else {                         /* refill, then copy (src/bufio.c:259-277) */
            size_t got = self->capacity;
            TRY(self->read_writer->read(self->read_writer->stream, self->bytes, &got));
            self->length = got;
            self->offset = 0;
            if (got < want) return HUF_ERROR_READ_WRITE;
            memcpy(dst, self->bytes, want);
            self->offset = want;
        }
    }
    self->have_been_processed += size;   /* only successful requests are counted */
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_bufio_read_uint8(huf_bufio_read_writer_t *self, uint8_t *byte)
{
    GUARD(self); GUARD(byte);
    return huf_bufio_read(self, byte, 1);
}

huf_error_t huf_bufio_write_uint8(huf_bufio_read_writer_t *self, uint8_t byte)
{
    GUARD(self);
    return huf_bufio_write(self, &byte, 1);
}

/* ------------------------------------------------------------------ histogram (src/histogram.c) */
huf_error_t huf_histogram_init(huf_histogram_t **self, size_t iota, size_t length)
{
    GUARD(self);
    if (!iota || !length) return HUF_ERROR_INVALID_ARGUMENT;
    huf_histogram_t *h = (huf_histogram_t *)calloc(1, sizeof(*h));
    if (!h) return HUF_ERROR_MEMORY_ALLOCATION;
    h->frequencies = (uint64_t *)calloc(length, sizeof(uint64_t));
    if (!h->frequencies) { free(h); return HUF_ERROR_MEMORY_ALLOCATION; }
    h->iota = iota;
    h->length = length;
    h->start = (size_t)-1;
    *self = h;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_histogram_free(huf_histogram_t **self)
{
    GUARD(self);
    if (*self) {
        free((*self)->frequencies);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_histogram_reset(huf_histogram_t *self)
{
    GUARD(self);
    memset(self->frequencies, 0, self->length * sizeof(uint64_t));
    self->start = (size_t)-1;
    return HUF_ERROR_SUCCESS;
}

/* Generic element width (1..8 bytes, little-endian), whole elements only. The GPU kernel
 * hist256_kernel is the iota == 1 case the codec uses; this host version exists because the
 * reference exports it with a host-pointer signature. */
huf_error_t huf_histogram_populate(huf_histogram_t *self, void *buf, size_t len)
{
    GUARD(self); GUARD(buf);
    if (self->iota > 8) return HUF_ERROR_INVALID_ARGUMENT;
    const uint8_t *p = (const uint8_t *)buf;
    for (size_t at = 0; at + self->iota <= len; at += self->iota) {
        uint64_t el = 0;
        memcpy(&el, p + at, self->iota);
        if (el >= self->length) return HUF_ERROR_INVALID_ARGUMENT;   /* the reference writes out of bounds */
        self->frequencies[el]++;
        if (self->start == (size_t)-1 || el < self->start) self->start = (size_t)el;
    }
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ symbol map (src/symbol.c) */
huf_error_t huf_symbol_mapping_element_init(huf_symbol_mapping_element_t **self, const uint8_t *coding, size_t length)
{
    GUARD(self); GUARD(coding);
    huf_symbol_mapping_element_t *e = (huf_symbol_mapping_element_t *)calloc(1, sizeof(*e));
    if (!e) return HUF_ERROR_MEMORY_ALLOCATION;
    e->coding = (uint8_t *)calloc(length + 1, 1);
    if (!e->coding) { free(e); return HUF_ERROR_MEMORY_ALLOCATION; }
    memcpy(e->coding, coding, length);
    e->length = length;
    *self = e;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_element_free(huf_symbol_mapping_element_t **self)
{
    GUARD(self);
    if (*self) {
        free((*self)->coding);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_init(huf_symbol_mapping_t **self, size_t length)
{
    GUARD(self);
    huf_symbol_mapping_t *m = (huf_symbol_mapping_t *)calloc(1, sizeof(*m));
    if (!m) return HUF_ERROR_MEMORY_ALLOCATION;
    m->symbols = (huf_symbol_mapping_element_t **)calloc(length ? length : 1, sizeof(*m->symbols));
    if (!m->symbols) { free(m); return HUF_ERROR_MEMORY_ALLOCATION; }
    m->length = length;
    *self = m;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_reset(huf_symbol_mapping_t *self)
{
    GUARD(self);
    for (size_t i = 0; i < self->length; i++)
        if (self->symbols[i]) huf_symbol_mapping_element_free(&self->symbols[i]);
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_free(huf_symbol_mapping_t **self)
{
    GUARD(self);
    if (*self) {
        huf_symbol_mapping_reset(*self);
        free((*self)->symbols);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_insert(huf_symbol_mapping_t *self, size_t position, huf_symbol_mapping_element_t *element)
{
    GUARD(self); GUARD(element);
    if (position >= self->length) return HUF_ERROR_INVALID_ARGUMENT;
    if (self->symbols[position]) huf_symbol_mapping_element_free(&self->symbols[position]);
    self->symbols[position] = element;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_symbol_mapping_get(huf_symbol_mapping_t *self, size_t position, huf_symbol_mapping_element_t **element)
{
    GUARD(self); GUARD(element);
    if (position >= self->length) return HUF_ERROR_INVALID_ARGUMENT;
    *element = self->symbols[position];
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ pointer tree (src/tree.c) */
huf_error_t huf_node_to_string(const huf_node_t *self, uint8_t *buf, size_t *len)
{
    GUARD(buf); GUARD(len);
    size_t n = 0;
    for (const huf_node_t *cur = self; cur && cur->parent && n < *len; cur = cur->parent)
        buf[n++] = (cur->parent->left == cur) ? '0' : '1';    /* leaf -> root, src/tree.c:23-41 */
    *len = n;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_tree_init(huf_tree_t **self)
{
    GUARD(self);
    huf_tree_t *t = (huf_tree_t *)calloc(1, sizeof(*t));
    if (!t) return HUF_ERROR_MEMORY_ALLOCATION;
    t->leaves = (huf_node_t **)calloc(HUF_HISTOGRAM_LEN, sizeof(huf_node_t *));
    if (!t->leaves) { free(t); return HUF_ERROR_MEMORY_ALLOCATION; }
    *self = t;
    return HUF_ERROR_SUCCESS;
}

static void free_nodes(huf_node_t *root)   /* iterative: foreign trees may be 1025 deep */
{
    huf_node_t *cur = root;
    while (cur) {
        if (cur->left) { huf_node_t *c = cur->left; cur->left = NULL; c->parent = cur; cur = c; }
        else if (cur->right) { huf_node_t *c = cur->right; cur->right = NULL; c->parent = cur; cur = c; }
        else {
            huf_node_t *up = (cur == root) ? NULL : cur->parent;
            free(cur);
            cur = up;
        }
    }
}

huf_error_t huf_tree_reset(huf_tree_t *self)
{
    GUARD(self);
    free_nodes(self->root);
    self->root = NULL;
    memset(self->leaves, 0, HUF_HISTOGRAM_LEN * sizeof(huf_node_t *));
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_tree_free(huf_tree_t **self)
{
    GUARD(self);
    if (*self) {
        free_nodes((*self)->root);
        free((*self)->leaves);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

/* Same selection rule as the device tree_kernel: smallest (rate, 511 - index) first; the
 * smaller becomes the left child; a lone survivor gets a left-only root (src/tree.c:292-427).
 * Consumes the histogram like the reference does. */
huf_error_t huf_tree_from_histogram(huf_tree_t *self, huf_histogram_t *histogram)
{
    GUARD(self); GUARD(histogram);
    if (histogram->length < HUF_HISTOGRAM_LEN) return HUF_ERROR_INVALID_ARGUMENT;
    uint64_t *rate = histogram->frequencies;
    huf_node_t *slot[HUF_HISTOGRAM_LEN] = {0};
    int next = HUF_ASCII_COUNT;
    for (;;) {
        int best = -1, second = -1;
        for (int i = next - 1; i >= 0; i--) {          /* descending index: ties keep the earlier hit */
            if (!rate[i]) continue;
            if (best < 0 || rate[i] < rate[best]) { second = best; best = i; }
            else if (second < 0 || rate[i] < rate[second]) second = i;
        }
        if (best < 0) break;
        if (next >= HUF_HISTOGRAM_LEN) return HUF_ERROR_FATAL;
        huf_node_t *parent = (huf_node_t *)calloc(1, sizeof(huf_node_t));
        if (!parent) return HUF_ERROR_MEMORY_ALLOCATION;
        parent->index = (int16_t)next;
        const int pick[2] = {best, second};
        for (int side = 0; side < 2; side++) {
            const int i = pick[side];
            if (i < 0) continue;
            if (!slot[i]) {
                slot[i] = (huf_node_t *)calloc(1, sizeof(huf_node_t));
                if (!slot[i]) { free(parent); return HUF_ERROR_MEMORY_ALLOCATION; }
                slot[i]->index = (int16_t)i;
            }
            slot[i]->parent = parent;
            if (side == 0) parent->left = slot[i]; else parent->right = slot[i];
            if (i < HUF_ASCII_COUNT) self->leaves[i] = slot[i];
        }
        rate[next] = rate[best] + (second >= 0 ? rate[second] : 0);
        rate[best] = 0;
        if (second >= 0) rate[second] = 0;
        slot[next] = parent;
        self->root = parent;
        next++;
        if (second < 0) break;
    }
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_tree_serialize(huf_tree_t *self, int16_t *buf, size_t *len)   /* preorder, -1 = absent */
{
    GUARD(self); GUARD(buf); GUARD(len);
    size_t n = 0;
    /* explicit stack of "right children still to emit" */
    const huf_node_t *stack[2 * HUF_HISTOGRAM_LEN + 4];
    int top = 0;
    const huf_node_t *cur = self->root;
    for (;;) {
        if (cur) {
            buf[n++] = cur->index;
            if (top >= (int)(sizeof(stack) / sizeof(stack[0]))) return HUF_ERROR_FATAL;
            stack[top++] = cur->right;
            cur = cur->left;
        } else {
            buf[n++] = HUF_LEAF_NODE;
            if (!top) break;
            cur = stack[--top];
        }
    }
    *len = n;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_tree_deserialize(huf_tree_t *self, const int16_t *buf, size_t len)
{
    GUARD(self); GUARD(buf);
    /* every entry other than -1 is a node, entries past `len` are absent children */
    huf_node_t **pending = (huf_node_t **)calloc(len + 1, sizeof(huf_node_t *));
    if (!pending) return HUF_ERROR_MEMORY_ALLOCATION;
    size_t top = 0, at = 0;
    huf_node_t **link = &self->root;
    huf_node_t *owner = NULL;
    for (;;) {
        huf_node_t *made = NULL;
        if (at < len) {
            const int16_t v = buf[at++];
            if (v != HUF_LEAF_NODE) {
                made = (huf_node_t *)calloc(1, sizeof(huf_node_t));
                if (!made) { free(pending); return HUF_ERROR_MEMORY_ALLOCATION; }
                made->index = v;
                made->parent = owner;
            }
        }
        if (made) {
            *link = made;
            pending[top++] = made;
            owner = made;
            link = &made->left;
            continue;
        }
        if (!top) break;
        owner = pending[--top];
        link = &owner->right;
    }
    free(pending);
    return HUF_ERROR_SUCCESS;
}

/* ------------------------------------------------------------------ GPU sessions of huf_encode/huf_decode
 * A session = one device context plus its staging buffers; a call holds one session from start to
 * end.  By default there is ONE session on device HUF_GPU_DEVICE (0): concurrent calls take turns.
 * HUF_GPU_DEVICES = "0,1,2" / "all" makes one session per listed device ("0,0": two on device 0), and
 * concurrent calls - disjoint configs on different threads are legal and parallel in the
 * reference, which has no global state (src/encoder.c:379-392) - run side by side, each on the
 * first session that is free: a multi-threaded C caller uses every listed GPU. */
#define LANE_MAX 8             /* copy lanes (threads) of a large host <-> device transfer */
typedef struct {
    void *h_a, *h_b;           /* pinned staging */
    size_t h_a_cap, h_b_cap;
    void *d_a, *d_b, *d_c;     /* device staging (d_c: one round of a descriptor-fed decode) */
    size_t d_a_cap, d_b_cap, d_c_cap;
    void *lane_pin;            /* LANE_MAX x 2 pinned slots of LANE_SLOT bytes (lane_copy) */
    hipStream_t lane_stream[LANE_MAX];
    hipEvent_t lane_ev[LANE_MAX][2];
    int lanes_ready;
} staging_t;

#define HUF_MAX_SESSIONS 32
typedef struct {
    int device;
    int busy;
    hufgpu_ctx_t *ctx;
    staging_t stage;
} session_t;

static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_pool_cv = PTHREAD_COND_INITIALIZER;
static session_t g_sessions[HUF_MAX_SESSIONS];
static int g_nsessions = 0;
static int g_relaxed = -1;
static __thread session_t *t_session = NULL;       /* the session the calling thread holds */
#define g_ctx (t_session->ctx)
#define g_stage (t_session->stage)

/* the device list, read once (no GPU call: a process without a GPU still gets its loud error from
 * session_acquire) */
static void session_pool_init(void)
{
    if (g_nsessions) return;
    const char *list = getenv("HUF_GPU_DEVICES");
    if (list && *list) {
        if (strcmp(list, "all") == 0) {
            int n = hufgpu_device_count();
            if (n > HUF_MAX_SESSIONS) n = HUF_MAX_SESSIONS;
            for (int i = 0; i < n; i++) g_sessions[g_nsessions++].device = i;
        } else {
            const char *p = list;
            while (*p && g_nsessions < HUF_MAX_SESSIONS) {
                char *end = NULL;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                if (v >= 0) g_sessions[g_nsessions++].device = (int)v;
                p = end;
                while (*p == ',' || *p == ' ') p++;
            }
        }
    }
    if (!g_nsessions) {
        const char *dev = getenv("HUF_GPU_DEVICE");
        g_sessions[g_nsessions++].device = dev ? atoi(dev) : 0;
    }
}

static void session_enter(void)
{
    pthread_mutex_lock(&g_pool_lock);
    session_pool_init();
    for (;;) {
        for (int i = 0; i < g_nsessions; i++)
            if (!g_sessions[i].busy) {
                g_sessions[i].busy = 1;
                t_session = &g_sessions[i];
                pthread_mutex_unlock(&g_pool_lock);
                return;
            }
        pthread_cond_wait(&g_pool_cv, &g_pool_lock);
    }
}

static void session_leave(void)
{
    pthread_mutex_lock(&g_pool_lock);
    t_session->busy = 0;
    t_session = NULL;
    pthread_cond_signal(&g_pool_cv);
    pthread_mutex_unlock(&g_pool_lock);
}

/* a second, third ... session for the calling call, if one is free right now (never waits) */
static session_t *session_try_extra(void)
{
    session_t *got = NULL;
    pthread_mutex_lock(&g_pool_lock);
    for (int i = 0; i < g_nsessions && !got; i++)
        if (!g_sessions[i].busy) {
            g_sessions[i].busy = 1;
            got = &g_sessions[i];
        }
    pthread_mutex_unlock(&g_pool_lock);
    return got;
}

static void session_release_extra(session_t *s)
{
    pthread_mutex_lock(&g_pool_lock);
    s->busy = 0;
    pthread_cond_signal(&g_pool_cv);
    pthread_mutex_unlock(&g_pool_lock);
}

static huf_error_t session_acquire(void)
{
    if (g_ctx) return HUF_ERROR_SUCCESS;
    int rc = hufgpu_ctx_create(&g_ctx, t_session->device);
    if (rc != HUF_ERROR_SUCCESS) {
        fprintf(stderr, "libhuffman: the codec needs an MI355X (gfx950) GPU and has no CPU fallback: %s\n",
                hufgpu_last_error(NULL));
        g_ctx = NULL;
        return HUF_ERROR_FATAL;
    }
    return HUF_ERROR_SUCCESS;
}

static huf_error_t grow_host(void **p, size_t *cap, size_t want)
{
    if (*cap >= want) return HUF_ERROR_SUCCESS;
    if (*p) (void)hipHostFree(*p);
    *p = NULL; *cap = 0;
    (void)hipSetDevice(t_session->device);
    if (hipHostMalloc(p, want, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return HUF_ERROR_MEMORY_ALLOCATION;
    }
    *cap = want;
    return HUF_ERROR_SUCCESS;
}

static huf_error_t grow_dev(void **p, size_t *cap, size_t want)
{
    if (*cap >= want) return HUF_ERROR_SUCCESS;
    if (*p) hufgpu_free(g_ctx, *p);
    *p = NULL; *cap = 0;
    TRY(hufgpu_malloc(g_ctx, p, want));
    *cap = want;
    return HUF_ERROR_SUCCESS;
}

/* Device -> a memory stream's buffer.  The bytes behind a stream's contents are usually pages that
 * were never touched (a fresh buffer, the caller's huf_memopen capacity): copied into as they are,
 * the copy spends its time in page faults (240 MiB: 28-35 ms for a 4.5 ms copy).  So the pages are
 * populated first - madvise(MADV_POPULATE_WRITE), contents untouched, a few threads on disjoint
 * parts; with the huge pages stream_alloc asked for that is 2-3 ms - and copied into afterwards
 * (not at the same time as ANY copy of this process, in either direction: the copies' page pinning
 * and the populating threads then fight for the address-space lock - populating under the copy
 * itself 56 ms, under the input's copy to the device still slower than one after the other). */
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
#define PREFAULT_MIN ((size_t)16 << 20)
typedef struct { char *p; size_t n; } prefault_t;

static void *prefault_main(void *arg)
{
    prefault_t *w = (prefault_t *)arg;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (uintptr_t)w->p & ~(page - 1), hi = ((uintptr_t)w->p + w->n + page - 1) & ~(page - 1);
    if (madvise((void *)lo, (size_t)(hi - lo), MADV_POPULATE_WRITE) != 0)
        for (char *q = w->p; q < w->p + w->n; q += page) (void)__atomic_fetch_or(q, 0, __ATOMIC_RELAXED);   /* older kernels: a write fault per page, contents kept (one atomic read-modify-write) */
    return NULL;
}

static int prefault_threads(void)
{
    static int n = -1;
    if (n < 0) {
        const char *e = getenv("HUF_GPU_PREFAULT_THREADS");
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        n = e ? atoi(e) : (int)(cpus >= 4 ? 4 : cpus);
        if (n < 0) n = 0;
        if (n > 16) n = 16;
    }
    return n;
}

/* populate [p, p + n) on helper threads; prefault_end() waits for them */
typedef struct {
    prefault_t part[16];
    pthread_t th[16];
    int started;            /* bit i: th[i] runs */
} prefault_job_t;

static void prefault_begin(prefault_job_t *j, char *p, size_t n, int also_here)
{
    j->started = 0;
    const int nthreads = prefault_threads();
    if (n < PREFAULT_MIN || nthreads <= 0) return;
    const size_t piece = ((n / (size_t)nthreads) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);   /* whole huge pages */
    for (int i = 0; i < nthreads; i++) {
        const size_t off = (size_t)i * piece;
        if (off >= n) break;
        j->part[i].p = p + off;
        j->part[i].n = (n - off < piece || i == nthreads - 1) ? n - off : piece;     /* the last part takes the rest */
        if (i == 0 && also_here) continue;                          /* the first part is the calling thread's */
        if (pthread_create(&j->th[i], NULL, prefault_main, &j->part[i]) == 0) j->started |= 1 << i;
        else prefault_main(&j->part[i]);
    }
    if (also_here) prefault_main(&j->part[0]);
}

static void prefault_end(prefault_job_t *j)
{
    for (int i = 0; i < 16; i++)
        if (j->started & (1 << i)) pthread_join(j->th[i], NULL);
    j->started = 0;
}

/* Large transfers between PAGEABLE host memory (a caller's buffer, a memory stream) and the device.
 * hipMemcpy from or to pageable memory runs at 10-18 GB/s here (the runtime pins or stages piece by piece on one
 * thread), a fifth of what the link carries.  lane_copy() cuts the transfer into pieces of LANE_SLOT bytes and gives
 * them to a few threads (lanes); every lane owns two pinned slots and a stream: memcpy into a slot, asynchronous copy
 * from it - while that runs, memcpy into the other slot (and the other way round for device -> host, where the
 * destination's pages are populated piece by piece by the lane that is about to fill them: no populate of the whole
 * buffer in front of the copy).  Returns when everything has arrived. */
#define LANE_SLOT ((size_t)8 << 20)
#define LANE_MIN ((size_t)32 << 20)        /* below this one hipMemcpy is as good */
typedef struct {
    staging_t *st;
    int device, lane, nlanes, to_device;
    char *host;
    char *dev;
    size_t n;
    int err;
} lane_job_t;

static int lane_count(void)
{
    static int n = -1;
    if (n < 0) {
        const char *e = getenv("HUF_GPU_COPY_LANES");
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        n = e ? atoi(e) : (int)(cpus >= 16 ? 6 : (cpus >= 8 ? 4 : (cpus >= 4 ? 2 : 1)));     /* (1 GiB of log text through huffmanfile: 0 lanes 5.2, 2: 5.3, 4: 5.7, 6: 6.0, 8: 4.7 GiB/s) */
        if (n < 0) n = 0;
        if (n > LANE_MAX) n = LANE_MAX;
    }
    return n;
}

static void *lane_main(void *arg)
{
    lane_job_t *j = (lane_job_t *)arg;
    staging_t *st = j->st;
    if (hipSetDevice(j->device) != hipSuccess) { j->err = 1; return NULL; }
    hipStream_t s = st->lane_stream[j->lane];
    char *slot[2] = {(char *)st->lane_pin + (size_t)(2 * j->lane) * LANE_SLOT, (char *)st->lane_pin + (size_t)(2 * j->lane + 1) * LANE_SLOT};
    const size_t pieces = (j->n + LANE_SLOT - 1) / LANE_SLOT;
    int k = 0;                                   /* this lane's pieces, in order: lane, lane + nlanes, ... */
    if (j->to_device) {
        for (size_t p = (size_t)j->lane; p < pieces; p += (size_t)j->nlanes, k++) {
            const size_t off = p * LANE_SLOT, len = (j->n - off < LANE_SLOT) ? j->n - off : LANE_SLOT;
            if (k >= 2 && hipEventSynchronize(st->lane_ev[j->lane][k & 1]) != hipSuccess) { j->err = 1; break; }
            memcpy(slot[k & 1], j->host + off, len);
            if (hipMemcpyAsync(j->dev + off, slot[k & 1], len, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipEventRecord(st->lane_ev[j->lane][k & 1], s) != hipSuccess) { j->err = 1; break; }
        }
    } else {
        /* device -> host: the copy of piece k + 1 is in flight while piece k goes from its slot to the destination */
        size_t p = (size_t)j->lane;
        size_t off = p * LANE_SLOT, len = 0;
        if (p < pieces) {
            len = (j->n - off < LANE_SLOT) ? j->n - off : LANE_SLOT;
            if (hipMemcpyAsync(slot[0], j->dev + off, len, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipEventRecord(st->lane_ev[j->lane][0], s) != hipSuccess) j->err = 1;
        }
        for (; p < pieces && !j->err; p += (size_t)j->nlanes, k++) {
            const size_t pn = p + (size_t)j->nlanes;
            size_t offn = 0, lenn = 0;
            if (pn < pieces) {
                offn = pn * LANE_SLOT;
                lenn = (j->n - offn < LANE_SLOT) ? j->n - offn : LANE_SLOT;
                if (hipMemcpyAsync(slot[(k + 1) & 1], j->dev + offn, lenn, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipEventRecord(st->lane_ev[j->lane][(k + 1) & 1], s) != hipSuccess) { j->err = 1; break; }
            }
            prefault_t w = {j->host + off, len};
            prefault_main(&w);                       /* (contents untouched; pages that are there already cost nothing) */
            if (hipEventSynchronize(st->lane_ev[j->lane][k & 1]) != hipSuccess) { j->err = 1; break; }
            memcpy(j->host + off, slot[k & 1], len);
            off = offn;
            len = lenn;
        }
    }
    if (hipStreamSynchronize(s) != hipSuccess) j->err = 1;
    return NULL;
}

/* host <-> device, n bytes; falls back to one plain copy for small transfers or when the lanes cannot be set up */
static huf_error_t lane_copy(int to_device, void *dev, void *host, size_t n)
{
    staging_t *st = &g_stage;
    const int nl = lane_count();
    if (n < LANE_MIN || nl <= 0) goto plain;
    (void)hipSetDevice(t_session->device);
    if (st->lanes_ready < 0) goto plain;             /* a set-up that failed once: plain copies from then on */
    if (st->lanes_ready == 0) {
        st->lanes_ready = -1;
        int made_streams = 0, made_events = 0, failed = 0;
        if (hipHostMalloc(&st->lane_pin, (size_t)2 * (size_t)nl * LANE_SLOT, hipHostMallocPortable) != hipSuccess) { st->lane_pin = NULL; failed = 1; }
        for (int i = 0; i < nl && !failed; i++) {
            if (hipStreamCreateWithFlags(&st->lane_stream[i], hipStreamNonBlocking) != hipSuccess) { failed = 1; break; }
            made_streams++;
            for (int e = 0; e < 2; e++) {
                if (hipEventCreateWithFlags(&st->lane_ev[i][e], hipEventDisableTiming) != hipSuccess) { failed = 1; break; }
                made_events++;
            }
        }
        if (failed) {                                /* give back what was made: nothing of it is looked at again */
            (void)hipGetLastError();
            for (int k = 0; k < made_events; k++) (void)hipEventDestroy(st->lane_ev[k / 2][k % 2]);
            for (int i = 0; i < made_streams; i++) (void)hipStreamDestroy(st->lane_stream[i]);
            if (st->lane_pin) (void)hipHostFree(st->lane_pin);
            st->lane_pin = NULL;
            goto plain;
        }
        st->lanes_ready = 1;
    }
    {
        (void)hipDeviceSynchronize();                /* what the device buffer is read from or written by has finished (the lanes' streams do not wait for others) */
        lane_job_t job[LANE_MAX];
        pthread_t th[LANE_MAX];
        int started = 0, bad = 0;
        if (!to_device) {
            const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
            const uintptr_t lo = ((uintptr_t)host + page - 1) & ~(page - 1), hi = ((uintptr_t)host + n) & ~(page - 1);
            if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
        }
        for (int i = 0; i < nl; i++) {
            job[i].st = st; job[i].device = t_session->device; job[i].lane = i; job[i].nlanes = nl; job[i].to_device = to_device;
            job[i].host = (char *)host; job[i].dev = (char *)dev; job[i].n = n; job[i].err = 0;
            if (i == 0) continue;                        /* lane 0 is the calling thread's */
            if (pthread_create(&th[i], NULL, lane_main, &job[i]) == 0) started |= 1 << i;
            else lane_main(&job[i]);
        }
        lane_main(&job[0]);
        for (int i = 1; i < nl; i++)
            if (started & (1 << i)) pthread_join(th[i], NULL);
        for (int i = 0; i < nl; i++) bad |= job[i].err;
        if (bad) { (void)hipGetLastError(); return HUF_ERROR_FATAL; }
        return HUF_ERROR_SUCCESS;
    }
plain:
    return (huf_error_t)(to_device ? hufgpu_memcpy_h2d(g_ctx, dev, host, n) : hufgpu_memcpy_d2h(g_ctx, host, dev, n));
}

/* ------------------------------------------------------------------ transfers in both directions at once
 * huf_encode / huf_decode between two memory streams (src/encoder.c:261-388, src/decoder.c:201-287 with the
 * reference's memory streams on both ends) move N bytes to the device and about as many back; round 4 did one after
 * the other - copy in, kernels, copy out, per round of 256 MiB - and reached 11 GiB/s over a link that carries 53
 * each way AT THE SAME TIME.  Here a session owns two sets of persistent copy threads ("lanes"), one per direction;
 * a lane has two pinned slots, a stream and its events.  The caller publishes SEGMENTS - (host address, device
 * address, bytes) - per direction; the lanes of that direction take the segments in order and share the pieces of
 * each (DX_SLOT bytes, dealt out round robin).
 *   host -> device: memcpy into a slot, asynchronous copy from it, the next piece into the other slot meanwhile.
 *     A lane reports a segment as ISSUED - its copies are on the lane's stream, an event behind them - and the
 *     caller makes the compute stream wait for those events: no host thread waits for a copy to arrive.
 *   device -> host: asynchronous copy into a slot, and while it runs the piece before it goes from the other slot
 *     to its destination, whose pages the lane populates first (d2h_to_memstream's comment says why not earlier).
 *     A lane reports a segment as DONE when its pieces are in place.
 * Rounds of HUF_GPU_ROUND_MB (32) then overlap as: copy-in of round i + 1 | kernels of round i | copy-out of round
 * i - 1, inside ONE session (two sessions on one GPU lose: profiles/r04/python_layer_sessions.txt). */
#define DX_LANES_MAX 8
#define DX_SLOT ((size_t)4 << 20)
#define DX_RING 16                       /* segments a direction may have in flight */
typedef struct {
    char *host, *dev;
    size_t n;
    int direct;                          /* host -> device: the host bytes lie in registered (pinned) pages - copied from where they are */
    int issued_left;                     /* lanes that have not yet put their pieces on their streams (host -> device) */
    int done_left;                       /* lanes that have not yet finished their pieces */
} dx_seg_t;
struct dx_pool;
typedef struct {
    struct dx_pool *pool;
    int dir, idx;
    pthread_t th;
    hipStream_t stream;
    hipEvent_t slot_ev[2];
    hipEvent_t seg_ev[DX_RING];          /* host -> device: behind the lane's last copy of a segment */
    char *slot[2];
} dx_lane_t;
typedef struct dx_pool {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int device;
    int ready;                           /* 0 not made, 1 usable, -1 could not be made */
    int err;                             /* a lane met a failing HIP call (sticky for the pool's life) */
    int nl[2];                           /* lanes per direction: [0] host -> device, [1] device -> host */
    int can_register;                    /* hipHostRegister works on this process's pageable memory (dx_register_input) */
    dx_lane_t lane[2][DX_LANES_MAX];
    dx_seg_t seg[2][DX_RING];
    uint64_t published[2];               /* segments ever published per direction (a segment's id is its number) */
    void *pin;
} dx_pool_t;

static int dx_lanes_per_dir(void)
{
    static int n = -1;
    if (n < 0) {
        const char *e = getenv("HUF_GPU_DUPLEX_LANES");
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        n = e ? atoi(e) : (int)(cpus >= 16 ? 5 : (cpus >= 8 ? 3 : (cpus >= 4 ? 2 : 1)));
        if (n < 0) n = 0;
        if (n > DX_LANES_MAX) n = DX_LANES_MAX;
    }
    return n;
}

static void dx_fail(dx_pool_t *P) { pthread_mutex_lock(&P->mu); P->err = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); (void)hipGetLastError(); }

static void *dx_lane_main(void *arg)
{
    dx_lane_t *L = (dx_lane_t *)arg;
    dx_pool_t *P = L->pool;
    const int dir = L->dir, nl = P->nl[dir];
    if (hipSetDevice(P->device) != hipSuccess) dx_fail(P);
    uint64_t cur = 0;                    /* the next segment of this direction this lane looks at */
    unsigned k = 0;                      /* pieces this lane has moved: k & 1 is the slot of the next */
    for (;;) {
        pthread_mutex_lock(&P->mu);
        while (cur >= P->published[dir]) pthread_cond_wait(&P->cv, &P->mu);
        const dx_seg_t sg = P->seg[dir][cur % DX_RING];
        pthread_mutex_unlock(&P->mu);
        const size_t pieces = (sg.n + DX_SLOT - 1) / DX_SLOT;
        /* piece q of segment `cur` is lane (q + cur) % nl's: short segments do not all start at lane 0 */
        size_t q = (size_t)(((uint64_t)L->idx + (uint64_t)nl - cur % (uint64_t)nl) % (uint64_t)nl);
        int bad = 0;
        if (dir == 0 && sg.direct) {
            /* registered pages: one asynchronous copy of the whole segment, by the lane whose turn it is */
            if (q == 0 && hipMemcpyAsync(sg.dev, sg.host, sg.n, hipMemcpyHostToDevice, L->stream) != hipSuccess) bad = 1;
            if (hipEventRecord(L->seg_ev[cur % DX_RING], L->stream) != hipSuccess) bad = 1;
        } else if (dir == 0) {
            for (; q < pieces && !bad; q += (size_t)nl, k++) {
                const size_t off = q * DX_SLOT, len = (sg.n - off < DX_SLOT) ? sg.n - off : DX_SLOT;
                if (k >= 2 && hipEventSynchronize(L->slot_ev[k & 1]) != hipSuccess) { bad = 1; break; }
                memcpy(L->slot[k & 1], sg.host + off, len);
                if (hipMemcpyAsync(sg.dev + off, L->slot[k & 1], len, hipMemcpyHostToDevice, L->stream) != hipSuccess ||
                    hipEventRecord(L->slot_ev[k & 1], L->stream) != hipSuccess) bad = 1;
            }
            if (hipEventRecord(L->seg_ev[cur % DX_RING], L->stream) != hipSuccess) bad = 1;
        } else {
            /* the copy of piece q + nl is in flight while piece q goes from its slot to the destination */
            size_t off = q * DX_SLOT, len = 0;
            if (q < pieces) {
                len = (sg.n - off < DX_SLOT) ? sg.n - off : DX_SLOT;
                if (hipMemcpyAsync(L->slot[k & 1], sg.dev + off, len, hipMemcpyDeviceToHost, L->stream) != hipSuccess ||
                    hipEventRecord(L->slot_ev[k & 1], L->stream) != hipSuccess) bad = 1;
            }
            for (; q < pieces && !bad; q += (size_t)nl, k++) {
                const size_t qn = q + (size_t)nl;
                size_t offn = 0, lenn = 0;
                if (qn < pieces) {
                    offn = qn * DX_SLOT;
                    lenn = (sg.n - offn < DX_SLOT) ? sg.n - offn : DX_SLOT;
                    if (hipMemcpyAsync(L->slot[(k + 1) & 1], sg.dev + offn, lenn, hipMemcpyDeviceToHost, L->stream) != hipSuccess ||
                        hipEventRecord(L->slot_ev[(k + 1) & 1], L->stream) != hipSuccess) { bad = 1; break; }
                }
                prefault_t w = {sg.host + off, len};
                prefault_main(&w);                   /* (contents untouched; pages that are there already cost nothing) */
                if (hipEventSynchronize(L->slot_ev[k & 1]) != hipSuccess) { bad = 1; break; }
                memcpy(sg.host + off, L->slot[k & 1], len);
                off = offn;
                len = lenn;
            }
        }
        pthread_mutex_lock(&P->mu);
        if (bad) { P->err = 1; (void)hipGetLastError(); }
        dx_seg_t *g = &P->seg[dir][cur % DX_RING];
        g->issued_left--;
        g->done_left--;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
        cur++;
    }
    return NULL;
}

/* the session's pool, made on first use; NULL when it cannot be made (the callers then move bytes the old way) */
static dx_pool_t *dx_get(void)
{
    static dx_pool_t pools[HUF_MAX_SESSIONS];
    dx_pool_t *P = &pools[t_session - g_sessions];
    if (P->ready > 0) return P->err ? NULL : P;
    if (P->ready < 0) return NULL;
    const int nl = dx_lanes_per_dir();
    P->ready = -1;
    if (nl <= 0) return NULL;
    (void)hipSetDevice(t_session->device);
    P->device = t_session->device;
    P->nl[0] = P->nl[1] = nl;
    {   /* can this process register pageable memory at all?  (HUF_GPU_REGISTER=0: never tried) */
        const char *e = getenv("HUF_GPU_REGISTER");
        void *probe = NULL;
        if (!(e && atoi(e) == 0) && posix_memalign(&probe, 4096, 1 << 16) == 0) {
            memset(probe, 1, 1 << 16);
            if (hipHostRegister(probe, 1 << 16, hipHostRegisterDefault) == hipSuccess) {
                (void)hipHostUnregister(probe);
                P->can_register = 1;
                P->nl[1] = nl + 3 < DX_LANES_MAX ? nl + 3 : DX_LANES_MAX;     /* (threads the other direction will rarely need) */
            } else (void)hipGetLastError();
            free(probe);
        }
    }
    pthread_mutex_init(&P->mu, NULL);
    pthread_cond_init(&P->cv, NULL);
    if (hipHostMalloc(&P->pin, (size_t)2 * (size_t)(P->nl[0] + P->nl[1]) * DX_SLOT, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); P->pin = NULL; return NULL; }
    int ok = 1, slots = 0;
    for (int d = 0; d < 2 && ok; d++)
        for (int i = 0; i < P->nl[d] && ok; i++, slots += 2) {
            dx_lane_t *L = &P->lane[d][i];
            L->pool = P; L->dir = d; L->idx = i;
            L->slot[0] = (char *)P->pin + (size_t)slots * DX_SLOT;
            L->slot[1] = L->slot[0] + DX_SLOT;
            if (hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking) != hipSuccess) ok = 0;
            for (int e = 0; e < 2 && ok; e++) if (hipEventCreateWithFlags(&L->slot_ev[e], hipEventDisableTiming) != hipSuccess) ok = 0;
            for (int e = 0; e < DX_RING && ok; e++) if (hipEventCreateWithFlags(&L->seg_ev[e], hipEventDisableTiming) != hipSuccess) ok = 0;
        }
    if (!ok) { (void)hipGetLastError(); return NULL; }        /* (what was made stays: a process makes at most one pool a session) */
    for (int d = 0; d < 2; d++)
        for (int i = 0; i < P->nl[d]; i++) {
            dx_lane_t *L = &P->lane[d][i];
            if (pthread_create(&L->th, NULL, dx_lane_main, L) != 0) return NULL;     /* (lanes that run wait for ever for work: harmless) */
            pthread_detach(L->th);
        }
    P->ready = 1;
    return P;
}

/* a segment for the lanes of direction `dir`; returns its id.  Waits while the direction's ring is full. */
static uint64_t dx_publish(dx_pool_t *P, int dir, void *host, void *dev, size_t n, int direct = 0)
{
    pthread_mutex_lock(&P->mu);
    const uint64_t id = P->published[dir];
    if (id >= DX_RING)
        while (P->seg[dir][id % DX_RING].done_left > 0 && !P->err) pthread_cond_wait(&P->cv, &P->mu);      /* the segment a ring ago */
    dx_seg_t *g = &P->seg[dir][id % DX_RING];
    g->host = (char *)host; g->dev = (char *)dev; g->n = n; g->direct = direct;
    g->issued_left = g->done_left = P->nl[dir];
    P->published[dir] = id + 1;
    pthread_cond_broadcast(&P->cv);
    pthread_mutex_unlock(&P->mu);
    return id;
}
/* host -> device segment `id`: every lane has its copies on its stream; the default stream (the kernels') waits for them */
static huf_error_t dx_wait_issued(dx_pool_t *P, uint64_t id)
{
    pthread_mutex_lock(&P->mu);
    while (P->seg[0][id % DX_RING].issued_left > 0 && !P->err) pthread_cond_wait(&P->cv, &P->mu);
    const int err = P->err;
    pthread_mutex_unlock(&P->mu);
    if (err) return HUF_ERROR_FATAL;
    for (int i = 0; i < P->nl[0]; i++)
        if (hipStreamWaitEvent((hipStream_t)0, P->lane[0][i].seg_ev[id % DX_RING], 0) != hipSuccess) { (void)hipGetLastError(); return HUF_ERROR_FATAL; }
    return HUF_ERROR_SUCCESS;
}
static huf_error_t dx_wait_done(dx_pool_t *P, int dir, uint64_t id)
{
    pthread_mutex_lock(&P->mu);
    while (P->seg[dir][id % DX_RING].done_left > 0 && !P->err) pthread_cond_wait(&P->cv, &P->mu);
    const int err = P->err;
    pthread_mutex_unlock(&P->mu);
    return err ? HUF_ERROR_FATAL : HUF_ERROR_SUCCESS;
}
/* everything published so far has been moved (a failed pool: the lanes still count their segments down) */
static void dx_drain(dx_pool_t *P)
{
    pthread_mutex_lock(&P->mu);
    for (int d = 0; d < 2; d++) {
        const uint64_t n = P->published[d];
        for (uint64_t id = n > DX_RING ? n - DX_RING : 0; id < n; id++)
            while (P->seg[d][id % DX_RING].done_left > 0) pthread_cond_wait(&P->cv, &P->mu);
    }
    pthread_mutex_unlock(&P->mu);
}

/* What a call reads from host memory has been written by somebody: its pages are there, and registering pages that are
 * there costs 2 ms per GiB on these boxes (tools/calib/host_link_probe.hip; pages never touched: 45 ms, the faults).  The
 * whole input is registered ONCE, before the first lane moves - a hipHostRegister beside the output lanes' page
 * populating brings both to a crawl (the address-space lock: 29 GiB/s where 50 were measured alone, and two threads
 * that register at once get a fifth of one thread's rate) - and the copies then run straight from the caller's pages:
 * no memcpy into a slot, 2 GiB of memory traffic per GiB and five busy threads less.  Returns the registered base (to
 * hand to dx_unregister_input) or NULL: a read-only mapping, pages somebody else has registered - the staged lanes
 * take the call then. */
static void *dx_register_input(dx_pool_t *P, const void *host, size_t n)
{
    if (!P->can_register || !n) return NULL;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (uintptr_t)host & ~(page - 1), hi = ((uintptr_t)host + n + page - 1) & ~(page - 1);   /* (the pages that hold its first and last byte are mapped) */
    if (hipHostRegister((void *)lo, (size_t)(hi - lo), hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    return (void *)lo;
}
static void dx_unregister_input(void *base)
{
    if (base && hipHostUnregister(base) != hipSuccess) (void)hipGetLastError();
}

/* bytes a round: HUF_GPU_ROUND_MB, else an eighth of the call between 8 and 32 MiB (a call of 64 MiB in eight rounds
 * still overlaps seven of them; a round much below 8 MiB is a piece or two for ten lanes) */
static uint64_t dx_round_bytes(uint64_t total, uint64_t blocksize)
{
    static long env = -1;
    if (env < 0) { const char *e = getenv("HUF_GPU_ROUND_MB"); env = (e && atoi(e) > 0) ? atoi(e) : 0; }
    if (env > 0) return (uint64_t)env << 20;
    /* (a round costs about 0.1 ms beside its transfers - its launches and the wait for its length; 64 MiB in rounds of 8 MiB:
     *  3.5 + 3.7 ms, of 16-24 MiB: 2.8-2.9 + 3.2) */
    uint64_t r = (total / 4) & ~(((uint64_t)1 << 20) - 1);
    if (r < ((uint64_t)16 << 20)) r = (uint64_t)16 << 20;
    if (r > ((uint64_t)32 << 20)) r = (uint64_t)32 << 20;
    /* A block is one workgroup's work up to 2 MiB (encode) / 4 MiB (decode): a round of 32 MiB in blocks of 1 MiB - the Python
     * layer's default - is 32 workgroups on 256 CUs, and a round then takes as long as ONE block does (1 GiB of log text: 16-20 ms
     * of an encode's 30 and 25 ms of a decode's 42 were that).  Rounds of at least 128 blocks, 256 MiB at most. */
    if (blocksize > ((uint64_t)128 << 10) && blocksize < ((uint64_t)4 << 20)) {
        uint64_t want = 128 * blocksize;
        if (want > ((uint64_t)256 << 20)) want = (uint64_t)256 << 20;
        if (want > r) r = want;
    }
    return r;
}
#define DX_MIN_BYTES ((uint64_t)32 << 20)      /* below this the rounds are too few to overlap anything */

static huf_error_t d2h_to_memstream(membuf_t *wmem, const void *d_src, size_t n)
{
    TRY(mem_reserve(wmem, n));
    char *dst = (char *)*wmem->buf + wmem->len;
    if (n >= LANE_MIN && lane_count() > 0) {
        TRY(lane_copy(0, (void *)d_src, dst, n));       /* (populates the pages piece by piece, beside the copies) */
    } else {
        prefault_job_t job;
        prefault_begin(&job, dst, n, 1);
        prefault_end(&job);
        TRY(hufgpu_memcpy_d2h(g_ctx, dst, d_src, n));
    }
    wmem->len += n;
    return HUF_ERROR_SUCCESS;
}

static int relaxed_tree(void)
{
    if (g_relaxed < 0) {
        const char *e = getenv("HUF_GPU_RELAXED_TREE");
        g_relaxed = (e && atoi(e) != 0) ? 1 : 0;
    }
    return g_relaxed;
}

/* Exported switch (not part of the reference API): 1 = accept tree_len 1025 on decode. */
void huf_gpu_set_relaxed_tree(int enabled) { g_relaxed = enabled ? 1 : 0; }

/* ------------------------------------------------------------------ encoder / decoder objects */
struct __huf_encoder {
    huf_config_t *config;
    huf_bufio_read_writer_t *bufio_writer;
    huf_bufio_read_writer_t *bufio_reader;
};
struct __huf_decoder {
    huf_config_t *config;
    huf_bufio_read_writer_t *bufio_writer;
    huf_bufio_read_writer_t *bufio_reader;
};

static huf_error_t codec_init(huf_config_t **cfg, huf_bufio_read_writer_t **w, huf_bufio_read_writer_t **r,
                              const huf_config_t *config)
{
    GUARD(config);
    if (!config->reader || !config->writer) return HUF_ERROR_INVALID_ARGUMENT;   /* the reference crashes */
    TRY(huf_config_init(cfg));
    memcpy(*cfg, config, sizeof(*config));       /* private copy: the caller's struct is never written */
    TRY(huf_bufio_read_writer_init(w, (*cfg)->writer, (*cfg)->writer_buffer_size));
    TRY(huf_bufio_read_writer_init(r, (*cfg)->reader, (*cfg)->reader_buffer_size));
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_encoder_init(huf_encoder_t **self, const huf_config_t *config)
{
    GUARD(self); GUARD(config);
    huf_encoder_t *e = (huf_encoder_t *)calloc(1, sizeof(*e));
    if (!e) return HUF_ERROR_MEMORY_ALLOCATION;
    *self = e;
    huf_error_t err = codec_init(&e->config, &e->bufio_writer, &e->bufio_reader, config);
    if (err == HUF_ERROR_SUCCESS && !e->config->blocksize) e->config->blocksize = e->config->length;   /* encoder.c:163-165 */
    if (err != HUF_ERROR_SUCCESS) huf_encoder_free(self);
    return err;
}

huf_error_t huf_encoder_free(huf_encoder_t **self)
{
    GUARD(self);
    if (*self) {
        huf_bufio_read_writer_free(&(*self)->bufio_writer);
        huf_bufio_read_writer_free(&(*self)->bufio_reader);
        huf_config_free(&(*self)->config);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

huf_error_t huf_decoder_init(huf_decoder_t **self, const huf_config_t *config)
{
    GUARD(self); GUARD(config);
    huf_decoder_t *d = (huf_decoder_t *)calloc(1, sizeof(*d));
    if (!d) return HUF_ERROR_MEMORY_ALLOCATION;
    *self = d;
    huf_error_t err = codec_init(&d->config, &d->bufio_writer, &d->bufio_reader, config);
    if (err != HUF_ERROR_SUCCESS) huf_decoder_free(self);
    return err;
}

huf_error_t huf_decoder_free(huf_decoder_t **self)
{
    GUARD(self);
    if (*self) {
        huf_bufio_read_writer_free(&(*self)->bufio_writer);
        huf_bufio_read_writer_free(&(*self)->bufio_reader);
        huf_config_free(&(*self)->config);
        free(*self);
    }
    *self = NULL;
    return HUF_ERROR_SUCCESS;
}

/* A stream made by huf_memopen() is this library's own object: the codec then copies between
 * its buffer and the device directly instead of through read()/write() and a staging buffer
 * (one host memcpy less per direction; the stream's cursor and length move exactly as the
 * callbacks would have moved them).  Any other stream goes through its callbacks. */
static membuf_t *own_memstream_reader(const huf_read_writer_t *rw) { return (rw && rw->read == memread) ? (membuf_t *)rw->stream : NULL; }
static membuf_t *own_memstream_writer(const huf_read_writer_t *rw) { return (rw && rw->write == memwrite) ? (membuf_t *)rw->stream : NULL; }
static int zero_copy_enabled(void)
{
    const char *e = getenv("HUF_GPU_ZERO_COPY");
    return !(e && atoi(e) == 0);
}

/* ------------------------------------------------------------------ fd streams: I/O next to the GPU work
 * A stream made by huf_fdopen() is this library's own object too: its read(2)/write(2) calls can
 * run on a helper thread while the calling thread drives the GPU, in order and one at a time per
 * descriptor.  Two pinned buffers per direction: the reader fills one while the other is encoded,
 * the writer drains one while the next result arrives.  Streams with foreign
 * callbacks are never touched from a helper thread (§8b: callbacks run serially on the caller's
 * thread). */
typedef struct {
    pthread_t thread;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int started;
    int fd;
    int writer;              /* 0: fills the slots from fd, 1: drains them to fd */
    char *buf[2];
    size_t len[2];           /* bytes in the slot (reader: what the read returned) */
    int full[2];             /* reader: filled, waiting for the consumer; writer: handed over, waiting for write(2) */
    uint64_t remaining;      /* reader: bytes still to be requested */
    size_t batch;            /* reader: bytes per request; writer: bytes a slot holds */
    int eof_ok;              /* reader: the end of the input is the consumer's business (decode), not a failure */
    int next;                /* writer: slot of the next fd_writer_push() piece */
    int quit;                /* consumer/producer side is done (or gave up) */
    huf_error_t err;
} fd_worker_t;

static void *fd_worker_main(void *arg)
{
    fd_worker_t *w = (fd_worker_t *)arg;
    for (int k = 0;; k ^= 1) {
        pthread_mutex_lock(&w->mu);
        if (w->writer) {
            while (!w->full[k] && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
            if (!w->full[k]) { pthread_mutex_unlock(&w->mu); break; }     /* quit and nothing handed over */
        } else {
            while (w->full[k] && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
            if (w->quit || !w->remaining) { pthread_mutex_unlock(&w->mu); break; }
        }
        pthread_mutex_unlock(&w->mu);
        huf_error_t err = HUF_ERROR_SUCCESS;
        size_t got = 0;
        if (w->writer) {
            if (w->err == HUF_ERROR_SUCCESS) err = fdwrite(&w->fd, w->buf[k], w->len[k]);   /* after a failure: drop */
        } else {
            got = w->remaining < w->batch ? (size_t)w->remaining : w->batch;
            const size_t want = got;
            err = fdread(&w->fd, w->buf[k], &got);
            if (err == HUF_ERROR_SUCCESS && got < want && !w->eof_ok) err = HUF_ERROR_READ_WRITE;   /* bufio.c:251-253 */
            w->remaining = (got < want) ? 0 : w->remaining - want;
        }
        pthread_mutex_lock(&w->mu);
        if (err != HUF_ERROR_SUCCESS && w->err == HUF_ERROR_SUCCESS) w->err = err;
        if (w->writer) w->full[k] = 0;
        else { w->len[k] = got; w->full[k] = 1; }
        pthread_cond_broadcast(&w->cv);
        const int stop = !w->writer && (err != HUF_ERROR_SUCCESS || !w->remaining);
        pthread_mutex_unlock(&w->mu);
        if (stop) break;
    }
    return NULL;
}

static huf_error_t fd_worker_start(fd_worker_t *w)
{
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    if (pthread_create(&w->thread, NULL, fd_worker_main, w) != 0) return HUF_ERROR_MEMORY_ALLOCATION;
    w->started = 1;
    return HUF_ERROR_SUCCESS;
}

/* the caller is done with the worker: a writer first drains what was handed over */
static huf_error_t fd_worker_finish(fd_worker_t *w)
{
    if (!w->started) return HUF_ERROR_SUCCESS;
    pthread_mutex_lock(&w->mu);
    w->quit = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
    pthread_join(w->thread, NULL);
    pthread_mutex_destroy(&w->mu);
    pthread_cond_destroy(&w->cv);
    w->started = 0;
    return w->err;
}

/* reader slot k: wait for its bytes (a short or failed read is reported with the slot it hit) */
static huf_error_t fd_reader_wait(fd_worker_t *w, int k, size_t want)
{
    pthread_mutex_lock(&w->mu);
    while (!w->full[k]) pthread_cond_wait(&w->cv, &w->mu);
    const huf_error_t err = (w->len[k] < want) ? (w->err != HUF_ERROR_SUCCESS ? w->err : HUF_ERROR_READ_WRITE)
                                                : HUF_ERROR_SUCCESS;
    pthread_mutex_unlock(&w->mu);
    return err;
}

static void fd_reader_release(fd_worker_t *w, int k)
{
    pthread_mutex_lock(&w->mu);
    w->full[k] = 0;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
}

/* writer slot k: wait until its previous content is on the descriptor */
static huf_error_t fd_writer_wait(fd_worker_t *w, int k)
{
    pthread_mutex_lock(&w->mu);
    while (w->full[k]) pthread_cond_wait(&w->cv, &w->mu);
    const huf_error_t err = w->err;
    pthread_mutex_unlock(&w->mu);
    return err;
}

static void fd_writer_submit(fd_worker_t *w, int k, size_t len)
{
    pthread_mutex_lock(&w->mu);
    w->len[k] = len;
    w->full[k] = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
}

/* len bytes at d_src -> the descriptor, through the slots (a piece per slot) */
static huf_error_t fd_writer_push(fd_worker_t *w, const void *d_src, uint64_t len)
{
    const char *p = (const char *)d_src;
    while (len) {
        const size_t n = len < w->batch ? (size_t)len : w->batch;
        const int k = w->next;
        TRY(fd_writer_wait(w, k));
        TRY(hufgpu_memcpy_d2h(g_ctx, w->buf[k], p, n));
        fd_writer_submit(w, k, n);
        w->next ^= 1;
        p += n;
        len -= n;
    }
    return HUF_ERROR_SUCCESS;
}

static int own_fd_of(const huf_read_writer_t *rw, int writer)
{
    if (!rw || !rw->stream) return -1;
    if (writer ? rw->write != fdwrite : rw->read != fdread) return -1;
    return *(const int *)rw->stream;
}

/* ------------------------------------------------------------------ huf_encode (src/encoder.c:261-388) */
#define SMALL_DECODE_BYTES ((uint64_t)128 << 10)  /* (round 6: streams of up to 128 KiB decode in one workgroup's chain with one wait - 64 KiB: 141 -> ~70 us) */
#define SMALL_CALL_BYTES ((uint64_t)32 << 10)     /* (1 B: 56 -> 31 us, 4 KiB: 78 -> 54; from 64 KiB on the bound-sized copy back costs more than the waits) */
static huf_error_t encode_rounds(huf_encoder_t *enc, uint64_t batch, membuf_t *rmem, membuf_t *wmem,
                                 fd_worker_t *rd, fd_worker_t *wr)
{
    const uint64_t length = enc->config->length;
    const uint64_t blocksize = enc->config->blocksize;
    const uint64_t bound = hufgpu_encode_bound(batch, blocksize);
    int round = 0;
    /* a small call between two memory streams: one synchronisation instead of three (hufgpu_encode_small) */
    if (rmem && wmem && length <= SMALL_CALL_BYTES && rmem->len - rmem->off >= length) {
        const uint64_t b8 = ((bound + 7u) & ~7ull) + 8u;
        TRY(grow_host(&g_stage.h_a, &g_stage.h_a_cap, length));
        TRY(grow_host(&g_stage.h_b, &g_stage.h_b_cap, b8));
        memcpy(g_stage.h_a, (const char *)*rmem->buf + rmem->off, length);
        uint64_t out_len = 0;
        const int rc = hufgpu_encode_small(g_ctx, g_stage.h_a, length, blocksize, g_stage.d_a, g_stage.d_b, g_stage.d_b_cap,
                                           g_stage.h_b, g_stage.h_b_cap, &out_len);
        if (rc != HUF_ERROR_SUCCESS) return (huf_error_t)rc;
        rmem->off += length;
        return memwrite(wmem, g_stage.h_b, out_len);
    }
    for (uint64_t done = 0; done < length; round ^= 1) {
        const uint64_t take = (length - done < batch) ? length - done : batch;
        /* one large read per round; a short read is an error exactly like the reference's
         * block read (src/encoder.c:296, src/bufio.c:251-253) */
        int rc = HUF_ERROR_SUCCESS;
        if (rmem) {
            if (rmem->len - rmem->off < take) {
                rmem->off = rmem->len;                          /* what a failed read would have consumed */
                rc = HUF_ERROR_READ_WRITE;
            } else {
                rc = lane_copy(1, g_stage.d_a, (char *)*rmem->buf + rmem->off, take);
                if (rc == HUF_ERROR_SUCCESS) rmem->off += take;
            }
        } else if (rd->started) {
            rc = fd_reader_wait(rd, round, take);
            if (rc == HUF_ERROR_SUCCESS) rc = hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, rd->buf[round], take);
            if (rc == HUF_ERROR_SUCCESS) fd_reader_release(rd, round);   /* the next read starts under the encode */
        } else {
            rc = huf_bufio_read(enc->bufio_reader, g_stage.h_a, take);
            if (rc == HUF_ERROR_SUCCESS) rc = hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, g_stage.h_a, take);
        }
        uint64_t out_len = 0;
        if (rc == HUF_ERROR_SUCCESS)
            rc = hufgpu_encode(g_ctx, g_stage.d_a, take, blocksize, g_stage.d_b, g_stage.d_b_cap, NULL, &out_len, NULL);
        if (rc != HUF_ERROR_SUCCESS) return (huf_error_t)rc;
        if (wmem) {
            TRY(d2h_to_memstream(wmem, g_stage.d_b, out_len));
        } else if (wr->started) {
            TRY(fd_writer_push(wr, g_stage.d_b, out_len));      /* waits for the write of two rounds ago */
        } else {
            TRY(hufgpu_memcpy_d2h(g_ctx, g_stage.h_b, g_stage.d_b, out_len));
            TRY(huf_bufio_write(enc->bufio_writer, g_stage.h_b, out_len));
        }
        done += take;
    }
    (void)bound;
    return HUF_ERROR_SUCCESS;
}

/* One huf_encode() over several sessions (HUF_GPU_DEVICES lists more than one and some are free):
 * memory stream -> memory stream only.  The input is cut into rounds of whole blocks - blocks are
 * independent (src/encoder.c:288-374, reset :360-373), so the stream is the rounds' streams one after
 * the other, byte for byte what one session writes.  Every session runs on a thread of its own:
 * input round to its device, encode, and - once the sizes of all earlier rounds are known - the
 * result to its place in the output.  With sessions on different GPUs the rounds travel over
 * different host links; with two sessions on ONE GPU a round's copy back runs beside the next
 * round's copy in (full duplex).  The output's pages are made present before the first copy (a
 * populate beside running copies fights them for the address-space lock, see d2h_to_memstream). */
typedef struct {
    const char *src;
    char *dst;
    uint64_t length, blocksize, round_bytes, nrounds;
    std::atomic<uint64_t> next;
    uint64_t *out_len;              /* per round, valid once known[k] */
    unsigned char *known;
    unsigned char *done;            /* per round: its stream is in place in dst (written under mu, read after the joins) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    std::atomic<int> err;
} fanout_t;

static std::atomic<int> g_fanout_decodes(0), g_fanout_encodes(0);     /* huf_gpu_fanouts() */

typedef struct { fanout_t *f; session_t *session; int extra; } fanout_worker_t;   /* extra: not the call's own session */

#define HUF_MAX_LINKS 64
static pthread_mutex_t g_link_lock[HUF_MAX_LINKS][2];               /* per device: [0] host -> device, [1] device -> host */
static pthread_once_t g_link_once = PTHREAD_ONCE_INIT;
static void link_locks_init(void)
{
    for (int i = 0; i < HUF_MAX_LINKS; i++) {
        pthread_mutex_init(&g_link_lock[i][0], NULL);
        pthread_mutex_init(&g_link_lock[i][1], NULL);
    }
}

static void fanout_fail(fanout_t *f, int err)
{
    pthread_mutex_lock(&f->mu);
    int none = HUF_ERROR_SUCCESS;
    f->err.compare_exchange_strong(none, err);
    pthread_cond_broadcast(&f->cv);
    pthread_mutex_unlock(&f->mu);
}

static void *fanout_main(void *arg)
{
    fanout_worker_t *w = (fanout_worker_t *)arg;
    fanout_t *f = w->f;
    t_session = w->session;                                          /* this thread's g_ctx / g_stage */
    int rc = session_acquire();
    const uint64_t bound = hufgpu_encode_bound(f->round_bytes, f->blocksize);
    if (rc == HUF_ERROR_SUCCESS) rc = grow_dev(&g_stage.d_a, &g_stage.d_a_cap, f->round_bytes);
    if (rc == HUF_ERROR_SUCCESS) rc = grow_dev(&g_stage.d_b, &g_stage.d_b_cap, bound);
    if (rc != HUF_ERROR_SUCCESS && w->extra) {
        /* an EXTRA session that cannot be set up (a device of HUF_GPU_DEVICES without memory left, a context
         * that cannot be created) has taken no round yet: the call goes on with the sessions that work */
        t_session = NULL;
        return NULL;
    }
    while (rc == HUF_ERROR_SUCCESS) {
        if (f->err.load() != HUF_ERROR_SUCCESS) break;                   /* another session failed */
        const uint64_t k = f->next.fetch_add(1);
        if (k >= f->nrounds) break;
        const uint64_t off = k * f->round_bytes;
        const uint64_t take = (f->length - off < f->round_bytes) ? f->length - off : f->round_bytes;
        uint64_t out_len = 0;
        /* sessions on one device take turns per direction: while one copies a result back the next
         * copies its input in (both at once in the SAME direction only share the link, and all
         * sessions would move through their phases in step) */
        pthread_mutex_t *dir = g_link_lock[(unsigned)w->session->device % HUF_MAX_LINKS];
        pthread_mutex_lock(&dir[0]);
        rc = hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, f->src + off, take);
        pthread_mutex_unlock(&dir[0]);
        if (rc == HUF_ERROR_SUCCESS)
            rc = hufgpu_encode(g_ctx, g_stage.d_a, take, f->blocksize, g_stage.d_b, g_stage.d_b_cap, NULL, &out_len, NULL);
        if (rc != HUF_ERROR_SUCCESS) break;
        /* publish this round's size, then wait for the sizes of all rounds in front of it */
        uint64_t before = 0;
        pthread_mutex_lock(&f->mu);
        f->out_len[k] = out_len;
        f->known[k] = 1;
        pthread_cond_broadcast(&f->cv);
        for (;;) {
            uint64_t j = 0;
            before = 0;
            while (j < k && f->known[j]) before += f->out_len[j++];
            if (j == k) break;                                       /* every earlier size is known: this round still lands, */
            if (f->err.load() != HUF_ERROR_SUCCESS) break;           /* even after another session failed behind it */
            pthread_cond_wait(&f->cv, &f->mu);
        }
        uint64_t j2 = 0;
        while (j2 < k && f->known[j2]) j2++;
        const int stop = j2 < k;                                         /* (only possible after a failure) */
        pthread_mutex_unlock(&f->mu);
        if (stop) break;
        pthread_mutex_lock(&dir[1]);
        rc = hufgpu_memcpy_d2h(g_ctx, f->dst + before, g_stage.d_b, out_len);
        pthread_mutex_unlock(&dir[1]);
        if (rc == HUF_ERROR_SUCCESS) {
            pthread_mutex_lock(&f->mu);
            f->done[k] = 1;
            pthread_mutex_unlock(&f->mu);
        }
    }
    if (rc != HUF_ERROR_SUCCESS) fanout_fail(f, rc);
    t_session = NULL;
    return NULL;
}

/* returns 1 when the call was done here (*result = its outcome), 0 when the ordinary path should run */
static int encode_fanout(huf_encoder_t *enc, membuf_t *rmem, membuf_t *wmem, huf_error_t *result)
{
    const uint64_t length = enc->config->length, blocksize = enc->config->blocksize;
    if (!rmem || !wmem || wmem->readonly || g_nsessions < 2) return 0;
    if (rmem->len - rmem->off < length) return 0;                     /* a short input: the ordinary path reports it */
    const char *env = getenv("HUF_GPU_BATCH_MB");
    uint64_t round_bytes = (uint64_t)(env && atoi(env) > 0 ? atoi(env) : 32) << 20;
    if (round_bytes < blocksize) round_bytes = blocksize;
    round_bytes -= round_bytes % blocksize;
    const uint64_t nrounds = (length + round_bytes - 1) / round_bytes;
    if (nrounds < 2) return 0;

    session_t *mine = t_session;
    session_t *extra[HUF_MAX_SESSIONS];
    int nextra = 0;
    while ((uint64_t)nextra + 1 < nrounds && nextra < HUF_MAX_SESSIONS - 1) {
        session_t *s = session_try_extra();
        if (!s) break;
        extra[nextra++] = s;
    }
    if (nextra == 0) return 0;                                        /* every other session is busy: one after the other */
    pthread_once(&g_link_once, link_locks_init);

    uint64_t bound = 0;
    for (uint64_t k = 0; k < nrounds; k++) {
        const uint64_t off = k * round_bytes;
        bound += hufgpu_encode_bound(length - off < round_bytes ? length - off : round_bytes, blocksize);
    }
    fanout_t f;
    f.src = (const char *)*rmem->buf + rmem->off;
    f.length = length;
    f.blocksize = blocksize;
    f.round_bytes = round_bytes;
    f.nrounds = nrounds;
    f.next.store(0);
    f.err.store(HUF_ERROR_SUCCESS);
    f.out_len = (uint64_t *)calloc(nrounds, sizeof(uint64_t));
    f.known = (unsigned char *)calloc(nrounds, 1);
    f.done = (unsigned char *)calloc(nrounds, 1);
    huf_error_t err = (f.out_len && f.known && f.done) ? mem_reserve(wmem, bound) : HUF_ERROR_MEMORY_ALLOCATION;
    if (err == HUF_ERROR_SUCCESS) {
        f.dst = (char *)*wmem->buf + wmem->len;
        prefault_job_t job;                                           /* about as many bytes as the stream will have */
        prefault_begin(&job, f.dst, (size_t)(length < bound ? length : bound), 1);
        prefault_end(&job);
        pthread_mutex_init(&f.mu, NULL);
        pthread_cond_init(&f.cv, NULL);
        fanout_worker_t workers[HUF_MAX_SESSIONS];
        pthread_t th[HUF_MAX_SESSIONS];
        int started = 0;
        for (int i = 0; i < nextra; i++) {
            workers[i + 1].f = &f;
            workers[i + 1].session = extra[i];
            workers[i + 1].extra = 1;
            if (pthread_create(&th[i], NULL, fanout_main, &workers[i + 1]) != 0) break;
            started++;
        }
        workers[0].f = &f;
        workers[0].session = mine;
        workers[0].extra = 0;
        fanout_main(&workers[0]);                                     /* this thread works with the call's own session */
        t_session = mine;
        for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
        pthread_mutex_destroy(&f.mu);
        pthread_cond_destroy(&f.cv);
        err = (huf_error_t)f.err.load();
        /* what the reference's unbuffered writer has delivered when it fails stays delivered: the rounds in
         * front of the first one that is not in place (all of them on success) */
        uint64_t total = 0, p = 0;
        while (p < nrounds && f.done[p]) total += f.out_len[p++];
        if (err == HUF_ERROR_SUCCESS && p < nrounds) err = HUF_ERROR_FATAL;   /* (cannot happen: every round was taken) */
        wmem->len += total;
        rmem->off += (p == nrounds) ? length : p * round_bytes;
        if (err == HUF_ERROR_SUCCESS) g_fanout_encodes.fetch_add(1);
    }
    for (int i = 0; i < nextra; i++) session_release_extra(extra[i]);
    free(f.out_len);
    free(f.known);
    free(f.done);
    *result = err;
    return 1;
}

/* HUF_GPU_DX_TRACE=1: where the calling thread of a duplex call spends its time, one line per call on stderr */
static int dx_trace(void)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("HUF_GPU_DX_TRACE"); on = (e && atoi(e) != 0) ? 1 : 0; }
    return on;
}
static double dx_now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
#define DX_T(acc, stmt) do { const double t_ = dx_trace() ? dx_now() : 0.0; stmt; if (dx_trace()) (acc) += dx_now() - t_; } while (0)

static int duplex_enabled(void)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("HUF_GPU_DUPLEX"); on = (e && atoi(e) == 0) ? 0 : 1; }
    return on;
}

/* room for `count` bytes behind the `pending` bytes that lie behind the stream's contents already (results of earlier
 * rounds, not yet counted in len).  A buffer that has to grow moves: the lanes that write into it finish first. */
static huf_error_t mem_reserve_behind(membuf_t *m, dx_pool_t *P, size_t pending, size_t count)
{
    if (m->cap - m->len >= pending + count) return HUF_ERROR_SUCCESS;
    dx_drain(P);
    m->len += pending;                               /* (what a grown buffer takes along) */
    const huf_error_t rc = mem_reserve(m, count);
    m->len -= pending;
    return rc;
}

/* huf_encode() between two memory streams in rounds whose transfers overlap (the comment above dx_lane_main): returns
 * 1 when it took the call (*result = what huf_encode returns), 0 when the call is not of that kind - the caller goes
 * on as before.  Rounds are whole blocks (src/encoder.c:288-374: blocks are independent), the stream is the rounds'
 * streams one after the other; a round that fails ends the call with the rounds in front of it delivered, as
 * encode_rounds does. */
static int encode_duplex(huf_encoder_t *enc, membuf_t *rmem, membuf_t *wmem, huf_error_t *result)
{
    const uint64_t length = enc->config->length, blocksize = enc->config->blocksize;
    if (!rmem || !wmem || !duplex_enabled() || length < DX_MIN_BYTES || rmem->len - rmem->off < length || blocksize == 0) return 0;
    uint64_t R = dx_round_bytes(length, blocksize < ((uint64_t)2 << 20) ? blocksize : 0)   /* (from 2 MiB on the encoder cuts blocks into chunks itself) */;
    if (R < blocksize) R = blocksize;
    R -= R % blocksize;
    if (length <= R + R / 2) return 0;
    dx_pool_t *P = dx_get();
    if (!P) return 0;
    const uint64_t bound = (hufgpu_encode_bound(R, blocksize) + 255u) & ~(uint64_t)255;
    if (grow_dev(&g_stage.d_a, &g_stage.d_a_cap, 2 * R) != HUF_ERROR_SUCCESS ||
        grow_dev(&g_stage.d_b, &g_stage.d_b_cap, 2 * bound) != HUF_ERROR_SUCCESS) return 0;
    if (!wmem->fixed && mem_reserve(wmem, hufgpu_encode_bound(length, blocksize)) != HUF_ERROR_SUCCESS) return 0;
    const uint64_t nr = (length + R - 1) / R;
    char *src = (char *)*rmem->buf + rmem->off;
    char *d_in[2] = {(char *)g_stage.d_a, (char *)g_stage.d_a + R};
    char *d_out[2] = {(char *)g_stage.d_b, (char *)g_stage.d_b + bound};
#define ROUND_BYTES(i) (((i) + 1) * R <= length ? R : length - (i) * R)
    void *const reg = dx_register_input(P, src, length);
    const int direct = reg != NULL;
    const uint64_t in0 = dx_publish(P, 0, src, d_in[0], ROUND_BYTES((uint64_t)0), direct);
    if (nr > 1) (void)dx_publish(P, 0, src + R, d_in[1], ROUND_BYTES((uint64_t)1), direct);
    uint64_t out0 = 0, out_total = 0, done_in = 0;
    huf_error_t err = HUF_ERROR_SUCCESS;
    double t_in = 0, t_out = 0, t_k = 0, t_end = 0;
    const double t_start = dx_trace() ? dx_now() : 0.0;
    for (uint64_t i = 0; i < nr; i++) {
        DX_T(t_in, err = dx_wait_issued(P, in0 + i));
        if (err == HUF_ERROR_SUCCESS && i >= 2) DX_T(t_out, err = dx_wait_done(P, 1, out0 + i - 2));      /* the round that used this output buffer */
        if (err != HUF_ERROR_SUCCESS) break;
        uint64_t out_len = 0;
        DX_T(t_k, err = (huf_error_t)hufgpu_encode(g_ctx, d_in[i & 1], ROUND_BYTES(i), blocksize, d_out[i & 1], bound, NULL, &out_len, NULL));
        done_in = (i + 1 < nr) ? (i + 1) * R : length;                                       /* (what a failed round has consumed, too) */
        if (err != HUF_ERROR_SUCCESS) break;
        err = mem_reserve_behind(wmem, P, out_total, out_len);
        if (err != HUF_ERROR_SUCCESS) break;
        const uint64_t id = dx_publish(P, 1, (char *)*wmem->buf + wmem->len + out_total, d_out[i & 1], out_len);
        if (i == 0) out0 = id;
        out_total += out_len;
        if (i + 2 < nr) (void)dx_publish(P, 0, src + (i + 2) * R, d_in[i & 1], ROUND_BYTES(i + 2), direct);   /* its kernels are done: the buffer is free */
    }
#undef ROUND_BYTES
    DX_T(t_end, dx_drain(P));
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); if (err == HUF_ERROR_SUCCESS) err = HUF_ERROR_FATAL; }   /* (copies of rounds a failure left behind) */
    dx_unregister_input(reg);
    if (dx_trace())
        fprintf(stderr, "encode_duplex: %llu rounds of %llu MiB, %.2f ms: waiting for input %.2f, for an output buffer %.2f, kernels (+ their wait) %.2f, the last copies %.2f; input registered=%d lanes %d/%d\n",
                (unsigned long long)nr, (unsigned long long)(R >> 20), (dx_now() - t_start) * 1e3, t_in * 1e3, t_out * 1e3, t_k * 1e3, t_end * 1e3, direct, P->nl[0], P->nl[1]);
    if (err == HUF_ERROR_SUCCESS && P->err) err = HUF_ERROR_FATAL;
    rmem->off += done_in;
    wmem->len += out_total;
    *result = err;
    return 1;
}

static huf_error_t encode_locked(huf_encoder_t *enc)
{
    const uint64_t length = enc->config->length;
    const uint64_t blocksize = enc->config->blocksize;
    if (blocksize > HUFGPU_MAX_BLOCK) {
        fprintf(stderr, "libhuffman: blocksize %llu exceeds the GPU kernel limit (%llu)\n",
                (unsigned long long)blocksize, (unsigned long long)HUFGPU_MAX_BLOCK);
        return HUF_ERROR_INVALID_ARGUMENT;
    }
    TRY(session_acquire());

    membuf_t *rmem = zero_copy_enabled() ? own_memstream_reader(enc->config->reader) : NULL;
    membuf_t *wmem = zero_copy_enabled() ? own_memstream_writer(enc->config->writer) : NULL;
    const int rfd = (rmem || !zero_copy_enabled()) ? -1 : own_fd_of(enc->config->reader, 0);
    const int wfd = (wmem || !zero_copy_enabled()) ? -1 : own_fd_of(enc->config->writer, 1);
    {
        huf_error_t fan = HUF_ERROR_SUCCESS;
        if (encode_fanout(enc, rmem, wmem, &fan) || encode_duplex(enc, rmem, wmem, &fan))
            return fan != HUF_ERROR_SUCCESS ? fan : huf_bufio_read_writer_flush(enc->bufio_writer);
    }

    /* bytes per round: whole blocks; smaller rounds when descriptor I/O runs next to the GPU
     * (the first read and the last write are not hidden) */
    const char *env = getenv("HUF_GPU_BATCH_MB");
    uint64_t batch = (uint64_t)(env && atoi(env) > 0 ? atoi(env) : ((rfd >= 0 || wfd >= 0) ? 32 : 256)) << 20;
    if (batch < blocksize) batch = blocksize;
    batch -= batch % blocksize;
    if (batch > length) batch = length;

    const uint64_t bound = hufgpu_encode_bound(batch, blocksize);
    if (!rmem) TRY(grow_host(&g_stage.h_a, &g_stage.h_a_cap, rfd >= 0 ? 2 * batch : batch));
    if (!wmem) TRY(grow_host(&g_stage.h_b, &g_stage.h_b_cap, wfd >= 0 ? 2 * bound : bound));
    TRY(grow_dev(&g_stage.d_a, &g_stage.d_a_cap, batch));
    TRY(grow_dev(&g_stage.d_b, &g_stage.d_b_cap, bound));

    fd_worker_t rd, wr;
    memset(&rd, 0, sizeof(rd));
    memset(&wr, 0, sizeof(wr));
    huf_error_t err = HUF_ERROR_SUCCESS;
    if (rfd >= 0) {
        rd.fd = rfd;
        rd.buf[0] = (char *)g_stage.h_a;
        rd.buf[1] = (char *)g_stage.h_a + batch;
        rd.remaining = length;
        rd.batch = (size_t)batch;
        err = fd_worker_start(&rd);
    }
    if (wfd >= 0 && err == HUF_ERROR_SUCCESS) {
        wr.fd = wfd;
        wr.writer = 1;
        wr.buf[0] = (char *)g_stage.h_b;
        wr.buf[1] = (char *)g_stage.h_b + bound;
        wr.batch = (size_t)bound;
        err = fd_worker_start(&wr);
    }
    if (err == HUF_ERROR_SUCCESS) err = encode_rounds(enc, batch, rmem, wmem, &rd, &wr);
    (void)fd_worker_finish(&rd);                                  /* its failures surfaced with their round */
    const huf_error_t werr = fd_worker_finish(&wr);               /* results of complete rounds still go out */
    if (err == HUF_ERROR_SUCCESS) err = werr;
    if (err != HUF_ERROR_SUCCESS) return err;
    return huf_bufio_read_writer_flush(enc->bufio_writer);        /* encoder.c:377 */
}

huf_error_t huf_encode(const huf_config_t *config)
{
    GUARD(config);
    huf_encoder_t *enc = NULL;
    TRY(huf_encoder_init(&enc, config));
    huf_error_t err = HUF_ERROR_SUCCESS;
    if (enc->config->length) {                    /* length 0: nothing is read or written */
        session_enter();
        err = encode_locked(enc);
        session_leave();
    }
    huf_encoder_free(&enc);
    return err;
}

/* ------------------------------------------------------------------ huf_decode (src/decoder.c:201-287) */
static huf_error_t read_upto(huf_read_writer_t *rw, uint8_t *dst, size_t want, size_t *got)
{
    size_t total = 0;
    while (total < want) {
        size_t n = want - total;
        TRY(rw->read(rw->stream, dst + total, &n));
        if (!n) break;
        total += n;
    }
    *got = total;
    return HUF_ERROR_SUCCESS;
}

/* like grow_dev(), but the first `keep` bytes survive */
static huf_error_t grow_dev_keep(void **p, size_t *cap, size_t want, size_t keep)
{
    if (*cap >= want) return HUF_ERROR_SUCCESS;
    void *bigger = NULL;
    TRY(hufgpu_malloc(g_ctx, &bigger, want));
    if (keep) {
        const int rc = hufgpu_memcpy_d2d(g_ctx, bigger, *p, keep);
        if (rc != HUF_ERROR_SUCCESS) { hufgpu_free(g_ctx, bigger); return (huf_error_t)rc; }
    }
    if (*p) hufgpu_free(g_ctx, *p);
    *p = bigger;
    *cap = want;
    return HUF_ERROR_SUCCESS;
}

/* the helper thread's next piece (it asked for `want` bytes) -> behind the `*loaded` stream bytes on the device */
static huf_error_t fd_reader_to_device(fd_worker_t *rd, int *rslot, uint64_t want, uint64_t *loaded, int *eof)
{
    const int k = *rslot;
    pthread_mutex_lock(&rd->mu);
    while (!rd->full[k]) pthread_cond_wait(&rd->cv, &rd->mu);
    const size_t got = rd->len[k];
    const huf_error_t rerr = rd->err;
    pthread_mutex_unlock(&rd->mu);
    if (rerr != HUF_ERROR_SUCCESS) return rerr;
    TRY(hufgpu_memcpy_h2d(g_ctx, (char *)g_stage.d_a + *loaded, rd->buf[k], got));
    fd_reader_release(rd, k);
    *rslot = k ^ 1;
    *loaded += got;
    if (got < want) *eof = 1;
    return HUF_ERROR_SUCCESS;
}

/* Decode with the input on a huf_fdopen() descriptor: the file is read by a helper thread in
 * pieces that go to the device as they arrive, and the stream is decoded in rounds of `piece`
 * compressed bytes - the block loop of src/decoder.c:218 cut at block boundaries: a round starts
 * where the previous one stopped and runs while fewer than its share of bytes is consumed, which
 * is the reference's loop condition with more check points.  Each round's output leaves through
 * the writer (a helper thread as well when that is a descriptor) while the next pieces are read
 * and decoded.  A round whose last block needs bytes that are not there yet is repeated once
 * they are; past `length` the descriptor is asked for more like the reference's on-demand reads. */
static huf_error_t decode_rounds_fd(huf_decoder_t *dec, fd_worker_t *rd, membuf_t *wmem, fd_worker_t *wr,
                                    uint64_t piece, uint32_t flags)
{
    const uint64_t length = dec->config->length;
    const uint64_t margin = 4u << 20;            /* what a round may look ahead before it is worth starting */
    uint64_t loaded = 0;                         /* bytes of the stream on the device (g_stage.d_a) */
    uint64_t requested = 0;                      /* of `length`, by the helper thread */
    int rslot = 0, eof = 0;
    uint64_t pos = 0;
    uint64_t out_cap = (piece + margin) * 8 + (1u << 20);
    TRY(grow_dev(&g_stage.d_a, &g_stage.d_a_cap, (size_t)length + 16));

    while (pos < length) {
        const uint64_t round_len = (length - pos < piece) ? length - pos : piece;
        /* input up to the round's end plus the margin, or all there is */
        while (!eof && requested < length && loaded < pos + round_len + margin) {
            TRY(fd_reader_to_device(rd, &rslot, (length - requested < piece) ? length - requested : piece, &loaded, &eof));
            requested = (length - requested < piece) ? length : requested + piece;
        }
        const uint64_t avail = loaded - pos;
        /* the round's bytes at an aligned address (the parallel block discovery wants that) */
        TRY(grow_dev(&g_stage.d_c, &g_stage.d_c_cap, (size_t)avail + 16));
        TRY(hufgpu_memcpy_d2d(g_ctx, g_stage.d_c, (const char *)g_stage.d_a + pos, avail));
        TRY(grow_dev(&g_stage.d_b, &g_stage.d_b_cap, out_cap));
        uint64_t raw = 0, used = 0;
        int rc = hufgpu_decode_stream(g_ctx, g_stage.d_c, avail, round_len, g_stage.d_b, g_stage.d_b_cap, flags, &raw, &used, NULL);
        if (rc == HUF_ERROR_MEMORY_ALLOCATION && out_cap < ((uint64_t)1 << 40)) {   /* output did not fit: enlarge */
            out_cap *= 4;
            continue;
        }
        if (rc == HUF_ERROR_READ_WRITE) {
            if (!eof && requested < length) {            /* more of the stream is on its way: take a piece, again */
                TRY(fd_reader_to_device(rd, &rslot, (length - requested < piece) ? length - requested : piece, &loaded, &eof));
                requested = (length - requested < piece) ? length : requested + piece;
                continue;
            }
            if (!eof) {                                   /* maybe the descriptor holds more than `length` */
                const size_t more_want = loaded < 65536 ? 65536 : (size_t)loaded;
                TRY(grow_dev_keep(&g_stage.d_a, &g_stage.d_a_cap, (size_t)loaded + more_want + 16, (size_t)loaded));
                size_t more = 0;
                while (more < more_want) {                /* the helper thread has finished: read here */
                    size_t n = more_want - more < rd->batch ? more_want - more : rd->batch;
                    const size_t asked = n;
                    TRY(fdread(&rd->fd, rd->buf[0], &n));
                    TRY(hufgpu_memcpy_h2d(g_ctx, (char *)g_stage.d_a + loaded + more, rd->buf[0], n));
                    more += n;
                    if (n < asked) { eof = 1; break; }
                }
                loaded += more;
                if (more) continue;
            }
        }
        /* bytes of the blocks that decoded completely are delivered even when a later block
         * fails, as the reference's unbuffered writer would have done */
        if (raw && wmem) {
            TRY(d2h_to_memstream(wmem, g_stage.d_b, raw));
        } else if (raw && wr->started) {
            TRY(fd_writer_push(wr, g_stage.d_b, raw));
        } else if (raw) {
            TRY(grow_host(&g_stage.h_b, &g_stage.h_b_cap, raw));
            TRY(hufgpu_memcpy_d2h(g_ctx, g_stage.h_b, g_stage.d_b, raw));
            TRY(huf_bufio_write(dec->bufio_writer, g_stage.h_b, raw));
        }
        if (rc != HUF_ERROR_SUCCESS) return (huf_error_t)rc;
        pos += used;
    }
    return HUF_ERROR_SUCCESS;
}

static huf_error_t decode_from_fd(huf_decoder_t *dec, int rfd, membuf_t *wmem, int wfd, uint32_t flags)
{
    const uint64_t length = dec->config->length;
    const char *env = getenv("HUF_GPU_BATCH_MB");
    uint64_t piece = (uint64_t)(env && atoi(env) > 0 ? atoi(env) : 32) << 20;
    if (piece > length) piece = length;
    if (piece < 65536) piece = 65536;                             /* also the size of the reads past `length` */
    TRY(grow_host(&g_stage.h_a, &g_stage.h_a_cap, 2 * piece));
    if (wfd >= 0) TRY(grow_host(&g_stage.h_b, &g_stage.h_b_cap, 2 * piece));

    fd_worker_t rd, wr;
    memset(&rd, 0, sizeof(rd));
    memset(&wr, 0, sizeof(wr));
    rd.fd = rfd;
    rd.buf[0] = (char *)g_stage.h_a;
    rd.buf[1] = (char *)g_stage.h_a + piece;
    rd.remaining = length;
    rd.batch = (size_t)piece;
    rd.eof_ok = 1;
    huf_error_t err = fd_worker_start(&rd);
    if (wfd >= 0 && err == HUF_ERROR_SUCCESS) {
        wr.fd = wfd;
        wr.writer = 1;
        wr.buf[0] = (char *)g_stage.h_b;
        wr.buf[1] = (char *)g_stage.h_b + piece;
        wr.batch = (size_t)piece;
        err = fd_worker_start(&wr);
    }
    if (err == HUF_ERROR_SUCCESS) err = decode_rounds_fd(dec, &rd, wmem, &wr, piece, flags);
    (void)fd_worker_finish(&rd);
    const huf_error_t werr = fd_worker_finish(&wr);               /* what was delivered before a failure still goes out */
    if (err == HUF_ERROR_SUCCESS) err = werr;
    if (err != HUF_ERROR_SUCCESS) return err;                     /* no flush on the error path (decoder.c:278-286) */
    return huf_bufio_read_writer_flush(dec->bufio_writer);
}

/* pieces = NULL: huf_decode().  pieces != NULL: huf_gpu_decode_blocks() - only the blocks that lie
 * completely inside `length` bytes, *pieces = their stream bytes, a cut-off last block is no error. */
/* One huf_decode() over several sessions (the twin of encode_fanout): memory stream -> memory stream, sessions
 * free.  Blocks are independent once they are found (src/decoder.c:218-276), and finding them is a device job
 * of a few milliseconds per GiB (hufgpu_block_index: every candidate header probed count-only, the chain
 * walked).  So: the stream goes to the call's own device, its block index comes back, the blocks are dealt out
 * in contiguous ranges balanced by COMPRESSED bytes, and every session - a thread of its own -
 * takes its range of the stream from host memory, decodes it with the indexed kernels and writes its output
 * where it belongs (the sum of the block_len fields in front of it).  Anything unusual - a stream the walk
 * cannot validate to its end, an error in any range - leaves the whole call to the ordinary path, which
 * reports what the reference reports; nothing has been committed by then. */

typedef struct {
    const char *src;                /* the stream in host memory */
    char *dst;                      /* the output's place in the writer's buffer */
    const uint64_t *offs;           /* nblocks + 1 header offsets (host) */
    const uint64_t *outoff;         /* nblocks + 1 output offsets (host) */
    uint64_t b0, b1;                /* this worker's blocks */
    uint32_t flags;
    session_t *session;
    int own;                        /* the call's own session: the stream is already on its device (at d_a) */
    const uint64_t *d_index;        /* own: the device index */
    int rc;
} dfan_worker_t;

static void *dfan_main(void *arg)
{
    dfan_worker_t *w = (dfan_worker_t *)arg;
    w->rc = HUF_ERROR_SUCCESS;
    if (w->b1 <= w->b0) return NULL;
    session_t *const before = t_session;
    t_session = w->session;
    int rc = session_acquire();
    const uint64_t nb = w->b1 - w->b0;
    const uint64_t s0 = w->offs[w->b0], s1 = w->offs[w->b1];
    const uint64_t raw = w->outoff[w->b1] - w->outoff[w->b0];
    pthread_mutex_t *dir = g_link_lock[(unsigned)w->session->device % HUF_MAX_LINKS];
    uint64_t got = 0;
    if (rc == HUF_ERROR_SUCCESS) rc = grow_dev(&g_stage.d_b, &g_stage.d_b_cap, raw + 16);
    if (rc == HUF_ERROR_SUCCESS && w->own) {
        rc = hufgpu_decode(g_ctx, g_stage.d_a, s1, w->d_index + w->b0, nb, g_stage.d_b, g_stage.d_b_cap, w->flags, &got, NULL);
    } else if (rc == HUF_ERROR_SUCCESS) {
        uint64_t *rel = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
        if (!rel) rc = HUF_ERROR_MEMORY_ALLOCATION;
        if (rc == HUF_ERROR_SUCCESS) rc = grow_dev(&g_stage.d_a, &g_stage.d_a_cap, s1 - s0 + 16);
        if (rc == HUF_ERROR_SUCCESS) rc = grow_dev(&g_stage.d_c, &g_stage.d_c_cap, (nb + 1) * sizeof(uint64_t));
        if (rc == HUF_ERROR_SUCCESS) {
            for (uint64_t i = 0; i <= nb; i++) rel[i] = w->offs[w->b0 + i] - s0;
            pthread_mutex_lock(&dir[0]);
            rc = hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, w->src + s0, s1 - s0);
            if (rc == HUF_ERROR_SUCCESS) rc = hufgpu_memcpy_h2d(g_ctx, g_stage.d_c, rel, (nb + 1) * sizeof(uint64_t));
            pthread_mutex_unlock(&dir[0]);
        }
        if (rc == HUF_ERROR_SUCCESS)
            rc = hufgpu_decode(g_ctx, g_stage.d_a, s1 - s0, (const uint64_t *)g_stage.d_c, nb, g_stage.d_b, g_stage.d_b_cap, w->flags,
                               &got, NULL);
        free(rel);
    }
    if (rc == HUF_ERROR_SUCCESS && got != raw) rc = HUF_ERROR_FATAL;       /* (the block_len fields said otherwise) */
    if (rc == HUF_ERROR_SUCCESS) {
        pthread_mutex_lock(&dir[1]);
        rc = hufgpu_memcpy_d2h(g_ctx, w->dst + w->outoff[w->b0], g_stage.d_b, raw);
        pthread_mutex_unlock(&dir[1]);
    }
    w->rc = rc;
    t_session = before;
    return NULL;
}

/* returns 1 when the call was done here (*result = its outcome), 0 when the ordinary path should run */
static int decode_fanout(huf_decoder_t *dec, membuf_t *rmem, membuf_t *wmem, uint32_t flags, huf_error_t *result)
{
    const uint64_t length = dec->config->length;
    if (!rmem || !wmem || wmem->readonly || g_nsessions < 2) return 0;
    if (rmem->len - rmem->off < length) return 0;
    const char *env = getenv("HUF_GPU_FANOUT_MIN_MB");
    const uint64_t min_bytes = (uint64_t)(env && atoi(env) > 0 ? atoi(env) : 64) << 20;
    if (length < min_bytes) return 0;
    session_t *mine = t_session;
    session_t *extra[HUF_MAX_SESSIONS];
    int nextra = 0;
    while (nextra < HUF_MAX_SESSIONS - 1) {
        session_t *s = session_try_extra();
        if (!s) break;
        extra[nextra++] = s;
    }
    if (nextra == 0) return 0;
    pthread_once(&g_link_once, link_locks_init);
    const char *src = (const char *)*rmem->buf + rmem->off;
    uint64_t *offs = NULL, *outoff = NULL;
    int done = 0;
    do {
        /* the stream on the call's own device, and its block index */
        if (grow_dev(&g_stage.d_a, &g_stage.d_a_cap, length + 16) != HUF_ERROR_SUCCESS) break;
        if (hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, src, length) != HUF_ERROR_SUCCESS) break;
        const uint64_t *d_index = NULL;
        uint64_t nb = 0, used = 0;
        if (hufgpu_block_index(g_ctx, g_stage.d_a, length, length, flags, &d_index, &nb, &used, NULL) != HUF_ERROR_SUCCESS) break;
        if (nb < 2 || used != length) break;                          /* not validated to its end: the ordinary path */
        offs = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
        outoff = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
        if (!offs || !outoff) break;
        if (hufgpu_memcpy_d2h(g_ctx, offs, d_index, (nb + 1) * sizeof(uint64_t)) != HUF_ERROR_SUCCESS) break;
        outoff[0] = 0;
        int sane = offs[0] == 0 && offs[nb] == length;
        for (uint64_t b = 0; b < nb && sane; b++) {
            if (offs[b + 1] < offs[b] + 10 || offs[b + 1] > length) { sane = 0; break; }
            uint64_t bl;
            memcpy(&bl, src + offs[b], sizeof(bl));                   /* block_len, little-endian (src/encoder.c:325) */
            if (bl > ((uint64_t)1 << 40)) { sane = 0; break; }
            outoff[b + 1] = outoff[b] + bl;
        }
        if (!sane) break;
        const uint64_t total = outoff[nb];
        if (mem_reserve(wmem, total) != HUF_ERROR_SUCCESS) break;
        char *dst = (char *)*wmem->buf + wmem->len;
        prefault_job_t job;
        prefault_begin(&job, dst, (size_t)total, 1);
        prefault_end(&job);
        /* contiguous block ranges, balanced by compressed bytes: worker w takes the blocks whose header lies in
         * its share of the stream */
        const int nw = nextra + 1;
        dfan_worker_t workers[HUF_MAX_SESSIONS];
        int created[HUF_MAX_SESSIONS];
        pthread_t th[HUF_MAX_SESSIONS];
        uint64_t b = 0;
        for (int w = 0; w < nw; w++) {
            const uint64_t want = (uint64_t)(((unsigned __int128)length * (unsigned)(w + 1)) / (unsigned)nw);
            uint64_t e = b;
            while (e < nb && (w == nw - 1 || offs[e] < want)) e++;
            workers[w].src = src; workers[w].dst = dst; workers[w].offs = offs; workers[w].outoff = outoff;
            workers[w].b0 = b; workers[w].b1 = e; workers[w].flags = flags;
            workers[w].session = (w == 0) ? mine : extra[w - 1];
            workers[w].own = (w == 0);
            workers[w].d_index = d_index;
            workers[w].rc = HUF_ERROR_SUCCESS;
            created[w] = 0;
            b = e;
        }
        for (int w = 1; w < nw; w++) {
            if (pthread_create(&th[w], NULL, dfan_main, &workers[w]) == 0) created[w] = 1;
            else workers[w].rc = HUF_ERROR_FATAL;
        }
        dfan_main(&workers[0]);
        t_session = mine;
        int ok = workers[0].rc == HUF_ERROR_SUCCESS;
        for (int w = 1; w < nw; w++) {
            if (created[w]) pthread_join(th[w], NULL);
            ok = ok && workers[w].rc == HUF_ERROR_SUCCESS;
        }
        if (!ok) break;                                               /* nothing committed: the ordinary path decides */
        wmem->len += total;
        rmem->off += length;
        g_fanout_decodes.fetch_add(1);
        *result = huf_bufio_read_writer_flush(dec->bufio_writer);
        done = 1;
    } while (0);
    free(offs);
    free(outoff);
    for (int i = 0; i < nextra; i++) session_release_extra(extra[i]);
    return done;
}

/* huf_decode() between two memory streams, the same way: the stream goes to the device segment by segment, is decoded in
 * rounds of HUF_GPU_ROUND_MB compressed bytes cut at block boundaries (decode_rounds_fd's loop: the block loop of
 * src/decoder.c:218 with more check points) and every round's output leaves while the next is decoded.  Returns 1 when
 * it took the call.  ANYTHING unusual - an error in any round, a last block that wants bytes beyond `length`, an output
 * that does not fit - leaves the call to the ordinary path (returns 0 with nothing committed), which reports what the
 * reference reports. */
static int decode_duplex(huf_decoder_t *dec, membuf_t *rmem, membuf_t *wmem, uint32_t flags, huf_error_t *result)
{
    if (!rmem || !wmem || !duplex_enabled()) return 0;
    const uint64_t length = dec->config->length;
    const uint64_t left = rmem->len - rmem->off;
    const uint64_t total = length < left ? length : left;
    /* (the first block's length field stands for the stream's block size: a hint for the rounds' size, nothing else) */
    uint64_t first_len = 0;
    if (total >= 8) memcpy(&first_len, (const char *)*rmem->buf + rmem->off, 8);
    const uint64_t R = dx_round_bytes(total, first_len);
    if (total < DX_MIN_BYTES || total <= R + R / 2) return 0;
    dx_pool_t *P = dx_get();
    if (!P) return 0;
    const uint64_t margin = 4u << 20;                /* what a round may look ahead (its last block's end) */
    const uint64_t out_cap = ((R + margin) * 8 + (1u << 20) + 255u) & ~(uint64_t)255;
    if (grow_dev(&g_stage.d_a, &g_stage.d_a_cap, total + 16) != HUF_ERROR_SUCCESS ||
        grow_dev(&g_stage.d_b, &g_stage.d_b_cap, 2 * out_cap) != HUF_ERROR_SUCCESS ||
        grow_dev(&g_stage.d_c, &g_stage.d_c_cap, R + margin + 16) != HUF_ERROR_SUCCESS) return 0;
    if (!wmem->fixed && mem_reserve(wmem, total + total / 4 + (1u << 20)) != HUF_ERROR_SUCCESS) return 0;
    char *src = (char *)*rmem->buf + rmem->off;
    char *d_out[2] = {(char *)g_stage.d_b, (char *)g_stage.d_b + out_cap};
    void *const reg = dx_register_input(P, src, total);
    const int direct = reg != NULL;
    const uint64_t nseg = (total + R - 1) / R;
    uint64_t pub = 0, waited = 0, loaded = 0, in0 = 0;   /* input segments published / the kernels' stream waits for / their bytes */
    uint64_t pos = 0, out_total = 0, out0 = 0, rounds = 0;
    int ok = 1;
    double t_in = 0, t_out = 0, t_k = 0, t_cp = 0, t_end = 0;
    const double t_start = dx_trace() ? dx_now() : 0.0;
    while (ok && pos < total) {
        const uint64_t round_len = (total - pos < R) ? total - pos : R;
        uint64_t need = pos + round_len + margin;
        if (need > total) need = total;
        uint64_t raw = 0, used = 0;
        for (;;) {
            while (ok && loaded < need) {            /* (at most three segments ahead of what has been asked for) */
                while (pub < nseg && pub < waited + 3) {
                    const uint64_t n = (pub + 1) * R <= total ? R : total - pub * R;
                    const uint64_t id = dx_publish(P, 0, src + pub * R, (char *)g_stage.d_a + pub * R, n, direct);
                    if (pub == 0) in0 = id;
                    pub++;
                }
                huf_error_t we = HUF_ERROR_SUCCESS;
                DX_T(t_in, we = dx_wait_issued(P, in0 + waited));
                if (we != HUF_ERROR_SUCCESS) { ok = 0; break; }
                waited++;
                loaded = waited * R < total ? waited * R : total;
            }
            if (!ok) break;
            const uint64_t avail = (loaded - pos < R + margin) ? loaded - pos : R + margin;   /* (whole segments arrive: more than was asked for) */
            /* the round's bytes at an aligned address (the parallel block discovery wants that); the buffer two rounds back is free */
            int rc = HUF_ERROR_SUCCESS;
            DX_T(t_cp, rc = hufgpu_memcpy_d2d(g_ctx, g_stage.d_c, (const char *)g_stage.d_a + pos, avail));
            if (rc != HUF_ERROR_SUCCESS) { ok = 0; break; }
            if (rounds >= 2) DX_T(t_out, rc = dx_wait_done(P, 1, out0 + rounds - 2));
            if (rc != HUF_ERROR_SUCCESS) { ok = 0; break; }
            DX_T(t_k, rc = hufgpu_decode_stream(g_ctx, g_stage.d_c, avail, round_len, d_out[rounds & 1], out_cap, flags, &raw, &used, NULL));
            /* (a round that already looks at all it may - R + margin bytes - and still wants more holds a block longer than
             *  that: more segments cannot help it, the ordinary path takes the call at once; round 5 loaded every remaining
             *  segment, decoding in vain each time, before it gave up) */
            if (rc == HUF_ERROR_READ_WRITE && loaded < total && avail < R + margin) {          /* the last block wants more of the stream: it is on its way */
                need = loaded + R < total ? loaded + R : total;
                continue;
            }
            if (rc != HUF_ERROR_SUCCESS || used == 0) ok = 0;
            break;
        }
        if (!ok) break;
        if (raw) {
            if (mem_reserve_behind(wmem, P, out_total, raw) != HUF_ERROR_SUCCESS) { ok = 0; break; }
            const uint64_t id = dx_publish(P, 1, (char *)*wmem->buf + wmem->len + out_total, d_out[rounds & 1], raw);
            if (rounds == 0) out0 = id;
            out_total += raw;
            rounds++;
        }
        pos += used;
    }
    DX_T(t_end, dx_drain(P));
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); ok = 0; }           /* (input segments still on their way) */
    dx_unregister_input(reg);
    if (dx_trace())
        fprintf(stderr, "decode_duplex: %llu rounds of %llu MiB, %.2f ms: waiting for input %.2f, for an output buffer %.2f, the round's copy %.2f, kernels (+ their waits) %.2f, the last copies %.2f; ok=%d\n",
                (unsigned long long)rounds, (unsigned long long)(R >> 20), (dx_now() - t_start) * 1e3, t_in * 1e3, t_out * 1e3, t_cp * 1e3, t_k * 1e3, t_end * 1e3, ok);
    if (!ok || P->err) return 0;                                                               /* nothing committed: the ordinary path takes the call */
    rmem->off += pos;
    wmem->len += out_total;
    *result = huf_bufio_read_writer_flush(dec->bufio_writer);
    return 1;
}

static huf_error_t decode_locked(huf_decoder_t *dec, uint64_t *pieces)
{
    const uint64_t length = dec->config->length;
    TRY(session_acquire());
    const uint32_t flags = relaxed_tree() ? HUFGPU_RELAXED_TREE : HUFGPU_STRICT_TREE;
    if (!pieces && zero_copy_enabled() && own_fd_of(dec->config->reader, 0) >= 0) {
        membuf_t *wm = own_memstream_writer(dec->config->writer);
        return decode_from_fd(dec, own_fd_of(dec->config->reader, 0), wm, wm ? -1 : own_fd_of(dec->config->writer, 1), flags);
    }

    /* The reference pulls bytes on demand and may run past `length` to finish the last block
     * (src/decoder.c:218); here: take `length` bytes, and if the device reports that a block
     * needs more input, ask the reader for more and decode again. */
    membuf_t *rmem = zero_copy_enabled() ? own_memstream_reader(dec->config->reader) : NULL;
    membuf_t *wmem = zero_copy_enabled() ? own_memstream_writer(dec->config->writer) : NULL;
    if (!pieces) {
        huf_error_t fan = HUF_ERROR_SUCCESS;
        if (decode_fanout(dec, rmem, wmem, flags, &fan) || decode_duplex(dec, rmem, wmem, flags, &fan)) return fan;
    }
    /* a small call between two memory streams: one synchronisation instead of three (hufgpu_decode_small).  Anything but
     * a clean decode - an error, a last block that wants bytes behind `length` - goes on below as if nothing had happened */
    if (!pieces && rmem && wmem && length <= SMALL_DECODE_BYTES && rmem->len - rmem->off >= length) {
        const uint64_t out_cap = (uint64_t)length * 8 + 64;
        const uint64_t h_need = ((out_cap + 7u) & ~7ull) + 64u;
        if (grow_host(&g_stage.h_a, &g_stage.h_a_cap, (size_t)length) == HUF_ERROR_SUCCESS &&
            grow_host(&g_stage.h_b, &g_stage.h_b_cap, (size_t)h_need) == HUF_ERROR_SUCCESS &&
            grow_dev(&g_stage.d_a, &g_stage.d_a_cap, (size_t)length + 16) == HUF_ERROR_SUCCESS &&
            grow_dev(&g_stage.d_b, &g_stage.d_b_cap, (size_t)out_cap) == HUF_ERROR_SUCCESS) {
            memcpy(g_stage.h_a, (const char *)*rmem->buf + rmem->off, (size_t)length);
            uint64_t raw = 0, used = 0;
            const int rc = hufgpu_decode_small(g_ctx, g_stage.h_a, length, length, flags, g_stage.d_a, g_stage.d_b, out_cap,
                                               g_stage.h_b, g_stage.h_b_cap, &raw, &used);
            if (rc == HUF_ERROR_FATAL) return HUF_ERROR_FATAL;
            if (rc == HUF_ERROR_SUCCESS) {
                rmem->off += (size_t)(used < length ? used : length);
                if (raw) TRY(memwrite(wmem, g_stage.h_b, (size_t)raw));
                return huf_bufio_read_writer_flush(dec->bufio_writer);
            }
        }
    }
    size_t avail = 0;
    const char *in_ptr = NULL;              /* host bytes [0, avail) of the input */
    const size_t start_off = rmem ? rmem->off : 0;
    if (rmem) {
        const size_t left = rmem->len - rmem->off;
        in_ptr = (const char *)*rmem->buf + rmem->off;
        avail = (size_t)length < left ? (size_t)length : left;
        rmem->off += avail;
    } else {
        size_t cap_in = (size_t)length + 4096;
        TRY(grow_host(&g_stage.h_a, &g_stage.h_a_cap, cap_in));
        TRY(read_upto(dec->config->reader, (uint8_t *)g_stage.h_a, (size_t)length, &avail));
    }

    uint64_t out_cap = (uint64_t)avail * 8 + (1u << 20);
    for (;;) {
        TRY(grow_dev(&g_stage.d_a, &g_stage.d_a_cap, avail + 16));
        TRY(grow_dev(&g_stage.d_b, &g_stage.d_b_cap, out_cap));
        uint64_t raw = 0, used = 0;
        int rc = rmem ? (int)lane_copy(1, g_stage.d_a, (void *)in_ptr, avail) : hufgpu_memcpy_h2d(g_ctx, g_stage.d_a, g_stage.h_a, avail);
        if (rc == HUF_ERROR_SUCCESS)
            rc = hufgpu_decode_stream(g_ctx, g_stage.d_a, avail, length, g_stage.d_b, g_stage.d_b_cap, flags, &raw, &used, NULL);
        if (rc == HUF_ERROR_FATAL) return (huf_error_t)rc;
        if (rc == HUF_ERROR_MEMORY_ALLOCATION && out_cap < ((uint64_t)1 << 40)) {   /* output did not fit: enlarge */
            out_cap *= 4;
            continue;
        }
        if (pieces) {
            /* a piece of a stream: what counts is the last block boundary inside it */
            uint64_t good_raw = 0, good_used = 0;
            TRY(hufgpu_decode_stream_complete(g_ctx, &good_raw, &good_used));
            if (rc == HUF_ERROR_READ_WRITE) { rc = HUF_ERROR_SUCCESS; raw = good_raw; used = good_used; }
            *pieces = (rc == HUF_ERROR_SUCCESS) ? used : good_used;
            if (rc != HUF_ERROR_SUCCESS) raw = good_raw;          /* the blocks in front of a damaged one */
        } else
        if (rc == HUF_ERROR_READ_WRITE && rmem) {  /* maybe the stream holds more than `length` */
            const size_t more_want = avail < 65536 ? 65536 : avail;
            const size_t left = rmem->len - rmem->off;
            const size_t more = more_want < left ? more_want : left;
            if (more) { rmem->off += more; avail += more; continue; }
        } else if (rc == HUF_ERROR_READ_WRITE) {   /* maybe the reader has more than `length` */
            size_t more_want = avail < 65536 ? 65536 : avail;
            if (g_stage.h_a_cap < avail + more_want) {
                void *bigger = NULL; size_t bigger_cap = 0;
                TRY(grow_host(&bigger, &bigger_cap, avail + more_want));
                memcpy(bigger, g_stage.h_a, avail);
                (void)hipHostFree(g_stage.h_a);
                g_stage.h_a = bigger; g_stage.h_a_cap = bigger_cap;
            }
            size_t more = 0;
            TRY(read_upto(dec->config->reader, (uint8_t *)g_stage.h_a + avail, more_want, &more));
            if (more) { avail += more; continue; }
        }
        /* the reference's unbuffered reader stops right behind the last block it took (src/decoder.c:
         * 218-276 pulls bytes on demand): the bytes looked at speculatively beyond that stay unread, so
         * that a caller can decode consecutive streams from one memstream */
        if (rmem) rmem->off = start_off + (size_t)(used < avail ? used : avail);
        /* bytes of the blocks that decoded completely are delivered even when a later block
         * fails, as the reference's unbuffered writer would have done */
        if (raw && wmem) {
            TRY(d2h_to_memstream(wmem, g_stage.d_b, raw));
        } else if (raw) {
            TRY(grow_host(&g_stage.h_b, &g_stage.h_b_cap, raw));
            TRY(hufgpu_memcpy_d2h(g_ctx, g_stage.h_b, g_stage.d_b, raw));
            TRY(huf_bufio_write(dec->bufio_writer, g_stage.h_b, raw));
        }
        if (rc != HUF_ERROR_SUCCESS) return (huf_error_t)rc;   /* no flush on the error path (decoder.c:278-286) */
        return huf_bufio_read_writer_flush(dec->bufio_writer);
    }
}

huf_error_t huf_decode(const huf_config_t *config)
{
    GUARD(config);
    huf_decoder_t *dec = NULL;
    TRY(huf_decoder_init(&dec, config));
    huf_error_t err = HUF_ERROR_SUCCESS;
    if (dec->config->length) {                    /* test/decode_test.c:32-36: empty input is fine */
        session_enter();
        err = decode_locked(dec, NULL);
        session_leave();
    }
    huf_decoder_free(&dec);
    return err;
}

int huf_gpu_decode_blocks(const huf_config_t *config, uint64_t *consumed)
{
    GUARD(config);
    GUARD(consumed);
    *consumed = 0;
    huf_decoder_t *dec = NULL;
    TRY(huf_decoder_init(&dec, config));
    huf_error_t err = HUF_ERROR_SUCCESS;
    if (dec->config->length) {
        session_enter();
        err = decode_locked(dec, consumed);
        session_leave();
    }
    huf_decoder_free(&dec);
    return err;
}

/* host -> host, for a binding that has to hand the result over as an object of its own (the
 * Python layer's `bytes`): the destination is fresh memory, so a plain memcpy runs at page-fault
 * speed (256 MiB: 35-40 ms).  Huge-page advice, then every thread makes its part present and copies it. */
typedef struct { char *dst; const char *src; size_t n; } copy_part_t;
static void *copy_part_main(void *arg)
{
    copy_part_t *c = (copy_part_t *)arg;
    prefault_t w = {c->dst, c->n};
    prefault_main(&w);
    memcpy(c->dst, c->src, c->n);
    return NULL;
}

int huf_gpu_copy_out(void *dst, const void *src, size_t n)
{
    if ((!dst || !src) && n) return HUF_ERROR_INVALID_ARGUMENT;
    const int nthreads = prefault_threads();
    if (n < PREFAULT_MIN || nthreads <= 1) {
        if (n) memcpy(dst, src, n);
        return HUF_ERROR_SUCCESS;
    }
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = ((uintptr_t)dst + page - 1) & ~(page - 1), hi = ((uintptr_t)dst + n) & ~(page - 1);
    if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
    copy_part_t part[16];
    pthread_t th[16];
    int started = 0;
    const size_t piece = ((n / (size_t)nthreads) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    for (int i = 0; i < nthreads; i++) {
        const size_t off = (size_t)i * piece;
        if (off >= n) break;
        part[i].dst = (char *)dst + off;
        part[i].src = (const char *)src + off;
        part[i].n = (n - off < piece || i == nthreads - 1) ? n - off : piece;        /* the last part takes the rest */
        if (i == 0) continue;
        if (pthread_create(&th[i], NULL, copy_part_main, &part[i]) == 0) started |= 1 << i;
        else copy_part_main(&part[i]);
    }
    copy_part_main(&part[0]);
    for (int i = 1; i < 16; i++)
        if (started & (1 << i)) pthread_join(th[i], NULL);
    return HUF_ERROR_SUCCESS;
}

/* sessions that hold a device context right now, and how many the device list allows */
/* how many huf_encode() / huf_decode() calls of this process were spread over several sessions */
int huf_gpu_fanouts(int *encodes, int *decodes)
{
    if (encodes) *encodes = g_fanout_encodes.load();
    if (decodes) *decodes = g_fanout_decodes.load();
    return g_fanout_encodes.load() + g_fanout_decodes.load();
}

int huf_gpu_sessions(int *configured)
{
    pthread_mutex_lock(&g_pool_lock);
    session_pool_init();
    int live = 0;
    for (int i = 0; i < g_nsessions; i++) live += g_sessions[i].ctx != NULL;
    if (configured) *configured = g_nsessions;
    pthread_mutex_unlock(&g_pool_lock);
    return live;
}

}  /* extern "C" */
