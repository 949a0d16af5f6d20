/*
 * huffman_gpu.h - device-resident entry points of the MI355X Huffman block codec.
 *
 * The reference has one boundary for the hot path: huf_encode()/huf_decode() over callback
 * streams (include/huffman/encoder.h:25-26, decoder.h:25-26).  Those are served by this
 * library too (include/huffman.h).  The functions below are the same per-block hot path with
 * the callback streams peeled off: plain pointers into HBM and sizes, no host copies.  They
 * are what huf_encode()/huf_decode() call internally after staging a batch, what the Python
 * layer calls when the data already lives on the GPU, and what bench.py times.
 *
 * C ABI only: no C++ or torch types; `stream` arguments are a hipStream_t passed as void*
 * (NULL = the device's default stream).  All functions return a huf_error_t value
 * (include/huffman.h); HIP failures map to HUF_ERROR_FATAL and hufgpu_last_error() carries
 * the text.  Nothing here falls back to the CPU.
 *
 * Replaces, per block (reference file:line):
 *   hufgpu_histogram      src/histogram.c:73-103   huf_histogram_populate (iota = 1)
 *   hufgpu_encode         src/encoder.c:288-374    block loop: histogram, tree.c:292-427 tree,
 *                                                   encoder.c:40-81 codes, tree.c:233-289
 *                                                   serialize, encoder.c:85-131 bit-pack
 *   hufgpu_decode         src/decoder.c:218-276    header parse, tree.c:138-227 deserialize,
 *                                                   decoder.c:34-96 tree walk
 */
#ifndef INCLUDE_huffman_gpu_h__
#define INCLUDE_huffman_gpu_h__

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-device workspace (side tables, ticket counters, profiling events).  Not thread-safe, and
 * its calls share that workspace: enqueue them on ONE stream at a time (or wait for the stream
 * before switching to another); use one context per concurrently used stream. */
typedef struct hufgpu_ctx hufgpu_ctx_t;

/* Decode flags. */
#define HUFGPU_STRICT_TREE  0u  /* tree_len > 1024 -> HUF_ERROR_BTREE_OVERFLOW (reference parity,
                                   src/decoder.c:237-239) */
#define HUFGPU_RELAXED_TREE 1u  /* accept the 1025-entry tree the encoder itself emits for blocks
                                   with all 256 byte values */

#define HUFGPU_SEQUENTIAL   2u  /* hufgpu_decode_stream only: skip the parallel block discovery and take the
                                   blocks strictly in order (diagnostics; results are identical) */

/* Largest block the kernels take (bytes).  Larger blocks -> HUF_ERROR_INVALID_ARGUMENT.  (Codes are
 * kept in 56 bits, which any block below F(57) = 3.6e11 bytes satisfies; the limit is what still
 * fits a device together with its stream and its output.)  hufgpu_histogram() returns 32-bit counts
 * and takes blocks below 2^32 bytes only. */
#define HUFGPU_MAX_BLOCK ((uint64_t)1 << 38)

/* Number of usable gfx950 devices; 0 when HIP is unusable (never an error by itself). */
int hufgpu_device_count(void);

/* Create/destroy a context bound to `device`. HUF_ERROR_FATAL when there is no such GPU. */
int hufgpu_ctx_create(hufgpu_ctx_t **ctx, int device);
int hufgpu_ctx_destroy(hufgpu_ctx_t *ctx);
int hufgpu_ctx_device(const hufgpu_ctx_t *ctx);      /* the device ordinal the context was made for (-1: no context) */

/* Text of the last failure on this context (or of the last global failure if ctx == NULL). */
const char *hufgpu_last_error(const hufgpu_ctx_t *ctx);

/* Number of blocks huf_encode produces for n input bytes (src/encoder.c:288-293). */
uint64_t hufgpu_block_count(uint64_t n, uint64_t blocksize);

/* Capacity (bytes) that always holds the encoded stream of n bytes. */
uint64_t hufgpu_encode_bound(uint64_t n, uint64_t blocksize);

/* 256-bin byte histogram of each block: d_hist[block * 256 + byte] (uint32). */
int hufgpu_histogram(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                     uint32_t *d_hist, void *stream);

/*
 * Encode n bytes at d_in into the libhuffman block stream at d_out (both in HBM).
 *   d_block_offsets : optional, nblocks+1 uint64 in HBM; receives the byte offset of every
 *                     block header in d_out, the last entry being the stream length.  This is
 *                     the in-process block index (the wire format itself stores no payload
 *                     length - SURVEY §0 fact 1).
 *   out_len         : optional host pointer; when given the call synchronises and stores the
 *                     stream length.  When NULL the call only enqueues work on the stream.
 */
int hufgpu_encode(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                  void *d_out, uint64_t out_cap, uint64_t *d_block_offsets,
                  uint64_t *out_len, void *stream);

/*
 * Decode a block stream whose block index is known (d_block_offsets from hufgpu_encode).
 *   raw_len  : optional host pointer; when given the call synchronises, stores the number of
 *              bytes delivered in d_out and returns the first error in stream order
 *              (HUF_ERROR_BTREE_OVERFLOW / _CORRUPTED / _READ_WRITE like src/decoder.c).  After an
 *              error these are the blocks in front of the failing one AND the symbols of the failing
 *              block that src/decoder.c:69-91 delivers before it stops (the block is decoded once
 *              more, in order, with its record as the whole input).
 *              When NULL the call only enqueues; fetch the result with hufgpu_decode_result().
 */
int hufgpu_decode(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t stream_len,
                  const uint64_t *d_block_offsets, uint64_t nblocks,
                  void *d_out, uint64_t out_cap, uint32_t flags,
                  uint64_t *raw_len, void *stream);

/*
 * The same pair with the encoder's SUB-INDEX: besides the block index the encoder can hand over
 * where, inside every block's payload, each group of 32 symbols starts (2 bytes per 32 symbols +
 * 8 bytes per 2 048 symbols + the 256 code lengths of every block; hufgpu_sub_index_bytes() bytes, 8-byte aligned, in HBM).  Like the
 * block index it is in-process side information - the stream is the reference's, byte for byte.
 * With it hufgpu_decode_sub() decodes every symbol once instead of finding the codeword starts by
 * decoding speculatively (src/decoder.c:34-96 has the same information implicitly: it walks the
 * bits in order).  The sub-index is VERIFIED while it is used: a group must decode to exactly its
 * recorded bit count with every walk inside the tree and inside the payload; a block for which
 * that fails (stale or foreign sub-index, damaged stream) is decoded again without it, so results
 * and error codes are those of hufgpu_decode() for ANY content of d_sub_index.
 *   raw_size, blocksize : the n and blocksize of the encode that produced stream and sub-index
 *                         (they fix the block count and the layout of d_sub_index).
 */
uint64_t hufgpu_sub_index_bytes(uint64_t n, uint64_t blocksize);
int hufgpu_encode_sub(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                      void *d_out, uint64_t out_cap, uint64_t *d_block_offsets,
                      void *d_sub_index, uint64_t *out_len, void *stream);
int hufgpu_decode_sub(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t stream_len,
                      const uint64_t *d_block_offsets, uint64_t raw_size, uint64_t blocksize,
                      const void *d_sub_index, void *d_out, uint64_t out_cap, uint32_t flags,
                      uint64_t *raw_len, void *stream);

/*
 * One logical input over the GPUs of a node: RCCL scatter / gather of block buffers.
 * Blocks are independent (src/encoder.c:288-374 resets all state between blocks), so rank r of G owns a contiguous range
 * of ceil(nblocks / G) blocks (hufgpu_shard_range) and the codec needs no collective; the data starts and ends on ONE
 * rank, the root.  Every rank of the communicator makes the same call with the same n_total, blocksize, root and flags;
 * the pointers count on the root only (NULL elsewhere).  Each movement is one group of ncclSend / ncclRecv of exactly-sized
 * buffers to computed offsets (RCCL has no scatterv / gatherv) on the object's own stream, plus all-gathers of one or two
 * words a rank: who is ready, every shard's compressed size (rank order = stream order: shard r starts at the sum of the
 * sizes in front of it), every rank's decode result.  The gathered stream is the reference's, byte for byte - the one a
 * single hufgpu_encode() of the whole input writes.  RCCL is looked up with dlopen at the first call (HUF_GPU_RCCL_LIB,
 * else librccl.so.1): without it these entry points return HUF_ERROR_FATAL and hufgpu_shard_last_error(NULL) says why.
 * Every call has a deadline (HUF_GPU_SHARD_TIMEOUT_MS when the object is made, default 120 000 ms; hufgpu_shard_set_timeout;
 * 0 = none): a rank that never arrives, or a communicator that reports an asynchronous error, makes the call return
 * HUF_ERROR_FATAL on the ranks that did arrive - the object's own communicator is aborted (ncclCommAbort), the object
 * is broken and every later call on it fails at once; hufgpu_shard_destroy() is still to be called.
 *
 *   hufgpu_shard_create   : nccl_comm = an existing ncclComm_t of the ranks (not destroyed with the object; nranks and rank
 *                           are the communicator's), or NULL: then the object makes its own from `id`
 *                           (HUFGPU_SHARD_ID_BYTES from hufgpu_shard_unique_id() on one rank, handed to all by the caller),
 *                           nranks and rank.  ctx = this rank's context: its device is the rank's GPU, and no other work
 *                           may be in flight on it during a sharded call.
 *   hufgpu_encode_sharded : d_in (root: n_total bytes) -> d_stream (root: room for hufgpu_encode_bound(n_total, blocksize)
 *                           bytes); *stream_len and shard_lens[nranks] (host, optional) are filled on EVERY rank.
 *                           HUFGPU_SHARD_INDEX: d_block_offsets (root: hufgpu_block_count(n_total, blocksize) + 1 words)
 *                           receives the block index of the whole stream.
 *   hufgpu_decode_sharded : d_stream + d_block_offsets (root) -> d_out (root: n_total bytes).  The stream is cut into
 *                           nranks shares of about equal compressed BYTES at block borders (hufgpu_shard_plan_decode);
 *                           a rank decodes its blocks from the block index alone.  HUFGPU_SHARD_OWN_LAYOUT: the stream is
 *                           the one this object's last hufgpu_encode_sharded (same root, n_total, blocksize) produced -
 *                           the shards are cut as they were then and every rank uses the block index and the sub-index
 *                           it kept (d_block_offsets is not read).  Errors: the first failing rank's in stream order,
 *                           returned on every rank; *raw_len = bytes in front of it (as hufgpu_decode()).
 *   legs_ms               : optional, 4 doubles: host milliseconds of the call's four legs (scatter, codec, control words,
 *                           gather; decode: plan, scatter, codec + results, gather), each synchronised.
 */
typedef struct hufgpu_shard hufgpu_shard_t;
#define HUFGPU_SHARD_ID_BYTES   128      /* sizeof(ncclUniqueId) */
#define HUFGPU_SHARD_INDEX      0x100u
#define HUFGPU_SHARD_OWN_LAYOUT 0x200u
int hufgpu_shard_unique_id(void *id);
int hufgpu_shard_create(hufgpu_shard_t **sh, hufgpu_ctx_t *ctx, void *nccl_comm, const void *id, int nranks, int rank);
int hufgpu_shard_destroy(hufgpu_shard_t *sh);
int hufgpu_shard_info(const hufgpu_shard_t *sh, int *nranks, int *rank);
const char *hufgpu_shard_last_error(const hufgpu_shard_t *sh);
int hufgpu_shard_set_timeout(hufgpu_shard_t *sh, uint32_t timeout_ms);
int hufgpu_shard_range(uint64_t n_total, uint64_t blocksize, int rank, int nranks, uint64_t *lo, uint64_t *hi);
/* first_block[nranks + 1] from a host copy of the block index (nblocks + 1 offsets, the last = the stream's length) */
int hufgpu_shard_plan_decode(const uint64_t *block_offsets, uint64_t nblocks, int nranks, uint64_t *first_block);
int hufgpu_encode_sharded(hufgpu_shard_t *sh, int root, const void *d_in, uint64_t n_total, uint64_t blocksize,
                          uint32_t flags, void *d_stream, uint64_t stream_cap, uint64_t *d_block_offsets,
                          uint64_t *stream_len, uint64_t *shard_lens, double *legs_ms);
int hufgpu_decode_sharded(hufgpu_shard_t *sh, int root, const void *d_stream, uint64_t stream_len,
                          const uint64_t *d_block_offsets, uint64_t n_total, uint64_t blocksize, uint32_t flags,
                          void *d_out, uint64_t out_cap, uint64_t *raw_len, double *legs_ms);

/* A small encode with one synchronisation instead of three: h_in_pinned (n bytes, pinned host memory) -> d_in ->
 * encode -> h_out_pinned: the stream's first hufgpu_encode_bound(n, blocksize) bytes and, 8-byte aligned behind them,
 * its length (h_out_cap >= bound rounded up to 8, + 8).  What huf_encode() uses for memory streams of up to 32 KiB. */
int hufgpu_encode_small(hufgpu_ctx_t *ctx, const void *h_in_pinned, uint64_t n, uint64_t blocksize, void *d_in,
                        void *d_out, uint64_t out_cap, void *h_out_pinned, uint64_t h_out_cap, uint64_t *out_len);

/* The decode's twin of hufgpu_encode_small(): a raw stream of `avail` bytes in PINNED host memory (of which the reference's
 * loop takes blocks while fewer than `length` bytes are consumed, src/decoder.c:218), decoded in order by one workgroup,
 * output and outcome back in pinned host memory, one synchronisation.  d_in: avail bytes of device memory; d_out /
 * out_cap: the device output buffer; h_out_pinned / h_out_cap: min(out_cap, 8 avail + 64) rounded up to 8, + 48 bytes.
 * Returns what hufgpu_decode_stream() returns for the same stream; *raw_len bytes at h_out_pinned are the result. */
int hufgpu_decode_small(hufgpu_ctx_t *ctx, const void *h_in_pinned, uint64_t avail, uint64_t length, uint32_t flags,
                        void *d_in, void *d_out, uint64_t out_cap, void *h_out_pinned, uint64_t h_out_cap,
                        uint64_t *raw_len, uint64_t *consumed);

/* Of the last enqueued hufgpu_decode() / hufgpu_decode_sub(): blocks that went through a slower decoder -
 * counters[0] = decoded again by the exact in-order-equivalent decoder (a damaged block, an unusual tree, a stale
 * sub-index), counters[1] = 0 (reserved: it counted the blocks round 4's one-pass decoder handed on).
 * Results never depend on these; they say what a slow decode was slow for.  Synchronises the stream. */
int hufgpu_decode_counters(hufgpu_ctx_t *ctx, uint32_t *counters);

/* Synchronise and report the outcome of the last enqueued hufgpu_decode() / hufgpu_decode_sub().
 * LIFETIME: d_stream, d_block_offsets and d_out of the enqueued call must stay valid until this returns - when a block
 * failed, this call reads the stream and the index once more and WRITES the failing block's symbols in front of the
 * failure to d_out (what src/decoder.c:69-91 delivers).  After it has returned the library holds none of them. */
int hufgpu_decode_result(hufgpu_ctx_t *ctx, uint64_t *raw_len);

/*
 * Decode a raw stream (no index): `avail` bytes are readable at d_stream, `length` compressed
 * bytes drive the block loop exactly like config->length in src/decoder.c:218.  Block
 * boundaries are discovered on the device: every byte offset is tested for a valid header,
 * every candidate is decoded to find its end (straight into d_out when all candidates together
 * fit it), the chain from offset 0 is followed, blocks that are not in place yet are decoded in
 * parallel and anything else (errors, odd headers, tails) goes to an exact in-order decoder -
 * results, errors and byte counts are those of the reference in every case.  Synchronous.
 * *raw_len bytes of d_out are the result; what lies behind them in d_out is unspecified.
 */
int hufgpu_decode_stream(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length,
                         void *d_out, uint64_t out_cap, uint32_t flags, uint64_t *raw_len,
                         uint64_t *consumed, void *stream);

/*
 * The block index of a raw stream (what hufgpu_decode() wants) without decoding the stream into an output
 * buffer: every candidate header is probed count-only and the chain from offset 0 is walked
 * (kernels/discover.hpp).  *d_index = device array of *nblocks + 1 header offsets owned by the context (valid
 * until its next decode call), the last one = *consumed, the stream offset behind the validated blocks.
 * *nblocks = 0: nothing could be validated (a damaged or tiny stream - hufgpu_decode_stream() reports what is
 * wrong with it).  d_stream must be 16-byte aligned.  For callers that spread the decode of ONE stream over
 * several devices (huf_decode() with HUF_GPU_DEVICES, src/decoder.c:218-276: blocks are independent once found).
 */
int hufgpu_block_index(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length, uint32_t flags,
                       const uint64_t **d_index, uint64_t *nblocks, uint64_t *consumed, void *stream);

/* Of the last hufgpu_decode_stream() call: the stream bytes and output bytes of the blocks that
 * decoded COMPLETELY (on success all of them; after an error the position in front of the failing
 * block - what a caller that feeds a stream piecewise keeps for its next piece). */
int hufgpu_decode_stream_complete(hufgpu_ctx_t *ctx, uint64_t *raw_len, uint64_t *consumed);

/* Deterministic synthetic inputs of SURVEY §8d, generated in HBM (kind: 0 const41,
 * 1 uniform256, 2 uniform255, 3 zipf255). `first` = index of the first byte of this shard in
 * the global sequence, so shards of one logical input can be produced on different GPUs. */
int hufgpu_fill(hufgpu_ctx_t *ctx, void *d_out, uint64_t n, int kind, uint64_t seed,
                uint64_t first, void *stream);

/* Plain device memory helpers so that C callers need not link HIP themselves. */
int hufgpu_malloc(hufgpu_ctx_t *ctx, void **d_ptr, uint64_t bytes);
int hufgpu_free(hufgpu_ctx_t *ctx, void *d_ptr);
int hufgpu_memcpy_h2d(hufgpu_ctx_t *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int hufgpu_memcpy_d2h(hufgpu_ctx_t *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int hufgpu_memcpy_d2d(hufgpu_ctx_t *ctx, void *d_dst, const void *d_src, uint64_t bytes);
int hufgpu_synchronize(hufgpu_ctx_t *ctx);

/* Bandwidth calibration: ONE launch of a hand-written kernel that only moves bytes, 16 bytes per lane and access
 * (kernels/fill.hpp) - kind 0: copy d_a -> d_b, 1: read d_a, 2: fill d_b; variant 0 .. HUFGPU_CALIB_VARIANTS - 1 =
 * workgroup shape and cache policy.  bytes: a multiple of 64 KiB; pointers 16-byte aligned.  bench.py times these
 * for the ceilings it prints beside the roofline (what a kernel that reads N and writes N can reach on this part). */
#define HUFGPU_CALIB_VARIANTS 8
int hufgpu_calib_bandwidth(hufgpu_ctx_t *ctx, int kind, int variant, const void *d_a, void *d_b, uint64_t bytes, void *stream);

/* Per-kernel timing. While enabled, every hufgpu_encode/hufgpu_decode call records HIP
 * events around each of its kernels on the stream it launches on (up to 256 calls are kept).
 * enabled: 1 = start a new record, 0 = pause (the record is kept), 2 = resume the record - so a
 * timed loop can sample every n-th call (an event costs ~5 us).  hufgpu_get_profile() returns, for
 * kind 0 = encode or 1 = decode, the per-stage time summed over the recorded calls, in launch order:
 *   encode: [hist256, tree, scan_sizes, pack]     decode: [prepare, decode]
 * (blocks shorter than 4 MiB run hist256 + tree + scan_sizes as ONE kernel: its time is stage 0,
 * stages 1 and 2 are empty).  No host synchronisation happens until hufgpu_get_profile(). */
int hufgpu_set_profiling(hufgpu_ctx_t *ctx, int enabled);
int hufgpu_get_profile(hufgpu_ctx_t *ctx, int kind, float *ms_sum, int max_stages,
                       int *n_stages, int *n_calls);

/* ---- extensions of the host API of include/huffman.h (not in the reference) ----
 * huf_gpu_set_relaxed_tree: 1 = huf_decode() accepts the 1025-entry trees of blocks that use all
 * 256 byte values (the reference's encoder writes them, its decoder returns error 5).
 * huf_gpu_memwrap: a read-only huf_read_writer_t over `length` bytes the caller already holds
 * (no copy into a huf_memopen() buffer); huf_encode()/huf_decode() send such a stream to the
 * device directly.  Close it with huf_memclose(); the bytes are never written or freed. */
struct __huf_read_writer;
struct __huf_encoder_config;
void huf_gpu_set_relaxed_tree(int enabled);
int huf_gpu_memwrap(struct __huf_read_writer **self, const void *data, size_t length);
/* huf_gpu_memwrap_out: a WRITER over `capacity` bytes of memory the caller provides (the buffer that is to hold
 * the result): huf_encode()/huf_decode() write there directly; a write that does not fit fails with
 * HUF_ERROR_MEMORY_ALLOCATION (the memory is never grown, moved or freed).  huf_memlen() = bytes written. */
int huf_gpu_memwrap_out(struct __huf_read_writer **self, void *buffer, size_t capacity);
/* huf_gpu_decode_blocks: huf_decode() for a caller that holds only a PIECE of a stream (a file read
 * in bounded rounds, src/decoder.c:218 has the whole stream behind its reader): the blocks that lie
 * completely inside config->length bytes are decoded and written, *consumed = their stream bytes.
 * A last block that is cut off is not an error here - it is left for the next call (*consumed <
 * config->length; 0 when not even one block is complete); every other error is huf_decode()'s.
 * The reader is never asked for more than config->length bytes. */
int huf_gpu_decode_blocks(const struct __huf_encoder_config *config, uint64_t *consumed);

/* huf_gpu_sessions: huf_encode()/huf_decode() calls hold one SESSION (a device context and its staging
 * buffers) each.  The environment decides how many there are: HUF_GPU_DEVICE=k (default 0) = one
 * session on device k, concurrent calls take turns; HUF_GPU_DEVICES="0,1,2" or "all" = one session per
 * listed device ("0,0" = two on device 0), and concurrent calls - disjoint configs on different
 * threads, parallel in the reference (src/encoder.c:379-392 has no global state) - run side by side
 * on different sessions, so a multi-threaded caller uses every listed GPU.  Returns the sessions that
 * hold a context so far; *configured = the length of the list. */
int huf_gpu_sessions(int *configured);

/* With several sessions configured and free, ONE huf_encode() / huf_decode() between memory streams is spread
 * over them: the encoder deals out rounds of whole blocks, the decoder finds the stream's blocks on one device
 * (hufgpu_block_index) and deals out block ranges balanced by compressed bytes (src/decoder.c:218-276: blocks
 * are independent once found).  Counts of the calls of this process that went that way; returns their sum. */
int huf_gpu_fanouts(int *encodes, int *decodes);

/* huf_gpu_copy_out: memcpy for a binding that must hand a result over as an object of its own (the
 * Python layer's `bytes`): `dst` is fresh memory, where a plain memcpy runs at page-fault speed.
 * Huge-page advice for the destination, then a few threads make their parts present and copy them. */
int huf_gpu_copy_out(void *dst, const void *src, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* INCLUDE_huffman_gpu_h__ */
