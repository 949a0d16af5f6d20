/*
 * hufgpu_api.hip - host side of the device-resident C ABI (include/huffman_gpu.h).
 *
 * Owns the per-device context (stream, workspace in HBM, pinned result words) and launches the
 * kernels of hufgpu_kernels.hip.  No CPU implementation of the codec lives here: if HIP or a
 * gfx950 device is unavailable every entry point fails with HUF_ERROR_FATAL and says why.
 */
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/huffman_gpu.h"
#include "hufgpu_common.h"
#include "hufgpu_kernels.hip"

using namespace hufgpu;

#ifndef HIST_THREADS
#define HIST_THREADS 256
#endif
#define PACK_THREADS 256
#ifndef DSUB_THREADS
#define DSUB_THREADS 256      /* decode_sub_kernel: four waves a workgroup, eight wave tiles a wave - a workgroup's table build is paid once per 64 KiB */
#endif
#ifndef DEC_THREADS
#define DEC_THREADS 512
#endif
#define SCAN_THREADS 1024
#define MAX_STAGES 8
#define PROF_SLOTS 256
#define PROF_ENCODE 0
#define PROF_DECODE 1

struct hufgpu_ctx {
    int device;
    hipStream_t stream;
    char err[512];

    /* encode workspace, sized for ws_blocks blocks */
    uint64_t ws_blocks;
    uint32_t *d_hist;
    hufcode_t *d_codetab;
    int16_t *d_treebuf;
    HufBlockMeta *d_meta;
    uint64_t *d_offsets;          /* used when the caller passes no index buffer */
    TwoLevel enc_sizes;           /* two-level prefix sums of the encoded block sizes */
    uint64_t ws_chunks;           /* blocks >= HUF_BIG_BLOCK: per-chunk counts, payload bits and first bits */
    uint32_t *d_chunk_hist;
    uint64_t *d_chunk_tot, *d_chunk_bits;

    /* decode workspace */
    uint64_t dws_blocks;
    HufDecodeMeta *d_dmeta;
    uint64_t *d_out_offsets;
    int32_t *d_status;
    TwoLevel dec_lens;            /* two-level prefix sums of the block lengths */
    uint32_t *d_fix_count;        /* decode_sub_kernel: blocks its sub-index could not verify */
    uint32_t *d_fix_blocks;
    uint32_t *d_fix_flag;

    /* raw-stream discovery workspace */
    uint64_t disc_wgs, disc_cands;
    uint32_t *d_wg_counts;
    void *d_disc_slots;           /* DISC_SLOTS candidates a discovery workgroup (kernels/discover.hpp, DiscSlot) */
    uint64_t *d_disc_masks;       /* 64 header-test verdicts per discovery thread */
    uint64_t *d_wg_base;
    uint64_t *d_cand, *d_cand_end, *d_chain;
    int32_t *d_cand_status;
    uint32_t *d_nxt;
    uint64_t *d_walk;             /* 5 result words of walk_kernel */
    uint64_t *d_spec_off;         /* speculative output offsets of the candidates (disc_cands + 1) */

    /* blocks of many MiB in a raw stream: the sub-index built for them (kernels/spec_index.hpp) */
    uint64_t big_lanes, big_sub_bytes;
    uint64_t *d_big_entry, *d_big_exit, *d_big_pre, *d_big_wgpre, *d_big_wgscratch, *d_big_first_pos, *d_big_first_g, *d_big_last_pos;
    uint32_t *d_big_cnt;
    void *d_big_sub;
    uint64_t *d_big_offs;         /* SPEC_WORDS status words, then the two-entry block index */

    uint64_t *d_result;           /* 8 words: err, raw_len, failing block / consumed, blocks, complete consumed, complete raw */
    uint64_t complete_used, complete_raw;   /* of the last hufgpu_decode_stream(): see hufgpu_decode_stream_complete() */
    uint64_t *h_result;           /* pinned mirror */
    uint64_t *d_zipf;             /* 255 cumulative weights */

    /* per-kernel timing: every profiled call records HIP events around its kernels into the
     * next slot; hufgpu_get_profile() sums the slots, so a timed loop needs no host sync */
    int profiling;
    int prof_used;
    int cur_slot;
    int n_stages;
    hipEvent_t (*ev)[MAX_STAGES + 1];
    int slot_stages[PROF_SLOTS];
    int slot_kind[PROF_SLOTS];
    int decode_pending;
    hipStream_t last_stream;
    /* the last enqueued indexed decode: hufgpu_decode_result() decodes a failing block once more, in order */
    const uint8_t *last_st;
    const uint64_t *last_offsets;
    uint8_t *last_out;
    uint64_t last_stream_len, last_out_cap, last_nblocks;
    int last_max_tree;
};

static char g_err[512] = "";

static void set_err(hufgpu_ctx *ctx, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof(g_err), "%s", buf);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", buf);
    fprintf(stderr, "libhuffman(gpu): %s\n", buf);
}

#define HIP_OK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err((ctx), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                              \
            return HUFE_FATAL;                                                              \
        }                                                                                   \
    } while (0)

extern "C" int hufgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int usable = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    return usable;
}

extern "C" const char *hufgpu_last_error(const hufgpu_ctx_t *ctx) { return ctx ? ctx->err : g_err; }

extern "C" uint64_t hufgpu_block_count(uint64_t n, uint64_t blocksize)
{
    if (n == 0) return 0;
    if (blocksize == 0) blocksize = n;            /* src/encoder.c:163-165 */
    return (n + blocksize - 1) / blocksize;
}

extern "C" uint64_t hufgpu_encode_bound(uint64_t n, uint64_t blocksize)
{
    /* per block: 10 + 2*1025 header; payload <= 9 bits per byte (an optimal prefix code never
     * costs more than the 8-bit fixed code, plus the wrap-root bit), +1 byte of padding */
    const uint64_t nb = hufgpu_block_count(n, blocksize);
    return nb * (HUF_HEADER_FIXED + 2ull * HUF_TREE_MAX + 1) + (n * 9 + 7) / 8 + 16;
}

extern "C" int hufgpu_ctx_create(hufgpu_ctx_t **out, int device)
{
    if (!out) return HUFE_ARGUMENT;
    *out = NULL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_err(NULL, "no HIP device available (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return HUFE_FATAL;
    }
    if (device < 0 || device >= n) {
        set_err(NULL, "device %d out of range (have %d)", device, n);
        return HUFE_ARGUMENT;
    }
    hipDeviceProp_t prop;
    HIP_OK(NULL, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(NULL, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
        return HUFE_FATAL;
    }
    hufgpu_ctx *ctx = (hufgpu_ctx *)calloc(1, sizeof(hufgpu_ctx));
    if (!ctx) return HUFE_MEMORY;
    ctx->device = device;
    HIP_OK(NULL, hipSetDevice(device));
    ctx->stream = NULL;   /* the device's default stream: ordered with every blocking stream (torch's default included) */
    HIP_OK(ctx, hipMalloc((void **)&ctx->d_result, 8 * sizeof(uint64_t)));
    HIP_OK(ctx, hipMalloc((void **)&ctx->d_walk, DISC_WORDS * sizeof(uint64_t)));
    HIP_OK(ctx, hipHostMalloc((void **)&ctx->h_result, 16 * sizeof(uint64_t), hipHostMallocDefault));

    /* zipf255 cumulative weights: w_r = floor(2^32 / r), r = 1..255 */
    uint64_t cum[255], acc = 0;
    for (int r = 1; r <= 255; r++) {
        acc += (1ull << 32) / (uint64_t)r;
        cum[r - 1] = acc;
    }
    HIP_OK(ctx, hipMalloc((void **)&ctx->d_zipf, sizeof(cum)));
    HIP_OK(ctx, hipMemcpy(ctx->d_zipf, cum, sizeof(cum), hipMemcpyHostToDevice));
    *out = ctx;
    return HUFE_OK;
}

static void free_two_level(TwoLevel *t);

static void free_encode_ws(hufgpu_ctx *c)
{
    free_two_level(&c->enc_sizes);
    (void)hipFree(c->d_hist);
    (void)hipFree(c->d_codetab);
    (void)hipFree(c->d_treebuf);
    (void)hipFree(c->d_meta);
    (void)hipFree(c->d_offsets);
    (void)hipFree(c->d_chunk_hist); (void)hipFree(c->d_chunk_tot); (void)hipFree(c->d_chunk_bits);
    c->d_chunk_hist = NULL; c->d_chunk_tot = NULL; c->d_chunk_bits = NULL; c->ws_chunks = 0;
    c->d_hist = NULL; c->d_codetab = NULL; c->d_treebuf = NULL; c->d_meta = NULL; c->d_offsets = NULL;
    c->ws_blocks = 0;
}

static void free_disc_ws(hufgpu_ctx *c, int which)
{
    if (which & 1) { (void)hipFree(c->d_wg_counts); (void)hipFree(c->d_wg_base); (void)hipFree(c->d_disc_masks); (void)hipFree(c->d_disc_slots); c->d_disc_slots = NULL; c->d_wg_counts = NULL; c->d_wg_base = NULL; c->d_disc_masks = NULL; c->disc_wgs = 0; }
    if (which & 2) {
        (void)hipFree(c->d_cand); (void)hipFree(c->d_cand_end); (void)hipFree(c->d_chain); (void)hipFree(c->d_cand_status); (void)hipFree(c->d_nxt); (void)hipFree(c->d_spec_off);
        c->d_cand = c->d_cand_end = c->d_chain = NULL; c->d_cand_status = NULL; c->d_nxt = NULL; c->d_spec_off = NULL; c->disc_cands = 0;
    }
}

static void free_big_ws(hufgpu_ctx *c, int which)
{
    if (which & 1) {
        (void)hipFree(c->d_big_entry); (void)hipFree(c->d_big_exit); (void)hipFree(c->d_big_pre); (void)hipFree(c->d_big_cnt);
        (void)hipFree(c->d_big_wgpre); (void)hipFree(c->d_big_wgscratch);
        (void)hipFree(c->d_big_first_pos); (void)hipFree(c->d_big_first_g); (void)hipFree(c->d_big_last_pos);
        c->d_big_entry = c->d_big_exit = c->d_big_pre = c->d_big_wgpre = c->d_big_wgscratch = NULL; c->d_big_cnt = NULL; c->big_lanes = 0;
        c->d_big_first_pos = c->d_big_first_g = c->d_big_last_pos = NULL;
    }
    if (which & 4) { (void)hipFree(c->d_big_sub); c->d_big_sub = NULL; c->big_sub_bytes = 0; }
    if (which & 8) { (void)hipFree(c->d_big_offs); c->d_big_offs = NULL; }
}

static void free_decode_ws(hufgpu_ctx *c)
{
    free_two_level(&c->dec_lens);
    (void)hipFree(c->d_dmeta);
    (void)hipFree(c->d_out_offsets);
    (void)hipFree(c->d_status);
    (void)hipFree(c->d_fix_count); (void)hipFree(c->d_fix_blocks); (void)hipFree(c->d_fix_flag);
    c->d_fix_count = NULL; c->d_fix_blocks = NULL; c->d_fix_flag = NULL;
    c->d_dmeta = NULL; c->d_out_offsets = NULL; c->d_status = NULL;
    c->dws_blocks = 0;
}

extern "C" int hufgpu_ctx_destroy(hufgpu_ctx_t *ctx)
{
    if (!ctx) return HUFE_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_encode_ws(ctx);
    free_decode_ws(ctx);
    free_disc_ws(ctx, 3);
    free_big_ws(ctx, 15);
    (void)hipFree(ctx->d_walk);
    (void)hipFree(ctx->d_result);
    (void)hipFree(ctx->d_zipf);
    (void)hipHostFree(ctx->h_result);
    if (ctx->ev) {
        for (int k = 0; k < PROF_SLOTS; k++)
            for (int i = 0; i <= MAX_STAGES; i++) (void)hipEventDestroy(ctx->ev[k][i]);
        free(ctx->ev);
    }
    free(ctx);
    return HUFE_OK;
}

/* workspace of a two-level prefix sum over `cap` blocks; counters start (and are left) at zero */
static int alloc_two_level(hufgpu_ctx *c, TwoLevel *t, uint64_t cap, bool with_min)
{
    const uint64_t groups = cap / SCAN_GROUP + 2;
    memset(t, 0, sizeof(*t));
    HIP_OK(c, hipMalloc((void **)&t->vals, cap * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&t->local, cap * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&t->gsum, groups * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&t->gprefix, groups * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&t->gcount, groups * SCAN_TICKET_STRIDE * sizeof(uint32_t)));
    HIP_OK(c, hipMalloc((void **)&t->done, sizeof(uint32_t)));
    if (with_min) HIP_OK(c, hipMalloc((void **)&t->gmin, groups * sizeof(uint64_t)));
    HIP_OK(c, hipMemset(t->gcount, 0, groups * SCAN_TICKET_STRIDE * sizeof(uint32_t)));
    HIP_OK(c, hipMemset(t->done, 0, sizeof(uint32_t)));
    HIP_OK(c, hipDeviceSynchronize());   /* the kernels may run on a non-blocking stream: the zeros must be there first */
    return HUFE_OK;
}

static void free_two_level(TwoLevel *t)
{
    (void)hipFree(t->vals); (void)hipFree(t->local); (void)hipFree(t->gsum); (void)hipFree(t->gprefix);
    (void)hipFree(t->gcount); (void)hipFree(t->done); (void)hipFree(t->gmin);
    memset(t, 0, sizeof(*t));
}

static int ensure_encode_ws(hufgpu_ctx *c, uint64_t nblocks)
{
    if (nblocks <= c->ws_blocks) return HUFE_OK;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    free_encode_ws(c);
    const uint64_t cap = nblocks + nblocks / 8 + 16;
    HIP_OK(c, hipMalloc((void **)&c->d_hist, cap * HUF_NSYM * sizeof(uint64_t)));   /* (64-bit counts for chunked blocks) */
    HIP_OK(c, hipMalloc((void **)&c->d_codetab, cap * HUF_NSYM * sizeof(hufcode_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_treebuf, cap * HUF_TREE_STRIDE * sizeof(int16_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_meta, cap * sizeof(HufBlockMeta)));
    HIP_OK(c, hipMalloc((void **)&c->d_offsets, (cap + 1) * sizeof(uint64_t)));
    int rc2 = alloc_two_level(c, &c->enc_sizes, cap, false);
    if (rc2) return rc2;
    c->ws_blocks = cap;
    return HUFE_OK;
}

static int ensure_chunk_ws(hufgpu_ctx *c, uint64_t nchunks)
{
    if (nchunks <= c->ws_chunks) return HUFE_OK;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_chunk_hist); (void)hipFree(c->d_chunk_tot); (void)hipFree(c->d_chunk_bits);
    c->d_chunk_hist = NULL; c->d_chunk_tot = NULL; c->d_chunk_bits = NULL; c->ws_chunks = 0;
    const uint64_t cap = nchunks + nchunks / 8 + 16;
    HIP_OK(c, hipMalloc((void **)&c->d_chunk_hist, cap * HUF_NSYM * sizeof(uint32_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_chunk_tot, cap * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_chunk_bits, cap * sizeof(uint64_t)));
    c->ws_chunks = cap;
    return HUFE_OK;
}

static int ensure_decode_ws(hufgpu_ctx *c, uint64_t nblocks)
{
    if (nblocks <= c->dws_blocks) return HUFE_OK;
    HIP_OK(c, hipStreamSynchronize(c->stream));
    free_decode_ws(c);
    const uint64_t cap = nblocks + nblocks / 8 + 16;
    HIP_OK(c, hipMalloc((void **)&c->d_dmeta, cap * sizeof(HufDecodeMeta)));
    HIP_OK(c, hipMalloc((void **)&c->d_out_offsets, (cap + 1) * sizeof(uint64_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_status, cap * sizeof(int32_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_fix_count, 2 * sizeof(uint32_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_fix_blocks, cap * sizeof(uint32_t)));
    HIP_OK(c, hipMalloc((void **)&c->d_fix_flag, cap * sizeof(uint32_t)));
    HIP_OK(c, hipMemset(c->d_fix_count, 0, 2 * sizeof(uint32_t)));
    HIP_OK(c, hipMemset(c->d_fix_flag, 0, cap * sizeof(uint32_t)));
    int rc2 = alloc_two_level(c, &c->dec_lens, cap, true);
    if (rc2) return rc2;
    c->dws_blocks = cap;
    return HUFE_OK;
}

static inline hipStream_t pick_stream(hufgpu_ctx *c, void *stream) { (void)c; return (hipStream_t)stream; }

#define STAGE_BEGIN(c, s, kind)                                                         \
    do {                                                                                \
        (c)->n_stages = 0;                                                              \
        (c)->cur_slot = -1;                                                             \
        if ((c)->profiling && (c)->prof_used < PROF_SLOTS) {                            \
            (c)->cur_slot = (c)->prof_used++;                                           \
            (c)->slot_kind[(c)->cur_slot] = (kind);                                     \
            (c)->slot_stages[(c)->cur_slot] = 0;                                        \
            HIP_OK((c), hipEventRecord((c)->ev[(c)->cur_slot][0], (s)));                \
        }                                                                               \
    } while (0)
#define STAGE_MARK(c, s)                                                                \
    do {                                                                                \
        if ((c)->cur_slot >= 0 && (c)->n_stages < MAX_STAGES) {                         \
            (c)->n_stages++;                                                            \
            (c)->slot_stages[(c)->cur_slot] = (c)->n_stages;                            \
            HIP_OK((c), hipEventRecord((c)->ev[(c)->cur_slot][(c)->n_stages], (s)));    \
        }                                                                               \
    } while (0)

extern "C" int hufgpu_set_profiling(hufgpu_ctx_t *ctx, int enabled)
{
    if (!ctx) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (enabled && !ctx->ev) {
        ctx->ev = (hipEvent_t(*)[MAX_STAGES + 1])calloc(PROF_SLOTS, sizeof(*ctx->ev));
        if (!ctx->ev) return HUFE_MEMORY;
        for (int k = 0; k < PROF_SLOTS; k++)
            for (int i = 0; i <= MAX_STAGES; i++) HIP_OK(ctx, hipEventCreate(&ctx->ev[k][i]));
    }
    ctx->profiling = enabled ? 1 : 0;
    if (enabled == 1) ctx->prof_used = 0;        /* 1 = start a new record, 2 = resume, 0 = pause (record kept) */
    ctx->cur_slot = -1;
    return HUFE_OK;
}

extern "C" int hufgpu_get_profile(hufgpu_ctx_t *ctx, int kind, float *ms_sum, int max_stages,
                                  int *n_stages, int *n_calls)
{
    if (!ctx || !ms_sum || !n_stages || !n_calls) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    *n_stages = 0;
    *n_calls = 0;
    for (int i = 0; i < max_stages; i++) ms_sum[i] = 0.f;
    for (int k = 0; k < ctx->prof_used; k++) {
        if (ctx->slot_kind[k] != kind) continue;
        const int ns = ctx->slot_stages[k] < max_stages ? ctx->slot_stages[k] : max_stages;
        if (ns <= 0) continue;
        HIP_OK(ctx, hipEventSynchronize(ctx->ev[k][ctx->slot_stages[k]]));
        for (int i = 0; i < ns; i++) {
            float ms = 0.f;
            HIP_OK(ctx, hipEventElapsedTime(&ms, ctx->ev[k][i], ctx->ev[k][i + 1]));
            ms_sum[i] += ms;
        }
        if (ns > *n_stages) *n_stages = ns;
        (*n_calls)++;
    }
    return HUFE_OK;
}

static int check_block_args(hufgpu_ctx *c, uint64_t n, uint64_t *blocksize)
{
    if (*blocksize == 0) *blocksize = n;
    if (*blocksize > HUFGPU_MAX_BLOCK) {
        set_err(c, "blocksize %llu exceeds the kernel limit of %llu bytes", (unsigned long long)*blocksize,
                (unsigned long long)HUFGPU_MAX_BLOCK);
        return HUFE_ARGUMENT;
    }
    return HUFE_OK;
}

extern "C" int hufgpu_histogram(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                                uint32_t *d_hist, void *stream)
{
    if (!ctx || (!d_in && n) || !d_hist) return HUFE_ARGUMENT;
    if (n == 0) return HUFE_OK;
    int rc = check_block_args(ctx, n, &blocksize);
    if (rc) return rc;
    if (blocksize > 0xffffffffull) {
        set_err(ctx, "hufgpu_histogram returns 32-bit counts: blocks of 2^32 bytes and more are not taken");
        return HUFE_ARGUMENT;
    }
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t nb = hufgpu_block_count(n, blocksize);
    hist256_kernel<HIST_THREADS><<<dim3((unsigned)nb), dim3(HIST_THREADS), 0, s>>>((const uint8_t *)d_in, n, blocksize, d_hist);
    HIP_OK(ctx, hipGetLastError());
    return HUFE_OK;
}

/* where the two arrays of a sub-index live inside the caller's buffer */
static HufSubIndex sub_index_view(void *d_sub, uint64_t n, uint64_t blocksize)
{
    HufSubIndex v;
    memset(&v, 0, sizeof(v));
    if (!d_sub || n == 0) return v;
    if (blocksize == 0) blocksize = n;
    const uint64_t nb = hufgpu_block_count(n, blocksize);
    v.gpb = ((blocksize + HUF_SUB_GROUP - 1) / HUF_SUB_GROUP + 7) & ~7ull;   /* rows of 16-byte multiples */
    v.tpb = (blocksize + HUF_SUB_TILE - 1) / HUF_SUB_TILE;
    v.tile_bits = (uint64_t *)d_sub;
    v.group_bits = (uint16_t *)((uint64_t *)d_sub + nb * v.tpb);
    v.lens = (uint8_t *)(v.group_bits + nb * v.gpb);        /* gpb is a multiple of 8: 16-byte aligned */
    return v;
}

extern "C" uint64_t hufgpu_sub_index_bytes(uint64_t n, uint64_t blocksize)
{
    if (n == 0) return 0;
    if (blocksize == 0) blocksize = n;
    const uint64_t nb = hufgpu_block_count(n, blocksize);
    const uint64_t gpb = ((blocksize + HUF_SUB_GROUP - 1) / HUF_SUB_GROUP + 7) & ~7ull;
    const uint64_t tpb = (blocksize + HUF_SUB_TILE - 1) / HUF_SUB_TILE;
    return nb * tpb * sizeof(uint64_t) + nb * gpb * sizeof(uint16_t) + nb * HUF_NSYM;
}

static int encode_impl(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                       void *d_out, uint64_t out_cap, uint64_t *d_block_offsets, void *d_sub_index,
                       uint64_t *out_len, void *stream)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (n == 0) {                                  /* src/encoder.c:288: nothing to do */
        if (out_len) *out_len = 0;
        if (d_block_offsets) {                     /* the index of an empty stream: its length, 0 */
            HIP_OK(ctx, hipSetDevice(ctx->device));
            HIP_OK(ctx, hipMemsetAsync(d_block_offsets, 0, sizeof(uint64_t), pick_stream(ctx, stream)));
        }
        return HUFE_OK;
    }
    if (!d_in || !d_out) return HUFE_ARGUMENT;
    int rc = check_block_args(ctx, n, &blocksize);
    if (rc) return rc;
    if (out_cap < hufgpu_encode_bound(n, blocksize)) {
        set_err(ctx, "output capacity %llu below hufgpu_encode_bound() = %llu", (unsigned long long)out_cap,
                (unsigned long long)hufgpu_encode_bound(n, blocksize));
        return HUFE_ARGUMENT;
    }
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const uint64_t nb = hufgpu_block_count(n, blocksize);
    if (nb > 0x7fffffffull) {
        set_err(ctx, "too many blocks (%llu)", (unsigned long long)nb);
        return HUFE_ARGUMENT;
    }
    rc = ensure_encode_ws(ctx, nb);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    uint64_t *offs = d_block_offsets ? d_block_offsets : ctx->d_offsets;
    const uint8_t *in = (const uint8_t *)d_in;
    if (d_sub_index && ((uintptr_t)d_sub_index & 7u)) {
        set_err(ctx, "the sub-index buffer must be 8-byte aligned");
        return HUFE_ARGUMENT;
    }
    const HufSubIndex sub = sub_index_view(d_sub_index, n, blocksize);

    STAGE_BEGIN(ctx, s, PROF_ENCODE);
    TwoLevel sizes = ctx->enc_sizes;
    static const bool fused_only = getenv("HUF_GPU_FUSED_HIST") && atoi(getenv("HUF_GPU_FUSED_HIST")) != 0;   /* (measurements: the one-launch form) */
    if (blocksize < HUF_CHUNKED_FROM) {
        /* counts, tree and the sums of the encoded sizes in one launch (the profile's "tree" and
         * "scan_sizes" stages are then empty) */
        sizes.total = offs + nb;
        if (blocksize >= HL_MIN_BLOCK && !fused_only) {
            /* counts with lane-private counters at the rate HBM delivers, then the trees as a launch of their
             * own (kernels/hist_lanes.hpp): 0.19 + 0.13 ms per GiB on zipf255 where the fused kernel takes 0.44 */
            hist_lanes_kernel<HL_THREADS><<<dim3((unsigned)nb), dim3(HL_THREADS), 0, s>>>(in, n, blocksize, ctx->d_hist);
            STAGE_MARK(ctx, s);
            tree_wave_kernel<<<dim3((unsigned)nb), dim3(64), 0, s>>>(ctx->d_hist, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, sizes);
            STAGE_MARK(ctx, s);
            STAGE_MARK(ctx, s);
        } else
        if (blocksize <= HT_PACKED_MAX_BLOCK)
            hist_tree_kernel<HIST_THREADS, true><<<dim3((unsigned)nb), dim3(HIST_THREADS), 0, s>>>(in, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, sizes);
        else
            hist_tree_kernel<HIST_THREADS, false><<<dim3((unsigned)nb), dim3(HIST_THREADS), 0, s>>>(in, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, sizes);
        if (!(blocksize >= HL_MIN_BLOCK && !fused_only)) {
            STAGE_MARK(ctx, s);
            STAGE_MARK(ctx, s);
            STAGE_MARK(ctx, s);
        }
    } else {
        /* blocks of HUF_BIG_BLOCK bytes and more are cut into chunks, one workgroup each (blocksize = 0:
         * the whole input is ONE block, src/encoder.c:163-165 - the reference's default) */
        const uint64_t cpb = (blocksize + HUF_CHUNK_SYMS - 1) / HUF_CHUNK_SYMS;
        const uint64_t nchunks = nb * cpb;
        if (nchunks > 0x7fffffffull) return HUFE_ARGUMENT;
        rc = ensure_chunk_ws(ctx, nchunks);
        if (rc) return rc;
        ChunkGeom geo;
        geo.n = n;
        geo.blocksize = blocksize;
        geo.cpb = (uint32_t)cpb;
        chunk_hist_kernel<HL_THREADS><<<dim3((unsigned)nchunks), dim3(HL_THREADS), 0, s>>>(in, geo, ctx->d_chunk_hist);
        if (blocksize < HUF_BIG_BLOCK) {
            /* rates below 2^23: the wave-per-block tree with 32-bit keys (its sums of the encoded sizes are not used
             * here: scan_sizes_kernel writes the index below) */
            sizes.total = offs + nb;
            block_hist32_kernel<<<dim3((unsigned)nb), dim3(HUF_NSYM), 0, s>>>(ctx->d_chunk_hist, (uint32_t)cpb, ctx->d_hist);
            STAGE_MARK(ctx, s);
            tree_wave_kernel<<<dim3((unsigned)nb), dim3(64), 0, s>>>(ctx->d_hist, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, sizes);
        } else {
            block_hist_kernel<<<dim3((unsigned)nb), dim3(HUF_NSYM), 0, s>>>(ctx->d_chunk_hist, (uint32_t)cpb, (uint64_t *)ctx->d_hist);
            STAGE_MARK(ctx, s);
            tree_kernel<uint64_t, uint64_t><<<dim3((unsigned)nb), dim3(64), 0, s>>>((const uint64_t *)ctx->d_hist, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta);
        }
        STAGE_MARK(ctx, s);
        scan_sizes_kernel<SCAN_THREADS><<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(ctx->d_meta, nb, offs);
        chunk_total_kernel<<<dim3((unsigned)nchunks), dim3(64), 0, s>>>(ctx->d_chunk_hist, (uint32_t)cpb, ctx->d_codetab, ctx->d_meta, ctx->d_chunk_tot);
        chunk_scan_kernel<SCAN_THREADS><<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(ctx->d_chunk_tot, (uint32_t)cpb, ctx->d_chunk_bits);
        STAGE_MARK(ctx, s);
        sizes.local = NULL;              /* pack reads the finished index */
        PackChunk ck;
        ck.chunk_bits = ctx->d_chunk_bits;
        ck.chunk_syms = HUF_CHUNK_SYMS;
        ck.cpb = (uint32_t)cpb;
        pack_chunk_kernel<PACK_THREADS, false><<<dim3((unsigned)nchunks), dim3(PACK_THREADS), 0, s>>>(in, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, offs, sizes, (uint8_t *)d_out, sub, ck);
    }
    if (blocksize >= HUF_CHUNKED_FROM) {
        /* (packed above) */
    } else if (blocksize <= 121392ull)   /* deepest possible code <= 24 bits: 32-bit code path only */
        pack_kernel<PACK_THREADS, true><<<dim3((unsigned)nb), dim3(PACK_THREADS), 0, s>>>(in, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, offs, sizes, (uint8_t *)d_out, sub);
    else
        pack_kernel<PACK_THREADS, false><<<dim3((unsigned)nb), dim3(PACK_THREADS), 0, s>>>(in, n, blocksize, ctx->d_codetab, ctx->d_treebuf, ctx->d_meta, offs, sizes, (uint8_t *)d_out, sub);
    STAGE_MARK(ctx, s);
    HIP_OK(ctx, hipGetLastError());

    if (out_len) {
        HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, offs + nb, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_OK(ctx, hipStreamSynchronize(s));
        *out_len = ctx->h_result[0];
    }
    return HUFE_OK;
}

extern "C" int hufgpu_encode(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                             void *d_out, uint64_t out_cap, uint64_t *d_block_offsets,
                             uint64_t *out_len, void *stream)
{
    return encode_impl(ctx, d_in, n, blocksize, d_out, out_cap, d_block_offsets, NULL, out_len, stream);
}

extern "C" int hufgpu_encode_sub(hufgpu_ctx_t *ctx, const void *d_in, uint64_t n, uint64_t blocksize,
                                 void *d_out, uint64_t out_cap, uint64_t *d_block_offsets,
                                 void *d_sub_index, uint64_t *out_len, void *stream)
{
    return encode_impl(ctx, d_in, n, blocksize, d_out, out_cap, d_block_offsets, d_sub_index, out_len, stream);
}

static int decode_chain(hufgpu_ctx *ctx, const uint8_t *st, uint64_t avail, uint64_t length, uint8_t *out,
                        uint64_t out_cap, int max_tree, hipStream_t s, uint64_t *raw, uint64_t *used,
                        uint64_t *good_used, uint64_t *good_raw);

extern "C" int hufgpu_decode_result(hufgpu_ctx_t *ctx, uint64_t *raw_len)
{
    if (!ctx) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (!ctx->decode_pending) {
        if (raw_len) *raw_len = 0;
        return HUFE_OK;
    }
    HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_result, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->last_stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->last_stream));
    ctx->decode_pending = 0;
    const uint8_t *last_st = ctx->last_st;
    ctx->last_st = NULL;                           /* the caller's buffers are not looked at again after this call */
    const uint64_t failing = ctx->h_result[2];
    if (failing == ~0ull) {                        /* every block decoded */
        if (raw_len) *raw_len = ctx->h_result[1];
        return HUFE_OK;
    }
    /* first failing block in stream order: its error code, and the bytes of the blocks before it */
    int32_t err = HUFE_FATAL;
    uint64_t before = 0;
    HIP_OK(ctx, hipMemcpyAsync(&err, ctx->d_status + failing, sizeof(err), hipMemcpyDeviceToHost, ctx->last_stream));
    HIP_OK(ctx, hipMemcpyAsync(&before, ctx->d_out_offsets + failing, sizeof(before), hipMemcpyDeviceToHost, ctx->last_stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->last_stream));
    if ((err == HUFE_RW || err == HUFE_CORRUPTED) && last_st && failing < ctx->last_nblocks && before <= ctx->last_out_cap) {
        /* src/decoder.c:69-91 delivers the symbols in front of the failure: the failing block once more by the
         * exact in-order decoder, its record [o0, o1) as the whole input (a walk that needs more fails like the
         * reference's reader at the end of its input) */
        uint64_t o[2] = {0, 0};
        HIP_OK(ctx, hipMemcpyAsync(o, ctx->last_offsets + failing, sizeof(o), hipMemcpyDeviceToHost, ctx->last_stream));
        HIP_OK(ctx, hipStreamSynchronize(ctx->last_stream));
        if (o[1] > ctx->last_stream_len) o[1] = ctx->last_stream_len;
        if (o[0] < o[1]) {
            uint64_t raw = 0, used = 0, gu = 0, gr = 0;
            const int rc = decode_chain(ctx, last_st + o[0], o[1] - o[0], 1, ctx->last_out + before, ctx->last_out_cap - before,
                                        ctx->last_max_tree, ctx->last_stream, &raw, &used, &gu, &gr);
            if (rc == err) before += raw;
        }
    }
    if (raw_len) *raw_len = before;
    if (err == HUFE_ARGUMENT) set_err(ctx, "block %llu is longer than the kernels support", (unsigned long long)failing);
    if (err == HUFE_MEMORY) set_err(ctx, "output buffer too small (block %llu)", (unsigned long long)failing);
    return err;
}

/* One small encode with ONE synchronisation (include/huffman_gpu.h): input from pinned host memory, the stream and its
 * length back into pinned host memory.  A call through the general entry points waits three times (input up, the length,
 * the stream back); for inputs of a few KiB those waits are most of the call. */
extern "C" int hufgpu_encode_small(hufgpu_ctx_t *ctx, const void *h_in_pinned, uint64_t n, uint64_t blocksize, void *d_in,
                                   void *d_out, uint64_t out_cap, void *h_out_pinned, uint64_t h_out_cap, uint64_t *out_len)
{
    if (!ctx || !h_in_pinned || !d_in || !d_out || !h_out_pinned || !out_len || n == 0) return HUFE_ARGUMENT;
    const uint64_t bound = hufgpu_encode_bound(n, blocksize);
    const uint64_t len_at = (bound + 7u) & ~7ull;
    if (h_out_cap < len_at + 8u || out_cap < bound) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_OK(ctx, hipMemcpyAsync(d_in, h_in_pinned, n, hipMemcpyHostToDevice, s));
    const int rc = encode_impl(ctx, d_in, n, blocksize, d_out, out_cap, NULL, NULL, NULL, (void *)s);
    if (rc != HUFE_OK) return rc;
    const uint64_t nb = hufgpu_block_count(n, blocksize ? blocksize : n);
    HIP_OK(ctx, hipMemcpyAsync(h_out_pinned, d_out, bound, hipMemcpyDeviceToHost, s));
    HIP_OK(ctx, hipMemcpyAsync((char *)h_out_pinned + len_at, ctx->d_offsets + nb, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_OK(ctx, hipStreamSynchronize(s));
    *out_len = *(const uint64_t *)((const char *)h_out_pinned + len_at);
    return (*out_len <= bound) ? HUFE_OK : HUFE_FATAL;
}

/* How many blocks of the last enqueued decode were handed on: counters[0] = to the exact decoder
 * (decode_fix_kernel), counters[1] = 0 (round 4's one-pass decoder, gone with round 5's clean-up).  Synchronises. */
extern "C" int hufgpu_decode_counters(hufgpu_ctx_t *ctx, uint32_t *counters)
{
    if (!ctx || !counters) return HUFE_ARGUMENT;
    counters[0] = counters[1] = 0;
    if (!ctx->d_fix_count) return HUFE_OK;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (ctx->last_stream || ctx->decode_pending) HIP_OK(ctx, hipStreamSynchronize(ctx->last_stream));
    HIP_OK(ctx, hipMemcpy(counters, ctx->d_fix_count, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HUFE_OK;
}

/* Bandwidth calibration (include/huffman_gpu.h): one launch of a kernel that only moves bytes. */
template <int KIND>
static int calib_launch(int variant, const uint8_t *a, uint8_t *b, uint64_t bytes, uint32_t *flag, hipStream_t s)
{
#define CALIB_CASE(V, T, P, NTL, NTS) case V: calib_bw_kernel<T, P, KIND, NTL, NTS><<<dim3((unsigned)(bytes / P)), dim3(T), 0, s>>>(a, b, flag); return P;
    switch (variant) {
        CALIB_CASE(0, 256, 16384, true, true)
        CALIB_CASE(1, 256, 16384, true, false)
        CALIB_CASE(2, 256, 16384, false, false)
        CALIB_CASE(3, 512, 65536, true, true)
        CALIB_CASE(4, 512, 65536, true, false)
        CALIB_CASE(5, 256, 4096, true, true)
        CALIB_CASE(6, 256, 4096, false, false)
        CALIB_CASE(7, 1024, 65536, true, true)
    }
#undef CALIB_CASE
    return 0;
}
extern "C" int hufgpu_calib_bandwidth(hufgpu_ctx_t *ctx, int kind, int variant, const void *d_a, void *d_b, uint64_t bytes, void *stream)
{
    if (!ctx || kind < 0 || kind > 2 || variant < 0 || variant >= HUFGPU_CALIB_VARIANTS) return HUFE_ARGUMENT;
    if ((kind != 2 && !d_a) || (kind != 1 && !d_b) || bytes == 0 || (bytes & 65535u) || (((uintptr_t)d_a | (uintptr_t)d_b) & 15u)) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    uint32_t *flag = (uint32_t *)ctx->d_result;          /* (a word nobody reads: the read-only kernel's "result") */
    int per = 0;
    if (kind == 0) per = calib_launch<0>(variant, (const uint8_t *)d_a, (uint8_t *)d_b, bytes, flag + 6, s);
    else if (kind == 1) per = calib_launch<1>(variant, (const uint8_t *)d_a, (uint8_t *)d_b, bytes, flag + 6, s);
    else per = calib_launch<2>(variant, (const uint8_t *)d_a, (uint8_t *)d_b, bytes, flag + 6, s);
    HIP_OK(ctx, hipGetLastError());
    return per ? HUFE_OK : HUFE_ARGUMENT;
}

static int decode_impl(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t stream_len,
                       const uint64_t *d_block_offsets, uint64_t nblocks, const HufSubIndex *sub, uint64_t blocksize,
                       void *d_out, uint64_t out_cap, uint32_t flags, uint64_t *raw_len, void *stream)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (nblocks == 0 || stream_len == 0) {         /* src/decoder.c:218, test/decode_test.c:32-36 */
        ctx->decode_pending = 0;
        if (raw_len) *raw_len = 0;
        return HUFE_OK;
    }
    if (!d_stream || !d_block_offsets || (!d_out && out_cap)) return HUFE_ARGUMENT;
    if (nblocks > 0x7fffffffull) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_decode_ws(ctx, nblocks);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    const int max_tree = (flags & HUFGPU_RELAXED_TREE) ? HUF_TREE_MAX : HUF_TREE_STRICT;
    const uint8_t *st = (const uint8_t *)d_stream;

    STAGE_BEGIN(ctx, s, PROF_DECODE);
    unsigned long long *res = (unsigned long long *)ctx->d_result;
    /* header parse + two-level sums of the block lengths; also (re)initialises result[1] and [2] */
    TwoLevel lens = ctx->dec_lens;
    lens.total = (uint64_t *)res + 1;
    lens.total2 = ctx->d_out_offsets + nblocks;
    lens.min_out = (uint64_t *)res + 2;
    decode_prepare_kernel<<<dim3((unsigned)((nblocks + SCAN_GROUP - 1) / SCAN_GROUP)), dim3(SCAN_GROUP), 0, s>>>(st, stream_len, d_block_offsets, nblocks, max_tree, ctx->d_dmeta, ctx->d_status, lens, ctx->d_fix_count);
    STAGE_MARK(ctx, s);
    if (sub && sub->tile_bits) {
        /* the encoder's sub-index: one table pass per symbol, verified; what cannot be verified is
         * decoded again by the self-synchronising decoder (decode_fix_kernel) */
        const uint64_t cpb = (blocksize + DSUB_CHUNK_SYMS - 1) / DSUB_CHUNK_SYMS;
        if (nblocks * cpb > 0x7fffffffull) return HUFE_ARGUMENT;
        DecFixList fix;
        fix.count = ctx->d_fix_count;
        fix.blocks = ctx->d_fix_blocks;
        fix.flag = ctx->d_fix_flag;
        decode_sub_kernel<DSUB_THREADS><<<dim3((unsigned)(nblocks * cpb)), dim3(DSUB_THREADS), 0, s>>>(st, stream_len, d_block_offsets, ctx->d_dmeta, ctx->d_out_offsets, lens, (uint8_t *)d_out, out_cap, ctx->d_status, res, *sub, blocksize, (uint32_t)cpb, fix);
        const unsigned fix_grid = (unsigned)(nblocks < 1024 ? nblocks : 1024);
        decode_fix_kernel<DEC_THREADS><<<dim3(fix_grid), dim3(DEC_THREADS), 0, s>>>(st, stream_len, d_block_offsets, ctx->d_dmeta, lens, (uint8_t *)d_out, out_cap, ctx->d_status, res, fix);
    } else {
        static const bool exact_only = getenv("HUF_GPU_EXACT_DECODE") && atoi(getenv("HUF_GPU_EXACT_DECODE")) != 0;   /* (measurements: the exact decoder for every block) */
        if (exact_only) {
            decode_kernel<DEC_THREADS><<<dim3((unsigned)nblocks), dim3(DEC_THREADS), 0, s>>>(st, stream_len, d_block_offsets, ctx->d_dmeta, ctx->d_out_offsets, lens, (uint8_t *)d_out, out_cap, ctx->d_status, res);
        } else {
            /* the lean self-synchronising decoder (kernels/decode_fast.hpp); what it cannot vouch for - a damaged
             * stream, an unusual tree - is decoded again by the exact one, which also reports the reference's error */
            DecFixList fix;
            fix.count = ctx->d_fix_count;
            fix.blocks = ctx->d_fix_blocks;
            fix.flag = ctx->d_fix_flag;
            const unsigned fix_grid = (unsigned)(nblocks < 1024 ? nblocks : 1024);
            DecodeFastArgs fa;
            fa.stream = st; fa.stream_len = stream_len; fa.offsets = d_block_offsets; fa.dmeta = ctx->d_dmeta; fa.out_offsets = ctx->d_out_offsets;
            fa.lens = lens; fa.out = (uint8_t *)d_out; fa.out_cap = out_cap; fa.status = ctx->d_status; fa.result = res; fa.fix = fix;
            decode_fast_kernel<DEC_THREADS><<<dim3((unsigned)nblocks), dim3(DEC_THREADS), 0, s>>>(fa);
            decode_fix_kernel<DEC_THREADS><<<dim3(fix_grid), dim3(DEC_THREADS), 0, s>>>(st, stream_len, d_block_offsets, ctx->d_dmeta, lens, (uint8_t *)d_out, out_cap, ctx->d_status, res, fix);
        }
    }
    STAGE_MARK(ctx, s);
    HIP_OK(ctx, hipGetLastError());
    ctx->decode_pending = 1;
    ctx->last_stream = s;
    ctx->last_st = st;
    ctx->last_stream_len = stream_len;
    ctx->last_offsets = d_block_offsets;
    ctx->last_nblocks = nblocks;
    ctx->last_out = (uint8_t *)d_out;
    ctx->last_out_cap = out_cap;
    ctx->last_max_tree = max_tree;
    if (raw_len) return hufgpu_decode_result(ctx, raw_len);
    return HUFE_OK;
}

extern "C" int hufgpu_decode(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t stream_len,
                             const uint64_t *d_block_offsets, uint64_t nblocks, void *d_out,
                             uint64_t out_cap, uint32_t flags, uint64_t *raw_len, void *stream)
{
    return decode_impl(ctx, d_stream, stream_len, d_block_offsets, nblocks, NULL, 0, d_out, out_cap, flags, raw_len, stream);
}

extern "C" int hufgpu_decode_sub(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t stream_len,
                                 const uint64_t *d_block_offsets, uint64_t raw_size, uint64_t blocksize,
                                 const void *d_sub_index, void *d_out, uint64_t out_cap, uint32_t flags,
                                 uint64_t *raw_len, void *stream)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (blocksize == 0) blocksize = raw_size;
    const uint64_t nblocks = hufgpu_block_count(raw_size, blocksize);
    if (d_sub_index && ((uintptr_t)d_sub_index & 7u)) {
        set_err(ctx, "the sub-index buffer must be 8-byte aligned");
        return HUFE_ARGUMENT;
    }
    const HufSubIndex sub = sub_index_view((void *)d_sub_index, raw_size, blocksize);
    return decode_impl(ctx, d_stream, stream_len, d_block_offsets, nblocks, &sub, blocksize, d_out, out_cap, flags, raw_len, stream);
}

/* The exact sequential decoder (one workgroup, blocks in order). */
static int decode_chain(hufgpu_ctx *ctx, const uint8_t *st, uint64_t avail, uint64_t length, uint8_t *out,
                        uint64_t out_cap, int max_tree, hipStream_t s, uint64_t *raw, uint64_t *used,
                        uint64_t *good_used, uint64_t *good_raw)
{
    decode_chain_kernel<DEC_THREADS><<<dim3(1), dim3(DEC_THREADS), 0, s>>>(st, avail, length, max_tree, out, out_cap, ctx->d_result, NULL, 0);
    HIP_OK(ctx, hipGetLastError());
    HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_result, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_OK(ctx, hipStreamSynchronize(s));
    *raw = ctx->h_result[1];
    *used = ctx->h_result[2];
    *good_used = ctx->h_result[4];
    *good_raw = ctx->h_result[5];
    return (int)ctx->h_result[0];
}

/* One small decode with ONE synchronisation (include/huffman_gpu.h), hufgpu_encode_small's twin: the raw stream from pinned
 * host memory, the in-order chain (decode_chain_lean_kernel: the block loop of src/decoder.c:218-276, one workgroup, the lean decoders in
 * front of the exact one), the output and the kernel's six
 * result words back into pinned host memory behind one another.  A call through the general entry points waits three
 * times (stream up, the result words, the output back): 62-140 microseconds where the kernel takes twenty. */
extern "C" int hufgpu_decode_small(hufgpu_ctx_t *ctx, const void *h_in_pinned, uint64_t avail, uint64_t length, uint32_t flags,
                                   void *d_in, void *d_out, uint64_t out_cap, void *h_out_pinned, uint64_t h_out_cap,
                                   uint64_t *raw_len, uint64_t *consumed)
{
    if (!ctx || !h_in_pinned || !d_in || !d_out || !h_out_pinned || !raw_len || !consumed || avail == 0) return HUFE_ARGUMENT;
    const uint64_t bound = out_cap < avail * 8u + 64u ? out_cap : avail * 8u + 64u;         /* (a symbol takes a bit at least) */
    /* what comes back with the result words: twice the stream and a bit - all of the output unless the stream is less than half
     * of it (round 6; until then the whole bound, eight times the stream, came back every time: 0.5 MiB for a call of 64 KiB).
     * The rest, if there is one, follows in a second copy. */
    const uint64_t first = 2u * avail + 4096u;
    const uint64_t copy = bound < first ? bound : first;
    const uint64_t res_at = (bound + 7u) & ~7ull;
    if (h_out_cap < res_at + 6u * sizeof(uint64_t)) return HUFE_ARGUMENT;
    *raw_len = *consumed = 0;
    if (length == 0) return HUFE_OK;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int max_tree = (flags & HUFGPU_RELAXED_TREE) ? HUF_TREE_MAX : HUF_TREE_STRICT;
    HIP_OK(ctx, hipMemcpyAsync(d_in, h_in_pinned, avail, hipMemcpyHostToDevice, s));
    decode_chain_lean_kernel<DEC_THREADS><<<dim3(1), dim3(DEC_THREADS), 0, s>>>((const uint8_t *)d_in, avail, length, max_tree, (uint8_t *)d_out, out_cap, ctx->d_result);
    HIP_OK(ctx, hipGetLastError());
    HIP_OK(ctx, hipMemcpyAsync(h_out_pinned, d_out, copy, hipMemcpyDeviceToHost, s));
    HIP_OK(ctx, hipMemcpyAsync((char *)h_out_pinned + res_at, ctx->d_result, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_OK(ctx, hipStreamSynchronize(s));
    const uint64_t *r = (const uint64_t *)((const char *)h_out_pinned + res_at);
    *raw_len = r[1];
    *consumed = r[2];
    ctx->complete_used = r[4];
    ctx->complete_raw = r[5];
    if (r[1] > bound) return HUFE_FATAL;                                                      /* (cannot be: more symbols than bits) */
    if (r[1] > copy) {
        HIP_OK(ctx, hipMemcpyAsync((char *)h_out_pinned + copy, (const char *)d_out + copy, r[1] - copy, hipMemcpyDeviceToHost, s));
        HIP_OK(ctx, hipStreamSynchronize(s));
    }
    return (int)r[0];
}

/* Leading blocks of HUF_BIG_BLOCK symbols and more (blocksize = 0 makes the whole input ONE block,
 * src/encoder.c:163-165): one workgroup per block would leave the device idle, so a sub-index is
 * built for each such block (kernels/spec_index.hpp) and decode_sub_kernel decodes - and verifies -
 * it chunk by chunk.  Stops at the first block this does not apply to or does not work for; the
 * caller's general path takes over at *pos / *rawpos and reports whatever is wrong there. */
static int decode_big_blocks(hufgpu_ctx *ctx, const uint8_t *st, uint64_t avail, uint64_t length, uint8_t *out,
                             uint64_t out_cap, uint32_t flags, hipStream_t s, void *stream, uint64_t *pos_io,
                             uint64_t *rawpos_io)
{
    const int max_tree = (flags & HUFGPU_RELAXED_TREE) ? HUF_TREE_MAX : HUF_TREE_STRICT;
    uint64_t pos = *pos_io, rawpos = *rawpos_io;
    if (!ctx->d_big_offs) HIP_OK(ctx, hipMalloc((void **)&ctx->d_big_offs, (SPEC_WORDS + 2) * sizeof(uint64_t)));
    unsigned long long *d_status = (unsigned long long *)ctx->d_big_offs;
    uint64_t *d_offs = ctx->d_big_offs + SPEC_WORDS;
    while (pos < length && avail - pos >= HUF_HEADER_FIXED) {
        spec_head_kernel<<<dim3(1), dim3(64), 0, s>>>(st, avail, pos, d_status);
        HIP_OK(ctx, hipGetLastError());
        HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, d_status, SPEC_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_OK(ctx, hipStreamSynchronize(s));
        const uint64_t block_len = ctx->h_result[SPEC_BLOCK_LEN];
        const long long tl = (long long)ctx->h_result[SPEC_TREE_LEN];
        const long long leaf = (long long)ctx->h_result[SPEC_LEAF];
        if (ctx->h_result[SPEC_FAIL] || block_len < HUF_BIG_BLOCK || block_len > HUFGPU_MAX_BLOCK) break;
        if (tl < 1 || tl > max_tree || block_len > out_cap - rawpos) break;
        const uint64_t pay_off = pos + HUF_HEADER_FIXED + 2ull * (uint64_t)tl;
        if (pay_off > avail) break;
        const uint64_t pay_bytes = avail - pay_off;

        const uint64_t sub_bytes = hufgpu_sub_index_bytes(block_len, block_len);
        /* (workspace that cannot be had - a block of many GiB needs a quarter of its size - is no
         * error: the general path takes the block) */
        if (sub_bytes > ctx->big_sub_bytes) {
            free_big_ws(ctx, 4);
            if (hipMalloc(&ctx->d_big_sub, sub_bytes) != hipSuccess) { (void)hipGetLastError(); ctx->d_big_sub = NULL; break; }
            ctx->big_sub_bytes = sub_bytes;
        }
        const HufSubIndex sub = sub_index_view(ctx->d_big_sub, block_len, block_len);
        uint64_t o1;
        if (leaf >= 0) {
            /* one 0 bit per symbol: nothing to find out */
            o1 = pay_off + ((block_len + 7) >> 3);
            if (o1 > avail) break;
            const uint64_t h_offs[2] = {pos, o1};
            HIP_OK(ctx, hipMemcpyAsync(d_offs, h_offs, sizeof(h_offs), hipMemcpyHostToDevice, s));
            HIP_OK(ctx, hipStreamSynchronize(s));
        } else {
            /* an encoder-made payload has at most 9 bits per symbol (8 + the wrap root's) */
            uint64_t max_bits = pay_bytes * 8;
            if (max_bits > 9 * block_len + 64) max_bits = 9 * block_len + 64;
            const uint64_t nlanes = (max_bits + SPEC_LANE_BITS - 1) / SPEC_LANE_BITS;
            if (nlanes == 0) break;
            if (nlanes > ctx->big_lanes) {
                free_big_ws(ctx, 1);
                const uint64_t cap = nlanes + nlanes / 8 + 16;
                if (hipMalloc((void **)&ctx->d_big_entry, cap * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_exit, cap * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_pre, (cap + 1) * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_wgpre, (cap / DEC_THREADS + 4) * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_wgscratch, (cap / DEC_THREADS + 4) * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_first_pos, cap * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_first_g, cap * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_last_pos, cap * sizeof(uint64_t)) != hipSuccess ||
                    hipMalloc((void **)&ctx->d_big_cnt, cap * sizeof(uint32_t)) != hipSuccess) {
                    (void)hipGetLastError();
                    free_big_ws(ctx, 1);
                    break;
                }
                ctx->big_lanes = cap;
            }
            SpecJob j;
            j.tree = st + pos + HUF_HEADER_FIXED;
            j.tree_len = (int)tl;
            j.pay = st + pay_off;
            j.pay_bytes = pay_bytes;
            j.max_bits = max_bits;
            j.block_len = block_len;
            j.nlanes = nlanes;
            j.entry = ctx->d_big_entry;
            j.exitp = ctx->d_big_exit;
            j.cnt = ctx->d_big_cnt;
            j.pre = ctx->d_big_pre;
            j.wg_pre = ctx->d_big_wgpre;
            j.first_pos = ctx->d_big_first_pos;
            j.first_g = ctx->d_big_first_g;
            j.last_pos = ctx->d_big_last_pos;
            j.status = d_status;
            const unsigned lane_wgs = (unsigned)((nlanes + DEC_THREADS - 1) / DEC_THREADS);
            spec_scan_kernel<DEC_THREADS><<<dim3(lane_wgs), dim3(DEC_THREADS), 0, s>>>(j, sub.lens);
            bool chain_ok = false;
            for (int attempt = 0; attempt < 2; attempt++) {
                spec_prefix_kernel<SCAN_THREADS><<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(j, (uint64_t)lane_wgs, ctx->d_big_wgscratch);
                spec_mark_kernel<DEC_THREADS><<<dim3(lane_wgs), dim3(DEC_THREADS), 0, s>>>(j, sub);
                spec_groups_kernel<<<dim3((unsigned)((nlanes + 8 + 255) / 256)), dim3(256), 0, s>>>(j, sub, pos, pay_off, d_offs);
                HIP_OK(ctx, hipGetLastError());
                HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, d_status, SPEC_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                HIP_OK(ctx, hipStreamSynchronize(s));
                if (getenv("HUF_GPU_TRACE"))
                    fprintf(stderr, "big block at %llu: attempt %d chain %llu fail %llu found %llu end_bits %llu lanes %llu\n", (unsigned long long)pos, attempt,
                            (unsigned long long)ctx->h_result[SPEC_CHAIN], (unsigned long long)ctx->h_result[SPEC_FAIL],
                            (unsigned long long)ctx->h_result[SPEC_FOUND], (unsigned long long)ctx->h_result[SPEC_END_BITS], (unsigned long long)nlanes);
                if (!ctx->h_result[SPEC_CHAIN] || ctx->h_result[SPEC_FAIL] || attempt == 1) {
                    chain_ok = !ctx->h_result[SPEC_CHAIN] && !ctx->h_result[SPEC_SHORT];
                    break;
                }
                /* some share did not fall into step before its first bit (a run of one byte value is
                 * a periodic bit string: a decoder can lock onto it one bit off): mend the chain, one
                 * share further per round, then sum and mark again.  A run of more than
                 * SPEC_REPAIR_ROUNDS shares (512 KiB of payload) is left to the general path. */
                bool mended = false;
                for (int round = 0; round < SPEC_REPAIR_ROUNDS && !mended; round += SPEC_REPAIR_BATCH) {
                    HIP_OK(ctx, hipMemsetAsync(d_status + SPEC_REPAIRED, 0, sizeof(uint64_t), s));
                    for (int k = 0; k < SPEC_REPAIR_BATCH; k++)      /* (a round that finds nothing to mend costs a few microseconds) */
                        spec_repair_kernel<DEC_THREADS><<<dim3(lane_wgs), dim3(DEC_THREADS), 0, s>>>(j);
                    HIP_OK(ctx, hipMemcpyAsync(ctx->h_result + SPEC_REPAIRED, d_status + SPEC_REPAIRED, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                    HIP_OK(ctx, hipStreamSynchronize(s));
                    mended = ctx->h_result[SPEC_REPAIRED] == 0;
                    if (getenv("HUF_GPU_TRACE")) fprintf(stderr, "  repair rounds %d..: %llu shares\n", round, (unsigned long long)ctx->h_result[SPEC_REPAIRED]);
                }
                if (!mended) break;
                HIP_OK(ctx, hipMemsetAsync(d_status + SPEC_CHAIN, 0, sizeof(uint64_t), s));
                HIP_OK(ctx, hipMemsetAsync(d_status + SPEC_SHORT, 0, sizeof(uint64_t), s));
                HIP_OK(ctx, hipMemsetAsync(d_status + SPEC_FOUND, 0, sizeof(uint64_t), s));
                spec_sum_kernel<DEC_THREADS><<<dim3(lane_wgs), dim3(DEC_THREADS), 0, s>>>(j);
            }
            if (!chain_ok) break;
            if (ctx->h_result[SPEC_FAIL] || !ctx->h_result[SPEC_FOUND]) break;
            o1 = pay_off + ((ctx->h_result[SPEC_END_BITS] + 7) >> 3);
            if (o1 > avail) break;
        }
        uint64_t got = 0;
        const int err = decode_impl(ctx, st, o1, d_offs, 1, &sub, block_len, out + rawpos, out_cap - rawpos, flags, &got, stream);
        if (getenv("HUF_GPU_TRACE")) {
            uint32_t nfix = 0;
            (void)hipMemcpy(&nfix, ctx->d_fix_count, sizeof(nfix), hipMemcpyDeviceToHost);
            fprintf(stderr, "big block at %llu: decode err %d, %llu bytes, blocks decoded again without the sub-index: %u\n", (unsigned long long)pos, err,
                    (unsigned long long)got, nfix);
        }
        if (err != HUFE_OK || got != block_len) break;   /* the general path decodes it again and says what is wrong */
        pos = o1;
        rawpos += block_len;
    }
    *pos_io = pos;
    *rawpos_io = rawpos;
    return HUFE_OK;
}

static int decode_stream_general(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length,
                                 void *d_out, uint64_t out_cap, uint32_t flags, uint64_t *raw_len,
                                 uint64_t *consumed, void *stream);

static int discover_chain(hufgpu_ctx_t *ctx, const uint8_t *st, uint64_t avail, uint64_t length, uint64_t scan_len, int max_tree,
                          uint8_t *out, uint64_t out_cap, hipStream_t s, uint64_t *m_out, uint64_t *resume_out,
                          bool *complete_out, uint64_t *in_place_out);

/* The block index of a raw stream without decoding it into anything: see include/huffman_gpu.h. */
#ifdef DFAST_DEBUG
extern "C" int hufgpu_debug_dfast(unsigned long long *out16, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(hufgpu::g_dfast_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(hufgpu::g_dfast_dbg), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int hufgpu_block_index(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length, uint32_t flags,
                                  const uint64_t **d_index, uint64_t *nblocks, uint64_t *consumed, void *stream)
{
    if (!ctx || !d_index || !nblocks || !consumed) return HUFE_ARGUMENT;
    *d_index = NULL; *nblocks = 0; *consumed = 0;
    if (length == 0) return HUFE_OK;
    if (!d_stream || ((uintptr_t)d_stream & 15u)) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int max_tree = (flags & HUFGPU_RELAXED_TREE) ? HUF_TREE_MAX : HUF_TREE_STRICT;
    const uint64_t scan_len = length < avail ? length : avail;
    if (scan_len < 4096) return HUFE_OK;
    uint64_t m = 0, resume = 0, in_place = ~0ull;
    bool complete = false;
    const int rc = discover_chain(ctx, (const uint8_t *)d_stream, avail, length, scan_len, max_tree, NULL, 0, s, &m, &resume,
                                  &complete, &in_place);
    if (rc != HUFE_OK) return rc;
    if (m == 0) return HUFE_OK;
    *d_index = ctx->d_chain;
    *nblocks = m;
    *consumed = resume;
    return HUFE_OK;
}

extern "C" int hufgpu_decode_stream(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length,
                                    void *d_out, uint64_t out_cap, uint32_t flags, uint64_t *raw_len,
                                    uint64_t *consumed, void *stream)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (raw_len) *raw_len = 0;
    if (consumed) *consumed = 0;
    if (length == 0) return HUFE_OK;                  /* src/decoder.c:218 */
    if ((!d_stream && avail) || (!d_out && out_cap)) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    uint64_t pos = 0, rawpos = 0;
    if (!(flags & HUFGPU_SEQUENTIAL) && avail >= HUF_BIG_BLOCK / 8) {
        const int rc = decode_big_blocks(ctx, (const uint8_t *)d_stream, avail, length, (uint8_t *)d_out, out_cap, flags,
                                         pick_stream(ctx, stream), stream, &pos, &rawpos);
        if (rc != HUFE_OK) return rc;
    }
    ctx->complete_used = pos;
    ctx->complete_raw = rawpos;
    if (pos >= length) {
        if (raw_len) *raw_len = rawpos;
        if (consumed) *consumed = pos;
        return HUFE_OK;
    }
    uint64_t raw2 = 0, used2 = 0;
    const int err = decode_stream_general(ctx, (const uint8_t *)d_stream + pos, avail - pos, length - pos,
                                          (uint8_t *)d_out + rawpos, out_cap - rawpos, flags, &raw2, &used2, stream);
    /* the general path reports ITS complete blocks (0 / 0 when it returned before decoding anything): the
     * totals are formed here, in one place */
    ctx->complete_used = pos + ctx->complete_used;
    ctx->complete_raw = rawpos + ctx->complete_raw;
    if (raw_len) *raw_len = rawpos + raw2;
    if (consumed) *consumed = pos + used2;
    return err;
}

/* The block chain of a raw stream (kernels/discover.hpp): candidates, probes, links, walk.  On return
 * ctx->d_chain holds the header offsets of the *m blocks the walk validated (+ the offset behind them),
 * *resume = the stream offset behind the validated blocks, *complete = the chain ends the stream exactly
 * as src/decoder.c:218 would, *in_place = output bytes the probes already put where they belong (~0: none;
 * only when `out` has room for every candidate).  *m = 0: nothing validated. */
static int discover_chain(hufgpu_ctx_t *ctx, const uint8_t *st, uint64_t avail, uint64_t length, uint64_t scan_len, int max_tree,
                          uint8_t *out, uint64_t out_cap, hipStream_t s, uint64_t *m_out, uint64_t *resume_out,
                          bool *complete_out, uint64_t *in_place_out)
{
    *m_out = 0; *resume_out = 0; *complete_out = false; *in_place_out = ~0ull;
    const uint64_t nwg = (scan_len + DISC_CHUNK - 1) / DISC_CHUNK;
    const uint64_t ngroups = (nwg + DISC_SCAN_GROUP - 1) / DISC_SCAN_GROUP;
    if (nwg > ctx->disc_wgs) {
        HIP_OK(ctx, hipStreamSynchronize(s));
        free_disc_ws(ctx, 1);
        const uint64_t cap = nwg + nwg / 8 + 16;
        const uint64_t gcap = (cap + DISC_SCAN_GROUP - 1) / DISC_SCAN_GROUP + 1;
        HIP_OK(ctx, hipMalloc((void **)&ctx->d_wg_counts, cap * sizeof(uint32_t)));
        HIP_OK(ctx, hipMalloc((void **)&ctx->d_wg_base, (cap + 1 + 2 * gcap) * sizeof(uint64_t)));     /* local sums, then the groups' bases and totals */
        HIP_OK(ctx, hipMalloc((void **)&ctx->d_disc_masks, cap * DISC_THREADS * sizeof(uint64_t)));
        HIP_OK(ctx, hipMalloc((void **)&ctx->d_disc_slots, cap * DISC_SLOTS * sizeof(DiscSlot)));
        ctx->disc_wgs = cap;
    }
    uint64_t *const group_base = ctx->d_wg_base + ctx->disc_wgs + 1;
    uint64_t *const group_total = group_base + (ctx->disc_wgs + DISC_SCAN_GROUP - 1) / DISC_SCAN_GROUP + 1;
    /* Round 6: ONE wait per call.  Everything that needs the number of candidates - the probes' launch, the sums, the links,
     * the walk - reads it on the device (ctx->d_walk, DISC_NCAND) and is launched as wide as the candidate arrays are:
     * surplus workgroups leave at once.  Only when there are no arrays yet (the context's first raw stream), or when the
     * stream turns out to hold more candidates than they take (the walk's result says so), does the host wait for the
     * count, make room and go again - what every call did until round 5. */
    for (int attempt = 0; attempt < 2; attempt++) {
        HIP_OK(ctx, hipMemsetAsync(ctx->d_walk, 0, DISC_WORDS * sizeof(uint64_t), s));
        discover_kernel<<<dim3((unsigned)nwg), dim3(DISC_THREADS), 0, s>>>(st, avail, scan_len, max_tree, ctx->d_wg_counts, (DiscSlot *)ctx->d_disc_slots, ctx->d_disc_masks);
        scan_counts_kernel<SCAN_THREADS><<<dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s>>>(ctx->d_wg_counts, nwg, ctx->d_wg_base, group_base, group_total, ctx->d_walk, ctx->disc_cands);
        HIP_OK(ctx, hipGetLastError());
        if (ctx->disc_cands == 0 || attempt == 1) {
            HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_walk + DISC_FOUND, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIP_OK(ctx, hipStreamSynchronize(s));
            const uint64_t found = ctx->h_result[0];
            if (found == 0 || found >= 0x7fffffffull) return HUFE_OK;
            if (found > ctx->disc_cands) {
                free_disc_ws(ctx, 2);
                const uint64_t cap = found + found / 8 + 16;
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_cand, cap * sizeof(uint64_t)));
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_cand_end, cap * sizeof(uint64_t)));
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_chain, (cap + 1) * sizeof(uint64_t)));
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_cand_status, cap * sizeof(int32_t)));
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_nxt, cap * sizeof(uint32_t)));
                HIP_OK(ctx, hipMalloc((void **)&ctx->d_spec_off, (cap + 1) * sizeof(uint64_t)));
                ctx->disc_cands = cap;
            }
            /* (the count kernel clamped DISC_NCAND to the capacity it was given: all of them now; h_result[0] is pinned and not
             *  written again before this copy has run - the next one into it is behind it on the stream) */
            HIP_OK(ctx, hipMemcpyAsync(ctx->d_walk + DISC_NCAND, ctx->h_result, sizeof(uint64_t), hipMemcpyHostToDevice, s));
        }
        const uint64_t width = ctx->disc_cands;                          /* launches are as wide as the arrays */
        /* (the candidates' block_len fields pass through d_cand_end, which the probes then overwrite with the ends) */
        place_cands_kernel<<<dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, s>>>(st, ctx->d_wg_counts, nwg, ctx->d_wg_base, group_base, (const DiscSlot *)ctx->d_disc_slots, ctx->d_disc_masks, ctx->d_cand, ctx->d_cand_end, width);
        cand_lens_kernel<SCAN_THREADS><<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(ctx->d_cand_end, ctx->d_walk, ctx->d_spec_off);
        /* (the list of candidates for the exact decoder lives in d_nxt, which link_kernel writes behind the probes; its count in DISC_REDO) */
        probe_kernel<DEC_THREADS><<<dim3((unsigned)width), dim3(DEC_THREADS), 0, s>>>(st, avail, ctx->d_cand, ctx->d_cand_end, ctx->d_cand_status, ctx->d_spec_off, out, out_cap, ctx->d_nxt, ctx->d_walk);
        /* (two forms, each at the lean probe's register budget; the one whose mode it is not leaves at once.  Count-only - every
         *  candidate on the list: hufgpu_block_index - takes a workgroup per candidate) */
        const unsigned exact_grid = (unsigned)(width < 1024 || !out ? width : 1024);
        probe_exact_kernel<DEC_THREADS, true><<<dim3(exact_grid), dim3(DEC_THREADS), 0, s>>>(st, avail, ctx->d_cand, ctx->d_cand_end, ctx->d_cand_status, ctx->d_spec_off, out, out_cap, ctx->d_nxt, ctx->d_walk);
        probe_exact_kernel<DEC_THREADS, false><<<dim3(exact_grid), dim3(DEC_THREADS), 0, s>>>(st, avail, ctx->d_cand, ctx->d_cand_end, ctx->d_cand_status, ctx->d_spec_off, out, out_cap, ctx->d_nxt, ctx->d_walk);
        link_kernel<<<dim3((unsigned)((width + 255) / 256)), dim3(256), 0, s>>>(ctx->d_cand, ctx->d_cand_end, ctx->d_cand_status, ctx->d_walk, length, ctx->d_nxt);
        walk_kernel<<<dim3(1), dim3(WALK_THREADS), 0, s>>>(ctx->d_cand, ctx->d_cand_end, ctx->d_nxt, ctx->d_chain, ctx->d_walk, ctx->d_spec_off, out_cap);
        HIP_OK(ctx, hipGetLastError());
        HIP_OK(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_walk, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_OK(ctx, hipStreamSynchronize(s));
        if (ctx->h_result[DISC_FOUND] > width) continue;                 /* more candidates than the arrays took: once more, with room */
        *m_out = ctx->h_result[0];
        *in_place_out = ctx->h_result[4];   /* bytes the probe already decoded into `out` for these m blocks */
        *complete_out = ctx->h_result[2] != 0;
        *resume_out = *complete_out ? ctx->h_result[3] : ctx->h_result[1];
        return HUFE_OK;
    }
    return HUFE_OK;
}

static int decode_stream_general(hufgpu_ctx_t *ctx, const void *d_stream, uint64_t avail, uint64_t length,
                                 void *d_out, uint64_t out_cap, uint32_t flags, uint64_t *raw_len,
                                 uint64_t *consumed, void *stream)
{
    hipStream_t s = pick_stream(ctx, stream);
    const int max_tree = (flags & HUFGPU_RELAXED_TREE) ? HUF_TREE_MAX : HUF_TREE_STRICT;
    const uint8_t *st = (const uint8_t *)d_stream;
    uint8_t *out = (uint8_t *)d_out;
    uint64_t raw = 0, used = 0;
    int err = HUFE_OK;
    ctx->complete_used = 0;              /* also what an early return (a failed HIP call) leaves behind */
    ctx->complete_raw = 0;

    /* ---- parallel path: discover the block chain, decode the validated prefix ---- */
    uint64_t prefix_raw = 0, resume = 0;
    bool complete = false;
    const uint64_t scan_len = length < avail ? length : avail;
    /* (below 64 KiB of stream the in-order chain is the faster of the two: one launch, 50-60 us a call where the discovery's
     *  launches and its two host round trips take 100-130 - tools/time_stream_small.py) */
    const bool try_parallel = (((uintptr_t)st & 15u) == 0) && scan_len >= 65536 && !(flags & HUFGPU_SEQUENTIAL);
    if (try_parallel) {
        uint64_t m = 0, in_place = ~0ull;
        const int drc = discover_chain(ctx, st, avail, length, scan_len, max_tree, out, out_cap, s, &m, &resume, &complete, &in_place);
        if (drc != HUFE_OK) return drc;
        {
            if (m > 0 && in_place != ~0ull) {
                prefix_raw = in_place;                 /* every candidate was a block: nothing to decode again */
            } else if (m > 0) {
                err = hufgpu_decode(ctx, st, resume, ctx->d_chain, m, out, out_cap, flags, &prefix_raw, stream);
                if (err != HUFE_OK) {                  /* cannot happen for probed blocks except for lack of room */
                    if (err == HUFE_MEMORY) { if (raw_len) *raw_len = prefix_raw; return err; }
                    prefix_raw = 0; resume = 0; complete = false;   /* start over, sequentially */
                }
            } else {
                resume = 0; complete = false;
            }
        }
    }
    ctx->complete_used = 0;
    ctx->complete_raw = 0;
    if (complete) {
        raw = prefix_raw;
        used = resume;
        err = HUFE_OK;
        ctx->complete_used = used;
        ctx->complete_raw = raw;
    } else {
        /* ---- exact sequential decoder for what is left (all of it when nothing was validated) ---- */
        uint64_t raw2 = 0, used2 = 0;
        STAGE_BEGIN(ctx, s, PROF_DECODE);
        uint64_t good_used = 0, good_raw = 0;
        err = decode_chain(ctx, st + resume, avail - resume, length - resume, out + prefix_raw,
                           out_cap - prefix_raw, max_tree, s, &raw2, &used2, &good_used, &good_raw);
        STAGE_MARK(ctx, s);
        raw = prefix_raw + raw2;
        used = resume + used2;
        ctx->complete_used = resume + good_used;
        ctx->complete_raw = prefix_raw + good_raw;
    }
    if (raw_len) *raw_len = raw;
    if (consumed) *consumed = used;
    if (err == HUFE_ARGUMENT) set_err(ctx, "a block is longer than the kernels support");
    if (err == HUFE_MEMORY) set_err(ctx, "output buffer too small");
    return err;
}

extern "C" int hufgpu_decode_stream_complete(hufgpu_ctx_t *ctx, uint64_t *raw_len, uint64_t *consumed)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (raw_len) *raw_len = ctx->complete_raw;
    if (consumed) *consumed = ctx->complete_used;
    return HUFE_OK;
}

extern "C" int hufgpu_fill(hufgpu_ctx_t *ctx, void *d_out, uint64_t n, int kind, uint64_t seed,
                           uint64_t first, void *stream)
{
    if (!ctx || (!d_out && n) || kind < 0 || kind > 3) return HUFE_ARGUMENT;
    if (n == 0) return HUFE_OK;
    if (kind == 1 && (first & 7)) {
        set_err(ctx, "uniform256 shards must start on an 8-byte boundary");
        return HUFE_ARGUMENT;
    }
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    fill_kernel<<<dim3(4096), dim3(256), 0, s>>>((uint8_t *)d_out, n, kind, seed, first, ctx->d_zipf);
    HIP_OK(ctx, hipGetLastError());
    return HUFE_OK;
}

extern "C" int hufgpu_ctx_device(const hufgpu_ctx_t *ctx) { return ctx ? ctx->device : -1; }

extern "C" int hufgpu_malloc(hufgpu_ctx_t *ctx, void **d_ptr, uint64_t bytes)
{
    if (!ctx || !d_ptr) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err(ctx, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return HUFE_MEMORY;
    }
    return HUFE_OK;
}

extern "C" int hufgpu_free(hufgpu_ctx_t *ctx, void *d_ptr)
{
    if (!ctx) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipFree(d_ptr));
    return HUFE_OK;
}

extern "C" int hufgpu_memcpy_h2d(hufgpu_ctx_t *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (!bytes) return HUFE_OK;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return HUFE_OK;
}

extern "C" int hufgpu_memcpy_d2h(hufgpu_ctx_t *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (!bytes) return HUFE_OK;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return HUFE_OK;
}

extern "C" int hufgpu_memcpy_d2d(hufgpu_ctx_t *ctx, void *d_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx) return HUFE_ARGUMENT;
    if (!bytes) return HUFE_OK;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return HUFE_OK;
}

extern "C" int hufgpu_synchronize(hufgpu_ctx_t *ctx)
{
    if (!ctx) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return HUFE_OK;
}

#ifdef DEC_PHASE_PROF
/* diagnostic builds only: cycle sums of the decode phases (thread 0 of every workgroup) */
extern "C" int hufgpu_debug_phase_cycles(hufgpu_ctx_t *ctx, unsigned long long *out16, int reset)
{
    if (!ctx || !out16) return HUFE_ARGUMENT;
    HIP_OK(ctx, hipDeviceSynchronize());
    HIP_OK(ctx, hipMemcpyFromSymbol(out16, HIP_SYMBOL(hufgpu::g_dec_prof), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        HIP_OK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(hufgpu::g_dec_prof), z, sizeof(z)));
    }
    return HUFE_OK;
}
#endif
