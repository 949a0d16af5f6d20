/*
 * hufgpu_sharded.hip - one logical input over the GPUs of a node, behind the C ABI (include/huffman_gpu.h,
 * SURVEY.md §8e): blocks are independent (src/encoder.c:288-374 resets everything between blocks), so rank r of G
 * owns a contiguous range of ceil(nblocks / G) blocks and the codec itself needs no collective.  What moves is whole
 * shards between a root and the ranks - RCCL has neither scatterv nor gatherv, so each movement is ONE group of
 * ncclSend/ncclRecv of exactly-sized buffers to computed offsets - and a few control words (all-gathers of one or two
 * uint64 a rank: who is ready, how long every compressed shard is, who decoded what).
 *
 * RCCL is not linked: its entry points are looked up with dlopen (HUF_GPU_RCCL_LIB, else librccl.so.1 - inside a
 * PyTorch process that is the copy torch already loaded), so the library loads and every single-GPU entry point works
 * on a machine without it.  Everything here is host code over the public entry points of hufgpu_api.hip plus one
 * eight-line kernel; all of a call's work - RCCL's and the codec's - is enqueued on the shard object's own stream.
 *
 * Nothing in this file has a timeout: a rank that never arrives holds the others inside RCCL.  Callers that must not
 * hang (bench.py) run the call on a thread they can give up on.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/huffman.h"
#include "../../include/huffman_gpu.h"

namespace {

typedef struct { char internal[HUFGPU_SHARD_ID_BYTES]; } rccl_id_t;      /* ncclUniqueId (rccl.h:40-43) */
typedef void *rccl_comm_t;
enum { RCCL_UINT8 = 1, RCCL_UINT64 = 5 };                                 /* ncclDataType_t (rccl.h:459-464) */

struct Rccl {
    void *lib;
    int (*GetUniqueId)(rccl_id_t *);
    int (*CommInitRank)(rccl_comm_t *, int, rccl_id_t, int);
    int (*CommDestroy)(rccl_comm_t);
    int (*CommCount)(rccl_comm_t, int *);
    int (*CommUserRank)(rccl_comm_t, int *);
    int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t);
    int (*Broadcast)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t);
    int (*Send)(const void *, size_t, int, int, rccl_comm_t, hipStream_t);
    int (*Recv)(void *, size_t, int, int, rccl_comm_t, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char *(*GetErrorString)(int);
    char why[256];
};
Rccl g_rccl;
pthread_once_t g_rccl_once = PTHREAD_ONCE_INIT;

void rccl_load(void)
{
    const char *names[4] = {getenv("HUF_GPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; i < 4 && !g_rccl.lib; i++) {
        if (!names[i] || !names[i][0]) continue;
        g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.lib) snprintf(g_rccl.why, sizeof g_rccl.why, "%s", dlerror());
        if (i == 0 && !g_rccl.lib) return;                 /* (a library that was asked for by name: no other is tried) */
    }
    if (!g_rccl.lib) return;
    struct { const char *name; void **to; } syms[] = {
        {"ncclGetUniqueId", (void **)&g_rccl.GetUniqueId}, {"ncclCommInitRank", (void **)&g_rccl.CommInitRank},
        {"ncclCommDestroy", (void **)&g_rccl.CommDestroy}, {"ncclCommCount", (void **)&g_rccl.CommCount},
        {"ncclCommUserRank", (void **)&g_rccl.CommUserRank}, {"ncclAllGather", (void **)&g_rccl.AllGather},
        {"ncclBroadcast", (void **)&g_rccl.Broadcast}, {"ncclSend", (void **)&g_rccl.Send}, {"ncclRecv", (void **)&g_rccl.Recv},
        {"ncclGroupStart", (void **)&g_rccl.GroupStart}, {"ncclGroupEnd", (void **)&g_rccl.GroupEnd},
        {"ncclGetErrorString", (void **)&g_rccl.GetErrorString}};
    for (auto &s : syms) {
        *s.to = dlsym(g_rccl.lib, s.name);
        if (!*s.to) {
            snprintf(g_rccl.why, sizeof g_rccl.why, "%s is missing from the RCCL library", s.name);
            g_rccl.lib = NULL;                              /* (the handle stays open: nothing of it is used) */
            return;
        }
    }
}

const Rccl *rccl(void)
{
    pthread_once(&g_rccl_once, rccl_load);
    return g_rccl.lib ? &g_rccl : NULL;
}

double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* dst[i] = src[i] - sub + add: a shard's block index moved between "from the shard's first byte" and "from the stream's" */
__global__ void shard_rebase_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint64_t n, uint64_t sub, uint64_t add)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] - sub + add;
}

}  // namespace

struct hufgpu_shard {
    hufgpu_ctx_t *ctx;
    rccl_comm_t comm;
    int owns_comm, nranks, rank, device;
    hipStream_t stream;
    char err[512];
    /* this rank's buffers, grown on demand and kept */
    uint8_t *d_raw;   uint64_t raw_cap;      /* my uncompressed shard (not on the root: there it is part of the caller's buffer) */
    uint8_t *d_comp;  uint64_t comp_cap;     /* my compressed shard */
    uint64_t *d_offs; uint64_t offs_cap;     /* its block index, from the shard's first byte (entries) */
    uint64_t *d_stage; uint64_t stage_cap;   /* block indexes on their way (entries) */
    void *d_sub;      uint64_t sub_cap;      /* its sub-index */
    uint64_t *d_words, *h_words;             /* control words: 4 * nranks + 8 on the device, pinned mirror */
    /* the layout of the last hufgpu_encode_sharded (HUFGPU_SHARD_OWN_LAYOUT) */
    int have_layout, enc_root;
    uint64_t enc_total, enc_bs, enc_len;
    uint64_t *enc_lens;                      /* [nranks] */
};

namespace {

int fail(hufgpu_shard_t *sh, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sh->err, sizeof sh->err, fmt, ap);
    va_end(ap);
    return code;
}
#define SH_HIP(sh, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
        return fail(sh, HUF_ERROR_FATAL, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
#define SH_RCCL(sh, call) do { int r_ = (call); if (r_ != 0) \
        return fail(sh, HUF_ERROR_FATAL, "%s failed: %s", #call, rccl()->GetErrorString(r_)); } while (0)
#define SH_HUF(sh, call) do { int r_ = (call); if (r_ != HUF_ERROR_SUCCESS) \
        return fail(sh, r_, "%s failed: %s", #call, hufgpu_last_error(sh->ctx)); } while (0)

void range_of(uint64_t n_total, uint64_t blocksize, int rank, int nranks, uint64_t *lo, uint64_t *hi)
{
    if (n_total == 0) { *lo = *hi = 0; return; }
    const uint64_t bs = blocksize ? blocksize : n_total;
    const uint64_t nblocks = (n_total + bs - 1) / bs, per = (nblocks + (uint64_t)nranks - 1) / (uint64_t)nranks;
    uint64_t b0 = (uint64_t)rank * per, b1;
    if (b0 > nblocks) b0 = nblocks;
    b1 = b0 + per < nblocks ? b0 + per : nblocks;
    *lo = b0 * bs < n_total ? b0 * bs : n_total;
    *hi = b1 * bs < n_total ? b1 * bs : n_total;
}

/* a buffer of at least `need` units of `unit` bytes (a failure is remembered, not returned: the ranks agree on it first) */
template <typename T>
bool grow(hufgpu_shard_t *sh, T **p, uint64_t *cap, uint64_t need, uint64_t unit)
{
    if (need <= *cap && *p) return true;
    if (*p) { hufgpu_free(sh->ctx, *p); *p = NULL; *cap = 0; }
    void *q = NULL;
    if (hufgpu_malloc(sh->ctx, &q, (need ? need : 1) * unit) != HUF_ERROR_SUCCESS) {
        fail(sh, HUF_ERROR_MEMORY_ALLOCATION, "%s", hufgpu_last_error(sh->ctx));
        return false;
    }
    *p = (T *)q;
    *cap = need ? need : 1;
    return true;
}

/* every rank contributes `k` words (h_words[0..k)); afterwards h_words[k + r * k + j] is word j of rank r, everywhere */
int all_words(hufgpu_shard_t *sh, int k)
{
    const Rccl *R = rccl();
    const size_t n = (size_t)sh->nranks * (size_t)k;
    SH_HIP(sh, hipMemcpyAsync(sh->d_words, sh->h_words, (size_t)k * 8, hipMemcpyHostToDevice, sh->stream));
    SH_RCCL(sh, R->AllGather(sh->d_words, sh->d_words + k, (size_t)k, RCCL_UINT64, sh->comm, sh->stream));
    SH_HIP(sh, hipMemcpyAsync(sh->h_words + k, sh->d_words + k, n * 8, hipMemcpyDeviceToHost, sh->stream));
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    return HUF_ERROR_SUCCESS;
}

/* ONE group: the root sends piece r of `d_root` (off[r], len[r]) to rank r; rank r receives into d_mine.  (The root's
 * own piece does not move.) */
int scatter(hufgpu_shard_t *sh, int root, const uint8_t *d_root, const uint64_t *off, const uint64_t *len, uint8_t *d_mine)
{
    const Rccl *R = rccl();
    SH_RCCL(sh, R->GroupStart());
    int r_ = 0;
    if (sh->rank == root) {
        for (int r = 0; r < sh->nranks && r_ == 0; r++)
            if (r != root && len[r]) r_ = R->Send(d_root + off[r], (size_t)len[r], RCCL_UINT8, r, sh->comm, sh->stream);
    } else if (len[sh->rank]) {
        r_ = R->Recv(d_mine, (size_t)len[sh->rank], RCCL_UINT8, root, sh->comm, sh->stream);
    }
    const int e_ = R->GroupEnd();
    if (r_ != 0 || e_ != 0) return fail(sh, HUF_ERROR_FATAL, "ncclSend/ncclRecv (scatter) failed: %s", R->GetErrorString(r_ ? r_ : e_));
    return HUF_ERROR_SUCCESS;
}

/* the inverse: rank r's d_mine (len[r] bytes) lands at off[r] of d_root */
int gather(hufgpu_shard_t *sh, int root, uint8_t *d_root, const uint64_t *off, const uint64_t *len, const uint8_t *d_mine)
{
    const Rccl *R = rccl();
    SH_RCCL(sh, R->GroupStart());
    int r_ = 0;
    if (sh->rank == root) {
        for (int r = 0; r < sh->nranks && r_ == 0; r++)
            if (r != root && len[r]) r_ = R->Recv(d_root + off[r], (size_t)len[r], RCCL_UINT8, r, sh->comm, sh->stream);
    } else if (len[sh->rank]) {
        r_ = R->Send(d_mine, (size_t)len[sh->rank], RCCL_UINT8, root, sh->comm, sh->stream);
    }
    const int e_ = R->GroupEnd();
    if (r_ != 0 || e_ != 0) return fail(sh, HUF_ERROR_FATAL, "ncclSend/ncclRecv (gather) failed: %s", R->GetErrorString(r_ ? r_ : e_));
    return HUF_ERROR_SUCCESS;
}

int rebase(hufgpu_shard_t *sh, uint64_t *dst, const uint64_t *src, uint64_t n, uint64_t sub, uint64_t add)
{
    if (!n) return HUF_ERROR_SUCCESS;
    shard_rebase_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sh->stream>>>(dst, src, n, sub, add);
    SH_HIP(sh, hipGetLastError());
    return HUF_ERROR_SUCCESS;
}

struct Legs {
    hufgpu_shard_t *sh;
    double *out, t;
    int i;
    Legs(hufgpu_shard_t *s, double *o) : sh(s), out(o), t(0), i(0) { if (out) { (void)hipStreamSynchronize(sh->stream); t = now_ms(); } }
    void next(void)                              /* the leg ends here (timed legs are synchronised: no overlap between them) */
    {
        if (!out) return;
        (void)hipStreamSynchronize(sh->stream);
        const double n = now_ms();
        out[i++] = n - t;
        t = n;
    }
};

}  // namespace

extern "C" int hufgpu_shard_range(uint64_t n_total, uint64_t blocksize, int rank, int nranks, uint64_t *lo, uint64_t *hi)
{
    if (!lo || !hi || nranks < 1 || rank < 0 || rank >= nranks) return HUF_ERROR_INVALID_ARGUMENT;
    range_of(n_total, blocksize, rank, nranks, lo, hi);
    return HUF_ERROR_SUCCESS;
}

extern "C" int hufgpu_shard_plan_decode(const uint64_t *block_offsets, uint64_t nblocks, int nranks, uint64_t *first_block)
{
    if (!first_block || nranks < 1 || (nblocks && !block_offsets)) return HUF_ERROR_INVALID_ARGUMENT;
    /* rank r takes the blocks whose header lies in its 1/nranks share of the stream's bytes: cut r = the first block
     * whose header lies at or behind ceil(total * r / nranks) */
    first_block[0] = 0;
    const uint64_t total = nblocks ? block_offsets[nblocks] : 0;
    for (int r = 1; r < nranks; r++) {
        const unsigned __int128 want = ((unsigned __int128)total * (unsigned)r + (unsigned)nranks - 1) / (unsigned)nranks;
        uint64_t lo = first_block[r - 1], hi = nblocks;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if ((unsigned __int128)block_offsets[mid] < want) lo = mid + 1; else hi = mid;
        }
        first_block[r] = lo;
    }
    first_block[nranks] = nblocks;
    return HUF_ERROR_SUCCESS;
}

extern "C" int hufgpu_shard_unique_id(void *id)
{
    const Rccl *R = rccl();
    if (!id) return HUF_ERROR_INVALID_ARGUMENT;
    if (!R) return HUF_ERROR_FATAL;
    return R->GetUniqueId((rccl_id_t *)id) == 0 ? HUF_ERROR_SUCCESS : HUF_ERROR_FATAL;
}

extern "C" const char *hufgpu_shard_last_error(const hufgpu_shard_t *sh)
{
    if (!sh) return rccl() ? "" : (g_rccl.why[0] ? g_rccl.why : "the RCCL library could not be loaded");
    return sh->err;
}

extern "C" int hufgpu_shard_destroy(hufgpu_shard_t *sh)
{
    if (!sh) return HUF_ERROR_SUCCESS;
    (void)hipSetDevice(sh->device);
    if (sh->stream) { (void)hipStreamSynchronize(sh->stream); }
    if (sh->owns_comm && sh->comm && rccl()) (void)rccl()->CommDestroy(sh->comm);
    if (sh->d_raw) hufgpu_free(sh->ctx, sh->d_raw);
    if (sh->d_comp) hufgpu_free(sh->ctx, sh->d_comp);
    if (sh->d_offs) hufgpu_free(sh->ctx, sh->d_offs);
    if (sh->d_stage) hufgpu_free(sh->ctx, sh->d_stage);
    if (sh->d_sub) hufgpu_free(sh->ctx, sh->d_sub);
    if (sh->d_words) hufgpu_free(sh->ctx, sh->d_words);
    if (sh->h_words) (void)hipHostFree(sh->h_words);
    if (sh->stream) (void)hipStreamDestroy(sh->stream);
    free(sh->enc_lens);
    free(sh);
    return HUF_ERROR_SUCCESS;
}

extern "C" int hufgpu_shard_create(hufgpu_shard_t **out, hufgpu_ctx_t *ctx, void *nccl_comm, const void *id, int nranks, int rank)
{
    if (!out || !ctx) return HUF_ERROR_INVALID_ARGUMENT;
    *out = NULL;
    const Rccl *R = rccl();
    if (!R) return HUF_ERROR_FATAL;                            /* hufgpu_shard_last_error(NULL) says why */
    if (nccl_comm) {                                           /* the caller's communicator says who we are */
        if (R->CommCount(nccl_comm, &nranks) != 0 || R->CommUserRank(nccl_comm, &rank) != 0) return HUF_ERROR_INVALID_ARGUMENT;
    } else if (!id) {
        return HUF_ERROR_INVALID_ARGUMENT;
    }
    if (nranks < 1 || rank < 0 || rank >= nranks) return HUF_ERROR_INVALID_ARGUMENT;
    hufgpu_shard_t *sh = (hufgpu_shard_t *)calloc(1, sizeof *sh);
    if (!sh) return HUF_ERROR_MEMORY_ALLOCATION;
    sh->ctx = ctx;
    sh->nranks = nranks;
    sh->rank = rank;
    sh->device = hufgpu_ctx_device(ctx);
    sh->enc_lens = (uint64_t *)calloc((size_t)nranks, sizeof(uint64_t));
    int rc = HUF_ERROR_FATAL;
    do {
        if (!sh->enc_lens) { rc = HUF_ERROR_MEMORY_ALLOCATION; break; }
        if (hipSetDevice(sh->device) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&sh->stream, hipStreamNonBlocking) != hipSuccess) break;
        const size_t words = 4 * (size_t)nranks + 8;
        void *p = NULL;
        if (hufgpu_malloc(ctx, &p, words * 8) != HUF_ERROR_SUCCESS) { rc = HUF_ERROR_MEMORY_ALLOCATION; break; }
        sh->d_words = (uint64_t *)p;
        if (hipHostMalloc((void **)&sh->h_words, words * 8, hipHostMallocDefault) != hipSuccess) { rc = HUF_ERROR_MEMORY_ALLOCATION; break; }
        if (nccl_comm) {
            sh->comm = nccl_comm;
        } else {
            rccl_id_t uid;
            memcpy(&uid, id, sizeof uid);
            if (R->CommInitRank(&sh->comm, nranks, uid, rank) != 0) { sh->comm = NULL; break; }
            sh->owns_comm = 1;
        }
        *out = sh;
        return HUF_ERROR_SUCCESS;
    } while (0);
    (void)hipGetLastError();
    hufgpu_shard_destroy(sh);
    return rc;
}

extern "C" int hufgpu_shard_info(const hufgpu_shard_t *sh, int *nranks, int *rank)
{
    if (!sh) return HUF_ERROR_INVALID_ARGUMENT;
    if (nranks) *nranks = sh->nranks;
    if (rank) *rank = sh->rank;
    return HUF_ERROR_SUCCESS;
}

/*
 * legs_ms (optional, 4 doubles): scatter of the input, encode, the size all-gather, gather of the stream (+ index).
 */
extern "C" int hufgpu_encode_sharded(hufgpu_shard_t *sh, int root, const void *d_in, uint64_t n_total, uint64_t blocksize,
                                     uint32_t flags, void *d_stream, uint64_t stream_cap, uint64_t *d_block_offsets,
                                     uint64_t *stream_len, uint64_t *shard_lens, double *legs_ms)
{
    if (!sh) return HUF_ERROR_INVALID_ARGUMENT;
    const int G = sh->nranks, me = sh->rank;
    if (root < 0 || root >= G) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "root %d of %d ranks", root, G);
    SH_HIP(sh, hipSetDevice(sh->device));
    sh->have_layout = 0;
    uint64_t lo[64], hi[64];
    if (G > 64) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "more than 64 ranks");
    for (int r = 0; r < G; r++) range_of(n_total, blocksize, r, G, &lo[r], &hi[r]);
    const uint64_t n = hi[me] - lo[me], nb = hufgpu_block_count(n, blocksize);
    const uint64_t bs_total = hufgpu_block_count(n_total, blocksize);

    /* 1. buffers; then everybody says whether it is ready (a rank that fails alone must not leave the others in a collective) */
    int ready = 1, why = HUF_ERROR_SUCCESS;
    if (me == root && n_total && (!d_in || !d_stream || stream_cap < hufgpu_encode_bound(n_total, blocksize) ||
                                  ((flags & HUFGPU_SHARD_INDEX) && !d_block_offsets))) {
        ready = 0; why = HUF_ERROR_INVALID_ARGUMENT;
        fail(sh, why, "the root needs the input, room for hufgpu_encode_bound(n_total) bytes of stream and, with HUFGPU_SHARD_INDEX, for the block index");
    }
    if (ready && !((me == root || grow(sh, &sh->d_raw, &sh->raw_cap, n, 1)) &&
                   grow(sh, &sh->d_comp, &sh->comp_cap, hufgpu_encode_bound(n, blocksize), 1) &&
                   grow(sh, &sh->d_offs, &sh->offs_cap, nb + 1, 8) &&
                   grow(sh, &sh->d_stage, &sh->stage_cap, nb + 1, 8) &&
                   grow(sh, &sh->d_sub, &sh->sub_cap, hufgpu_sub_index_bytes(n, blocksize), 1))) {
        ready = 0; why = HUF_ERROR_MEMORY_ALLOCATION;
    }
    sh->h_words[0] = (uint64_t)ready;
    { const int rc = all_words(sh, 1); if (rc) return rc; }
    for (int r = 0; r < G; r++)
        if (!sh->h_words[1 + r]) return ready ? fail(sh, HUF_ERROR_FATAL, "rank %d is not ready", r) : why;

    Legs legs(sh, legs_ms);
    /* 2. the input shards leave the root */
    uint64_t len[64];
    for (int r = 0; r < G; r++) len[r] = hi[r] - lo[r];
    { const int rc = scatter(sh, root, (const uint8_t *)d_in, lo, len, sh->d_raw); if (rc) return rc; }
    legs.next();
    /* 3. every rank encodes its blocks (stream and side tables stay here: HUFGPU_SHARD_OWN_LAYOUT decodes with them) */
    const uint8_t *src = me == root ? (const uint8_t *)d_in + lo[me] : sh->d_raw;
    if (n) SH_HUF(sh, hufgpu_encode_sub(sh->ctx, src, n, blocksize, sh->d_comp, sh->comp_cap, sh->d_offs, sh->d_sub, NULL, sh->stream));
    legs.next();
    /* 4. how long every shard is: one word a rank (the length stands at the end of the shard's block index) */
    if (n) SH_HIP(sh, hipMemcpyAsync(sh->d_words, sh->d_offs + nb, 8, hipMemcpyDeviceToDevice, sh->stream));
    else SH_HIP(sh, hipMemsetAsync(sh->d_words, 0, 8, sh->stream));
    SH_RCCL(sh, rccl()->AllGather(sh->d_words, sh->d_words + 1, 1, RCCL_UINT64, sh->comm, sh->stream));
    SH_HIP(sh, hipMemcpyAsync(sh->h_words + 1, sh->d_words + 1, (size_t)G * 8, hipMemcpyDeviceToHost, sh->stream));
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    uint64_t start[65];
    start[0] = 0;
    for (int r = 0; r < G; r++) { sh->enc_lens[r] = sh->h_words[1 + r]; start[r + 1] = start[r] + sh->enc_lens[r]; }
    if (start[G] > hufgpu_encode_bound(n_total, blocksize)) return fail(sh, HUF_ERROR_FATAL, "the shards are longer than the bound of the whole");
    legs.next();
    /* 5. the compressed shards to their places in the root's stream: rank order = stream order */
    { const int rc = gather(sh, root, (uint8_t *)d_stream, start, sh->enc_lens, sh->d_comp); if (rc) return rc; }
    if (me == root && sh->enc_lens[me])
        SH_HIP(sh, hipMemcpyAsync((uint8_t *)d_stream + start[me], sh->d_comp, sh->enc_lens[me], hipMemcpyDeviceToDevice, sh->stream));
    if (flags & HUFGPU_SHARD_INDEX) {
        /* the block index of the whole: every shard's entries counted from the stream's first byte, the last entry = the length */
        uint64_t boff[64], blen[64];
        for (int r = 0; r < G; r++) {
            boff[r] = 8 * hufgpu_block_count(lo[r], blocksize ? blocksize : n_total);   /* (whole blocks in front of the shard) */
            blen[r] = 8 * hufgpu_block_count(hi[r] - lo[r], blocksize);
        }
        { const int rc = rebase(sh, sh->d_stage, sh->d_offs, nb, 0, start[me]); if (rc) return rc; }
        { const int rc = gather(sh, root, (uint8_t *)d_block_offsets, boff, blen, (const uint8_t *)sh->d_stage); if (rc) return rc; }
        if (me == root) {
            if (nb) SH_HIP(sh, hipMemcpyAsync((uint8_t *)d_block_offsets + boff[me], sh->d_stage, nb * 8, hipMemcpyDeviceToDevice, sh->stream));
            sh->h_words[0] = start[G];
            SH_HIP(sh, hipMemcpyAsync(d_block_offsets + bs_total, sh->h_words, 8, hipMemcpyHostToDevice, sh->stream));
        }
    }
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    legs.next();
    sh->have_layout = 1;
    sh->enc_root = root;
    sh->enc_total = n_total;
    sh->enc_bs = blocksize;
    sh->enc_len = start[G];
    if (stream_len) *stream_len = start[G];
    if (shard_lens) memcpy(shard_lens, sh->enc_lens, (size_t)G * 8);
    return HUF_ERROR_SUCCESS;
}

/*
 * legs_ms (optional, 4 doubles): the plan (foreign streams: index to the host, broadcast), scatter of the stream, decode
 * + the result all-gather, gather of the output.
 */
extern "C" int hufgpu_decode_sharded(hufgpu_shard_t *sh, int root, const void *d_stream, uint64_t stream_len,
                                     const uint64_t *d_block_offsets, uint64_t n_total, uint64_t blocksize, uint32_t flags,
                                     void *d_out, uint64_t out_cap, uint64_t *raw_len, double *legs_ms)
{
    if (!sh) return HUF_ERROR_INVALID_ARGUMENT;
    const int G = sh->nranks, me = sh->rank;
    if (root < 0 || root >= G) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "root %d of %d ranks", root, G);
    if (G > 64) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "more than 64 ranks");
    SH_HIP(sh, hipSetDevice(sh->device));
    const bool own = (flags & HUFGPU_SHARD_OWN_LAYOUT) != 0;
    const uint32_t dflags = flags & (HUFGPU_RELAXED_TREE);
    const uint64_t bs = blocksize ? blocksize : n_total;
    const uint64_t nblocks = hufgpu_block_count(n_total, blocksize);
    if (raw_len) *raw_len = 0;

    int ready = 1, why = HUF_ERROR_SUCCESS;
    if (own && !(sh->have_layout && sh->enc_root == root && sh->enc_total == n_total && sh->enc_bs == blocksize)) {
        ready = 0; why = HUF_ERROR_INVALID_ARGUMENT;
        fail(sh, why, "HUFGPU_SHARD_OWN_LAYOUT: no hufgpu_encode_sharded of this root, size and block size went before");
    }
    if (ready && me == root && n_total && (!d_stream || !d_out || out_cap < n_total || (!own && !d_block_offsets) ||
                                           (own && stream_len != sh->enc_len))) {
        ready = 0; why = HUF_ERROR_INVALID_ARGUMENT;
        fail(sh, why, "the root needs the stream, its block index (or the layout of this object's last encode) and room for n_total bytes");
    }
    Legs legs(sh, legs_ms);
    /* 1. which blocks a rank decodes: b[r] .. b[r + 1], and where their bytes lie in the stream */
    uint64_t b[65], cstart[65];
    if (own) {
        uint64_t lo, hi;
        cstart[0] = 0;
        for (int r = 0; r < G; r++) {
            range_of(n_total, blocksize, r, G, &lo, &hi);
            b[r] = hufgpu_block_count(lo, bs);
            cstart[r + 1] = cstart[r] + sh->enc_lens[r];
        }
        b[G] = nblocks;
        sh->h_words[0] = (uint64_t)ready;
        { const int rc = all_words(sh, 1); if (rc) return rc; }
        for (int r = 0; r < G; r++)
            if (!sh->h_words[1 + r]) return ready ? fail(sh, HUF_ERROR_FATAL, "rank %d is not ready", r) : why;
    } else {
        /* the root reads the block index, cuts the stream into G shares of about equal BYTES at block borders and tells
         * everybody: words 0..G = first blocks, G+1..2G+1 = their offsets, 2G+2 = ready */
        uint64_t *plan = sh->h_words;
        const int W = 2 * G + 3;
        memset(plan, 0, (size_t)W * 8);
        if (me == root && ready) {
            uint64_t *h_offs = (uint64_t *)malloc((size_t)(nblocks + 1) * 8);
            if (!h_offs) { ready = 0; why = HUF_ERROR_MEMORY_ALLOCATION; }
            else {
                hipError_t e = hipMemcpyAsync(h_offs, d_block_offsets, (size_t)(nblocks + 1) * 8, hipMemcpyDeviceToHost, sh->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(sh->stream);
                if (e != hipSuccess) { ready = 0; why = HUF_ERROR_FATAL; fail(sh, why, "reading the block index failed: %s", hipGetErrorString(e)); }
                else if (h_offs[nblocks] != stream_len) { ready = 0; why = HUF_ERROR_INVALID_ARGUMENT; fail(sh, why, "the block index does not end at the stream's length"); }
                else {
                    hufgpu_shard_plan_decode(h_offs, nblocks, G, plan);
                    for (int r = 0; r <= G; r++) plan[G + 1 + r] = h_offs[plan[r]];
                }
                free(h_offs);
            }
        }
        plan[2 * G + 2] = (uint64_t)ready;
        if (me == root) SH_HIP(sh, hipMemcpyAsync(sh->d_words, plan, (size_t)W * 8, hipMemcpyHostToDevice, sh->stream));
        SH_RCCL(sh, rccl()->Broadcast(sh->d_words, sh->d_words, (size_t)W, RCCL_UINT64, root, sh->comm, sh->stream));
        SH_HIP(sh, hipMemcpyAsync(plan, sh->d_words, (size_t)W * 8, hipMemcpyDeviceToHost, sh->stream));
        SH_HIP(sh, hipStreamSynchronize(sh->stream));
        if (!plan[2 * G + 2]) return me == root ? why : fail(sh, HUF_ERROR_FATAL, "the root is not ready");
        for (int r = 0; r <= G; r++) { b[r] = plan[r]; cstart[r] = plan[G + 1 + r]; }
    }
    const uint64_t nb = b[me + 1] - b[me];
    uint64_t rlo[64], rlen[64], clen[64];
    for (int r = 0; r < G; r++) {
        rlo[r] = b[r] * bs < n_total ? b[r] * bs : n_total;
        const uint64_t rhi = b[r + 1] * bs < n_total ? b[r + 1] * bs : n_total;
        rlen[r] = rhi - rlo[r];
        clen[r] = cstart[r + 1] - cstart[r];
    }
    const uint64_t n = rlen[me];
    /* buffers (a rank that cannot have them says so in the result all-gather: it takes part in every movement until then,
     * into no buffer - nothing is sent to a rank that is not ready, so the readiness goes first for foreign streams too) */
    int mine_ok = 1;
    if (!((me == root || (grow(sh, &sh->d_raw, &sh->raw_cap, n, 1) && grow(sh, &sh->d_comp, &sh->comp_cap, clen[me], 1))) &&
          (own || (grow(sh, &sh->d_offs, &sh->offs_cap, nb + 1, 8) && (me != root || grow(sh, &sh->d_stage, &sh->stage_cap, nblocks + (uint64_t)G + 1, 8))))))
        mine_ok = 0;
    if (!own) {
        sh->h_words[0] = (uint64_t)mine_ok;
        { const int rc = all_words(sh, 1); if (rc) return rc; }
        for (int r = 0; r < G; r++)
            if (!sh->h_words[1 + r]) return mine_ok ? fail(sh, HUF_ERROR_FATAL, "rank %d is out of memory", r) : HUF_ERROR_MEMORY_ALLOCATION;
    } else if (!mine_ok) {
        return HUF_ERROR_MEMORY_ALLOCATION;         /* (own layout: the buffers are the encode's, they are there) */
    }
    legs.next();
    /* 2. the compressed shards (and, for a foreign stream, each one's block index counted from its first byte) */
    { const int rc = scatter(sh, root, (const uint8_t *)d_stream, cstart, clen, sh->d_comp); if (rc) return rc; }
    const uint64_t *my_offs = sh->d_offs;
    if (!own) {
        uint64_t ioff[64], ilen[64];
        if (me == root) {
            uint64_t at = 0;
            for (int r = 0; r < G; r++) {                      /* rank r's entries b[r] .. b[r + 1] inclusive, rebased, one after the other */
                const uint64_t cnt = b[r + 1] - b[r] + 1;
                { const int rc = rebase(sh, sh->d_stage + at, d_block_offsets + b[r], cnt, cstart[r], 0); if (rc) return rc; }
                ioff[r] = at * 8; ilen[r] = cnt * 8;
                at += cnt;
            }
            my_offs = sh->d_stage + ioff[me] / 8;
        } else {
            for (int r = 0; r < G; r++) { ioff[r] = 0; ilen[r] = (b[r + 1] - b[r] + 1) * 8; }
        }
        { const int rc = scatter(sh, root, (const uint8_t *)sh->d_stage, ioff, ilen, (uint8_t *)sh->d_offs); if (rc) return rc; }
    }
    legs.next();
    /* 3. decode; then everybody learns how it went everywhere */
    const uint8_t *src = me == root ? (const uint8_t *)d_stream + cstart[me] : sh->d_comp;
    uint8_t *dst = me == root ? (uint8_t *)d_out + rlo[me] : sh->d_raw;
    uint64_t got = 0;
    int err = HUF_ERROR_SUCCESS;
    if (nb) {
        if (own) err = hufgpu_decode_sub(sh->ctx, src, clen[me], sh->d_offs, n, blocksize, sh->d_sub, dst, n, dflags, &got, sh->stream);
        else err = hufgpu_decode(sh->ctx, src, clen[me], my_offs, nb, dst, n, dflags, &got, sh->stream);
        if (err) fail(sh, err, "decoding blocks %llu..%llu failed: %s", (unsigned long long)b[me], (unsigned long long)b[me + 1], hufgpu_last_error(sh->ctx));
        else if (got != n) { err = HUF_ERROR_READ_WRITE; fail(sh, err, "blocks %llu..%llu hold %llu bytes, not %llu", (unsigned long long)b[me], (unsigned long long)b[me + 1], (unsigned long long)got, (unsigned long long)n); }
    }
    sh->h_words[0] = (uint64_t)err;
    sh->h_words[1] = got;
    { const int rc = all_words(sh, 2); if (rc) return rc; }
    legs.next();
    uint64_t total = 0;
    for (int r = 0; r < G; r++) {                              /* the first error in stream order, as one decoder would report it */
        const int e = (int)sh->h_words[2 + 2 * r];
        total += sh->h_words[2 + 2 * r + 1];
        if (e) {
            if (raw_len) *raw_len = total;                     /* (the blocks in front of the failing shard and what it delivered) */
            return e == err && r == me ? err : fail(sh, e, "rank %d failed to decode its blocks", r);
        }
    }
    /* 4. the output shards to the root */
    { const int rc = gather(sh, root, (uint8_t *)d_out, rlo, rlen, sh->d_raw); if (rc) return rc; }
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    legs.next();
    if (raw_len) *raw_len = total;
    return HUF_ERROR_SUCCESS;
}
