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
 * Round 6: no call can hang.  Every sharded call has a deadline (HUF_GPU_SHARD_TIMEOUT_MS, default 120 000; 0 = none;
 * hufgpu_shard_set_timeout): the call's body runs on a helper thread, inside it every wait for the stream is a poll of
 * hipStreamQuery and ncclCommGetAsyncError against the deadline, and the calling thread waits for the helper against the
 * same deadline - a rank that never arrives holds the others in the stream (the real RCCL: its kernels spin) or in the
 * host call (a transport that blocks there: connection set-up, tests/mock_rccl).  On expiry, or on an asynchronous error of
 * the communicator, the communicator is aborted (ncclCommAbort: pending kernels leave, blocked host calls return) if it is
 * the object's own, the object is marked broken - every later call returns HUF_ERROR_FATAL at once - and the call returns
 * HUF_ERROR_FATAL.  Nothing is re-executed and no process is replaced; what the caller does with a broken group is the
 * caller's business (bench.py: the rank exits non-zero).
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
#include <unistd.h>
#include <vector>

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
    int (*CommAbort)(rccl_comm_t);                       /* optional (every RCCL has them; a stand-in transport may not) */
    int (*CommGetAsyncError)(rccl_comm_t, int *);
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
    g_rccl.CommAbort = (int (*)(rccl_comm_t))dlsym(g_rccl.lib, "ncclCommAbort");
    g_rccl.CommGetAsyncError = (int (*)(rccl_comm_t, int *))dlsym(g_rccl.lib, "ncclCommGetAsyncError");
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
    hipStream_t stream2;                     /* the root's own shard, beside the movements of the others' */
    hipEvent_t ev;
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
    /* deadlines */
    uint32_t timeout_ms;                     /* of every call; 0 = none */
    double deadline;                         /* of the call that is running (now_ms() scale); 0 = none */
    pthread_mutex_t mu;                      /* guards comm / broken / aborted between a call's helper and its caller */
    int broken;                              /* a call timed out or the communicator failed: every later call fails at once */
    int orphans;                             /* helper threads a timed-out call left behind: the object is not freed while one may run */
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

/* the communicator is done for: abort it (ours) so that whatever waits inside it returns, and mark the object */
void break_group(hufgpu_shard_t *sh)
{
    rccl_comm_t victim = NULL;
    pthread_mutex_lock(&sh->mu);
    sh->broken = 1;
    sh->have_layout = 0;
    if (sh->owns_comm && sh->comm) { victim = sh->comm; sh->comm = NULL; sh->owns_comm = 0; }
    pthread_mutex_unlock(&sh->mu);
    if (victim && rccl() && rccl()->CommAbort) (void)rccl()->CommAbort(victim);
}

int timed_out(hufgpu_shard_t *sh, const char *where)
{
    break_group(sh);
    return fail(sh, HUF_ERROR_FATAL, "timed out after %u ms %s: a rank of the group did not arrive (HUF_GPU_SHARD_TIMEOUT_MS); the group is broken", sh->timeout_ms, where);
}

/* Wait for the object's stream: a poll against the call's deadline that also asks the communicator for asynchronous
 * errors (a peer that died) - never a blind hipStreamSynchronize when there is a deadline. */
int wait_stream(hufgpu_shard_t *sh, hipStream_t st)
{
    if (sh->deadline == 0.0) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(sh, HUF_ERROR_FATAL, "hipStreamSynchronize failed: %s", hipGetErrorString(e)); }
        return HUF_ERROR_SUCCESS;
    }
    const Rccl *R = rccl();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return HUF_ERROR_SUCCESS;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); break_group(sh); return fail(sh, HUF_ERROR_FATAL, "hipStreamQuery failed: %s", hipGetErrorString(e)); }
        if ((spins & 31u) == 31u) {
            pthread_mutex_lock(&sh->mu);
            rccl_comm_t c = sh->comm;
            const int broken = sh->broken;
            pthread_mutex_unlock(&sh->mu);
            if (broken) return fail(sh, HUF_ERROR_FATAL, "the group was broken while this call waited");
            int async = 0;
            if (c && R && R->CommGetAsyncError && R->CommGetAsyncError(c, &async) == 0 && async != 0 && async != 7 /* ncclInProgress */) {
                break_group(sh);
                return fail(sh, HUF_ERROR_FATAL, "the communicator reports an asynchronous error: %s", R->GetErrorString(async));
            }
            if (now_ms() > sh->deadline) return timed_out(sh, "waiting for the stream");
        }
        if (spins > 4000u) usleep(50);                 /* (the first ~0.2 ms: a spin - the control all-gathers take tens of microseconds) */
    }
}

#define SH_HIP(sh, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
        return fail(sh, HUF_ERROR_FATAL, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
#define SH_RCCL(sh, call) do { int r_ = (call); if (r_ != 0) { break_group(sh); \
        return fail(sh, HUF_ERROR_FATAL, "%s failed: %s", #call, rccl()->GetErrorString(r_)); } } while (0)
#define SH_WAIT(sh) do { const int w_ = wait_stream(sh, (sh)->stream); if (w_) return w_; } while (0)

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
    SH_WAIT(sh);
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
    if (r_ != 0 || e_ != 0) { break_group(sh); return fail(sh, HUF_ERROR_FATAL, "ncclSend/ncclRecv (scatter) failed: %s", R->GetErrorString(r_ ? r_ : e_)); }
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
    if (r_ != 0 || e_ != 0) { break_group(sh); return fail(sh, HUF_ERROR_FATAL, "ncclSend/ncclRecv (gather) failed: %s", R->GetErrorString(r_ ? r_ : e_)); }
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
    Legs(hufgpu_shard_t *s, double *o) : sh(s), out(o), t(0), i(0) { if (out) { (void)wait_stream(sh, sh->stream); t = now_ms(); } }
    void next(void)                              /* the leg ends here (timed legs are synchronised: no overlap between them) */
    {
        if (!out) return;
        (void)wait_stream(sh, sh->stream);
        const double n = now_ms();
        out[i++] = n - t;
        t = n;
    }
};

/* ---- a call's body on a helper thread, its caller waiting against the deadline ---- */
struct Call {
    hufgpu_shard_t *sh;
    int (*body)(Call *);
    /* arguments (both calls; what a call does not have stays 0) */
    int root;
    const void *d_in;
    uint64_t n_total, blocksize, stream_len, cap;
    uint32_t flags;
    void *d_out;
    uint64_t *d_index;
    const uint64_t *d_index_in;
    int want_legs;
    /* results: the caller's memory is written by the caller, and only by a call that came back in time */
    uint64_t out_len;
    std::vector<uint64_t> lens;
    double legs[4];
    int rc;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int done, refs;
};

void call_release(Call *c)
{
    pthread_mutex_lock(&c->mu);
    const int left = --c->refs;
    pthread_mutex_unlock(&c->mu);
    if (left == 0) { pthread_mutex_destroy(&c->mu); pthread_cond_destroy(&c->cv); delete c; }
}

void *call_main(void *arg)
{
    Call *c = (Call *)arg;
    int rc = hipSetDevice(c->sh->device) == hipSuccess ? c->body(c) : HUF_ERROR_FATAL;
    pthread_mutex_lock(&c->mu);
    c->rc = rc;
    c->done = 1;
    pthread_cond_broadcast(&c->cv);
    pthread_mutex_unlock(&c->mu);
    call_release(c);
    return NULL;
}

/* runs c->body; true = the body came back (c->rc), false = the deadline passed first (the group is broken, the helper is
 * left to itself and frees the call when it returns) */
bool run_call(Call *c)
{
    hufgpu_shard_t *sh = c->sh;
    pthread_mutex_init(&c->mu, NULL);
    pthread_cond_init(&c->cv, NULL);
    c->done = 0;
    if (sh->timeout_ms == 0) {                       /* no deadline: on the caller's thread */
        sh->deadline = 0.0;
        c->refs = 1;
        c->rc = c->body(c);
        return true;
    }
    sh->deadline = now_ms() + (double)sh->timeout_ms;
    c->refs = 2;
    pthread_t th;
    if (pthread_create(&th, NULL, call_main, c) != 0) {
        c->refs = 1;
        c->rc = fail(sh, HUF_ERROR_FATAL, "pthread_create failed");
        return true;
    }
    /* the helper polls the same deadline wherever it waits for the stream; what it cannot poll is a host call that blocks
     * (connection set-up to a rank that is not there): a grace period for the first, then the abort from here */
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    const uint64_t ns = (uint64_t)until.tv_nsec + ((uint64_t)sh->timeout_ms + 500ull) * 1000000ull;
    until.tv_sec += (time_t)(ns / 1000000000ull);
    until.tv_nsec = (long)(ns % 1000000000ull);
    pthread_mutex_lock(&c->mu);
    int w = 0;
    while (!c->done && w == 0) w = pthread_cond_timedwait(&c->cv, &c->mu, &until);
    const int done = c->done;
    pthread_mutex_unlock(&c->mu);
    if (done) {
        pthread_join(th, NULL);
        return true;
    }
    pthread_mutex_lock(&sh->mu);
    sh->orphans++;
    pthread_mutex_unlock(&sh->mu);
    (void)timed_out(sh, "inside the transport");      /* aborts the communicator: the helper's host call returns */
    /* a moment for the helper to come back (it then still is this call's thread to join) */
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += 2;
    pthread_mutex_lock(&c->mu);
    w = 0;
    while (!c->done && w == 0) w = pthread_cond_timedwait(&c->cv, &c->mu, &until);
    const int late = c->done;
    pthread_mutex_unlock(&c->mu);
    if (late) {
        pthread_join(th, NULL);
        pthread_mutex_lock(&sh->mu);
        sh->orphans--;
        pthread_mutex_unlock(&sh->mu);
        /* (the helper's own words - "the all-gather failed" - are the abort's echo: what happened is the deadline) */
        fail(sh, HUF_ERROR_FATAL, "timed out after %u ms inside the transport: a rank of the group did not arrive (HUF_GPU_SHARD_TIMEOUT_MS); the group is broken", sh->timeout_ms);
    } else {
        pthread_detach(th);
    }
    return false;
}

int encode_body(Call *c);
int decode_body(Call *c);

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

extern "C" int hufgpu_shard_set_timeout(hufgpu_shard_t *sh, uint32_t timeout_ms)
{
    if (!sh) return HUF_ERROR_INVALID_ARGUMENT;
    sh->timeout_ms = timeout_ms;
    return HUF_ERROR_SUCCESS;
}

extern "C" int hufgpu_shard_destroy(hufgpu_shard_t *sh)
{
    if (!sh) return HUF_ERROR_SUCCESS;
    (void)hipSetDevice(sh->device);
    pthread_mutex_lock(&sh->mu);
    const int orphans = sh->orphans;
    pthread_mutex_unlock(&sh->mu);
    if (orphans) return HUF_ERROR_SUCCESS;           /* a helper of a timed-out call may still run on this object: it is left alone, not freed */
    if (sh->stream && !sh->broken) { (void)hipStreamSynchronize(sh->stream); }
    if (sh->owns_comm && sh->comm && rccl()) (void)rccl()->CommDestroy(sh->comm);
    if (sh->d_raw) hufgpu_free(sh->ctx, sh->d_raw);
    if (sh->d_comp) hufgpu_free(sh->ctx, sh->d_comp);
    if (sh->d_offs) hufgpu_free(sh->ctx, sh->d_offs);
    if (sh->d_stage) hufgpu_free(sh->ctx, sh->d_stage);
    if (sh->d_sub) hufgpu_free(sh->ctx, sh->d_sub);
    if (sh->d_words) hufgpu_free(sh->ctx, sh->d_words);
    if (sh->h_words) (void)hipHostFree(sh->h_words);
    if (sh->ev) (void)hipEventDestroy(sh->ev);
    if (sh->stream2) (void)hipStreamDestroy(sh->stream2);
    if (sh->stream) (void)hipStreamDestroy(sh->stream);
    pthread_mutex_destroy(&sh->mu);
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
    pthread_mutex_init(&sh->mu, NULL);
    sh->ctx = ctx;
    sh->nranks = nranks;
    sh->rank = rank;
    sh->device = hufgpu_ctx_device(ctx);
    sh->enc_lens = (uint64_t *)calloc((size_t)nranks, sizeof(uint64_t));
    sh->timeout_ms = 120000u;
    if (const char *t = getenv("HUF_GPU_SHARD_TIMEOUT_MS")) {
        char *end = NULL;
        const unsigned long v = strtoul(t, &end, 10);
        if (end != t && v <= 0xfffffffful) sh->timeout_ms = (uint32_t)v;
    }
    int rc = HUF_ERROR_FATAL;
    do {
        if (!sh->enc_lens) { rc = HUF_ERROR_MEMORY_ALLOCATION; break; }
        if (hipSetDevice(sh->device) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&sh->stream, hipStreamNonBlocking) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&sh->stream2, hipStreamNonBlocking) != hipSuccess) break;
        if (hipEventCreateWithFlags(&sh->ev, hipEventDisableTiming) != hipSuccess) break;
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
    if (root < 0 || root >= sh->nranks) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "root %d of %d ranks", root, sh->nranks);
    if (sh->broken) return fail(sh, HUF_ERROR_FATAL, "the group is broken (an earlier call timed out or its communicator failed)");
    SH_HIP(sh, hipSetDevice(sh->device));
    Call *c = new Call();
    c->sh = sh; c->body = encode_body; c->root = root; c->d_in = d_in; c->n_total = n_total; c->blocksize = blocksize;
    c->flags = flags; c->d_out = d_stream; c->cap = stream_cap; c->d_index = d_block_offsets; c->want_legs = legs_ms != NULL;
    if (!run_call(c)) return HUF_ERROR_FATAL;                  /* (sh->err says it; the call is the helper's to free) */
    const int rc = c->rc;
    if (rc == HUF_ERROR_SUCCESS) {
        if (stream_len) *stream_len = c->out_len;
        if (shard_lens) memcpy(shard_lens, c->lens.data(), (size_t)sh->nranks * 8);
        if (legs_ms) memcpy(legs_ms, c->legs, sizeof c->legs);
    }
    call_release(c);
    return rc;
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
    if (raw_len) *raw_len = 0;
    if (root < 0 || root >= sh->nranks) return fail(sh, HUF_ERROR_INVALID_ARGUMENT, "root %d of %d ranks", root, sh->nranks);
    if (sh->broken) return fail(sh, HUF_ERROR_FATAL, "the group is broken (an earlier call timed out or its communicator failed)");
    SH_HIP(sh, hipSetDevice(sh->device));
    Call *c = new Call();
    c->sh = sh; c->body = decode_body; c->root = root; c->d_in = d_stream; c->stream_len = stream_len; c->d_index_in = d_block_offsets;
    c->n_total = n_total; c->blocksize = blocksize; c->flags = flags; c->d_out = d_out; c->cap = out_cap; c->want_legs = legs_ms != NULL;
    if (!run_call(c)) return HUF_ERROR_FATAL;
    const int rc = c->rc;
    if (raw_len) *raw_len = c->out_len;                         /* (after an error: what a single decoder would have delivered) */
    if (rc == HUF_ERROR_SUCCESS && legs_ms) memcpy(legs_ms, c->legs, sizeof c->legs);
    call_release(c);
    return rc;
}

namespace {

int encode_body(Call *c)
{
    hufgpu_shard_t *sh = c->sh;
    const int G = sh->nranks, me = sh->rank, root = c->root;
    const uint64_t n_total = c->n_total, blocksize = c->blocksize;
    const uint32_t flags = c->flags;
    const void *d_in = c->d_in;
    void *d_stream = c->d_out;
    uint64_t *d_block_offsets = c->d_index;
    sh->have_layout = 0;
    std::vector<uint64_t> lo(G), hi(G), len(G), start(G + 1);
    for (int r = 0; r < G; r++) { range_of(n_total, blocksize, r, G, &lo[r], &hi[r]); len[r] = hi[r] - lo[r]; }
    const uint64_t n = hi[me] - lo[me], nb = hufgpu_block_count(n, blocksize);
    const uint64_t bs_total = hufgpu_block_count(n_total, blocksize);

    /* 1. buffers; then everybody says whether it is ready (a rank that fails alone must not leave the others in a collective) */
    int ready = 1, why = HUF_ERROR_SUCCESS;
    if (me == root && n_total && (!d_in || !d_stream || c->cap < hufgpu_encode_bound(n_total, blocksize) ||
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

    Legs legs(sh, c->want_legs ? c->legs : NULL);
    /* 2. the input shards leave the root - and the root's own shard, which does not move, is encoded on a stream of its
     *    own beside the sends (timed legs: one after the other).  From here to the size all-gather a rank that fails
     *    keeps its error to itself and goes on: the others are on their way into that collective (round 5 returned at
     *    once and left them there). */
    int err = HUF_ERROR_SUCCESS;
    const uint8_t *src = me == root ? (const uint8_t *)d_in + lo[me] : sh->d_raw;
    const bool beside = me == root && !c->want_legs && n != 0;
    { const int rc = scatter(sh, root, (const uint8_t *)d_in, lo.data(), len.data(), sh->d_raw); if (rc) return rc; }   /* (a failed group: the communicator is gone, nobody waits) */
    legs.next();
    /* 3. every rank encodes its blocks (stream and side tables stay here: HUFGPU_SHARD_OWN_LAYOUT decodes with them) */
    if (n) {
        err = hufgpu_encode_sub(sh->ctx, src, n, blocksize, sh->d_comp, sh->comp_cap, sh->d_offs, sh->d_sub, NULL, beside ? sh->stream2 : sh->stream);
        if (err) fail(sh, err, "hufgpu_encode_sub failed: %s", hufgpu_last_error(sh->ctx));
        else if (beside && (hipEventRecord(sh->ev, sh->stream2) != hipSuccess || hipStreamWaitEvent(sh->stream, sh->ev, 0) != hipSuccess)) {
            (void)hipGetLastError();
            err = fail(sh, HUF_ERROR_FATAL, "ordering the root's encode behind its sends failed");
        }
    }
    legs.next();
    /* 4. how every rank fared and how long its shard is: two words a rank (the length stands at the end of the shard's block index) */
    {
        hipError_t e = hipSuccess;
        sh->h_words[0] = (uint64_t)err;
        e = hipMemcpyAsync(sh->d_words, sh->h_words, 8, hipMemcpyHostToDevice, sh->stream);
        if (e == hipSuccess) e = (n && !err) ? hipMemcpyAsync(sh->d_words + 1, sh->d_offs + nb, 8, hipMemcpyDeviceToDevice, sh->stream)
                                             : hipMemsetAsync(sh->d_words + 1, 0, 8, sh->stream);
        if (e != hipSuccess) {                                     /* (nothing to put into the collective: the group cannot go on) */
            (void)hipGetLastError();
            break_group(sh);
            return fail(sh, HUF_ERROR_FATAL, "staging the shard's length failed: %s", hipGetErrorString(e));
        }
    }
    SH_RCCL(sh, rccl()->AllGather(sh->d_words, sh->d_words + 2, 2, RCCL_UINT64, sh->comm, sh->stream));
    SH_HIP(sh, hipMemcpyAsync(sh->h_words + 2, sh->d_words + 2, (size_t)G * 16, hipMemcpyDeviceToHost, sh->stream));
    SH_WAIT(sh);
    for (int r = 0; r < G; r++) {                                  /* the first error in rank order, on every rank */
        const int e = (int)sh->h_words[2 + 2 * r];
        if (e) return (e == err && r == me) ? err : fail(sh, e, "rank %d failed to encode its blocks", r);
    }
    start[0] = 0;
    for (int r = 0; r < G; r++) { sh->enc_lens[r] = sh->h_words[2 + 2 * r + 1]; start[r + 1] = start[r] + sh->enc_lens[r]; }
    if (start[G] > hufgpu_encode_bound(n_total, blocksize)) return fail(sh, HUF_ERROR_FATAL, "the shards are longer than the bound of the whole");
    legs.next();
    /* 5. the compressed shards to their places in the root's stream: rank order = stream order */
    { const int rc = gather(sh, root, (uint8_t *)d_stream, start.data(), sh->enc_lens, sh->d_comp); if (rc) return rc; }
    if (me == root && sh->enc_lens[me])
        SH_HIP(sh, hipMemcpyAsync((uint8_t *)d_stream + start[me], sh->d_comp, sh->enc_lens[me], hipMemcpyDeviceToDevice, sh->stream));
    if (flags & HUFGPU_SHARD_INDEX) {
        /* the block index of the whole: every shard's entries counted from the stream's first byte, the last entry = the length */
        std::vector<uint64_t> boff(G), blen(G);
        for (int r = 0; r < G; r++) {
            boff[r] = 8 * hufgpu_block_count(lo[r], blocksize ? blocksize : n_total);   /* (whole blocks in front of the shard) */
            blen[r] = 8 * hufgpu_block_count(hi[r] - lo[r], blocksize);
        }
        { const int rc = rebase(sh, sh->d_stage, sh->d_offs, nb, 0, start[me]); if (rc) return rc; }
        { const int rc = gather(sh, root, (uint8_t *)d_block_offsets, boff.data(), blen.data(), (const uint8_t *)sh->d_stage); if (rc) return rc; }
        if (me == root) {
            if (nb) SH_HIP(sh, hipMemcpyAsync((uint8_t *)d_block_offsets + boff[me], sh->d_stage, nb * 8, hipMemcpyDeviceToDevice, sh->stream));
            sh->h_words[0] = start[G];
            SH_HIP(sh, hipMemcpyAsync(d_block_offsets + bs_total, sh->h_words, 8, hipMemcpyHostToDevice, sh->stream));
        }
    }
    SH_WAIT(sh);
    legs.next();
    sh->have_layout = 1;
    sh->enc_root = root;
    sh->enc_total = n_total;
    sh->enc_bs = blocksize;
    sh->enc_len = start[G];
    c->out_len = start[G];
    c->lens.assign(sh->enc_lens, sh->enc_lens + G);
    return HUF_ERROR_SUCCESS;
}

int decode_body(Call *c)
{
    hufgpu_shard_t *sh = c->sh;
    const int G = sh->nranks, me = sh->rank, root = c->root;
    const uint64_t n_total = c->n_total, blocksize = c->blocksize, stream_len = c->stream_len, out_cap = c->cap;
    const uint32_t flags = c->flags;
    const void *d_stream = c->d_in;
    const uint64_t *d_block_offsets = c->d_index_in;
    void *d_out = c->d_out;
    const bool own = (flags & HUFGPU_SHARD_OWN_LAYOUT) != 0;
    const uint32_t dflags = flags & (HUFGPU_RELAXED_TREE);
    const uint64_t bs = blocksize ? blocksize : n_total;
    const uint64_t nblocks = hufgpu_block_count(n_total, blocksize);
    c->out_len = 0;

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
    /* a foreign stream's index goes into the buffers the last encode's layout lives in: that layout is gone */
    if (!own) sh->have_layout = 0;
    Legs legs(sh, c->want_legs ? c->legs : NULL);
    /* 1. which blocks a rank decodes: b[r] .. b[r + 1], and where their bytes lie in the stream */
    std::vector<uint64_t> b(G + 1), cstart(G + 1);
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
                    /* the cuts of a damaged index must not send the root beyond its stream: they rise and end inside it */
                    for (int r = 0; r < G && ready; r++)
                        if (plan[G + 1 + r] > plan[G + 2 + r] || plan[G + 2 + r] > stream_len) {
                            ready = 0; why = HUF_ERROR_INVALID_ARGUMENT;
                            fail(sh, why, "the block index does not rise (block %llu): the stream cannot be cut along it", (unsigned long long)plan[r + 1]);
                        }
                }
                free(h_offs);
            }
        }
        plan[2 * G + 2] = (uint64_t)ready;
        if (me == root) SH_HIP(sh, hipMemcpyAsync(sh->d_words, plan, (size_t)W * 8, hipMemcpyHostToDevice, sh->stream));
        SH_RCCL(sh, rccl()->Broadcast(sh->d_words, sh->d_words, (size_t)W, RCCL_UINT64, root, sh->comm, sh->stream));
        SH_HIP(sh, hipMemcpyAsync(plan, sh->d_words, (size_t)W * 8, hipMemcpyDeviceToHost, sh->stream));
        SH_WAIT(sh);
        if (!plan[2 * G + 2]) return me == root ? why : fail(sh, HUF_ERROR_FATAL, "the root is not ready");
        for (int r = 0; r <= G; r++) { b[r] = plan[r]; cstart[r] = plan[G + 1 + r]; }
    }
    const uint64_t nb = b[me + 1] - b[me];
    std::vector<uint64_t> rlo(G), rlen(G), clen(G);
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
    { const int rc = scatter(sh, root, (const uint8_t *)d_stream, cstart.data(), clen.data(), sh->d_comp); if (rc) return rc; }
    const uint64_t *my_offs = sh->d_offs;
    if (!own) {
        std::vector<uint64_t> ioff(G), ilen(G);
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
        { const int rc = scatter(sh, root, (const uint8_t *)sh->d_stage, ioff.data(), ilen.data(), (uint8_t *)sh->d_offs); if (rc) return rc; }
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
            c->out_len = total;                                /* (the blocks in front of the failing shard and what it delivered) */
            return e == err && r == me ? err : fail(sh, e, "rank %d failed to decode its blocks", r);
        }
    }
    /* 4. the output shards to the root */
    { const int rc = gather(sh, root, (uint8_t *)d_out, rlo.data(), rlen.data(), sh->d_raw); if (rc) return rc; }
    SH_WAIT(sh);
    legs.next();
    c->out_len = total;
    return HUF_ERROR_SUCCESS;
}

}  // namespace
