/*
 * mock_rccl.cpp - TEST ONLY.  The fourteen RCCL entry points libhuffman_amd/csrc/hufgpu_sharded.hip looks up, over named
 * pipes between processes that SHARE ONE GPU, so that the multi-rank control flow of hufgpu_encode_sharded /
 * hufgpu_decode_sharded (who sends what to whom, in which group, at which offset) runs on a one-GPU box - RCCL itself
 * refuses two ranks on one device.  HUF_GPU_RCCL_LIB points the library at this file's .so; MOCK_RCCL_DIR names the
 * directory the pipes live in.  Semantics kept: operations inside ncclGroupStart/End are deferred to the group's end and
 * run concurrently there (a thread a peer and direction); an operation waits for the work enqueued on its stream before
 * it touches the buffer, and the caller's thread returns when the data has arrived.  Not kept: asynchrony (every
 * operation is complete when the call returns - a peer that never arrives holds the CALL, where the real RCCL would hold
 * the stream), performance, error reporting beyond "it failed".  ncclCommAbort makes every blocked operation of the
 * communicator return a failure within a few milliseconds (the pipes are opened and read without blocking, in a poll loop).
 * build: hipcc -O1 -fPIC -shared tests/mock_rccl/mock_rccl.cpp -o <somewhere>/libmock_rccl.so -lpthread
 */
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {
struct Comm { int nranks, rank, device; char tag[33]; int fd_send[64], fd_recv[64]; volatile int aborted; };   /* pipes stay open for the communicator's life: data written is not lost between two groups */
struct Op { int send; void *dev; size_t bytes; int peer; Comm *c; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
const size_t type_size[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};

void pipe_path(const Comm *c, int from, int to, char *out, size_t n)
{
    const char *dir = getenv("MOCK_RCCL_DIR");
    snprintf(out, n, "%s/%s_%d_%d", dir ? dir : "/tmp", c->tag, from, to);
}

/* (the descriptors are non-blocking: a peer that is not there yet - a read end without a writer reads 0 bytes, a full pipe
 *  refuses - is waited for in steps of a few milliseconds, so that an abort of the communicator is seen) */
bool io_all(const Comm *c, int fd, char *p, size_t n, bool wr)
{
    while (n) {
        if (c->aborted) return false;
        const ssize_t k = wr ? write(fd, p, n) : read(fd, p, n);
        if (k > 0) { p += k; n -= (size_t)k; continue; }
        if (k < 0 && errno == EINTR) continue;
        if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK) return false;
        struct pollfd pf = {fd, (short)(wr ? POLLOUT : POLLIN), 0};
        (void)poll(&pf, 1, 5);
        if (k == 0 && !wr) usleep(2000);                /* (no writer yet: poll returns at once with POLLHUP) */
    }
    return true;
}

/* all operations of one peer and direction, in the order they were issued */
struct Lane { std::vector<Op> ops; bool ok; pthread_t th; };
void *lane_main(void *arg)
{
    Lane *l = (Lane *)arg;
    l->ok = false;
    const Op &first = l->ops[0];
    if (hipSetDevice(first.c->device) != hipSuccess) return NULL;
    char path[512];
    pipe_path(first.c, first.send ? first.c->rank : first.peer, first.send ? first.peer : first.c->rank, path, sizeof path);
    int *slot = first.send ? &first.c->fd_send[first.peer] : &first.c->fd_recv[first.peer];
    if (*slot < 0) {
        if (mkfifo(path, 0600) != 0 && errno != EEXIST) return NULL;
        /* (a write end opens only when the read end is there: ENXIO until then) */
        while (!first.c->aborted && (*slot = open(path, (first.send ? O_WRONLY : O_RDONLY) | O_NONBLOCK)) < 0 && errno == ENXIO) usleep(2000);
    }
    const int fd = *slot;
    if (fd < 0) return NULL;
    bool ok = true;
    for (const Op &op : l->ops) {
        char *host = (char *)malloc(op.bytes ? op.bytes : 1);
        if (!host) { ok = false; break; }
        if (op.send) {
            ok = hipMemcpy(host, op.dev, op.bytes, hipMemcpyDeviceToHost) == hipSuccess && io_all(op.c, fd, host, op.bytes, true);
        } else {
            ok = io_all(op.c, fd, host, op.bytes, false) && hipMemcpy(op.dev, host, op.bytes, hipMemcpyHostToDevice) == hipSuccess;
        }
        free(host);
        if (!ok) break;
    }
    l->ok = ok;
    return NULL;
}

int run(std::vector<Op> &ops)
{
    if (ops.empty()) return 0;
    for (const Op &op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;      /* what was enqueued before the operation is done */
    std::vector<Lane *> lanes;
    for (const Op &op : ops) {
        Lane *found = NULL;
        for (Lane *l : lanes)
            if (l->ops[0].send == op.send && l->ops[0].peer == op.peer && l->ops[0].c == op.c) found = l;
        if (!found) { found = new Lane(); lanes.push_back(found); }
        found->ops.push_back(op);
    }
    for (Lane *l : lanes) pthread_create(&l->th, NULL, lane_main, l);
    int rc = 0;
    for (Lane *l : lanes) { pthread_join(l->th, NULL); if (!l->ok) rc = 1; delete l; }
    ops.clear();
    return rc;
}

int issue(const Op &op)
{
    g_ops.push_back(op);
    return g_depth ? 0 : run(g_ops);
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    FILE *f = fopen("/dev/urandom", "rb");
    unsigned char rnd[16] = {0};
    if (f) { if (fread(rnd, 1, 16, f) != 16) rnd[0] = (unsigned char)getpid(); fclose(f); }
    for (int i = 0; i < 16; i++) snprintf(id->internal + 2 * i, 3, "%02x", rnd[i]);
    return 0;
}
int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    Comm *c = (Comm *)calloc(1, sizeof *c);
    c->nranks = nranks; c->rank = rank;
    memcpy(c->tag, id.internal, 32);
    if (nranks > 64) { free(c); return 1; }
    for (int i = 0; i < 64; i++) c->fd_send[i] = c->fd_recv[i] = -1;
    if (hipGetDevice(&c->device) != hipSuccess) { free(c); return 1; }
    *comm = c;
    return 0;
}
int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    for (int i = 0; i < 64; i++) {
        if (c->fd_send[i] >= 0) close(c->fd_send[i]);
        if (c->fd_recv[i] >= 0) close(c->fd_recv[i]);
    }
    free(c);
    return 0;
}
/* every blocked operation of the communicator returns a failure; the object itself is left to the process's end (its
 * operations may still be on their way out) */
int ncclCommAbort(void *comm) { ((Comm *)comm)->aborted = 1; return 0; }
int ncclCommGetAsyncError(void *comm, int *err) { *err = ((Comm *)comm)->aborted ? 1 : 0; return 0; }
int ncclCommCount(void *comm, int *n) { *n = ((Comm *)comm)->nranks; return 0; }
int ncclCommUserRank(void *comm, int *r) { *r = ((Comm *)comm)->rank; return 0; }
const char *ncclGetErrorString(int r) { return r ? "mock RCCL: the operation failed" : "no error"; }
int ncclGroupStart(void) { g_depth++; return 0; }
int ncclGroupEnd(void) { return --g_depth ? 0 : run(g_ops); }
int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t s)
{
    return issue(Op{1, (void *)buf, count * type_size[type], peer, (Comm *)comm, s});
}
int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t s)
{
    return issue(Op{0, buf, count * type_size[type], peer, (Comm *)comm, s});
}
int ncclAllGather(const void *send, void *recv, size_t count, int type, void *comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_size[type];
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpy((char *)recv + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    g_depth++;
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank) continue;
        g_ops.push_back(Op{1, (void *)send, bytes, r, c, s});
        g_ops.push_back(Op{0, (char *)recv + (size_t)r * bytes, bytes, r, c, s});
    }
    return --g_depth ? 0 : run(g_ops);
}
int ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, void *comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_size[type];
    g_depth++;
    if (c->rank == root) {
        for (int r = 0; r < c->nranks; r++)
            if (r != root) g_ops.push_back(Op{1, (void *)send, bytes, r, c, s});
    } else {
        g_ops.push_back(Op{0, recv, bytes, root, c, s});
    }
    int rc = --g_depth ? 0 : run(g_ops);
    if (rc == 0 && c->rank == root && send != recv) rc = hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess;
    return rc;
}
}
