/* What the host link gives on this box, for the design of huf_host.cpp's transfers:
 *   1. pinned buffers: host -> device alone, device -> host alone, both at once (two streams)
 *   2. hipHostRegister / hipHostUnregister of pageable memory by piece size, pages touched and untouched,
 *      and the copy rate from / into registered memory
 *   3. the same with several threads, each registering and copying its own pieces
 * build: hipcc -O2 --offload-arch=gfx950 tools/calib/host_link_probe.hip -o /tmp/host_link_probe -lpthread */
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef struct { char *host; char *dev; size_t n, piece; int to_device, nthreads, idx; double secs; } job_t;
static void *reg_worker(void *arg)
{
    job_t *j = (job_t *)arg;
    hipStream_t s;
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t pieces = j->n / j->piece;
    const double t0 = now();
    for (size_t p = (size_t)j->idx; p < pieces; p += (size_t)j->nthreads) {
        char *h = j->host + p * j->piece;
        CK(hipHostRegister(h, j->piece, hipHostRegisterDefault));
        if (j->to_device) CK(hipMemcpyAsync(j->dev + p * j->piece, h, j->piece, hipMemcpyHostToDevice, s));
        else CK(hipMemcpyAsync(h, j->dev + p * j->piece, j->piece, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        CK(hipHostUnregister(h));
    }
    j->secs = now() - t0;
    CK(hipStreamDestroy(s));
    return NULL;
}

int main(void)
{
    const size_t N = (size_t)1 << 30;
    char *d_a, *d_b, *pin_a, *pin_b;
    CK(hipSetDevice(0));
    CK(hipMalloc((void **)&d_a, N)); CK(hipMalloc((void **)&d_b, N));
    CK(hipHostMalloc((void **)&pin_a, N, hipHostMallocPortable)); CK(hipHostMalloc((void **)&pin_b, N, hipHostMallocPortable));
    memset(pin_a, 1, N); memset(pin_b, 2, N);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d_a, pin_a, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        double t1 = now();
        CK(hipMemcpyAsync(pin_b, d_b, N, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        double t2 = now();
        CK(hipMemcpyAsync(d_a, pin_a, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin_b, d_b, N, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        double t3 = now();
        /* in pieces of 4 MiB on each stream */
        for (size_t o = 0; o < N; o += (size_t)4 << 20) {
            CK(hipMemcpyAsync(d_a + o, pin_a + o, (size_t)4 << 20, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(pin_b + o, d_b + o, (size_t)4 << 20, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        double t4 = now();
        if (rep) printf("pinned 1 GiB: h2d %.1f GiB/s, d2h %.1f GiB/s, both at once %.1f GiB/s each way (%.2f ms), in 4 MiB pieces %.1f (%.2f ms)\n",
                        1 / (t1 - t0), 1 / (t2 - t1), 1 / (t3 - t2), (t3 - t2) * 1e3, 1 / (t4 - t3), (t4 - t3) * 1e3);
    }
    /* register / unregister by piece size */
    for (int touched = 0; touched < 2; touched++)
        for (size_t piece = (size_t)4 << 20; piece <= ((size_t)64 << 20); piece *= 4) {
            char *m = (char *)mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            madvise(m, N, MADV_HUGEPAGE);
            if (touched) memset(m, 3, N);
            double treg = 0, tcpy = 0, tun = 0;
            const size_t pieces = ((size_t)256 << 20) / piece;
            for (size_t p = 0; p < pieces; p++) {
                double a = now();
                CK(hipHostRegister(m + p * piece, piece, hipHostRegisterDefault));
                double b = now();
                CK(hipMemcpyAsync(d_a + p * piece, m + p * piece, piece, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
                double c = now();
                CK(hipHostUnregister(m + p * piece));
                double d = now();
                treg += b - a; tcpy += c - b; tun += d - c;
            }
            printf("register %s pieces of %2zu MiB (256 MiB): register %.3f ms, copy %.3f ms (%.1f GiB/s), unregister %.3f ms per piece -> %.1f GiB/s one thread\n",
                   touched ? "touched  " : "untouched", piece >> 20, treg / pieces * 1e3, tcpy / pieces * 1e3, piece / 1073741824.0 / (tcpy / pieces),
                   tun / pieces * 1e3, 0.25 / (treg + tcpy + tun));
            munmap(m, N);
        }
    /* several threads, each its own pieces: host -> device from touched memory, device -> host into untouched memory, and both at once */
    for (int nt = 2; nt <= 8; nt *= 2)
        for (size_t piece = (size_t)4 << 20; piece <= ((size_t)16 << 20); piece *= 4) {
            char *src = (char *)mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            char *dst = (char *)mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            madvise(src, N, MADV_HUGEPAGE); madvise(dst, N, MADV_HUGEPAGE);
            memset(src, 5, N);
            job_t jobs[16]; pthread_t th[16];
            double t0 = now();
            for (int i = 0; i < nt; i++) { jobs[i] = (job_t){src, d_a, N, piece, 1, nt, i, 0}; pthread_create(&th[i], NULL, reg_worker, &jobs[i]); }
            for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
            double t1 = now();
            for (int i = 0; i < nt; i++) { jobs[i] = (job_t){dst, d_b, N, piece, 0, nt, i, 0}; pthread_create(&th[i], NULL, reg_worker, &jobs[i]); }
            for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
            double t2 = now();
            munmap(dst, N);
            dst = (char *)mmap(NULL, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            madvise(dst, N, MADV_HUGEPAGE);
            for (int i = 0; i < nt; i++) { jobs[i] = (job_t){src, d_a, N, piece, 1, nt, i, 0}; pthread_create(&th[i], NULL, reg_worker, &jobs[i]); }
            for (int i = 0; i < nt; i++) { jobs[8 + i] = (job_t){dst, d_b, N, piece, 0, nt, i, 0}; pthread_create(&th[8 + i], NULL, reg_worker, &jobs[8 + i]); }
            for (int i = 0; i < nt; i++) { pthread_join(th[i], NULL); pthread_join(th[8 + i], NULL); }
            double t3 = now();
            printf("%d threads a direction, pieces of %2zu MiB, 1 GiB: h2d %.1f GiB/s, d2h into fresh pages %.1f GiB/s, both at once %.1f GiB/s each way (%.1f ms)\n",
                   nt, piece >> 20, 1 / (t1 - t0), 1 / (t2 - t1), 1 / (t3 - t2), (t3 - t2) * 1e3);
            munmap(src, N); munmap(dst, N);
        }
    return 0;
}
