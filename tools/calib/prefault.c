/* How fast can fresh anonymous pages be made present?  (The first copy into a fresh buffer is bound by this.) */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
typedef struct { char *p; size_t n; int mode; } job_t;
static void *work(void *a)
{
    job_t *j = (job_t *)a;
    if (j->mode == 0) { if (madvise(j->p, j->n, MADV_POPULATE_WRITE)) perror("madvise"); }
    else for (size_t i = 0; i < j->n; i += 4096) ((volatile char *)j->p)[i] = 0;
    return NULL;
}
static void run(const char *name, size_t n, int threads, int mode, int huge)
{
    char *p = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (huge) madvise(p, n, MADV_HUGEPAGE);
    pthread_t th[64]; job_t jobs[64];
    double t0 = now();
    size_t piece = (n / threads + 4095) & ~(size_t)4095;
    for (int i = 0; i < threads; i++) {
        size_t off = (size_t)i * piece;
        jobs[i].p = p + off; jobs[i].n = off + piece <= n ? piece : n - off; jobs[i].mode = mode;
        pthread_create(&th[i], NULL, work, &jobs[i]);
    }
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    double t1 = now();
    memset(p, 1, n);
    double t2 = now();
    printf("%-44s %2d threads: populate %7.2f ms, memset after %6.2f ms\n", name, threads, t1 - t0, t2 - t1);
    munmap(p, n);
}
int main(void)
{
    size_t n = (size_t)240 << 20;
    for (int t = 1; t <= 16; t *= 2) run("madvise(POPULATE_WRITE)", n, t, 0, 0);
    for (int t = 1; t <= 16; t *= 2) run("touch a byte per page", n, t, 1, 0);
    for (int t = 1; t <= 16; t *= 4) run("MADV_HUGEPAGE + touch", n, t, 1, 1);
    for (int t = 1; t <= 16; t *= 4) run("MADV_HUGEPAGE + madvise(POPULATE_WRITE)", n, t, 0, 1);
    char *p = malloc(n); double t0 = now(); memset(p, 1, n); printf("malloc + memset: %.2f ms\n", now() - t0); free(p);
    return 0;
}
