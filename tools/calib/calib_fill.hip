// Store-bandwidth ceilings on gfx950 (MI355X) for the shape of the one-symbol decode (a fill of
// 64 KiB per workgroup next to an 8 KiB read): which part of the gap to a plain fill is the grid
// shape, the store policy, the LDS footprint, the read or the dependent loads in front of it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__device__ __forceinline__ void st16(uint4 *p, uint4 v)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p)); else *p = v;
}

// MODE bit 0: non-temporal stores, bit 1: LDS footprint of the decode kernel, bit 2: 16-byte read per
// thread before the fill (checked after it), bit 3: the read's address depends on a loaded offset
template <int THREADS, int PER_WG, int MODE>
__global__ __launch_bounds__(THREADS) void fill_kernel(uint8_t *out, const uint8_t *src, const uint64_t *offs, uint32_t *flag)
{
    __shared__ uint32_t lds[(MODE & 2) ? 10240 : 1];
    const int tid = threadIdx.x;
    const uint64_t blk = blockIdx.x;
    if (MODE & 2) lds[tid] = tid;
    uint4 r = make_uint4(0, 0, 0, 0);
    if (MODE & 4) {
        const uint64_t o = (MODE & 8) ? offs[blk] : blk * (PER_WG / 8);
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + o) + tid);
        r = make_uint4(t.x, t.y, t.z, t.w);
    }
    uint4 *q = reinterpret_cast<uint4 *>(out + blk * PER_WG);
    const uint4 v = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    if (MODE & 16) {         // groups of ITERS workgroups: in step i all of them write to block G+i (piece j each)
        constexpr int ITERS = PER_WG / 16 / THREADS;
        const uint64_t g0 = blk - blk % ITERS, j = blk % ITERS;
#pragma unroll
        for (int i = 0; i < ITERS; i++)
            st16<(MODE & 1) != 0>(reinterpret_cast<uint4 *>(out + (g0 + i) * PER_WG) + j * THREADS + tid, v);
    } else {
#pragma unroll
    for (int i = 0; i < PER_WG / 16 / THREADS; i++) st16<(MODE & 1) != 0>(q + i * THREADS + tid, v);
    }
    if (MODE & 4) {
        if (__syncthreads_or((r.x | r.y | r.z | r.w) != 0u) && tid == 0) flag[0] = 1;
    }
    if ((MODE & 2) && lds[(tid * 7) & 511] == 0xffffffffu) flag[1] = 1;
}

// read PER_WG bytes per workgroup (nt loads), TRANSPOSED: step i of all workgroups of a group reads block G+i
template <int THREADS, int PER_WG, bool TRANSPOSED>
__global__ __launch_bounds__(THREADS) void read_kernel(const uint8_t *in, uint32_t *flag)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int ITERS = PER_WG / 16 / THREADS;
    const int tid = threadIdx.x;
    const uint64_t blk = blockIdx.x;
    const uint64_t g0 = blk - blk % ITERS, j = blk % ITERS;
    v4u acc = {0, 0, 0, 0};
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
        const v4u *p = TRANSPOSED ? reinterpret_cast<const v4u *>(in + (g0 + i) * PER_WG) + j * THREADS + tid
                                  : reinterpret_cast<const v4u *>(in + blk * PER_WG) + i * THREADS + tid;
        acc |= __builtin_nontemporal_load(p);
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) flag[2] = 1;
}

template <typename F>
static float time_ms(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++) launch();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 20; i++) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 20;
}

int main()
{
    const uint64_t N = 1ull << 30;
    uint8_t *out, *src; uint64_t *offs; uint32_t *flag;
    CHECK(hipMalloc(&out, N)); CHECK(hipMalloc(&src, N / 8 + 65536)); CHECK(hipMalloc(&offs, 16384 * 8)); CHECK(hipMalloc(&flag, 64));
    CHECK(hipMemset(src, 0, N / 8 + 65536)); CHECK(hipMemset(flag, 0, 64));
    uint64_t h[16384];
    for (int i = 0; i < 16384; i++) h[i] = (uint64_t)i * 8192;
    CHECK(hipMemcpy(offs, h, sizeof(h), hipMemcpyHostToDevice));
#define RUN(T, P, M, label) printf("%-58s %.4f ms\n", label, time_ms([&] { fill_kernel<T, P, M><<<dim3((unsigned)(N / P)), dim3(T), 0, 0>>>(out, src, offs, flag); }))
    RUN(256, 4096, 0, "262144 x 256 thr, 4 KiB, plain stores");
    RUN(256, 4096, 1, "262144 x 256 thr, 4 KiB, nt stores");
    RUN(512, 8192, 0, "131072 x 512 thr, 8 KiB, plain stores");
    RUN(256, 16384, 0, "65536 x 256 thr, 16 KiB, plain stores");
    RUN(256, 16384, 1, "65536 x 256 thr, 16 KiB, nt stores");
    RUN(512, 65536, 0, "16384 x 512 thr, 64 KiB, plain stores");
    RUN(512, 65536, 1, "16384 x 512 thr, 64 KiB, nt stores");
    RUN(512, 65536, 3, "  + 40 KiB LDS per workgroup");
    RUN(512, 65536, 5, "  + 8 KiB read (no LDS)");
    RUN(512, 65536, 7, "  + 8 KiB read + LDS");
    RUN(512, 65536, 15, "  + 8 KiB read at a loaded offset + LDS");
    RUN(512, 65536, 6, "  plain stores + 8 KiB read + LDS");
    RUN(512, 65536, 17, "16384 x 512 thr, 64 KiB transposed over 8 workgroups, nt");
    RUN(512, 65536, 16, "16384 x 512 thr, 64 KiB transposed over 8 workgroups, plain");
    RUN(512, 65536, 23, "  transposed nt + 8 KiB read + LDS");
    RUN(512, 65536, 22, "  transposed plain + 8 KiB read + LDS");
    RUN(256, 65536, 17, "16384 x 256 thr, 64 KiB transposed over 16 workgroups, nt");
    RUN(256, 65536, 16, "16384 x 256 thr, 64 KiB transposed over 16 workgroups, plain");
    RUN(256, 65536, 1, "16384 x 256 thr, 64 KiB, nt stores");
    RUN(1024, 65536, 1, "16384 x 1024 thr, 64 KiB, nt stores");
#define RUNR(T, P, TR, label) printf("%-58s %.4f ms\n", label, time_ms([&] { read_kernel<T, P, TR><<<dim3((unsigned)(N / P)), dim3(T), 0, 0>>>(out, flag); }))
    RUNR(256, 4096, false, "read 262144 x 256 thr, 4 KiB");
    RUNR(256, 16384, false, "read 65536 x 256 thr, 16 KiB");
    RUNR(256, 65536, false, "read 16384 x 256 thr, 64 KiB");
    RUNR(256, 65536, true, "read 16384 x 256 thr, 64 KiB transposed over 16");
    RUNR(512, 65536, false, "read 16384 x 512 thr, 64 KiB");
    RUNR(1024, 65536, false, "read 16384 x 1024 thr, 64 KiB");
    RUNR(1024, 65536, true, "read 16384 x 1024 thr, 64 KiB transposed over 4");
    CHECK(hipDeviceSynchronize());
    return 0;
}
