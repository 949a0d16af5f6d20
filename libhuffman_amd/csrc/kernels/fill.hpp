/* fill.hpp - synthetic inputs of SURVEY 8d generated in HBM.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "util.hpp"

namespace hufgpu {

/* ======================================================================================
 * Synthetic inputs of SURVEY §8d (libhuffman_amd/datagen.py is the numpy twin).
 * ==================================================================================== */
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_kernel(uint8_t *__restrict__ out, uint64_t n, int kind, uint64_t seed, uint64_t first,
                            const uint64_t *__restrict__ zipf_cum)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = first + i;                 /* global byte index */
        uint8_t v;
        if (kind == 0) v = 0x41;
        else if (kind == 1) v = (uint8_t)(splitmix64_at(seed, (g >> 3) + 1) >> (8 * (g & 7)));
        else if (kind == 2) v = (uint8_t)(splitmix64_at(seed, g + 1) % 255ull);
        else {
            const uint64_t u = splitmix64_at(seed, g + 1) % zipf_cum[254];
            int lo = 0, hi = 255;                     /* number of cum[r] <= u */
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (zipf_cum[mid] <= u) lo = mid + 1; else hi = mid;
            }
            v = (uint8_t)lo;
        }
        out[i] = v;
    }
}

/* ======================================================================================
 * Bandwidth calibration (bench.py's ceilings; tools/calib): what a kernel that does nothing but move bytes reaches
 * on this part - 16 bytes per lane and access, every workgroup a contiguous piece, grid = bytes / PER_WG.
 * KIND 0: copy (read a, write b), 1: read a only (the OR of everything decides one store nobody takes), 2: fill b.
 * NTL / NTS: non-temporal loads / stores.
 * ==================================================================================== */
template <int THREADS, int PER_WG, int KIND, bool NTL, bool NTS>
__global__ __launch_bounds__(THREADS) void calib_bw_kernel(const uint8_t *__restrict__ a, uint8_t *__restrict__ b, uint32_t *__restrict__ flag)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int ITERS = PER_WG / 16 / THREADS;
    const v4u *src = reinterpret_cast<const v4u *>(a + (uint64_t)blockIdx.x * PER_WG) + threadIdx.x;
    v4u *dst = reinterpret_cast<v4u *>(b + (uint64_t)blockIdx.x * PER_WG) + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
    v4u v[ITERS > 8 ? 8 : ITERS];
#pragma unroll 1
    for (int i0 = 0; i0 < ITERS; i0 += 8) {
#pragma unroll
        for (int i = 0; i < 8 && i0 + i < ITERS; i++) {
            if (KIND == 2) v[i] = v4u{0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
            else v[i] = NTL ? __builtin_nontemporal_load(src + (i0 + i) * THREADS) : src[(i0 + i) * THREADS];
        }
#pragma unroll
        for (int i = 0; i < 8 && i0 + i < ITERS; i++) {
            if (KIND == 1) acc |= v[i];
            else if (NTS) __builtin_nontemporal_store(v[i], dst + (i0 + i) * THREADS);
            else dst[(i0 + i) * THREADS] = v[i];
        }
    }
    if (KIND == 1 && (acc.x | acc.y | acc.z | acc.w) == 0x12345678u) flag[0] = 1;
}

}  // namespace hufgpu
