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

}  // namespace hufgpu
