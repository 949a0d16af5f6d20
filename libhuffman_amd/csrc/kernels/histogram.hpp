/* histogram.hpp - hist256_kernel: per-block byte counts (src/histogram.c:73-103).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "util.hpp"

namespace hufgpu {

/* ======================================================================================
 * hist256 - replaces huf_histogram_populate (src/histogram.c:73-103, iota = 1).
 *
 * One workgroup per block; every wavefront owns a private 256-bin histogram in LDS so that
 * LDS atomics of different waves never collide; the wave copies are summed at the end.
 * Each lane reads 16 contiguous bytes per step (a wave reads 1 KiB, fully coalesced).
 * Runs of one byte value are folded before touching LDS: a 16-byte chunk of one value costs
 * one atomic, and a whole wave-step of one value costs one atomic for the wave - that is the
 * common case on BASELINE config 2 (all 0x41), where per-byte atomics would serialise 64-way.
 * ==================================================================================== */
#ifndef HIST_COPIES
#define HIST_COPIES 4
#endif
#ifndef HIST_THREADS
#define HIST_THREADS 256
#endif
#define HIST_SIDE_STRIDE HIST_THREADS      /* hist_tree's per-lane words: [hot0: one per lane][hot1: one per lane], bank = lane */

/* `one` is what a single occurrence adds: 1, or 1 << 16 when two 16-bit counters share a word */
__device__ __forceinline__ void hist_add_bytes(uint32_t *h, uint32_t w, uint32_t one)
{
    atomicAdd(&h[w & 0xffu], one);
    atomicAdd(&h[(w >> 8) & 0xffu], one);
    atomicAdd(&h[(w >> 16) & 0xffu], one);
    atomicAdd(&h[w >> 24], one);
}

/* rot4 = the lane's bank rotation in every byte (0: none): the count of byte b is kept at word b ^ rot */
__device__ __forceinline__ void hist_add_chunk(uint32_t *h, uint4 v, uint32_t one = 1u, uint32_t rot4 = 0u)
{
    const uint32_t b = v.x & 0xffu;
    const uint32_t rep = b * 0x01010101u;
    const bool uni = (v.x == rep) & (v.y == rep) & (v.z == rep) & (v.w == rep);
    const unsigned long long act = __ballot(1);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    const unsigned long long same = __ballot(uni && b == b0);
    if (same == act) {                       /* the whole wave step holds one byte value */
        if ((unsigned)lane_id() == (unsigned)__builtin_ctzll(act))
            atomicAdd(&h[b0 ^ (rot4 & 0xffu)], one * 16u * (uint32_t)__popcll(act));
        return;
    }
    if (uni) {
        atomicAdd(&h[b ^ (rot4 & 0xffu)], one * 16u);
        return;
    }
    hist_add_bytes(h, v.x ^ rot4, one);
    hist_add_bytes(h, v.y ^ rot4, one);
    hist_add_bytes(h, v.z ^ rot4, one);
    hist_add_bytes(h, v.w ^ rot4, one);
}

/* The same with two byte values (hot0, hot1; 0x100 = none) taken out of the conflicts: their
 * occurrences go to words of the lane's own (side, side + 1 = those words' indices relative to h)
 * instead of the shared bins.  On skewed data the lanes of a wave that meet a frequent byte in one
 * ds_add serialise on its bin - 72 % of the LDS cycles of the Zipf histogram were such conflicts. */
__device__ __forceinline__ int hist_hot_index(uint32_t b, uint32_t hot0, uint32_t hot1, int side)
{
    return (b == hot0) ? side : ((b == hot1) ? side + HIST_SIDE_STRIDE : (int)b);
}

__device__ __forceinline__ void hist_add_bytes_hot(uint32_t *h, uint32_t w, uint32_t one, uint32_t hot0,
                                                   uint32_t hot1, int side)
{
#pragma unroll
    for (int k = 0; k < 4; k++) atomicAdd(&h[hist_hot_index((w >> (8 * k)) & 0xffu, hot0, hot1, side)], one);
}

__device__ __forceinline__ void hist_add_chunk_hot(uint32_t *h, uint4 v, uint32_t one, uint32_t hot0,
                                                   uint32_t hot1, int side)
{
    const uint32_t b = v.x & 0xffu;
    const uint32_t rep = b * 0x01010101u;
    const bool uni = (v.x == rep) & (v.y == rep) & (v.z == rep) & (v.w == rep);
    if (uni) {
        atomicAdd(&h[hist_hot_index(b, hot0, hot1, side)], one * 16u);
        return;
    }
    hist_add_bytes_hot(h, v.x, one, hot0, hot1, side);
    hist_add_bytes_hot(h, v.y, one, hot0, hot1, side);
    hist_add_bytes_hot(h, v.z, one, hot0, hot1, side);
    hist_add_bytes_hot(h, v.w, one, hot0, hot1, side);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void hist256_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                          uint64_t blocksize, uint32_t *__restrict__ hist)
{
    /* HIST_COPIES private histograms per wavefront, selected by lane: hot symbols of skewed data
     * then collide HIST_COPIES times less inside one ds_add (SQ_LDS_BANK_CONFLICT was 82 % of the
     * LDS cycles with one copy on Zipf data) */
    constexpr int WAVES = THREADS / 64;
    constexpr int COPIES = WAVES * HIST_COPIES;
    __shared__ uint32_t s_hist[COPIES * HUF_NSYM];

    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    const int tid = (int)threadIdx.x;

    for (int i = tid; i < COPIES * HUF_NSYM; i += THREADS) s_hist[i] = 0;
    __syncthreads();

    uint32_t *mine = s_hist + ((tid >> 6) * HIST_COPIES + (tid & (HIST_COPIES - 1))) * HUF_NSYM;
    const uint8_t *p = in + base;
    const uint64_t head = dmin<uint64_t>(len, (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    if ((uint64_t)tid < head) atomicAdd(&mine[p[tid]], 1u);

    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    uint64_t i = (uint64_t)tid;
    for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {           /* four loads in flight per lane */
        const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                    v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
        hist_add_chunk(mine, v0);
        hist_add_chunk(mine, v1);
        hist_add_chunk(mine, v2);
        hist_add_chunk(mine, v3);
    }
    for (; i < nvec; i += THREADS) hist_add_chunk(mine, load_stream16(q + i));

    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < len) atomicAdd(&mine[p[tail0 + tid]], 1u);   /* < 16 bytes */
    __syncthreads();

    for (int b = tid; b < HUF_NSYM; b += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < COPIES; w++) sum += s_hist[w * HUF_NSYM + b];
        hist[blk * HUF_NSYM + b] = sum;
    }
}

}  // namespace hufgpu
