/* hist_tree.hpp - hist_tree_kernel: histogram + tree + size sums of a block in one launch.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "histogram.hpp"
#include "tree.hpp"
#include "offsets.hpp"

namespace hufgpu {


#ifndef HT_COPIES
#define HT_COPIES 2
#endif

/* hist256 + tree in one launch: the block's byte counts never leave the CU.  All waves count;
 * then waves 1.. retire and wave 0 builds the tree in the LDS the histogram copies occupied.
 * The tree rounds are latency bound and the counting is memory bound, so on a CU the tree of
 * one block runs under the counting of the next ones.  The wave that finishes a group of blocks
 * last also prefix-sums the group's encoded sizes (no scan launch between this kernel and pack). */
#ifndef HTP_ARRAYS
#define HTP_ARRAYS 2        /* packed mode: 256-word arrays per wave, each holding two 16-bit copies (1: 0.79, 2: 0.71, 4: 0.86 ms on Zipf) */
#endif
#define HT_PACKED_MAX_BLOCK 131072u     /* a wave counts a quarter of the block: < 65 536 per 16-bit counter */

/* PACKED: the block is at most HT_PACKED_MAX_BLOCK bytes, so the private histograms use 16-bit
 * counters, two per word (lane parity picks the half): four copies per wave in the LDS of two
 * (hot symbols of skewed data collide half as often).  The totals are accumulated in place in the
 * last array, which lies behind the 7 KiB TreeLds: 8 KiB per workgroup = 20 resident groups per
 * CU instead of 13, and the latency-bound tree waves are what the kernel waits for on
 * multi-symbol data (uniform bytes 0.68 -> 0.60 ms, Zipf 0.77 -> 0.71 ms per GiB). */
template <int THREADS, bool PACKED>
__global__ __launch_bounds__(THREADS) void hist_tree_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                            uint64_t blocksize, hufcode_t *__restrict__ codetab,
                                                            int16_t *__restrict__ treebuf,
                                                            HufBlockMeta *__restrict__ meta, TwoLevel sizes)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int COPIES = WAVES * (PACKED ? HTP_ARRAYS : HT_COPIES);   /* 256-word arrays */
    constexpr size_t HBYTES = (size_t)COPIES * HUF_NSYM * sizeof(uint32_t);
    constexpr size_t TBYTES = HUF_NSYM * sizeof(uint32_t);
    /* totals: the last histogram array when that lies behind the tree's area (summed in place:
     * a thread reads and writes only its own bin there), else right behind the tree's area */
    constexpr size_t TOT_OFF = (HBYTES >= sizeof(TreeLds) + TBYTES) ? HBYTES - TBYTES : sizeof(TreeLds);
    constexpr size_t UBYTES = (HBYTES > TOT_OFF + TBYTES) ? HBYTES : TOT_OFF + TBYTES;
    static_assert(TOT_OFF >= sizeof(TreeLds) && TOT_OFF % 16 == 0, "totals must survive the tree's initialisation");
    __shared__ __attribute__((aligned(16))) uint8_t s_union[UBYTES];
    __shared__ uint32_t s_side[2 * THREADS];   /* a lane's counts of its wave's two most frequent bytes (see below) */
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_union);
    uint32_t *s_tot = reinterpret_cast<uint32_t *>(s_union + TOT_OFF);

    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    const int tid = (int)threadIdx.x;

    for (int i = tid; i < COPIES * HUF_NSYM; i += THREADS) s_hist[i] = 0;
    s_side[tid] = 0;
    s_side[THREADS + tid] = 0;
    __syncthreads();
    uint32_t *mine;
    uint32_t one = 1u;
    /* Array a of a wave keeps the count of byte b at word b ^ rot(a): the frequent bytes of skewed
     * data then sit in different LDS banks in different arrays.  Without it a wave's arrays all hold
     * byte 0 in bank 0, byte 1 in bank 1, ... and lanes that meet frequent bytes in one ds_add
     * collide on those banks even when they use different arrays. */
    constexpr int ARRS = PACKED ? HTP_ARRAYS : HT_COPIES;
    const uint32_t arr = PACKED ? ((uint32_t)(tid >> 1) & (HTP_ARRAYS - 1)) : ((uint32_t)tid & (HT_COPIES - 1));
    const uint32_t rot = arr * (32u / ARRS);
    const uint32_t rot4 = rot * 0x01010101u;
    if (PACKED) {
        mine = s_hist + ((tid >> 6) * HTP_ARRAYS + arr) * HUF_NSYM;
        one = (tid & 1) ? 0x10000u : 1u;
    } else {
        mine = s_hist + ((tid >> 6) * HT_COPIES + arr) * HUF_NSYM;
    }
    auto rotated = [rot4](uint4 v) { return make_uint4(v.x ^ rot4, v.y ^ rot4, v.z ^ rot4, v.w ^ rot4); };
    const uint8_t *p = in + base;
    const uint64_t head = dmin<uint64_t>(len, (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    if ((uint64_t)tid < head) atomicAdd(&mine[p[tid] ^ rot], one);
    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    uint64_t i = (uint64_t)tid;
    uint32_t hot = 0x100u, hot1 = 0x100u;                        /* no byte is singled out yet */
    const int side = (int)(&s_side[tid] - mine);           /* the lane's word for hot0; hot1's is THREADS words on */
    if (i + 3 * THREADS < nvec) {                                /* the first four chunks, then a look at the counts */
        const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                    v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
        hist_add_chunk(mine, v0, one, rot4);
        hist_add_chunk(mine, v1, one, rot4);
        hist_add_chunk(mine, v2, one, rot4);
        hist_add_chunk(mine, v3, one, rot4);
        i += 4 * THREADS;
        /* the wave's two most frequent bytes so far (its own copies; a wave's LDS operations are
         * in order): from 1/16 of the 4 KiB seen on, their occurrences are counted per lane */
        constexpr int ARR = PACKED ? HTP_ARRAYS : HT_COPIES;
        const uint32_t *wave_hist = s_hist + (tid >> 6) * ARR * HUF_NSYM;
        uint32_t cand[4];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const uint32_t bin = (uint32_t)(tid & 63) + 64u * qd;
            uint32_t c = 0;
#pragma unroll
            for (int a = 0; a < ARR; a++) {
                const uint32_t x = wave_hist[a * HUF_NSYM + (bin ^ (uint32_t)(a * (32 / ARR)))];
                c += PACKED ? ((x & 0xffffu) + (x >> 16)) : x;
            }
            cand[qd] = (c << 8) | bin;
        }
        uint32_t best = dmax(dmax(cand[0], cand[1]), dmax(cand[2], cand[3]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) best = dmax(best, wave_xor_any(best, o));
        if ((best >> 8) >= 256u && (best >> 8) < 4096u) {        /* (all 4 096 equal: the run paths take it) */
            hot = best & 0xffu;
            uint32_t second = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) second = dmax(second, cand[qd] == best ? 0u : cand[qd]);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) second = dmax(second, wave_xor_any(second, o));
            if ((second >> 8) >= 256u) hot1 = second & 0xffu;
        }
    }
    if (hot < 0x100u) {
        for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {
            const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                        v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
            hist_add_chunk_hot(mine, rotated(v0), one, hot ^ rot, hot1 ^ rot, side);
            hist_add_chunk_hot(mine, rotated(v1), one, hot ^ rot, hot1 ^ rot, side);
            hist_add_chunk_hot(mine, rotated(v2), one, hot ^ rot, hot1 ^ rot, side);
            hist_add_chunk_hot(mine, rotated(v3), one, hot ^ rot, hot1 ^ rot, side);
        }
        for (; i < nvec; i += THREADS) hist_add_chunk_hot(mine, rotated(load_stream16(q + i)), one, hot ^ rot, hot1 ^ rot, side);
        atomicAdd(&mine[hot ^ rot], s_side[tid]);                /* a lane's own words: the values are final */
        if (hot1 < 0x100u) atomicAdd(&mine[hot1 ^ rot], s_side[THREADS + tid]);
    }
    for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {           /* four loads in flight per lane */
        const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                    v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
        hist_add_chunk(mine, v0, one, rot4);
        hist_add_chunk(mine, v1, one, rot4);
        hist_add_chunk(mine, v2, one, rot4);
        hist_add_chunk(mine, v3, one, rot4);
    }
    for (; i < nvec; i += THREADS) hist_add_chunk(mine, load_stream16(q + i), one, rot4);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < len) atomicAdd(&mine[p[tail0 + tid] ^ rot], one);   /* < 16 bytes */
    __syncthreads();
    for (int b = tid; b < HUF_NSYM; b += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < COPIES; w++) {
            const uint32_t x = s_hist[w * HUF_NSYM + (b ^ ((w % ARRS) * (32 / ARRS)))];
            sum += PACKED ? ((x & 0xffffu) + (x >> 16)) : x;
        }
        s_tot[b] = sum;
    }
    __syncthreads();                     /* copies are dead from here on; s_tot is complete */
    if (tid >= 64) return;               /* ended waves do not take part in anything below */
    uint32_t rate[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rate[j] = s_tot[(tid & 63) + 64 * j];
    const uint64_t bytes = tree_fast_wave(rate, *reinterpret_cast<TreeLds *>(s_union), blk, codetab, treebuf, meta);
    /* stream offsets (the reference's running file position): summed here, see two_level_arrive */
    two_level_arrive(sizes, blk, gridDim.x, bytes);
}

}  // namespace hufgpu
