/* hist_lanes.hpp - hist_lanes_kernel: per-block byte counts with LANE-PRIVATE counters (src/histogram.c:73-103),
   and tree_wave_kernel: the tree of a block from those counts (src/tree.c:292-427) as a launch of its own.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "util.hpp"
#include "tree.hpp"
#include "offsets.hpp"

namespace hufgpu {

/* ======================================================================================
 * hist_lanes_kernel - the 256 counts of every block, one workgroup per block.
 *
 * The fused kernel (hist_tree.hpp) counts with LDS atomics on a few shared copies: lanes that meet the
 * same byte value - or just the same LDS bank - in one ds_add serialise (63 % of its LDS cycles were
 * conflicts), and taking the two hottest values out of that costs four VALU instructions per byte.
 * Here every LANE has a counter of its own for every byte value: row b of the 64 KiB array holds the 64
 * lanes' counters of byte b, so a ds_add of 64 lanes touches 64 different words in 64 consecutive banks'
 * worth of addresses - two passes of the 32 banks, never more, whatever the data is (runs of one byte
 * included).  The LDS address of (byte, lane) is byte << 8 | lane << 2: ONE v_perm_b32 per byte puts
 * byte k of the loaded dword into bits 8..15 over the lane's column offset - no extract, no shift, no
 * compare.  All waves of the workgroup share the array (lane l of every wave adds to the same word; the
 * adds are atomic), the rows are summed at the end with 16-byte reads and DPP row sums.
 * LDS: 64 KiB in round 3 (two workgroups of 512 threads per CU), 32 KiB since round 4 (16-bit counters, below): three
 * workgroups per CU - its 72 registers a lane set that now; every thread has all of its loads in flight
 * before it counts (128 bytes per thread at 64 KiB blocks), so the kernel runs at the rate HBM delivers.
 * ==================================================================================== */
#define HL_THREADS 512
#define HL_MIN_BLOCK 32768u       /* below this the 64 KiB of counters cost more to zero and sum than the block to count: the fused kernel */
/* Round 4: 16-bit counters.  Lanes l and l + 32 of a wave share a word (low and high half); an LDS instruction serves
 * lanes 0-31 and 32-63 in different cycles anyway, so the 32 lanes of a half still meet 32 different banks - two passes
 * per ds_add as before - and the array is 32 KiB: four workgroups per CU instead of two, half as much to zero and to
 * sum (the registers, not the LDS, then decide: three workgroups).  A counter sees lane l of all eight waves: 1 024 bytes of a 64 KiB block, 32 768 of the largest block that
 * comes here (below 2 MiB; chunks of 256 KiB above that) - 16 bits hold it.  The address is byte << 7 | (lane & 31) << 2:
 * the v_perm puts the byte over twice the column, one shift halves both. */
#ifndef HL_AHEAD
#define HL_AHEAD 8                /* 16-byte vectors a thread requests before it counts the first */
#endif
#ifndef HL_WAVES_PER_SIMD
#define HL_WAVES_PER_SIMD 4       /* (the least the compiler has to leave room for: 8 waves a workgroup, two workgroups per CU; with 72 registers
                                     it is three) */
#endif
#define HL_LDS_BYTES (HUF_NSYM * 32 * 4)
#define HL_MAX_PER_COUNTER 65535u      /* what a 16-bit counter holds: a counter sees the bytes of one lane of all the waves - a 64th of what a
                                          workgroup counts, + the up to 31 bytes a lane takes in front of and behind the vectors (asserted
                                          at the kernels: hufgpu_api.hip's block sizes for them) */

/* col = (lane & 31) << 3 (twice the column's byte offset), one = 1 or 1 << 16 (the lane's half of the word) */
__device__ __forceinline__ void hl_add_dword(uint8_t *hl_lds, uint32_t w, uint32_t col, uint32_t one)
{
    /* D.byte0 = col.byte0, D.byte1 = w.byte k, D.byte2 = D.byte3 = 0; >> 1: byte << 7 | (lane & 31) << 2 */
    atomicAdd(reinterpret_cast<uint32_t *>(hl_lds + (__builtin_amdgcn_perm(w, col, 0x0c0c0400u) >> 1)), one);
    atomicAdd(reinterpret_cast<uint32_t *>(hl_lds + (__builtin_amdgcn_perm(w, col, 0x0c0c0500u) >> 1)), one);
    atomicAdd(reinterpret_cast<uint32_t *>(hl_lds + (__builtin_amdgcn_perm(w, col, 0x0c0c0600u) >> 1)), one);
    atomicAdd(reinterpret_cast<uint32_t *>(hl_lds + (__builtin_amdgcn_perm(w, col, 0x0c0c0700u) >> 1)), one);
}
__device__ __forceinline__ void hl_add_byte(uint8_t *hl_lds, uint32_t byte, uint32_t col, uint32_t one)
{
    atomicAdd(reinterpret_cast<uint32_t *>(hl_lds + (((byte << 8) | col) >> 1)), one);
}

__device__ __forceinline__ void hl_add_vec(uint8_t *hl_lds, uint4 v, uint32_t col, uint32_t one)
{
    hl_add_dword(hl_lds, v.x, col, one);
    hl_add_dword(hl_lds, v.y, col, one);
    hl_add_dword(hl_lds, v.z, col, one);
    hl_add_dword(hl_lds, v.w, col, one);
}

/* the 256 counts of `len` bytes at `p` -> out[256] (the whole workgroup; hl_lds = its 64 KiB) */
template <int THREADS>
__device__ __forceinline__ void hl_count(uint8_t *hl_lds, const uint8_t *__restrict__ p, uint64_t len, uint32_t *__restrict__ out)
{
    const int tid = (int)threadIdx.x;
    const uint32_t col = (uint32_t)(tid & 31) << 3;
    const uint32_t one = (tid & 32) ? 0x10000u : 1u;

    {
        uint4 *z = reinterpret_cast<uint4 *>(hl_lds);
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int i = 0; i < HL_LDS_BYTES / 16 / THREADS; i++) z[i * THREADS + tid] = zero;
    }
    const uint64_t head = dmin<uint64_t>(len, (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    /* the first eight vectors of every thread are requested before the zeroed array is waited for */
    uint4 v[HL_AHEAD];
    uint64_t i = (uint64_t)tid;
    const bool full8 = i + (HL_AHEAD - 1) * THREADS < nvec;
    if (full8) {
#pragma unroll
        for (int k = 0; k < HL_AHEAD; k++) v[k] = load_stream16(q + i + (uint64_t)k * THREADS);
    }
    __syncthreads();
    if ((uint64_t)tid < head) hl_add_byte(hl_lds, (uint32_t)p[tid], col, one);
    if (full8) {
#pragma unroll
        for (int k = 0; k < HL_AHEAD; k++) hl_add_vec(hl_lds, v[k], col, one);
        i += HL_AHEAD * THREADS;
    }
    for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {
        const uint4 a = load_stream16(q + i), b = load_stream16(q + i + THREADS), c = load_stream16(q + i + 2 * THREADS),
                    d = load_stream16(q + i + 3 * THREADS);
        hl_add_vec(hl_lds, a, col, one);
        hl_add_vec(hl_lds, b, col, one);
        hl_add_vec(hl_lds, c, col, one);
        hl_add_vec(hl_lds, d, col, one);
    }
    for (; i < nvec; i += THREADS) hl_add_vec(hl_lds, load_stream16(q + i), col, one);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < len) hl_add_byte(hl_lds, (uint32_t)p[tail0 + tid], col, one);
    __syncthreads();
    /* row sums: a wave takes four rows per step - lane l reads words 4 (l % 16) .. + 3 of row (l / 16) - adds its
     * four words and then the sixteen lanes of the row (DPP row_shr 1, 2, 4, 8: the sum ends in the row's lane 15) */
    constexpr int WAVES = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
    /* a row is 32 words of two 16-bit counters: lane l reads words 2 (l % 16), + 1 of row (l / 16) - a wave four rows a step */
#pragma unroll
    for (int it = 0; it < HUF_NSYM / (4 * WAVES); it++) {
        const int row = (it * WAVES + wave) * 4 + (lane >> 4);
        const uint2 x = *reinterpret_cast<const uint2 *>(hl_lds + row * 128 + (lane & 15) * 8);
        uint32_t s = (x.x & 0xffffu) + (x.x >> 16) + (x.y & 0xffffu) + (x.y >> 16);
        s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
        s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
        s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, true);    /* row_shr:4 */
        s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xf, true);    /* row_shr:8 */
        if ((lane & 15) == 15) out[row] = s;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS, HL_WAVES_PER_SIMD) void hist_lanes_kernel(const uint8_t *__restrict__ in, uint64_t n, uint64_t blocksize,
                                                             uint32_t *__restrict__ hist)
{
    __shared__ __attribute__((aligned(16))) uint8_t hl_lds[HL_LDS_BYTES];      /* [256 byte values][32 columns] words of two 16-bit counters (round 3: [256][64] uint32) */
    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    hl_count<THREADS>(hl_lds, in + base, len, hist + blk * HUF_NSYM);
}

/* tree_wave_kernel - tree_fast_wave (tree.hpp) as a launch of its own: one wavefront per block, counts from
 * hist_lanes_kernel; also sums the encoded sizes (two_level_arrive), like the fused kernel's tree wave. */
__global__ __launch_bounds__(64, 8) void tree_wave_kernel(const uint32_t *__restrict__ hist, hufcode_t *__restrict__ codetab,
                                                       int16_t *__restrict__ treebuf, HufBlockMeta *__restrict__ meta,
                                                       TwoLevel sizes)
{
    __shared__ TreeLds L;
    const uint64_t blk = blockIdx.x;
    const int lane = lane_id();
    uint32_t rate[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rate[j] = hist[blk * HUF_NSYM + lane + 64 * j];
    const uint64_t bytes = tree_fast_wave(rate, L, blk, codetab, treebuf, meta);
    two_level_arrive(sizes, blk, gridDim.x, bytes);
}

}  // namespace hufgpu
