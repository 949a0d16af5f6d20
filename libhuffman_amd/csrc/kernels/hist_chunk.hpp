/* hist_chunk.hpp - the histogram side of chunked blocks: chunk_hist_kernel, block_hist_kernel,
   chunk_total_kernel, chunk_scan_kernel (src/histogram.c:73-103 for blocks of many MiB).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "histogram.hpp"
#include "offsets.hpp"
#include "hist_lanes.hpp"

namespace hufgpu {

/* A block of blocksize >= HUF_BIG_BLOCK bytes is cut into chunks of HUF_CHUNK_SYMS symbols (the
 * stream's last block may have fewer).  Chunk c of block b = input bytes
 * [b * blocksize + c * HUF_CHUNK_SYMS, ...).  cpb = chunks per block. */
#define HUF_CHUNK_SYMS 262144u          /* 32 pack tiles: the sub-index' tiles and groups align with chunks */
#define HUF_BIG_BLOCK  (1ull << 22)     /* from here on 64-bit tree keys (a rate may reach 2^23) and, when decoding raw streams, a sub-index built on the device */
#define HUF_CHUNKED_FROM (1ull << 21)   /* from here on blocks are ENCODED in chunks: one workgroup per block leaves most of the 256 CUs
                                           without work when a GiB is only a few hundred blocks (pack at 2 / 3 MiB blocks: 0.81 / 1.19 ms per
                                           GiB, chunked 0.57; at 1 MiB both forms take 0.64) */

/* hist_lanes.hpp counts with 16-bit counters, one per lane and byte value: a counter sees a 64th of what a workgroup counts
 * (+ 32 bytes of head and tail) - whole blocks below HUF_CHUNKED_FROM, chunks of HUF_CHUNK_SYMS above it */
static_assert(HUF_CHUNKED_FROM / 64 + 32 <= HL_MAX_PER_COUNTER && HUF_CHUNK_SYMS / 64 + 32 <= HL_MAX_PER_COUNTER,
              "a lane's 16-bit counter would overflow into its neighbour's");

struct ChunkGeom {
    uint64_t n, blocksize;
    uint32_t cpb;
    __device__ __forceinline__ bool locate(uint64_t chunk, uint64_t &base, uint64_t &len) const
    {
        const uint64_t blk = chunk / cpb, c = chunk % cpb;
        const uint64_t b0 = blk * blocksize;
        const uint64_t blen = dmin<uint64_t>(blocksize, n - b0);
        const uint64_t s0 = c * (uint64_t)HUF_CHUNK_SYMS;
        if (s0 >= blen) { base = 0; len = 0; return false; }
        base = b0 + s0;
        len = dmin<uint64_t>(HUF_CHUNK_SYMS, blen - s0);
        return true;
    }
};

/* byte counts of every chunk: lane-private counters (hist_lanes.hpp: hl_count), one workgroup per chunk */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void chunk_hist_kernel(const uint8_t *__restrict__ in, ChunkGeom geo,
                                                             uint32_t *__restrict__ chunk_hist)
{
    __shared__ __attribute__((aligned(16))) uint8_t hl_lds[HL_LDS_BYTES];
    uint64_t base, len;
    if (!geo.locate(blockIdx.x, base, len)) {            /* (a chunk behind the stream's last, short block) */
        for (int b = (int)threadIdx.x; b < HUF_NSYM; b += THREADS) chunk_hist[(uint64_t)blockIdx.x * HUF_NSYM + b] = 0;
        return;
    }
    hl_count<THREADS>(hl_lds, in + base, len, chunk_hist + (uint64_t)blockIdx.x * HUF_NSYM);
}

/* byte counts of every block = the sums over its chunks (64-bit: a block may be longer than 2^32 bytes) */
__global__ __launch_bounds__(HUF_NSYM) void block_hist_kernel(const uint32_t *__restrict__ chunk_hist, uint32_t cpb,
                                                              uint64_t *__restrict__ hist)
{
    /* (eight registers, all of its allocation, and a 64-bit shift by the last of them: the gfx950 hazard of
     *  DESIGN.md 3.3, caught by the build's ISA check - one register of slack) */
    asm volatile("; one VGPR more than the kernel uses" ::: "v8");
    const uint64_t blk = blockIdx.x;
    uint64_t sum = 0;
    for (uint32_t c = 0; c < cpb; c++) sum += chunk_hist[(blk * cpb + c) * HUF_NSYM + threadIdx.x];
    hist[blk * HUF_NSYM + threadIdx.x] = sum;
}

/* the same with 32-bit sums (blocks below HUF_BIG_BLOCK: the wave-per-block tree of tree.hpp takes those) */
__global__ __launch_bounds__(HUF_NSYM) void block_hist32_kernel(const uint32_t *__restrict__ chunk_hist, uint32_t cpb,
                                                                uint32_t *__restrict__ hist)
{
    const uint64_t blk = blockIdx.x;
    uint32_t sum = 0;
    for (uint32_t c = 0; c < cpb; c++) sum += chunk_hist[(blk * cpb + c) * HUF_NSYM + threadIdx.x];
    hist[blk * HUF_NSYM + threadIdx.x] = sum;
}

/* payload bits of every chunk = sum over the bytes of count * code length (one wave per chunk) */
__global__ __launch_bounds__(64) void chunk_total_kernel(const uint32_t *__restrict__ chunk_hist, uint32_t cpb,
                                                         const hufcode_t *__restrict__ codetab, const HufBlockMeta *__restrict__ meta,
                                                         uint64_t *__restrict__ chunk_tot)
{
    asm volatile("; one VGPR more than the kernel uses (the build's ISA check: a 64-bit shift by the last of sixteen)" ::: "v16");
    const uint64_t chunk = blockIdx.x, blk = chunk / cpb;
    /* a one-symbol block has no code table (the wave-per-block tree does not write one): one bit a symbol */
    const bool one = meta[blk].tree_len == 5;
    uint64_t bits = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int s = lane_id() + 64 * j;
        bits += (uint64_t)chunk_hist[chunk * HUF_NSYM + s] * (one ? 1ull : (uint64_t)(codetab[blk * HUF_NSYM + s] & 0xffu));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) bits += shfl_xor_u64(bits, o);
    if (lane_id() == 0) chunk_tot[chunk] = bits;
}

/* first payload bit of every chunk = exclusive sums of the chunk totals inside each block */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void chunk_scan_kernel(const uint64_t *__restrict__ chunk_tot, uint32_t cpb,
                                                             uint64_t *__restrict__ chunk_bits)
{
    const uint64_t blk = blockIdx.x;
    const uint64_t *src = chunk_tot + blk * cpb;
    (void)chunked_excl_scan<THREADS>(cpb, chunk_bits + blk * cpb, [src](uint64_t i) { return src[i]; });
}

}  // namespace hufgpu
