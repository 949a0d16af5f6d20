/* spec_index.hpp - a sub-index for a block of many MiB that came WITHOUT one (a raw stream through
   huf_decode(), src/decoder.c:205-283, written with blocksize = 0 or a blocksize of many MiB):
   spec_head_kernel, spec_scan_kernel, spec_prefix_kernel, spec_mark_kernel, spec_groups_kernel.

   The block's payload is cut into lanes of SPEC_LANE_BITS bits.  Every lane starts decoding
   SPEC_OVERLAP bits before its share: a Huffman decoder that starts in the middle of a codeword
   falls into step with the true codewords after a few symbols, so at its share's first bit the
   lane is (almost surely) on a codeword boundary.  Pass 1 (scan) counts the codewords that begin
   in every share; a prefix sum turns that into the symbol index each lane starts with; pass 2
   (mark) decodes the shares again and writes down where every 32nd symbol begins - which is the
   sub-index the encoder would have written (pack.hpp).  Nothing here is trusted: the result goes
   to decode_sub_kernel, which verifies every group and every chunk against the payload and hands
   the block to the exact decoder if anything is off; a lane that did not fall into step shows as
   a broken chain (exit of lane i != entry of lane i + 1) and the whole attempt is dropped.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "decode.hpp"
#include "offsets.hpp"
#include "pack.hpp"

namespace hufgpu {

#define SPEC_LANE_BITS 4096u
#define SPEC_OVERLAP   512u

/* status words of one attempt */
enum { SPEC_FAIL = 0, SPEC_END_BITS = 1, SPEC_FOUND = 2, SPEC_BLOCK_LEN = 3, SPEC_TREE_LEN = 4, SPEC_LEAF = 5, SPEC_WORDS = 8 };

struct SpecJob {
    const uint8_t *tree;       /* the block's serialized tree */
    int tree_len;
    const uint8_t *pay;        /* its payload */
    uint64_t pay_bytes;        /* readable bytes from there (to the end of what the caller holds) */
    uint64_t max_bits;         /* payload bits the lanes cover */
    uint64_t block_len;
    uint64_t nlanes;
    uint64_t *entry;           /* first codeword start at or after the share's first bit */
    uint64_t *exitp;           /* first codeword start at or after the share's end */
    uint32_t *cnt;             /* codewords that start inside the share */
    uint64_t *pre;             /* nlanes + 1: symbols before each share */
    uint64_t *gstart;          /* payload bit of every 32nd symbol */
    unsigned long long *status;
};

/* header of the block at stream[pos...): length, tree length and - for the five-entry tree of a
 * one-symbol block - its byte */
__global__ __launch_bounds__(64) void spec_head_kernel(const uint8_t *__restrict__ stream, uint64_t avail, uint64_t pos,
                                                       unsigned long long *__restrict__ status)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < SPEC_WORDS; i++) status[i] = 0;
    status[SPEC_LEAF] = ~0ull;
    if (avail - pos < HUF_HEADER_FIXED) { status[SPEC_FAIL] = 1; return; }
    uint64_t bl = 0;
    for (int i = 0; i < 8; i++) bl |= (uint64_t)stream[pos + i] << (8 * i);
    const int16_t tl = (int16_t)((uint32_t)stream[pos + 8] | ((uint32_t)stream[pos + 9] << 8));
    status[SPEC_BLOCK_LEN] = bl;
    status[SPEC_TREE_LEN] = (unsigned long long)(long long)tl;
    if (tl == 5 && avail - pos >= HUF_HEADER_FIXED + 10) {
        const int leaf = single_leaf_symbol(stream + pos + HUF_HEADER_FIXED);
        if (leaf >= 0) status[SPEC_LEAF] = (unsigned long long)leaf;
    }
}

/* MSB-first bit source on global memory, one per lane (64-bit positions) */
struct GlobReader {
    const uint8_t *pay;
    uint64_t nbytes;
    uint64_t b;          /* the bits at the position, left aligned */
    int32_t avail;       /* valid bits in b */
    uint64_t gf;         /* next 32-bit word to append */
    __device__ __forceinline__ void load(uint64_t pos)
    {
        const uint64_t g = pos >> 5;
        const uint32_t o = (uint32_t)(pos & 31u);
        const uint64_t w = ((uint64_t)load_be32(pay, 4 * g, nbytes) << 32) | load_be32(pay, 4 * g + 4, nbytes);
        b = w << o;
        avail = 64 - (int32_t)o;
        gf = g + 2;
    }
    __device__ __forceinline__ uint32_t index() const { return (uint32_t)(b >> (64 - DEC_LUT_BITS)); }
    __device__ __forceinline__ uint32_t top() const { return (uint32_t)(b >> 63); }
    __device__ __forceinline__ void consume(uint32_t adv)      /* adv <= 31 */
    {
        b <<= adv;
        avail -= (int32_t)adv;
        if (avail <= 32) {
            b |= (uint64_t)load_be32(pay, 4 * gf, nbytes) << (32 - avail);
            avail += 32;
            gf++;
        }
    }
};

/* one table lookup at pos; true = a codeword was taken.  A walk that leaves the tree resumes a
 * few bits on (the table's `skip`), as the self-synchronising decoder does (decode.hpp). */
template <int THREADS>
__device__ __forceinline__ bool spec_step(const DecShared<THREADS> &sh, GlobReader &rd, uint64_t &pos)
{
    const uint32_t e = sh.lut[rd.index()];
    if (e < DEC_E_LONG) {
        const uint32_t adv = dec_e_adv(e);
        pos += adv;
        rd.consume(adv);
        return e < DEC_E_BAD;
    }
    uint32_t node = e & 0x7ffu;
    const uint64_t p0 = pos;
    rd.consume(DEC_LUT_BITS);
    pos += DEC_LUT_BITS;
    for (;;) {
        const uint32_t bit = rd.top();
        rd.consume(1);
        pos++;
        const uint32_t nx = dec_child(sh.lr[node], bit);
        if (nx == DEC_NULL) {
            pos = p0 + 1;
            rd.load(pos);
            return false;
        }
        node = nx;
        if (sh.lr[node] == DEC_LEAF_LR) return true;
    }
}

/* pass 1: entry, exit and codeword count of every share */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_scan_kernel(SpecJob j)
{
    __shared__ DecShared<THREADS> sh;
    int leaf;
    const int rc = dec_build_tables<THREADS, false>(sh, j.tree, j.tree_len, &leaf);
    if (rc != HUFE_OK || leaf >= 0) {
        if (threadIdx.x == 0) j.status[SPEC_FAIL] = 1;
        return;
    }
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (lane >= j.nlanes) return;
    const uint64_t lo = lane * SPEC_LANE_BITS;
    const uint64_t hi = dmin<uint64_t>(lo + SPEC_LANE_BITS, j.max_bits);
    uint64_t pos = lo > SPEC_OVERLAP ? lo - SPEC_OVERLAP : 0;
    GlobReader rd;
    rd.pay = j.pay;
    rd.nbytes = j.pay_bytes;
    rd.load(pos);
    while (pos < lo) (void)spec_step<THREADS>(sh, rd, pos);
    j.entry[lane] = pos;
    uint32_t c = 0;
    while (pos < hi) c += spec_step<THREADS>(sh, rd, pos) ? 1u : 0u;
    j.exitp[lane] = pos;
    j.cnt[lane] = c;
}

/* symbols before every share */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_prefix_kernel(SpecJob j)
{
    const uint32_t *cnt = j.cnt;
    const uint64_t total = chunked_excl_scan<THREADS>(j.nlanes, j.pre, [cnt](uint64_t i) { return (uint64_t)cnt[i]; });
    if (threadIdx.x == 0) j.pre[j.nlanes] = total;
}

/* pass 2: where every 32nd symbol begins, and where the block's last symbol ends */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_mark_kernel(SpecJob j)
{
    __shared__ DecShared<THREADS> sh;
    int leaf;
    const int rc = dec_build_tables<THREADS, false>(sh, j.tree, j.tree_len, &leaf);
    if (rc != HUFE_OK || leaf >= 0) return;               /* pass 1 said so already */
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (lane >= j.nlanes) return;
    uint64_t s = j.pre[lane];
    if (s >= j.block_len) return;                         /* the block ends before this share */
    uint64_t pos = j.entry[lane];
    /* the chain: this lane begins where the one before it ended (lane 0 begins at bit 0) */
    if (lane > 0 ? (j.exitp[lane - 1] != pos) : (pos != 0)) {
        j.status[SPEC_FAIL] = 1;
        return;
    }
    if (lane + 1 == j.nlanes && s + j.cnt[lane] < j.block_len) {
        j.status[SPEC_FAIL] = 1;                          /* the block does not end inside the covered bits */
        return;
    }
    const uint64_t stop = j.exitp[lane];
    GlobReader rd;
    rd.pay = j.pay;
    rd.nbytes = j.pay_bytes;
    rd.load(pos);
    while (pos < stop) {
        if ((s & (HUF_SUB_GROUP - 1)) == 0) j.gstart[s / HUF_SUB_GROUP] = pos;
        if (spec_step<THREADS>(sh, rd, pos)) {
            s++;
            if (s == j.block_len) {
                j.status[SPEC_END_BITS] = pos;
                j.status[SPEC_FOUND] = 1;
                break;
            }
        }
    }
}

/* the sub-index (HufSubIndex of ONE block, blocksize = block_len) from the group starts, and the
 * two-entry block index [pos, pos + encoded size) */
__global__ __launch_bounds__(256) void spec_groups_kernel(SpecJob j, HufSubIndex sub, uint64_t pos, uint64_t pay_off,
                                                          uint64_t *__restrict__ offs)
{
    const uint64_t ngroups = (j.block_len + HUF_SUB_GROUP - 1) / HUF_SUB_GROUP;
    const uint64_t end_bits = j.status[SPEC_END_BITS];
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < ngroups) {
        const uint64_t a = j.gstart[g];
        const uint64_t b = (g + 1 < ngroups) ? j.gstart[g + 1] : end_bits;
        const uint64_t d = b - a;
        sub.group_bits[g] = (uint16_t)(d > 0xffffull ? 0xffffull : d);
        if ((g & (HUF_SUB_TILE / HUF_SUB_GROUP - 1)) == 0) sub.tile_bits[g / (HUF_SUB_TILE / HUF_SUB_GROUP)] = a;
    } else if (g < sub.gpb) {
        sub.group_bits[g] = 0;
    }
    if (blockIdx.x == 0) {
        sub.lens[threadIdx.x] = 0;                        /* no code lengths: the decoder builds its tables from the tree */
        if (threadIdx.x == 0) {
            offs[0] = pos;
            offs[1] = pay_off + ((end_bits + 7) >> 3);
        }
    }
}

}  // namespace hufgpu
