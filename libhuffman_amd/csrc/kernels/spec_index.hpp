/* spec_index.hpp - a sub-index for a block of many MiB that came WITHOUT one (a raw stream through
   huf_decode(), src/decoder.c:205-283, written with blocksize = 0 or a blocksize of many MiB):
   spec_head_kernel, spec_scan_kernel, spec_prefix_kernel, spec_mark_kernel, spec_groups_kernel.

   The block's payload is cut into lanes of SPEC_LANE_BITS bits.  Every lane starts decoding
   SPEC_OVERLAP bits before its share: a Huffman decoder that starts in the middle of a codeword
   falls into step with the true codewords after a few symbols, so at its share's first bit the
   lane is (almost surely) on a codeword boundary.  Pass 1 (scan) counts the codewords that begin
   in every share; a prefix sum turns that into the symbol index each lane starts with; pass 2
   (mark) decodes the shares again and writes down the bits between every 32nd symbol's start and the
   next - which is the sub-index the encoder would have written (pack.hpp).  Nothing here is trusted: the result goes
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
#define SPEC_REPAIR_ROUNDS 1024    /* launches spent on mending a broken chain before the general path takes the block ... */
#define SPEC_REPAIR_BATCH  16      /* ... checked by the host once per this many */

/* status words of one attempt */
enum { SPEC_FAIL = 0, SPEC_END_BITS = 1, SPEC_FOUND = 2, SPEC_BLOCK_LEN = 3, SPEC_TREE_LEN = 4, SPEC_LEAF = 5, SPEC_END_LANE = 6, SPEC_END_OPEN = 7,
       SPEC_CHAIN = 8, SPEC_REPAIRED = 9, SPEC_SHORT = 10, SPEC_WORDS = 11 };

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
    uint64_t *pre;             /* symbols before each share inside its workgroup's shares */
    uint64_t *wg_pre;          /* workgroups + 1: codeword counts of the workgroups, then their exclusive sums (the last = all) */
    uint64_t *first_pos;       /* per share: payload bit of the first group start inside it (a group = 32 symbols) ... */
    uint64_t *first_g;         /* ... and that group's index; ~0: the share holds no group start */
    uint64_t *last_pos;        /* per share: payload bit of the last group start inside it (~0: none) */
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

/* A lane's window on the payload: SPEC_COL_WORDS big-endian words from the word that holds its
 * position, staged in a COLUMN of the wave's LDS slice (word k of lane l at slice[k * 64 + l]: the
 * lanes of a wave read and write different banks whatever their positions are).  The lane loads them
 * itself (three 16-byte loads and a 4-byte one at a 4-byte aligned address, one v_perm per word for
 * byte order and the payload's misalignment), decodes until fewer than 64 bits are left in front of
 * the staged end, and stages again from where it stands.  No lane waits for a load inside the symbol
 * loop (a refilled bit buffer made the whole wave wait for one lane's refill at nearly every symbol:
 * 11 ms per GiB of payload). */
#define SPEC_COL_WORDS 12u
#define SPEC_COL_GUARD 64u           /* bits kept in front of the staged end: a table codeword is 12, a walk gives up there */

struct ColStage {
    uint32_t *col;             /* &slice[lane] */
    const uint8_t *pay;
    uint64_t nbytes;
    uint64_t g0;               /* payload word staged at col[0] */

    __device__ __forceinline__ void stage(uint64_t g)
    {
        struct __attribute__((packed, aligned(4))) Q4 { uint32_t x, y, z, w; };
        g0 = g;
        const uint64_t byte0 = 4 * g;
        if (byte0 + 4ull * SPEC_COL_WORDS + 8ull <= nbytes) {
            const uintptr_t a = (uintptr_t)(pay + byte0);
            const uint32_t m = (uint32_t)(a & 3u);
            const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
            const uint32_t *q = reinterpret_cast<const uint32_t *>(a - m);
            const Q4 v0 = *reinterpret_cast<const Q4 *>(q), v1 = *reinterpret_cast<const Q4 *>(q + 4),
                     v2 = *reinterpret_cast<const Q4 *>(q + 8);
            const uint32_t x = q[12];
            col[0 * 64] = __builtin_amdgcn_perm(v0.y, v0.x, sel);
            col[1 * 64] = __builtin_amdgcn_perm(v0.z, v0.y, sel);
            col[2 * 64] = __builtin_amdgcn_perm(v0.w, v0.z, sel);
            col[3 * 64] = __builtin_amdgcn_perm(v1.x, v0.w, sel);
            col[4 * 64] = __builtin_amdgcn_perm(v1.y, v1.x, sel);
            col[5 * 64] = __builtin_amdgcn_perm(v1.z, v1.y, sel);
            col[6 * 64] = __builtin_amdgcn_perm(v1.w, v1.z, sel);
            col[7 * 64] = __builtin_amdgcn_perm(v2.x, v1.w, sel);
            col[8 * 64] = __builtin_amdgcn_perm(v2.y, v2.x, sel);
            col[9 * 64] = __builtin_amdgcn_perm(v2.z, v2.y, sel);
            col[10 * 64] = __builtin_amdgcn_perm(v2.w, v2.z, sel);
            col[11 * 64] = __builtin_amdgcn_perm(x, v2.w, sel);
        } else {
#pragma unroll 1
            for (uint32_t k = 0; k < SPEC_COL_WORDS; k++) col[k * 64] = load_be32(pay, byte0 + 4ull * k, nbytes);
        }
    }
    /* position relative to the staged words; the caller keeps rel + SPEC_COL_GUARD <= 32 * SPEC_COL_WORDS */
    __device__ __forceinline__ uint32_t rel(uint64_t pos) const { return (uint32_t)(pos - 32ull * g0); }
    __device__ __forceinline__ bool covers(uint64_t pos) const { return rel(pos) + SPEC_COL_GUARD <= 32u * SPEC_COL_WORDS; }
    __device__ __forceinline__ uint32_t window(uint32_t r) const          /* the 32 bits at staged bit r */
    {
        const uint32_t w = r >> 5, o = r & 31u;
        const uint64_t b = (((uint64_t)col[w * 64] << 32) | col[(w + 1u) * 64]) << o;
        return (uint32_t)(b >> 32);
    }
    __device__ __forceinline__ uint32_t bit(uint32_t r) const { return (col[(r >> 5) * 64] >> (31u - (r & 31u))) & 1u; }
};

/* one table lookup at pos (the stage covers it); true = a codeword was taken.  A walk that leaves
 * the tree resumes a few bits on (the table's `skip`), as the self-synchronising decoder does
 * (decode.hpp); a walk that would leave the staged words (a code of more than 52 bits: no encoder
 * makes one for a block that fits a device) is given up the same way - whatever comes of it is
 * verified later. */
template <int THREADS>
__device__ __forceinline__ bool spec_step(const DecShared<THREADS> &sh, const ColStage &st, uint64_t &pos)
{
    const uint32_t r = st.rel(pos);
    const uint32_t e = sh.lut[st.window(r) >> (32 - DEC_LUT_BITS)];
    if (e < DEC_E_LONG) {
        pos += dec_e_adv(e);
        return e < DEC_E_BAD;
    }
    uint32_t node = e & 0x7ffu;
    uint32_t p = r + DEC_LUT_BITS;
    for (;;) {
        if (p >= 32u * SPEC_COL_WORDS) { pos += 1; return false; }
        const uint32_t nx = dec_child(sh.lr[node], st.bit(p));
        p++;
        if (nx == DEC_NULL) { pos += 1; return false; }
        node = nx;
        if (sh.lr[node] == DEC_LEAF_LR) { pos += p - r; return true; }
    }
}

template <int THREADS>
__device__ __forceinline__ ColStage spec_stage_of(DecShared<THREADS> &sh, const SpecJob &j)
{
    static_assert(sizeof(DecShared<THREADS>::pay) + sizeof(DecShared<THREADS>::mark) >= (THREADS / 64) * 64 * SPEC_COL_WORDS * 4,
                  "one column slice per wave");
    static_assert(offsetof(DecShared<THREADS>, mark) == offsetof(DecShared<THREADS>, pay) + sizeof(DecShared<THREADS>::pay), "one area");
    ColStage st;
    st.col = sh.pay + (threadIdx.x >> 6) * (64 * SPEC_COL_WORDS) + (threadIdx.x & 63u);
    st.pay = j.pay;
    st.nbytes = j.pay_bytes;
    st.g0 = 0;
    return st;
}

/* pass 1: entry, exit and codeword count of every share */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_scan_kernel(SpecJob j, uint8_t *__restrict__ lens_out)
{
    __shared__ DecShared<THREADS> sh;
    int leaf;
    const int rc = dec_build_tables<THREADS, false>(sh, j.tree, j.tree_len, &leaf);
    if (rc != HUFE_OK || leaf >= 0) {
        if (threadIdx.x == 0) j.status[SPEC_FAIL] = 1;
        return;
    }
    __syncthreads();                                      /* the table build used the stage area as scratch */
    if (blockIdx.x == 0) {
        /* the code length of every byte value = the depth of its leaf, level by level from the root:
         * with them decode_sub_kernel builds its tables in a few parallel steps (and checks them
         * against the tree, as it checks an encoder's) */
        constexpr int ENT = DecShared<THREADS>::ENT;
        uint16_t *depth = reinterpret_cast<uint16_t *>(sh.pay);
        for (int i = threadIdx.x; i < ENT; i += THREADS) depth[i] = (i == 0) ? 0 : 0xffffu;
        for (int i = threadIdx.x; i < HUF_NSYM; i += THREADS) lens_out[i] = 0;
        __syncthreads();
        for (uint32_t dlev = 0; dlev < (uint32_t)ENT; dlev++) {
            bool any = false;
            for (int i = threadIdx.x; i < j.tree_len; i += THREADS) {
                if (depth[i] != dlev) continue;
                const uint32_t lr = sh.lr[i];
                if (lr == DEC_LEAF_LR) {
                    if (dlev <= 255u) lens_out[(uint8_t)sh.ent[i]] = (uint8_t)dlev;
                    continue;
                }
                const uint32_t l = lr & 0xffffu, r = lr >> 16;
                if (l != DEC_NULL) { depth[l] = (uint16_t)(dlev + 1u); any = true; }
                if (r != DEC_NULL) { depth[r] = (uint16_t)(dlev + 1u); any = true; }
            }
            if (!__syncthreads_or(any ? 1 : 0)) break;
        }
        __syncthreads();
    }
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    uint32_t c = 0;
    if (lane < j.nlanes) {
        const uint64_t lo = lane * SPEC_LANE_BITS;
        const uint64_t hi = dmin<uint64_t>(lo + SPEC_LANE_BITS, j.max_bits);
        uint64_t pos = lo > SPEC_OVERLAP ? lo - SPEC_OVERLAP : 0;
        ColStage st = spec_stage_of<THREADS>(sh, j);
        uint64_t entry = ~0ull;
        while (pos < hi) {
            st.stage(pos >> 5);
            while (pos < hi && st.covers(pos)) {
                if (pos >= lo && entry == ~0ull) entry = pos;
                const bool counted = pos >= lo;
                if (spec_step<THREADS>(sh, st, pos) && counted) c++;
            }
        }
        if (entry == ~0ull) entry = pos;                  /* the share holds no codeword start (or is empty) */
        j.entry[lane] = entry;
        j.exitp[lane] = pos;
        j.cnt[lane] = c;
    }
    /* symbols before each share inside this workgroup's shares; the workgroup's count for the
     * one-workgroup scan over workgroups (a scan over all shares in one workgroup took 0.8 ms per
     * GiB of payload) */
    __syncthreads();                                      /* every wave is done with its column slice */
    uint32_t total;
    const uint32_t before = block_excl_scan_u32<THREADS>(c, sh.part, total);
    if (lane < j.nlanes) j.pre[lane] = before;
    if (threadIdx.x == 0) j.wg_pre[blockIdx.x] = total;
}

/* A share whose run-in did not reach a codeword boundary - it lies in a stretch where the payload
 * looks the same one bit on (a long run of one short code: zeros in a sparse file) - begins where its
 * left neighbour ended instead: one round mends every share whose neighbour was right, so a stretch
 * of n such shares takes n rounds (each a launch that touches only the broken shares).  The host
 * repeats it until nothing changes; then spec_sum_kernel redoes the sums. */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_repair_kernel(SpecJob j)
{
    __shared__ DecShared<THREADS> sh;
    __shared__ int s_any;
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    uint64_t from = 0;
    bool need = false;
    if (lane > 0 && lane < j.nlanes) {
        from = j.exitp[lane - 1];
        need = j.entry[lane] != from;
    }
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if (need) s_any = 1;
    __syncthreads();
    if (!s_any) return;
    int leaf;
    const int rc = dec_build_tables<THREADS, false>(sh, j.tree, j.tree_len, &leaf);
    if (rc != HUFE_OK || leaf >= 0) return;
    __syncthreads();
    if (!need) return;
    const uint64_t hi = dmin<uint64_t>((lane + 1) * SPEC_LANE_BITS, j.max_bits);
    ColStage st = spec_stage_of<THREADS>(sh, j);
    uint64_t pos = from;
    uint32_t c = 0;
    while (pos < hi) {
        st.stage(pos >> 5);
        while (pos < hi && st.covers(pos))
            if (spec_step<THREADS>(sh, st, pos)) c++;
    }
    j.entry[lane] = from;
    j.exitp[lane] = pos;
    j.cnt[lane] = c;
    atomicAdd(&j.status[SPEC_REPAIRED], 1ull);
}

/* the sums of spec_scan_kernel's tail again, after a repair */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_sum_kernel(SpecJob j)
{
    __shared__ uint32_t s_part[THREADS / 64];
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const uint32_t c = lane < j.nlanes ? j.cnt[lane] : 0u;
    uint32_t total;
    const uint32_t before = block_excl_scan_u32<THREADS>(c, s_part, total);
    if (lane < j.nlanes) j.pre[lane] = before;
    if (threadIdx.x == 0) j.wg_pre[blockIdx.x] = total;
}

/* symbols before every workgroup's shares (in place: counts in, exclusive sums out, the total last) */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_prefix_kernel(SpecJob j, uint64_t nwg, uint64_t *__restrict__ scratch)
{
    const uint64_t *cnt = j.wg_pre;
    const uint64_t total = chunked_excl_scan<THREADS>(nwg, scratch, [cnt](uint64_t i) { return cnt[i]; });
    __syncthreads();
    for (uint64_t i = threadIdx.x; i < nwg; i += THREADS) j.wg_pre[i] = scratch[i];
    if (threadIdx.x == 0) j.wg_pre[nwg] = total;
}

/* pass 2: the bits of every group of 32 symbols, the first bit of every tile, the bit behind the block's last symbol */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_mark_kernel(SpecJob j, HufSubIndex sub)
{
    __shared__ DecShared<THREADS> sh;
    int leaf;
    const int rc = dec_build_tables<THREADS, false>(sh, j.tree, j.tree_len, &leaf);
    if (rc != HUFE_OK || leaf >= 0) return;               /* pass 1 said so already */
    __syncthreads();                                      /* the table build used the stage area as scratch */
    const uint64_t lane = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    if (lane >= j.nlanes) return;
    uint64_t s = j.wg_pre[blockIdx.x] + j.pre[lane];
    if (s >= j.block_len) return;                         /* the block ends before this share */
    uint64_t pos = j.entry[lane];
    /* the chain: this lane begins where the one before it ended (lane 0 begins at bit 0) */
    if (lane > 0 ? (j.exitp[lane - 1] != pos) : (pos != 0)) {
        j.status[SPEC_CHAIN] = 1;                         /* (spec_repair_kernel can mend this) */
        return;
    }
    if (lane + 1 == j.nlanes && s + j.cnt[lane] < j.block_len) {
        j.status[SPEC_SHORT] = 1;                         /* the block does not end inside the covered bits (or the counts
                                                             are those of a broken chain) */
        return;
    }
    const uint64_t stop = j.exitp[lane];
    ColStage st = spec_stage_of<THREADS>(sh, j);
    /* A group's bits = the distance of two group starts: both inside this share -> written here;
     * the group that begins in an earlier share ends at this share's FIRST start -> first_pos /
     * first_g / last_pos, put together by spec_groups_kernel (a start every 32nd symbol written
     * down as 8 bytes for a later pass was a third of this kernel's time). */
    uint64_t prev = ~0ull, fpos = ~0ull, fg = ~0ull;
    uint64_t gnext = (s + HUF_SUB_GROUP - 1) / HUF_SUB_GROUP;                 /* the next group whose start this share sees */
    uint32_t k = 0;                                                           /* symbols taken in this share */
    uint32_t next_mark = (uint32_t)(gnext * HUF_SUB_GROUP - s);               /* ... k at that start */
    const uint64_t left = j.block_len - s;
    const uint32_t rem = left > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)left; /* the block ends after this many (a share holds < 2^13) */
    bool done = false;
    while (pos < stop && !done) {
        st.stage(pos >> 5);
        while (pos < stop && st.covers(pos)) {
            if (k == next_mark) {                                             /* (remembering the starts in registers and working
                                                                                 them off once per staged window was not faster) */
                const uint64_t g = gnext++;
                next_mark += HUF_SUB_GROUP;
                if (prev != ~0ull) {
                    const uint64_t d = pos - prev;
                    sub.group_bits[g - 1] = (uint16_t)(d > 0xffffull ? 0xffffull : d);
                } else {
                    fpos = pos;
                    fg = g;
                }
                prev = pos;
                if ((g & (HUF_SUB_TILE / HUF_SUB_GROUP - 1)) == 0) sub.tile_bits[g / (HUF_SUB_TILE / HUF_SUB_GROUP)] = pos;
            }
            if (spec_step<THREADS>(sh, st, pos)) {
                k++;
                if (k == rem) { done = true; break; }
            }
        }
    }
    if (done) {
        if (prev != ~0ull) {
            const uint64_t d = pos - prev;
            sub.group_bits[(j.block_len - 1) / HUF_SUB_GROUP] = (uint16_t)(d > 0xffffull ? 0xffffull : d);
        }
        j.status[SPEC_END_OPEN] = (prev == ~0ull) ? 1 : 0;                    /* the last group began in an earlier share */
        j.status[SPEC_END_LANE] = lane;
        j.status[SPEC_END_BITS] = pos;
        j.status[SPEC_FOUND] = 1;
    }
    j.first_pos[lane] = fpos;
    j.first_g[lane] = fg;
    j.last_pos[lane] = prev;
}

/* the groups that span shares, the padding of the sub-index row, and the two-entry block index
 * [pos, pos + encoded size) */
__global__ __launch_bounds__(256) void spec_groups_kernel(SpecJob j, HufSubIndex sub, uint64_t pos, uint64_t pay_off,
                                                          uint64_t *__restrict__ offs)
{
    const uint64_t ngroups = (j.block_len + HUF_SUB_GROUP - 1) / HUF_SUB_GROUP;
    const uint64_t end_bits = j.status[SPEC_END_BITS], end_lane = j.status[SPEC_END_LANE];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j.status[SPEC_FAIL] || j.status[SPEC_CHAIN] || j.status[SPEC_SHORT] || !j.status[SPEC_FOUND]) return;
    /* the last group start in front of share i */
    auto start_before = [&](uint64_t i) {
        while (i > 0) {
            const uint64_t p = j.last_pos[--i];
            if (p != ~0ull) return p;
        }
        return (uint64_t)0;
    };
    if (t <= end_lane) {
        const uint64_t g = j.first_g[t];
        if (g != ~0ull && g > 0 && t > 0) {
            const uint64_t d = j.first_pos[t] - start_before(t);
            sub.group_bits[g - 1] = (uint16_t)(d > 0xffffull ? 0xffffull : d);
        }
    }
    if (t == 0 && j.status[SPEC_END_OPEN]) {
        const uint64_t d = end_bits - start_before(end_lane);
        sub.group_bits[ngroups - 1] = (uint16_t)(d > 0xffffull ? 0xffffull : d);
    }
    if (t < sub.gpb - ngroups) sub.group_bits[ngroups + t] = 0;
    if (blockIdx.x == 0) {                                /* (the code lengths were written by spec_scan_kernel) */
        if (threadIdx.x == 0) {
            offs[0] = pos;
            offs[1] = pay_off + ((end_bits + 7) >> 3);
        }
    }
}

}  // namespace hufgpu
