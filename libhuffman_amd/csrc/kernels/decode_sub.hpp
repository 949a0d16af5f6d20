/* decode_sub.hpp - decode_sub_kernel / decode_fix_kernel: indexed decode with the encoder's sub-index
   (src/decoder.c:34-96 restated as one table pass per symbol).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "decode.hpp"
#include "pack.hpp"

namespace hufgpu {

/* ======================================================================================
 * Decode with the sub-index pack_kernel writes (pack.hpp, HufSubIndex): the payload bits of every
 * group of 32 symbols and the first payload bit of every tile of 8 192 symbols.
 *
 * The self-synchronising decoder (decode.hpp) has to find out where codewords start: every symbol
 * is decoded at least twice (count pass + write pass) plus the synchronisation rounds.  Here a
 * lane is TOLD where its 32 symbols start, decodes them once through the same 2^12-entry table
 * and stores them as two 16-byte words.  What it is told is verified, not trusted:
 *   (a) the block's first tile starts at payload bit 0,
 *   (b) a lane's 32 codewords take exactly the bits its group is said to have, none of its walks
 *       leaves the tree and none needs bits past the payload,
 *   (c) a chunk (the part of a block one workgroup decodes) ends where the next chunk is said to
 *       start.
 * By induction over the lanes these make the output equal to that of the in-order decoder of
 * src/decoder.c:34-96.  A block that fails any of them - a stale or foreign sub-index, a damaged
 * stream - is appended to a list and decoded again by decode_fix_kernel with the exact
 * self-synchronising decoder, which also produces the reference's error code: a wrong sub-index
 * costs time, never correctness.
 * ==================================================================================== */
#define DSUB_SPL 32                         /* symbols per lane = HUF_SUB_GROUP */
#define DSUB_SLACK_WORDS 16                 /* staged behind the last needed word: 31 table codewords + two refills of a lane that runs wild */
#define DSUB_MAX_GROUP_BITS (DSUB_SPL * HUF_CODE_MAXBITS)

/* 64-bit left-aligned bit buffer over the linearly staged payload words (big-endian words) */
struct LinReader {
    const uint32_t *st;
    uint32_t hi, lo;
    int32_t avail;
    uint32_t gf;         /* next staged word to append */

    __device__ __forceinline__ void load(uint32_t pos)
    {
        const uint32_t g = pos >> 5, off = pos & 31u;
        const uint64_t b = (((uint64_t)st[g] << 32) | st[g + 1]) << off;
        hi = (uint32_t)(b >> 32);
        lo = (uint32_t)b;
        avail = (int32_t)(64u - off);
        gf = g + 2;
    }
    __device__ __forceinline__ uint32_t index() const { return hi >> (32 - DEC_LUT_BITS); }
    __device__ __forceinline__ uint32_t pos() const { return (gf << 5) - (uint32_t)avail; }
    __device__ __forceinline__ void consume(uint32_t adv)
    {
        const uint64_t b = (((uint64_t)hi << 32) | lo) << adv;
        hi = (uint32_t)(b >> 32);
        lo = (uint32_t)b;
        avail -= (int32_t)adv;
    }
    __device__ __forceinline__ void refill()                   /* needs avail <= 32 */
    {
        const uint64_t t = (uint64_t)st[gf] << (32 - avail);
        hi |= (uint32_t)(t >> 32);
        lo |= (uint32_t)t;
        avail += 32;
        gf++;
    }
};

/* bit-serial walk behind a `long` table entry on the linear stage; result as dec_rare_packed */
template <int THREADS>
__device__ __forceinline__ uint64_t dec_rare_lin(const DecShared<THREADS> &sh, const uint32_t *st, uint32_t e,
                                                 uint32_t pos, uint32_t lim)
{
    uint32_t node = e & 0x7ffu;
    uint32_t p = pos + DEC_LUT_BITS;
    for (;;) {
        if (p >= lim) return (uint64_t)CW_EXH << 40;
        const uint32_t bit = (st[p >> 5] >> (31u - (p & 31u))) & 1u;
        p++;
        const uint32_t nx = bit ? sh.right[node] : sh.left[node];
        if (nx == DEC_NULL) return ((uint64_t)CW_BAD << 40) | p;
        node = nx;
        if (sh.left[node] == DEC_NULL && sh.right[node] == DEC_NULL) break;
    }
    return ((uint64_t)CW_OK << 40) | ((uint64_t)(uint8_t)sh.ent[node] << 32) | p;
}

/* One table step of a lane: returns the entry (low byte = symbol); *ok is cleared when the lookup is
 * not a codeword.  The rare paths sit behind one wave-uniform branch. */
template <int THREADS>
__device__ __forceinline__ uint32_t dsub_next(const DecShared<THREADS> &sh, LinReader &rd, uint32_t lim, bool &ok)
{
    uint32_t e = sh.lut[rd.index()];
    if (__builtin_expect(__ballot(e >= DEC_E_BAD) != 0ull, 0)) {
        if (e >= DEC_E_LONG) {
            const uint64_t r = dec_rare_lin<THREADS>(sh, rd.st, e, rd.pos(), lim);
            if ((int)(r >> 40) == CW_OK) {
                rd.load((uint32_t)r);
                e = (uint32_t)(r >> 32) & 0xffu;           /* advance 0: the reader already stands behind it */
            } else {
                ok = false;
                e = 0x0100u;
            }
        } else if (e >= DEC_E_BAD) {
            ok = false;
            e = 0x0100u;                                   /* keep moving: the lane's result is discarded anyway */
        }
    }
    rd.consume(e >> 8);
    return e;
}

/* The symbols [sym0, sym1) of a block (sym0 a multiple of THREADS * 32) with the block's sub-index.
 * Tables are in sh.  Returns true (workgroup-uniform) when everything was verified; *end_bit = the
 * payload bit behind the chunk's last symbol. */
template <int THREADS>
__device__ bool decode_payload_sub(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes,
                                   uint64_t block_len, uint64_t sym0, uint64_t sym1,
                                   const uint64_t *__restrict__ tile_bits,
                                   const uint16_t *__restrict__ group_bits, uint8_t *gout, uint64_t *end_bit)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int TILE = THREADS * DSUB_SPL;
    /* the staged words take the place of the self-synchronising decoder's payload image and marks */
    uint32_t *stage = sh.pay;
    constexpr uint32_t STAGE_WORDS = (uint32_t)((sizeof(sh.pay) + sizeof(sh.mark)) / sizeof(uint32_t));
    static_assert(offsetof(DecShared<THREADS>, mark) == offsetof(DecShared<THREADS>, pay) + sizeof(sh.pay), "stage = pay + mark");
    constexpr uint32_t CAP_BITS = (STAGE_WORDS - DSUB_SLACK_WORDS - 2) * 32u;
    static_assert(CAP_BITS >= 64u * DSUB_MAX_GROUP_BITS + 32u, "one wave's groups always fit the stage");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t pay_bits = pay_bytes * 8ull;
    uint32_t *s_wtot = sh.part;                    /* [WAVES] */
    bool ok = true;
    (void)block_len;

    uint64_t T0 = tile_bits[sym0 / HUF_SUB_TILE];  /* first payload bit of the chunk, as told */
    if (sym0 == 0 && T0 != 0) ok = false;          /* (a) */

    for (uint64_t t0 = sym0; t0 < sym1; t0 += TILE) {
        const uint64_t my0 = t0 + (uint64_t)tid * DSUB_SPL;
        uint32_t nsym = 0, gb = 0;
        if (my0 < sym1) {
            nsym = (uint32_t)dmin<uint64_t>(DSUB_SPL, sym1 - my0);
            gb = group_bits[my0 / DSUB_SPL];
            if (gb > DSUB_MAX_GROUP_BITS) { gb = DSUB_MAX_GROUP_BITS; ok = false; }
        }
        /* exclusive sums of the groups' bits: in the wave by shuffles, across waves through LDS */
        uint32_t inc = gb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wtot[wave] = inc;
        __syncthreads();
        uint32_t wtot[WAVES], tile_total = 0, mybase = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            wtot[w] = uni32(s_wtot[w]);
            if (w == wave) mybase = tile_total;
            tile_total += wtot[w];
        }
        const uint32_t ex = mybase + inc - gb;     /* my first bit relative to T0 */
        if (T0 + tile_total > pay_bits) ok = false;                /* (b): bits past the payload */

        /* One pass stages the bits of all waves; only when they do not fit the stage (codes far
         * longer than the 9-bit average) every wave gets a pass of its own. */
        const bool single = (uint32_t)(T0 & 31ull) + tile_total <= CAP_BITS;
        const int npass = single ? 1 : WAVES;
#pragma unroll 1
        for (int ps = 0; ps < npass; ps++) {
            uint32_t pbase = 0, pbits = tile_total;                /* the pass's first bit (relative to T0) and bit count */
            if (!single) {
                pbits = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    if (w < ps) pbase += wtot[w];
                    if (w == ps) pbits = wtot[w];
                }
            }
            const bool mine = single || wave == ps;
            const uint64_t first = T0 + pbase;
            const uint64_t origin = first & ~31ull;                /* payload bit of stage word 0 */
            const uint32_t lead = (uint32_t)(first - origin);
            const uint32_t need_bits = lead + pbits;
            const uint32_t nwords = ((need_bits + 31u) >> 5) + DSUB_SLACK_WORDS + 2u;   /* <= STAGE_WORDS */
            const uint32_t lim = ((need_bits + 31u) >> 5) * 32u + 64u;                  /* bits a walk may look at */
            /* ---- stage: big-endian words of payload bytes [origin / 8 + 4 i, + 4) ---- */
            {
                const uint64_t byte0 = origin >> 3;
                int tl = tid;
                asm volatile("" : "+v"(tl));
                if (byte0 + 4ull * nwords + 8ull <= pay_bytes) {
                    const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)(pay + byte0));
                    const uint32_t m = (uint32_t)(a & 3u);
                    const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(a - m);
                    for (uint32_t i = (uint32_t)tl; i < nwords; i += THREADS)
                        stage[i] = __builtin_amdgcn_perm(q[i + 1], q[i], sel);
                } else {
                    for (uint32_t i = (uint32_t)tl; i < nwords; i += THREADS)
                        stage[i] = load_be32(pay, byte0 + 4ull * i, pay_bytes);
                }
            }
            __syncthreads();
            if (mine && nsym) {
                const uint32_t s = lead + (ex - pbase);
                LinReader rd;
                rd.st = stage;
                rd.load(s);
                uint8_t *dst = gout + my0;
                if (nsym == DSUB_SPL) {
                    uint32_t w[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        uint32_t acc = 0;
                        acc = __builtin_amdgcn_alignbit(dsub_next<THREADS>(sh, rd, lim, ok), acc, 8);
                        acc = __builtin_amdgcn_alignbit(dsub_next<THREADS>(sh, rd, lim, ok), acc, 8);
                        if (rd.avail <= 32) rd.refill();
                        acc = __builtin_amdgcn_alignbit(dsub_next<THREADS>(sh, rd, lim, ok), acc, 8);
                        acc = __builtin_amdgcn_alignbit(dsub_next<THREADS>(sh, rd, lim, ok), acc, 8);
                        if (rd.avail <= 32) rd.refill();
                        w[k] = acc;
                    }
                    if ((((uintptr_t)dst) & 15u) == 0) {
                        store_stream16(reinterpret_cast<uint4 *>(dst), make_uint4(w[0], w[1], w[2], w[3]));
                        store_stream16(reinterpret_cast<uint4 *>(dst) + 1, make_uint4(w[4], w[5], w[6], w[7]));
                    } else {
#pragma unroll
                        for (int k = 0; k < 32; k++) dst[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                    }
                } else {                                            /* the block's last, short group */
                    for (uint32_t k = 0; k < nsym; k++) {
                        dst[k] = (uint8_t)dsub_next<THREADS>(sh, rd, lim, ok);
                        if (rd.avail <= 32) rd.refill();
                    }
                }
                if (rd.pos() - s != gb) ok = false;                 /* (b): exactly the bits of the group */
            }
            __syncthreads();                                        /* the stage is rewritten by the next pass */
        }
        T0 += tile_total;
    }
    *end_bit = T0;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

/* Work list of blocks the sub-index path could not verify (zeroed state between launches: the
 * flags by decode_fix_kernel, the count by decode_prepare_kernel of the next call). */
struct DecFixList {
    uint32_t *count;      /* [1] */
    uint32_t *blocks;     /* [nblocks] */
    uint32_t *flag;       /* [nblocks] block already listed */
};

#define DSUB_CHUNK_SYMS 65536u        /* symbols one workgroup decodes: four tiles of 512 x 32 */

template <int THREADS>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void decode_sub_kernel(
    const uint8_t *__restrict__ stream, uint64_t stream_len, const uint64_t *__restrict__ offsets,
    const HufDecodeMeta *__restrict__ dmeta, uint64_t *__restrict__ out_offsets, TwoLevel lens,
    uint8_t *__restrict__ out, uint64_t out_cap, int32_t *__restrict__ status,
    unsigned long long *__restrict__ result, HufSubIndex sub, uint64_t blocksize, uint32_t cpb, DecFixList fix)
{
    __shared__ DecShared<THREADS> sh;
    static_assert(DSUB_CHUNK_SYMS % (THREADS * DSUB_SPL) == 0 && (THREADS * DSUB_SPL) % HUF_SUB_TILE == 0, "chunks are whole tiles");
    const int tid = (int)threadIdx.x;
    const uint64_t blk = blockIdx.x / cpb;
    const uint32_t c = (uint32_t)(blockIdx.x % cpb);
    const HufDecodeMeta m = dmeta[blk];
    const uint64_t obase = lens.gprefix[blk / SCAN_GROUP] + lens.local[blk];
    if (tid == 0 && c == 0) out_offsets[blk] = obase;
    if (m.status != HUFE_OK || m.block_len == 0) return;            /* header errors were recorded by decode_prepare */
    const uint64_t sym0 = (uint64_t)c * DSUB_CHUNK_SYMS;
    if (sym0 >= m.block_len) return;
    if (obase + m.block_len > out_cap) {
        if (tid == 0 && c == 0) {
            status[blk] = HUFE_MEMORY;
            atomicMin(&result[2], (unsigned long long)blk);
        }
        return;
    }
    if (m.block_len > blocksize) {
        /* more symbols than the encode this sub-index belongs to put into a block (a damaged header,
         * or not that encode's stream at all): the chunks do not cover it */
        if (tid == 0 && c == 0 && atomicExch(&fix.flag[blk], 1u) == 0u) fix.blocks[atomicAdd(fix.count, 1u)] = (uint32_t)blk;
        return;
    }
    const uint64_t sym1 = dmin<uint64_t>(m.block_len, sym0 + DSUB_CHUNK_SYMS);
    const uint64_t o0 = offsets[blk];
    const uint64_t o1 = dmin<uint64_t>(offsets[blk + 1], stream_len);
    const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
    const uint8_t *tree = stream + o0 + HUF_HEADER_FIXED;
    const uint8_t *pay = tree + 2 * (int)m.tree_len;
    bool good;
    int leaf = m.leaf;
    int rc = HUFE_OK;
    if (leaf < 0) rc = dec_build_tables<THREADS, false>(sh, tree, m.tree_len, &leaf);
    if (rc != HUFE_OK) {
        good = false;
    } else if (leaf >= 0) {
        /* one 0 bit per symbol: the chunk's bits start at payload bit sym0 (a multiple of 8) */
        uint64_t eb = 0, produced = 0;
        good = (sym0 >> 3) <= pay_bytes &&
               decode_single_leaf<THREADS, true>(sh, (uint32_t)leaf, pay + (sym0 >> 3), sym1 - sym0,
                                                 pay_bytes - (sym0 >> 3), out + obase + sym0, &eb, &produced) == HUFE_OK;
    } else {
        uint64_t end_bit = 0;
        good = decode_payload_sub<THREADS>(sh, pay, pay_bytes, m.block_len, sym0, sym1, sub.tile_bits + blk * sub.tpb,
                                           sub.group_bits + blk * sub.gpb, out + obase, &end_bit);
        /* (c) the next chunk starts where this one ends */
        if (good && sym1 < m.block_len && sub.tile_bits[blk * sub.tpb + sym1 / HUF_SUB_TILE] != end_bit) good = false;
    }
    if (!good && tid == 0) {
        if (atomicExch(&fix.flag[blk], 1u) == 0u) fix.blocks[atomicAdd(fix.count, 1u)] = (uint32_t)blk;
    }
}

/* The listed blocks again, whole, with the exact self-synchronising decoder (= decode_kernel's body). */
template <int THREADS>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void decode_fix_kernel(
    const uint8_t *__restrict__ stream, uint64_t stream_len, const uint64_t *__restrict__ offsets,
    const HufDecodeMeta *__restrict__ dmeta, TwoLevel lens, uint8_t *__restrict__ out, uint64_t out_cap,
    int32_t *__restrict__ status, unsigned long long *__restrict__ result, DecFixList fix)
{
    __shared__ DecShared<THREADS> sh;
    const int tid = (int)threadIdx.x;
    const uint32_t n = uni32(*fix.count);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint64_t blk = fix.blocks[i];
        if (tid == 0) fix.flag[blk] = 0;
        const HufDecodeMeta m = dmeta[blk];
        const uint64_t obase = lens.gprefix[blk / SCAN_GROUP] + lens.local[blk];
        const uint64_t o0 = offsets[blk];
        const uint64_t o1 = dmin<uint64_t>(offsets[blk + 1], stream_len);
        const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
        uint64_t end_bits = 0, produced = 0;
        __syncthreads();
        const int err = decode_block<THREADS>(sh, stream + o0 + HUF_HEADER_FIXED, m.tree_len, m.block_len, pay_bytes,
                                              out + obase, &end_bits, &produced);
        if (tid == 0 && err != HUFE_OK) {
            status[blk] = err;
            atomicMin(&result[2], (unsigned long long)blk);
        }
    }
}

}  // namespace hufgpu
