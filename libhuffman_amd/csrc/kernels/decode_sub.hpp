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
#define DSUB_L2_BITS 6u                     /* a second-level table takes codes of up to 12 + 6 bits */
#define DSUB_L2_ENTRIES 1024u               /* it lives in DecShared::ent */
#define DSUB_SLACK_WORDS 16                 /* staged behind the last needed word: 31 table codewords + two refills of a lane that runs wild */
#define DSUB_MAX_GROUP_BITS (DSUB_SPL * HUF_CODE_MAXBITS)
#define DSUB_CHUNK_SYMS 65536u               /* symbols one workgroup decodes: four tiles of 512 x 32 */

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
        const uint32_t nx = dec_child(sh.lr[node], bit);
        if (nx == DEC_NULL) return ((uint64_t)CW_BAD << 40) | p;
        node = nx;
        if (sh.lr[node] == DEC_LEAF_LR) break;
    }
    return ((uint64_t)CW_OK << 40) | ((uint64_t)(uint8_t)sh.ent[node] << 32) | p;
}

/* ======================================================================================
 * Tables from the stream's tree AND the encoder's code lengths, in a few parallel steps.
 *
 * dec_build_tables (decode.hpp) finds every node's right child by a search and walks the tree once
 * per table entry: 0.29 of the kernel's 0.78 ms per GiB.  With the code length of every byte value
 * from the sub-index the same tables follow from prefix sums - and the lengths can be CHECKED
 * against the serialized tree in parallel, which nothing short of walking it can do without them:
 *   - the tree has the shape every encoder-made tree has: 4K+1 entries, K leaves (a node followed
 *     by two -1), 2K nodes, 2K+1 markers, a root with a left child only;
 *   - with d_k = claimed length of the k-th leaf in preorder, the code of leaf k is the sum of
 *     2^-d_j over j < k (preorder visits leaves in code order) and the lengths sum to 1/2 exactly;
 *   - leaf 0 is entry d_0 (its d_0 ancestors precede it), the last leaf is followed only by its
 *     markers and the root's, and between leaf k and leaf k+1 lie their 3 + d_(k+1) - (d_k - t_k)
 *     entries, t_k = trailing one bits of leaf k's code (up past t_k right children, then down the
 *     left spine of the next subtree).
 * Leaf 0's depth is fixed by the stream, and each further depth by the entry positions and the
 * depths before it: if every check holds the claimed lengths ARE the tree's.  Anything else -
 * another shape of tree, a stale sub-index - returns false and the caller walks the tree.
 * Codes longer than 12 bits get a `long` entry and are decoded by a binary search over the leaves'
 * codes (dec_rare_fast); the child links are not built.  Lengths above 32 take the walk.
 * LDS: code[256] (left-aligned in 32 bits), length[256], byte[256] of the leaves in preorder in
 * sh.lr; sh.fastk = K.
 * ==================================================================================== */
template <int THREADS>
struct DsubFastLds {
    __device__ static __forceinline__ uint32_t *code(DecShared<THREADS> &sh) { return sh.lr; }
    __device__ static __forceinline__ uint8_t *len(DecShared<THREADS> &sh) { return reinterpret_cast<uint8_t *>(sh.lr + 256); }
    __device__ static __forceinline__ uint8_t *sym(DecShared<THREADS> &sh) { return reinterpret_cast<uint8_t *>(sh.lr + 320); }
    __device__ static __forceinline__ const uint32_t *code(const DecShared<THREADS> &sh) { return sh.lr; }
    __device__ static __forceinline__ const uint8_t *len(const DecShared<THREADS> &sh) { return reinterpret_cast<const uint8_t *>(sh.lr + 256); }
    __device__ static __forceinline__ const uint8_t *sym(const DecShared<THREADS> &sh) { return reinterpret_cast<const uint8_t *>(sh.lr + 320); }
};

/* largest k < K with code[k] <= v (code[0] = 0) */
__device__ __forceinline__ uint32_t dsub_leaf_of(const uint32_t *code, uint32_t K, uint32_t v)
{
    uint32_t lo = 0, hi = K;
#pragma unroll
    for (int it = 0; it < 8; it++) {                     /* K <= 256 */
        const uint32_t mid = (lo + hi) >> 1;
        if (hi - lo > 1u) {
            if (code[mid] <= v) lo = mid; else hi = mid;
        }
    }
    return lo;
}

template <int THREADS>
__device__ bool dsub_fast_tables(DecShared<THREADS> &sh, const uint8_t *tree, int tree_len, const uint8_t *__restrict__ lens_g)
{
    typedef DsubFastLds<THREADS> F;
    constexpr int ENT = DecShared<THREADS>::ENT;
    static_assert(THREADS * 2 >= ENT - 2 && THREADS >= 256, "two entries per thread");
    const int tid = (int)threadIdx.x;
    const uint32_t K = (uint32_t)(tree_len - 1) >> 2;
    if (tid == 0) { sh.fastk = 0; sh.l2n = 0; }
    if (tree_len < 9 || tree_len > HUF_TREE_MAX || ((tree_len - 1) & 3) != 0) return false;     /* uniform */
    uint8_t *s_lens = reinterpret_cast<uint8_t *>(sh.pay);                      /* [256] the claimed lengths by byte value */
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(sh.pay + 64);                /* [256] entry index of the k-th leaf */
    __syncthreads();                                                            /* previous user of sh is done */
    {
        /* entries 2t and 2t+1 from two aligned 32-bit loads per thread, as in dec_build_tables */
        const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
        const uint32_t mis = (uint32_t)(a & 3u);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(a - mis);
        const uint32_t nbytes = mis + 2u * (uint32_t)tree_len;
        for (int t = tid; 2 * t < ENT; t += THREADS) {
            const uint32_t lo = (4u * (uint32_t)t < nbytes) ? q[t] : 0u;
            const uint32_t hi = (4u * (uint32_t)t + 4u < nbytes) ? q[t + 1] : 0u;
            const uint32_t two = mis ? __builtin_amdgcn_alignbit(hi, lo, 8u * mis) : lo;
            const int i = 2 * t;
            sh.ent[i] = (i < tree_len) ? (int16_t)(two & 0xffffu) : (int16_t)-1;
            if (i + 1 < ENT) sh.ent[i + 1] = (i + 1 < tree_len) ? (int16_t)(two >> 16) : (int16_t)-1;
        }
        if (tid < 64) reinterpret_cast<uint32_t *>(s_lens)[tid] = reinterpret_cast<const uint32_t *>(lens_g)[tid];
    }
    __syncthreads();
    bool ok = true;
    /* ---- shape: leaves, nodes, markers ---- */
    {
        const int i0 = 2 * tid;
        const int e0 = sh.ent[i0], e1 = sh.ent[i0 + 1], e2 = (i0 + 2 < ENT) ? sh.ent[i0 + 2] : -1, e3 = (i0 + 3 < ENT) ? sh.ent[i0 + 3] : -1;
        const bool n0 = e0 != -1, n1 = e1 != -1;                                /* (entries at or past tree_len read -1) */
        const bool l0 = n0 && e1 == -1 && e2 == -1 && i0 + 2 < tree_len;
        const bool l1 = n1 && e2 == -1 && e3 == -1 && i0 + 3 < tree_len;
        uint32_t tot;
        const uint32_t mine = (uint32_t)l0 + (uint32_t)l1 + (((uint32_t)n0 + (uint32_t)n1) << 16);
        const uint32_t ex = block_excl_scan_u32<THREADS>(mine, sh.part, tot);
        if ((tot & 0xffffu) != K || (tot >> 16) != 2u * K) ok = false;
        if (tid == 0 && (sh.ent[0] == -1 || sh.ent[tree_len - 1] != -1)) ok = false;
        uint32_t k = ex & 0xffffu;
        if (l0 && k < 256u) { s_pos[k] = (uint16_t)i0; F::sym(sh)[k] = (uint8_t)e0; k++; }
        if (l1 && k < 256u) { s_pos[k] = (uint16_t)(i0 + 1); F::sym(sh)[k] = (uint8_t)e1; }
    }
    if (!__syncthreads_and(ok ? 1 : 0)) return false;
    /* ---- claimed lengths -> codes; they must fill the left half of the code space exactly ---- */
    uint32_t d = 0;
    {
        uint64_t width = 0;
        if ((uint32_t)tid < K) {
            d = s_lens[F::sym(sh)[tid]];
            if (d < 2u || d > 32u) ok = false;
            else width = 1ull << (32u - d);
        }
        uint64_t tot;
        const uint64_t ex = block_excl_scan<THREADS, uint64_t>(width, reinterpret_cast<uint64_t *>(sh.wtile), tot);
        if (tot != (1ull << 31)) ok = false;
        if ((uint32_t)tid < K) {
            F::code(sh)[tid] = (uint32_t)ex;
            F::len(sh)[tid] = (uint8_t)d;
        }
    }
    if (!__syncthreads_and(ok ? 1 : 0)) return false;
    /* ---- the entry positions the lengths imply are the stream's ---- */
    if ((uint32_t)tid < K) {
        const uint32_t k = (uint32_t)tid;
        const uint32_t bits = F::code(sh)[k] >> (32u - d);                      /* the d code bits */
        const uint32_t t = (uint32_t)__builtin_ctz(~bits);                      /* trailing ones (< d: codes start with 0) */
        const uint32_t pos = s_pos[k];
        if (k == 0 && pos != d) ok = false;
        if (k + 1 < K) {
            const uint32_t dn = F::len(sh)[k + 1];
            if (dn + t < d || (uint32_t)s_pos[k + 1] != pos + 3u + (dn + t - d)) ok = false;
        } else if (pos + 4u != (uint32_t)tree_len) ok = false;                  /* leaf, its two markers, the root's */
    }
    if (!__syncthreads_and(ok ? 1 : 0)) return false;
    /* ---- the table: eight consecutive entries per thread, one 16-byte store ---- */
    {
        static_assert((1 << DEC_LUT_BITS) == THREADS * 8, "eight entries per thread");
        const uint32_t *code = F::code(sh);
        const uint32_t x0 = (uint32_t)tid * 8u;
        uint32_t k = dsub_leaf_of(code, K, x0 << (32 - DEC_LUT_BITS));
        uint32_t e[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t idx = x0 + j;
            const uint32_t v = idx << (32 - DEC_LUT_BITS);
            if (v >> 31) {
                /* the first bit leaves the tree (the root has no right child); the run of bits that fail
                 * the same way, as in dec_build_tables */
                const uint32_t skip = dmin<uint32_t>((uint32_t)__clz((int)~v), (uint32_t)DEC_LUT_BITS);
                e[j] = DEC_E_BAD | DEC_E_NOCW | (skip << 8) | 1u;
            } else {
                while (k + 1u < K && code[k + 1u] <= v) k++;
                const uint32_t dk = F::len(sh)[k];
                e[j] = (dk <= (uint32_t)DEC_LUT_BITS) ? ((dk << 8) | (uint32_t)F::sym(sh)[k]) : (uint32_t)DEC_E_LONG;
            }
        }
        *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
    /* ---- second level: the subtree below a 12-bit prefix whose codes are at most DSUB_L2_BITS longer
     *      gets a table of its own in the LDS the tree entries occupied (they are not needed any more):
     *      entry 0x8000 | (bits - 1) << 10 | offset in the first table, (length << 8) | byte in the second.
     *      Codes beyond that keep their `long` entry. ---- */
    __syncthreads();                                     /* the first table is written, the tree entries are done with */
    {
        uint16_t *l2 = reinterpret_cast<uint16_t *>(sh.ent);
        const uint32_t *code = F::code(sh);
        const uint8_t *len = F::len(sh);
        uint32_t size = 0, nb = 0, run_end = 0;
        const uint32_t k = (uint32_t)tid;
        const uint32_t P = (k < K) ? (code[k] >> (32 - DEC_LUT_BITS)) : 0u;
        if (k < K && d > (uint32_t)DEC_LUT_BITS &&
            (k == 0 || len[k - 1] <= DEC_LUT_BITS || (code[k - 1] >> (32 - DEC_LUT_BITS)) != P)) {
            /* first leaf below its prefix: the leaves below one prefix are neighbours (preorder = code order) */
            uint32_t maxd = d, jn = k + 1u;
            while (jn < K && jn - k <= (1u << DSUB_L2_BITS) && (code[jn] >> (32 - DEC_LUT_BITS)) == P) {
                maxd = dmax<uint32_t>(maxd, len[jn]);
                jn++;
            }
            const bool closed = !(jn < K && (code[jn] >> (32 - DEC_LUT_BITS)) == P);
            if (closed && maxd <= (uint32_t)DEC_LUT_BITS + DSUB_L2_BITS) {
                nb = maxd - DEC_LUT_BITS;
                size = 1u << nb;
                run_end = jn;
            }
        }
        uint32_t total;
        const uint32_t off = block_excl_scan_u32<THREADS>(size, sh.part, total);
        if (size && off + size <= DSUB_L2_ENTRIES) {
            for (uint32_t jn = k; jn < run_end; jn++) {
                const uint32_t dj = len[jn];
                const uint32_t first = (code[jn] >> (32 - DEC_LUT_BITS - nb)) & (size - 1u);
                const uint32_t count = 1u << (DEC_LUT_BITS + nb - dj);
                const uint16_t entry = (uint16_t)((dj << 8) | (uint32_t)F::sym(sh)[jn]);
                for (uint32_t i = 0; i < count; i++) l2[off + first + i] = entry;
            }
            sh.lut[P] = (uint16_t)(0x8000u | ((nb - 1u) << 10) | off);
        }
        if (tid == 0) sh.l2n = dmin<uint32_t>(total, DSUB_L2_ENTRIES);
    }
    if (tid == 0) sh.fastk = K;
    __syncthreads();
    return true;
}

/* a second-level entry of the first table resolved with the 32 bits at the codeword's start */
template <int THREADS>
__device__ __forceinline__ uint32_t dsub_l2(const DecShared<THREADS> &sh, uint32_t e, uint32_t bits32)
{
    const uint32_t nb = ((e >> 10) & 7u) + 1u;
    return reinterpret_cast<const uint16_t *>(sh.ent)[(e & 0x3ffu) + ((bits32 << DEC_LUT_BITS) >> (32u - nb))];
}
#define DSUB_IS_L2(e) (((e) & 0xC000u) == 0x8000u)

/* a codeword longer than the table's 12 bits with dsub_fast_tables' tables: the leaf whose code
 * interval holds the 32 bits at the position; result as dec_rare_packed */
template <int THREADS>
__device__ __forceinline__ uint64_t dec_rare_fast(const DecShared<THREADS> &sh, const uint32_t *st, uint32_t pos, uint32_t lim)
{
    typedef DsubFastLds<THREADS> F;
    const uint32_t g = pos >> 5, o = pos & 31u;
    const uint32_t w = o ? ((st[g] << o) | (st[g + 1] >> (32u - o))) : st[g];
    if (w >> 31) return ((uint64_t)CW_BAD << 40) | (pos + 1u);
    const uint32_t k = dsub_leaf_of(F::code(sh), sh.fastk, w);
    const uint32_t p = pos + (uint32_t)F::len(sh)[k];
    if (p > lim) return (uint64_t)CW_EXH << 40;
    return ((uint64_t)CW_OK << 40) | ((uint64_t)F::sym(sh)[k] << 32) | p;
}

/* One table step of a lane: returns the entry (low byte = symbol); *ok is cleared when the lookup is
 * not a codeword.  The rare paths sit behind one wave-uniform branch. */
template <int THREADS>
__device__ __forceinline__ uint32_t dsub_next(const DecShared<THREADS> &sh, LinReader &rd, uint32_t lim, bool &ok)
{
    uint32_t e = sh.lut[rd.index()];
    if (DSUB_IS_L2(e)) e = dsub_l2<THREADS>(sh, e, rd.hi);
    if (__builtin_expect(__ballot(e >= DEC_E_BAD) != 0ull, 0)) {
        if (e >= DEC_E_LONG) {
            const uint64_t r = sh.fastk ? dec_rare_fast<THREADS>(sh, rd.st, rd.pos(), lim) : dec_rare_lin<THREADS>(sh, rd.st, e, rd.pos(), lim);
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

/* LDS of the sub-index path inside DecShared: the area of the self-synchronising decoder's payload
 * image and marks holds one private slice of staged payload words per WAVE and, at its end, the
 * chunk's group bit counts (fetched by LDS-DMA while the tables are built; the table build's
 * scratch lies in front of them). */
template <int THREADS>
struct DsubLds {
    static constexpr int WAVES = THREADS / 64;
    static constexpr uint32_t AREA_WORDS = (uint32_t)((sizeof(DecShared<THREADS>::pay) + sizeof(DecShared<THREADS>::mark)) / sizeof(uint32_t));
    static constexpr uint32_t GROUPS = DSUB_CHUNK_SYMS / DSUB_SPL;          /* groups per chunk */
    static constexpr uint32_t WTILES = GROUPS / 64;                         /* wave tiles (64 groups) per chunk */
    static constexpr uint32_t GB_WORDS = GROUPS / 2;                        /* two 16-bit counts per word */
    static constexpr uint32_t SLICE_WORDS = ((AREA_WORDS - GB_WORDS) / WAVES) & ~3u;
    static constexpr uint32_t CAP_BITS = (SLICE_WORDS - DSUB_SLACK_WORDS - 2 - 4) * 32u;   /* (- 4: slices are written 16 bytes at a time) */
    static_assert(offsetof(DecShared<THREADS>, mark) == offsetof(DecShared<THREADS>, pay) + sizeof(DecShared<THREADS>::pay), "one area");
    static_assert(offsetof(DecShared<THREADS>, pay) % 16 == 0 && SLICE_WORDS % 4 == 0, "16-byte slice stores");
    static_assert(CAP_BITS >= DSUB_MAX_GROUP_BITS + 32u, "one lane's group always fits a slice");
    static_assert(SLICE_WORDS <= 3 * 256, "three staging steps of 64 lanes x 4 words cover a slice");
    static_assert((AREA_WORDS - GB_WORDS) * 4 >= ((DecShared<THREADS>::ENT + 1) / 2) * 4 + 2 * 2048 * 2, "the table build's scratch lies in front of the group counts");
    static_assert(WTILES <= 64 && WTILES % WAVES == 0 && WTILES <= sizeof(DecShared<THREADS>::wtile) / 4, "a wave scans the tile totals in one step");
    __device__ static __forceinline__ uint32_t *slice(DecShared<THREADS> &sh, int wave) { return sh.pay + (uint32_t)wave * SLICE_WORDS; }
    __device__ static __forceinline__ uint32_t *gb(DecShared<THREADS> &sh) { return sh.pay + (AREA_WORDS - GB_WORDS); }
};

/* The chunk's group bit counts are requested first (LDS-DMA: no register waits for them), so that
 * their latency passes under the table build.  grp = the chunk's first count (4-byte aligned: rows
 * of the sub-index are padded), ngrp = groups the chunk has. */
template <int THREADS>
__device__ __forceinline__ void dsub_prefetch(DecShared<THREADS> &sh, const uint16_t *__restrict__ grp, uint32_t ngrp)
{
    const uint32_t *g32 = reinterpret_cast<const uint32_t *>(grp);
    const uint32_t nd = (ngrp + 1u) >> 1;
    const uint32_t wave = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t k = 0; k < DsubLds<THREADS>::GB_WORDS / THREADS; k++) {
        const uint32_t d = k * THREADS + threadIdx.x;
        if (d < nd) __builtin_amdgcn_global_load_lds(g32 + d, DsubLds<THREADS>::gb(sh) + k * THREADS + wave * 64u, 4, 0, 0);
    }
}

/* A group again, step by step with the rare paths (long codes; the block's last, short group).
 * Out of line: inlined, its state competes with the hot loop's for the 64 VGPRs. */
template <int THREADS>
__device__ __noinline__ bool dsub_redo_group(const DecShared<THREADS> &sh, const uint32_t *stage, uint32_t s, uint32_t nsym,
                                             uint32_t lim, uint32_t gb, uint8_t *dst)
{
    bool ok = true;
    LinReader rd;
    rd.st = stage;
    rd.load(s);
    for (uint32_t k = 0; k < nsym; k++) {
        dst[k] = (uint8_t)dsub_next<THREADS>(sh, rd, lim, ok);
        if (rd.avail <= 32) rd.refill();
    }
    return ok && rd.pos() - s == gb;
}

/* The symbols [sym0, sym1) of a block (sym0 a multiple of DSUB_CHUNK_SYMS) with the block's sub-index.
 * Tables are in sh, dsub_prefetch has been called.  Returns true (workgroup-uniform) when everything
 * was verified; *end_bit = the payload bit behind the chunk's last symbol.  T0 = the chunk's first
 * payload bit, as told.
 *
 * After one scan of the group counts every WAVE is on its own: a wave tile = 64 groups = 2 048
 * symbols; wave w takes tiles w, w + 8, ...; it stages the tile's payload words in its private LDS
 * slice (LDS operations of one wave are in order: no barrier), decodes, stores.  The waves of a
 * workgroup drift apart, so one wave's memory latency passes under the other waves' decoding; the
 * version with workgroup-wide tiles spent 2/3 of its time in barriers behind loads. */
template <int THREADS>
__device__ bool decode_payload_sub(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes,
                                   uint64_t sym0, uint64_t sym1, uint64_t T0, uint8_t *gout, uint64_t *end_bit)
{
    typedef DsubLds<THREADS> L;
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CAP_BITS = L::CAP_BITS;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uint16_t *s_gb = reinterpret_cast<const uint16_t *>(L::gb(sh));
    uint32_t *stage = L::slice(sh, wave);
    const uint32_t ngrp = (uint32_t)((sym1 - sym0 + DSUB_SPL - 1) / DSUB_SPL);
    bool ok = true;
    unsigned long long pt = DPROF_T();

    /* the group counts requested by dsub_prefetch have landed (every wave waits for its own
     * requests, the barrier makes them everybody's) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sym0 == 0 && T0 != 0) ok = false;          /* (a) */

    /* bits of every wave tile, then their exclusive sums (lane q of every wave holds tile q's) */
#pragma unroll
    for (uint32_t k = 0; k < L::WTILES / WAVES; k++) {
        const uint32_t q = (uint32_t)wave * (L::WTILES / WAVES) + k;
        const uint32_t g = q * 64u + (uint32_t)lane;
        uint32_t v = (g < ngrp) ? (uint32_t)s_gb[g] : 0u;
        if (v > DSUB_MAX_GROUP_BITS) { v = DSUB_MAX_GROUP_BITS; ok = false; }
        v = wave_lane_u32(wave_incl_scan_u32(v), 63);
        if (lane == 0) sh.wtile[q] = v;
    }
    __syncthreads();
    const uint32_t tbits_q = ((uint32_t)lane < L::WTILES) ? sh.wtile[lane] : 0u;
    const uint32_t tincl = wave_incl_scan_u32(tbits_q);
    const uint32_t tstart_q = tincl - tbits_q;                       /* lane q: first bit of tile q relative to T0 */
    const uint32_t chunk_bits = wave_lane_u32(tincl, 63);
    if (T0 + chunk_bits > pay_bits) ok = false;                      /* (b): bits past the payload */
    const bool use_l2 = uni32(sh.l2n) != 0u;                          /* the block has codes in a second-level table */
    DPROF_ADD(8, pt);

#pragma unroll 1
    for (uint32_t q = (uint32_t)wave; q * 64u < ngrp; q += WAVES) {
        pt = DPROF_T();
        const uint32_t g = q * 64u + (uint32_t)lane;
        const uint64_t my0 = sym0 + (uint64_t)g * DSUB_SPL;
        uint32_t nsym = 0, gb = 0;
        if (g < ngrp) {
            nsym = (uint32_t)dmin<uint64_t>(DSUB_SPL, sym1 - my0);
            gb = dmin<uint32_t>((uint32_t)s_gb[g], DSUB_MAX_GROUP_BITS);
        }
        const uint32_t incl = wave_incl_scan_u32(gb);
        const uint32_t ex = incl - gb;                               /* my first bit relative to the tile's */
        const uint64_t tstart = T0 + wave_lane_u32(tstart_q, uni32(q));

        /* one pass stages the bits of all 64 lanes; only when they do not fit the slice (codes far
         * longer than the 9-bit average) the lanes are taken in several runs */
        uint32_t l0 = 0;
        while (l0 < 64u) {
            const uint32_t base = wave_lane_u32(ex, uni32(l0));
            const uint64_t first = tstart + base;
            const uint64_t origin = first & ~31ull;                  /* payload bit of stage word 0 */
            const uint32_t lead = (uint32_t)(first - origin);
            const unsigned long long over = __ballot((uint32_t)lane >= l0 && lead + (incl - base) > CAP_BITS);
            const uint32_t l1 = over ? (uint32_t)__builtin_ctzll(over) : 64u;      /* > l0: one group always fits */
            const uint32_t need_bits = lead + (wave_lane_u32(incl, uni32(l1 - 1u)) - base);
            const uint32_t nwords = ((need_bits + 31u) >> 5) + DSUB_SLACK_WORDS + 2u;   /* <= SLICE_WORDS - 4 */
            const uint32_t lim = ((need_bits + 31u) >> 5) * 32u + 64u;                  /* bits a walk may look at */
#ifndef DSUB_ABLATE_STAGE
            {
                /* Four words per lane and step: 20 payload bytes at a 4-byte aligned address (one
                 * 16-byte and one 4-byte load), four v_perm (byte order and the payload's byte
                 * misalignment in one selector), one 16-byte LDS store; all loads before the first use. */
                struct __attribute__((packed, aligned(4))) Q4 { uint32_t x, y, z, w; };
                constexpr int STEPS = 3;
                const uint64_t byte0 = origin >> 3;
                if (byte0 + 4ull * nwords + 24ull <= pay_bytes) {
                    const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)(pay + byte0));
                    const uint32_t m = (uint32_t)(a & 3u);
                    const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
                    const uint32_t *qw = reinterpret_cast<const uint32_t *>(a - m);
                    Q4 v[STEPS];
                    uint32_t x[STEPS];
                    /* every lane loads (lanes past the needed words load the last ones again and do not
                     * store them): a load under a condition leaves its 15 destination registers
                     * "maybe unchanged", which the compiler then carries around the whole tile loop */
                    const uint32_t last4 = (nwords - 1u) & ~3u;
#pragma unroll
                    for (int k = 0; k < STEPS; k++) {
                        const uint32_t i4 = dmin<uint32_t>(4u * ((uint32_t)lane + 64u * (uint32_t)k), last4);
                        v[k] = *reinterpret_cast<const Q4 *>(qw + i4);
                        x[k] = qw[i4 + 4];
                    }
#pragma unroll
                    for (int k = 0; k < STEPS; k++) {
                        const uint32_t i4 = 4u * ((uint32_t)lane + 64u * (uint32_t)k);
                        if (i4 < nwords)
                            *reinterpret_cast<uint4 *>(stage + i4) =
                                make_uint4(__builtin_amdgcn_perm(v[k].y, v[k].x, sel), __builtin_amdgcn_perm(v[k].z, v[k].y, sel),
                                           __builtin_amdgcn_perm(v[k].w, v[k].z, sel), __builtin_amdgcn_perm(x[k], v[k].w, sel));
                    }
                } else {
                    for (uint32_t i = (uint32_t)lane; i < nwords; i += 64u)
                        stage[i] = load_be32(pay, byte0 + 4ull * i, pay_bytes);
                }
            }
#endif
            DPROF_ADD(9, pt); pt = DPROF_T();
#ifdef DSUB_ABLATE_LANES
            if (false) {
#else
            if ((uint32_t)lane >= l0 && (uint32_t)lane < l1 && nsym) {
#endif
                const uint32_t s = lead + (ex - base);
                LinReader rd;
                rd.st = stage;
                rd.load(s);
                uint8_t *dst = gout + my0;
                uint32_t special = 0;                               /* OR of the table entries: bits 14/15 = not a leaf */
                if (nsym == DSUB_SPL) {
                    /* The common case has no branch: an entry that is not a leaf (a `long` code, a
                     * walk that leaves the tree) advances by its 5-bit field like a leaf and is only
                     * remembered; such a lane decodes its group again below, step by step.
                     * Two rounds of 16 symbols (rolled: the unrolled form does not fit 64 VGPRs), each
                     * stored as 16 bytes with the DEFAULT cache policy: a lane's store covers half of a
                     * 32-byte sector and the other half follows a round later; streaming (nt) stores
                     * then reach HBM as partial writes (1.05 -> 0.81 ms per GiB without nt; turning the
                     * wave's 2 KiB round in LDS for whole-line nt stores needs 8 more registers, and with
                     * 80 VGPRs = 3 workgroups per CU the kernel takes 0.97 ms). */
                    const bool aligned = (((uintptr_t)dst) & 15u) == 0;
#ifndef DSUB_BUF_READER
                    /* Position-based window: every two symbols the 64 bits at the position are read
                     * again from the stage (one ds_read2_b32, one 64-bit shift) - no bit buffer to
                     * refill, no branch: 8 wave instructions per symbol where the refilled 64-bit buffer
                     * (-DDSUB_BUF_READER) has 14, most of them in the refill block that some lane needs at
                     * every test; more LDS reads instead (zipf255 0.65 -> 0.62 ms, uniform bytes +-0). */
#ifndef DSUB_WINDOW_V1
                    /* The position register holds (payload bit - 1) + 8 * (LDS byte address of the
                     * stage): (Q >> 3) & ~3 IS the LDS address of the word pair, and the 32 bits at the
                     * position are ONE v_alignbit_b32 of the pair by ~Q (shift amounts 0..31: the pair
                     * is the one that holds bit position - 1, so the position is never the pair's first
                     * bit).  The second table index comes from a 32-bit shift of that register: 13 simple
                     * instructions per two symbols where the 64-bit window (-DDSUB_WINDOW_V1) had 15 with
                     * two 64-bit shifts (zipf255 0.643 -> 0.626 ms).
                     * (Measured and dropped: a lane decoding the two halves of its group side by side -
                     * the encoder also wrote the bits of every group's first half - to have two
                     * independent chains of LDS reads per lane.  At 64 VGPRs the tile loop spills
                     * (0.88 ms), at 78 VGPRs = 3 workgroups per CU it takes 0.72 ms: the loop is bound by
                     * VALU + LDS throughput, not by the latency of its dependent reads.) */
                    typedef const __attribute__((address_space(3))) uint32_t *lds_words;
                    typedef const __attribute__((address_space(3))) uint16_t *lds_halves;
                    const uint32_t q0 = s - 1u + 8u * (uint32_t)(uintptr_t)(lds_words)stage;
                    const uint32_t lut_addr = (uint32_t)(uintptr_t)(lds_halves)sh.lut;
/* L2 = the block has second-level entries (sh.l2n): a lookup that meets one - decided for the whole
 * wave by a ballot - goes on to the second table with the bits behind the 12-bit prefix.  Blocks
 * without such codes (zipf255, uniform bytes) run the loop without the two ballots per window. */
#define DSUB_WINDOW(Q, acc, L2)                                                                               \
                    {                                                                                         \
                        lds_words wp_ = (lds_words)(uintptr_t)(((Q) >> 3) & ~3u);                              \
                        const uint32_t d1_ = __builtin_amdgcn_alignbit(wp_[0], wp_[1], ~(Q));                 \
                        uint32_t e1_ = *(lds_halves)(uintptr_t)(lut_addr + ((d1_ >> 19) & 0x1ffeu));          \
                        if (L2 && __ballot(DSUB_IS_L2(e1_))) {                                                \
                            if (DSUB_IS_L2(e1_)) e1_ = dsub_l2<THREADS>(sh, e1_, d1_);                        \
                        }                                                                                     \
                        const uint32_t l1_ = (e1_ >> 8) & 31u;                                                \
                        const uint32_t d2_ = d1_ << l1_;                                                      \
                        uint32_t e2_ = *(lds_halves)(uintptr_t)(lut_addr + ((d2_ >> 19) & 0x1ffeu));          \
                        if (L2 && __ballot(DSUB_IS_L2(e2_))) {                                                \
                            /* (the window has 32 - l1 bits left: a second code of the second level behind  \
                             *  a long first one is left to the step-by-step path) */                        \
                            if (DSUB_IS_L2(e2_))                                                              \
                                e2_ = (l1_ + DEC_LUT_BITS + DSUB_L2_BITS <= 32u) ? dsub_l2<THREADS>(sh, e2_, d2_) : (uint32_t)DEC_E_LONG; \
                        }                                                                                     \
                        special |= e1_ | e2_;                                                                 \
                        acc = __builtin_amdgcn_alignbit(e1_, acc, 8);                                         \
                        acc = __builtin_amdgcn_alignbit(e2_, acc, 8);                                         \
                        (Q) += l1_ + ((e2_ >> 8) & 31u);                                                      \
                    }
#define DSUB_ROUNDS(L2)                                                                                        \
                    _Pragma("unroll 1")                                                                       \
                    for (int h = 0; h < 2; h++) {                                                             \
                        uint32_t w[4];                                                                        \
                        _Pragma("unroll")                                                                     \
                        for (int k = 0; k < 4; k++) {                                                         \
                            uint32_t acc = 0;                                                                 \
                            _Pragma("unroll")                                                                 \
                            for (int j = 0; j < 2; j++) DSUB_WINDOW(Q, acc, L2)                               \
                            w[k] = acc;                                                                       \
                        }                                                                                     \
                        if (aligned) {                                                                        \
                            reinterpret_cast<uint4 *>(dst)[h] = make_uint4(w[0], w[1], w[2], w[3]);           \
                        } else {                                                                              \
                            _Pragma("unroll")                                                                 \
                            for (int k = 0; k < 16; k++) dst[16 * h + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3))); \
                        }                                                                                     \
                    }
                    uint32_t Q = q0;
                    if (use_l2) { DSUB_ROUNDS(true) } else { DSUB_ROUNDS(false) }
#undef DSUB_ROUNDS
#undef DSUB_WINDOW
                    const uint32_t p = s + (Q - q0);
#else
                    uint32_t p = s;
#pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        uint32_t w[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            uint32_t acc = 0;
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                const uint32_t *wp = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(stage) + ((p >> 3) & ~3u));
                                uint64_t b = (((uint64_t)wp[0] << 32) | wp[1]) << (p & 31u);
                                const uint32_t e1 = sh.lut[(uint32_t)(b >> 32) >> (32 - DEC_LUT_BITS)];
                                const uint32_t l1 = (e1 >> 8) & 31u;
                                b <<= l1;
                                const uint32_t e2 = sh.lut[(uint32_t)(b >> 32) >> (32 - DEC_LUT_BITS)];
                                special |= e1 | e2;
                                acc = __builtin_amdgcn_alignbit(e1, acc, 8);
                                acc = __builtin_amdgcn_alignbit(e2, acc, 8);
                                p += l1 + ((e2 >> 8) & 31u);
                            }
                            w[k] = acc;
                        }
                        if (aligned) {
                            reinterpret_cast<uint4 *>(dst)[h] = make_uint4(w[0], w[1], w[2], w[3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; k++) dst[16 * h + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                        }
                    }
#endif
                    rd.load(p);
                }
#else
#pragma unroll 1
                    for (int h = 0; h < 2; h++) {
                        uint32_t w[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            uint32_t acc = 0;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const uint32_t e = sh.lut[rd.index()];
                                special |= e;
                                acc = __builtin_amdgcn_alignbit(e, acc, 8);
                                rd.consume((e >> 8) & 31u);
                                if (j & 1) { if (rd.avail <= 32) rd.refill(); }
                            }
                            w[k] = acc;
                        }
#ifdef DSUB_ABLATE_STORES
                        if (w[0] == 0x12345678u && w[3] == 0x9abcdef0u) dst[0] = 1;
                        else if (false) {
#else
                        if (aligned) {
#endif
                            reinterpret_cast<uint4 *>(dst)[h] = make_uint4(w[0], w[1], w[2], w[3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; k++) dst[16 * h + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                        }
                    }
                }
#endif
                bool group_ok = rd.pos() - s == gb;                 /* (b): exactly the bits of the group */
                if (__builtin_expect(__ballot(nsym != DSUB_SPL || (special & 0xC000u)) != 0ull, 0)) {
                    if (nsym != DSUB_SPL || (special & 0xC000u))    /* the block's last, short group; groups with long codes */
                        group_ok = dsub_redo_group<THREADS>(sh, stage, s, nsym, lim, gb, dst);
                }
                if (!group_ok) ok = false;
            }
            DPROF_ADD(10, pt); pt = DPROF_T();
            l0 = l1;
        }
    }
    *end_bit = T0 + chunk_bits;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

/* Work list of blocks the sub-index path could not verify (zeroed state between launches: the
 * flags by decode_fix_kernel, the count by decode_prepare_kernel of the next call). */
struct DecFixList {
    uint32_t *count;      /* [1] */
    uint32_t *blocks;     /* [nblocks] */
    uint32_t *flag;       /* [nblocks] block already listed */
};


#ifndef DSUB_WAVES_PER_SIMD
#define DSUB_WAVES_PER_SIMD 8
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS, DSUB_WAVES_PER_SIMD) void decode_sub_kernel(
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
    /* everything about the block is the same in all lanes: kept in SGPRs (held in VGPRs these
     * values pushed the tile loop's state out to scratch, 8 reloads per wave tile) */
    HufDecodeMeta m = dmeta[blk];
    m.block_len = uni64(m.block_len);
    m.tree_len = (int16_t)uni32((uint32_t)(uint16_t)m.tree_len);
    m.leaf = (int16_t)uni32((uint32_t)(uint16_t)m.leaf);
    m.status = (int32_t)uni32((uint32_t)m.status);
    const uint64_t obase = uni64(lens.gprefix[blk / SCAN_GROUP] + lens.local[blk]);
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
    const uint64_t o0 = uni64(offsets[blk]);
    const uint64_t o1 = dmin<uint64_t>(uni64(offsets[blk + 1]), stream_len);
    const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
    const uint8_t *tree = stream + o0 + HUF_HEADER_FIXED;
    const uint8_t *pay = tree + 2 * (int)m.tree_len;
    bool good;
    int leaf = m.leaf;
    int rc = HUFE_OK;
    if (leaf < 0)
        dsub_prefetch<THREADS>(sh, sub.group_bits + blk * sub.gpb + sym0 / DSUB_SPL,
                               (uint32_t)((sym1 - sym0 + DSUB_SPL - 1) / DSUB_SPL));
    const uint64_t T0 = uni64(sub.tile_bits[blk * sub.tpb + sym0 / HUF_SUB_TILE]);   /* first payload bit of the chunk, as told */
#ifndef DSUB_ABLATE_TABLES      /* (diagnostic builds: what the kernel costs without one of its phases) */
    if (leaf < 0) {
#ifndef DSUB_NO_FAST_TABLES
        if (!dsub_fast_tables<THREADS>(sh, tree, m.tree_len, sub.lens + blk * HUF_NSYM))
#endif
            rc = dec_build_tables<THREADS, false>(sh, tree, m.tree_len, &leaf);
    }
#endif
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
        good = decode_payload_sub<THREADS>(sh, pay, pay_bytes, sym0, sym1, T0, out + obase, &end_bit);
        /* (c) the next chunk starts where this one ends */
        if (good && sym1 < m.block_len && sub.tile_bits[blk * sub.tpb + sym1 / HUF_SUB_TILE] != end_bit) good = false;
    }
#ifdef DSUB_ABLATE_VERIFY       /* (diagnostic builds with a phase removed produce garbage: do not decode it again) */
    good = true;
#endif
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
