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
 * group of 32 symbols and the first payload bit of every tile of 2 048 symbols (= 64 groups = what
 * one wavefront takes at a time).
 *
 * The self-synchronising decoder (decode.hpp) has to find out where codewords start: every symbol
 * is decoded at least twice (count pass + write pass) plus the synchronisation rounds.  Here a
 * lane is TOLD where its 32 symbols start, decodes them once through a table of the first 12 code bits (a second
 * level for codes of up to 18 bits) and stores them as one 32-byte sector.  What it is told is
 * verified, not trusted:
 *   (a) the block's first tile starts at payload bit 0,
 *   (b) a lane's 32 codewords take exactly the bits its group is said to have, none of its walks
 *       leaves the tree and none needs bits past the payload,
 *   (c) a tile ends where the next tile of the block is said to start (the block's last: inside
 *       the payload).
 * By induction over the lanes these make the output equal to that of the in-order decoder of
 * src/decoder.c:34-96.  A block that fails any of them - a stale or foreign sub-index, a damaged
 * stream - is appended to a list and decoded again by decode_fix_kernel with the exact
 * self-synchronising decoder, which also produces the reference's error code: a wrong sub-index
 * costs time, never correctness.
 *
 * Round 5: every wave is on its own from the table build on (round 4 staged a chunk's 2 048 group
 * counts in LDS and summed them across the workgroup to find the waves' starts - 4 KiB, a load
 * through LDS-DMA, two scans and two barriers - where the sub-index now simply says where a wave
 * tile starts), and the payload words of a tile lie in LDS as one private COLUMN per lane: word g
 * of a lane's group at row D - 1 - g of the wave's slice, a row = the 64 lanes' words side by side.
 * A lane's reads fall into its own bank whatever the positions of the other lanes are (round 4: a
 * linear stage, the window read - two dwords a lane at data-dependent offsets - cost 10 LDS cycles
 * where it now costs 4; SQ_LDS_BANK_CONFLICT was 60 % of the kernel's LDS cycles).  The price: a
 * lane loads its own twelve dwords (three 16-byte loads at the lane's own 4-byte aligned address;
 * neighbours overlap, the overlap comes from the vector cache) instead of a 16-byte share of the
 * tile's, and a group must fit eleven words behind its first (at most 320 bits, i.e. 10 bits a
 * symbol; tiles beyond that are staged and decoded step by step).
 * ==================================================================================== */
#define DSUB_SPL 32                         /* symbols per lane = HUF_SUB_GROUP */
#define DSUB_L2_BITS 6u                     /* a second-level table takes codes of up to 12 + 6 bits */
#define DSUB_L2_ENTRIES 768u                /* it lives in DsubShared::ent */
#define DSUB_SLACK_WORDS 16                 /* (step-by-step path) staged behind the last needed word: what a lane that runs wild (32 look-ups of at most 18 bits) or a long code's walk may look at */
#define DSUB_MAX_GROUP_BITS (DSUB_SPL * HUF_CODE_MAXBITS)
#define DSUB_CHUNK_SYMS 65536u               /* symbols one workgroup decodes: 32 wave tiles */
#define DSUB_LUT_BITS 11u                    /* the first-level table is indexed by the 11 bits BEHIND a codeword's first bit: that bit is 0 in
                                                every code of an encoder-made tree (the root has a left child only, tree.c:410-413), and the
                                                2 048 entries of a first bit 1 all said the same thing.  The first bits are OR-ed as they go by. */
#define DSUB_ROWS 12                         /* words of a lane's column */
#define DSUB_COL_BITS (32u * (DSUB_ROWS - 1))   /* a lane's lead + group bits may be this much: the window at the group's last
                                                codeword reads the word behind the one it starts in */
static_assert(HUF_SUB_TILE == 64 * DSUB_SPL, "a sub-index tile is a wave tile");

/* The staged payload words of a wave lie in its LDS slice in REVERSED order: word g (big-endian, 32 payload bits)
 * at top[-g].  The hot loop's position register then counts DOWN, and both the pair's LDS address and the
 * v_alignbit_b32 shift amount are plain bit fields of it (see DSUB_WINDOW). */
__device__ __forceinline__ uint32_t rev_word(const uint32_t *top, uint32_t g) { return top[-(int32_t)g]; }

/* 64-bit left-aligned bit buffer over the reversed stage: the step-by-step path (long codes, short groups) */
struct RevReader {
    const uint32_t *top;
    uint32_t hi, lo;
    int32_t avail;
    uint32_t gf;         /* next staged word to append */

    __device__ __forceinline__ void load(uint32_t pos)
    {
        const uint32_t g = pos >> 5, off = pos & 31u;
        const uint64_t b = (((uint64_t)rev_word(top, g) << 32) | rev_word(top, g + 1)) << off;
        hi = (uint32_t)(b >> 32);
        lo = (uint32_t)b;
        avail = (int32_t)(64u - off);
        gf = g + 2;
    }
    __device__ __forceinline__ uint32_t index() const { return (hi >> (31 - DSUB_LUT_BITS)) & ((1u << DSUB_LUT_BITS) - 1u); }
    __device__ __forceinline__ bool first_bit() const { return (hi >> 31) != 0u; }
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
        const uint64_t t = (uint64_t)rev_word(top, gf) << (32 - avail);
        hi |= (uint32_t)(t >> 32);
        lo |= (uint32_t)t;
        avail += 32;
        gf++;
    }
};

/* Table entries of the sub-index path (uint16), laid out for the hot loop - the code length in the low five
 * bits is a shift amount as it stands, the sum of two entries carries the sum of their lengths in its low byte:
 *   leaf      byte << 8 | len                               len = 1..12 (first table), 13..18 (second level)
 *   level 2   (offset / 4) << 8 | (bits / 2 - 1) << 6 | 0x20  the second-level table of a 12-bit prefix
 *   long      0xFE00                                          a code the tables do not hold
 *   bad       0xFF00                                          the walk leaves the tree
 * An entry that is not a leaf has length 0; `long` and `bad` have a zero low BYTE: a lane that meets one stands
 * still for the rest of its group, and its last look-up says so. */
#define DSE_L2 0x20u
#define DSE_LONG 0xFE00u
#define DSE_BAD 0xFF00u
#define DSE_LEN(e) ((e) & 31u)
#define DSE_IS_L2(e) (((e) & 0x3fu) == DSE_L2)

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
template <class SH>
struct FastLdsOf {
    __device__ static __forceinline__ uint32_t *code(SH &sh) { return sh.lr; }
    __device__ static __forceinline__ uint8_t *len(SH &sh) { return reinterpret_cast<uint8_t *>(sh.lr + 256); }
    __device__ static __forceinline__ uint8_t *sym(SH &sh) { return reinterpret_cast<uint8_t *>(sh.lr + 320); }
    __device__ static __forceinline__ const uint32_t *code(const SH &sh) { return sh.lr; }
    __device__ static __forceinline__ const uint8_t *len(const SH &sh) { return reinterpret_cast<const uint8_t *>(sh.lr + 256); }
    __device__ static __forceinline__ const uint8_t *sym(const SH &sh) { return reinterpret_cast<const uint8_t *>(sh.lr + 320); }
};
template <int THREADS>
using DsubFastLds = FastLdsOf<DecShared<THREADS>>;        /* (decode_fast.hpp keeps its leaves the same way) */

/* LDS of decode_sub_kernel.  (Member names as in DecShared where the table build uses them.) */
template <int THREADS>
struct DsubShared {
    static constexpr int WAVES = THREADS / 64;
    static constexpr uint32_t SLICE_WORDS = 64u * DSUB_ROWS;      /* a wave's columns: row r of lane l at slice[64 r + l] */
    uint16_t lut[1 << DSUB_LUT_BITS];                            /* first-level table */
    uint32_t lr[384];                                            /* the leaves in preorder: code[256] (left-aligned), length[256], byte[256] (FastLdsOf) */
    int16_t ent[DSUB_L2_ENTRIES];                                /* second-level tables */
    /* (the few words a workgroup shares lie behind the second-level tables, in front of the slices: with them the
     *  struct is 78 rows of 256 bytes for four waves - eight workgroups a CU with room to spare; behind the slices they
     *  would make it 80, which is all of the CU's 160 KiB and one allocation granule from seven workgroups) */
    uint32_t part[WAVES];
    uint32_t wtile[3 * WAVES];                                   /* partial sums of the table build's code scan */
    uint32_t fastk;                                              /* leaves of the tables */
    uint32_t l2n;                                                /* entries of the second-level tables (0: none) */
    uint32_t firstone;                                           /* decode_single_leaf */
    uint8_t redo[THREADS];                                       /* per lane: bit i = the group of the wave's i-th tile is to be decoded again, step by step */
    uint64_t late[4];                                            /* what only the step-by-step path behind the tile loop needs - the payload's address and bytes,
                                                                    the chunk's output, the tile starts as told - parked here, not in scalar registers around the loop */
    __attribute__((aligned(256))) uint32_t pay[WAVES * SLICE_WORDS];  /* the waves' slices (the table build's scratch before that); a ROW - 256 bytes -
                                                                    at a multiple of 256: a row's number and a lane's offset in it are bit fields of an LDS address */
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

/* (Latency is what this costs - a workgroup builds its tables before it can do anything else: the tree entries
 * go from global memory straight into the registers that classify them, the checks do not vote one by one
 * - what follows a failed check works on clamped values and is thrown away by the one vote at the end -, the
 * scans are DPP scans with one barrier each, and blocks without codes beyond 12 bits skip the second level.) */
/* what a thread of the table build needs from global memory: the three aligned dwords that hold tree entries
 * 2t .. 2t+3, and (threads 0..63) four of the claimed lengths - requested early, used by dsub_fast_tables */
struct DsubTreeWords {
    uint32_t d[4], lens4, mis;
};
/* entries a thread classifies: 2 (512 threads) or 4 (256); it looks at two more behind them */
template <int THREADS>
__device__ __forceinline__ DsubTreeWords dsub_tree_request(const uint8_t *tree, int tree_len, const uint8_t *__restrict__ lens_g)
{
    constexpr uint32_t EPT = 1024 / THREADS;
    constexpr int NW = EPT / 2 + 2;                              /* aligned dwords that hold EPT + 2 entries at any byte misalignment */
    static_assert((THREADS == 256 || THREADS == 512) && NW <= 4, "two or four entries a thread");
    DsubTreeWords w;
    const int tid = (int)threadIdx.x;
    const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
    w.mis = (uint32_t)(a & 3u);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a - w.mis);
    const uint32_t nbytes = w.mis + 2u * (uint32_t)(tree_len > 0 ? tree_len : 0);      /* bytes from q[0] to the tree's end */
    const uint32_t t4 = 2u * EPT * (uint32_t)tid;                                       /* byte offset of the thread's first dword */
#pragma unroll
    for (int i = 0; i < 4; i++) w.d[i] = (i < NW && t4 + 4u * (uint32_t)i < nbytes) ? q[(EPT / 2) * tid + i] : 0u;
    w.lens4 = (tid < 64) ? reinterpret_cast<const uint32_t *>(lens_g)[tid] : 0u;
    return w;
}

template <int THREADS, class SH>
__device__ bool dsub_fast_tables(SH &sh, int tree_len, const DsubTreeWords &tw)
{
    typedef FastLdsOf<SH> F;
    constexpr int ENT = HUF_TREE_MAX + 1;
    constexpr int WAVES = THREADS / 64;
    static_assert((THREADS == 256 || THREADS == 512) && ENT == 1026, "1 024 entries over the threads, the 1 025th on the side");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t K = (uint32_t)(tree_len - 1) >> 2;
    if (tid == 0) { sh.fastk = 0; sh.l2n = 0; }
    if (tree_len < 9 || tree_len > HUF_TREE_MAX || ((tree_len - 1) & 3) != 0) return false;     /* uniform */
    uint8_t *s_lens = reinterpret_cast<uint8_t *>(sh.pay);                      /* [256] the claimed lengths by byte value */
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(sh.pay + 64);                /* [256] entry index of the k-th leaf */
    uint16_t *s_ent16 = reinterpret_cast<uint16_t *>(sh.pay + 192);            /* [256] the table entry of the k-th leaf's codewords */
    uint16_t *s_mark = reinterpret_cast<uint16_t *>(sh.pay + 512);              /* [2048] twice the number of the leaf that begins at a table entry (0: none, or leaf 0) */
    uint32_t *s_cpart = sh.wtile;                                               /* [3][WAVES] partial sums of the code scan */
    static_assert(sizeof(sh.wtile) >= 3 * WAVES * sizeof(uint32_t), "partials of the code scan");
    bool ok = true;
    /* ---- shape: leaves, nodes, markers.  Thread t classifies entries EPT t .. EPT t + EPT - 1 and looks at the two
     *      behind them (aligned dwords, shifted by the tree's byte misalignment; entries at or past tree_len read
     *      -1) ---- */
    constexpr int EPT = 1024 / THREADS;
    {
        const uint32_t mis = tw.mis;
        if (tid < 64) reinterpret_cast<uint32_t *>(s_lens)[tid] = tw.lens4;
        if (tid < 256) reinterpret_cast<uint4 *>(s_mark)[tid] = make_uint4(0u, 0u, 0u, 0u);
        const int i0 = EPT * tid;
        int e[EPT + 2];
#pragma unroll
        for (int j = 0; j < EPT + 2; j += 2) {
            const uint32_t two = mis ? __builtin_amdgcn_alignbit(tw.d[j / 2 + 1], tw.d[j / 2], 8u * mis) : tw.d[j / 2];
            e[j] = (i0 + j < tree_len) ? (int)(int16_t)(two & 0xffffu) : -1;
            e[j + 1] = (i0 + j + 1 < tree_len) ? (int)(int16_t)(two >> 16) : -1;
        }
        uint32_t mine = 0;
        bool leaf[EPT];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const bool node = e[j] != -1;
            leaf[j] = node && e[j + 1] == -1 && e[j + 2] == -1 && i0 + j + 2 < tree_len;
            if (i0 + j == tree_len - 1 && node) ok = false;                     /* the last entry is a marker */
            mine += (uint32_t)leaf[j] + ((uint32_t)node << 16);
        }
        if (tid == 0 && e[0] == -1) ok = false;                                 /* the root */
        if (tid == THREADS - 1 && i0 + EPT == tree_len - 1 && e[EPT] != -1) ok = false;    /* (entry 1024 has no thread of its own) */
        const uint32_t inc = wave_incl_scan_u32(mine);
        if (lane == 63) sh.part[wave] = inc;
        __syncthreads();                                                        /* (also: s_lens is written, the marks are cleared) */
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const uint32_t x = sh.part[i];
            if (i < wave) base += x;
            tot += x;
        }
        if ((tot & 0xffffu) != K || (tot >> 16) != 2u * K) ok = false;
        uint32_t k = (base + inc - mine) & 0xffffu;                             /* leaves in front of my entries */
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            if (leaf[j] && k < 256u) { s_pos[k] = (uint16_t)(i0 + j); F::sym(sh)[k] = (uint8_t)e[j]; k++; }
        }
    }
    __syncthreads();
    /* ---- claimed lengths -> codes; they must fill the left half of the code space exactly.  A code's share of
     *      the 32-bit code space is 2^(32 - d) <= 2^30: scanned as two 16-bit halves (DPP, no 64-bit shuffles).
     *      (Round 5: only the waves that hold leaves - K <= 256: the first four at most - do any of this; the
     *      table build is a third of the kernel's vector instructions for a block of 64 KiB.) ---- */
    constexpr uint32_t EPF = 2048u / THREADS;                                   /* table entries a thread fills */
    constexpr uint32_t WSPAN = 64u * EPF;                                       /* ... a wave */
    const bool leafwave = uni32((uint32_t)wave * 64u) < K;
    uint32_t d = 2;
    bool anylong;
    {
        uint32_t whi = 0, wlo = 0, ihi = 0, ilo = 0;
        if (leafwave) {
            if ((uint32_t)tid < K) {
                d = s_lens[F::sym(sh)[tid]];
                if (d < 2u || d > 32u) { ok = false; d = 2; }
                else if (d >= 16u) wlo = 1u << (32u - d);                       /* <= 2^16 */
                else whi = 1u << (16u - d);                                     /* 2^(32 - d) >> 16 */
            }
            ihi = wave_incl_scan_u32(whi);
            ilo = wave_incl_scan_u32(wlo);
        }
        const unsigned long long lg = __ballot(d > (uint32_t)DEC_LUT_BITS);
        if (lane == 63) {
            s_cpart[wave] = ihi;
            s_cpart[WAVES + wave] = ilo;
            s_cpart[2 * WAVES + wave] = lg != 0ull;
        }
        __syncthreads();
        uint32_t lf = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) lf |= s_cpart[2 * WAVES + i];
        anylong = uni32(lf) != 0u;
        if (leafwave) {
            uint64_t base = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < (WAVES < 4 ? WAVES : 4); i++) {                 /* (the waves behind the fourth hold no leaves) */
                const uint64_t x = ((uint64_t)s_cpart[i] << 16) + s_cpart[WAVES + i];
                if (i < wave) base += x;
                tot += x;
            }
            if (tot != (1ull << 31)) ok = false;
            if ((uint32_t)tid < K) {
                const uint32_t c = (uint32_t)(base + (((uint64_t)(ihi - whi)) << 16) + (ilo - wlo));
                F::code(sh)[tid] = c;
                F::len(sh)[tid] = (uint8_t)d;
                /* what the table's entries of this leaf hold, and a mark at the first of them (entry = the code's
                 * first 12 bits; the codes of leaves that share an entry are longer than 12 bits, any of them may
                 * leave its mark) and at the first entry of every wave of the fill below that the leaf covers: the marks
                 * are leaf numbers (doubled: a byte offset into s_ent16) in code order, an entry's leaf is the last mark
                 * at or in front of it, and a wave finds one at its first entry */
                const uint32_t x = c >> (32 - DEC_LUT_BITS);
                const uint32_t span = d <= (uint32_t)DEC_LUT_BITS ? 1u << ((uint32_t)DEC_LUT_BITS - d) : 1u;
                s_ent16[tid] = d <= (uint32_t)DEC_LUT_BITS ? (uint16_t)(((uint32_t)F::sym(sh)[tid] << 8) | d) : (uint16_t)DSE_LONG;
                s_mark[x & 2047u] = (uint16_t)(2u * (uint32_t)tid);             /* (x < 2048 when the lengths are right; if not, ok is false) */
                for (uint32_t bb = (x & ~(WSPAN - 1u)) + WSPAN; bb < x + span && bb < 2048u; bb += WSPAN)   /* the fill waves' first entries behind x */
                    s_mark[bb] = (uint16_t)(2u * (uint32_t)tid);
            }
        }
    }
    __syncthreads();
    /* ---- the entry positions the lengths imply are the stream's ---- */
    if (leafwave && (uint32_t)tid < K) {
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
    /* ---- the table (entry = the 11 bits behind a codeword's first): an entry's leaf is the last mark at or in front
     *      of it - the marks of a thread's entries, a running maximum over the lanes in front (DPP), and the leaf's
     *      entry from s_ent16 (round 4: a binary search per thread and a walk per entry, 180 vector instructions
     *      where this takes 30) ---- */
    {
        static_assert(EPF == 8 || EPF == 4, "eight or four entries a thread, one store");
        const uint32_t x0 = (uint32_t)tid * EPF;
        uint32_t mw[EPF / 2];
        if constexpr (EPF == 8) {
            const uint4 m = *reinterpret_cast<const uint4 *>(s_mark + x0);
            mw[0] = m.x; mw[1] = m.y; mw[2] = m.z; mw[3] = m.w;
        } else {
            const uint2 m = *reinterpret_cast<const uint2 *>(s_mark + x0);
            mw[0] = m.x; mw[1] = m.y;
        }
        uint32_t t2 = mw[0];
#pragma unroll
        for (uint32_t i = 1; i < EPF / 2; i++) t2 = pk_max_u16(t2, mw[i]);
        t2 = dmax<uint32_t>(t2 & 0xffffu, t2 >> 16);                            /* the thread's last mark (0: none) */
        uint32_t r = wave_excl_max_u32(t2);                                     /* the last mark in front of my entries */
        uint32_t e[EPF];
        const __attribute__((address_space(3))) uint8_t *ent_b = (const __attribute__((address_space(3))) uint8_t *)s_ent16;
#pragma unroll
        for (uint32_t j = 0; j < EPF; j++) {
            r = dmax<uint32_t>(r, (mw[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            e[j] = *(const __attribute__((address_space(3))) uint16_t *)(ent_b + r);
        }
        if constexpr (EPF == 8)
            *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        else
            *reinterpret_cast<uint2 *>(sh.lut + x0) = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
    }
    /* ---- second level: the subtree below a 12-bit prefix whose codes are at most DSUB_L2_BITS longer
     *      gets a table of its own (2, 4 or 6 more bits) in sh.ent.  Codes beyond that keep their `long`
     *      entry. ---- */
    if (anylong) {
        __syncthreads();                                 /* the first table is written */
        uint16_t *l2 = reinterpret_cast<uint16_t *>(sh.ent);
        const uint32_t *code = F::code(sh);
        const uint8_t *len = F::len(sh);
        uint32_t size = 0, nb = 0, run_end = 0;
        const uint32_t k = (uint32_t)tid;
        const uint32_t P = (k < K) ? (code[k] >> (32 - DEC_LUT_BITS)) : 0u;
        if (k < K && d > (uint32_t)DEC_LUT_BITS &&
            (k == 0 || len[k - 1] <= DEC_LUT_BITS || (code[k - 1] >> (32 - DEC_LUT_BITS)) != P)) {
            /* first leaf below its prefix: the leaves below one prefix are neighbours (preorder = code order) */
            uint32_t maxd = d, mind = d, jn = k + 1u;
            while (jn < K && jn - k <= (1u << DSUB_L2_BITS) && (code[jn] >> (32 - DEC_LUT_BITS)) == P) {
                maxd = dmax<uint32_t>(maxd, len[jn]);
                mind = dmin<uint32_t>(mind, len[jn]);
                jn++;
            }
            const bool closed = !(jn < K && (code[jn] >> (32 - DEC_LUT_BITS)) == P);
            if (closed && maxd <= (uint32_t)DEC_LUT_BITS + DSUB_L2_BITS && mind > (uint32_t)DEC_LUT_BITS) {
                nb = (maxd - DEC_LUT_BITS + 1u) & ~1u;
                size = 1u << nb;
                run_end = jn;
            }
        }
        uint32_t total;
        const uint32_t off = block_excl_scan_u32<THREADS>(size, sh.part, total);          /* (sizes are multiples of 4: so are the offsets) */
        if (size && off + size <= DSUB_L2_ENTRIES) {
            for (uint32_t jn = k; jn < run_end; jn++) {
                const uint32_t dj = len[jn];
                const uint32_t first = (code[jn] >> (32 - DEC_LUT_BITS - nb)) & (size - 1u);
                const uint32_t count = 1u << (DEC_LUT_BITS + nb - dj);
                const uint16_t entry = (uint16_t)(((uint32_t)F::sym(sh)[jn] << 8) | dj);
                for (uint32_t i = 0; i < count && first + i < size; i++) l2[off + first + i] = entry;
            }
            sh.lut[P] = (uint16_t)(((off >> 2) << 8) | ((nb / 2u - 1u) << 6) | DSE_L2);
        }
        if (tid == 0) sh.l2n = dmin<uint32_t>(total, DSUB_L2_ENTRIES);
    }
    if (tid == 0 && __builtin_expect(true, 1)) sh.fastk = K;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

/* a second-level entry of the first table resolved with the 32 bits at the codeword's start */
template <class SH>
__device__ __forceinline__ uint32_t dsub_l2(const SH &sh, uint32_t e, uint32_t bits32)
{
    const uint32_t nb = ((e >> 5) & 6u) + 2u;
    return reinterpret_cast<const uint16_t *>(sh.ent)[((e >> 8) << 2) + ((bits32 << DEC_LUT_BITS) >> (32u - nb))];
}

/* a codeword longer than the table's 12 bits with dsub_fast_tables' tables: the leaf whose code
 * interval holds the 32 bits at the position; result as dec_rare_packed */
template <class SH>
__device__ __forceinline__ uint64_t dec_rare_fast(const SH &sh, const uint32_t *top, uint32_t pos, uint32_t lim)
{
    typedef FastLdsOf<SH> F;
    const uint32_t g = pos >> 5, o = pos & 31u;
    const uint32_t w = o ? ((rev_word(top, g) << o) | (rev_word(top, g + 1) >> (32u - o))) : rev_word(top, g);
    if (w >> 31) return ((uint64_t)CW_BAD << 40) | (pos + 1u);
    const uint32_t k = dsub_leaf_of(F::code(sh), sh.fastk, w);
    const uint32_t p = pos + (uint32_t)F::len(sh)[k];
    if (p > lim) return (uint64_t)CW_EXH << 40;
    return ((uint64_t)CW_OK << 40) | ((uint64_t)F::sym(sh)[k] << 32) | p;
}

/* One table step of a lane on the step-by-step path: returns the entry (high byte = symbol); *ok is cleared
 * when the lookup is not a codeword.  The rare paths sit behind one wave-uniform branch. */
template <class SH>
__device__ __forceinline__ uint32_t dsub_next(const SH &sh, RevReader &rd, uint32_t lim, bool &ok)
{
    uint32_t e = rd.first_bit() ? (uint32_t)DSE_BAD : (uint32_t)sh.lut[rd.index()];   /* (a first bit of 1 leaves the tree: the root has no right child) */
    if (__builtin_expect(__ballot(DSE_LEN(e) == 0u) != 0ull, 0)) {
        if (DSE_IS_L2(e)) {
            e = dsub_l2(sh, e, rd.hi);
        } else if (e == DSE_LONG) {
            const uint64_t r = dec_rare_fast(sh, rd.top, rd.pos(), lim);
            if ((int)(r >> 40) == CW_OK) {
                rd.load((uint32_t)r);
                e = ((uint32_t)(r >> 32) & 0xffu) << 8;    /* advance 0: the reader already stands behind it */
            } else {
                ok = false;
                e = 1u;
            }
        } else if (DSE_LEN(e) == 0u) {
            ok = false;
            e = 1u;                                        /* keep moving: the lane's result is discarded anyway */
        }
    }
    rd.consume(DSE_LEN(e));
    return e;
}

/* The step-by-step path (long codes, tiles that do not fit the columns, the block's last, short group) stages
 * into the wave's slice linearly and in reversed order: word g at top[-g]. */
template <int THREADS>
struct DsubLds {
    static constexpr uint32_t SLICE_WORDS = DsubShared<THREADS>::SLICE_WORDS;
    static constexpr uint32_t CAP_BITS = (SLICE_WORDS - DSUB_SLACK_WORDS - 2 - 4) * 32u;
    static_assert(CAP_BITS >= DSUB_MAX_GROUP_BITS + 32u, "one lane's group always fits a slice");
    __device__ static __forceinline__ uint32_t *slice(DsubShared<THREADS> &sh, int wave) { return sh.pay + (uint32_t)wave * SLICE_WORDS; }
};

/* A group again, step by step with the rare paths (long codes; the block's last, short group).
 * Out of line: inlined, its state competes with the hot loop's for the 64 VGPRs. */
template <int THREADS>
__device__ __noinline__ bool dsub_redo_group(const DsubShared<THREADS> &sh, const uint32_t *top, uint32_t s, uint32_t nsym,
                                             uint32_t lim, uint32_t gb, uint8_t *dst)
{
    bool ok = true;
    RevReader rd;
    rd.top = top;
    rd.load(s);
    for (uint32_t k = 0; k < nsym; k++) {
        dst[k] = (uint8_t)(dsub_next(sh, rd, lim, ok) >> 8);
        if (rd.avail <= 32) rd.refill();
    }
    return ok && rd.pos() - s == gb;
}

/* A wave tile the columns cannot take (a group of more than 320 bits: codes far longer than the 9-bit average; a
 * tile whose loads would run past the stream's end), or lanes of a tile that met a code the tables do not hold:
 * the lanes in several runs, staged word by word, decoded step by step.  Rare, and out of line for the hot path's
 * registers.  first_bit = the tile's first payload bit, ex / incl = the lane's exclusive / inclusive bit counts
 * inside the tile, active = the lane's group is to be decoded. */
template <int THREADS>
__device__ __noinline__ bool dsub_tile_slow(const DsubShared<THREADS> &sh, uint32_t *top, const uint8_t *pay, uint64_t pay_bytes,
                                            uint64_t first_bit, uint32_t ex, uint32_t incl, uint32_t nsym, bool active, uint8_t *dst)
{
    constexpr uint32_t CAP_BITS = DsubLds<THREADS>::CAP_BITS;
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t gb = incl - ex;
    bool ok = true;
    uint32_t l0 = 0;
    while (l0 < 64u) {
        const uint32_t base = wave_lane_u32(ex, uni32(l0));
        const uint64_t first = first_bit + base;
        const uint64_t origin = first & ~31ull;                  /* payload bit of stage word 0 */
        const uint32_t lead = (uint32_t)(first - origin);
        const unsigned long long over = __ballot(lane >= l0 && lead + (incl - base) > CAP_BITS);
        const uint32_t l1 = over ? (uint32_t)__builtin_ctzll(over) : 64u;      /* > l0: one group always fits */
        const uint32_t need_bits = lead + (wave_lane_u32(incl, uni32(l1 - 1u)) - base);
        const uint32_t nwords = ((need_bits + 31u) >> 5) + DSUB_SLACK_WORDS + 2u;   /* <= SLICE_WORDS - 4 */
        const uint32_t lim = ((need_bits + 31u) >> 5) * 32u + 64u;                  /* bits a walk may look at */
        for (uint32_t i = lane; i < nwords; i += 64u)
            top[-(int32_t)i] = load_be32(pay, (origin >> 3) + 4ull * i, pay_bytes);
        if (lane >= l0 && lane < l1 && nsym && active) {
            if (!dsub_redo_group<THREADS>(sh, top, lead + (ex - base), nsym, lim, gb, dst)) ok = false;
        }
        l0 = l1;
    }
    return ok;
}

/* The symbols [sym0, sym1) of a block (sym0 a multiple of DSUB_CHUNK_SYMS) with the block's sub-index.
 * Returns true (workgroup-uniform) when everything was verified.  told = the first payload bit of the chunk's wave
 * tiles as the sub-index has them (told[q] for the chunk's tile q; told[ntiles] is looked at only when the block goes
 * on behind the chunk), grp = the chunk's group bit counts; readable = bytes that may be loaded from `pay` on (to the
 * end of the stream).
 *
 * Wave w takes tiles w, w + 8, ...: the lanes' bit counts, one scan, the check against the next tile's start, and
 * every lane requests the twelve dwords from the one that holds its first bit on.  The words of the wave's NEXT tile
 * are requested before it decodes this one and wait in twelve registers: the HBM latency passes under the decoding.
 * No barrier after the table build. */
template <int THREADS>
__device__ __forceinline__ bool decode_payload_sub(DsubShared<THREADS> &sh, const uint8_t *tree, int tree_len, const uint8_t *lens_g,
                                   const uint8_t *pay, uint64_t pay_bytes, uint64_t readable,
                                   uint64_t sym0, uint64_t sym1, bool block_goes_on, const uint64_t *__restrict__ told,
                                   const uint16_t *__restrict__ grp, uint8_t *gout)
{
    /* the table build's words from global memory are on their way while the first tile is set up */
    const DsubTreeWords tw = dsub_tree_request<THREADS>(tree, tree_len, lens_g);
    typedef DsubLds<THREADS> L;
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t ROWS = DSUB_ROWS;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = (int)uni32((uint32_t)tid >> 6);       /* (the compiler does not know that a wave's threads share tid >> 6) */
    const uint64_t pay_bits = pay_bytes * 8ull;
    uint32_t *slice = L::slice(sh, wave);
    uint32_t *top = slice + (L::SLICE_WORDS - 1u);                   /* step-by-step path: staged word g at top[-g] */
    const uint32_t nchunk = (uint32_t)(sym1 - sym0);
    const uint32_t ngrp = (nchunk + DSUB_SPL - 1) / DSUB_SPL;
    const uint32_t ntiles = (ngrp + 63u) / 64u;
    bool ok = true;
    unsigned long long pt = DPROF_T();

    typedef const __attribute__((address_space(3))) uint32_t *lds_words;
    typedef __attribute__((address_space(3))) uint32_t *lds_words_w;
    typedef const __attribute__((address_space(3))) uint16_t *lds_halves;
    typedef uint32_t dwords4 __attribute__((ext_vector_type(4)));
    /* The position of a lane: R = 32 * (number of the slice's last row) - position, position = bits from the first
     * bit of the lane's word 0, rows numbered from LDS address 0 in units of 256 bytes (the slices lie at multiples of
     * that).  Word g lies at row last - g, so R >> 5 IS the row of word g + 1 - the lower of the two rows that hold
     * the 32 bits at the position - and the low five bits of R are the amount v_alignbit_b32 shifts that pair by
     * (0..31; at amount 0 the position is the first bit of the LOWER row's word and the upper row is not looked at) -
     * no negation, and a codeword is R -= len.
     * The position REGISTER holds R with a gap: P = (R >> 5) << 8 | (R & 31), bits 5..7 zero.  The row's LDS address is
     * then P & 0xff00 | 4 * lane - one v_and_or_b32, no shift - and v_alignbit_b32 takes P as it stands.  P -= n for
     * n <= 32 is an ordinary subtraction and the gap cleared: a borrow out of the low five bits runs through the gap
     * (it becomes 111) into the row.  (Round 5, tools/calib/valu_rate.hip: on this chip v_add/sub/and/or/xor/mov and
     * v_lshrrev with VGPR or literal operands issue in 2.5 cycles a wave, everything else - three-operand forms, SDWA,
     * DPP, left shifts, an SGPR operand - in 4; a look-up pair was 12.5 instructions of which 7.5 slow, and is 11.5
     * with 6.5 slow.) */
#define DSUB_GAP(R_) ((((R_) << 3) & 0xff00u) | ((R_) & 31u))
    const uint32_t lane4 = 4u * (uint32_t)lane;
    const uint32_t slice_a = uni32((uint32_t)(uintptr_t)(lds_words)slice);                             /* LDS byte address of the wave's slice */
    const uint32_t r_top = 32u * ((slice_a >> 8) + ROWS - 1u);                                           /* R of position 0 */
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(lds_halves)sh.lut;
    const uintptr_t pay_a = (uintptr_t)uni64((uint64_t)(uintptr_t)pay);
    const uintptr_t cout_a = (uintptr_t)uni64((uint64_t)(uintptr_t)(gout + sym0));           /* the chunk's first output byte */
    const uint64_t end_a = uni64((uint64_t)pay_a + readable);                                 /* the stream's end */

    /* A tile's set-up.  The lane's column starts with the aligned 32-bit word of MEMORY that holds the lane's first
     * bit (the staged words are then byte-swapped dwords, whatever the payload's alignment).
         *   ex         : payload bits of the tile in front of the lane's group
     *   lead       : bits of the column's word 0 in front of the group (0..31)
     *   quick      : every group fits its column and the loads stay inside the stream (wave-uniform)
     *   V[3]       : the column's twelve dwords, requested here
     * and check (c) / (a) of the tile.  (The loads are issued - and waited for further down - for every tile, also
     * when there is nothing to load: the block's first bytes then, not stored.  Loads under a condition leave the
     * compiler with "maybe pending" registers at the top of the loop, and its wait for those is a wait for the
     * previous tile's STORES as well.) */
    uint4 V[3];
    uint32_t state_n = 0;                                           /* the lane's part in the tile that was set up last: see DSUB_STATE */
    bool quick_n = false;
    /* What the sub-index says about a tile is requested TWO tiles ahead (the words themselves one tile ahead): the
     * tile's start and the next tile's through the scalar cache (the sub-index is not written by this kernel: constant
     * address space), the lane's bit count by one 2-byte load.  Asked for where they are used, each tile began with two
     * memory round trips one after the other. */
    typedef const __attribute__((address_space(4))) uint64_t *const_u64;
    const const_u64 told_c = (const_u64)(uintptr_t)uni64((uint64_t)(uintptr_t)told);
    uint64_t t_f = 0, tn_f = 0;                                      /* fetched: the tile's first bit, the next tile's */
    uint32_t gb_f = 0;                                               /* fetched: my group's bits */
#define DSUB_FETCH(Q_)                                                                                         \
    {                                                                                                         \
        const uint32_t q_ = (Q_);                                                                             \
        const bool valid_ = q_ < ntiles;                                                                      \
        const uint32_t qq_ = valid_ ? q_ : 0u;                                                                \
        const bool more_ = valid_ && (qq_ + 1u < ntiles || block_goes_on);                                    \
        t_f = told_c[qq_];                                                                                    \
        tn_f = told_c[more_ ? qq_ + 1u : qq_];                                                                \
        /* (through a buffer resource over the chunk's counts: a lane without a group reads beyond it and gets 0 - no        \
         *  condition, no clamp, and a 32-bit offset instead of an address pair) */                                            \
        uint32_t l2_;                      /* (2 * lane from 4 * lane, which the windows keep: not one more register held) */ \
        asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(l2_) : "v"(lane4));                                     \
        gb_f = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(grp_rsrc, valid_ ? (qq_ << 7) | l2_ : 0xfffffffeu, 0, 0); \
    }
/* (Everything about a tile relative to the chunk's first, in 32-bit scalar arithmetic: a chunk's payload is less
 * than 2^22 bits long; a told start that is not inside that is wrong.  The fetch for the tile after the next
 * stands BETWEEN this tile's arithmetic and its loads: what it asks for is then complete when the words are, and
 * the top of the loop waits for nothing - behind the words it would wait for the previous tile's stores as well.) */
#define DSUB_SETUP(Q_, FETCH_Q_)                                                                               \
    {                                                                                                         \
        const uint32_t q_ = (Q_);                                                                             \
        const bool valid_ = q_ < ntiles;                                                                      \
        const uint32_t qq_ = valid_ ? q_ : 0u;                                                                \
        const uint64_t d_ = t_f - T0;                                                   /* the tile's first bit from the chunk's */ \
        const uint32_t rel_t_ = (uint32_t)d_;                                                                 \
        const bool more_ = valid_ && (qq_ + 1u < ntiles || block_goes_on);                                    \
        const uint32_t g_ = qq_ * 64u + (uint32_t)lane;                                                       \
        const uint32_t gb_n = dmin<uint32_t>(gb_f, DSUB_MAX_GROUP_BITS);                                      \
        const uint32_t nsym_n = (valid_ && g_ < ngrp) ? dmin<uint32_t>(DSUB_SPL, nchunk - g_ * DSUB_SPL) : 0u; \
        const uint32_t incl_ = wave_incl_scan_u32(gb_n);                                                      \
        const uint32_t tb_ = wave_lane_u32(incl_, 63);                                                        \
        uint32_t dhi_ = uni32((uint32_t)(d_ >> 32));                                                          \
        asm volatile("" : "+s"(dhi_));     /* (a 32-bit scalar compare: not d_ < 2^32 with the constant in a register pair) */ \
        const bool fine_ = valid_ && chunk_fine && dhi_ == 0u && rel_t_ <= room0 && tb_ <= room0 - rel_t_;   /* (b) */ \
        if (valid_ && !fine_) ok = false;                                                                     \
        if (fine_ && more_ && t_f + tb_ != tn_f) ok = false;                             /* (c) */              \
        const uint32_t rel_ = uni32(lead0 + (fine_ ? rel_t_ : 0u)) + (incl_ - gb_n);     /* my first bit from the chunk's aligned first word */ \
        const uint32_t lead_n = rel_ & 31u;                                                                   \
        /* one register a lane and tile: where the position register starts (15 bits), where (b) wants it to stand  \
         * behind the group (15 bits), and 2 = all 32 symbols / 1 = the block's last, short group / 0 = none */       \
        state_n = DSUB_GAP(r_top - lead_n) | (DSUB_GAP((r_top - lead_n - gb_n) & 0xfffu) << 15) | ((nsym_n == DSUB_SPL ? 2u : (nsym_n ? 1u : 0u)) << 30); \
        const uint32_t off_ = (rel_ >> 5) << 2;                                         /* my column's word 0, bytes from that word */ \
        /* (the last lane's twelve dwords inside the stream: then everybody's) */                              \
        quick_n = fine_ && wave_lane_u32(off_, 63) + 4u * ROWS <= left0 && __ballot(lead_n + gb_n > DSUB_COL_BITS) == 0ull; \
        DSUB_FETCH(FETCH_Q_)                                                                                  \
        _Pragma("unroll")                                                                                     \
        for (int k = 0; k < 3; k++) {                                                                         \
            const dwords4 v_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off_ + 16u * (uint32_t)k, 0, 0); \
            V[k] = make_uint4(v_.x, v_.y, v_.z, v_.w);                                                        \
        }                                                                                                     \
    }
/* the loaded words into the column: word g (byte-swapped) at row ROWS - 1 - g.  (The words are waited for and swapped for
 * every tile, also one that stages nothing: a wait under a condition leaves the compiler with "maybe pending" registers at
 * the top of the loop, and its wait for THOSE - before it overwrites them - is s_waitcnt vmcnt(0): the previous tile's
 * stores and the fetch just issued as well.) */
#define DSUB_TO_SLICE()                                                                                        \
    {                                                                                                         \
        lds_words_w col_ = (lds_words_w)(uintptr_t)(slice_a + lane4);   /* row 0 of my column */               \
        uint32_t wv_[12] = {V[0].x, V[0].y, V[0].z, V[0].w, V[1].x, V[1].y, V[1].z, V[1].w, V[2].x, V[2].y, V[2].z, V[2].w}; \
        _Pragma("unroll")                                                                                     \
        for (int g = 0; g < (int)ROWS; g++) {                                                                 \
            wv_[g] = __builtin_bswap32(wv_[g]);                                                               \
            asm volatile("" : "+v"(wv_[g]));                                                                  \
        }                                                                                                     \
        if (quick_n) {                                                                                        \
            _Pragma("unroll")                                                                                 \
            for (int g = 0; g < (int)ROWS; g++) col_[64 * ((int)ROWS - 1 - g)] = wv_[g];                      \
        }                                                                                                     \
    }
    static_assert(DSUB_ROWS == 12, "three 16-byte loads a lane");

    sh.redo[tid] = 0;                                               /* (the lane's own byte: no barrier) */
    if (tid == 0) {                                                  /* (read behind the tile loop: the table build's barriers lie in between) */
        sh.late[0] = (uint64_t)(uintptr_t)pay;
        sh.late[1] = pay_bytes;
        sh.late[2] = (uint64_t)(uintptr_t)(gout + sym0);
        sh.late[3] = (uint64_t)(uintptr_t)told;
    }
    /* the chunk's first tile, as told: (a), and where its first bit lies */
    const uint64_t T0 = told_c[0];
    uint64_t room64;
    const bool chunk_fine = !__builtin_sub_overflow(pay_bits, T0, &room64) && !(sym0 == 0 && T0 != 0);
    if (!chunk_fine) ok = false;
    const uint32_t room0 = (uint32_t)(room64 >> 32) != 0u ? 0xffffffffu : (uint32_t)room64;   /* payload bits from the chunk's first on (all that matter) */
    const uintptr_t a0 = pay_a + (uintptr_t)((chunk_fine ? T0 : 0ull) >> 3);
    const uint32_t lead0 = ((uint32_t)T0 & 7u) + 8u * (uint32_t)(a0 & 3u);
    const uintptr_t base0 = a0 & ~(uintptr_t)3;                     /* (may begin up to 3 bytes in front of the payload: header and tree lie there) */
    const uint64_t left64 = end_a - (uint64_t)base0;                /* readable bytes from there on */
    const uint32_t left0 = (uint32_t)(left64 >> 32) != 0u ? 0xffffffffu : (uint32_t)left64;
    /* The words come through a buffer resource from that word to the stream's end: a tile that stages nothing loads
     * whatever its offsets say and gets zeros where that is beyond the end - no safe address to point idle loads at,
     * and 32-bit offsets instead of address pairs.  A tile whose loads reach beyond the end - the stream's last - is
     * not quick: what a load that lies across the end returns (the hardware zeroes at least the dword that does, bytes
     * of the stream among them) is not relied upon. */
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base0, (short)0, (int)left0, 0x00020000);
    const __amdgpu_buffer_rsrc_t grp_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)grp, (short)0, (int)(2u * ngrp), 0x00020000);
    uint32_t q = (uint32_t)wave;
    DSUB_FETCH(q)
    DSUB_SETUP(q, q + WAVES)
    /* ... and the wave's first tile is on its way while the tables are built: the quick way from the sub-index's
     * code lengths, checked against the tree; any other tree leaves the block to the exact decoder */
    {
        unsigned long long kt = DPROF_T();
        if (!dsub_fast_tables<THREADS>(sh, tree_len, tw)) return false;          /* (workgroup-uniform) */
        DPROF_ADD(3, kt);
    }
    const bool use_l2 = uni32(sh.l2n) != 0u;                          /* the block has codes in a second-level table */
    DSUB_TO_SLICE()
    /* what the loop leaves for later (bit i = the wave's i-th tile): tiles that are not `quick`, and - per lane -
     * groups to be decoded again step by step.  Those go through dsub_tile_slow BEHIND the loop: a call inside
     * it would have the loop's registers saved and restored around a path that is next to never taken. */
    uint32_t slow_tiles = 0;
    /* The lane's 32 bytes leave through a buffer resource over the chunk's output: a lane that has nothing to store
     * (a tile that is not quick, the block's last, short group, lanes behind the block's end) stores at an offset beyond
     * the resource and the hardware drops it.  No branch around the stores - and that is the point: between the
     * request of the next tile's words and the wait for them lie the same two stores on every path, so the wait is
     * s_waitcnt vmcnt(2), not a wait for the stores' acknowledgement. */
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)cout_a, (short)0, (int)nchunk, 0x00020000);
    const bool out_ok = (cout_a & 3u) == 0;                          /* (dword stores: any other output address takes the step-by-step path) */

#pragma unroll 1
    for (uint32_t ti = 0; q < ntiles; q += WAVES, ti++) {
        pt = DPROF_T();
        const uint32_t my0 = (q * 64u + (uint32_t)lane) * DSUB_SPL;     /* relative to the chunk: 32-bit arithmetic, one offset register */
        const uint32_t state = state_n;                                 /* this tile's */
        const bool cur_quick = quick_n && out_ok;
        /* the next tile's words are requested now and arrive while this one is decoded (what the sub-index says
         * about the tile behind it, too) */
        DSUB_SETUP(q + WAVES, q + 2 * WAVES)
        DPROF_ADD(9, pt); pt = DPROF_T();
        if (!cur_quick) slow_tiles |= 1u << ti;
        {
            /* Every lane runs the loop, whatever its state (a lane without a whole group reads stale columns and its
             * result goes nowhere).  Every two symbols the 32 bits at the position are read again from the column
             * (one ds_read2st64_b32, one v_alignbit_b32) - no bit buffer to refill - and looked up twice; the
             * entries' low five bits shift the window and their sum moves the position as they stand.  An entry that
             * is not a leaf has length 0: the lane stands still from then on (a `long` code, a walk that leaves the
             * tree: the low byte is 0) and its last look-up tells; such a lane decodes its group again behind the
             * loop, step by step.  The lane's 32 bytes leave as two 16-byte stores back to back, DEFAULT cache
             * policy: one whole 32-byte sector (a store per 16 symbols left half-written sectors in L2 for a round,
             * and one in nine of them was evicted like that and written twice; streaming stores reach HBM as partial
             * writes altogether).
             * (Measured and dropped: a 64-bit buffer with refills, 14 instead of 8 instructions per symbol; a lane
             * decoding the two halves of its group side by side.) */
            const bool whole = (state >> 30) == 2u;
            uint32_t R = state & 0x7fffu;                               /* (with the gap: see above) */
            uint32_t special = 0, e_last = 0, firsts = 0;
            const uint32_t o_ = (cur_quick && whole) ? my0 : 0x7fffff00u;
/* L2 = the block has second-level entries (sh.l2n): a lookup that meets one - decided for the whole
 * wave by a ballot - goes on to the second table with the bits behind the 12-bit prefix; one that cannot
 * (too few bits left in the window) stays what it is and is remembered in `special`.  Blocks
 * without such codes (zipf255, uniform bytes) run the loop without the ballots. */
#define DSUB_WINDOW(PAIR, L2)                                                                                 \
            {                                                                                             \
                lds_words wp_ = (lds_words)(uintptr_t)((R & 0xff00u) | lane4);                            \
                const uint32_t d1_ = __builtin_amdgcn_alignbit(wp_[64], wp_[0], R);                        \
                uint32_t e1_ = *(lds_halves)(uintptr_t)(lut_addr + ((d1_ >> 19) & 0xffeu));               \
                if (L2 && __ballot(DSE_IS_L2(e1_))) {                                                     \
                    if (DSE_IS_L2(e1_)) e1_ = dsub_l2(sh, e1_, d1_);                                      \
                }                                                                                         \
                const uint32_t d2_ = d1_ << (e1_ & 31u);                                                  \
                uint32_t e2_ = *(lds_halves)(uintptr_t)(lut_addr + ((d2_ >> 19) & 0xffeu));               \
                if (L2 && __ballot(DSE_IS_L2(e2_))) {                                                     \
                    /* (the window has 32 - len bits left) */                                             \
                    if (DSE_IS_L2(e2_) && DSE_LEN(e1_) + DEC_LUT_BITS + DSUB_L2_BITS <= 32u)              \
                        e2_ = dsub_l2(sh, e2_, d2_);                                                      \
                }                                                                                         \
                if (L2) {                                                                                 \
                    special |= e1_ | e2_;                                                                 \
                    asm volatile("" : "+v"(special));     /* (now: not 32 entries kept for one big OR at the end) */ \
                }                                                                                         \
                firsts |= d1_ | d2_;          /* (one v_or3_b32: every codeword's first bit, at bit 31) */   \
                asm volatile("" : "+v"(firsts));     /* (now: not 32 windows kept for one big OR at the end) */    \
                /* the two bytes, and behind them the two lengths: R -= len1 + len2 is then ONE v_dot4c_i32_i8 with \
                 * the bytes (0, 0, -1, -1) - not an addition and an SDWA subtraction.  Codes of the first table are  \
                 * at most 12 bits: one step; with second-level codes (up to 18 bits) a length at a time. */         \
                PAIR = __builtin_amdgcn_perm(e2_, e1_, 0x04000501u);                                      \
                if (L2) {                                                                                 \
                    R = (uint32_t)__builtin_amdgcn_sdot4((int)PAIR, (int)0x00ff0000, (int)R, false) & 0xff1fu; \
                    R = (uint32_t)__builtin_amdgcn_sdot4((int)PAIR, (int)0xff000000, (int)R, false) & 0xff1fu; \
                } else {                                                                                  \
                    R = (uint32_t)__builtin_amdgcn_sdot4((int)PAIR, (int)0xffff0000, (int)R, false) & 0xff1fu; \
                }                                                                                         \
                e_last = e2_;                                                                             \
                /* (the windows of this form one after the other: interleaved they hold forty registers, and \
                 *  what lives around the tile loop is spilled for them - in the other form's path as well) */ \
                if (L2) __builtin_amdgcn_sched_barrier(0);                                                \
            }
#define DSUB_ROUNDS(L2)                                                                                        \
            {                                                                                             \
                uint32_t w[8];                                                                            \
                _Pragma("unroll")                                                                         \
                for (int k = 0; k < 8; k++) {                                                             \
                    uint32_t p01, p23;                                                                    \
                    DSUB_WINDOW(p01, L2)                                                                  \
                    DSUB_WINDOW(p23, L2)                                                                  \
                    w[k] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);                                  \
                }                                                                                         \
                a4_.x = w[0]; a4_.y = w[1]; a4_.z = w[2]; a4_.w = w[3];                                   \
                b4_.x = w[4]; b4_.y = w[5]; b4_.z = w[6]; b4_.w = w[7];                                   \
            }
            /* (two loops, whole: the compiler otherwise moves their common first look-up in front of the branch
             *  and spills the position register around it) */
            dwords4 a4_, b4_;
            if (use_l2) { asm volatile("; codes in a second-level table" : "+v"(R)); DSUB_ROUNDS(true) }
            else { asm volatile("; every code in the first table" : "+v"(R)); DSUB_ROUNDS(false) }
            /* (the stores behind the two forms, not in each: every path to the wait for the next tile's words then
             *  has the same two stores behind those loads, and the wait is s_waitcnt vmcnt(4..2) - in the forms, the
             *  compiler sees a path around them and waits for the stores' acknowledgement, vmcnt(0)) */
            __builtin_amdgcn_raw_buffer_store_b128(a4_, out_rsrc, o_, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(b4_, out_rsrc, o_ + 16u, 0, 0);
#undef DSUB_ROUNDS
#undef DSUB_WINDOW
            /* a group to be decoded again: the block's last, short one; one that met a code the tables do not hold; a
             * first bit of 1 (the step-by-step path says what the reference says).  Else (b): exactly the group's bits */
            const bool redo = cur_quick && (state >> 30) != 0u &&
                              (!whole || DSE_LEN(e_last) == 0u || (special & DSE_L2) != 0u || (firsts >> 31) != 0u);
            if (__builtin_expect(__ballot(redo) != 0ull, 0)) {
                if (redo) sh.redo[tid] |= (uint8_t)(1u << ti);
            }
            if (cur_quick && whole && !redo && R != ((state >> 15) & 0x7fffu)) ok = false;
        }
        DPROF_ADD(10, pt);
        /* the columns are free: the next tile's words (the wait for them counts the stores behind them out) */
        DSUB_TO_SLICE()
    }
#undef DSUB_TO_SLICE
#undef DSUB_SETUP
#undef DSUB_FETCH
    const uint32_t redo_tiles = sh.redo[tid];
    if (__builtin_expect(slow_tiles != 0u || __ballot(redo_tiles != 0u) != 0ull, 0)) {
        const uint8_t *const pay_l = (const uint8_t *)(uintptr_t)uni64(sh.late[0]);
        const uint64_t pay_bytes_l = uni64(sh.late[1]), pay_bits_l = pay_bytes_l * 8ull;
        uint8_t *const out_l = (uint8_t *)(uintptr_t)uni64(sh.late[2]);
        const uint64_t *const told_l = (const uint64_t *)(uintptr_t)uni64(sh.late[3]);
#pragma unroll 1
        for (uint32_t ti = 0; ti * WAVES + (uint32_t)wave < ntiles; ti++) {
            const bool whole = ((slow_tiles >> ti) & 1u) != 0u;
            const bool mine = whole || ((redo_tiles >> ti) & 1u) != 0u;
            if (!__ballot(mine)) continue;
            q = (uint32_t)wave + ti * WAVES;
            const uint32_t g = q * 64u + (uint32_t)lane;
            const uint32_t my0 = g * DSUB_SPL;
            uint32_t nsym = 0, gb = 0;
            if (g < ngrp) {
                nsym = dmin<uint32_t>(DSUB_SPL, nchunk - my0);
                gb = dmin<uint32_t>((uint32_t)grp[g], DSUB_MAX_GROUP_BITS);
            }
            const uint32_t incl = wave_incl_scan_u32(gb);
            const uint64_t tfirst = uni64(told_l[q]);
            /* (a tile whose told start or bits lie outside the payload was counted out by its set-up: ok is false) */
            if (tfirst <= pay_bits_l && (uint64_t)wave_lane_u32(incl, 63) <= pay_bits_l - tfirst) {
                if (!dsub_tile_slow<THREADS>(sh, top, pay_l, pay_bytes_l, tfirst, incl - gb, incl, nsym, mine, out_l + my0)) ok = false;
            }
        }
    }
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
__global__ __launch_bounds__(THREADS, DSUB_WAVES_PER_SIMD) 
void decode_sub_kernel(
    const uint8_t *__restrict__ stream, uint64_t stream_len, const uint64_t *__restrict__ offsets,
    const HufDecodeMeta *__restrict__ dmeta, uint64_t *__restrict__ out_offsets, TwoLevel lens,
    uint8_t *__restrict__ out, uint64_t out_cap, int32_t *__restrict__ status,
    unsigned long long *__restrict__ result, HufSubIndex sub, uint64_t blocksize, uint32_t cpb, DecFixList fix)
{
    __shared__ DsubShared<THREADS> sh;
    static_assert(DSUB_CHUNK_SYMS % (THREADS * DSUB_SPL) == 0 && DSUB_CHUNK_SYMS % HUF_SUB_TILE == 0, "chunks are whole tiles");
    const int tid = (int)threadIdx.x;
    const uint64_t blk = blockIdx.x / cpb;
    const uint32_t c = (uint32_t)(blockIdx.x % cpb);
    unsigned long long kt = DPROF_T();
    /* Everything about the block is the same in all lanes: kept in SGPRs (held in VGPRs these values pushed the
     * tile loop's state out to scratch), and everything is REQUESTED before the first of it is looked at: five
     * scalar loads, one wait (one after the other they were a sixth of a workgroup's life). */
    HufDecodeMeta m = dmeta[blk];
    const uint64_t out_group = lens.gprefix[blk / SCAN_GROUP], out_local = lens.local[blk];
    const uint64_t off0 = offsets[blk], off1 = offsets[blk + 1];
    pin_uniform(m.block_len); pin_uniform(out_group); pin_uniform(out_local); pin_uniform(off0); pin_uniform(off1);
    m.block_len = uni64(m.block_len);
    m.tree_len = (int16_t)uni32((uint32_t)(uint16_t)m.tree_len);
    m.leaf = (int16_t)uni32((uint32_t)(uint16_t)m.leaf);
    m.status = (int32_t)uni32((uint32_t)m.status);
    const uint64_t obase = uni64(out_group + out_local);
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
    const uint64_t o0 = uni64(off0);
    const uint64_t o1 = dmin<uint64_t>(uni64(off1), stream_len);
    const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
    const uint8_t *tree = stream + o0 + HUF_HEADER_FIXED;
    const uint8_t *pay = tree + 2 * (int)m.tree_len;
    bool good;
    if (m.leaf >= 0) {
        /* one 0 bit per symbol: the chunk's bits start at payload bit sym0 (a multiple of 8) */
        uint64_t eb = 0, produced = 0;
        good = (sym0 >> 3) <= pay_bytes &&
               decode_single_leaf<THREADS, true>(sh, (uint32_t)m.leaf, pay + (sym0 >> 3), sym1 - sym0,
                                                 pay_bytes - (sym0 >> 3), out + obase + sym0, &eb, &produced) == HUFE_OK;
    } else {
        DPROF_ADD(2, kt);
        good = decode_payload_sub<THREADS>(sh, tree, m.tree_len, sub.lens + blk * HUF_NSYM, pay, pay_bytes,
                                           stream_len - (uint64_t)(pay - stream), sym0, sym1, sym1 < m.block_len,
                                           sub.tile_bits + blk * sub.tpb + sym0 / HUF_SUB_TILE,
                                           sub.group_bits + blk * sub.gpb + sym0 / DSUB_SPL, out + obase);
    }
    DPROF_ADD(11, kt);              /* (the workgroup's life, thread 0) */
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
