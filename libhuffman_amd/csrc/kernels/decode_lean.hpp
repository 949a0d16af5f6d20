/* decode_lean.hpp - decode_lean_kernel: indexed decode WITHOUT the encoder's sub-index (what huf_decode() and
   streams written by the reference get; src/decoder.c:34-96), round 4 form: every symbol is decoded ONCE.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "decode.hpp"
#include "decode_sub.hpp"

namespace hufgpu {

/* ======================================================================================
 * decode_fast_kernel (decode_fast.hpp) finds the codeword starts by decoding every share of the payload from
 * its own first bit, decodes it again from where the left neighbour ended (a decoder that starts anywhere
 * falls into step after a few codewords, but nearly never ON the share's first bit), and a third time to
 * write: 62 vector instructions per symbol.  Here the speculation is moved in front of the share:
 *
 *   - a lane first walks a RUN-IN of LEAN_RUNIN bits that belong to its left neighbour; when it reaches its
 *     own share it stands on a codeword start of the real track in 97-99 % of all cases (tools/sim/sim_sync.py:
 *     zipf255 97.2 %, log text 98.8 %, uniform bytes 91-92 % at 96 bits).  No counting, no symbols: 9
 *     instructions per codeword of the run-in;
 *   - from there it decodes its share ONCE, four symbols to a 32-bit register (the loop of decode_sub_kernel:
 *     one window read and two table look-ups per two symbols, no branch but the wave's early exit), and KEEPS
 *     them: up to 4 x LEAN_W symbols in LEAN_W registers.  Only whole groups of four are tracked (start of the
 *     last group that began inside the share); that group is then decoded once more, symbol by symbol, to find
 *     the first codeword start at or behind the share's end and the symbols in front of it;
 *   - a lane is RIGHT if its start is its left neighbour's end (lane 0 starts at the segment's true first bit:
 *     induction over the lanes, as in decode_fast).  The few lanes that are not - the run-in had not fallen into
 *     step, more symbols than registers, a code the hot loop does not take - are compacted into the first lanes
 *     of the workgroup and walked again from the right start, step by step, until every start is its neighbour's
 *     end (a wave with ONE such lane would cost what a wave of 64 costs; 15 of 512 compacted cost one wave);
 *   - counts are prefix-summed, right lanes store their registers (unaligned 32-bit stores), the compacted
 *     lanes decode once more into HBM.
 *   - shares are not fixed in size: the payload left is cut into 512 equal shares of about 32 symbols per
 *     segment (the block index gives the payload's size), so the last segment of a block is as full as the
 *     first and no lane decodes what follows the block.
 *
 * The tables are built from the serialized tree alone, in parallel but for one chain: in preorder a leaf's
 * depth is (left turns on its path) + (right turns); the left turns are a prefix sum over the tree entries
 * (+1 node, -1 marker), the right turns are the one bits of the leaf's code, and the code of leaf k+1 is the
 * code of leaf k plus 2^-depth(k).  One wave walks that chain with scalar instructions (four per leaf); the
 * claimed depths are then CHECKED against the tree exactly as dsub_fast_tables checks the sub-index's.
 *
 * Verified, not proven: whatever this kernel cannot vouch for - an unusual tree, starts that do not settle in
 * LEAN_MAX_ROUNDS rounds, a walk out of the tree or past the payload on the final track - is left to
 * decode_fast_list_kernel and, behind that, to the exact decoder and its error codes.
 * ==================================================================================== */
#ifndef LEAN_W
#define LEAN_W 12u                                  /* registers of packed symbols per lane */
#endif
#ifndef LEAN_TARGET_SYMS
#define LEAN_TARGET_SYMS 34u                        /* symbols per lane and segment aimed at (4 x LEAN_W hold them with room for the spread) */
#endif
#ifndef LEAN_RUNIN_CW2
#define LEAN_RUNIN_CW2 27u                          /* run-in = this many HALF codewords of the block's average length (13.5 codewords) ... */
#endif
#define LEAN_RUNIN_MIN 64u                          /* ... but at least / at most so many bits */
#define LEAN_RUNIN_MAX 160u
#define LEAN_SB_MAX 288u                            /* bits per share at most (the stage) */
#define LEAN_SB_MIN (LEAN_RUNIN_MAX + 32u)          /* a lane's run-in stays behind the segment's first bit */
#define LEAN_SLACK_WORDS 48u                        /* staged behind the last share: lanes that finished early keep decoding until their wave is done */
#define LEAN_MAX_ROUNDS 6
#define LEAN_FIX_LANES 128u                         /* lanes one round can decode again (their symbols pass through LDS) */
#define LEAN_WALK_LANES 8u                          /* lanes of a segment the step-by-step walk may take ... */
#define LEAN_WALK_BYTES 160u                        /* ... and the symbols of one (a share holds at most LEAN_SB_MAX / 2) */
#define LEAN_L2_BITS 6u
#define LEAN_L2_ENTRIES 1024u

/* Table entries (uint16), the sub-index path's layout (decode_sub.hpp) with a moving `bad`:
 *   leaf      byte << 8 | len                               len = 1..12 (first table), 13..18 (second level)
 *   level 2   (offset / 4) << 8 | (bits / 2 - 1) << 6 | 0x20  the second-level table of a 12-bit prefix
 *   long      0xFE80                                          a code the tables do not hold (binary search over the codes)
 *   skip      0x0080 | n                                      the window starts with n one bits: no codeword starts on any of them
 *                                                             (every code starts with 0: the wrap root has no right child, src/tree.c:410-413)
 * Bit 7 or bit 5 set: not a codeword the hot loop takes; the low five bits are what a walk moves on by. */
#define LNE_L2 0x20u
#define LNE_ODD 0x80u
#define LNE_LONG 0xFE80u
#define LNE_SPECIAL (LNE_L2 | LNE_ODD)

#ifdef DEC_PHASE_PROF
#define LPROF_CNT(slot, v) atomicAdd(&g_dec_prof[slot], (unsigned long long)(v))
#define LEAN_FAIL(r) do { if (threadIdx.x == 0) atomicAdd(&g_lean_fail[r], 1ull); return false; } while (0)
#else
#define LEAN_FAIL(r) return false
#define LPROF_CNT(slot, v) do { } while (0)
#endif

#ifdef LEAN_FORCE_SLOW_STAGE      /* (diagnostic builds: every segment staged word by word) */
#define LEAN_DEBUG_SLOW_STAGE quick = false;
#else
#define LEAN_DEBUG_SLOW_STAGE
#endif

template <int THREADS>
struct LeanShared {
    static constexpr uint32_t STAGE_WORDS = (uint32_t)THREADS * (LEAN_SB_MAX / 32u) + LEAN_SLACK_WORDS + 8u;
    static constexpr uint32_t SLOT_WORDS = LEAN_W;
    uint16_t lut[1 << DEC_LUT_BITS];
    uint16_t l2[LEAN_L2_ENTRIES];
    uint32_t code[HUF_NSYM];             /* left-aligned codes of the leaves in preorder (= ascending) */
    uint8_t len[HUF_NSYM];
    uint8_t sym[HUF_NSYM];
    __attribute__((aligned(16))) uint32_t stage[STAGE_WORDS];   /* REVERSED: payload word g of the segment at stage[STAGE_WORDS - 1 - g] */
    uint32_t slots[LEAN_FIX_LANES * SLOT_WORDS];                /* the symbols of the lanes decoded again, on their way to the owners */
    /* (stage and slots together are the segment's OUTPUT once everything is decoded: bytes in place, flushed 16 at a time) */
    uint32_t E[THREADS];                 /* where lane i's track ends: the first codeword start at or behind its share's end */
    uint32_t S2[THREADS];                /* lanes decoded again: flags << 28 | start << 8 | count */
    uint16_t list[THREADS];
    uint8_t wid[THREADS];                /* walked lanes: which of wbuf's rows (0xff: none yet) */
    uint8_t wbuf[LEAN_WALK_LANES * LEAN_WALK_BYTES];
    uint32_t part[THREADS / 64];
    uint32_t cpart[3 * (THREADS / 64)];
    uint32_t nfix, nwalk;
    uint32_t fastk, l2n;
    uint32_t firstone;                   /* decode_single_leaf */
};

typedef const __attribute__((address_space(3))) uint32_t *lean_lds_words;
typedef const __attribute__((address_space(3))) uint16_t *lean_lds_halves;

/* the 32 payload bits at the position of R (R = r_origin - position; see decode_sub.hpp: with the words in reversed
 * order (R >> 3) & ~3 is the LDS address of the pair's second word and R's low five bits are the alignbit amount) */
__device__ __forceinline__ uint32_t lean_window(uint32_t R)
{
    lean_lds_words wp = (lean_lds_words)(uintptr_t)((R >> 3) & ~3u);
    return __builtin_amdgcn_alignbit(wp[1], wp[0], R);
}
__device__ __forceinline__ uint32_t lean_lut(uint32_t lut_addr, uint32_t d)
{
    return *(lean_lds_halves)(uintptr_t)(lut_addr + ((d >> 19) & 0x1ffeu));
}
template <int THREADS>
__device__ __forceinline__ uint32_t lean_l2(const LeanShared<THREADS> &sh, uint32_t e, uint32_t bits32)
{
    const uint32_t nb = ((e >> 5) & 6u) + 2u;
    return sh.l2[((e >> 8) << 2) + ((bits32 << DEC_LUT_BITS) >> (32u - nb))];
}

/* ======================================================================================
 * tables from the serialized tree (see the head of the file)
 * ==================================================================================== */
struct LeanTreeWords {
    uint32_t d0, d1, d2, mis;
};
template <int THREADS>
__device__ __forceinline__ LeanTreeWords lean_tree_request(const uint8_t *tree, int tree_len)
{
    LeanTreeWords w;
    const int tid = (int)threadIdx.x;
    const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
    w.mis = (uint32_t)(a & 3u);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a - w.mis);
    const uint32_t nbytes = w.mis + 2u * (uint32_t)(tree_len > 0 ? tree_len : 0);      /* bytes from q[0] to the tree's end */
    const uint32_t t4 = 4u * (uint32_t)tid;
    w.d0 = (t4 < nbytes) ? q[tid] : 0u;
    w.d1 = (t4 + 4u < nbytes) ? q[tid + 1] : 0u;
    w.d2 = (t4 + 8u < nbytes) ? q[tid + 2] : 0u;
    return w;
}

template <int THREADS>
__device__ bool lean_tables(LeanShared<THREADS> &sh, int tree_len, const LeanTreeWords &tw)
{
    constexpr int WAVES = THREADS / 64;
    static_assert(THREADS * 2 >= HUF_TREE_MAX - 1 && THREADS >= 256 && WAVES <= 8, "two entries per thread");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t K = (uint32_t)(tree_len - 1) >> 2;
    if (tid == 0) { sh.fastk = 0; sh.l2n = 0; }
    if (tree_len < 9 || tree_len > HUF_TREE_MAX || ((tree_len - 1) & 3) != 0) return false;     /* uniform */
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(sh.stage);                   /* [256] entry index of the k-th leaf */
    uint8_t *s_left = reinterpret_cast<uint8_t *>(sh.stage + 128);              /* [256] left turns on the way to the k-th leaf */
    uint32_t *s_cpart = sh.cpart;
    bool ok = true;
    /* ---- shape: leaves, nodes, markers.  Thread t looks at entries 2t .. 2t+3 (three aligned dwords, shifted
     *      by the tree's byte misalignment; entries at or past tree_len read -1).  Left turns on the way to entry i
     *      = nodes in front of it - markers in front of it (a node opens a left subtree, the marker that ends the
     *      subtree closes it). ---- */
    bool l0, l1;
    int e0, e1;
    {
        const uint32_t mis = tw.mis, d0 = tw.d0, d1 = tw.d1, d2 = tw.d2;
        if (tid < 64) reinterpret_cast<uint32_t *>(s_left)[tid] = 0x3f3f3f3fu;
        const uint32_t two01 = mis ? __builtin_amdgcn_alignbit(d1, d0, 8u * mis) : d0;
        const uint32_t two23 = mis ? __builtin_amdgcn_alignbit(d2, d1, 8u * mis) : d1;
        const int i0 = 2 * tid;
        e0 = (i0 < tree_len) ? (int)(int16_t)(two01 & 0xffffu) : -1;
        e1 = (i0 + 1 < tree_len) ? (int)(int16_t)(two01 >> 16) : -1;
        const int e2 = (i0 + 2 < tree_len) ? (int)(int16_t)(two23 & 0xffffu) : -1;
        const int e3 = (i0 + 3 < tree_len) ? (int)(int16_t)(two23 >> 16) : -1;
        const bool n0 = e0 != -1, n1 = e1 != -1;
        l0 = n0 && e1 == -1 && e2 == -1 && i0 + 2 < tree_len;
        l1 = n1 && e2 == -1 && e3 == -1 && i0 + 3 < tree_len;
        if (tid == 0 && !n0) ok = false;                                        /* the root */
        if ((i0 == tree_len - 1 && n0) || (i0 + 1 == tree_len - 1 && n1)) ok = false;      /* the last entry is a marker */
        if (tid == THREADS - 1 && i0 + 2 == tree_len - 1 && e2 != -1) ok = false;          /* (entry 1024 has no thread of its own) */
        const uint32_t mine = (uint32_t)l0 + (uint32_t)l1 + (((uint32_t)n0 + (uint32_t)n1) << 16);
        const uint32_t inc = wave_incl_scan_u32(mine);
        if (lane == 63) sh.part[wave] = inc;
        __syncthreads();                                                        /* (also: s_left is preset) */
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const uint32_t x = sh.part[i];
            if (i < wave) base += x;
            tot += x;
        }
        if ((tot & 0xffffu) != K || (tot >> 16) != 2u * K) ok = false;
        const uint32_t before = base + inc - mine;
        uint32_t k = before & 0xffffu;
        const uint32_t nb = before >> 16;                                       /* nodes in front of entry i0 */
        if (l0 && k < 256u) {
            s_pos[k] = (uint16_t)i0;
            sh.sym[k] = (uint8_t)e0;
            s_left[k] = (uint8_t)dmin<uint32_t>(2u * nb - (uint32_t)i0, 63u);
            k++;
        }
        if (l1 && k < 256u) {
            s_pos[k] = (uint16_t)(i0 + 1);
            sh.sym[k] = (uint8_t)e1;
            s_left[k] = (uint8_t)dmin<uint32_t>(2u * (nb + (uint32_t)n0) - (uint32_t)(i0 + 1), 63u);
        }
    }
    __syncthreads();
    /* ---- the one chain: depth(k) = left(k) + ones(code(k)), code(k+1) = code(k) + 2^-depth(k).  Wave 0, every value
     *      the same in all lanes: scalar instructions; lane l holds the left turns of leaves 4l .. 4l+3 and takes
     *      their depths. ---- */
    if (wave == 0) {
        /* (what the chain waits for is one scalar instruction after the other: count the ones, subtract from 32 - left,
         *  shift, add - four per leaf; the depths are taken out of the chain as 32 - depth and put right afterwards) */
        const uint32_t vl4 = 0xa0a0a0a0u - reinterpret_cast<const uint32_t *>(s_left)[lane];     /* (32 - left) + 128 per byte (left <= 63: no borrow between the bytes) */
        uint32_t vd = 0;
        uint32_t code = 0;
        const uint32_t nl = uni32((K + 3u) >> 2);
#pragma unroll 1
        for (uint32_t l = 0; l < nl; l++) {
            const uint32_t four = wave_lane_u32(vl4, l);
            uint32_t pack = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t lneg = ((four >> (8u * j)) & 0xffu) - 128u;          /* 32 - left (may be negative) */
                const uint32_t sh = lneg - (uint32_t)__builtin_popcount(code);     /* 32 - depth */
                code += 1u << (sh & 31u);
                pack |= (sh & 0xffu) << (8u * j);
            }
            vd = ((uint32_t)lane == l) ? pack : vd;
        }
        /* depth = 32 - sh, valid for sh in 0..30 (anything else becomes 33: refused below) */
        uint32_t dd = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t shb = (vd >> (8u * j)) & 0xffu;
            dd |= (shb <= 30u ? 32u - shb : 33u) << (8u * j);
        }
        reinterpret_cast<uint32_t *>(sh.len)[lane] = dd;
    }
    __syncthreads();
    /* ---- claimed depths -> codes; they must fill the left half of the code space exactly.  A code's share of
     *      the 32-bit code space is 2^(32 - d) <= 2^30: scanned as two 16-bit halves (DPP, no 64-bit shuffles) ---- */
    uint32_t d = 2;
    bool anylong;
    {
        uint32_t whi = 0, wlo = 0;
        if ((uint32_t)tid < K) {
            d = sh.len[tid];
            if (d < 2u || d > 32u) { ok = false; d = 2; }
            else if (d >= 16u) wlo = 1u << (32u - d);                           /* <= 2^16 */
            else whi = 1u << (16u - d);                                         /* 2^(32 - d) >> 16 */
        }
        const uint32_t ihi = wave_incl_scan_u32(whi), ilo = wave_incl_scan_u32(wlo);
        const unsigned long long lg = __ballot(d > (uint32_t)DEC_LUT_BITS);
        if (lane == 63) {
            s_cpart[wave] = ihi;
            s_cpart[WAVES + wave] = ilo;
            s_cpart[2 * WAVES + wave] = lg != 0ull;
        }
        __syncthreads();
        uint64_t base = 0, tot = 0;
        uint32_t lf = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const uint64_t x = ((uint64_t)s_cpart[i] << 16) + s_cpart[WAVES + i];
            if (i < wave) base += x;
            tot += x;
            lf |= s_cpart[2 * WAVES + i];
        }
        anylong = uni32(lf) != 0u;
        if (tot != (1ull << 31)) ok = false;
        if ((uint32_t)tid < K) sh.code[tid] = (uint32_t)(base + (((uint64_t)(ihi - whi)) << 16) + (ilo - wlo));
    }
    __syncthreads();
    /* ---- the entry positions the depths imply are the stream's (decode_sub.hpp: leaf 0 is entry d_0; between leaf k
     *      and leaf k+1 lie 3 + d_(k+1) - (d_k - t_k) entries, t_k = trailing one bits of leaf k's code; the last
     *      leaf is followed by its two markers and the root's) ---- */
    if ((uint32_t)tid < K) {
        const uint32_t k = (uint32_t)tid;
        const uint32_t bits = sh.code[k] >> (32u - d);                          /* the d code bits */
        const uint32_t t = (uint32_t)__builtin_ctz(~bits);                      /* trailing ones (< d: codes start with 0) */
        const uint32_t pos = s_pos[k];
        if (k == 0 && pos != d) ok = false;
        if (k + 1 < K) {
            const uint32_t dn = sh.len[k + 1];
            if (dn + t < d || (uint32_t)s_pos[k + 1] != pos + 3u + (dn + t - d)) ok = false;
        } else if (pos + 4u != (uint32_t)tree_len) ok = false;
    }
    /* ---- the table: eight consecutive entries per thread, one 16-byte store ---- */
    {
        static_assert((1 << DEC_LUT_BITS) == THREADS * 8, "eight entries per thread");
        const uint32_t *code = sh.code;
        const uint32_t x0 = (uint32_t)tid * 8u;
        uint32_t k = dsub_leaf_of(code, K, x0 << (32 - DEC_LUT_BITS));
        uint32_t e[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t idx = x0 + j;
            const uint32_t v = idx << (32 - DEC_LUT_BITS);
            if (v >> 31) {
                /* the first bit leaves the tree, and so does every one bit that follows it at once */
                e[j] = LNE_ODD | dmin<uint32_t>((uint32_t)__clz((int)~v), (uint32_t)DEC_LUT_BITS);
            } else {
                while (k + 1u < K && code[k + 1u] <= v) k++;
                const uint32_t dk = sh.len[k];
                e[j] = (dk <= (uint32_t)DEC_LUT_BITS) ? (((uint32_t)sh.sym[k] << 8) | dk) : (uint32_t)LNE_LONG;
            }
        }
        *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
    /* ---- second level: the subtree below a 12-bit prefix whose codes are at most LEAN_L2_BITS longer gets a table
     *      of its own (2, 4 or 6 more bits).  Codes beyond that keep their `long` entry. ---- */
    if (anylong) {
        __syncthreads();                                 /* the first table is written */
        uint16_t *l2 = sh.l2;
        const uint32_t *code = sh.code;
        const uint8_t *len = sh.len;
        uint32_t size = 0, nb = 0, run_end = 0;
        const uint32_t k = (uint32_t)tid;
        const uint32_t P = (k < K) ? (code[k] >> (32 - DEC_LUT_BITS)) : 0u;
        if (k < K && d > (uint32_t)DEC_LUT_BITS &&
            (k == 0 || len[k - 1] <= DEC_LUT_BITS || (code[k - 1] >> (32 - DEC_LUT_BITS)) != P)) {
            /* first leaf below its prefix: the leaves below one prefix are neighbours (preorder = code order) */
            uint32_t maxd = d, mind = d, jn = k + 1u;
            while (jn < K && jn - k <= (1u << LEAN_L2_BITS) && (code[jn] >> (32 - DEC_LUT_BITS)) == P) {
                maxd = dmax<uint32_t>(maxd, len[jn]);
                mind = dmin<uint32_t>(mind, len[jn]);
                jn++;
            }
            const bool closed = !(jn < K && (code[jn] >> (32 - DEC_LUT_BITS)) == P);
            if (closed && maxd <= (uint32_t)DEC_LUT_BITS + LEAN_L2_BITS && mind > (uint32_t)DEC_LUT_BITS) {
                nb = (maxd - DEC_LUT_BITS + 1u) & ~1u;
                size = 1u << nb;
                run_end = jn;
            }
        }
        uint32_t total;
        const uint32_t off = block_excl_scan_u32<THREADS>(size, sh.part, total);          /* (sizes are multiples of 4: so are the offsets) */
        if (size && off + size <= LEAN_L2_ENTRIES) {
            for (uint32_t jn = k; jn < run_end; jn++) {
                const uint32_t dj = len[jn];
                const uint32_t first = (code[jn] >> (32 - DEC_LUT_BITS - nb)) & (size - 1u);
                const uint32_t count = 1u << (DEC_LUT_BITS + nb - dj);
                const uint16_t entry = (uint16_t)(((uint32_t)sh.sym[jn] << 8) | dj);
                for (uint32_t i = 0; i < count && first + i < size; i++) l2[off + first + i] = entry;
            }
            sh.lut[P] = (uint16_t)(((off >> 2) << 8) | ((nb / 2u - 1u) << 6) | LNE_L2);
        }
        if (tid == 0) sh.l2n = dmin<uint32_t>(total, LEAN_L2_ENTRIES);
    }
    if (tid == 0) sh.fastk = K;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

/* ======================================================================================
 * one codeword, step by step (the compacted lanes; the last group of a share).  Everything the tables hold.
 * ==================================================================================== */
struct LeanStep {
    uint32_t len;        /* bits to move on by (0: nothing can be said - cannot happen with lean_tables' tables) */
    uint32_t sym;
    bool odd;            /* not a codeword: a run of one bits skipped */
};
template <int THREADS>
__device__ __forceinline__ LeanStep lean_step(const LeanShared<THREADS> &sh, uint32_t lut_addr, uint32_t R)
{
    const uint32_t d = lean_window(R);
    uint32_t e = lean_lut(lut_addr, d);
    LeanStep s;
    s.odd = false;
    if (__builtin_expect(__ballot((e & LNE_SPECIAL) != 0u) != 0ull, 0)) {
        if (e & LNE_L2) {
            e = lean_l2<THREADS>(sh, e, d);
        } else if (e == LNE_LONG) {
            const uint32_t k = dsub_leaf_of(sh.code, sh.fastk, d);
            s.len = sh.len[k];
            s.sym = sh.sym[k];
            return s;
        } else if (e & LNE_ODD) {
            s.odd = true;
        }
    }
    s.len = e & 31u;
    s.sym = e >> 8;
    return s;
}

/* A share walked step by step from `start`: everything the tables hold, and it knows where the payload ends.
 * Counts the codewords that start in front of `hi` and end at or in front of `lim`, and notes their symbols in `row`
 * (LEAN_WALK_BYTES of LDS).  Returns flags << 28 | end << 8 | count (count <= 255: codes have at least two bits and a
 * share at most LEAN_SB_MAX). */
#define LEAN_F_ODD 1u            /* the track met bits that are no codeword */
#define LEAN_F_EXH 2u            /* a codeword needs bits past the payload */
#define LEAN_F_WALK 4u           /* the symbols are in a row of wbuf, not in registers */
template <int THREADS>
__device__ __noinline__ uint32_t lean_walk(const LeanShared<THREADS> &sh, uint32_t lut_addr, uint32_t r_origin, bool mine,
                                           uint32_t start, uint32_t hi, uint32_t lim, uint8_t *row)
{
    uint32_t R = r_origin - start;
    const uint32_t Rhi = r_origin - hi;
    uint32_t c = 0, flags = 0;
    bool go = mine && start < hi;
    while (__any(go)) {
        const LeanStep s = lean_step<THREADS>(sh, lut_addr, R);
        if (go) {
            const uint32_t pos = r_origin - R;
            if (s.len == 0u || pos + s.len > lim) {
                flags |= LEAN_F_EXH;
                go = false;
            } else {
                R -= s.len;
                if (s.odd) flags |= LEAN_F_ODD;
                else {
                    if (c < LEAN_WALK_BYTES) row[c] = (uint8_t)s.sym;
                    c++;
                }
                if (R <= Rhi || c >= 255u) go = false;
            }
        }
    }
    return (flags << 28) | ((r_origin - R) << 8) | c;
}

/* A share decoded once, four symbols to a word, from the codeword start `start` to the first codeword start at or
 * behind `hi`.  The words go to w[] (TO_LDS = false: registers; the loop is rolled, k is the same in all lanes and w[k]
 * is written through the register index) or to slot[] (LDS).  What is tracked is the start of the last group that began
 * inside the share: that group is decoded once more, symbol by symbol, for the end and the count.  The last word holds
 * up to three symbols of the NEXT share; they are that share's first symbols, whoever stores them.
 * flags: LEAN_F_ODD = a look-up was not a codeword the loop takes (a long code, bits that are no codeword), or the
 * share holds more symbols than 4 x LEAN_W: end and count are not to be used. */
template <int THREADS, bool L2, bool TO_LDS>
__device__ __forceinline__ void lean_share(const LeanShared<THREADS> &sh, uint32_t lut_addr, uint32_t r_origin, bool mine,
                                           uint32_t start, uint32_t hi, uint32_t (&w)[LEAN_W], uint32_t *slot,
                                           uint32_t &end_out, uint32_t &cnt_out, uint32_t &flags_out)
{
    const uint32_t Rhi = r_origin - hi;
    uint32_t R = mine ? r_origin - start : Rhi;
    uint32_t Rg = R, ng = 0, special = 0;
#define LEAN_WINDOW(PAIR)                                                                                     \
    {                                                                                                         \
        const uint32_t d1_ = lean_window(R);                                                                  \
        uint32_t e1_ = lean_lut(lut_addr, d1_);                                                               \
        if (L2 && __ballot((e1_ & LNE_L2) != 0u)) {                                                           \
            if (e1_ & LNE_L2) e1_ = lean_l2<THREADS>(sh, e1_, d1_);                                           \
        }                                                                                                     \
        const uint32_t d2_ = d1_ << (e1_ & 31u);                                                              \
        uint32_t e2_ = lean_lut(lut_addr, d2_);                                                               \
        if (L2 && __ballot((e2_ & LNE_L2) != 0u)) {                                                           \
            if ((e2_ & LNE_L2) && (e1_ & 31u) + DEC_LUT_BITS + LEAN_L2_BITS <= 32u)                           \
                e2_ = lean_l2<THREADS>(sh, e2_, d2_);                                                         \
        }                                                                                                     \
        seen |= e1_ | e2_;                                                                                    \
        R -= (e1_ + e2_) & 0xffu;                                                                             \
        PAIR = __builtin_amdgcn_perm(e2_, e1_, 0x0c0c0501u);                                                  \
    }
#pragma unroll 1
    for (uint32_t k = 0; k < LEAN_W; k++) {
        const bool act = R > Rhi;
        if (!__any(act)) break;
        Rg = act ? R : Rg;
        ng += act ? 1u : 0u;
        uint32_t p01, p23, seen = 0;
        LEAN_WINDOW(p01)
        LEAN_WINDOW(p23)
        special |= act ? seen : 0u;                                    /* (a lane that is done decodes on, whatever lies there, until its wave is) */
        const uint32_t word = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
        if (TO_LDS) slot[k] = word;
        else w[k] = word;
    }
#undef LEAN_WINDOW
    uint32_t flags = 0;
    if (mine && R > Rhi) flags |= LEAN_F_ODD;                          /* more symbols than registers */
    if ((special & LNE_SPECIAL) != 0u) flags |= LEAN_F_ODD;
    uint32_t cnt = 0, end = dmax<uint32_t>(start, hi);
    if (__any(mine && ng != 0u)) {
        uint32_t Rs = Rg, c = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const bool in = mine && ng != 0u && Rs > Rhi;
            const uint32_t d1 = lean_window(Rs);
            uint32_t e1 = lean_lut(lut_addr, d1);
            if (L2 && __ballot((e1 & LNE_L2) != 0u)) {
                if (e1 & LNE_L2) e1 = lean_l2<THREADS>(sh, e1, d1);
            }
            if (in) {
                Rs -= e1 & 31u;
                c++;
            }
        }
        if (mine && ng != 0u) {
            cnt = 4u * (ng - 1u) + c;
            end = r_origin - Rs;
        }
    }
    end_out = end;
    cnt_out = cnt;
    flags_out = flags;
}

/* ======================================================================================
 * a block's payload
 * ==================================================================================== */
template <int THREADS>
__device__ __forceinline__ bool decode_payload_lean(LeanShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable,
                                                    uint64_t block_len, uint8_t *gout)
{
    typedef LeanShared<THREADS> L;
    constexpr uint32_t T = (uint32_t)THREADS;
    constexpr uint32_t OUT_BYTES = (L::STAGE_WORDS + LEAN_FIX_LANES * L::SLOT_WORDS) * 4u;
    static_assert(LEAN_W >= 4 && LEAN_W <= 16, "registers of symbols");
    static_assert(offsetof(L, slots) == offsetof(L, stage) + sizeof(L::stage), "the output image runs on from the stage into the slots");
    static_assert(LEAN_WALK_BYTES >= LEAN_SB_MAX / 2u && LEAN_WALK_LANES < 255u, "a walked share's symbols fit a row");
    typedef uint32_t dwords4 __attribute__((ext_vector_type(4)));
    typedef dwords4 dwords4_a4 __attribute__((aligned(4)));
    typedef const __attribute__((address_space(1))) dwords4_a4 *global_q4;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t *top = sh.stage + (L::STAGE_WORDS - 1u);                  /* staged word g at top[-g] */
    uint8_t *ostage = reinterpret_cast<uint8_t *>(sh.stage);           /* the output image */
    const uint32_t r_origin = 8u * (uint32_t)(uintptr_t)(lean_lds_words)(top - 1) + 32u;      /* R of position 0 */
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(lean_lds_halves)sh.lut;
    const bool use_l2 = uni32(sh.l2n) != 0u;
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uintptr_t pay_a = (uintptr_t)uni64((uint64_t)(uintptr_t)pay);
    const uintptr_t end_a = pay_a + (uintptr_t)readable;
    /* the run-in: so many codewords of the block's average length */
    const uint32_t runin = uni32(dmin<uint32_t>(dmax<uint32_t>((uint32_t)((pay_bits * LEAN_RUNIN_CW2 / 2u) / block_len), LEAN_RUNIN_MIN), LEAN_RUNIN_MAX));
    uint64_t true_start = 0, produced = 0;

    /* A segment's geometry: the payload left in equal shares of about LEAN_TARGET_SYMS symbols; word 0 of the stage is the
     * aligned 32-bit word of memory that holds the segment's first bit.  Its words are requested as soon as it is known
     * (the first segment's here, the others' while the segment before is stored) and wait in registers. */
    uint32_t sb = 0, first = 0, nlive = 0, need_words = 0;
    bool quick = false;
    /* (The words are loaded and staged by the FIRST HALF of the workgroup's waves, the output image is flushed by the
     *  second half: a wave's wait for its loads is then not a wait for the stores issued around them - vmcnt counts a
     *  wave's loads and stores together, in order.) */
    constexpr uint32_t LT = 3u * T / 4u;                               /* loading threads */
    constexpr uint32_t LSTEPS = (L::STAGE_WORDS + 4u * LT - 1u) / (4u * LT);
    const bool loader = tid < LT;
    dwords4 V[LSTEPS];
#define LEAN_GEOMETRY()                                                                                        \
    {                                                                                                         \
        const uint64_t rem_bits_ = pay_bits - true_start, rem_syms_ = block_len - produced;                   \
        if ((rem_bits_ | rem_syms_) >> 31) {                                                                  \
            uint64_t nseg_ = (rem_syms_ + (uint64_t)T * LEAN_TARGET_SYMS - 1u) / ((uint64_t)T * LEAN_TARGET_SYMS); \
            const uint64_t nseg_b_ = (rem_bits_ + (uint64_t)T * LEAN_SB_MAX - 1u) / ((uint64_t)T * LEAN_SB_MAX);  \
            if (nseg_b_ > nseg_) nseg_ = nseg_b_;                                                             \
            sb = (uint32_t)dmin<uint64_t>((rem_bits_ + nseg_ * T - 1u) / (nseg_ * T), LEAN_SB_MAX);           \
            nlive = (uint32_t)dmin<uint64_t>((rem_bits_ + LEAN_SB_MIN - 1u) / LEAN_SB_MIN, T);                \
        } else {                                                                                              \
            const uint32_t rb_ = (uint32_t)rem_bits_, rs_ = (uint32_t)rem_syms_;                              \
            uint32_t nseg_ = (rs_ + T * LEAN_TARGET_SYMS - 1u) / (T * LEAN_TARGET_SYMS);                      \
            const uint32_t nseg_b_ = (rb_ + T * LEAN_SB_MAX - 1u) / (T * LEAN_SB_MAX);                        \
            if (nseg_b_ > nseg_) nseg_ = nseg_b_;                                                             \
            sb = (rb_ + nseg_ * T - 1u) / (nseg_ * T);                                                        \
            nlive = T;                                                                                        \
        }                                                                                                     \
        sb = uni32(dmin<uint32_t>(dmax<uint32_t>(sb, LEAN_SB_MIN), LEAN_SB_MAX));                             \
        if (rem_bits_ < (uint64_t)T * sb) nlive = ((uint32_t)rem_bits_ + sb - 1u) / sb;     /* lanes whose share begins inside the payload */ \
        const uintptr_t a_ = pay_a + (uintptr_t)(true_start >> 3);                                            \
        first = (uint32_t)(true_start & 7u) + 8u * (uint32_t)(a_ & 3u);                                       \
        need_words = ((first + nlive * sb + 31u) >> 5) + LEAN_SLACK_WORDS;                                    \
        const uintptr_t a0_ = a_ & ~(uintptr_t)3;                                                             \
        quick = (uint64_t)a0_ + 4ull * (((uint64_t)need_words + 3u) & ~3ull) <= (uint64_t)end_a;              \
        LEAN_DEBUG_SLOW_STAGE                                                                                 \
        first = uni32(first);                                                                                 \
        nlive = uni32(nlive);                                                                                 \
        need_words = uni32(need_words);                                                                       \
        if (quick && loader) {                                                                                \
            const global_q4 q_ = (global_q4)(uintptr_t)uni64((uint64_t)a0_);                                  \
            uint32_t t_ = tid;                                                                                \
            asm volatile("" : "+v"(t_));             /* (indices computed here, not kept in registers around the loop) */ \
            _Pragma("unroll")                                                                                 \
            for (uint32_t k = 0; k < LSTEPS; k++) {                                                           \
                const uint32_t i4_ = dmin<uint32_t>(t_ + LT * k, (need_words - 1u) >> 2);     /* (every lane loads: no branch) */ \
                V[k] = q_[i4_];                                                                               \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll")                                                                                 \
            for (uint32_t k = 0; k < LSTEPS; k++) V[k] = dwords4{0u, 0u, 0u, 0u};      /* (defined on every path: not kept alive around the loop) */ \
        }                                                                                                     \
    }
    if (true_start >= pay_bits) LEAN_FAIL(1);
    LEAN_GEOMETRY()

    while (produced < block_len) {
        __syncthreads();                                               /* the previous segment's output is flushed */
        unsigned long long pt = DPROF_T();
        if (tid == 0) LPROF_CNT(12, 1);
        if (quick) {
            if (loader) {
                uint32_t t_ = tid;
                asm volatile("" : "+v"(t_));
#pragma unroll
                for (uint32_t k = 0; k < LSTEPS; k++) {
                    const uint32_t i4 = 4u * (t_ + LT * k);
                    if (i4 < need_words)
                        *reinterpret_cast<uint4 *>(top - (i4 + 3u)) =
                            make_uint4(__builtin_bswap32(V[k].w), __builtin_bswap32(V[k].z), __builtin_bswap32(V[k].y), __builtin_bswap32(V[k].x));
                }
            }
        } else {
            /* the stream ends right behind the payload: word by word from the first bit's BYTE, zeros past the end */
            first = (uint32_t)(true_start & 7u);
            for (uint32_t i = tid; i < need_words; i += T) top[-(int32_t)i] = load_be32(pay, (true_start >> 3) + 4ull * i, readable);
        }
        sh.wid[tid] = 0xff;
        if (tid == 0) { sh.nfix = 0; sh.nwalk = 0; }
        __syncthreads();
        DPROF_ADD(1, pt); pt = DPROF_T();
        const uint64_t rem_bits = pay_bits - true_start, rem_syms = block_len - produced;
        const uint32_t pay_rel = (uint32_t)dmin<uint64_t>(rem_bits + first, 0x000ffff0ull);   /* stage position of the payload's end */
        const uint32_t lo = first + tid * sb, hi = lo + sb;
        const bool live = tid < nlive;

        /* ---- run-in: from `runin` bits in front of the share to the first codeword start inside it, on whatever track
         *      that is (lane 0 stands on the segment's true first bit) ---- */
        const uint32_t Rlo = r_origin - lo;
        uint32_t R = (live && tid != 0u) ? Rlo + runin : Rlo;
#ifdef LEAN_ABLATE_RUNIN
        R = Rlo;
#endif
#pragma unroll 1
        for (uint32_t it = 0; it < LEAN_RUNIN_MAX / 4u + 2u; it++) {
            const bool act = R > Rlo;
            if (!__any(act)) break;
            const uint32_t d1 = lean_window(R);
            uint32_t e1 = lean_lut(lut_addr, d1);
            if (use_l2 && __ballot((e1 & LNE_L2) != 0u)) {
                if (e1 & LNE_L2) e1 = lean_l2<THREADS>(sh, e1, d1);
            }
            const uint32_t n1 = e1 & 31u;
            const uint32_t d2 = d1 << n1;
            uint32_t e2 = lean_lut(lut_addr, d2);
            if (use_l2 && __ballot((e2 & LNE_L2) != 0u)) {
                if ((e2 & LNE_L2) && n1 + DEC_LUT_BITS + LEAN_L2_BITS <= 32u) e2 = lean_l2<THREADS>(sh, e2, d2);
            }
            const uint32_t n2 = (e2 & LNE_L2) ? 0u : (e2 & 31u);
            const uint32_t t1 = R - (act ? n1 : 0u);
            R = t1 - ((t1 > Rlo) ? n2 : 0u);
        }
#ifdef LEAN_ABLATE_RUNIN
        R = Rlo;
#endif
        if (R > Rlo) R = Rlo;                                          /* (a `long` code in the run-in: not arrived; the share's first bit then) */
        uint32_t start = r_origin - R;
        DPROF_ADD(2, pt); pt = DPROF_T();

        /* ---- the share, once ---- */
        uint32_t w[LEAN_W];
#pragma unroll
        for (uint32_t k = 0; k < LEAN_W; k++) w[k] = 0;
        uint32_t end, cnt, flags;
#ifdef LEAN_ABLATE_SHARE
        end = hi; cnt = live ? 32u : 0u; flags = 0;
#else
        if (use_l2) lean_share<THREADS, true, false>(sh, lut_addr, r_origin, live, start, hi, w, nullptr, end, cnt, flags);
        else lean_share<THREADS, false, false>(sh, lut_addr, r_origin, live, start, hi, w, nullptr, end, cnt, flags);
#endif
        /* a share that reaches the payload's end is walked step by step (which knows where the payload ends) */
        if (live && hi + 96u > pay_rel) flags |= LEAN_F_ODD;
        if (live) sh.E[tid] = end;
        __syncthreads();
        DPROF_ADD(3, pt); pt = DPROF_T();

        /* ---- right is: my start is my left neighbour's end.  Everyone else: compacted, decoded again from there by the
         *      first lanes of the workgroup (the same loop, the words through LDS; what that loop cannot take: walked),
         *      until it holds for all ---- */
        bool walked = false;
        uint32_t prev = (tid == 0u) ? first : sh.E[tid - 1u];
        bool bad = live && ((flags & LEAN_F_ODD) != 0u || start != prev);
#ifdef LEAN_ABLATE_SETTLE
        bad = false;
#endif
        if (bad) LPROF_CNT(9, 1);
        int rounds = 0;
        bool settled = true;
        int settle_why = 0;
        /* (Only a lane whose LEFT neighbour looks right is decoded again: its start then is final unless something further
         *  left is still wrong.  Taking the end of a neighbour that is itself to be redone would put a lane that WAS right
         *  on a wrong track, and with codes that fall into step slowly that damage runs to the right, a lane per round.) */
        for (;;) {
            const unsigned long long bm = __ballot(bad);
            if (lane == 63u) sh.part[tid >> 6] = (uint32_t)(bm >> 63);
            unsigned long long st_ = DPROF_T();
            if (!__syncthreads_or(bad ? 1 : 0)) { DPROF_ADD(8, st_); break; }
            DPROF_ADD(8, st_); st_ = DPROF_T();
            if (++rounds > LEAN_MAX_ROUNDS) { settled = false; settle_why = 2; break; }
            if (tid == 0) LPROF_CNT(10, 1);
            const bool left_bad = lane ? ((bm >> (lane - 1u)) & 1ull) != 0ull : (tid != 0u && sh.part[(tid >> 6) - 1u] != 0u);
            const bool fixme = bad && !left_bad;
            uint32_t myslot = 0;
            {
                const unsigned long long m = __ballot(fixme);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&sh.nfix, (uint32_t)__popcll(m));
                base = wave_lane_u32(base, 0);
                myslot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (fixme && myslot < LEAN_FIX_LANES) sh.list[myslot] = (uint16_t)tid;
            }
            __syncthreads();
            DPROF_ADD(11, st_); st_ = DPROF_T();
            const uint32_t n = uni32(sh.nfix);
            if (n > LEAN_FIX_LANES) { settled = false; settle_why = 3; break; }        /* (uniform) */
            if (tid < ((n + 63u) & ~63u)) {                            /* (whole waves) */
                const bool mine = tid < n;
                const uint32_t li = mine ? sh.list[tid] : 0u;
                const uint32_t st = (li == 0u) ? first : sh.E[li - 1u];
                const uint32_t hi_li = first + (li + 1u) * sb;
                uint32_t e2, c2, f2;
                uint32_t dummy[LEAN_W];
                uint32_t *slot = sh.slots + tid * L::SLOT_WORDS;
                if (use_l2) lean_share<THREADS, true, true>(sh, lut_addr, r_origin, mine, st, hi_li, dummy, slot, e2, c2, f2);
                else lean_share<THREADS, false, true>(sh, lut_addr, r_origin, mine, st, hi_li, dummy, slot, e2, c2, f2);
                if (mine && hi_li + 96u > pay_rel) f2 |= LEAN_F_ODD;
                /* what the loop cannot take: step by step, the symbols into a row of wbuf (the lane keeps its row) */
                const bool towalk = mine && (f2 & LEAN_F_ODD) != 0u;
                if (__any(towalk)) {
                    uint32_t row = 0xffu;
                    if (towalk) {
                        row = sh.wid[li];
                        if (row == 0xffu) {
                            row = atomicAdd(&sh.nwalk, 1u);
                            if (row < LEAN_WALK_LANES) sh.wid[li] = (uint8_t)row;
                        }
                    }
                    const bool can = towalk && row < LEAN_WALK_LANES;
                    const uint32_t res = lean_walk<THREADS>(sh, lut_addr, r_origin, can, st, hi_li, pay_rel, sh.wbuf + (can ? row : 0u) * LEAN_WALK_BYTES);
                    if (can) {
                        e2 = (res >> 8) & 0xfffffu;
                        c2 = res & 0xffu;
                        f2 = (res >> 28) | LEAN_F_WALK;
                    }
                }
                if (mine) {
                    sh.S2[li] = (f2 << 28) | (st << 8) | c2;
                    sh.E[li] = e2;
                }
            }
            DPROF_ADD(13, st_); st_ = DPROF_T();
            __syncthreads();
            DPROF_ADD(14, st_); st_ = DPROF_T();
            if (tid == 0) sh.nfix = 0;
            if (fixme) {
                const uint32_t v = sh.S2[tid];
                start = (v >> 8) & 0xfffffu;
                cnt = v & 0xffu;
                flags = v >> 28;
                walked = (flags & LEAN_F_WALK) != 0u;
                if (!walked) {
                    const uint32_t *slot = sh.slots + myslot * L::SLOT_WORDS;
#pragma unroll
                    for (uint32_t k = 0; k < LEAN_W; k++) w[k] = (4u * k < cnt) ? slot[k] : 0u;     /* (behind the last group the slot holds whatever it held) */
                }
            }
            prev = (tid == 0u) ? first : sh.E[tid - 1u];
            /* (a lane the loop could not take and the walk had no row for stays bad: the rounds run out) */
            bad = live && (start != prev || ((flags & LEAN_F_ODD) != 0u && !walked));
            DPROF_ADD(15, st_);
        }
        /* (stage and slots have been read for the last time: zeroed now, they are the output image; the barriers of the sums
         *  below stand between this and the lanes' words) */
        for (uint32_t z = 16u * tid; z < OUT_BYTES; z += 16u * T) *reinterpret_cast<uint4 *>(ostage + z) = make_uint4(0u, 0u, 0u, 0u);
        if (!settled) LEAN_FAIL(settle_why);
        if (uni32(sh.nwalk) > LEAN_WALK_LANES) LEAN_FAIL(4);
        DPROF_ADD(4, pt); pt = DPROF_T();

        /* ---- counts -> places ---- */
        if (!live) cnt = 0;
        uint32_t seg_total;
        const uint32_t ex = block_excl_scan_u32<THREADS>(cnt, sh.part, seg_total);
        seg_total = uni32(seg_total);
        const bool last_seg = (uint64_t)seg_total >= rem_syms;
        if (!last_seg && nlive < T) LEAN_FAIL(5);                      /* the payload ends, the block does not: the exact decoder says how */
        const uint32_t take = last_seg ? (uint32_t)rem_syms : seg_total;
        if (take == 0u) LEAN_FAIL(6);
        uint32_t quota = 0;
        if (ex < take) quota = dmin<uint32_t>(take - ex, cnt);
        /* the track that delivers symbols met bits that are no codeword */
        const bool lane_ok = !(quota != 0u && walked && (flags & LEAN_F_ODD) != 0u);
        uint8_t *gseg = gout + produced;
        const uint32_t shift = (uint32_t)((uintptr_t)gseg & 15u);      /* the image is laid out like the destination's 16-byte lines */
        if (shift + take + 8u > OUT_BYTES) LEAN_FAIL(7);               /* (more symbols than the image holds: a segment far denser than the block) */
        const uint32_t last_end = uni32(sh.E[nlive - 1u]);
        if (!__syncthreads_and(lane_ok ? 1 : 0)) LEAN_FAIL(8);         /* (also: every walk has read the stage) */
        DPROF_ADD(5, pt); pt = DPROF_T();
#ifdef DEC_PHASE_PROF
        if (blockIdx.x == gridDim.x - 1 && tid == 0) {           /* (the stream's last block, segment by segment) */
            const uint32_t sidx = (uint32_t)atomicAdd(&g_lean_fail[15], 1ull);
            if (sidx < 8) {
                unsigned long long *d = g_lean_fail + 16 + 8 * sidx;
                d[0] = true_start; d[1] = ((unsigned long long)first << 32) | (quick ? 1u : 0u) | ((unsigned long long)sb << 8); d[2] = ((unsigned long long)nlive << 32) | need_words;
                d[3] = ((unsigned long long)seg_total << 32) | take; d[4] = ((unsigned long long)last_end << 32) | start; d[5] = ((unsigned long long)cnt << 32) | (uint32_t)rounds;
                d[6] = produced; d[7] = ((unsigned long long)(uint32_t)((uintptr_t)pay & 3u) << 32) | pay_rel;
            }
        }
#endif

        /* ---- the symbols into the output image: whole words (the bytes of the last one that are not mine are my right
         *      neighbour's first symbols, the same values from whoever writes them); walked lanes byte by byte ---- */
#ifndef LEAN_ABLATE_IMAGE
        /* ---- the symbols into the output image.  A lane's bytes begin anywhere, and a 32-bit LDS store that is not
         *      aligned costs what 64 one-lane stores cost: the lane shifts its words to the image's word grid instead and
         *      ORs them into the zeroed image (the bytes in front of its first symbol are zero; the bytes behind its last
         *      are its right neighbour's first symbols - the same values from whoever brings them).  Walked lanes: bytes. ---- */
        {
            const uint32_t at = shift + ex;
            const uint32_t sh8 = 8u * (at & 3u);
            uint32_t *img = reinterpret_cast<uint32_t *>(ostage) + (at >> 2);
            const uint32_t span = (at & 3u) + quota;                   /* bytes from the first word's first byte to my last symbol */
            if (!walked && quota != 0u) {
                uint32_t prevw = 0;
#pragma unroll
                for (uint32_t k = 0; k <= LEAN_W; k++) {
                    const uint32_t cur = k < LEAN_W ? w[k] : 0u;
                    /* word k of the grid: the top bytes of my word k - 1 and the low bytes of my word k */
                    const uint32_t v = sh8 ? __builtin_amdgcn_alignbit(cur, prevw, 32u - sh8) : cur;
                    if (4u * k < span) atomicOr(img + k, v);
                    prevw = cur;
                }
            }
            if (__any(walked && quota != 0u)) {
                if (walked && quota != 0u) {
                    const uint8_t *row = sh.wbuf + (uint32_t)sh.wid[tid] * LEAN_WALK_BYTES;
                    uint8_t *o = ostage + at;
                    for (uint32_t c = 0; c < quota; c++) o[c] = row[c];
                }
            }
        }
#endif
        produced += take;
        true_start += (uint64_t)(last_end - first);
        /* the next segment's words are requested before this one's stores */
        const uint32_t out_first = shift, out_total = shift + take;
        if (produced < block_len) {
            if (true_start >= pay_bits) LEAN_FAIL(1);                   /* input exhausted: the exact decoder says how */
            LEAN_GEOMETRY()
        }
        __syncthreads();
        DPROF_ADD(6, pt); pt = DPROF_T();
#ifndef LEAN_ABLATE_FLUSH
        {
            uint8_t *g0 = gseg - shift;                                 /* 16-byte aligned */
            uint32_t t_ = tid;
            asm volatile("" : "+v"(t_));                                /* (computed here: not a value kept, or spilled, around the loop) */
            for (uint32_t c16 = 16u * (t_ - LT); c16 < out_total; c16 += 16u * (T - LT)) {       /* (the storing half: tid - LT wraps for the others) */
                const uint4 v = *reinterpret_cast<const uint4 *>(ostage + c16);
                if (c16 >= out_first && c16 + 16u <= out_total) {
                    *reinterpret_cast<uint4 *>(g0 + c16) = v;
                } else {
                    const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t b = 0; b < 16; b++)
                        if (c16 + b >= out_first && c16 + b < out_total) g0[c16 + b] = (uint8_t)(vw[b >> 2] >> (8u * (b & 3u)));
                }
            }
        }
#endif
        DPROF_ADD(7, pt);
    }
#undef LEAN_GEOMETRY
    return true;
}

/* Work list of blocks this kernel leaves to decode_fast_list_kernel. */
struct LeanTodo {
    uint32_t *count;      /* [1] zeroed by decode_prepare_kernel */
    uint32_t *blocks;     /* [nblocks] */
};

#ifndef LEAN_WAVES_PER_SIMD
#define LEAN_WAVES_PER_SIMD 6
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS, LEAN_WAVES_PER_SIMD) void decode_lean_kernel(
    const uint8_t *__restrict__ stream, uint64_t stream_len, const uint64_t *__restrict__ offsets,
    const HufDecodeMeta *__restrict__ dmeta, uint64_t *__restrict__ out_offsets, TwoLevel lens,
    uint8_t *__restrict__ out, uint64_t out_cap, int32_t *__restrict__ status,
    unsigned long long *__restrict__ result, LeanTodo todo)
{
    __shared__ LeanShared<THREADS> sh;
    const int tid = (int)threadIdx.x;
    const uint64_t blk = blockIdx.x;
    HufDecodeMeta m = dmeta[blk];
    const uint64_t out_group = lens.gprefix[blk / SCAN_GROUP], out_local = lens.local[blk];
    const uint64_t off0 = offsets[blk], off1 = offsets[blk + 1];
    pin_uniform(m.block_len); pin_uniform(out_group); pin_uniform(out_local); pin_uniform(off0); pin_uniform(off1);
    m.block_len = uni64(m.block_len);
    m.tree_len = (int16_t)uni32((uint32_t)(uint16_t)m.tree_len);
    m.leaf = (int16_t)uni32((uint32_t)(uint16_t)m.leaf);
    m.status = (int32_t)uni32((uint32_t)m.status);
    const uint64_t obase = uni64(out_group + out_local);
    if (tid == 0) out_offsets[blk] = obase;
    if (m.status != HUFE_OK || m.block_len == 0) return;             /* header errors were recorded by decode_prepare */
    if (obase + m.block_len > out_cap) {
        if (tid == 0) {
            status[blk] = HUFE_MEMORY;
            atomicMin(&result[2], (unsigned long long)blk);
        }
        return;
    }
    const uint64_t o0 = uni64(off0);
    const uint64_t o1 = dmin<uint64_t>(uni64(off1), stream_len);
    const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
    const uint8_t *tree = stream + o0 + HUF_HEADER_FIXED;
    const uint8_t *pay = tree + 2 * (int)m.tree_len;
    bool good;
    if (m.leaf >= 0) {
        uint64_t eb = 0, produced = 0;
        good = decode_single_leaf<THREADS, true>(sh, (uint32_t)m.leaf, pay, m.block_len, pay_bytes, out + obase, &eb, &produced) == HUFE_OK;
    } else {
        const LeanTreeWords tw = lean_tree_request<THREADS>(tree, m.tree_len);
        unsigned long long kt = DPROF_T();
        const bool tables = lean_tables<THREADS>(sh, m.tree_len, tw);
        DPROF_ADD(0, kt);
#ifdef DEC_PHASE_PROF
        if (!tables && tid == 0) atomicAdd(&g_lean_fail[9], 1ull);
#endif
#ifdef LEAN_ABLATE_PAYLOAD
        good = tables;
        if (false)
#endif
        good = tables &&
               decode_payload_lean<THREADS>(sh, pay, pay_bytes, stream_len - (uint64_t)(pay - stream), m.block_len, out + obase);
    }
#if defined(LEAN_ABLATE_PAYLOAD) || defined(LEAN_ABLATE_SETTLE) || defined(LEAN_ABLATE_RUNIN) || defined(LEAN_ABLATE_IMAGE) || defined(LEAN_ABLATE_FLUSH) || defined(LEAN_ABLATE_SHARE)
    good = true;                      /* (timing experiments: the output is wrong and stays wrong) */
#endif
    if (!good && tid == 0) todo.blocks[atomicAdd(todo.count, 1u)] = (uint32_t)blk;
}

}  // namespace hufgpu
