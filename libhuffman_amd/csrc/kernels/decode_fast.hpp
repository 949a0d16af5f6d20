/* decode_fast.hpp - decode_fast_kernel: indexed decode WITHOUT the encoder's sub-index (what huf_decode() and
   streams written by the reference get; src/decoder.c:34-96), lean form.
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
 * The self-synchronising decoder of decode.hpp is exact for ANY stream, and pays for it: its count pass and
 * its re-synchronisation rounds keep, per lane and 32-bit word, where the lane's track first entered the word -
 * 54 instructions per symbol before a single byte is written (1.47 x 10^9 wave instructions per GiB, 0.12 of
 * the HBM roofline).  This is the same idea without the bookkeeping, verified instead of proven, with
 * decode_fix_kernel (= the exact decoder) behind it for every block where anything is off:
 *
 *   - (round 5: blocks without codes of more than 12 bits whose payload the shorter shares do not cut into more segments
 *     stage every lane's share in a COLUMN of its own - DFAST_COL_ROWS below; what follows describes the linear stage, the
 *     passes are the same on both: decode_payload_fast_impl<THREADS, COL>)
 *   - the payload is staged LINEARLY in segments of 512 x 288 bits; lane i owns the codewords that start
 *     in its 288 bits.  288 = 9 words: lanes that read "their" word k touch words 9 i + k - 64 different
 *     banks' worth of addresses, no conflicts, no interleaved layout, and the 32 bits at a position are one
 *     ds_read2_b32 + one v_alignbit_b32 of a position register that IS the LDS address (decode_sub.hpp);
 *   - scan: every lane walks from its start to the first codeword start at or behind its end, two table
 *     look-ups per window, counting - 10 instructions per symbol, no branch but the loop's;
 *   - lane 0 starts at the segment's true first codeword, every other lane first at its own first bit
 *     (speculation: a Huffman decoder falls into step within a few codewords), then at its left
 *     neighbour's end, again and again until no start changes.  By induction over the lanes the tracks are
 *     then the in-order decoder's: usually after two scans;
 *   - counts are prefix-summed and every lane decodes its symbols once more, four to a 32-bit store.  THIS
 *     pass checks what the in-order decoder would have met: a walk that leaves the tree or needs bits past
 *     the payload before the block is complete sends the block to the exact decoder, which delivers the
 *     reference's error code and byte count.  So does a block whose tree is not an ordinary one, whose
 *     starts do not settle in 64 rounds, or whose codes run past the staged words.
 * ==================================================================================== */
#define DFAST_SUBW 9u                               /* words per lane */
#define DFAST_SUB_BITS (32u * DFAST_SUBW)
#define DFAST_SLACK_WORDS 16u                       /* staged behind the segment: the last lane's window, a walk of a long code */
#define DFAST_RUNIN DFAST_SUB_BITS                   /* bits of a lane's share its first, speculative scan walks: all of them.
                                                       (96: nearly every WAVE then holds a lane that has not fallen into step, and a wave
                                                       rescans as long as its slowest lane - zipf255 1.87 -> 2.12 ms; codes of one length,
                                                       uniform bytes, never fall into step at all) */
#define DFAST_MAX_ROUNDS 64
/* Round 5: blocks without `long` codes stage every lane's share in a COLUMN of its own (decode_sub.hpp's layout: word r of the
 * lane at row r of its wave's slice, a row = the 64 lanes' words side by side): a lane's window read - two dwords at a
 * data-dependent word - then falls into the lane's own bank whatever the other lanes' positions are (ds_read2st64_b32: 4 LDS
 * cycles where the linear stage's took ~10).  A lane loads its own words from memory (three 16-byte loads at its 4-byte
 * aligned address; neighbours overlap by a word, from the vector cache).  Shares are at most 256 bits there: the column holds
 * the word with the bit in front of the share's first (positions are kept minus one), the share, the window behind its last
 * codeword start and the up to three symbols a lane decodes beyond its own to store a whole word - eleven rows. */
#define DFAST_COL_ROWS 11u                          /* ... of a block that has a table of pairs behind its stage */
#define DFAST_COL_SUB_BITS 256u
#define DFAST_COL_ROWS_WIDE 12u                     /* ... of a block without one (codes too long for pairs, blocks below 32 KiB): the stage runs on into
                                                       the pairs' place and the shares are the linear stage's 288 bits */
#ifndef DFAST_PAIRS_FROM
#define DFAST_PAIRS_FROM 32768u                      /* symbols of a block from which its scans read the table of pairs (dfast_pair_table) */
#endif
#define DREG_E_LONG 0u                                /* decode_regs.hpp's table: twelve bits that are the beginning of a longer code (a length of 0) */
#ifndef DREG_MIN_BLOCK
#define DREG_MIN_BLOCK 8192u                         /* symbols of a block from which decode_regs.hpp's path takes it.  (32 768 until round 6b: smaller blocks have codes
                                                        beyond 12 bits, which the path then declined - with the tables built twice for it.  Now, zipf255 with the block
                                                        index alone: 16 KiB blocks 3.15 -> 2.46 ms per GiB, 8 KiB blocks 5.74 -> 4.99; a 16 KiB huf_decode 81 -> 73 us.) */
#endif
#ifndef DFAST_JUMP_FROM_ROUND
#define DFAST_JUMP_FROM_ROUND 1                      /* the round loop's pass from which runs of one byte value are looked for (dfast_run_jump) */
#endif
/* (Round 4 also measured, and dropped again: a run-in of 96-192 bits in FRONT of a share before its first scan - 1.72 -> 1.66 ms per
 *  GiB on zipf255, 1.11 -> 1.31-1.54 on uniform bytes - and the same with the moved starts compacted into one wave's rescan - 1.73 ->
 *  1.77-1.91: profiles/r04/dfast_pre_runin.txt, dfast_compact.txt.) */

template <int THREADS>
struct DfastLds {
    static constexpr uint32_t AREA_WORDS = (uint32_t)((sizeof(DecShared<THREADS>::pay) + sizeof(DecShared<THREADS>::mark)) / sizeof(uint32_t));
    static constexpr uint32_t SEG_WORDS = (uint32_t)THREADS * DFAST_SUBW;
    static constexpr uint32_t STAGE_WORDS = SEG_WORDS + DFAST_SLACK_WORDS;                     /* the linear stage */
    static constexpr uint32_t COL_WORDS = (uint32_t)THREADS * DFAST_COL_ROWS;                   /* the column stage: [wave][row][lane] */
    static_assert((uint32_t)THREADS * DFAST_COL_ROWS_WIDE <= AREA_WORDS, "the wide column stage (no pairs behind it) fits pay + marks");
    static constexpr uint32_t AHEAD_WORDS = COL_WORDS > STAGE_WORDS ? COL_WORDS : STAGE_WORDS;  /* what lies in front of the table of pairs */
    static_assert(offsetof(DecShared<THREADS>, mark) == offsetof(DecShared<THREADS>, pay) + sizeof(DecShared<THREADS>::pay), "one area");
    static_assert(STAGE_WORDS + 4u <= AREA_WORDS, "the linear stage fits the area of the interleaved stage and its marks");
    static_assert(offsetof(DecShared<THREADS>, pay) % 16 == 0, "16-byte stage stores");
    /* the table of pairs (dfast_pair_table) behind the stage: the rest of the marks' area and the first entries of `ent` */
    static constexpr uint32_t PAIR_WORD = AHEAD_WORDS + 4u;
    static constexpr uint32_t PAIR_WORDS = (1u << DEC_LUT_BITS) / 2u;
    static_assert(offsetof(DecShared<THREADS>, ent) == offsetof(DecShared<THREADS>, mark) + sizeof(DecShared<THREADS>::mark), "ent runs on from the marks");
    static_assert(offsetof(DecShared<THREADS>, lr) == offsetof(DecShared<THREADS>, ent) + sizeof(DecShared<THREADS>::ent), "lr runs on from ent");
    static_assert((PAIR_WORD + PAIR_WORDS) * 4u <= sizeof(DecShared<THREADS>::pay) + sizeof(DecShared<THREADS>::mark) + sizeof(DecShared<THREADS>::ent) + sizeof(DecShared<THREADS>::lr),
                  "the pairs fit (they run on into ent and lr: blocks that have pairs have no long codes, and nothing else reads those two once the tables stand)");

    static_assert((PAIR_WORD * 4u) % 16u == 0u, "16-byte stores of the pairs");
    __device__ static __forceinline__ uint16_t *pairs(DecShared<THREADS> &sh) { return reinterpret_cast<uint16_t *>(sh.pay + PAIR_WORD); }
};

#ifdef DFAST_DEBUG
__device__ unsigned long long g_dfast_dbg[16];
#define DFAST_DBG(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_dfast_dbg[i], (unsigned long long)(v)); } while (0)
#define DFAST_DBGW(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dfast_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define DFAST_DBG(i, v) do { } while (0)
#define DFAST_DBGW(i, v) do { } while (0)
#endif

/* 64-bit left-aligned bit buffer over the linearly staged payload words (big-endian words): the edges of a
 * lane's output and the step-by-step path */
template <bool COL>
struct LinReaderT {
    const uint32_t *st;  /* linear stage: its word 0; column stage: the lane's word 0 (positions are then the lane's own: bits from that word) */
    uint32_t hi, lo;
    int32_t avail;
    uint32_t gf;         /* next staged word to append */
    __device__ __forceinline__ uint32_t word(uint32_t g) const { return COL ? st[64u * g] : st[g]; }

    __device__ __forceinline__ void load(uint32_t pos)
    {
        const uint32_t g = pos >> 5, off = pos & 31u;
        const uint64_t b = (((uint64_t)word(g) << 32) | word(g + 1)) << off;
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
        const uint64_t t = (uint64_t)word(gf) << (32 - avail);
        hi |= (uint32_t)(t >> 32);
        lo |= (uint32_t)t;
        avail += 32;
        gf++;
    }
};
typedef LinReaderT<false> LinReader;

/* the 32 bits at position register Q (decode_sub.hpp's convention: Q = position - 1 + a bias): linear stage - the bias is 8 x the
 * stage's LDS byte address, the word pair lies at Q >> 3; column stage - the bias is minus the position of the lane's column
 * word 0, the pair is rows Q >> 5 and + 1 of the column at LDS byte address cb (one ds_read2st64_b32) */
typedef const __attribute__((address_space(3))) uint32_t *dfast_lds_words;
typedef const __attribute__((address_space(3))) uint16_t *dfast_lds_halves;
template <bool COL>
__device__ __forceinline__ uint32_t dfast_bits_at(uint32_t Q, uint32_t cb)
{
    if (COL) {
        uint32_t a_;                                         /* ((Q >> 5) << 8) + cb in two instructions (the compiler makes three of it) */
        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a_) : "v"(Q & ~31u), "v"(cb));
        dfast_lds_words wp = (dfast_lds_words)(uintptr_t)a_;
        return __builtin_amdgcn_alignbit(wp[0], wp[64], ~Q);
    }
    dfast_lds_words wp = (dfast_lds_words)(uintptr_t)((Q >> 3) & ~3u);
    return __builtin_amdgcn_alignbit(wp[0], wp[1], ~Q);
}

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

/* One table step of a lane (tables of dec_build_tables): returns the entry (low byte = symbol); *ok is cleared
 * when the lookup is not a codeword.  The rare paths sit behind one wave-uniform branch. */
template <int THREADS, bool COL>
__device__ __forceinline__ uint32_t dfast_next(const DecShared<THREADS> &sh, LinReaderT<COL> &rd, uint32_t lim, bool &ok)
{
    uint32_t e = sh.lut[rd.index()];
    if (__builtin_expect(__ballot(e >= DEC_E_BAD) != 0ull, 0)) {
        if (COL && e >= DEC_E_LONG) {                     /* (the column stage is for blocks without such entries) */
            ok = false;
            e = 0x0100u;
        } else if (e >= DEC_E_LONG) {
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

/* Round 4, the table the scans of a block WITHOUT `long` entries read: what the next 12 bits hold as a whole.
 *   P[x] = (bits of the first codeword in the 12 bits x, and of the second if it lies completely inside) | (how many) << 12
 * for zipf255 1.5 codewords a look-up, for log text 1.9 (uniform bytes: one, their codes have 8 bits); a `bad` entry of the
 * table of single codewords (decode.hpp) is one look-up that advances by its `skip`, as it is there.  The kernel is bound
 * by vector instruction issue (46 a symbol, 21 of them in the scans): the sum of the entries a track has met IS its
 * position (low 12 bits) and its count (the bits above), one v_add3_u32 for two look-ups.
 * Eight entries per thread; P lies behind the stage (DfastLds::pairs). */
template <int THREADS>
__device__ __forceinline__ void dfast_pair_table(DecShared<THREADS> &sh)
{
    static_assert((1 << DEC_LUT_BITS) == THREADS * 8, "eight entries per thread");
    constexpr uint32_t MASK = (1u << DEC_LUT_BITS) - 1u;
    const uint32_t x0 = (uint32_t)threadIdx.x * 8u;
    uint32_t p[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t idx = x0 + j;
        const uint32_t e = sh.lut[idx];
        const uint32_t len = dec_e_adv(e);
        const uint32_t e2 = sh.lut[(idx << len) & MASK];                   /* the bits behind, zeros behind those: right for a codeword that ends inside */
        const bool two = e < DEC_E_BAD && e2 < DEC_E_BAD && len + (e2 >> 8) <= (uint32_t)DEC_LUT_BITS;
        p[j] = two ? (len + (e2 >> 8)) | 0x2000u : len | 0x1000u;
    }
    *reinterpret_cast<uint4 *>(DfastLds<THREADS>::pairs(sh) + x0) = make_uint4(p[0] | (p[1] << 16), p[2] | (p[3] << 16), p[4] | (p[5] << 16), p[6] | (p[7] << 16));
}

/* One scan of a lane: from `start` to the first codeword start at or behind `hi` (positions are bits of the
 * staged segment).  *end = that position, *cnt = table look-ups taken on the way (= codewords on a track that
 * meets no walk out of the tree; the write pass checks that).  qbase = 8 x the LDS byte address of the stage,
 * lut_addr = the LDS byte address of the table, pair_addr = that of the table of pairs.
 *   DFAST_LONGS (the block's table has `long` entries: codes of more than 12 bits, walked bit by bit behind a ballot):
 * look-up by look-up in the table of single codewords, every one asked whether it still is the lane's.
 *   DFAST_PAIRS (blocks of 32 KiB and more with codes of six bits and less: most): window by window - two look-ups
 * in the table of pairs, up to four codewords, 13 vector instructions - as long as a window BEGINS in front of `hi`;
 * the last window is then looked at again, codeword by codeword, for the end and the count.
 *   DFAST_SINGLES (the rest): the same with the table of single codewords.  (The end has to be the TRACK's - the first codeword start at or behind `hi`
 * - and not the end of whatever window crossed `hi`: two tracks that have met have the same codewords, not the
 * same windows, and an end that depends on the windows moves every lane to the right of a lane that moved.) */
enum { DFAST_SINGLES = 0, DFAST_LONGS = 1, DFAST_PAIRS = 2 };
template <int THREADS, int MODE, bool COL>
__device__ __forceinline__ void dfast_scan(const DecShared<THREADS> &sh, const uint32_t *stage, uint32_t qbase, uint32_t cb, uint32_t lut_addr, uint32_t pair_addr,
                                           uint32_t start, uint32_t hi, uint32_t lim, uint32_t *end, uint32_t *cnt)
{
    static_assert(!(COL && MODE == DFAST_LONGS), "long codes are walked on the linear stage");
    constexpr bool LONGS = MODE == DFAST_LONGS;
    uint32_t Q = start - 1u + qbase;                 /* (position - 1) + 8 x stage address: see decode_sub.hpp */
    const uint32_t hiQ = hi - 1u + qbase;
    uint32_t c = 0;
    DFAST_DBGW(8, 1);
    if (MODE == DFAST_SINGLES) {
        /* the same with the table of single codewords (blocks whose codes are too long for pairs, small blocks): two
         * look-ups, two codewords and 17 vector instructions per window */
        uint32_t Qg = Q, ng = 0;
        for (;;) {
            const bool act = Q < hiQ;
            if (!__any(act)) break;
            DFAST_DBGW(9, 1);
            Qg = act ? Q : Qg;
            ng += act ? 1u : 0u;
            const uint32_t d1 = dfast_bits_at<COL>(Q, cb);
            const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d1 >> 19) & 0x1ffeu));
            const uint32_t l1 = (e1 >> 8) & 31u;
            const uint32_t d2 = d1 << l1;
            const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d2 >> 19) & 0x1ffeu));
            /* (a leaf or `bad` entry advances by at most DEC_LUT_BITS: the second look-up always has its whole index.)  A lane
             * that is done walks on until its wave is: what lies there is read and ignored. */
            Q += l1 + ((e2 >> 8) & 31u);
        }
        if (ng != 0u) {
            const uint32_t d1 = dfast_bits_at<COL>(Qg, cb);
            const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d1 >> 19) & 0x1ffeu));
            const uint32_t l1 = (e1 >> 8) & 31u;
            const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(lut_addr + (((d1 << l1) >> 19) & 0x1ffeu));
            const uint32_t l2 = (e2 >> 8) & 31u;
            const uint32_t t1 = Qg + l1;
            const bool take2 = t1 < hiQ && l2 != 0u;
            c = 2u * (ng - 1u) + 1u + (take2 ? 1u : 0u);
            Q = t1 + (take2 ? l2 : 0u);
        } else {
            Q = start - 1u + qbase;
        }
        *end = Q + 1u - qbase;
        *cnt = c;
        return;
    }
    if (MODE == DFAST_PAIRS) {
        const uint32_t Q0 = Q;
        uint32_t S = 0, Sg = 0;                      /* the entries met: bits in the low 12 bits (a share and a window: < 4 096), codewords above */
        for (;;) {
            const bool act = Q < hiQ;
            if (!__any(act)) break;
            DFAST_DBGW(9, 1);
            const uint32_t d1 = dfast_bits_at<COL>(Q, cb);
            const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(pair_addr + ((d1 >> 19) & 0x1ffeu));
            const uint32_t d2 = d1 << (e1 & 31u);      /* (v_lshlrev_b32 takes the low five bits itself) */
            const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(pair_addr + ((d2 >> 19) & 0x1ffeu));
            /* (an entry advances by at most DEC_LUT_BITS: the second look-up always has its whole index.)  A lane that is done
             * stands still while its wave walks on. */
            if (act) {
                Sg = S;
                S += e1 + e2;
            }
            Q = Q0 + (S & 0xfffu);
        }
        const bool any = S != 0u;                      /* (every entry counts at least one) */
        if (__ballot(any)) {
            /* the last window again: its codewords start at Qg, b1 (the first entry's second, if it has one), b2, b3 (the
             * second entry's second) and the window ends at b4 >= hi */
            const uint32_t Qg = Q0 + (Sg & 0xfffu);
            const uint32_t d1 = dfast_bits_at<COL>(Qg, cb);
            const uint32_t i1 = (d1 >> 19) & 0x1ffeu;
            const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(pair_addr + i1);
            const uint32_t a1 = (*(dfast_lds_halves)(uintptr_t)(lut_addr + i1) >> 8) & 31u;
            const uint32_t i2 = ((d1 << (e1 & 31u)) >> 19) & 0x1ffeu;
            const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(pair_addr + i2);
            const uint32_t a2 = (*(dfast_lds_halves)(uintptr_t)(lut_addr + i2) >> 8) & 31u;
            const uint32_t b1 = Qg + a1, b2 = Qg + (e1 & 31u), b3 = b2 + a2, b4 = b2 + (e2 & 31u);
            const bool in1 = b1 < hiQ, in2 = b2 < hiQ, in3 = b3 < hiQ;
            const uint32_t endQ = !in1 ? b1 : !in2 ? b2 : !in3 ? b3 : b4;
            const uint32_t cn = (Sg >> 12) + 1u + ((in1 && (e1 & 0x2000u)) ? 1u : 0u) + (in2 ? 1u : 0u) + ((in3 && (e2 & 0x2000u)) ? 1u : 0u);
            if (any) {
                Q = endQ;
                c = cn;
            }
        }
        *end = Q + 1u - qbase;
        *cnt = c;
        return;
    }
    for (;;) {
        const bool act = Q < hiQ;
        if (!__any(act)) break;
        DFAST_DBGW(9, 1);
        const uint32_t d1 = dfast_bits_at<COL>(Q, cb);
        const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d1 >> 19) & 0x1ffeu));
        uint32_t l1 = (e1 >> 8) & 31u;
        const uint32_t d2 = d1 << l1;
        const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d2 >> 19) & 0x1ffeu));
        uint32_t l2 = (e2 >> 8) & 31u;
        /* the second look-up counts when the window still held a whole table index behind the first codeword */
        if (l1 > 20u) l2 = 0;
        if (LONGS) {
            if (e2 >= DEC_E_LONG) l2 = 0;            /* (it is the next window's first) */
            if (__builtin_expect(__ballot(act && e1 >= DEC_E_LONG) != 0ull, 0)) {
                if (act && e1 >= DEC_E_LONG) {       /* a code of more than 12 bits: walked bit by bit */
                    const uint32_t pos = Q + 1u - qbase;
                    const uint64_t r = dec_rare_lin<THREADS>(sh, stage, e1, pos, lim);
                    const int st = (int)(r >> 40);
                    l1 = (st == CW_EXH) ? (hi > pos ? hi - pos : 1u) : (uint32_t)r - pos;   /* (walks out of the tree resume behind the failing bit) */
                    l2 = 0;
                }
            }
        }
        /* a lane that has arrived stands still; a second look-up that starts at or behind `hi` is the neighbour's */
        const uint32_t t1 = Q + (act ? l1 : 0u);
        const bool take2 = t1 < hiQ && l2 != 0u;
        c += (act ? 1u : 0u) + (take2 ? 1u : 0u);
        Q = t1 + (take2 ? l2 : 0u);
    }
    *end = Q + 1u - qbase;
    *cnt = c;
}

/* Runs of one byte value.  A run is a periodic bit string, and a speculative lane inside it locks onto the pattern
 * a few bits off: nothing in the run tells the phases apart, so the lanes of a run are put right one per round,
 * from the left (a 16 KiB run of zeros in a 64 KiB block: 114 lanes).  But inside a run everything is known from
 * its beginning: every codeword start is the first one plus a multiple of the code length.
 *
 * dfast_run_at: does ONE codeword, repeated, fill the stage from bit `pos` to bit `hi` and a codeword further?
 * Returns its length (1..12), or 0.  (The 32 bits at pos + 32 k, k = 0..10; the string is periodic with period L
 * when every one of them, shifted on by L bits, is itself again.) */
template <bool COL>
__device__ __forceinline__ uint32_t dfast_run_at(uint32_t qbase, uint32_t cb, uint32_t lut_addr, uint32_t pos, uint32_t hi)
{
    const uint32_t Q0 = pos - 1u + qbase;
    uint32_t w[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const uint32_t Q = Q0 + 32u * (uint32_t)k;
        w[k] = dfast_bits_at<COL>(Q, cb);
    }
    const uint32_t e = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((w[0] >> 19) & 0x1ffeu));
    const uint32_t L = e >> 8;
    if (e >= DEC_E_BAD || L == 0u) return 0u;                          /* not a codeword of the table */
    bool same = true;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        /* the 32 bits at pos + 32 k + L; only those that begin in front of hi count */
        const bool counts = pos + 32u * (uint32_t)k < hi;
        if (counts && __builtin_amdgcn_alignbit(w[k], w[k + 1], 32u - L) != w[k]) same = false;
    }
    return same ? L : 0u;
}

/* dfast_run_jump: a lane that has just been put right (`changed`) and holds one codeword over and over is the
 * beginning of a run, as far as this wave is concerned; the lanes to its right take the codeword starts that follow
 * from it - if THEIR shares hold the same repetition from there on, which each of them checks in its own bits -
 * and a stretch of any length inside the wave settles in this round.  A guess that does not hold is found out like
 * any wrong start: by the neighbour's end in the next round.  Returns (start, end, count, moved) of the lane.
 * (Out of line: it runs in the rounds after the second only, and its registers are its own.) */
template <bool COL>
__device__ __noinline__ uint4 dfast_run_jump(uint32_t qbase, uint32_t cb, uint32_t lut_addr, bool changed, bool dead, uint32_t hi, uint32_t sb, uint32_t pay_rel,
                                             uint32_t start, uint32_t end, uint32_t cnt0)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t myL = (changed && start < hi) ? dfast_run_at<COL>(qbase, cb, lut_addr, start, hi) : 0u;
    /* the nearest such lane to the left (max-scan of lane indices), its start and code length */
    int src = (myL != 0u) ? (int)lane : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(src, o);
        if ((int)lane >= o && t > src) src = t;
    }
    const uint32_t P = (uint32_t)__shfl((int)start, src < 0 ? 0 : src);
    const uint32_t L = (uint32_t)__shfl((int)myL, src < 0 ? 0 : src);
    bool pass = false;
    uint32_t cand = 0;
    if (src >= 0 && src < (int)lane && !dead && !changed) {
        /* the first codeword start at or behind my first bit: P plus a multiple of L */
        const uint32_t lo = hi - sb;
        const uint32_t back = (lo - P) % L;
        cand = lo + (back ? L - back : 0u);
        pass = cand >= hi || (cand < pay_rel && dfast_run_at<COL>(qbase, cb, lut_addr, cand, hi) == L);
    }
    /* ... as far as every lane on the way agrees: the last lane that does not, against the source */
    int bad = (src >= 0 && src < (int)lane && !pass) ? (int)lane : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(bad, o);
        if ((int)lane >= o && t > bad) bad = t;
    }
    if (pass && bad <= src && cand != start) {
        const uint32_t cnt = (cand < hi) ? (hi - cand + L - 1u) / L : 0u;
        return make_uint4(cand, cand + cnt * L, cnt, 1u);
    }
    return make_uint4(start, end, cnt0, 0u);                 /* (start, end, count, moved) */
}

/* Write pass of a lane: its first `quota` symbols, from `start`, to g[0 .. quota).  Returns the position behind
 * the last one; *ok is cleared when a look-up was not a codeword (a walk out of the tree, bits past `lim`). */
/* Round 4: `whole` = the lane may store its last, partly filled word WHOLE: the symbols behind its own are the next
 * lane's first (the same values from whoever stores them), and they stay inside the block's output.  Its symbols then
 * need no step-by-step tail (up to three look-ups with a bit buffer and byte stores: a third of this pass); the position
 * behind its last symbol is the end its scan found, and the return value is not used. */
template <int THREADS, bool COL>
__device__ __forceinline__ uint32_t dfast_write(const DecShared<THREADS> &sh, const uint32_t *stage, uint32_t qbase, uint32_t cb, uint32_t lut_addr,
                                                uint32_t start, uint32_t quota, uint32_t lim, uint8_t *g, bool *ok_out, bool whole = false)
{
    /* (column stage: `stage` is the lane's column word 0 and org = the position of that word's first bit: the step-by-step
     *  reader works in the lane's own positions) */
    bool ok = true;
    const uint32_t org = COL ? 0u - qbase : 0u;                      /* (qbase = -org there) */
    LinReaderT<COL> rd;
    rd.st = stage;
    rd.load(start - org);
    /* (the words go out from the lane's first symbol on, wherever the output stands: 32-bit stores need no
     * alignment on gfx950, and the symbols in front of a 4-byte boundary, one step-by-step look-up each, cost as
     * much as the two words behind them) */
    const uint32_t head = 0;
    const uint32_t p0 = start;
    typedef uint32_t __attribute__((aligned(1))) unaligned_u32;
    unaligned_u32 *gw = reinterpret_cast<unaligned_u32 *>(g + head);
    const uint32_t words = whole ? (quota - head + 3u) >> 2 : (quota - head) >> 2;
    /* whole words: four table entries folded into one register, two per window, no branch; an entry that is
     * not a leaf advances like one and is only remembered */
    uint32_t Q = p0 - 1u + qbase;
    uint32_t special = 0;
    /* four symbols = one word: two windows of two look-ups */
#define DFAST_WORD(ACC)                                                                                        \
    {                                                                                                         \
        uint32_t pr_[2];                                                                                      \
        _Pragma("unroll")                                                                                     \
        for (int j = 0; j < 2; j++) {                                                                         \
            const uint32_t d1 = dfast_bits_at<COL>(Q, cb);                                   \
            const uint32_t e1 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d1 >> 19) & 0x1ffeu));            \
            const uint32_t l1 = (e1 >> 8) & 31u;                                                               \
            const uint32_t d2 = d1 << l1;                                                                     \
            const uint32_t e2 = *(dfast_lds_halves)(uintptr_t)(lut_addr + ((d2 >> 19) & 0x1ffeu));            \
            special |= e1 | e2;                                                                               \
            pr_[j] = __builtin_amdgcn_perm(e2, e1, 0x0c0c0400u);          /* the two bytes */                  \
            Q += l1 + ((e2 >> 8) & 31u);                                                                      \
        }                                                                                                     \
        ACC = __builtin_amdgcn_perm(pr_[1], pr_[0], 0x05040100u);                                             \
    }
    /* Round 5: sixteen symbols a store.  A wave's store instruction is 64 addresses in 64 different cache lines whatever
     * its width, and the address path takes them one by one: with a dword a lane and store the write pass was four
     * times as long as the scan that walks the same bits (tools/phase_fast.py: 40 % of the kernel). */
    typedef uint32_t unaligned_q4 __attribute__((ext_vector_type(4), aligned(1)));
    uint32_t k = 0;
    for (; k + 4u <= words; k += 4u) {
        uint32_t w0, w1, w2, w3;
        DFAST_WORD(w0)
        DFAST_WORD(w1)
        DFAST_WORD(w2)
        DFAST_WORD(w3)
        unaligned_q4 v4;
        v4.x = w0; v4.y = w1; v4.z = w2; v4.w = w3;
        *reinterpret_cast<unaligned_q4 *>(g + head + 4u * k) = v4;
    }
    for (; k < words; k++) {
        uint32_t acc;
        DFAST_WORD(acc)
        gw[k] = acc;
    }
#undef DFAST_WORD
    uint32_t p1 = Q + 1u - qbase;
    if (__builtin_expect(__ballot((special & 0xC000u) != 0u) != 0ull, 0)) {
        if (special & 0xC000u) {                     /* a long code (or worse) among them: the words again, step by step */
            rd.load(p0 - org);
            uint8_t *b = g + head;
            for (uint32_t c = 0; c < 4u * words; c++) {
                b[c] = (uint8_t)dfast_next<THREADS, COL>(sh, rd, lim, ok);
                if (rd.avail <= 32) rd.refill();
            }
            p1 = rd.pos() + org;
        }
    }
    if (__ballot(!whole)) {
        if (!whole) {
            rd.load(p1 - org);
            for (uint32_t c = head + 4u * words; c < quota; c++) {
                g[c] = (uint8_t)dfast_next<THREADS, COL>(sh, rd, lim, ok);
                if (rd.avail <= 32) rd.refill();
            }
            p1 = rd.pos() + org;
        }
    }
    *ok_out = ok;
    return p1;
}

/* The column stage of a segment (blocks without long codes): lane t's column holds the payload words from the one with bit
 * lo_t - 1 on (lo_t = the first bit of its share; lane 0: the segment's true first bit), DFAST_COL_ROWS of them, big-endian:
 * twelve dwords at the lane's own 4-byte aligned address (the payload's bytes need not be aligned: one v_perm_b32 per word
 * puts them right, as the linear stage does).  word0 = the index of the lane's first word among the segment's words (from
 * seg0; -1 for a lane 0 whose share starts with the segment).  A lane whose words reach beyond `readable` takes them byte by
 * byte with zeros behind the end (the stream's last segment). */
template <int THREADS>
__device__ __forceinline__ void dfast_stage_col(uint32_t *col, const uint8_t *pay, uint64_t seg0, uint64_t readable, int32_t word0, bool wanted, const bool wide)
{
    if (!wanted) return;
    const int64_t first = (int64_t)(seg0 >> 3) + 4 * (int64_t)word0;       /* payload byte of the column's word 0 (-4: the block's tree ends there) */
    uint32_t w[DFAST_COL_ROWS_WIDE];
    if (first + 4 * (int64_t)(DFAST_COL_ROWS_WIDE + 1u) <= (int64_t)readable) {
        struct __attribute__((packed, aligned(4))) Q4 { uint32_t x, y, z, w; };
        const uintptr_t a = (uintptr_t)((intptr_t)(uintptr_t)pay + (intptr_t)first);
        const uint32_t m = (uint32_t)(a & 3u);
        const uint32_t *qw = reinterpret_cast<const uint32_t *>(a - m);
        const Q4 v0 = *reinterpret_cast<const Q4 *>(qw), v1 = *reinterpret_cast<const Q4 *>(qw + 4), v2 = *reinterpret_cast<const Q4 *>(qw + 8);
        const uint32_t x = qw[12];
        const uint32_t d[13] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, x};
        const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
#pragma unroll
        for (uint32_t r = 0; r < DFAST_COL_ROWS_WIDE; r++) w[r] = __builtin_amdgcn_perm(d[r + 1], d[r], sel);
    } else {
#pragma unroll
        for (uint32_t r = 0; r < DFAST_COL_ROWS_WIDE; r++) {
            const int64_t off = first + 4 * (int64_t)r;
            w[r] = off >= 0 ? load_be32(pay, (uint64_t)off, readable) : 0u;
        }
    }
#pragma unroll
    for (uint32_t r = 0; r < DFAST_COL_ROWS; r++) col[64u * r] = w[r];
    if (wide) col[64u * DFAST_COL_ROWS] = w[DFAST_COL_ROWS];
}

/* The segment that begins at bit seg0 of the payload into the linear stage: need_words words, big-endian. */
template <int THREADS>
__device__ __forceinline__ void dfast_stage(uint32_t *stage, const uint8_t *pay, uint64_t seg0, uint64_t readable, uint32_t need_words, uint64_t produced)
{
    typedef DfastLds<THREADS> L;
    const int tid = (int)threadIdx.x;
    {
            /* four words per thread and step from 20 bytes at a 4-byte aligned address, as decode_sub stages */
            struct __attribute__((packed, aligned(4))) Q4 { uint32_t x, y, z, w; };
            const uint64_t byte0 = seg0 >> 3;
            constexpr uint32_t STEPS = (L::STAGE_WORDS + 4u * THREADS - 1u) / (4u * THREADS);
            if (byte0 + 4ull * (4ull * THREADS * STEPS) + 24ull <= readable) {
                const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)(pay + byte0));
                const uint32_t m = (uint32_t)(a & 3u);
                const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
                const uint32_t *qw = reinterpret_cast<const uint32_t *>(a - m);
                Q4 v[STEPS];
                uint32_t x[STEPS];
#pragma unroll
                for (uint32_t k = 0; k < STEPS; k++) {
                    const uint32_t i4 = dmin<uint32_t>(4u * ((uint32_t)tid + (uint32_t)THREADS * k), (need_words - 1u) & ~3u);   /* (nothing past the words the shares need) */
                    v[k] = *reinterpret_cast<const Q4 *>(qw + i4);
                    x[k] = qw[i4 + 4];
                }
#pragma unroll
                for (uint32_t k = 0; k < STEPS; k++) {
                    const uint32_t i4 = 4u * ((uint32_t)tid + (uint32_t)THREADS * k);
                    if (i4 < need_words)
                        *reinterpret_cast<uint4 *>(stage + i4) =
                            make_uint4(__builtin_amdgcn_perm(v[k].y, v[k].x, sel), __builtin_amdgcn_perm(v[k].z, v[k].y, sel),
                                       __builtin_amdgcn_perm(v[k].w, v[k].z, sel), __builtin_amdgcn_perm(x[k], v[k].w, sel));
                }
            } else {
                for (uint32_t i = (uint32_t)tid; i < need_words; i += THREADS) stage[i] = load_be32(pay, byte0 + 4ull * i, readable);
            }
        }
}

/* A block's payload (tables in sh, built by dec_build_tables<THREADS, true>).  Returns true (workgroup-uniform)
 * when block_len symbols were written and everything the in-order decoder would have checked held.
 * readable = bytes that may be loaded from `pay` on (to the end of the stream: what lies behind the block's
 * payload is never part of a track that passes the checks, so it need not be zeroed). */
template <int THREADS, bool COL>
__device__ __forceinline__ bool decode_payload_fast_impl(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, uint64_t block_len,
                                    uint8_t *gout, uint64_t *end_bits, uint64_t hint_bytes, const bool longs, const bool pairs)
{
    /* end_bits: the caller does not know where the payload ends (the raw-stream probe: pay_bytes = the rest of the stream) and
     * wants to be told.  hint_bytes (with end_bits): where it probably ends - the next header candidate; taken for the end
     * as the guesses below are, and given up like them by the segment whose symbols come short of the block. */
    typedef DfastLds<THREADS> L;
    constexpr int WAVES = THREADS / 64;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    /* linear stage: runs on into the marks' area; column stage: this lane's column, word 0 (its wave's slice: DFAST_COL_ROWS rows of 64 words) */
    const bool wide = COL && !pairs;                                  /* (no table of pairs behind the stage: twelve rows, shares of 288 bits) */
    uint32_t *stage = COL ? sh.pay + (uint32_t)wave * ((wide ? DFAST_COL_ROWS_WIDE : DFAST_COL_ROWS) * 64u) + (uint32_t)lane : sh.pay;
    const uint32_t cb = (uint32_t)(uintptr_t)(dfast_lds_words)stage;
    uint32_t qbase = COL ? 0u : 8u * cb;                               /* (column stage: minus the position of the column's word 0, per lane and segment) */
    const uint32_t SUB_BITS = (COL && !wide) ? DFAST_COL_SUB_BITS : DFAST_SUB_BITS;
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(dfast_lds_halves)sh.lut;
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uint32_t lim = (L::STAGE_WORDS - 2u) * 32u;                  /* bits a walk may look at */
    const uint32_t pair_addr = (uint32_t)(uintptr_t)(dfast_lds_halves)L::pairs(sh);
    if (pairs) dfast_pair_table<THREADS>(sh);                          /* (read behind the stage's first barrier) */
    uint64_t true_start = 0, produced = 0;
    bool ok = true;
    bool trust = true;                                                 /* (uniform) guesses at the block's end are allowed */
    while (produced < block_len) {
        if (true_start >= pay_bits) { ok = false; DFAST_DBG(0, 1); break; }             /* input exhausted: the exact decoder says how */
        const uint64_t seg0 = true_start & ~31ull;
        /* Round 4: the payload left is cut into EQUAL shares (the block index says where it ends): the last segment of a block
         * is as full as the others instead of a quarter full on average, and every scan is that much shorter (zipf255 at
         * 64 KiB: 3.2 segments of 288-bit shares -> 4 of 232).  The raw-stream probe does not know the end: 288 as before. */
        uint32_t sb = SUB_BITS;
        const bool hinted = uni32((end_bits && trust && hint_bytes * 8ull > seg0 && hint_bytes <= pay_bytes) ? 1u : 0u) != 0u;
        if (!end_bits || hinted) {
            const uint64_t rem = (hinted ? hint_bytes * 8ull : pay_bits) - seg0;
            const uint64_t nseg = SUB_BITS == DFAST_SUB_BITS ? (rem + (uint64_t)THREADS * DFAST_SUB_BITS - 1u) / ((uint64_t)THREADS * DFAST_SUB_BITS)
                                                             : (rem + (uint64_t)THREADS * DFAST_COL_SUB_BITS - 1u) / ((uint64_t)THREADS * DFAST_COL_SUB_BITS);   /* (constant divisors) */
            const uint64_t even = (rem + nseg * THREADS - 1u) / (nseg * THREADS);
            sb = (uint32_t)dmin<uint64_t>(dmax<uint64_t>(even, 64u), SUB_BITS);
        }
        sb = uni32(sb);
        const uint32_t need_words = uni32(dmin<uint32_t>(((uint32_t)THREADS * sb + 31u) / 32u + DFAST_SLACK_WORDS, L::STAGE_WORDS));
        const uint32_t pay_rel = (uint32_t)dmin<uint64_t>(pay_bits - seg0, 0xfffffff0ull);   /* payload bits from seg0 on */
        const uint32_t first = (uint32_t)(true_start - seg0);
        const uint32_t hi = ((uint32_t)tid + 1u) * sb;
        __syncthreads();                                               /* the previous segment's readers are done */
        unsigned long long pt = DPROF_T();
        if (COL) {
            /* the column's word 0 is the one that holds the bit in FRONT of the lane's first: positions are kept minus one */
            const int32_t lo1 = (int32_t)(tid == 0 ? first : hi - sb) - 1;
            const int32_t word0 = lo1 >> 5;                                /* (-1 for a lane 0 whose share starts with the segment) */
            qbase = 0u - (uint32_t)(32 * word0);
            dfast_stage_col<THREADS>(stage, pay, seg0, readable, word0, hi - sb < pay_rel, wide);
        } else {
            dfast_stage<THREADS>(stage, pay, seg0, readable, need_words, produced);
        }
        __syncthreads();
        DPROF_ADD(1, pt); pt = DPROF_T();
        /* speculation: every lane but the first starts at its own first bit - a decoder that starts anywhere falls
         * into step within a few codewords, and a lane that has not is found out below */
        uint32_t start = tid == 0 ? first : hi - sb;
        /* a lane whose share lies behind the payload holds nothing and ends where its share ends (the block's last
         * segment is a quarter full on average: its other lanes scanned what follows the block, and passed every
         * change of their neighbour's on) */
        bool dead = hi - sb >= pay_rel;
        const uint64_t remaining = block_len - produced;
        /* A caller that does not know where the payload ends (the raw-stream probe: pay_bytes = the rest of the
         * stream) has no dead lanes by the test above, and the lanes behind the block's last symbol scan the next
         * block's header and payload, round after round.  Two guesses at where the block ends, both checked by the
         * symbol count of the segment (a segment that ends short of the block although lanes were taken for dead is
         * done again without guessing):
         *  - before the first scan: the bits per symbol of the block so far, a sixteenth more, and 1 024 bits; */
        bool guessed = false;                                          /* (uniform) lanes may have been taken for dead */
        if (hinted) {
            /* (round 4) the next candidate's offset: nearly always the end, and then this is the indexed decoder's segment */
            const uint32_t bound = (uint32_t)dmin<uint64_t>(hint_bytes * 8ull - seg0, 0xfffffff0ull);
            if (!dead && hi - sb >= bound) dead = true;
            guessed = (uint32_t)(THREADS - 1) * sb >= bound;
        } else
        if (end_bits && trust && produced != 0) {
            const float est = (float)remaining * ((float)true_start / (float)produced);
            const float lim_f = (float)first + est * 1.0625f + 1024.0f;
            const uint32_t bound = lim_f < 4.0e9f ? (uint32_t)lim_f : 0xffffffffu;
            if (!dead && hi - sb >= bound) dead = true;
            guessed = (uint32_t)(THREADS - 1) * sb >= bound;
        }
        uint32_t end = hi, cnt = 0;
        if (__ballot(!dead)) {
            if constexpr (!COL) {
                if (longs) dfast_scan<THREADS, DFAST_LONGS, false>(sh, stage, qbase, cb, lut_addr, pair_addr, dead ? hi : start, hi, lim, &end, &cnt);
                else if (pairs) dfast_scan<THREADS, DFAST_PAIRS, false>(sh, stage, qbase, cb, lut_addr, pair_addr, dead ? hi : start, hi, lim, &end, &cnt);
                else dfast_scan<THREADS, DFAST_SINGLES, false>(sh, stage, qbase, cb, lut_addr, pair_addr, dead ? hi : start, hi, lim, &end, &cnt);
            } else {
                if (pairs) dfast_scan<THREADS, DFAST_PAIRS, true>(sh, stage, qbase, cb, lut_addr, pair_addr, dead ? hi : start, hi, lim, &end, &cnt);
                else dfast_scan<THREADS, DFAST_SINGLES, true>(sh, stage, qbase, cb, lut_addr, pair_addr, dead ? hi : start, hi, lim, &end, &cnt);
            }
            if (dead) { end = hi; cnt = 0; }
        }
        /*  - after it: the speculative counts are right to a few symbols either way; a lane in front of which they
         *    already hold the rest of the block and a margin is taken for dead. */
        if (end_bits && trust && !hinted) {
            uint32_t spec_total;
            const uint32_t exs = block_excl_scan_u32<THREADS>(cnt, sh.part, spec_total);
            if (!dead && (uint64_t)exs >= remaining + 128u + ((uint32_t)tid >> 2)) {
                dead = true;
                end = hi;
                cnt = 0;
            }
            guessed = guessed || (uint64_t)uni32(spec_total) >= remaining + 128u;
        }
        if (lane == 63) sh.wend[wave] = end;
        __syncthreads();
        DPROF_ADD(2, pt); pt = DPROF_T();
        int rounds = 0;
        for (;;) {
            /* left neighbour's end: a DPP move inside the wave, LDS across the wave seams */
            uint32_t ns = wave_up1_u32(end);
            if (lane == 0) ns = (tid == 0) ? first : sh.wend[wave - 1];
            const int changed = (ns != start) && !dead;
            __syncthreads();                                           /* everyone has read sh.wend */
            if (__ballot(changed != 0)) {
                DFAST_DBGW(rounds == 0 ? 6 : rounds == 1 ? 7 : 14, 1);
                if (changed) start = ns;
                uint32_t e2 = end, c2 = cnt;
                if constexpr (!COL) {
                    if (longs) dfast_scan<THREADS, DFAST_LONGS, false>(sh, stage, qbase, cb, lut_addr, pair_addr, changed ? start : hi, hi, lim, &e2, &c2);
                    else if (pairs) dfast_scan<THREADS, DFAST_PAIRS, false>(sh, stage, qbase, cb, lut_addr, pair_addr, changed ? start : hi, hi, lim, &e2, &c2);
                    else dfast_scan<THREADS, DFAST_SINGLES, false>(sh, stage, qbase, cb, lut_addr, pair_addr, changed ? start : hi, hi, lim, &e2, &c2);
                } else {
                    if (pairs) dfast_scan<THREADS, DFAST_PAIRS, true>(sh, stage, qbase, cb, lut_addr, pair_addr, changed ? start : hi, hi, lim, &e2, &c2);
                    else dfast_scan<THREADS, DFAST_SINGLES, true>(sh, stage, qbase, cb, lut_addr, pair_addr, changed ? start : hi, hi, lim, &e2, &c2);
                }
                if (changed) { end = e2; cnt = c2; }
            }
            /* ---- runs of one byte value (only when the starts have not settled in two rounds): dfast_run_jump ---- */
            int jumped = 0;
            if (rounds >= DFAST_JUMP_FROM_ROUND && __ballot(changed != 0)) {
                DFAST_DBGW(15, 1);
                const uint4 r = dfast_run_jump<COL>(qbase, cb, lut_addr, changed != 0, dead, hi, sb, pay_rel, start, end, cnt);
                start = r.x; end = r.y; cnt = r.z; jumped = (int)r.w;
            }
            if (lane == 63) sh.wend[wave] = end;
            if (!__syncthreads_or(changed | jumped)) break;
            DFAST_DBG(10, 1);
            if (++rounds > DFAST_MAX_ROUNDS) { ok = false; DFAST_DBG(1, 1); break; }    /* (uniform: every thread counts the same rounds) */
        }
        if (!ok) break;
        DPROF_ADD(3, pt); pt = DPROF_T();
        uint32_t seg_total;
        const uint32_t ex = block_excl_scan_u32<THREADS>(cnt, sh.part, seg_total);
        seg_total = uni32(seg_total);
        if (guessed) DFAST_DBG(12, 1);
        if (guessed && (uint64_t)seg_total < remaining) {             /* a guess that did not hold: the segment again, without */
            DFAST_DBG(13, 1);
            trust = false;
            continue;
        }
        trust = true;
        const uint32_t take = (uint32_t)dmin<uint64_t>(seg_total, remaining);
        uint32_t quota = 0;
        if (ex < take) {
            quota = take - ex;
            if (quota > cnt) quota = cnt;
        }
        DPROF_ADD(4, pt); pt = DPROF_T();
        bool lane_ok = true;
        if (quota) {
            /* (all of the lane's symbols, and the whole last word still inside this block's output) */
            const bool whole = quota == cnt && produced + ex + ((quota + 3u) & ~3u) <= block_len;
            uint32_t qe = dfast_write<THREADS, COL>(sh, stage, qbase, cb, lut_addr, start, quota, lim, gout + produced + ex, &lane_ok, whole);
            if (whole) qe = end;
#ifdef DFAST_DEBUG
            if (!lane_ok) atomicAdd(&g_dfast_dbg[4], 1ull);
            if (qe > pay_rel) atomicAdd(&g_dfast_dbg[5], 1ull);
#endif
            if (qe > pay_rel) { lane_ok = false; }                         /* a codeword of the block needs bits past the payload */
            if (end_bits && ex + quota == take && (uint64_t)take == remaining) sh.qend = qe;      /* behind the block's last symbol */
        }
        const uint32_t last_end = uni32(sh.wend[WAVES - 1]);
        if (!__syncthreads_and(lane_ok ? 1 : 0)) { ok = false; DFAST_DBG(2, 1); break; }
        DPROF_ADD(5, pt);
        DFAST_DBG(11, 1);
        produced += take;
        if (take == 0) { ok = false; DFAST_DBG(3, 1); break; }                          /* (no progress: cannot happen on a track that holds codewords) */
        if (end_bits && produced == block_len) *end_bits = seg0 + (uint64_t)uni32(sh.qend);      /* payload bits up to and including the last symbol */
        true_start = seg0 + last_end;
    }
    return ok;
}

/* The block's payload, by the form of stage its tables allow: blocks with `long` entries (codes of more than 12 bits, walked bit
 * by bit along the staged words) on the linear stage, all others on the column stage. */
template <int THREADS, bool ALLOW_COL = true>
__device__ __forceinline__ bool decode_payload_fast(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, uint64_t block_len,
                                    uint8_t *gout, uint64_t *end_bits = nullptr, uint64_t hint_bytes = 0)
{
    const int tid = (int)threadIdx.x;
    /* does the table hold `long` entries at all?  (eight entries per thread) */
    bool longs, pairs = false;
    {
        const uint32_t *t = reinterpret_cast<const uint32_t *>(sh.lut) + 4 * tid;      /* (the table is 4-byte aligned) */
        const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        const uint32_t any = (t0 | t1 | t2 | t3) & 0x80008000u;            /* bit 15: long (bad entries have bit 14 only) */
        static_assert((1 << DEC_LUT_BITS) == THREADS * 8 && DEC_E_LONG == 0xC000u && DEC_E_BAD == 0x4000u, "eight entries per thread");
        longs = __syncthreads_or(any != 0u) != 0;
        /* the table of pairs pays for itself (its build, the longer look at a scan's last window) when codewords are short
         * enough to come in pairs - one of six bits or less: it fits the 12 bits twice - and the block is long enough
         * (1 GiB: zipf255 1.63 -> 1.54 ms, log text 1.53 -> 1.47; uniform bytes, without a pair, 1.03 -> 1.08 with it, zipf255 in
         * 16 KiB blocks 2.97 -> 3.06: those keep the table of singles) */
        if (!longs && block_len >= DFAST_PAIRS_FROM) {
            const uint32_t m = dmin<uint32_t>(dmin<uint32_t>(dmin<uint32_t>(t0 & 0xffffu, t0 >> 16), dmin<uint32_t>(t1 & 0xffffu, t1 >> 16)),
                                              dmin<uint32_t>(dmin<uint32_t>(t2 & 0xffffu, t2 >> 16), dmin<uint32_t>(t3 & 0xffffu, t3 >> 16)));
            pairs = __syncthreads_or(m < 0x0700u) != 0;                     /* a leaf is (bits << 8) | byte */
        }
    }
#ifdef DFAST_LINEAR_ONLY
    return decode_payload_fast_impl<THREADS, false>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, longs, pairs);
#else
    if constexpr (!ALLOW_COL) return decode_payload_fast_impl<THREADS, false>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, longs, pairs);
    if (longs) return decode_payload_fast_impl<THREADS, false>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, true, false);
    /* The column stage's shares are 256 bits at most, the linear stage's 288: a payload that the shorter shares cut into one
     * segment more keeps the linear stage (uniform bytes in 64 KiB blocks: 589 824 bits = 4 x 512 x 288 exactly - five segments
     * of columns, 1.12 -> 1.58 ms per GiB, and shares of 288 bits = 32 of its 9-bit codes keep every speculative start right). */
    const uint64_t est_bits = end_bits ? hint_bytes * 8ull : pay_bytes * 8ull;      /* (the probe: the next candidate's offset, 0 = none) */
    const uint64_t seg_col = (est_bits + (uint64_t)THREADS * DFAST_COL_SUB_BITS - 1u) / ((uint64_t)THREADS * DFAST_COL_SUB_BITS);
    const uint64_t seg_lin = (est_bits + (uint64_t)THREADS * DFAST_SUB_BITS - 1u) / ((uint64_t)THREADS * DFAST_SUB_BITS);
    if (pairs && uni32(seg_col != seg_lin ? 1u : 0u) != 0u)          /* (without pairs the columns' shares are 288 bits too) */
        return decode_payload_fast_impl<THREADS, false>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, false, pairs);
    return decode_payload_fast_impl<THREADS, true>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, false, pairs);
#endif
}

/* (Round 4 also walked TWO shares a lane side by side - two chains of dependent LDS reads in flight together: bit-exact and
 * slower, zipf255 1.58 -> 2.18 ms per GiB, uniform bytes unchanged; profiles/r04/dfast_two_chains.txt and notebook_r04.md have
 * the numbers, the code went with round 5's clean-up.) */
template <int THREADS, bool ALLOW_COL = true>
__device__ __forceinline__ bool decode_payload_dfast(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, uint64_t block_len,
                                    uint8_t *gout, uint64_t *end_bits = nullptr, uint64_t hint_bytes = 0)
{
    return decode_payload_fast<THREADS, ALLOW_COL>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes);
}

/* ======================================================================================
 * Round 4: the table of an ENCODER-MADE tree without walking it.  dec_build_tables finds every node's right child by a
 * search and walks the tree twelve levels deep for each of the 4 096 entries - about 1 150 vector instructions per thread
 * and block, a seventh of this kernel's.  For the trees an encoder writes (4K + 1 entries, K leaves, a root with a left
 * child only, every other node two children) the depths follow from two facts: in
 * preorder a leaf's depth is (left turns on its path) + (right turns), the left turns are a prefix sum over the entries
 * (+1 node, -1 marker), the right turns are the one bits of the leaf's code, and code(k + 1) = code(k) + 2^-depth(k) -
 * one chain of four scalar instructions per leaf, walked by ONE wave; the claimed depths are then checked against the
 * entries' positions exactly as dsub_fast_tables (decode_sub.hpp) checks the sub-index's, and the table is filled eight
 * consecutive entries per thread.  Entries are dec_build_tables' (decode.hpp).  Returns false - nothing of the tables
 * is to be used then, the caller walks the tree - for any other shape of tree and for blocks with codes of more than
 * DEC_LUT_BITS bits (their `long` entries need the child links only the walk builds).
 * ==================================================================================== */
/* where decode_regs.hpp's table build keeps the leaves in preorder: codes (left-aligned) and bytes in sh.leaves - they stay, behind
 * everything the columns overwrite (pay, marks, ent and most of lr), for the block's codes beyond the table's 12 bits
 * (dreg_long_entry); the lengths, which only the build needs, where FastLdsOf has them. */
template <class SH>
struct DregLeafLdsOf {
    static_assert(sizeof(SH::leaves) >= 320u * 4u && sizeof(SH) <= 40960u, "the leaves fit, and four workgroups a CU");
    __device__ static __forceinline__ uint32_t *code(SH &sh) { return sh.leaves; }
    __device__ static __forceinline__ uint8_t *len(SH &sh) { return reinterpret_cast<uint8_t *>(sh.lr + 256); }
    __device__ static __forceinline__ uint8_t *sym(SH &sh) { return reinterpret_cast<uint8_t *>(sh.leaves + 256); }
};
template <class SH, bool REGS> struct DfastLeafLds : FastLdsOf<SH> {};
template <class SH> struct DfastLeafLds<SH, true> : DregLeafLdsOf<SH> {};

template <int THREADS, bool SPEC, bool REGS = false>
__device__ bool dfast_tables_from_tree(DecShared<THREADS> &sh, const uint8_t *tree, int tree_len)
{
    typedef DfastLeafLds<DecShared<THREADS>, REGS> F;
    constexpr int WAVES = THREADS / 64;
    static_assert(THREADS * 2 >= HUF_TREE_MAX - 1 && THREADS >= 256 && WAVES <= 8, "two entries per thread");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t K = (uint32_t)(tree_len - 1) >> 2;
    if (tree_len < 9 || tree_len > HUF_TREE_MAX || ((tree_len - 1) & 3) != 0) return false;     /* uniform */
    __syncthreads();                                                            /* the previous user of sh is done */
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(sh.pay);                     /* [256] entry index of the k-th leaf */
    uint8_t *s_left = reinterpret_cast<uint8_t *>(sh.pay + 128);                /* [256] left turns on the way to the k-th leaf */
    uint32_t *s_cpart = sh.wtile;
    static_assert(sizeof(sh.wtile) >= 3 * WAVES * sizeof(uint32_t), "partials of the code scan");
    bool ok = true;
    /* the three aligned dwords that hold entries 2t .. 2t + 3 */
    uint32_t d0, d1, d2, mis;
    {
        const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
        mis = (uint32_t)(a & 3u);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(a - mis);
        const uint32_t nbytes = mis + 2u * (uint32_t)tree_len;
        const uint32_t t4 = 4u * (uint32_t)tid;
        d0 = (t4 < nbytes) ? q[tid] : 0u;
        d1 = (t4 + 4u < nbytes) ? q[tid + 1] : 0u;
        d2 = (t4 + 8u < nbytes) ? q[tid + 2] : 0u;
    }
    if (tid == 0) { sh.efflen = tree_len; sh.badsym = DEC_NO_BAD; sh.firstone = DEC_NO_BAD; sh.qend = 0; }     /* (what dec_build_tables sets) */
    bool l0, l1;
    int e0, e1;
    {
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
        __syncthreads();
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
            F::sym(sh)[k] = (uint8_t)e0;
            s_left[k] = (uint8_t)dmin<uint32_t>(2u * nb - (uint32_t)i0, 63u);
            k++;
        }
        if (l1 && k < 256u) {
            s_pos[k] = (uint16_t)(i0 + 1);
            F::sym(sh)[k] = (uint8_t)e1;
            s_left[k] = (uint8_t)dmin<uint32_t>(2u * (nb + (uint32_t)n0) - (uint32_t)(i0 + 1), 63u);
        }
    }
    __syncthreads();
    if (wave == 0) {                                                            /* the chain (scalar: every value the same in all lanes) */
        const uint32_t vl4 = 0xa0a0a0a0u - reinterpret_cast<const uint32_t *>(s_left)[lane];
        uint32_t vd = 0;
        uint32_t code = 0;
        const uint32_t nl = uni32((K + 3u) >> 2);
#pragma unroll 1
        for (uint32_t l = 0; l < nl; l++) {
            const uint32_t four = wave_lane_u32(vl4, l);
            uint32_t pack = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t lneg = ((four >> (8u * j)) & 0xffu) - 128u;
                const uint32_t sft = lneg - (uint32_t)__builtin_popcount(code);     /* 32 - depth */
                code += 1u << (sft & 31u);
                pack |= (sft & 0xffu) << (8u * j);
            }
            vd = ((uint32_t)lane == l) ? pack : vd;
        }
        uint32_t dd = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t shb = (vd >> (8u * j)) & 0xffu;
            dd |= (shb <= 30u ? 32u - shb : 33u) << (8u * j);
        }
        reinterpret_cast<uint32_t *>(F::len(sh))[lane] = dd;
    }
    __syncthreads();
    uint32_t d = 2;
    bool anylong;
    {
        uint32_t whi = 0, wlo = 0;
        if ((uint32_t)tid < K) {
            d = F::len(sh)[tid];
            if (d < 2u || d > 32u) { ok = false; d = 2; }
            else if (d >= 16u) wlo = 1u << (32u - d);
            else whi = 1u << (16u - d);
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
        if ((uint32_t)tid < K) F::code(sh)[tid] = (uint32_t)(base + (((uint64_t)(ihi - whi)) << 16) + (ilo - wlo));
    }
    __syncthreads();
    if ((uint32_t)tid < K) {                                                    /* the entry positions the depths imply are the stream's */
        const uint32_t k = (uint32_t)tid;
        const uint32_t bits = F::code(sh)[k] >> (32u - d);
        const uint32_t t = (uint32_t)__builtin_ctz(~bits);
        const uint32_t pos = s_pos[k];
        if (k == 0 && pos != d) ok = false;
        if (k + 1 < K) {
            const uint32_t dn = F::len(sh)[k + 1];
            if (dn + t < d || (uint32_t)s_pos[k + 1] != pos + 3u + (dn + t - d)) ok = false;
        } else if (pos + 4u != (uint32_t)tree_len) ok = false;
    }
    if (anylong && !REGS) ok = false;                                           /* (uniform: codes beyond the table take the walk's child links; decode_regs.hpp
                                                                                    finds such a code among the leaves kept here: sh.l2n, sh.fastk) */
    if (REGS && tid == 0) { sh.l2n = anylong ? 1u : 0u; sh.fastk = K; }
    if (!__syncthreads_and(ok ? 1 : 0)) return false;
    if constexpr (REGS) {
        /* Round 6, decode_regs.hpp's table: decode_sub's entry format (byte << 8 | length) over the 12 bits at a position.  A
         * first bit of 0: a leaf - every node below the root's only child has two children -, or DREG_E_LONG when the twelve
         * bits are not a whole code: a length of 0, to the pass "stand still" (round 6b; until then blocks with such codes
         * were not this path's).  A first bit of 1 leaves the tree: length = the run of ones (at most 12),
         * byte 0 - on a speculative track that is what puts it into step (tracks in a run of ones all land on the 0 behind
         * it), on the block's true track the first bits are checked (decode_regs.hpp). */
        static_assert((1 << DEC_LUT_BITS) == THREADS * 8, "eight entries per thread");
        const uint32_t *code = F::code(sh);
        const uint32_t x0 = (uint32_t)tid * 8u;
        uint32_t e[8];
        if (x0 < (1u << (DEC_LUT_BITS - 1))) {
            uint32_t k = dsub_leaf_of(code, K, x0 << (32 - DEC_LUT_BITS));
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t v = (x0 + j) << (32 - DEC_LUT_BITS);
                while (k + 1u < K && code[k + 1u] <= v) k++;
                const uint32_t lk = (uint32_t)F::len(sh)[k];
                e[j] = lk > (uint32_t)DEC_LUT_BITS ? DREG_E_LONG : ((uint32_t)F::sym(sh)[k] << 8) | lk;
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t v = (x0 + j) << (32 - DEC_LUT_BITS);
                e[j] = dmin<uint32_t>((uint32_t)__clz((int)~v), (uint32_t)DEC_LUT_BITS);
            }
        }
        *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        __syncthreads();
        return true;
    }
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
                /* the first bit leaves the tree (bits = 1), and so does every one bit that follows it at once (skip) */
                const uint32_t skip = dmin<uint32_t>((uint32_t)__clz((int)~v), (uint32_t)DEC_LUT_BITS);
                e[j] = DEC_E_BAD | DEC_E_NOCW | (skip << 8) | 1u;
            } else {
                while (k + 1u < K && code[k + 1u] <= v) k++;
                e[j] = ((uint32_t)F::len(sh)[k] << 8) | (uint32_t)F::sym(sh)[k];
            }
        }
        *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        __syncthreads();
        if (SPEC) {
            /* a failing run and the codeword behind it in ONE entry when both fit the window (dec_build_tables) */
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t ej = e[j];
                if (ej >= DEC_E_BAD && ej < DEC_E_LONG && (ej & 0x7fu) == 1u) {
                    const uint32_t run = dec_e_adv(ej);
                    const uint32_t idx = x0 + j;
                    const uint32_t e2 = sh.lut[(idx << run) & ((1u << DEC_LUT_BITS) - 1u)];
                    if (run < (uint32_t)DEC_LUT_BITS && e2 < DEC_E_BAD && run + (e2 >> 8) <= (uint32_t)DEC_LUT_BITS)
                        sh.lut[idx] = (uint16_t)(DEC_E_BAD | ((run + (e2 >> 8)) << 8) | 1u);
                }
            }
        }
    }
    __syncthreads();
    return true;
}

template <int THREADS, class BuildTables>
__device__ __forceinline__ int decode_payload_regs(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, uint64_t block_len,
                                                   uint8_t *gout, uint64_t *end_bits, uint64_t hint_bytes, BuildTables build_tables);      /* decode_regs.hpp */

/* decode_fast_kernel's arguments: ONE struct, so that the code behind decode_payload_regs can read them again from where they lie
 * (the kernel argument segment) instead of holding them in scalar registers through the whole decode - round 6b: at eight waves a
 * SIMD a wave has 80 of those, the register path's loop needs most, and what does not fit is brought back by v_readlane, a vector
 * instruction.  Only the fall-back paths and the list of blocks for the exact decoder need anything after the payload. */
struct DecodeFastArgs {
    const uint8_t *stream;
    uint64_t stream_len;
    const uint64_t *offsets;
    const HufDecodeMeta *dmeta;
    uint64_t *out_offsets;
    TwoLevel lens;
    uint8_t *out;
    uint64_t out_cap;
    int32_t *status;
    unsigned long long *result;
    DecFixList fix;
};

/* where block blk's tree and payload lie and where its bytes go */
struct DfastBlock {
    HufDecodeMeta m;
    uint64_t obase, pay_bytes, readable;
    const uint8_t *tree, *pay;
};
__device__ __forceinline__ DfastBlock dfast_locate(const DecodeFastArgs &a, uint64_t blk)
{
    DfastBlock b;
    b.m = a.dmeta[blk];
    b.m.block_len = uni64(b.m.block_len);
    b.m.tree_len = (int16_t)uni32((uint32_t)(uint16_t)b.m.tree_len);
    b.m.leaf = (int16_t)uni32((uint32_t)(uint16_t)b.m.leaf);
    b.m.status = (int32_t)uni32((uint32_t)b.m.status);
    b.obase = uni64(a.lens.gprefix[blk / SCAN_GROUP] + a.lens.local[blk]);
    const uint64_t o0 = uni64(a.offsets[blk]);
    const uint64_t o1 = dmin<uint64_t>(uni64(a.offsets[blk + 1]), a.stream_len);
    b.pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)b.m.tree_len);
    b.tree = a.stream + o0 + HUF_HEADER_FIXED;
    b.pay = b.tree + 2 * (int)b.m.tree_len;
    b.readable = a.stream_len - (uint64_t)(b.pay - a.stream);
    return b;
}

/* One indexed block (the body of decode_fast_kernel). */
template <int THREADS>
__device__ __forceinline__ void decode_fast_block(DecShared<THREADS> &sh, const DecodeFastArgs &a)
{
    const int tid = (int)threadIdx.x;
    int regs = 0;                                       /* (uniform) what decode_regs.hpp made of the block: 0 = it declined (another shape of tree, a short or a dull block) */
    {
        const uint64_t blk = blockIdx.x;
        const DfastBlock b = dfast_locate(a, blk);
        if (tid == 0) a.out_offsets[blk] = b.obase;
        if (b.m.status != HUFE_OK || b.m.block_len == 0) return;         /* header errors were recorded by decode_prepare */
        if (b.obase + b.m.block_len > a.out_cap) {
            if (tid == 0) {
                a.status[blk] = HUFE_MEMORY;
                atomicMin(&a.result[2], (unsigned long long)blk);
            }
            return;
        }
        if (b.m.leaf >= 0) {
            /* a block of one byte value: done here, with what is at hand (reading the arguments again costs such a block - which is
             * nothing but this function - a third of its time: const41 0.24 -> 0.32 ms) */
            uint64_t eb = 0, produced = 0;
            const bool fine = decode_single_leaf<THREADS, true>(sh, (uint32_t)b.m.leaf, b.pay, b.m.block_len, b.pay_bytes, a.out + b.obase, &eb, &produced) == HUFE_OK;
            if (!fine && tid == 0) {
                if (atomicExch(&a.fix.flag[blk], 1u) == 0u) a.fix.blocks[atomicAdd(a.fix.count, 1u)] = (uint32_t)blk;
            }
            return;
        }
#ifndef DFAST_NO_REGS
        if (b.m.leaf < 0 && b.m.block_len >= DREG_MIN_BLOCK)
            regs = decode_payload_regs<THREADS>(sh, b.pay, b.pay_bytes, b.readable, b.m.block_len, a.out + b.obase, nullptr, 0,
                                                [&]() { return dfast_tables_from_tree<THREADS, true, true>(sh, b.tree, b.m.tree_len) ? (uni32(sh.l2n) != 0u ? 2 : 1) : 0; });
        if (regs == 1) return;                          /* (DREG_OK: most blocks of most streams end here) */
#endif
    }
    /* Everything from here on reads the arguments AGAIN, through a pointer the compiler cannot see through: nothing of the above
     * is held in registers across the payload for it. */
    const DecodeFastArgs *again = (const DecodeFastArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(again));
    const DecodeFastArgs &r = *again;
    const uint64_t blk = blockIdx.x;
    bool good;
    if (regs != 0) {
        good = false;                                   /* (DREG_FAILED) */
    } else {
        const DfastBlock b = dfast_locate(r, blk);
        int leaf = b.m.leaf;
        int rc = HUFE_OK;
        unsigned long long kt = DPROF_T();
        /* (dfast_tables_from_tree from 32 KiB of symbols on: the chain is a latency - 3 us a block - that four workgroups per CU hide
         *  next to a long payload and not next to a short one.  1 GiB in 64 KiB blocks: zipf255 1.72 -> 1.70 ms, uniform bytes 1.11 -> 1.05,
         *  log text 1.75 -> 1.68; in 16 KiB blocks 3.2 -> 3.6, in 4 KiB blocks 12.7 -> 15.2 with it) */
#ifndef DFAST_NO_REGS
        if (leaf < 0) rc = dec_build_tables<THREADS, true>(sh, b.tree, b.m.tree_len, &leaf);
#else
        if (leaf < 0 && !(b.m.block_len >= 32768u && dfast_tables_from_tree<THREADS, true>(sh, b.tree, b.m.tree_len)))
            rc = dec_build_tables<THREADS, true>(sh, b.tree, b.m.tree_len, &leaf);
#endif
        DPROF_ADD(6, kt);
        if (rc != HUFE_OK) {
            good = false;
        } else if (leaf >= 0) {
            uint64_t eb = 0, produced = 0;
            good = decode_single_leaf<THREADS, true>(sh, (uint32_t)leaf, b.pay, b.m.block_len, b.pay_bytes, r.out + b.obase, &eb, &produced) == HUFE_OK;
        } else {
            good = decode_payload_dfast<THREADS>(sh, b.pay, b.pay_bytes, b.readable, b.m.block_len, r.out + b.obase);
        }
    }
    if (!good && tid == 0) {
        if (atomicExch(&r.fix.flag[blk], 1u) == 0u) r.fix.blocks[atomicAdd(r.fix.count, 1u)] = (uint32_t)blk;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void decode_fast_kernel(DecodeFastArgs a)
{
    __shared__ DecShared<THREADS> sh;
    decode_fast_block<THREADS>(sh, a);
}

}  // namespace hufgpu
