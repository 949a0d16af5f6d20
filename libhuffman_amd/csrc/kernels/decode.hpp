/* decode.hpp - decode_prepare_kernel, decode_kernel, decode_chain_kernel (src/decoder.c:34-96, 218-276; src/tree.c:138-227).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "offsets.hpp"

namespace hufgpu {

/* ======================================================================================
 * header parse of src/decoder.c:218-252 for every indexed block + output offsets
 * ==================================================================================== */
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p)
{
    /* (one 8-byte load: global loads need no alignment on gfx950, and the stream is little-endian like the device) */
    typedef uint64_t __attribute__((aligned(1))) unaligned_u64;
    return *reinterpret_cast<const unaligned_u64 *>(p);
}

/* first 10 bytes of a block header at stream + o0, by aligned 32-bit loads (the words that hold
 * at least one stream byte are readable) */
__device__ __forceinline__ void load_header10(const uint8_t *stream, uint64_t stream_len, uint64_t o0,
                                              uint64_t &block_len, int16_t &tree_len)
{
    const uintptr_t a = (uintptr_t)(stream + o0);
    const uint32_t m = (uint32_t)(a & 3u);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a - m);
    const uintptr_t end = (uintptr_t)(stream + stream_len);
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = ((uintptr_t)(q + k) < end) ? q[k] : 0u;
    const uint32_t sh = 8u * m;
    uint32_t d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = m ? ((w[k] >> sh) | (w[k + 1] << (32u - sh))) : w[k];
    block_len = (uint64_t)d[0] | ((uint64_t)d[1] << 32);
    tree_len = (int16_t)(d[2] & 0xffffu);
}

/* The tree every one-symbol block carries, [root, leaf, -1, -1, -1]: the
 * leaf's byte, or -1 for any other five entries (those go through the general tree build). */
__device__ __forceinline__ int single_leaf_symbol(const uint8_t *tree)
{
    int16_t e5[5];
#pragma unroll
    for (int i = 0; i < 5; i++) e5[i] = (int16_t)((uint16_t)tree[2 * i] | ((uint16_t)tree[2 * i + 1] << 8));
    const bool one = e5[0] != -1 && e5[1] != -1 && e5[2] == -1 && e5[3] == -1 && e5[4] == -1;
    return one ? (int)(uint8_t)e5[1] : -1;
}

/* decode_prepare_kernel - header parse of src/decoder.c:218-252 for every indexed block, one block
 * per thread, and the sums of the block lengths (= where each block's output starts) as a
 * two-level prefix: a workgroup is one SCAN_GROUP.  result words: [0] unused, [1] total raw
 * bytes (sum of block_len), [2] first failing block in stream order (~0 = none), [3] unused.
 * Word [2] is written here (minimum over the header errors) and lowered with atomicMin by the
 * decode kernel: first error in stream order wins, like the reference's abort. */
__global__ __launch_bounds__(SCAN_GROUP) void decode_prepare_kernel(const uint8_t *__restrict__ stream,
                                                                    uint64_t stream_len,
                                                                    const uint64_t *__restrict__ offsets,
                                                                    uint64_t nblocks, int max_tree_len,
                                                                    HufDecodeMeta *__restrict__ dmeta,
                                                                    int32_t *__restrict__ status, TwoLevel lens,
                                                                    uint32_t *__restrict__ fix_count)
{
    __shared__ uint64_t s_part[SCAN_GROUP / 64];
    __shared__ unsigned long long s_bad;
    if (threadIdx.x == 0) s_bad = ~0ull;
    if (fix_count && blockIdx.x == 0 && threadIdx.x == 0) { fix_count[0] = 0; fix_count[1] = 0; }   /* the work list of decode_sub / decode_fast starts empty */
    const uint64_t b = (uint64_t)blockIdx.x * SCAN_GROUP + threadIdx.x;
    HufDecodeMeta m;
    m.block_len = 0;
    m.tree_len = 0;
    m.leaf = -1;
    m.status = HUFE_OK;
    if (b < nblocks) {
        const uint64_t o0 = offsets[b];
        const uint64_t o1 = dmin<uint64_t>(offsets[b + 1], stream_len);
        if (o0 > o1 || o1 - o0 < HUF_HEADER_FIXED) {
            m.status = HUFE_RW;                                /* decoder.c:220-234 short read */
        } else {
            uint64_t bl;
            int16_t tl;
            load_header10(stream, stream_len, o0, bl, tl);
            if (tl < 0 || tl > max_tree_len) m.status = HUFE_OVERFLOW;          /* decoder.c:237-239 */
            else if (o1 - o0 < HUF_HEADER_FIXED + 2ull * (uint64_t)tl) m.status = HUFE_RW;   /* :248-252 */
            else {
                /* A block cannot hold more symbols than its payload has bits: a larger block_len (a
                 * damaged header) is decoded as far as the input goes and then fails like the
                 * reference's reader does at the end of its input (decoder.c:53-56). */
                const uint64_t pay_bits = (o1 - o0 - HUF_HEADER_FIXED - 2ull * (uint64_t)tl) * 8ull;
                if (bl > pay_bits) bl = pay_bits + 1;
                if (bl > HUF_MAX_BLOCK_LEN) m.status = HUFE_ARGUMENT;            /* beyond kernel limits */
                else {
                    m.block_len = bl;
                    m.tree_len = tl;
                    if (tl == 5) m.leaf = (int16_t)single_leaf_symbol(stream + o0 + HUF_HEADER_FIXED);
                }
            }
        }
        dmeta[b] = m;
        status[b] = m.status;
    }
    __syncthreads();
    if (m.status != HUFE_OK) atomicMin(&s_bad, (unsigned long long)b);
    uint64_t total;
    const uint64_t ex = block_excl_scan<SCAN_GROUP, uint64_t>(m.block_len, s_part, total);
    if (b < nblocks) lens.local[b] = ex;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (threadIdx.x == 0) {
        handover_store(lens.gsum + blockIdx.x, total);
        handover_store(lens.gmin + blockIdx.x, s_bad);
    }
    two_level_finish(lens, gridDim.x);
}

/* ======================================================================================
 * decode - replaces huf_tree_deserialize (src/tree.c:138-227) and __huf_decode_block
 * (src/decoder.c:34-96).
 *
 * One workgroup per block.
 *  1. The serialized tree is turned into child arrays in parallel.  With S(i) = number of
 *     open child slots before entry i (S(0) = 1, +1 after a node entry, -1 after a -1
 *     marker), entry j+1 is the left child of node j and the first later entry with the same
 *     S as j is its right child; entries after S reaches 0, or past the buffer, do not exist
 *     (tree.c:152-160: a missing entry is a NULL child).
 *  2. Trees whose root has a single leaf child on the left (every block of one distinct byte,
 *     e.g. BASELINE config 2) need no table: every symbol is one 0 bit, a 1 bit leaves the
 *     tree.  The payload is checked for a set bit and the output is a fill.
 *  3. Otherwise a 2^LUT_BITS-entry table in LDS maps the next LUT_BITS stream bits to
 *     {leaf, length}, {inner node to continue the bit walk from} or {walk left the tree}.
 *  4. The payload is processed in segments of THREADS x 128 bits, staged in LDS as big-endian
 *     words in a [word-in-subsequence][lane] layout (lane-consecutive = bank-consecutive).
 *     Every lane decodes one 128-bit subsequence through a 64-bit register bit buffer (one
 *     LDS word per 32 bits consumed + one table read per symbol).  Only lane 0 knows where
 *     its first codeword starts; the others start at their subsequence boundary and keep a
 *     128-bit map of the codeword starts they found.  Each round a lane whose left neighbour
 *     reported a different end position re-decodes from there only until it lands on a
 *     codeword start it already knows (the tracks have merged; counts follow from popcounts),
 *     until no start changes any more (self-synchronisation; exact for any stream, worst case
 *     one lane per round).  Symbol counts are prefix-summed and the symbols are decoded once
 *     more straight into HBM (32-bit stores, bytes at the edges).  Exactly block_len symbols
 *     are produced; pad bits are ignored (decoder.c:89-91).
 * ==================================================================================== */
#define DEC_LUT_BITS 12
#ifndef DEC_SUB_WORDS
#define DEC_SUB_WORDS 8               /* 32-bit words per lane subsequence (power of two) */
#endif
#define DEC_SUB_BITS (32 * DEC_SUB_WORDS)
#define DEC_NULL 0xffffu
#define DEC_LEAF_LR 0xffffffffu       /* lr[] of a childless node */
#define DEC_XCOLS ((40 + DEC_SUB_WORDS - 1) / DEC_SUB_WORDS + 1)   /* >= 40 extra words > (1025 + LUT_BITS)/32: deepest bit walk */
#define DEC_EXH 0xffffffffu           /* "a codeword ran past the readable payload" */
#define DEC_NO_BAD 0xffffffffu

#ifdef DEC_PHASE_PROF
__device__ unsigned long long g_dec_prof[16];
#define DPROF_T() (__builtin_readcyclecounter())
#define DPROF_ADD(slot, t0) do { if (threadIdx.x == 0) atomicAdd(&g_dec_prof[slot], (unsigned long long)(__builtin_readcyclecounter() - (t0))); } while (0)
#else
#define DPROF_T() 0ull
#define DPROF_ADD(slot, t0) do { (void)(t0); } while (0)
#endif

template <int THREADS>
struct DecShared {
    static constexpr int ENT = HUF_TREE_MAX + 1;
    static constexpr int COLS = THREADS + DEC_XCOLS;
    uint16_t lut[1 << DEC_LUT_BITS];
    __attribute__((aligned(16))) uint32_t pay[DEC_SUB_WORDS * COLS];  /* segment word i at pay[(i % W) * COLS + i / W]: lane-consecutive = bank-consecutive
                                            (a padded linear layout has a cheaper address but costs 2 KiB = one workgroup per CU) */
    uint16_t mark[DEC_SUB_WORDS][THREADS];  /* (codewords before << 5 | offset) of lane l's first visit to each word */
    int16_t ent[ENT];                    /* the tree's entries (behind the marks: decode_fast_kernel's table of pairs runs on into it when the
                                            block has no codes that need the entries, decode_fast.hpp) */
    __attribute__((aligned(4))) uint32_t lr[ENT];   /* children of entry i: left in the low half, right in the high half, DEC_NULL = none.  (Behind `ent`
                                            since round 5: blocks without `long` codes need neither after the tables are built, and decode_fast's
                                            column stage + table of pairs run on into both.) */
    __attribute__((aligned(16))) uint32_t wend[THREADS / 64];   /* end position of the last lane of each wave (neighbours use shuffles).  (Aligned: `ent` in
                                            front of it is not a multiple of 16 bytes, and decode_sub reads wtile[] sixteen bytes at a time.) */
    uint32_t part[THREADS / 64];
    uint32_t wtile[32];                  /* decode_sub: payload bits of the chunk's wave tiles */
    uint32_t fastk;                      /* decode_sub: leaves of the tables dsub_fast_tables built (0: the tables are dec_build_tables') */
    uint32_t l2n;                        /* decode_sub: entries of the second-level table in `ent` (0: none) */
    int efflen;
    uint32_t badsym;                     /* segment symbol index of the first walk that left the tree */
    uint32_t firstone;                   /* single-leaf trees: first set payload bit */
    uint32_t qend;                       /* segment bit right after the block's last symbol */
    uint32_t leaves[320];                /* decode_regs.hpp: the leaves' codes in preorder (256 words) and their bytes (64), kept for the block's codes
                                            beyond the table (the struct with them: 40 880 of the 40 960 bytes a workgroup has at four a CU) */
};

/* big-endian 32-bit word of payload bytes [off, off+4), zero beyond nbytes */
__device__ __forceinline__ uint32_t load_be32(const uint8_t *pay, uint64_t off, uint64_t nbytes)
{
    if (off >= nbytes) return 0;
    const uint64_t remain = nbytes - off;
    const uintptr_t a = (uintptr_t)(pay + off);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t m = (uint32_t)(a & 3u);
    uint32_t v = q[0];
    if (m) {
        const uint32_t hi = (remain > 4u - m) ? q[1] : 0u;
        v = (v >> (8 * m)) | (hi << (32 - 8 * m));
    }
    v = __builtin_bswap32(v);
    if (remain < 4) v &= 0xffffffffu << (8 * (4 - (uint32_t)remain));
    return v;
}

/* Two-word MSB-first window over the staged segment: w0 = word g, w1 = word g+1.  A table
 * codeword is at most DEC_LUT_BITS long, so after it the position is in word g or g+1. */
template <int COLS>
__device__ __forceinline__ uint32_t pay_slot(uint32_t i) { return (i & (DEC_SUB_WORDS - 1)) * COLS + i / DEC_SUB_WORDS; }

/* Two-word MSB-first window over the staged segment, kept so that a symbol costs as few vector
 * instructions as possible (the decode kernel is bound by VALU issue, 4 cycles per wave64
 * instruction): the pair is held delayed, {d0,d1} = {word g, word g+1} >> 20, and the position
 * inside word g as s = 31 - (pos & 31).  Then ONE v_alignbit_b32 by s (shift amounts 0..31, no
 * 64-bit shift, no special case at a word start) puts the 12 bits at the position at bits 1..12
 * of its result - masked, that is the byte offset of their table entry - a codeword of len bits
 * is s -= len, and s < 0 says "moved into word g + 1". */
template <int COLS>
struct BitReader {
    const uint32_t *pay;
    static constexpr uint32_t DELAY = 32 - DEC_LUT_BITS;   /* 20 */
    uint32_t d0, d1;     /* {word g, word g+1} >> DELAY */
    uint32_t wl;         /* word g + 1 as staged */
    uint32_t r;          /* g relative to the first word of the lane's subsequence (0..DEC_SUB_WORDS-1) */
    uint32_t waddr;      /* LDS byte offset of word g + 1 inside `pay` */
    int32_t s;

    __device__ __forceinline__ uint32_t word(uint32_t i) const { return pay[pay_slot<COLS>(i)]; }
    /* sub_w0 = first word of the lane's subsequence (a multiple of DEC_SUB_WORDS) */
    __device__ __forceinline__ void load(uint32_t pos, uint32_t sub_w0)
    {
        const uint32_t g = pos >> 5;
        r = g - sub_w0;
        s = (int32_t)(31u - (pos & 31u));
        const uint32_t w0 = word(g);
        waddr = 4u * pay_slot<COLS>(g + 1);
        wl = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(pay) + waddr);
        d0 = w0 >> DELAY;
        d1 = __builtin_amdgcn_alignbit(w0, wl, DELAY);
    }
    /* byte offset of the table entry for the DEC_LUT_BITS bits at the position */
    __device__ __forceinline__ uint32_t lut_offset() const
    {
        return __builtin_amdgcn_alignbit(d0, d1, (uint32_t)s) & (((1u << DEC_LUT_BITS) - 1u) << 1);
    }
    __device__ __forceinline__ uint32_t pos(uint32_t sub_w0) const { return ((r + sub_w0) << 5) + (31u - (uint32_t)s); }
    /* The position moved into word g + 1 (s is back in 0..31); only called while r + 1 <
     * DEC_SUB_WORDS.  The staged layout puts word i at (i % W) * COLS + i / W, so the lane's own
     * words are COLS apart and the first word of the next lane's subsequence, the only other one
     * ever appended here, sits one slot behind the lane's first word. */
    __device__ __forceinline__ void step_next(uint32_t wrap_addr)
    {
        r++;
        waddr += 4u * COLS;
        if (r == DEC_SUB_WORDS - 1) waddr = wrap_addr;
        const uint32_t wn = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(pay) + waddr);
        d0 = wl >> DELAY;
        d1 = __builtin_amdgcn_alignbit(wl, wn, DELAY);
        wl = wn;
    }
};

/* Buffered reader: up to 64 payload bits left-aligned in a register pair.  The table index is one
 * shift of the high half, a codeword is one 64-bit shift, and - what it is for - a refill is
 * only ever needed every SECOND codeword (a refill leaves >= 33 bits, two table codewords take
 * <= 24).  A wave executes the word-change code whenever ANY of its lanes crosses a word, i.e.
 * practically every iteration, so halving how often that code runs is worth more than anything
 * inside the per-codeword path (issue cost, MI355X, 8 waves/SIMD, tools/calib: simple VOP2 ~2.5
 * cycles, VOP3 / v_cmp ~4.5, scalar ~4.5). */
template <int COLS>
struct BufReader {
    const uint32_t *pay;
    uint32_t hi, lo;     /* bit buffer: the next stream bit is bit 31 of hi; bits past `avail` are 0 */
    int32_t avail;       /* valid bits */
    uint32_t gf;         /* staged word that the next refill appends */

    __device__ __forceinline__ uint32_t word(uint32_t i) const { return pay[pay_slot<COLS>(i)]; }
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

enum { CW_OK = 0, CW_BAD = 1, CW_EXH = 2 };

__device__ __forceinline__ uint32_t dec_child(uint32_t lr, uint32_t bit) { return bit ? (lr >> 16) : (lr & 0xffffu); }

/* Table entries (uint16):
 *   leaf    (len << 8) | symbol                     len = 1..DEC_LUT_BITS
 *   bad     0x4000 | nocw | (skip << 8) | bits      the walk leaves the tree at bit `bits` of the
 *                                                   window; a speculative track resumes `skip` (5 bits)
 *                                                   bits on; nocw (0x2000) is clear when skip also
 *                                                   covers the codeword that follows the failing run
 *   long    0xC000 | node                           still inside the tree after DEC_LUT_BITS bits
 * so bits 8..13 are "advance by" for leaf and bad alike.  Every code of an encoder-made tree
 * starts with 0 (the wrap root has no right child, src/tree.c:410-413), so a lane that starts
 * its subsequence in the middle of a codeword runs into `bad` all the time until it has
 * synchronised: that path has to be as cheap as a symbol, and `skip` jumps over a whole run of
 * bits that would fail the same way (bits == 1: the run of equal leading bits). */
#define DEC_E_BAD  0x4000u
#define DEC_E_LONG 0xC000u
#define DEC_E_NOCW 0x2000u           /* in a `bad` entry: no codeword was taken (the lookup does not count as one) */
__device__ __forceinline__ uint32_t dec_e_adv(uint32_t e) { return (e >> 8) & 0x1fu; }

/* Bit-serial walk for `long` entries (and the verdict of a `bad` one), on the staged words.
 * CW_OK: sym, npos = position after the codeword.  CW_BAD: the walk left the tree, npos =
 * position after the failing bit.  CW_EXH: the walk needs bits past the readable payload.
 * Result packed in registers (no stack): bits 0-31 npos, 32-39 sym, 40-41 status. */
template <int THREADS>
__device__ __forceinline__ uint64_t dec_rare_packed(const DecShared<THREADS> &sh, uint32_t e, uint32_t pos, uint32_t pay_rel)
{
    if (e < DEC_E_LONG)                          /* the table walk already left the tree */
        return ((uint64_t)CW_BAD << 40) | (uint64_t)(pos + (e & 0x7fu));
    uint32_t node = e & 0x7ffu;
    uint32_t p = pos + DEC_LUT_BITS;
    for (;;) {
        if (p >= pay_rel) return (uint64_t)CW_EXH << 40;
        const uint32_t w = sh.pay[pay_slot<DecShared<THREADS>::COLS>(p >> 5)];
        const uint32_t bit = (w >> (31u - (p & 31u))) & 1u;
        p++;
        const uint32_t nx = dec_child(sh.lr[node], bit);
        if (nx == DEC_NULL) return ((uint64_t)CW_BAD << 40) | p;
        node = nx;
        if (sh.lr[node] == DEC_LEAF_LR) break;
    }
    return ((uint64_t)CW_OK << 40) | ((uint64_t)(uint8_t)sh.ent[node] << 32) | p;
}

/* Per-lane decode state that survives the synchronisation rounds.  The lane's track is also
 * summarised in sh.mark: for every 32-bit word of the subsequence, where the track first
 * visited it and how many codewords it had decoded before that visit. */
struct LaneTrack {
    uint32_t start;    /* first codeword of this lane (segment bits) */
    uint32_t end;      /* first codeword at/after the lane's limit, or DEC_EXH */
    uint32_t cnt;      /* codewords that start inside the lane's subsequence */
    int32_t lastbad;   /* last word of the subsequence in which a walk left the tree, -1 = none */
};

#define DEC_NO_MARK 0xffffu
__device__ __forceinline__ uint16_t dec_mark(uint32_t count, uint32_t pos) { return (uint16_t)((count << 5) | (pos & 31u)); }

/* Count pass.  A track is the sequence of positions the decoder visits from `start` (a walk
 * that leaves the tree resumes a bit - or a run of such bits - later; only speculative starts
 * ever do that on a valid stream).
 * MERGE = false: decode everything.  MERGE = true: tr/sh.mark describe the lane's previous
 * track; decode from the new `start` only until the new track enters a word at exactly the
 * position where the previous track entered it - from there on the two are identical, so the
 * old end stays valid and the counts differ by a constant.
 * CHECK = false when no table codeword that starts before the lane's limit can reach the end of
 * the readable payload (all lanes but one or two per block): no per-symbol bound test, and a
 * `bad` entry costs two extra instructions.
 * The common iteration is v_alignbit, 2 x index, table read, special test, s -= advance, sign
 * test.  The subsequence limit is only looked at on a word change (it is word aligned), and the
 * codeword count is the wave-uniform iteration count minus the lane's non-codeword lookups.
 * Of the walks that left the tree only the word of the LAST one is remembered (enough to tell,
 * after a merge, whether the surviving part of the old track had one); the exact first one of
 * the final track is searched afterwards, by dec_first_bad, on corrupt streams only. */
template <int THREADS, bool MERGE, bool CHECK>
__device__ __forceinline__ void dec_scan_impl(DecShared<THREADS> &sh, LaneTrack &tr, uint32_t start,
                                              uint32_t sub_lo, uint32_t pay_rel)
{
    const int tid = (int)threadIdx.x;
    const uint32_t limit = sub_lo + DEC_SUB_BITS;
    const uint32_t sub_w0 = sub_lo >> 5;
    constexpr uint32_t DONE = 0x1000u;   /* rd.r of a lane that has left the loop (the loop's only exit test) */
    uint32_t c = 0, pos = start;
    int32_t nlast = -1;           /* like LaneTrack::lastbad, for the part decoded here */
    uint32_t lw = DEC_SUB_WORDS;  /* word of the latest mark; DEC_SUB_WORDS = none written */
    uint32_t old_c = 0;
    bool merged = false;
    if (pos < limit) {
        BitReader<DecShared<THREADS>::COLS> rd;
        rd.pay = sh.pay;
        rd.load(pos, sub_w0);
        lw = rd.r;
        for (uint32_t k = 0; k < lw; k++) sh.mark[k][tid] = DEC_NO_MARK;     /* nothing visits these */
        if (MERGE) {
            const uint32_t old = sh.mark[lw][tid];
            if (old != DEC_NO_MARK && (old & 31u) == (pos & 31u)) { merged = true; old_c = old >> 5; }
        }
        if (!merged) {
            uint16_t *mk = &sh.mark[lw][tid];                    /* mark of the current word */
            *mk = dec_mark(0, pos);
            /* word DEC_SUB_WORDS of the subsequence = first word of the next lane's */
            const uint32_t wrap_addr = 4u * pay_slot<DecShared<THREADS>::COLS>(sub_w0 + DEC_SUB_WORDS);
            uint32_t it = 0;      /* table lookups done: the same in every lane that is still in the loop (an SGPR) */
            uint32_t miss = 0;    /* lookups of this lane that were not codewords */
            do {
                uint32_t e = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(sh.lut) + rd.lut_offset());
                bool slow = e >= DEC_E_LONG;
                if (CHECK) slow = (e >= DEC_E_BAD) || (rd.pos(sub_w0) + dec_e_adv(e) > pay_rel);
                if (__builtin_expect(__ballot(e >= DEC_E_BAD || slow) != 0ull, 0)) {
                    if (slow) {
                        const uint32_t p = rd.pos(sub_w0);
                        uint32_t np = DEC_EXH;                        /* needs bits past the payload (decoder.c:53-56) */
                        bool codeword = false;
                        if (e >= DEC_E_LONG) {
                            const uint64_t r = dec_rare_packed<THREADS>(sh, e, p, pay_rel);
                            const uint32_t npos = (uint32_t)r;
                            const int st = (int)(r >> 40);
                            if (st == CW_OK && npos <= pay_rel) { np = npos; codeword = true; }
                            else if (st == CW_BAD && npos <= pay_rel) { nlast = (int32_t)rd.r; np = p + 1; }
                        } else if (CHECK && e >= DEC_E_BAD) {
                            if (p + (e & 0x7fu) <= pay_rel) { nlast = (int32_t)rd.r; np = p + 1; }   /* a real payload bit left the tree */
                        }
                        if (!codeword) miss++;
                        e = 0;                                        /* the common part has nothing left to do */
                        if (np >= limit) { pos = np; c = it + 1 - miss; lw = rd.r; rd.r = DONE; }
                        else {
                            const uint32_t nr = (np >> 5) - sub_w0;
                            bool stop = false;
                            if (nr != rd.r) {                         /* words a long walk jumps over are never visited */
                                for (uint32_t k = rd.r + 1; k < nr; k++) sh.mark[k][tid] = DEC_NO_MARK;
                                mk = &sh.mark[nr][tid];
                                const uint32_t old = *mk;
                                if (MERGE && old != DEC_NO_MARK && (old & 31u) == (np & 31u)) {
                                    merged = true; old_c = old >> 5; pos = np; c = it + 1 - miss; lw = nr; stop = true;
                                } else *mk = dec_mark(it + 1 - miss, np);
                            }
                            if (stop) rd.r = DONE;
                            else rd.load(np, sub_w0);
                        }
                    } else if (e >= DEC_E_BAD) {                      /* left the tree: resume after the run */
                        nlast = (int32_t)rd.r;                        /* words only grow: the latest is the last */
                        miss += (e >> 13) & 1u;                       /* DEC_E_NOCW: not a codeword */
                    }
                }
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(it) : : "scc");
                rd.s -= (int32_t)dec_e_adv(e);
                if (rd.s < 0) {                                       /* the track enters the next word (never after the slow path: it advanced by 0) */
                    rd.s += 32;
                    const uint32_t off = 31u - (uint32_t)rd.s;
                    if (rd.r == DEC_SUB_WORDS - 1) {                  /* ... which is past the lane's limit */
                        pos = limit + off; c = it - miss; lw = DEC_SUB_WORDS - 1; rd.r = DONE;
                    } else {
                        mk += THREADS;                                /* &sh.mark[r + 1][tid] */
                        const uint32_t old = MERGE ? (uint32_t)*mk : (uint32_t)DEC_NO_MARK;
                        if (MERGE && old != DEC_NO_MARK && (old & 31u) == off) {
                            merged = true; old_c = old >> 5; lw = rd.r + 1;
                            pos = ((sub_w0 + lw) << 5) + off; c = it - miss; rd.r = DONE;
                        } else {
                            *mk = (uint16_t)(((it - miss) << 5) | off);
                            rd.step_next(wrap_addr);
                        }
                    }
                }
            } while (rd.r != DONE);
        }
    }
    if (MERGE && merged) {
        /* identical from `pos` on: later marks keep their positions, their counts shift; what the
         * old track met from word lw on, the new one meets too */
        const uint32_t delta = c - old_c;                      /* modulo 2^32, may be "negative" */
        for (uint32_t k = lw + 1; k < DEC_SUB_WORDS; k++) {
            const uint32_t r = sh.mark[k][tid];
            if (r != DEC_NO_MARK) sh.mark[k][tid] = (uint16_t)(r + (delta << 5));
        }
        sh.mark[lw][tid] = dec_mark(c, pos);
        if (tr.lastbad < (int32_t)lw) tr.lastbad = nlast;      /* the old track's events before word lw are gone */
        tr.cnt += delta;
        /* tr.end unchanged */
    } else {
        for (uint32_t k = (lw == DEC_SUB_WORDS ? 0u : lw + 1); k < DEC_SUB_WORDS; k++) sh.mark[k][tid] = DEC_NO_MARK;
        tr.cnt = c;
        tr.end = pos;
        tr.lastbad = nlast;
    }
    tr.start = start;
}

/* The first count pass of a segment (every lane starts in the first word of its subsequence, no
 * previous track to merge with, no bound checks), organised BY WORD: for each of the lane's
 * words, an inner loop decodes while the position is still inside that word, then ALL lanes
 * change word together.  In dec_scan_impl a wave runs the word-change code whenever any of its 64
 * lanes crosses a word - every iteration, 17 of the 34 VALU instructions of an iteration - here
 * it runs 8 times per subsequence; the price is that the wave waits per word for the lane with
 * the most codewords in it (53 inner iterations instead of 42 on Zipf data, simulated). */
template <int THREADS>
__device__ __forceinline__ void dec_scan_words(DecShared<THREADS> &sh, LaneTrack &tr, uint32_t start,
                                               uint32_t sub_lo, uint32_t pay_rel)
{
    const int tid = (int)threadIdx.x;
    const uint32_t limit = sub_lo + DEC_SUB_BITS;
    const uint32_t sub_w0 = sub_lo >> 5;
    constexpr uint32_t DONE = 0x1000u;
    BitReader<DecShared<THREADS>::COLS> rd;
    rd.pay = sh.pay;
    rd.load(start, sub_w0);                                   /* rd.r == 0 */
    uint16_t *mk = &sh.mark[0][tid];
    *mk = dec_mark(0, start);
    const uint32_t wrap_addr = 4u * pay_slot<DecShared<THREADS>::COLS>(sub_w0 + DEC_SUB_WORDS);
    uint32_t c = 0;               /* codewords decoded so far */
    uint32_t pos = 0, lw = DEC_SUB_WORDS - 1;
    int32_t nlast = -1;
#pragma unroll 1
    for (uint32_t r = 0; r < DEC_SUB_WORDS; r++) {
        int32_t s_keep = 0;
        if (rd.r == r) {          /* not the lanes that a long codeword carried past this word, or out */
            while (rd.s >= 0) {
                uint32_t e = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(sh.lut) + rd.lut_offset());
                if (__builtin_expect(__ballot(e >= DEC_E_BAD) != 0ull, 0)) {
                    if (e >= DEC_E_LONG) {
                        const uint32_t p = rd.pos(sub_w0);
                        const uint64_t rr = dec_rare_packed<THREADS>(sh, e, p, pay_rel);
                        const uint32_t npos = (uint32_t)rr;
                        const int st = (int)(rr >> 40);
                        uint32_t np = DEC_EXH;                        /* needs bits past the payload (decoder.c:53-56) */
                        if (st == CW_OK && npos <= pay_rel) np = npos;
                        else {
                            if (st == CW_BAD && npos <= pay_rel) { nlast = (int32_t)r; np = p + 1; }
                            c--;                                      /* not a codeword: undo the count below */
                        }
                        e = 0;
                        if (np >= limit) { pos = np; lw = r; rd.r = DONE; rd.s = -1; }
                        else {
                            const uint32_t nr = (np >> 5) - sub_w0;
                            rd.load(np, sub_w0);
                            if (nr != r) {                            /* words a long walk jumps over are never visited */
                                for (uint32_t k = r + 1; k < nr; k++) sh.mark[k][tid] = DEC_NO_MARK;
                                mk = &sh.mark[nr][tid];
                                *mk = dec_mark(c + 1, np);
                                s_keep = rd.s;                        /* resumes when the word loop gets there */
                                rd.s = -1;
                            }
                        }
                    } else if (e >= DEC_E_BAD) {                      /* left the tree: resume after the run */
                        nlast = (int32_t)r;
                        c -= (e >> 13) & 1u;                          /* DEC_E_NOCW: not a codeword */
                    }
                }
                c++;
                rd.s -= (int32_t)dec_e_adv(e);
            }
        }
        /* every lane that is still in word r has crossed into word r + 1 */
        if (rd.r == r) {
            rd.s += 32;
            const uint32_t off = 31u - (uint32_t)rd.s;
            if (r == DEC_SUB_WORDS - 1) { pos = limit + off; rd.r = DONE; }
            else {
                mk += THREADS;
                *mk = (uint16_t)((c << 5) | off);
                rd.step_next(wrap_addr);
            }
        } else if (rd.s < 0 && rd.r != DONE) rd.s = s_keep;            /* jumped ahead in this word */
    }
    for (uint32_t k = lw + 1; k < DEC_SUB_WORDS; k++) sh.mark[k][tid] = DEC_NO_MARK;
    tr.cnt = c;
    tr.end = pos;
    tr.lastbad = nlast;
    tr.start = start;
}

/* Lanes near the end of the payload (one or two per block) take the bound-checked loop. */
template <int THREADS, bool MERGE>
__device__ __forceinline__ void dec_scan(DecShared<THREADS> &sh, LaneTrack &tr, uint32_t start,
                                         uint32_t sub_lo, uint32_t pay_rel)
{
    if (sub_lo + DEC_SUB_BITS + DEC_LUT_BITS <= pay_rel) {
        if (!MERGE && start - sub_lo < 32u) dec_scan_words<THREADS>(sh, tr, start, sub_lo, pay_rel);
        else
            dec_scan_impl<THREADS, MERGE, false>(sh, tr, start, sub_lo, pay_rel);
    } else dec_scan_impl<THREADS, MERGE, true>(sh, tr, start, sub_lo, pay_rel);
}

/* Codewords a (final) track decodes from `start` before the first walk that leaves the tree
 * on a real payload bit (src/decoder.c:69-71); DEC_NO_BAD if it reaches `limit` or the end of
 * the payload first.  Only run for lanes whose track has such an event: corrupt streams. */
template <int THREADS>
__device__ __noinline__ uint32_t dec_first_bad(const DecShared<THREADS> &sh, uint32_t start, uint32_t limit, uint32_t pay_rel)
{
    uint32_t pos = start, c = 0;
    while (pos < limit) {
        const uint32_t g = pos >> 5, off = pos & 31u;
        const uint32_t w0 = sh.pay[pay_slot<DecShared<THREADS>::COLS>(g)];
        const uint32_t w1 = sh.pay[pay_slot<DecShared<THREADS>::COLS>(g + 1)];
        const uint32_t win = off ? ((w0 << off) | (w1 >> (32u - off))) : w0;
        const uint32_t e = sh.lut[win >> (32 - DEC_LUT_BITS)];
        if (e < DEC_E_BAD) {
            if (pos + dec_e_adv(e) > pay_rel) return DEC_NO_BAD;
            pos += dec_e_adv(e);
        } else {
            const uint64_t r = dec_rare_packed<THREADS>(sh, e, pos, pay_rel);
            const uint32_t npos = (uint32_t)r;
            const int st = (int)(r >> 40);
            if (st == CW_BAD && npos <= pay_rel) return c;
            if (st != CW_OK || npos > pay_rel) return DEC_NO_BAD;
            pos = npos;
        }
        c++;
    }
    return DEC_NO_BAD;
}

/* Write pass: the lane's first `quota` symbols go to g[0..quota) (STORE) or nowhere (probe).
 * Returns the position after the last one.  The track has been validated by the count pass:
 * every lookup is a codeword. */
template <int THREADS, bool STORE>
__device__ __forceinline__ uint32_t dec_write(const DecShared<THREADS> &sh, uint32_t start, uint32_t pay_rel,
                                              uint32_t quota, uint8_t *g)
{
    BufReader<DecShared<THREADS>::COLS> rd;
    rd.pay = sh.pay;
    rd.load(start);
    auto next = [&]() -> uint32_t {              /* table entry of the next codeword: low byte = symbol */
        uint32_t e = sh.lut[rd.index()];
        if (__builtin_expect(__ballot(e >= DEC_E_BAD) != 0ull, 0)) {
            if (e >= DEC_E_BAD) {
                const uint64_t r = dec_rare_packed<THREADS>(sh, e, rd.pos(), pay_rel);
                rd.load((uint32_t)r);
                e = (uint32_t)(r >> 32) & 0xffu;
            }
        }
        rd.consume(e >> 8);
        return e;
    };
    if (!STORE) {                       /* probe mode: only the position after the quota is wanted */
        for (uint32_t c = 0; c < quota; c++) {
            (void)next();
            if (rd.avail <= 32) rd.refill();
        }
        return rd.pos();
    }
    /* bytes up to the first 4-byte boundary of the output, whole words (four table entries folded
     * into one register with v_alignbit, one 32-bit store, a refill check per two codewords), the
     * bytes that are left */
    const uint32_t head = dmin<uint32_t>(quota, (4u - (uint32_t)((uintptr_t)g & 3u)) & 3u);
    for (uint32_t c = 0; c < head; c++) {
        g[c] = (uint8_t)next();
        if (rd.avail <= 32) rd.refill();
    }
    uint32_t *gw = reinterpret_cast<uint32_t *>(g + head);
    const uint32_t words = (quota - head) >> 2;
    for (uint32_t k = 0; k < words; k++) {
        uint32_t acc = 0;
        acc = __builtin_amdgcn_alignbit(next(), acc, 8);
        acc = __builtin_amdgcn_alignbit(next(), acc, 8);
        if (rd.avail <= 32) rd.refill();
        acc = __builtin_amdgcn_alignbit(next(), acc, 8);
        acc = __builtin_amdgcn_alignbit(next(), acc, 8);
        if (rd.avail <= 32) rd.refill();
        gw[k] = acc;
    }
    for (uint32_t c = head + 4u * words; c < quota; c++) {
        g[c] = (uint8_t)next();
        if (rd.avail <= 32) rd.refill();
    }
    return rd.pos();
}

/* Set bits among the payload bits [0, have) that the aligned 16 bytes at address p hold (v = those
 * 16 bytes; the payload starts at address a0 and `end` is one past the last byte with a needed bit). */
__device__ __forceinline__ uint32_t single_leaf_stray_bits(uint4 v, uintptr_t p, uintptr_t a0, uintptr_t end, uint64_t have)
{
    if (p >= a0 && p + 16 < end) return v.x | v.y | v.z | v.w;      /* interior: every bit counts */
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uintptr_t addr = p + (uintptr_t)k;
        uint32_t byte = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
        if (addr < a0 || addr >= end) byte = 0;
        else if (addr == end - 1 && (have & 7u)) byte &= 0xff00u >> (uint32_t)(have & 7u);   /* first bits = high bits */
        any |= byte;
    }
    return any;
}

/* dst[0, n) = symv by the whole workgroup: bytes up to the first 16-byte boundary, 16-byte
 * streaming stores, bytes behind the last boundary */
template <int THREADS>
__device__ __forceinline__ void fill_bytes(uint8_t *dst, uint64_t n, uint32_t symv)
{
    const int tid = (int)threadIdx.x;
    const uint32_t rep = symv * 0x01010101u;
    const uint64_t head = dmin<uint64_t>(n, (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u);
    if ((uint64_t)tid < head) dst[tid] = (uint8_t)symv;
    uint4 *q = reinterpret_cast<uint4 *>(dst + head);
    const uint64_t nvec = (n - head) >> 4;
    const uint4 v4 = make_uint4(rep, rep, rep, rep);
    for (uint64_t i = (uint64_t)tid; i < nvec; i += THREADS) store_stream16(q + i, v4);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < n) dst[tail0 + tid] = (uint8_t)symv;
}

/* Trees whose root has one leaf child on the left: every symbol is a single 0 bit and a 1 bit
 * leaves the tree (src/decoder.c:69-71).  The output is a fill and the payload only has to be
 * free of set bits: its loads are issued, then the fill (as far as the input could reach at
 * all), then the loaded words are looked at - memory operations complete in order, so neither
 * waits for the other.  Only a payload with a set bit is read again for the bit's position; what
 * the fill wrote behind it is unspecified, as everywhere behind the end of a failed decode.
 * Aligned 16-byte loads: the chunks that hold a payload byte are readable (never across a page). */
template <int THREADS, bool STORE, class SH>
__device__ int decode_single_leaf(SH &sh, uint32_t symv, const uint8_t *pay, uint64_t block_len,
                                  uint64_t pay_bytes, uint8_t *gout, uint64_t *end_bits, uint64_t *produced_out)
{
    const int tid = (int)threadIdx.x;
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uint64_t have = dmin<uint64_t>(block_len, pay_bits);       /* bits we may look at */
    const uintptr_t a0 = (uintptr_t)pay, end = a0 + (uintptr_t)((have + 7) >> 3);
    uintptr_t p = (a0 & ~(uintptr_t)15) + 16u * (uintptr_t)tid;
    uint4 v0 = make_uint4(0, 0, 0, 0);
    if (p < end) v0 = load_stream16(reinterpret_cast<const uint4 *>(p));
    if (STORE) fill_bytes<THREADS>(gout, have, symv);
    uint32_t any = (p < end) ? single_leaf_stray_bits(v0, p, a0, end, have) : 0u;
    for (p += 16u * THREADS; p < end; p += 16u * THREADS)
        any |= single_leaf_stray_bits(load_stream16(reinterpret_cast<const uint4 *>(p)), p, a0, end, have);
    uint64_t good = have;
    if (__syncthreads_or(any != 0u)) {                               /* a 1 bit: which one is the first? */
        if (tid == 0) sh.firstone = DEC_NO_BAD;
        __syncthreads();
        const uint64_t nwords = (have + 31) >> 5;
        for (uint64_t w = (uint64_t)tid; w < nwords; w += THREADS) {
            uint32_t v = load_be32(pay, w * 4, pay_bytes);
            const uint64_t left_bits = have - (w << 5);
            if (left_bits < 32) v &= ~(0xffffffffu >> (uint32_t)left_bits);
            if (v) { atomicMin(&sh.firstone, (uint32_t)((w << 5) + (uint32_t)__clz(v))); break; }
        }
        __syncthreads();
        good = sh.firstone;
        *produced_out = good;
        return HUFE_CORRUPTED;                                       /* decoder.c:69-71 */
    }
    *produced_out = good;
    if (have < block_len) return HUFE_RW;                            /* decoder.c:53-56 */
    *end_bits = block_len;
    return HUFE_OK;
}

/* Steps 0-3 of a block's decode: child links from the serialized tree, then the lookup table.
 * Returns HUFE_OK with the tables in sh (*single_leaf = -1), HUFE_OK with *single_leaf = the byte of
 * a tree whose root has one leaf child on the left (no table is built: the payload is all zero
 * bits), or HUFE_CORRUPTED for a NULL root.  SPEC = the table also folds "failing run + the codeword
 * behind it" into one entry, which only the speculative (self-synchronising) lanes profit from.
 * (Inlined by force: out of line it is compiled without its callers' register bound - 70 registers - and a kernel's count is
 *  the largest of its own and its callees': decode_fast_kernel and probe_kernel fell to three workgroups a CU when the
 *  compiler stopped inlining it; tests/test_isa_check.py watches the count.) */
template <int THREADS, bool SPEC = true>
__device__ __forceinline__ int dec_build_tables(DecShared<THREADS> &sh, const uint8_t *tree, int tree_len, int *single_leaf)
{
    constexpr int ENT = DecShared<THREADS>::ENT;
    const int tid = (int)threadIdx.x;
    *single_leaf = -1;

    /* ---- 0. the tree every one-symbol block carries, [root, leaf, -1, -1, -1],
     *         is recognised straight from its five entries; other shapes of single-leaf trees are
     *         caught after the general tree build below ---- */
    unsigned long long pt = DPROF_T();
    if (tree_len == 5) {
        const int leaf = single_leaf_symbol(tree);
        if (leaf >= 0) { *single_leaf = leaf; return HUFE_OK; }
    }

    /* ---- 1. tree ---- */
    __syncthreads();           /* previous user of sh is done */
    uint16_t *s_open = reinterpret_cast<uint16_t *>(&sh.pay[0]);   /* S(i); payload not staged yet */
    static_assert(sizeof(sh.pay) >= ENT * sizeof(uint16_t), "S(i) scratch must fit");
    {
        /* entries 2t and 2t+1 from two aligned 32-bit loads per thread (a dword that holds a tree byte
         * is readable), shifted by the tree's byte misalignment */
        const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
        const uint32_t mis = (uint32_t)(a & 3u);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(a - mis);
        const uint32_t nbytes = mis + 2u * (uint32_t)tree_len;           /* bytes from q[0] to the tree's end */
        for (int t = tid; 2 * t < ENT; t += THREADS) {
            const uint32_t lo = (4u * (uint32_t)t < nbytes) ? q[t] : 0u;
            const uint32_t hi = (4u * (uint32_t)t + 4u < nbytes) ? q[t + 1] : 0u;
            const uint32_t two = mis ? __builtin_amdgcn_alignbit(hi, lo, 8u * mis) : lo;
            const int i = 2 * t;
            sh.ent[i] = (i < tree_len) ? (int16_t)(two & 0xffffu) : (int16_t)-1;
            if (i + 1 < ENT) sh.ent[i + 1] = (i + 1 < tree_len) ? (int16_t)(two >> 16) : (int16_t)-1;
            sh.lr[i] = DEC_LEAF_LR;
            if (i + 1 < ENT) sh.lr[i + 1] = DEC_LEAF_LR;
        }
    }
    if (tid == 0) { sh.efflen = tree_len; sh.badsym = DEC_NO_BAD; sh.firstone = DEC_NO_BAD; sh.qend = 0; }
    __syncthreads();
    {
        constexpr int PER = (ENT + THREADS - 1) / THREADS;
        int local[PER];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            local[k] = (i < tree_len) ? ((sh.ent[i] != -1) ? 1 : -1) : 0;
            sum += local[k];
        }
        uint32_t tot;
        /* signed values scanned in two's complement */
        const uint32_t ex = block_excl_scan_u32<THREADS>((uint32_t)sum, sh.part, tot);
        int run = 1 + (int)ex;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            if (i < ENT) {
                s_open[i] = (uint16_t)(run < 0 ? 0 : run);
                if (i < tree_len && run <= 0) atomicMin(&sh.efflen, i);
                run += local[k];
            }
        }
    }
    __syncthreads();
    const int eff = (int)uni32((uint32_t)sh.efflen);
    /* Child links: the left child of node j is entry j+1; its right child is the entry behind the
     * left subtree, and that is the first r > j with S(r) <= S(j) (inside the left subtree more
     * slots are open than before j; it is complete when the count is back at S(j)).  Entries at or
     * past `eff` do not exist (NULL).  "First later entry with a value <= mine" for all entries at
     * once: a min-tree over S in heap order (node h covers leaves [h << k, (h + 1) << k) - (N >> k)),
     * leaves past `eff` hold 0 so that every search ends; a search climbs while the node to the
     * right has no such entry and then descends to the first one that has (<= 2 x 11 reads).
     * (The first version settled subtree sizes bottom-up, one tree level per round and barrier.) */
    {
        constexpr int N = 2048, TOP = 11;                 /* leaves (>= ENT + 1), levels above them */
        static_assert(N > ENT && (N >> TOP) == 1 && THREADS * 4 == N, "one thread per four leaves");
        uint16_t *s_min = reinterpret_cast<uint16_t *>(&sh.pay[(ENT + 1) / 2]);      /* behind S(i) */
        static_assert(sizeof(sh.pay) >= ((ENT + 1) / 2) * sizeof(uint32_t) + 2 * N * sizeof(uint16_t), "min-tree must fit");
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = 4 * tid + q;
            v[q] = (i < eff) ? (uint32_t)s_open[i] : 0u;
            s_min[N + i] = (uint16_t)v[q];
        }
        const uint32_t a0 = dmin<uint32_t>(v[0], v[1]), a1 = dmin<uint32_t>(v[2], v[3]);
        s_min[(N >> 1) + 2 * tid] = (uint16_t)a0;
        s_min[(N >> 1) + 2 * tid + 1] = (uint16_t)a1;
        uint32_t m = dmin<uint32_t>(a0, a1);
        s_min[(N >> 2) + tid] = (uint16_t)m;                          /* level 2: one node per thread */
#pragma unroll
        for (int k = 3; k <= 8; k++) {                                /* levels 3..8 inside the wave */
            m = dmin<uint32_t>(m, wave_xor_any(m, 1 << (k - 3)));
            if ((tid & ((1 << (k - 2)) - 1)) == 0) s_min[(N >> k) + (tid >> (k - 2))] = (uint16_t)m;
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll 1
            for (int h = (N >> 8) - 1; h >= 1; h--) s_min[h] = (uint16_t)dmin<uint32_t>(s_min[2 * h], s_min[2 * h + 1]);
        }
        __syncthreads();
        for (int j = tid; j < eff; j += THREADS) {
            if (sh.ent[j] == -1) continue;
            const int l = j + 1;
            uint32_t links = DEC_LEAF_LR;
            if (l < eff && sh.ent[l] != -1) links = 0xffff0000u | (uint32_t)l;
            const uint32_t s = s_open[j];
            uint32_t k = 0, p = (uint32_t)l;                          /* node (k, p): leaves [p << k, (p + 1) << k) */
            while (s_min[(N >> k) + p] > s) {                         /* nothing in this node: the next one, as high up as it starts */
                p++;
                while ((p & 1u) == 0u && k < (uint32_t)TOP) { p >>= 1; k++; }
            }
            while (k > 0) {                                           /* the first leaf inside that qualifies */
                k--;
                p <<= 1;
                if (s_min[(N >> k) + p] > s) p++;
            }
            if ((int)p < eff && sh.ent[p] != -1) links = (links & 0xffffu) | (p << 16);
            sh.lr[j] = links;
        }
    }
    __syncthreads();
    /* tree_len == 0 or a tree that starts with -1 is a NULL root: the reference crashes,
     * the decision is BTREE_CORRUPTED */
    if (!(eff > 0 && sh.ent[0] != -1)) return HUFE_CORRUPTED;

    DPROF_ADD(0, pt); pt = DPROF_T();

    /* ---- 2. single-leaf tree: every symbol is one 0 bit ---- */
    {
        const uint32_t lr0 = sh.lr[0];
        const uint32_t l0 = lr0 & 0xffffu;
        if (l0 != DEC_NULL && (lr0 >> 16) == DEC_NULL && sh.lr[l0] == DEC_LEAF_LR) {
            *single_leaf = (int)(uint8_t)sh.ent[l0];
            return HUFE_OK;
        }
    }

    /* ---- 3. lookup table ----
     * Two hops of DEC_LUT_BITS/2 bits: first the node (or verdict) reached after the high half
     * of the index, kept in the low 64 table slots for a moment, then every entry continues
     * from there - half the dependent LDS steps of walking all 12 bits per entry. */
    constexpr int HALF = DEC_LUT_BITS / 2;
    /* one step of a walk: (node, its children) and the next bit -> verdict, or the child and ITS
     * children (one dependent LDS read per level; a leaf's byte is a second read at the very end) */
    auto walk_step = [&](uint32_t &node, uint32_t &lrn, uint32_t bit, int b, uint32_t &e) -> bool {
        const uint32_t nx = dec_child(lrn, bit);
        if (nx == DEC_NULL) { e = (2u << 14) | ((uint32_t)(b + 1) << 8); return true; }
        const uint32_t lrx = sh.lr[nx];
        if (lrx == DEC_LEAF_LR) { e = ((uint32_t)(b + 1) << 8) | ((uint32_t)(uint8_t)sh.ent[nx]); return true; }
        node = nx;
        lrn = lrx;
        return false;
    };
    uint32_t hop1 = 0;
    if (tid < (1 << HALF)) {
        uint32_t node = 0, lrn = sh.lr[0], e = 0;
        bool done = false;
#pragma unroll
        for (int b = 0; b < HALF; b++) {
            const uint32_t bit = ((uint32_t)tid >> (HALF - 1 - b)) & 1u;
            if (!done) done = walk_step(node, lrn, bit, b, e);
        }
        hop1 = done ? e : ((1u << 14) | node);           /* type 1 here: "continue from node" */
    }
    __syncthreads();                                      /* nobody reads the LUT region yet: reuse the end of it */
    uint16_t *s_hop = sh.lut + (1 << DEC_LUT_BITS) - (1 << HALF);
    if (tid < (1 << HALF)) s_hop[tid] = (uint16_t)hop1;
    __syncthreads();
    {
        constexpr int PERL = (1 << DEC_LUT_BITS) / THREADS;
        uint16_t mine[PERL > 0 ? PERL : 1];
        /* the thread's PERL walks advance level by level TOGETHER: their LDS reads are independent, so
         * a level costs one LDS latency, not PERL of them (the walks one after the other, three
         * dependent reads per level, were 14k of the table build's 30k cycles) */
        uint32_t we[PERL > 0 ? PERL : 1], wnode[PERL > 0 ? PERL : 1], wlr[PERL > 0 ? PERL : 1];
        bool wdone[PERL > 0 ? PERL : 1];
#pragma unroll
        for (int k = 0; k < PERL; k++) {
            const int idx = tid + k * THREADS;
            we[k] = s_hop[idx >> HALF];
            wdone[k] = (we[k] >> 14) != 1u;
            wnode[k] = we[k] & 0x7ffu;
        }
#pragma unroll
        for (int k = 0; k < PERL; k++) wlr[k] = wdone[k] ? DEC_LEAF_LR : sh.lr[wnode[k]];
#pragma unroll
        for (int b = HALF; b < DEC_LUT_BITS; b++)
        {
#pragma unroll
            for (int k = 0; k < PERL; k++) {
                const int idx = tid + k * THREADS;
                const uint32_t bit = ((uint32_t)idx >> (DEC_LUT_BITS - 1 - b)) & 1u;
                if (!wdone[k]) wdone[k] = walk_step(wnode[k], wlr[k], bit, b, we[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < PERL; k++) {
            const int idx = tid + k * THREADS;
            uint32_t e = wdone[k] ? we[k] : ((1u << 14) | wnode[k]);
            if ((e >> 14) == 1u) e = DEC_E_LONG | (e & 0x7ffu);
            else if ((e >> 14) == 2u) {
                /* bad: resume one bit on; when the very first bit fails, every bit of the run of
                 * equal bits after it fails the same way */
                const uint32_t bits = (e >> 8) & 0xfu;
                uint32_t skip = 1;
                if (bits == 1u) {
                    const uint32_t top = (uint32_t)idx << (32 - DEC_LUT_BITS);
                    skip = dmin<uint32_t>((uint32_t)__clz((int)((top >> 31) ? ~top : top)), (uint32_t)DEC_LUT_BITS);
                }
                e = DEC_E_BAD | DEC_E_NOCW | (skip << 8) | bits;
            }
            mine[k] = (uint16_t)e;
        }
        __syncthreads();                                  /* all reads of s_hop are done */
#pragma unroll
        for (int k = 0; k < PERL; k++) sh.lut[tid + k * THREADS] = mine[k];
        /* A speculative lane that meets a run of failing bits decodes the codeword behind the run
         * in its next iteration; when run + codeword fit the window, one entry does both
         * (DEC_E_NOCW clear), which helps data with short codes (uniform bytes 2.61 -> 2.48 ms). */
        __syncthreads();
        if (SPEC) {
#pragma unroll
        for (int k = 0; k < PERL; k++) {
            const uint32_t e = mine[k];
            if (e >= DEC_E_BAD && e < DEC_E_LONG && (e & 0x7fu) == 1u) {
                const uint32_t run = dec_e_adv(e);
                const uint32_t idx = (uint32_t)(tid + k * THREADS);
                const uint32_t e2 = sh.lut[(idx << run) & ((1u << DEC_LUT_BITS) - 1u)];
                if (run < (uint32_t)DEC_LUT_BITS && e2 < DEC_E_BAD && run + (e2 >> 8) <= (uint32_t)DEC_LUT_BITS)
                    sh.lut[idx] = (uint16_t)(DEC_E_BAD | ((run + (e2 >> 8)) << 8) | 1u);
            }
        }
        }
    }
    __syncthreads();
    DPROF_ADD(1, pt);
    return HUFE_OK;
}

/* Decode one block whose header has been parsed.  `tree` points at the tree_len int16 entries,
 * the payload follows them and at most pay_bytes of it may be read.  Writes block_len bytes
 * to gout.  Returns HUFE_*; *end_bits = payload bits consumed up to and including the last
 * symbol (valid on success); *produced_out = symbols delivered (also on failure). */
template <int THREADS, bool STORE = true>
__device__ int decode_block(DecShared<THREADS> &sh, const uint8_t *tree, int tree_len,
                            uint64_t block_len, uint64_t pay_bytes, uint8_t *gout, uint64_t *end_bits,
                            uint64_t *produced_out)
{
    constexpr int COLS = DecShared<THREADS>::COLS;
    const int tid = (int)threadIdx.x;
    *produced_out = 0;
    const uint8_t *pay = tree + 2 * tree_len;
    const uint64_t pay_bits = pay_bytes * 8ull;
    {
        int leaf = -1;
        const int rc = dec_build_tables<THREADS, true>(sh, tree, tree_len, &leaf);
        if (rc != HUFE_OK) return rc;
        if (leaf >= 0)
            return decode_single_leaf<THREADS, STORE>(sh, (uint32_t)leaf, pay, block_len, pay_bytes, gout, end_bits,
                                                      produced_out);
    }
    unsigned long long pt = DPROF_T();
    /* ---- 4. payload ---- */
    uint64_t true_start = 0;      /* bit where the next undecoded codeword starts */
    uint64_t produced = 0;        /* symbols written so far */
    int err = HUFE_OK;
    const uint32_t sub_lo = (uint32_t)tid * DEC_SUB_BITS;

    while (produced < block_len) {
        if (true_start >= pay_bits) { err = HUFE_RW; break; }          /* input exhausted */
        /* segment origin: the 32-bit word that holds true_start */
        pt = DPROF_T();
        const uint64_t seg0 = true_start & ~31ull;
        const uint64_t byte0 = seg0 >> 3;
        /* (the lane index is laundered so that the staging addresses are recomputed per segment:
         * hoisted out of this loop they do not fit in 64 VGPRs and are spilled to scratch, which
         * showed up as +20 % HBM traffic of the kernel) */
        int tl = tid;
        asm volatile("" : "+v"(tl));
        if (byte0 + 4ull * (DEC_SUB_WORDS * COLS) + 8ull <= pay_bytes) {
            /* the whole staged window lies inside the payload: two aligned loads and ONE v_perm per
             * word (byte order and the payload's byte misalignment in one selector, same for all) */
            const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)(pay + byte0));   /* block-uniform: SGPR base */
            const uint32_t m = (uint32_t)(a & 3u);
            const uint32_t sel = (m << 24) | ((m + 1u) << 16) | ((m + 2u) << 8) | (m + 3u);
            const uint32_t *q = reinterpret_cast<const uint32_t *>(a - m);
            for (int i = tl; i < DEC_SUB_WORDS * COLS; i += THREADS)
                sh.pay[pay_slot<COLS>((uint32_t)i)] = __builtin_amdgcn_perm(q[i + 1], q[i], sel);
        } else {
            for (int i = tl; i < DEC_SUB_WORDS * COLS; i += THREADS)
                sh.pay[pay_slot<COLS>((uint32_t)i)] = load_be32(pay, byte0 + 4ull * i, pay_bytes);
        }
        __syncthreads();
        const uint32_t pay_rel = (uint32_t)dmin<uint64_t>(pay_bits - seg0, 0xfffffff0ull);
        const uint32_t first_start = (uint32_t)(true_start - seg0);
        DPROF_ADD(2, pt); pt = DPROF_T();

        LaneTrack tr;
        dec_scan<THREADS, false>(sh, tr, tid == 0 ? first_start : sub_lo, sub_lo, pay_rel);
        if ((tid & 63) == 63) sh.wend[tid >> 6] = tr.end;
        __syncthreads();
        DPROF_ADD(3, pt); pt = DPROF_T();
        for (;;) {
            /* left neighbour's end: a shuffle inside the wave, LDS across the wave seams */
            uint32_t ns = wave_up1_u32(tr.end);
            if ((tid & 63) == 0) ns = (tid == 0) ? first_start : sh.wend[(tid >> 6) - 1];
            const int changed = (ns != tr.start);
            __syncthreads();                               /* everyone has read sh.wend */
            if (changed) {
                dec_scan<THREADS, true>(sh, tr, ns, sub_lo, pay_rel);
            }
            if ((tid & 63) == 63) sh.wend[tid >> 6] = tr.end;
            if (!__syncthreads_or(changed)) break;
        }

        DPROF_ADD(4, pt); pt = DPROF_T();
        /* output positions */
        uint32_t seg_total;
        const uint32_t ex = block_excl_scan_u32<THREADS>(tr.cnt, sh.part, seg_total);
        const uint64_t remaining = block_len - produced;
        /* the first walk that left the tree, in stream order, is a real error if it happens
         * before the block is complete (src/decoder.c:69-71); later ones are padding/garbage.
         * Symbols decoded before it are still delivered, like the reference's writer does. */
        if (tr.lastbad >= 0 && (uint64_t)ex < remaining) {
            const uint32_t bad_at = dec_first_bad<THREADS>(sh, tr.start, sub_lo + DEC_SUB_BITS, pay_rel);
            if (bad_at != DEC_NO_BAD && (uint64_t)ex + bad_at < remaining) atomicMin(&sh.badsym, ex + bad_at);
        }
        __syncthreads();
        const uint32_t badsym = uni32(sh.badsym);
        seg_total = uni32(seg_total);
        const uint32_t good = (badsym != DEC_NO_BAD) ? badsym : seg_total;
        const uint32_t take = (uint32_t)dmin<uint64_t>(good, remaining);
        /* (plain ifs: the select/min form of this was observed to misbehave when compiled inside
         * the previous version of this kernel by ROCm 7.2 hipcc) */
        uint32_t quota = 0;
        if (ex < take) {
            quota = take - ex;
            if (quota > tr.cnt) quota = tr.cnt;
        }
        DPROF_ADD(5, pt); pt = DPROF_T();
        if (STORE) {
            if (quota) {
                const uint32_t qe = dec_write<THREADS, true>(sh, tr.start, pay_rel, quota, gout + produced + ex);
                if (ex + quota == take && remaining <= good) sh.qend = qe;   /* block's last symbol */
            }
        } else if (quota && ex + quota == take && remaining <= good) {
            /* probe: only the lane that holds the block's last symbol walks, to find where it ends */
            sh.qend = dec_write<THREADS, false>(sh, tr.start, pay_rel, quota, nullptr);
        }
        const uint32_t last_end = uni32(sh.wend[THREADS / 64 - 1]);
        __syncthreads();
        DPROF_ADD(6, pt);
        produced += take;
        if (badsym != DEC_NO_BAD) { err = HUFE_CORRUPTED; break; }
        if (produced < block_len) {
            if (last_end == DEC_EXH) { err = HUFE_RW; break; }
            true_start = seg0 + last_end;
        } else {
            true_start = seg0 + uni32(sh.qend);
        }
    }
    if (err == HUFE_OK) *end_bits = true_start;
    *produced_out = produced;
    return err;
}

/* Indexed decode: one workgroup per block, block extents from the in-process index. */
#ifndef DEC_WAVES_PER_SIMD
#define DEC_WAVES_PER_SIMD 8      /* 4 workgroups of 512 per CU: caps the kernel at 64 VGPRs (no scratch), +15 % over 3 workgroups */
#endif
template <int THREADS>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void decode_kernel(const uint8_t *__restrict__ stream,
                                                         uint64_t stream_len,
                                                         const uint64_t *__restrict__ offsets,
                                                         const HufDecodeMeta *__restrict__ dmeta,
                                                         uint64_t *__restrict__ out_offsets, TwoLevel lens,
                                                         uint8_t *__restrict__ out, uint64_t out_cap,
                                                         int32_t *__restrict__ status,
                                                         unsigned long long *__restrict__ result)
{
    __shared__ DecShared<THREADS> sh;
    const int tid = (int)threadIdx.x;
    const uint64_t blk = blockIdx.x;
    const HufDecodeMeta m = dmeta[blk];
    int err = m.status;
    const uint64_t obase = lens.gprefix[blk / SCAN_GROUP] + lens.local[blk];
    if (tid == 0) out_offsets[blk] = obase;          /* hufgpu_decode_result: bytes before a failing block */
    if (err == HUFE_OK && m.block_len > 0) {
        const uint64_t o0 = offsets[blk];
        const uint64_t o1 = dmin<uint64_t>(offsets[blk + 1], stream_len);
        if (obase + m.block_len > out_cap) {
            err = HUFE_MEMORY;
        } else {
            const uint64_t pay_bytes = o1 - (o0 + HUF_HEADER_FIXED + 2ull * (uint64_t)m.tree_len);
            uint64_t end_bits = 0, produced = 0;
            const uint8_t *tree = stream + o0 + HUF_HEADER_FIXED;
            if (m.leaf >= 0)                   /* decode_prepare has read the tree: straight to the fill */
                err = decode_single_leaf<THREADS, true>(sh, (uint32_t)m.leaf, tree + 10, m.block_len, pay_bytes,
                                                        out + obase, &end_bits, &produced);
            else
                err = decode_block<THREADS>(sh, tree, m.tree_len, m.block_len, pay_bytes, out + obase, &end_bits,
                                            &produced);
        }
    }
    if (tid == 0 && err != m.status) {       /* header errors were recorded by decode_prepare */
        status[blk] = err;
        if (err != HUFE_OK) atomicMin(&result[2], (unsigned long long)blk);
    }
}

/* Raw-stream decode (no index): the block loop of src/decoder.c:218-276 run by ONE workgroup.
 * Blocks are taken strictly in order because a block's end is only known once block_len
 * symbols have been decoded; inside a block all lanes work in parallel.
 * result[0] = error, [1] = bytes written, [2] = reader bytes consumed, [3] = blocks done,
 * [4] / [5] = stream bytes / output bytes of the blocks that decoded completely. */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void decode_chain_kernel(const uint8_t *__restrict__ stream,
                                                               uint64_t avail, uint64_t length,
                                                               int max_tree_len, uint8_t *__restrict__ out,
                                                               uint64_t out_cap, uint64_t *__restrict__ result,
                                                               uint64_t *__restrict__ block_offsets,
                                                               uint64_t max_index)
{
    __shared__ DecShared<THREADS> sh;
    uint64_t rd = 0, wr = 0, nblk = 0;
    uint64_t good_rd = 0, good_wr = 0;                                /* behind the last block that decoded completely */
    int err = HUFE_OK;
    while (length > rd) {                                             /* decoder.c:218 */
        good_rd = rd;
        good_wr = wr;
        if (block_offsets && nblk < max_index) {
            if (threadIdx.x == 0) block_offsets[nblk] = rd;
        }
        if (avail - rd < 8) { err = HUFE_RW; break; }                 /* decoder.c:220-224 */
        const uint64_t block_len = load_u64_unaligned(stream + rd);
        rd += 8;
        if (avail - rd < 2) { err = HUFE_RW; break; }                 /* decoder.c:231-234 */
        const int16_t tl = (int16_t)((uint16_t)stream[rd] | ((uint16_t)stream[rd + 1] << 8));
        rd += 2;
        if (tl < 0 || tl > max_tree_len) { err = HUFE_OVERFLOW; break; }   /* decoder.c:237-239 */
        if (avail - rd < 2ull * (uint64_t)tl) { err = HUFE_RW; break; }    /* decoder.c:248-252 */
        const uint8_t *tree = stream + rd;
        rd += 2ull * (uint64_t)tl;
        if (block_len == 0) { nblk++; continue; }
        /* more symbols than payload bits left (a damaged header): decode what is there, then fail
         * where the reference's reader runs out of input (decoder.c:53-56) */
        uint64_t want = block_len;
        if (want > (avail - rd) * 8ull) want = (avail - rd) * 8ull + 1;
        /* ... and no more than the output has room for: an error inside that part is the
         * stream's first error; only a block that decodes cleanly up to there needs more room */
        const bool capped = want > out_cap - wr;
        if (capped) want = out_cap - wr;
        if (want > HUF_MAX_BLOCK_LEN) { err = HUFE_ARGUMENT; break; }
        uint64_t end_bits = 0, produced = 0;
        if (want) err = decode_block<THREADS>(sh, tree, tl, want, avail - rd, out + wr, &end_bits, &produced);
        if (err != HUFE_OK) { wr += produced; break; }   /* symbols before the failure stay delivered */
        if (capped) { wr += want; err = HUFE_MEMORY; break; }
        rd += (end_bits + 7) >> 3;
        wr += block_len;
        nblk++;
    }
    if (threadIdx.x == 0) {
        result[0] = (uint64_t)err;
        result[1] = wr;
        result[2] = rd;
        result[3] = nblk;
        result[4] = (err == HUFE_OK) ? rd : good_rd;
        result[5] = (err == HUFE_OK) ? wr : good_wr;
        if (block_offsets && nblk < max_index) block_offsets[nblk] = rd;
    }
}

}  // namespace hufgpu
