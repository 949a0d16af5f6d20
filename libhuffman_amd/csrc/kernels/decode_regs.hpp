/* decode_regs.hpp - decode_payload_regs: the payload of a block decoded with the block index alone (what huf_decode()
   and streams written by the reference get; src/decoder.c:34-96), round 6 form: every codeword is looked up ONCE per
   synchronisation round and the decoded bytes wait in registers for their place in the output.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "decode.hpp"
#include "decode_sub.hpp"
#include "decode_fast.hpp"

namespace hufgpu {

/* ======================================================================================
 * decode_fast.hpp's lean decoder walks every codeword 3.2 times: a scan that counts, 1.2 scans after the starts have
 * moved, and a write pass that decodes what the scans only measured - 42.5 vector instructions a symbol where
 * decode_sub_kernel, which is TOLD where a lane's symbols start, takes 10.5 (profiles/r05/dfast_budget.txt).  The
 * scans could not keep their symbols because a lane does not know where its bytes go before every lane in front of it
 * has counted.  Here they are kept all the same - in registers:
 *
 *   - a lane's share of a segment is at most 384 payload bits and (the shares are cut to the block's bits per
 *     symbol) about 48 symbols, at most 64: sixteen registers of four bytes (DREG_ITERS);
 *   - ONE pass form: from the lane's start, four symbols an iteration with decode_sub's loop (entries byte << 8 | length
 *     over the 12 bits at a position, a position register that counts down with a gap, both lengths of a window
 *     subtracted by one v_dot4c) - 27 vector instructions per four symbols, the iteration's four bytes into register k;
 *     a lane whose position has passed its share's end is switched off; the four lengths of its last iteration give
 *     the end (the first codeword start at or behind the share's end) and the count;
 *   - the rounds are decode_fast's: lane 0 starts at the segment's true first codeword, every other lane at its own
 *     first bit, then at its left neighbour's end, until no start changes.  A wave in which a start changed runs the
 *     pass again (every lane of it: a lane whose start stood decodes what it decoded before);
 *   - then the counts are summed and every lane stores its registers: whole 16-byte quads, the rest in dwords - the
 *     bytes behind a lane's last symbol in its last dword are its right neighbour's first (the iteration decoded
 *     them on the same track), as in decode_fast's write pass.
 *
 * What the in-order decoder would have checked is checked: every codeword's first bit is 0 (the trees this path
 * takes have a root with a left child only and two children everywhere else - dfast_tables_from_tree - so the 2 048
 * entries of a first bit 0 are all leaves and a 1 leaves the tree: such an entry advances by the run of ones, which is
 * what puts a speculative track into step), no codeword of the block needs bits past the payload,
 * the symbols add up to block_len.  A block that fails any of it goes to the exact decoder (decode_fix_kernel /
 * probe_exact_kernel), which has the reference's error code and byte count.
 * Codes beyond the table's 12 bits (up to 32): DregLeaves below.  Blocks of fewer than 8 192 symbols, with a code beyond 32 bits
 * or a tree of another shape are decode_fast.hpp's, and so are - for the raw-stream probe - blocks of fewer than four bits a symbol.
 * A share that holds more than 64 codewords (codes far shorter than the block's average) has the segment done again
 * with shares of half the bits; 128 bits cannot hold more than 64 codewords (no code of these trees has fewer than 2).
 * ==================================================================================== */
#ifndef DREG_ROWS
#define DREG_ROWS 15u                          /* words of a lane's column: up to 31 bits in front of the share, the share, the 48 bits the last iteration's
                                                  other three codewords may take behind it (32 x rows >= 31 + share + 48 - 1) */
#endif
#define DREG_SUB_BITS (32u * DREG_ROWS - 96u)  /* 15 rows: 384 (round 6b; 12 rows and 288 bits until then: a 64 KiB block of zipf255 was four segments
                                                  of 240 bits, now three of 320 - what a segment costs beside its passes a quarter less often, and the
                                                  wave's slowest lane closer to its average one) */
#ifndef DREG_ITERS
#define DREG_ITERS 16                          /* iterations of four symbols a pass can take: 12 or 16 */
#endif
#ifndef DREG_TARGET_SYMS
#define DREG_TARGET_SYMS (DREG_ITERS == 16 ? (DREG_ROWS >= 15u ? 48u : 40u) : 30u)   /* symbols a share is cut for (of 4 x DREG_ITERS it may hold) */
#endif
#ifndef DREG_MIN_SHARE_BITS
#define DREG_MIN_SHARE_BITS 64u                 /* a block whose shares (DREG_TARGET_SYMS symbols at its bits a symbol) would be shorter is not this path's ... */
#define DREG_MIN_SHARE_BITS_PROBE 192u          /* ... and for the raw-stream probe, whose failures cost more, one of fewer than four bits a symbol: decode_payload_regs */
#endif
#define DREG_MAX_BLOCK (1u << 26)              /* symbols of a block this path takes: 32 x 2^26 payload bits are positions of 32 bits */
#define DREG_SAFE_BITS (8u * DREG_ITERS)       /* a share of so many bits cannot hold more codewords than the registers take: none has fewer than 2 bits */

typedef const __attribute__((address_space(3))) uint32_t *dreg_lds_words;
typedef const __attribute__((address_space(3))) uint16_t *dreg_lds_halves;

/* The position register (decode_sub.hpp): R = 32 x (the LDS row of the column's word 0) - bits from that word's first bit, a
 * row = 256 bytes of LDS counted from address 0, the column's word g at row (word 0's row) - g; kept with a gap,
 * P = (R >> 5) << 8 | (R & 31): P & 0xff00 is the LDS address of the lower of the two rows that hold the 32 bits at the
 * position, P as it stands the amount v_alignbit_b32 shifts the pair by, a codeword P = (P - len) & 0xff1f. */
__device__ __forceinline__ uint32_t dreg_gap(uint32_t R) { return ((R << 3) & 0xff00u) | (R & 31u); }
__device__ __forceinline__ uint32_t dreg_ungap(uint32_t P) { return ((P >> 3) & 0x1fe0u) | (P & 31u); }

__device__ __forceinline__ uint32_t dreg_bits_at(uint32_t P, uint32_t lane4)
{
    dreg_lds_words wp = (dreg_lds_words)(uintptr_t)((P & 0xff00u) | lane4);
    return __builtin_amdgcn_alignbit(wp[64], wp[0], P);
}
__device__ __forceinline__ uint32_t dreg_entry(uint32_t lut_addr, uint32_t d)
{
    return *(dreg_lds_halves)(uintptr_t)(lut_addr + ((d >> 19) & 0x1ffeu));
}

/* Codes beyond the table's twelve bits (round 6b).  Their table entry is DREG_E_LONG = 0: a length of 0 - a lane that meets one
 * stands still, every further look-up of its iteration is the same entry (the window is shifted by 0), and the iteration's
 * LAST length byte says so: no other entry has a length of 0.  The pass then decodes what the lane is short of one codeword at a
 * time (dreg_repair), the long ones by a binary search over the leaves' codes which dfast_tables_from_tree left in sh.leaves
 * (the walk's child links are not built) - a byte seen once in 5 000 costs a wave some hundred instructions in every third
 * pass.  Blocks without such codes run a pass that does not ask (dreg_pass<false>). */
typedef const __attribute__((address_space(3))) uint8_t *dreg_lds_bytes;
struct DregLeaves {
    uint32_t code_a;            /* LDS address of [K] the leaves' codes in preorder = ascending, left-aligned in 32 bits; leaf k's length (2 .. 32)
                                   from the distance to the next code: the codes of a full tree lie 2^(32 - length) apart, the last one ends
                                   at 2^31 (every code begins with the root's 0) */
    uint32_t sym_a;             /* LDS address of [K] their bytes */
    uint32_t K;
    bool any;                   /* (uniform) the block has codes beyond the table */
};
/* byte << 8 | length of the codeword whose (first) 32 bits are d */
__device__ __forceinline__ uint32_t dreg_long_entry(uint32_t code_a, uint32_t sym_a, uint32_t K, uint32_t d)
{
    dreg_lds_words code = (dreg_lds_words)(uintptr_t)code_a;
    uint32_t lo = 0, hi = K;                             /* largest k < K with code[k] <= d (code[0] = 0): dsub_leaf_of on LDS addresses */
#pragma unroll 1
    for (int it = 0; it < 8; it++) {                     /* K <= 256 */
        const uint32_t mid = (lo + hi) >> 1;
        if (hi - lo > 1u) {
            if (code[mid] <= d) lo = mid; else hi = mid;
        }
    }
    const uint32_t next = lo + 1u < K ? code[lo + 1u] : 0x80000000u;
    const uint32_t byte = ((dreg_lds_bytes)(uintptr_t)sym_a)[lo];
    return (byte << 8) | ((uint32_t)__clz((int)(next - code[lo])) + 1u);
}
/* An iteration in which a lane (`hit`) met a long code: its first codewords stand - their bytes in S, their lengths in L4 - and P
 * stands at the long one; the rest of its four one by one.  Called under the iteration's EXEC (the wave's active lanes).  Loops,
 * not unrolled: there are sixteen copies of this in a pass.  (Out of line - one copy - it cost more than it saved: the sixteen
 * registers of symbols around a call are sixteen registers saved and restored, zipf255 1.01 -> 1.33 ms with it.) */
__device__ __forceinline__ void dreg_repair(uint32_t lut_addr, uint32_t lane4, const DregLeaves &lv, bool hit, uint32_t &P, uint32_t &S, uint32_t &L4, uint32_t &firsts)
{
    /* (lengths that stand are not 0: bit 7 of a byte of nz says "not zero") */
    const uint32_t nz = (((L4 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | L4) & 0x80808080u;
    const uint32_t nv = hit ? (uint32_t)__builtin_popcount(nz) : 4u;
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const bool go = j >= nv;
        if (__any(go)) {
            const uint32_t d = dreg_bits_at(P, lane4);
            uint32_t e = dreg_entry(lut_addr, d);
            const bool lg = go && (e & 0xffu) == 0u;
            if (__any(lg)) {
                const uint32_t e2 = dreg_long_entry(lv.code_a, lv.sym_a, lv.K, d);
                e = lg ? e2 : e;
            }
            if (go) {
                const uint32_t len = e & 0xffu;                        /* (at most 32: one borrow) */
                firsts |= d;
                S = (S & ~(0xffu << (8u * j))) | ((e >> 8) << (8u * j));
                L4 = (L4 & ~(0xffu << (8u * j))) | (len << (8u * j));
                P = (P - len) & 0xff1fu;
            }
        }
    }
}

/* what a pass leaves in a lane besides the sixteen registers */
struct DregTrack {
    uint32_t Rend;       /* (as R, not gapped) the first codeword start at or behind the share's end (the start, for a lane that holds nothing) */
    uint32_t cnt;        /* codewords that start in front of the share's end */
    bool bad;            /* a codeword of the lane's iterations began with a 1 (the last iteration's may lie behind the share's end: whoever
                            needs to know exactly walks the lane's codewords again, decode_payload_regs) */
    bool over;           /* sixteen iterations were not enough */
};

/* One pass of a lane: from Pstart while the position lies in front of Phi (both as position registers). */
/* (the sixteen registers are sixteen variables, DregSyms' members: an array of them the compiler turns into ONE value of sixteen
 *  registers in a row, moved, spilled and reloaded whole) */
struct DregSyms { uint32_t s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15; };
template <bool LONG>
__device__ __forceinline__ void dreg_pass(uint32_t lut_addr, uint32_t lane4, const DregLeaves &lv, uint32_t Rfloor, uint32_t Pstart, uint32_t Phi, DregSyms &sym, DregTrack &t)
{
    uint32_t P = Pstart, ng = 0, firsts = 0, L4 = 0;
#define DREG_WINDOW(PAIR)                                                                                      \
    {                                                                                                         \
        const uint32_t d1_ = dreg_bits_at(P, lane4);                                                          \
        const uint32_t e1_ = dreg_entry(lut_addr, d1_);                                                       \
        const uint32_t d2_ = d1_ << (e1_ & 31u);                                                              \
        const uint32_t e2_ = dreg_entry(lut_addr, d2_);                                                       \
        firsts |= d1_ | d2_;                                                                                  \
        asm volatile("" : "+v"(firsts));                                                                      \
        PAIR = __builtin_amdgcn_perm(e2_, e1_, 0x04000501u);           /* byte 1, byte 2, length 1, length 2 */ \
        P = (uint32_t)__builtin_amdgcn_sdot4((int)PAIR, (int)0xffff0000, (int)P, false) & 0xff1fu;            \
    }
    /* one iteration: four symbols into register S.  The chain of iterations ends when no lane of the wave has a position in
     * front of its share's end any more; a lane that is done is switched OFF while its wave goes on (a branch, not selects):
     * its position, the four lengths of its last iteration and that iteration's number stand where it left them - round 6b:
     * until then a done lane walked on and three selects an iteration kept what the end needed (31 -> 28 instructions). */
#define DREG_ITER(S, K)                                                                                        \
    {                                                                                                         \
        const bool act = P > Phi;                                                                             \
        if (!__any(act)) goto done;                                                                           \
        if (act) {                                                                                            \
            uint32_t p01, p23;                                                                                \
            DREG_WINDOW(p01)                                                                                  \
            DREG_WINDOW(p23)                                                                                  \
            S = __builtin_amdgcn_perm(p23, p01, 0x05040100u);                                                 \
            L4 = __builtin_amdgcn_perm(p23, p01, 0x07060302u);     /* lengths 1 .. 4 */                       \
            ng = (K);                            /* (positions only grow: the active iterations are the first ng) */ \
            if (LONG) {                                                                                       \
                const bool hit_ = p23 < 0x01000000u; /* the iteration's last look-up was DREG_E_LONG */        \
                if (__builtin_expect(__any(hit_), 0)) dreg_repair(lut_addr, lane4, lv, hit_, P, S, L4, firsts); \
            }                                                                                                 \
        }                                                                                                     \
    }
    DREG_ITER(sym.s0, 1u) DREG_ITER(sym.s1, 2u) DREG_ITER(sym.s2, 3u) DREG_ITER(sym.s3, 4u)
    DREG_ITER(sym.s4, 5u) DREG_ITER(sym.s5, 6u) DREG_ITER(sym.s6, 7u) DREG_ITER(sym.s7, 8u)
    DREG_ITER(sym.s8, 9u) DREG_ITER(sym.s9, 10u) DREG_ITER(sym.s10, 11u) DREG_ITER(sym.s11, 12u)
#if DREG_ITERS == 16
    DREG_ITER(sym.s12, 13u) DREG_ITER(sym.s13, 14u) DREG_ITER(sym.s14, 15u) DREG_ITER(sym.s15, 16u)
#endif
done:
#undef DREG_ITER
#undef DREG_WINDOW
    t.over = P > Phi;
    t.bad = (firsts >> 31) != 0u;
    /* (a long code in the last iteration can carry it past the column's last row - three more codewords of up to 32 bits where
     *  the rows allow for 48 bits: what was looked up there is not the payload's, and the lane's last dword, whose bytes behind its
     *  own symbols go to the right neighbour's place, must not be stored.  Such a lane ran out of rows as another runs out of
     *  registers: the segment again, with shares of half the bits.  Every codeword that ENDS inside the column was decoded from
     *  its own bits alone - no code is the beginning of another.) */
    if (LONG && dreg_ungap(P) < Rfloor) t.over = true;
    /* the last iteration's codewords began at R1 + length 1, R1, R2, R3 and it ended at R4 = where the lane stands: from the
     * four lengths, no look-up (until round 6b the iteration was looked up again: four dependent table reads a pass) */
    const uint32_t RPhi = dreg_ungap(Phi);
    const uint32_t R4 = dreg_ungap(P);
    const uint32_t R3 = R4 + (L4 >> 24), R2 = R3 + ((L4 >> 16) & 0xffu), R1 = R2 + ((L4 >> 8) & 0xffu);
    const bool in1 = R1 > RPhi, in2 = R2 > RPhi, in3 = R3 > RPhi;
    t.cnt = ng != 0u ? 4u * ng - 3u + (in1 ? 1u : 0u) + (in2 ? 1u : 0u) + (in3 ? 1u : 0u) : 0u;
    t.Rend = !in1 ? R1 : !in2 ? R2 : !in3 ? R3 : R4;
}

/* the position behind the first n codewords from Pfrom; *first_bits |= their first bits (bit 31).  (One lane a block: the
 * one that holds the block's last symbol and codewords behind it.) */
__device__ __forceinline__ uint32_t dreg_walk(uint32_t lut_addr, uint32_t lane4, const DregLeaves &lv, uint32_t Pfrom, uint32_t n, uint32_t *first_bits)
{
    uint32_t P = Pfrom, f = 0;
#pragma unroll 1
    while (__any(n != 0u)) {
        const uint32_t d = dreg_bits_at(P, lane4);
        uint32_t e = dreg_entry(lut_addr, d);
        if (lv.any) {
            const bool lg = n != 0u && (e & 0xffu) == 0u;
            if (__any(lg)) {
                const uint32_t e2 = dreg_long_entry(lv.code_a, lv.sym_a, lv.K, d);
                e = lg ? e2 : e;
            }
        }
        if (n != 0u) {
            f |= d;
            P = (P - (e & 0xffu)) & 0xff1fu;
            n--;
        }
    }
    *first_bits |= f;
    return P;
}

/* The lane's column: payload words word0 .. word0 + 11 of the segment that begins at bit seg0 (big-endian), word r at row
 * DREG_ROWS - 1 - r.  In two steps, so that the words are on their way while something else happens (the table build; the
 * sums and the checks of the segment before): dreg_request asks for thirteen dwords from the 4-byte aligned word that
 * holds the column's first byte - through a buffer resource over the readable bytes: a lane that wants nothing, or whose
 * words reach beyond the end, asks beyond it and gets zeros, no branch around the loads -, dreg_commit puts the bytes right
 * (one v_perm_b32 a word, whatever the payload's alignment) and writes the column; a lane whose words reach beyond
 * `readable` takes them byte by byte there, with zeros behind the end (the stream's last segment).  A column is read by
 * its own lane only: no barrier around any of this. */
struct DregWords {
    uint32_t d[DREG_ROWS + 1u];
    bool fast;
};
typedef uint32_t dreg_dwords4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ DregWords dreg_request(__amdgpu_buffer_rsrc_t rsrc, uint32_t seg0, uint32_t readable, uint32_t word0, bool wanted)
{
    DregWords q;
    const uint32_t first = (seg0 >> 3) + 4u * word0;                   /* payload byte of the column's word 0 (positions fit 32 bits: decode_payload_regs) */
    q.fast = wanted && first + 4u * (DREG_ROWS + 1u) <= readable;
    const uint32_t off = q.fast ? first : 0xffffff00u;       /* (the resource begins at the aligned word that holds payload byte 0) */
    static_assert(DREG_ROWS == 12u || DREG_ROWS == 15u, "thirteen or sixteen dwords");
    const dreg_dwords4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    const dreg_dwords4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0);
    const dreg_dwords4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32u, 0, 0);
    q.d[0] = v0.x; q.d[1] = v0.y; q.d[2] = v0.z; q.d[3] = v0.w;
    q.d[4] = v1.x; q.d[5] = v1.y; q.d[6] = v1.z; q.d[7] = v1.w;
    q.d[8] = v2.x; q.d[9] = v2.y; q.d[10] = v2.z; q.d[11] = v2.w;
#if DREG_ROWS == 15u
    const dreg_dwords4 v3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48u, 0, 0);
    q.d[12] = v3.x; q.d[13] = v3.y; q.d[14] = v3.z; q.d[15] = v3.w;
#else
    q.d[12] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 48u, 0, 0);
#endif
    return q;
}

__device__ __forceinline__ void dreg_commit(uint32_t *col, const DregWords &q, uint32_t sel, const uint8_t *pay, uint32_t seg0, uint32_t readable, uint32_t word0, bool wanted)
{
    /* (word by word: twelve results held for one block of stores are twelve registers more at the kernel's fullest point) */
    if (q.fast) {
#pragma unroll
        for (uint32_t r = 0; r < DREG_ROWS; r++) col[64u * (DREG_ROWS - 1u - r)] = __builtin_amdgcn_perm(q.d[r + 1], q.d[r], sel);
    }
    if (__builtin_expect(__ballot(wanted && !q.fast) != 0ull, 0)) {
        if (wanted && !q.fast) {
            const uint32_t first = (seg0 >> 3) + 4u * word0;
#pragma unroll 1
            for (uint32_t r = 0; r < DREG_ROWS; r++) col[64u * (DREG_ROWS - 1u - r)] = load_be32(pay, first + 4u * r, readable);
        }
    }
}

/* Runs of one byte value (decode_fast.hpp, dfast_run_at / dfast_run_jump, on this stage): does ONE codeword, repeated,
 * fill the column from position `pos` (register P0) to `hi` and a codeword further?  Its length, or 0. */
__device__ __forceinline__ uint32_t dreg_run_at(uint32_t lut_addr, uint32_t lane4, uint32_t P0, uint32_t pos, uint32_t hi)
{
    /* (one window after the other, three registers: this runs next to the sixteen that hold the lane's symbols) */
    uint32_t w = dreg_bits_at(P0, lane4);
    const uint32_t L = dreg_entry(lut_addr, w) & 31u;
    if ((w >> 31) != 0u || L == 0u) return 0u;
    bool same = true;
#pragma unroll 1
    for (uint32_t k = 0; k < (DREG_SUB_BITS + 31u) / 32u + 1u; k++) {
        const uint32_t wn = dreg_bits_at(P0 - ((k + 1u) << 8), lane4);       /* (32 bits on: one row down) */
        const bool counts = pos + 32u * k < hi;
        if (counts && __builtin_amdgcn_alignbit(w, wn, 32u - L) != w) same = false;
        w = wn;
    }
    return same ? L : 0u;
}

/* A lane that has just been put right and holds one codeword over and over is the beginning of a run as far as this wave
 * is concerned: the lanes to its right take the codeword starts that follow from it if THEIR shares hold the same
 * repetition from there on.  Returns (start, end, moved): a lane that moved runs its pass in the next round. */
__device__ __forceinline__ uint4 dreg_run_jump(uint32_t lut_addr, uint32_t lane4, uint32_t rbase, bool changed, bool dead, uint32_t hi, uint32_t lo, uint32_t pay_rel,
                                            uint32_t start, uint32_t end)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t myL = (changed && start < hi) ? dreg_run_at(lut_addr, lane4, dreg_gap(rbase - start), start, hi) : 0u;
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
    if (src >= 0 && src < (int)lane && !dead && !changed && lo >= P) {
        const uint32_t back = (lo - P) % L;
        cand = lo + (back ? L - back : 0u);
        pass = cand >= hi || (cand < pay_rel && dreg_run_at(lut_addr, lane4, dreg_gap(rbase - cand), cand, hi) == L);
    }
    int bad = (src >= 0 && src < (int)lane && !pass) ? (int)lane : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(bad, o);
        if ((int)lane >= o && t > bad) bad = t;
    }
    if (pass && bad <= src && cand != start) {
        const uint32_t cnt = (cand < hi) ? (hi - cand + L - 1u) / L : 0u;
        return make_uint4(cand, cand + cnt * L, 1u, 0u);
    }
    return make_uint4(start, end, 0u, 0u);
}

#define DREG_MAX_ROUNDS 64

/* block_excl_scan_u32 (util.hpp) without the barrier behind the reads: for partial words that nobody writes again before another
 * barrier has passed (the segment's sums: the next segment's rounds lie in between; the probe's extra sum has words of its own). */
template <int THREADS>
__device__ __forceinline__ uint32_t dreg_excl_scan(uint32_t v, uint32_t *s_part, uint32_t &total)
{
    constexpr int WAVES = THREADS / 64;
    static_assert(WAVES == 8 || WAVES == 4, "the waves' words are summed by the first lanes of a row of sixteen");
    const uint32_t wave = uni32(threadIdx.x >> 6);
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane_id() == 63) s_part[wave] = inc;
    __syncthreads();
    /* (the waves' words summed by the lanes - lane j reads word j mod WAVES, three DPP steps, two v_readlane: eight words read by
     *  every lane and added under eight comparisons of the wave's number were thirty instructions, half of them the comparisons'
     *  masks coming back from where the compiler had put them) */
    uint32_t x = s_part[lane_id() & (WAVES - 1)];
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
    if (WAVES == 8) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    /* row_shr:4 */
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, WAVES - 1);
    const uint32_t base = wave != 0u ? (uint32_t)__builtin_amdgcn_readlane((int)x, (int)wave - 1) : 0u;
    return base + inc - v;
}

/* do the waves' words (one a wave, written in front of the last barrier) hold anything but zeros? */
template <int WAVES>
__device__ __forceinline__ bool dreg_any_word(const uint32_t *words)
{
    return __ballot(words[lane_id() & (WAVES - 1)] != 0u) != 0ull;
}

/* ceil(n / d) for n < 2^31, d >= 1, a quotient below 2^18: v_rcp's estimate (off by a few units in the seventh digit) cut off and
 * put right by two comparisons - uniform values, five instructions where the division proper takes forty */
__device__ __forceinline__ uint32_t dreg_ceil_div(uint32_t n, uint32_t d)
{
    uint32_t q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d) * 0.999999f);
    q += q * d < n ? 1u : 0u;
    q += q * d < n ? 1u : 0u;
    return q;
}

/* a segment: where it begins (bit seg0 of the payload, a multiple of 32; its first codeword at seg0 + first) and the bits of a share */
struct DregSeg { uint32_t seg0, sb, first; bool hinted; };
template <int THREADS>
__device__ __forceinline__ DregSeg dreg_plan(uint32_t ts, bool trust, uint32_t shrink, uint32_t cap, uint32_t fixlen, bool probing, uint32_t hint_bytes, uint32_t pay_bytes)
{
    DregSeg g;
    const uint32_t pay_bits = pay_bytes * 8u;
    g.seg0 = ts & ~31u;
    g.first = (uint32_t)(ts - g.seg0);
    const uint32_t SUB = dmax<uint32_t>(cap >> shrink, 64u);
    uint32_t sb = SUB;
    g.hinted = uni32((probing && trust && hint_bytes * 8u > g.seg0) ? 1u : 0u) != 0u;      /* (a hint beyond the payload is 0 here) */
    if ((!probing || g.hinted) && pay_bits > g.seg0) {
        /* equal shares of what is left (decode_fast.hpp: the block's last segment as full as the others) */
        const uint32_t rem = (g.hinted ? hint_bytes * 8u : pay_bits) - g.seg0;
        if (rem < (1u << 31)) {
            /* (two integer divisions are some eighty vector instructions a segment: dreg_ceil_div.  Exact, because a quotient that
             *  IS a whole number must stay one: uniform bytes, 589 824 bits = 4 x 512 x 288, are four segments of 288 bits - not
             *  five of 231, and not four of 289 and a fifth for the crumbs) */
            const uint32_t nseg = dreg_ceil_div(rem, (uint32_t)THREADS * SUB);
            uint32_t even = dreg_ceil_div(rem, nseg * (uint32_t)THREADS);
            /* (a block whose codes all have one length - decode_payload_regs: shares of whole codewords) */
            if (fixlen != 0u) even = fixlen * dreg_ceil_div(even, fixlen);
            /* (not below 192 bits while the registers allow: a speculative track needs a hundred bits or so to fall into step,
             *  and a short payload spread thin over all the lanes - 4 KiB: 60 bits a lane - is put right one lane a round) */
            sb = dmin<uint32_t>(dmax<uint32_t>(even, 192u), SUB);
        }
    }
    g.sb = uni32(sb);
    return g;
}

/* A block's payload.  build_tables() (workgroup-uniform result) fills sh.lut with dfast_tables_from_tree<THREADS, SPEC, true>'s
 * table - it runs HERE, behind the request of the first segment's words, which then arrive while the tables are built.
 * Returns DREG_NO_TABLES when it declined (nothing was decoded: the caller takes another path), DREG_OK when block_len
 * symbols were written and everything the in-order decoder would have checked held, DREG_FAILED otherwise.  Arguments as
 * decode_payload_fast_impl (decode_fast.hpp): end_bits = the caller does not know where the payload ends (the raw-stream
 * probe: pay_bytes = the rest of the stream) and wants to be told, hint_bytes = where it probably ends. */
enum { DREG_NO_TABLES = 0, DREG_OK = 1, DREG_FAILED = 2, DREG_LONG = 3 };
template <int THREADS, bool LONG, class BuildTables>
__device__ __forceinline__ int decode_payload_regs_as(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes64, uint64_t readable64, uint64_t block_len64,
                                                   uint8_t *gout, uint64_t *end_bits, uint64_t hint_bytes64, BuildTables build_tables)
{
    /* Positions in 32 bits (round 6b: the 64-bit ones were two scalar registers each in a kernel that has eighty, and what did
     * not fit went through v_readlane - a vector instruction - every time it was needed): no code of this path is longer than
     * 32 bits, so a block of up to 2^26 symbols ends within 2^31 bits of its payload's start and nothing behind that is
     * anyone's business (the probe's "rest of the stream", an index entry with slack behind the block). */
    if (block_len64 > DREG_MAX_BLOCK) return DREG_NO_TABLES;                     /* (uniform) */
    const uint32_t block_len = (uint32_t)block_len64;
    const uint32_t pay_bytes = (uint32_t)dmin<uint64_t>(pay_bytes64, (uint64_t)block_len * 4u + 16u);
    const uint32_t readable = (uint32_t)dmin<uint64_t>(readable64, 0xfffffe00ull);
    const uint32_t hint_bytes = hint_bytes64 <= (uint64_t)pay_bytes ? (uint32_t)hint_bytes64 : 0u;      /* (a hint beyond the payload is none) */
    constexpr int WAVES = THREADS / 64;
    static_assert(offsetof(DecShared<THREADS>, pay) % 256 == 0, "a row of a column is 256 bytes at a multiple of 256: its number is a bit field of an LDS address");
    /* (the columns run on from pay + marks into ent and lr: the tree's entries are not this path's, and what the table build keeps
     *  in lr - the leaves' codes - it is done with before the first column is written) */
    static_assert((uint32_t)THREADS * DREG_ROWS * 4u <= sizeof(DecShared<THREADS>::pay) + sizeof(DecShared<THREADS>::mark) + sizeof(DecShared<THREADS>::ent) + sizeof(DecShared<THREADS>::lr),
                  "the columns fit pay + marks + ent + lr");
    static_assert(offsetof(DecShared<THREADS>, lr) == offsetof(DecShared<THREADS>, ent) + sizeof(DecShared<THREADS>::ent) &&
                  offsetof(DecShared<THREADS>, ent) == offsetof(DecShared<THREADS>, mark) + sizeof(DecShared<THREADS>::mark), "one area");
    static_assert(sizeof(DecShared<THREADS>) <= 65536 - 256, "rows are numbered in eight bits");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = (int)uni32((uint32_t)tid >> 6);
    uint32_t *slice = sh.pay + (uint32_t)wave * (DREG_ROWS * 64u);
    const uint32_t slice_a = uni32((uint32_t)(uintptr_t)(dreg_lds_words)slice);
    const uint32_t lane4 = 4u * (uint32_t)lane;
    const uint32_t r_top = 32u * ((slice_a >> 8) + DREG_ROWS - 1u);         /* R of the first bit of a column's word 0 */
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(dreg_lds_halves)sh.lut;
    const uint32_t pay_bits = pay_bytes * 8u;
    /* the payload through a buffer resource from the aligned word that holds its first byte to the end of what may be read */
    const uintptr_t pay_a = (uintptr_t)uni64((uint64_t)(uintptr_t)pay);
    const uint32_t mis = (uint32_t)(pay_a & 3u);
    const uint32_t sel = (mis << 24) | ((mis + 1u) << 16) | ((mis + 2u) << 8) | (mis + 3u);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(pay_a - mis), (short)0, (int)(readable + mis), 0x00020000);
    /* shares cut for DREG_TARGET_SYMS symbols at the block's bits per symbol (the probe: at the hint's; without one, whole) */
    uint32_t cap = DREG_SUB_BITS, fixlen = 0;
    {
        const uint32_t known = end_bits ? hint_bytes * 8u : pay_bits;
        if (known != 0u) {
            const float per_sym = (float)known / (float)block_len;
            const float est = (float)DREG_TARGET_SYMS * per_sym;
            /* Blocks of few bits a symbol: the sixteen registers hold their shares down to a hundred bits and less, in which a
             * speculative track falls into step late.  With the block index they are still this path's down to shares of 64 bits
             * (`tools/time_lowentropy.py`, ms per GiB in 64 KiB / 1 MiB blocks, this path against decode_fast.hpp's scans: two byte
             * values 0.68 / 0.56 against 1.23 / 1.33, zeros with 1 % of random bytes 0.92 / 0.74 against 1.40 / 1.00, geometric bytes
             * 2.55 / 1.66 against 4.28 / 2.02).  The PROBE of a raw stream keeps them away unless all their codes have one length
             * (two byte values 0.90 against 1.39 ms a call, four 0.87 against 1.14): what the rounds do not settle there goes to the
             * exact decoder, and geometric bytes in 64 KiB blocks hold blocks that take THAT milliseconds (11.4 against 4.5 ms a
             * call with them through here; round 6 did not get to the bottom of either figure). */
            const uint32_t Lfix = (uint32_t)(per_sym + 0.001f);
            const bool one_length = Lfix >= 2u && Lfix <= DEC_LUT_BITS && known - Lfix * block_len < 8u;
            if (est < (float)((end_bits && !one_length) ? DREG_MIN_SHARE_BITS_PROBE : DREG_MIN_SHARE_BITS)) return DREG_NO_TABLES;
            cap = est >= (float)DREG_SUB_BITS ? DREG_SUB_BITS : dmax<uint32_t>((uint32_t)est, 64u);
            /* Incompressible bytes: every code 9 bits (8 and the root's 0), the payload block_len x 9 bits and a byte's padding.
             * Shares of whole codewords then - a lane's own first bit IS a codeword's and the speculative pass the only one,
             * where codes of one length never fall into step by themselves (1 GiB of uniform bytes: 0.78 ms against 1.06 with
             * shares of 384 bits).  A block that only looks like it loses nothing. */
            if (one_length) {
                fixlen = Lfix;
                cap = Lfix * (uint32_t)(((float)cap + 0.5f) * __builtin_amdgcn_rcpf((float)Lfix));
            }
        }
    }
    cap = uni32(cap);
    fixlen = uni32(fixlen);
    /* (plain functions of plain values, no closures: a closure over a dozen locals is an object the compiler keeps whole -
     *  sixteen registers in a row, spilled and reloaded as one) */
#define DREG_PLAN(TS, TRUST, SHRINK) dreg_plan<THREADS>((TS), (TRUST), (SHRINK), cap, fixlen, end_bits != nullptr, hint_bytes, pay_bytes)
#define DREG_REL(G) (pay_bits > (G).seg0 ? pay_bits - (G).seg0 : 0u)       /* payload bits from seg0 on */
#define DREG_WORD0(G) ((tid == 0 ? (G).first : (uint32_t)tid * (G).sb) >> 5)
#define DREG_WANTED(G) ((uint32_t)tid * (G).sb < DREG_REL(G))
    uint32_t shrink = 0;                                               /* (uniform) halvings of the shares after a lane ran out of registers */
    uint32_t true_start = 0, produced = 0;
    uint32_t seg0_last = 0;
    bool ok = true;
    bool lanes_ok = true;                                              /* (per lane) every segment's lane_ok so far */
    bool trust = true;
    unsigned long long pt0 = DPROF_T();
    DregSeg g = DREG_PLAN(0, true, 0);
    const DregWords q = dreg_request(rsrc, g.seg0, readable, DREG_WORD0(g), DREG_WANTED(g));
    DPROF_ADD(10, pt0); pt0 = DPROF_T();
    {
        const int built = build_tables();                              /* (uniform) 0: declined, 1: the table stands, 2: and the block has codes beyond it */
        if (built == 0) return DREG_NO_TABLES;
        if (!LONG && built == 2) return DREG_LONG;                     /* nothing was decoded; the table stands */
    }
    typedef DregLeafLdsOf<DecShared<THREADS>> LF;
    DregLeaves lv;
    lv.code_a = (uint32_t)(uintptr_t)(dreg_lds_words)LF::code(sh);
    lv.sym_a = (uint32_t)(uintptr_t)(dreg_lds_bytes)LF::sym(sh);
    lv.K = uni32(sh.fastk);
    lv.any = LONG;
    const uint32_t Rfloor = r_top - 32u * (DREG_ROWS - 1u) - 32u;      /* R behind the column's last bit */
    DPROF_ADD(11, pt0); pt0 = DPROF_T();
    dreg_commit(slice + lane, q, sel, pay, g.seg0, readable, DREG_WORD0(g), DREG_WANTED(g));
    DPROF_ADD(12, pt0);
    bool have = true;                                                  /* (uniform) the columns hold segment g */
    while (produced < block_len) {
        if (true_start >= pay_bits) { ok = false; DFAST_DBG(0, 1); break; }
        unsigned long long pt = DPROF_T();
        DregSyms sym;                                                  /* (the segment's: nothing of them lives from one segment to the next) */
        if (!have) {                                                   /* a segment done again with other shares: staged on the spot */
            g = DREG_PLAN(true_start, trust, shrink);
            const DregWords q2 = dreg_request(rsrc, g.seg0, readable, DREG_WORD0(g), DREG_WANTED(g));
            dreg_commit(slice + lane, q2, sel, pay, g.seg0, readable, DREG_WORD0(g), DREG_WANTED(g));
        }
        have = false;
        const uint32_t seg0 = g.seg0;
        const uint32_t sb = g.sb, first = g.first;
        const bool hinted = g.hinted;
        const uint32_t pay_rel = DREG_REL(g);
        const uint32_t hi = ((uint32_t)tid + 1u) * sb;
        const uint32_t lo = tid == 0 ? first : hi - sb;
        const uint32_t rbase = r_top + 32u * (lo >> 5);               /* R = rbase - position */
        uint32_t start = lo;
        bool dead = hi - sb >= pay_rel;
        const uint32_t remaining = block_len - produced;
        bool guessed = false;
        if (hinted) {
            const uint32_t bound = hint_bytes * 8u - seg0;
            if (!dead && hi - sb >= bound) dead = true;
            guessed = (uint32_t)(THREADS - 1) * sb >= bound;
        } else if (end_bits && trust && produced != 0) {
            const float est = (float)remaining * ((float)true_start / (float)produced);
            const float lim_f = (float)first + est * 1.0625f + 1024.0f;
            const uint32_t bound = lim_f < 4.0e9f ? (uint32_t)lim_f : 0xffffffffu;
            if (!dead && hi - sb >= bound) dead = true;
            guessed = (uint32_t)(THREADS - 1) * sb >= bound;
        }
        /* codewords that start at or behind the payload's end are nobody's */
        const uint32_t hi_eff = dmin<uint32_t>(hi, pay_rel);
        uint32_t Phi = dreg_gap(rbase - (dead ? lo : hi_eff));
        DregTrack t;
        t.Rend = 0; t.cnt = 0; t.bad = false; t.over = false;
        uint32_t end = hi, cnt = 0;
        bool need = !dead;
        int rounds = 0;
        DPROF_ADD(1, pt); pt = DPROF_T();
        for (;;) {
            if (__ballot(need)) {
                DFAST_DBGW(8, 1);
                DFAST_DBGW(rounds == 0 ? 6 : rounds == 1 ? 7 : 14, 1);
                dreg_pass<LONG>(lut_addr, lane4, lv, Rfloor, dreg_gap(rbase - (dead ? lo : start)), Phi, sym, t);
                /* (a track of more than 64 look-ups - a speculative one through entries that leave the tree, bit by bit; or,
                 *  when the rounds are over, the lane's true one - ends where the share does for now: its neighbour is not
                 *  sent in front of its own column) */
                end = (dead || t.over) ? hi : rbase - t.Rend;
                cnt = dead ? 0u : t.cnt;
            }
            if (rounds == 0 && end_bits && trust && !hinted) {
                /* (the probe without a hint) the speculative counts are right to a few symbols either way: a lane in front of
                 * which they already hold the rest of the block and a margin is taken for dead */
                uint32_t spec_total;
                const uint32_t exs = dreg_excl_scan<THREADS>(cnt, sh.wtile + 2 * WAVES, spec_total);
                if (!dead && exs >= remaining + 128u + ((uint32_t)tid >> 2)) {
                    dead = true;
                    end = hi;
                    cnt = 0;
                    Phi = dreg_gap(rbase - lo);
                }
                guessed = guessed || uni32(spec_total) >= remaining + 128u;
            }
            if (lane == 63) sh.wend[wave] = end;
            __syncthreads();
            if (rounds == 0) { DPROF_ADD(2, pt); pt = DPROF_T(); }
            uint32_t ns = wave_up1_u32(end);
            if (lane == 0) ns = (tid == 0) ? first : sh.wend[wave - 1];
            const bool changed = (ns != start) && !dead;
            if (changed) start = ns;
            bool jumped = false;
            if (rounds >= DFAST_JUMP_FROM_ROUND && __ballot(changed)) {
                DFAST_DBGW(15, 1);
                const uint4 r = dreg_run_jump(lut_addr, lane4, rbase, changed, dead, hi_eff, hi - sb, pay_rel, start, end);
                start = r.x; end = r.y; jumped = r.z != 0u;
            }
            need = changed || jumped;
            /* does any lane of the workgroup go on?  (One barrier: a wave's word is written behind the barrier above and read
             * behind this one, and the next round's barrier above lies in front of its next writer; everyone has read sh.wend.) */
            const uint32_t wave_needs = __ballot(need) != 0ull ? 1u : 0u;   /* (the vote of the whole wave: outside the lane's branch) */
            if (lane == 0) sh.wtile[wave] = wave_needs;
            __syncthreads();
            if (!dreg_any_word<WAVES>(sh.wtile)) break;
            DFAST_DBG(10, 1);
            if (++rounds > DREG_MAX_ROUNDS) { ok = false; DFAST_DBG(1, 1); break; }
        }
        if (!ok) break;
        DPROF_ADD(3, pt); pt = DPROF_T();
        /* where the next segment begins is known: its words are asked for NOW, in front of this segment's stores - loads
         * and stores come back in the order they went out, and behind the stores the words would wait for every one of
         * them - and arrive under the sums */
        const uint32_t last_end = uni32(sh.wend[WAVES - 1]);
        const uint32_t next_start = seg0 + last_end;
        const DregSeg gn = DREG_PLAN(next_start, true, shrink);
        /* (behind the payload's last byte there is no next segment: its lanes ask beyond the resource and nothing is fetched -
         *  one request in five was for the first kilobytes of the NEXT block, which that block's workgroup reads again) */
        const bool more = uni32(next_start + 7u < (gn.hinted ? hint_bytes * 8u : pay_bits) ? 1u : 0u) != 0u;
        const DregWords qn = dreg_request(rsrc, gn.seg0, readable, DREG_WORD0(gn), more && DREG_WANTED(gn));
        /* the counts' sum; above it the lanes whose true track ran out of registers (counts: at most 64 a lane, 2^15 a segment) */
        uint32_t seg_total;
        const uint32_t ex = dreg_excl_scan<THREADS>(cnt | ((uint32_t)(t.over && !dead) << 16), sh.part, seg_total) & 0xffffu;
        seg_total = uni32(seg_total);
        if ((seg_total >> 16) != 0u) {                                 /* a share of more than 64 codewords: the segment again, shares of half the bits */
            DFAST_DBG(9, 1);
            if ((cap >> shrink) <= 64u) { ok = false; break; }         /* (shares of 64 bits hold what a lane can take: this is not a payload of this tree - the exact decoder says what it is) */
            shrink++;
            continue;
        }
        if (guessed) DFAST_DBG(12, 1);
        if (guessed && seg_total < remaining) {             /* a guess that did not hold: the segment again, without */
            DFAST_DBG(13, 1);
            trust = false;
            continue;
        }
        trust = true;
        const uint32_t take = dmin<uint32_t>(seg_total, remaining);
        uint32_t quota = 0;
        if (ex < take) {
            quota = take - ex;
            if (quota > cnt) quota = cnt;
        }
        DPROF_ADD(4, pt); pt = DPROF_T();
        bool lane_ok = true;
        uint32_t qe = end;                                             /* the position behind the lane's last symbol of the block */
        const bool partial = quota != 0u && quota < cnt;               /* the block ends inside this lane's codewords */
        /* such a lane's end, and whether a first bit of 1 that a lane's pass saw belongs to one of ITS codewords (the pass looks at
         * whole iterations: the last one's may begin behind the share's end - the right neighbour's, who sees them too, or nobody's
         * behind the block's last symbol): its codewords again, one by one.  A lane or two a block. */
        const bool again = quota != 0u && (partial || t.bad);
        if (__ballot(again)) {
            uint32_t fb = 0;
            const uint32_t Pq = dreg_walk(lut_addr, lane4, lv, dreg_gap(rbase - start), again ? quota : 0u, &fb);
            if (again) {
                qe = rbase - dreg_ungap(Pq);
                lane_ok = (fb >> 31) == 0u;
            }
        }
        /* this segment's columns have been read for the last time: the next segment's words into them (the wait for them
         * counts only what went out before them) */
        DPROF_ADD(7, pt); pt = DPROF_T();
        if (more && produced + take < block_len) {
            dreg_commit(slice + lane, qn, sel, pay, gn.seg0, readable, DREG_WORD0(gn), DREG_WANTED(gn));
            have = true;
        }
        DPROF_ADD(8, pt); pt = DPROF_T();
        if (quota != 0u) {
            if (qe > pay_rel) lane_ok = false;                         /* a codeword of the block needs bits past the payload */
            if (end_bits && ex + quota == take && take == remaining) sh.qend = qe;
            /* the registers: all of the lane's symbols and the whole last dword still inside this block's output, or
             * whole dwords and the rest byte by byte */
            uint8_t *gp = gout + (produced + ex);
            const bool whole = !partial && produced + ex + ((quota + 3u) & ~3u) <= block_len;
            const uint32_t nd = whole ? (quota + 3u) >> 2 : quota >> 2;
            typedef uint32_t __attribute__((aligned(1))) unaligned_u32;
            typedef uint32_t unaligned_q4 __attribute__((ext_vector_type(4), aligned(1)));
#define DREG_QUAD(Q4, A, B, C, D)                                                                              \
            if (nd >= 4u * (Q4) + 4u) {                                                                       \
                unaligned_q4 v4;                                                                              \
                v4.x = sym.A; v4.y = sym.B; v4.z = sym.C; v4.w = sym.D;                                       \
                *reinterpret_cast<unaligned_q4 *>(gp + 16u * (Q4)) = v4;                                      \
            } else {                                                                                          \
                if (nd > 4u * (Q4)) *reinterpret_cast<unaligned_u32 *>(gp + 16u * (Q4)) = sym.A;              \
                if (nd > 4u * (Q4) + 1u) *reinterpret_cast<unaligned_u32 *>(gp + 16u * (Q4) + 4u) = sym.B;    \
                if (nd > 4u * (Q4) + 2u) *reinterpret_cast<unaligned_u32 *>(gp + 16u * (Q4) + 8u) = sym.C;    \
            }
            DREG_QUAD(0, s0, s1, s2, s3)
            DREG_QUAD(1, s4, s5, s6, s7)
            DREG_QUAD(2, s8, s9, s10, s11)
#if DREG_ITERS == 16
            DREG_QUAD(3, s12, s13, s14, s15)
#endif
#undef DREG_QUAD
            if (!whole && (quota & 3u) != 0u) {
                const uint32_t kq = quota >> 2;
                uint32_t tw = sym.s0;
#define DREG_PICK(K, S) tw = kq == (K) ? sym.S : tw;
                DREG_PICK(1, s1) DREG_PICK(2, s2) DREG_PICK(3, s3) DREG_PICK(4, s4) DREG_PICK(5, s5) DREG_PICK(6, s6) DREG_PICK(7, s7)
                DREG_PICK(8, s8) DREG_PICK(9, s9) DREG_PICK(10, s10) DREG_PICK(11, s11)
#if DREG_ITERS == 16
                DREG_PICK(12, s12) DREG_PICK(13, s13) DREG_PICK(14, s14) DREG_PICK(15, s15)
#endif
#undef DREG_PICK
                for (uint32_t c = 0; c < (quota & 3u); c++) gp[(quota & ~3u) + c] = (uint8_t)(tw >> (8u * c));
            }
        }
        DPROF_ADD(9, pt); pt = DPROF_T();
        /* (is every lane content?  Asked once, behind the block's last segment - round 6b; until then a vote and a barrier a
         *  segment, and a second barrier in the sums.  Neither is needed: the columns are their lanes' own, sh.wend was read for
         *  the last time in front of the sums' barrier, the sums' partial words are written again only behind the next rounds'
         *  barriers - the probe's extra sum in round 0 has words of its own for that reason.) */
        lanes_ok = lanes_ok && lane_ok;
        DFAST_DBG(11, 1);
        produced += take;
        if (take == 0) { ok = false; DFAST_DBG(3, 1); break; }
        if (produced == block_len) seg0_last = seg0;
        true_start = next_start;
        g = gn;
    }
    {
        const uint32_t wave_bad = __ballot(!lanes_ok) != 0ull ? 1u : 0u;
        if (lane == 0) sh.wtile[WAVES + wave] = wave_bad;
        __syncthreads();                                               /* (also: sh.qend is everyone's) */
        if (dreg_any_word<WAVES>(sh.wtile + WAVES)) { ok = false; DFAST_DBG(2, 1); }
    }
    if (ok && end_bits) *end_bits = (uint64_t)seg0_last + (uint64_t)uni32(sh.qend);
    return ok ? DREG_OK : DREG_FAILED;
#undef DREG_PLAN
#undef DREG_REL
#undef DREG_WORD0
#undef DREG_WANTED
}

/* A block's payload: by the pass that does not ask for codes beyond the table, and - when the table build says the block has some -
 * by the one that does (two instances of everything above: one pass with the question in it was 3 % slower on blocks that have
 * none, two passes chosen from inside the round loop 30 % - the sixteen registers came out of the choice as copies).
 * build_tables() -> 0: declined, 1: the table stands, 2: and the block has codes beyond it. */
template <int THREADS, class BuildTables>
__device__ __forceinline__ int decode_payload_regs(DecShared<THREADS> &sh, const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, uint64_t block_len,
                                                   uint8_t *gout, uint64_t *end_bits, uint64_t hint_bytes, BuildTables build_tables)
{
    int r = decode_payload_regs_as<THREADS, false>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, build_tables);
    if (r == DREG_LONG) r = decode_payload_regs_as<THREADS, true>(sh, pay, pay_bytes, readable, block_len, gout, end_bits, hint_bytes, []() { return 2; });
    return r;
}

/* The in-order chain of decode.hpp (decode_chain_kernel: the block loop of src/decoder.c:218-276, one workgroup) with the
 * lean decoder in front of the exact one: a block with an encoder-shaped tree goes through decode_payload_regs first - told
 * where the stream ends as a hint at where its payload does, which is right for the stream of one block that a small call
 * usually is - and only what that cannot vouch for (a damaged block, another shape of tree, a one-symbol block) through
 * decode_block, which has the reference's error codes and byte counts.  What hufgpu_decode_small runs (streams of up to
 * 32 KiB: the call is a latency, and the exact decoder's 54 instructions a symbol on one workgroup were most of it -
 * 4 KiB 76 us, 64 KiB 141 us a call in round 5).  result[] as decode_chain_kernel's. */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void decode_chain_lean_kernel(const uint8_t *__restrict__ stream, uint64_t avail, uint64_t length,
                                                                    int max_tree_len, uint8_t *__restrict__ out, uint64_t out_cap,
                                                                    uint64_t *__restrict__ result)
{
    __shared__ DecShared<THREADS> sh;
    uint64_t rd = 0, wr = 0, nblk = 0;
    uint64_t good_rd = 0, good_wr = 0;
    int err = HUFE_OK;
    unsigned long long pk = DPROF_T();                                /* (tools/phase_fast.py small: slots 13 = the kernel, 14 = a block's header, 15 = its payload) */
    while (length > rd) {                                             /* decoder.c:218 */
        unsigned long long ph = DPROF_T();
        good_rd = rd;
        good_wr = wr;
        if (avail - rd < 8) { err = HUFE_RW; break; }                 /* decoder.c:220-224 */
        const uint64_t block_len = uni64(load_u64_unaligned(stream + rd));
        rd += 8;
        if (avail - rd < 2) { err = HUFE_RW; break; }                 /* decoder.c:231-234 */
        const int16_t tl = (int16_t)uni32((uint32_t)stream[rd] | ((uint32_t)stream[rd + 1] << 8));
        rd += 2;
        if (tl < 0 || tl > max_tree_len) { err = HUFE_OVERFLOW; break; }   /* decoder.c:237-239 */
        if (avail - rd < 2ull * (uint64_t)tl) { err = HUFE_RW; break; }    /* decoder.c:248-252 */
        const uint8_t *tree = stream + rd;
        rd += 2ull * (uint64_t)tl;
        if (block_len == 0) { nblk++; continue; }
        uint64_t want = block_len;
        if (want > (avail - rd) * 8ull) want = (avail - rd) * 8ull + 1;
        const bool capped = want > out_cap - wr;
        if (capped) want = out_cap - wr;
        if (want > HUF_MAX_BLOCK_LEN) { err = HUFE_ARGUMENT; break; }
        uint64_t end_bits = 0, produced = 0;
        if (want == block_len && tl >= 9 && tl <= HUF_TREE_MAX && block_len >= 256u) {
            const uint64_t hint = length > rd ? length - rd : 0;
            __syncthreads();
            DPROF_ADD(14, ph); ph = DPROF_T();
            /* (blocks of a few KiB have codes beyond the table's 12 bits - a byte seen once in 4 096 - and the path of this file
             *  declines them: decode_fast.hpp's lean decoder walks such codes, with the tables of the tree's walk) */
            const int lean = block_len < DREG_MIN_BLOCK ? DREG_NO_TABLES :
                decode_payload_regs<THREADS>(sh, stream + rd, avail - rd, avail - rd, block_len, out + wr, &end_bits, hint,
                                             [&]() { return dfast_tables_from_tree<THREADS, true, true>(sh, tree, tl) ? (uni32(sh.l2n) != 0u ? 2 : 1) : 0; });
            bool done = lean == DREG_OK;
            if (lean == DREG_NO_TABLES) {
                int leaf = -1;
                end_bits = 0;
                __syncthreads();
                done = dec_build_tables<THREADS, true>(sh, tree, tl, &leaf) == HUFE_OK && leaf < 0 &&
                       decode_payload_dfast<THREADS>(sh, stream + rd, avail - rd, avail - rd, block_len, out + wr, &end_bits, hint);
            }
            DPROF_ADD(15, ph);
            if (done) {
                rd += (end_bits + 7) >> 3;
                wr += block_len;
                nblk++;
                continue;
            }
            end_bits = 0;
            __syncthreads();
        }
        if (want) err = decode_block<THREADS>(sh, tree, tl, want, avail - rd, out + wr, &end_bits, &produced);
        if (err != HUFE_OK) { wr += produced; break; }   /* symbols before the failure stay delivered */
        if (capped) { wr += want; err = HUFE_MEMORY; break; }
        rd += (end_bits + 7) >> 3;
        wr += block_len;
        nblk++;
    }
    DPROF_ADD(13, pk);
    if (threadIdx.x == 0) {
        result[0] = (uint64_t)err;
        result[1] = wr;
        result[2] = rd;
        result[3] = nblk;
        result[4] = (err == HUFE_OK) ? rd : good_rd;
        result[5] = (err == HUFE_OK) ? wr : good_wr;
    }
}

}  // namespace hufgpu
