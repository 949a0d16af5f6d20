/* pack.hpp - pack_kernel: block header and MSB-first bit-pack (src/encoder.c:85-131, 322-339).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "offsets.hpp"

namespace hufgpu {

/* ======================================================================================
 * pack_kernel - replaces the header emission (src/encoder.c:322-339) and __huf_encode_block
 * + huf_bit_write (src/encoder.c:85-131, src/bufio.c:18-23).
 *
 * One workgroup per block.  The block record [u64 len][i16 tree_len][tree][payload] is a bit
 * string that starts at byte offsets[blk] of the output; it is produced as big-endian 32-bit
 * words aligned with the 4-byte words of the destination (stream bit b, MSB first inside each
 * byte, is bit 31-(b&31) of word b>>5; a finished word is byte-swapped and stored).
 *
 * Payload tiles of THREADS*32 symbols.  Every lane loads 32 contiguous input bytes, looks the
 * codes up in the LDS table and keeps them in registers; the workgroup prefix-sums the per-lane
 * bit counts; then each lane shifts its codes through a 64-bit accumulator and stores every
 * word that ENDS inside its bit range straight to HBM.  32 symbols are at least 32 bits, so
 * every lane owns at least one word end: the only thing a lane needs from its left neighbour
 * is the neighbour's unfinished tail (< 32 bits), one __shfl_up (LDS for the wave seams, the
 * header tail / previous tile for lane 0).  No LDS image, no atomics.  The record's first and
 * last word are byte-masked because neighbouring blocks own the rest of those words.
 * ==================================================================================== */
#define PACK_SPT 32
#define PACK_STAGE_WORDS 3328          /* 13 KiB: a 256x32-symbol tile at up to ~12.9 bits per symbol */

/* Sub-index of a block's payload (in-process, like the block index; the wire format is untouched):
 * the encoder knows where every lane's PACK_SPT symbols start, the decoder would otherwise have to
 * find out by decoding everything twice (self-synchronisation).  Per group of HUF_SUB_GROUP symbols
 * the number of payload bits they occupy, per tile of HUF_SUB_TILE symbols the payload bit the tile
 * starts at (a tile = the 64 groups one wavefront of the decoder takes at a time).  Block b owns groups [b * gpb, (b+1) * gpb) and tiles [b * tpb, (b+1) * tpb), gpb/tpb =
 * ceil(blocksize / group or tile).  The decoder VERIFIES what it is told (a group must decode to
 * exactly its bit count), so a wrong or stale sub-index costs time, never correctness. */
#define HUF_SUB_GROUP PACK_SPT
#define HUF_SUB_TILE  (64 * PACK_SPT)     /* 2 048 symbols: what one wave decodes at a time (decode_sub.hpp; round 4: 8 192, and the decoder
                                            summed a block's group counts in LDS to find its waves' starts) */
struct HufSubIndex {
    uint64_t *tile_bits;      /* NULL = no sub-index */
    uint16_t *group_bits;
    uint8_t *lens;            /* [nblocks][256] code length of every byte value (0: absent): lets the decoder
                                 build its tables in a few parallel steps and CHECK them against the stream's
                                 tree (decode_sub.hpp, dsub_fast_tables) instead of walking that tree */
    uint64_t gpb, tpb;
};

template <typename CodeT>
struct PackAcc {
    uint64_t acc;       /* right-aligned bits not yet emitted */
    uint32_t nacc;      /* number of them (< 32 between pushes) */
    uint32_t first;     /* first finished word (its leading bits belong to the left neighbour) */
    bool have_first;
    uint32_t *gw;       /* where the next finished word goes (LDS stage or HBM, fixed per tile) */

    __device__ __forceinline__ void emit(uint32_t word)
    {
        if (!have_first) { first = word; have_first = true; }
        else *gw = __builtin_bswap32(word);
        gw++;
    }
    __device__ __forceinline__ void push32(uint32_t code, uint32_t len)     /* len <= 32 */
    {
        acc = (acc << len) | code;
        nacc += len;
        if (nacc >= 32) {
            nacc -= 32;
            emit((uint32_t)(acc >> nacc));
        }
    }
    /* two codes of at most 16 bits each as ONE push: half as many shifts of the accumulator and
     * "is a word finished?" tests (a wave runs the emit block whenever any of its lanes finishes a
     * word, i.e. at nearly every test) */
    __device__ __forceinline__ void push_pair(uint32_t e0, uint32_t e1)
    {
        const uint32_t l1 = e1 & 0xffu;
        push32(((e0 >> 8) << l1) | (e1 >> 8), (e0 & 0xffu) + l1);
    }
    /* three codes of at most 10 bits each (uniform bytes: 9) */
    __device__ __forceinline__ void push_triple(uint32_t e0, uint32_t e1, uint32_t e2)
    {
        const uint32_t l1 = e1 & 0xffu, l2 = e2 & 0xffu;
        push32(((((e0 >> 8) << l1) | (e1 >> 8)) << l2) | (e2 >> 8), (e0 & 0xffu) + l1 + l2);
    }
    __device__ __forceinline__ uint32_t tail() const { return (uint32_t)(acc & ((1ull << nacc) - 1ull)); }
    __device__ __forceinline__ void push(CodeT e)
    {
        uint32_t len = (uint32_t)(e & 0xffu);
        if constexpr (sizeof(CodeT) == 8) {
            const uint64_t c = e >> 8;
            if (len > 32) {                      /* long code: high part first */
                push32((uint32_t)(c >> 32), len - 32);
                len = 32;
            }
            push32((uint32_t)c, len);
        } else {
            push32((uint32_t)(e >> 8), len);
        }
    }
};

/* Codes of at most 24 bits (the short-code launch; blocks of any launch whose longest code is that short): ONE
 * 32-bit register of bits not yet emitted.  At most 31 of them are valid between pushes (what lies above is
 * history on its way out), a push adds at most 31, and the word that becomes complete is cut from the register
 * pair {bits that left the register, register} - no 64-bit shift anywhere: half the registers of the 64-bit
 * accumulator, cheaper instructions, and nothing for the gfx950 last-VGPR shift hazard to bite. */
#ifndef PACK_ACC64           /* (-DPACK_ACC64: the 64-bit accumulator for every code width, as in round 2 - the build that
                                showed the hazard at seven waves per SIMD, kept for tests/test_isa_check.py) */
template <>
struct PackAcc<uint32_t> {
    uint32_t acc;       /* right-aligned bits not yet emitted (the low `nacc` of them count) */
    uint32_t nacc;      /* number of them (< 32 between pushes) */
    uint32_t first;     /* first finished word (its leading bits belong to the left neighbour) */
    bool have_first;
    uint32_t *gw;       /* where the next finished word goes (LDS stage or HBM, fixed per tile) */

    __device__ __forceinline__ void emit(uint32_t word)
    {
        if (!have_first) { first = word; have_first = true; }
        else *gw = __builtin_bswap32(word);
        gw++;
    }
    __device__ __forceinline__ void push32(uint32_t code, uint32_t len)     /* len <= 31 */
    {
        const uint32_t out = acc >> ((32u - len) & 31u);     /* the bits the shift pushes out (used only when len >= 1) */
        acc = (acc << len) | code;
        nacc += len;
        if (nacc >= 32) {
            nacc -= 32;
            emit(__builtin_amdgcn_alignbit(out, acc, nacc));
        }
    }
    __device__ __forceinline__ void push_pair(uint32_t e0, uint32_t e1)      /* two codes of at most 15 bits each */
    {
        const uint32_t l1 = e1 & 0xffu;
        push32(((e0 >> 8) << l1) | (e1 >> 8), (e0 & 0xffu) + l1);
    }
    __device__ __forceinline__ void push_triple(uint32_t e0, uint32_t e1, uint32_t e2)   /* three codes of at most 10 bits each */
    {
        const uint32_t l1 = e1 & 0xffu, l2 = e2 & 0xffu;
        push32(((((e0 >> 8) << l1) | (e1 >> 8)) << l2) | (e2 >> 8), (e0 & 0xffu) + l1 + l2);
    }
    __device__ __forceinline__ void push(uint32_t e) { push32(e >> 8, e & 0xffu); }
    __device__ __forceinline__ uint32_t tail() const { return acc & ((1u << nacc) - 1u); }
};
#endif

/* byte j of the block header (encoder.c:325-339, little-endian fields) */
__device__ __forceinline__ uint32_t header_byte(uint32_t j, uint64_t block_len, uint32_t tree_len,
                                                const int16_t *__restrict__ tb)
{
    if (j < 8) return (uint32_t)(block_len >> (8 * j)) & 0xffu;
    if (j < 10) return (tree_len >> (8 * (j - 8))) & 0xffu;
    const uint16_t e = (uint16_t)tb[(j - 10) >> 1];
    return (e >> (8 * (j & 1))) & 0xffu;
}

/* The block header (encoder.c:322-339) and - for a block of one distinct byte - its whole payload.  Whole aligned
 * words are stored here, the unfinished last word becomes the incoming tail of the payload's first lane
 * (s_tail[WAVES]).  Returns false when the record is complete (one distinct byte); *hdr_end_out = the header's end
 * in bytes from A0. */
template <int THREADS>
__device__ __forceinline__ bool pack_header(uint64_t len, const int16_t *__restrict__ tb, uint32_t tree_len,
                                            uint8_t *g_a0, uint32_t *g_w0, uint32_t rec_lo, uint64_t rec_hi,
                                            uint32_t *s_tail, uint32_t *hdr_end_out)
{
    constexpr int WAVES = THREADS / 64;
    const int tid = (int)threadIdx.x;
    /* ---- header: whole aligned words are stored here, the unfinished last word becomes the
     *      incoming tail of the payload's first lane ---- */
    const uint32_t hdr_bytes = HUF_HEADER_FIXED + 2u * tree_len;
    const uint32_t hdr_end = rec_lo + hdr_bytes;                 /* relative to A0 */
    for (uint32_t w = tid; w < (hdr_end >> 2); w += THREADS) {
        uint32_t v = 0;                                          /* little-endian memory word */
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t bp = 4 * w + k;
            if (bp >= rec_lo) v |= header_byte(bp - rec_lo, len, tree_len, tb) << (8 * k);
        }
        if (4 * w >= rec_lo) g_w0[w] = v;
        else {
            for (uint32_t k = rec_lo - 4 * w; k < 4; k++) g_a0[4 * w + k] = (uint8_t)(v >> (8 * k));
        }
    }
    if (tree_len == 5) {
        /* One distinct byte in the block: its code is the single bit 0 (tree.c:410-413 with one
         * leaf), so the payload is ceil(len/8) zero bytes - nothing of the input needs reading
         * again (the histogram already saw it). */
        for (uint32_t bp = (hdr_end & ~3u) + tid; bp < hdr_end; bp += THREADS)       /* header bytes of the seam word */
            g_a0[bp] = (uint8_t)header_byte(bp - rec_lo, len, tree_len, tb);
        const uint64_t z0 = hdr_end, z1 = rec_hi;                /* zero bytes [z0, z1) relative to A0 */
        const uint64_t a0 = dmin<uint64_t>((z0 + 15) & ~15ull, z1);
        const uint64_t a1 = dmax<uint64_t>(a0, z1 & ~15ull);
        /* g_a0 is 4-byte aligned; 16-byte stores need the absolute address aligned */
        const uint64_t skew = (uint64_t)((uintptr_t)g_a0 & 15u);
        const uint64_t b0 = dmin<uint64_t>(((z0 + skew + 15) & ~15ull) - skew, z1);
        const uint64_t b1 = dmax<uint64_t>(b0, ((z1 + skew) & ~15ull) - skew);
        (void)a0; (void)a1;
        for (uint64_t bp = z0 + tid; bp < b0; bp += THREADS) g_a0[bp] = 0;
        uint4 *q = reinterpret_cast<uint4 *>(g_a0 + b0);
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t i = (uint64_t)tid; i < ((b1 - b0) >> 4); i += THREADS) store_pack16(q + i, zero4);
        for (uint64_t bp = b1 + tid; bp < z1; bp += THREADS) g_a0[bp] = 0;
        return false;
    }
    if (tid == 0) {
        uint32_t t = 0;                                          /* big-endian partial word */
        for (uint32_t bp = hdr_end & ~3u; bp < hdr_end; bp++)
            t = (t << 8) | header_byte(bp - rec_lo, len, tree_len, tb);
        s_tail[WAVES] = t;                                       /* carry: value of the (hdr_end&3)*8 leading bits */
    }
    __syncthreads();
    *hdr_end_out = hdr_end;
    return true;
}

template <int THREADS, typename CodeT, int GROUP = 1>
__device__ __forceinline__ void pack_block(const uint8_t *__restrict__ src, uint64_t len,
                                           const hufcode_t *__restrict__ codes64,
                                           const int16_t *__restrict__ tb, uint32_t tree_len,
                                           uint8_t *__restrict__ out, uint64_t dst0, uint64_t dst1,
                                           CodeT *s_code, uint32_t *s_part, uint32_t *s_tail, uint32_t *s_stage,
                                           uint64_t *__restrict__ sub_tiles, uint16_t *__restrict__ sub_groups)
{
    static_assert(HUF_SUB_TILE == 64 * PACK_SPT, "a sub-index tile is one wave of a pack tile");
    constexpr int TILE = THREADS * PACK_SPT;
    constexpr int WAVES = THREADS / 64;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    uint8_t *g_a0 = out + (dst0 & ~3ull);
    const uint32_t rec_lo = (uint32_t)(dst0 & 3ull);            /* record bytes relative to A0 */
    const uint64_t rec_hi = rec_lo + (dst1 - dst0);
    uint32_t *g_w0 = reinterpret_cast<uint32_t *>(g_a0);

    for (int i = tid; i < HUF_NSYM; i += THREADS) s_code[i] = (CodeT)codes64[i];

    uint32_t hdr_end;
    if (!pack_header<THREADS>(len, tb, tree_len, g_a0, g_w0, rec_lo, rec_hi, s_tail, &hdr_end)) return;

    uint64_t bitpos = (uint64_t)hdr_end * 8ull;                  /* relative to A0 bit 0 */


    for (uint64_t t0 = 0; t0 < len; t0 += TILE) {
        /* ---- load + look up ---- */
        const uint64_t my0 = t0 + (uint64_t)tid * PACK_SPT;
        uint32_t nsym = 0;
        CodeT code[PACK_SPT];
        uint32_t mybits = 0;
        if (my0 < len) {
            nsym = (uint32_t)dmin<uint64_t>(PACK_SPT, len - my0);
            const uint8_t *p = src + my0;
            if (nsym == PACK_SPT && (((uintptr_t)p) & 15u) == 0) {
                const uint4 v0 = load_stream16(reinterpret_cast<const uint4 *>(p));
                const uint4 v1 = load_stream16(reinterpret_cast<const uint4 *>(p) + 1);
                const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int k = 0; k < PACK_SPT; k++) code[k] = s_code[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
            } else {
#pragma unroll
                for (int k = 0; k < PACK_SPT; k++) code[k] = (k < (int)nsym) ? s_code[p[k]] : (CodeT)0;
            }
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) mybits += (uint32_t)(code[k] & 0xffu);
        } else {
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) code[k] = 0;
        }
        uint32_t tile_bits;
        const uint32_t ex = block_excl_scan_u32<THREADS>(mybits, s_part, tile_bits);
        if (sub_groups) {                                        /* the sub-index: 2 bytes per 32 symbols */
            if (nsym) sub_groups[my0 / PACK_SPT] = (uint16_t)mybits;
            if (lane == 0 && nsym) sub_tiles[my0 / HUF_SUB_TILE] = bitpos + ex - (uint64_t)hdr_end * 8ull;   /* (my0 = t0 + wave * 2 048) */
        }

        /* ---- shift the codes out ----
         * Finished words go to an LDS stage laid out like the destination (stage word i <-> HBM
         * address stage_addr + 4 i, both 16-byte aligned), then the workgroup flushes the stage
         * with 16-byte stores: lanes' words are adjacent in memory but not in time, so storing
         * them one by one costs a partially filled store instruction per word.  A tile whose
         * codes are too long for the stage (possible only far above the 9-bit average) stores
         * straight to HBM instead. */
        const uint64_t s = bitpos + ex;                          /* my first bit */
        const uint64_t w_lo = bitpos >> 5, w_hi = (bitpos + tile_bits) >> 5;   /* tile's finished words [w_lo, w_hi) */
        const uintptr_t stage_addr = (uintptr_t)(g_w0 + w_lo) & ~(uintptr_t)15;
        const uint32_t i_lo = (uint32_t)(((uintptr_t)(g_w0 + w_lo) - stage_addr) >> 2);
        const uint32_t i_hi = i_lo + (uint32_t)(w_hi - w_lo);
        const bool staged = i_hi + 2 <= PACK_STAGE_WORDS;        /* wave-uniform */
        PackAcc<CodeT> a;
        a.acc = 0;
        a.nacc = (uint32_t)(s & 31u);                            /* leading bits come from the left */
        a.have_first = false;
        a.first = 0;
        uint32_t *const g_first = g_w0 + (s >> 5);
        uint32_t *const s_first = s_stage + (i_lo + (uint32_t)((s >> 5) - w_lo));
        if (staged) {
            a.gw = s_first;
            if constexpr (GROUP == 3) {
#pragma unroll
                for (int k = 0; k + 2 < PACK_SPT; k += 3) a.push_triple((uint32_t)code[k], (uint32_t)code[k + 1], (uint32_t)code[k + 2]);
                static_assert(PACK_SPT % 3 == 2, "the last two symbols go as a pair");
                a.push_pair((uint32_t)code[PACK_SPT - 2], (uint32_t)code[PACK_SPT - 1]);
            } else if constexpr (GROUP == 2) {
#pragma unroll
                for (int k = 0; k < PACK_SPT; k += 2) a.push_pair((uint32_t)code[k], (uint32_t)code[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < PACK_SPT; k++) a.push(code[k]);  /* absent symbols have len 0 */
            }
        } else {
            a.gw = g_first;
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) a.push(code[k]);
        }
        const uint32_t nwords = (uint32_t)(a.gw - (staged ? s_first : g_first));   /* finished words of this lane */
        const uint32_t tail_val = a.tail();

        /* ---- tails hop one lane to the right ---- */
        uint32_t in_tail = wave_up1_u32(tail_val);
        if (lane == 63) s_tail[wave] = tail_val;
        __syncthreads();
        if (lane == 0) in_tail = (wave == 0) ? s_tail[WAVES] : s_tail[wave - 1];
        const uint32_t n_in = (uint32_t)(s & 31u);
        const bool is_last = (nsym > 0) && (my0 + nsym == len);  /* holds the block's last symbol */
        if (a.have_first) {
            const uint32_t word = __builtin_bswap32(a.first | (n_in ? (in_tail << (32 - n_in)) : 0u));
            if (staged) *s_first = word;
            else *g_first = word;
        }
        uint32_t out_tail = tail_val;
        if (!a.have_first && nsym > 0) {
            /* only the block's last lane can be shorter than a word: its tail continues the
             * neighbour's */
            out_tail = (n_in ? (in_tail << (a.nacc - n_in)) : 0u) | tail_val;
        }
        if (is_last && a.nacc) {
            /* zero-padded final byte(s) (encoder.c:123-128); bytes past the record belong to
             * the next block */
            const uint32_t word = out_tail << (32 - a.nacc);
            const uint32_t nbytes = (a.nacc + 7) >> 3;
            uint8_t *b = reinterpret_cast<uint8_t *>(g_first + nwords);
            for (uint32_t k = 0; k < nbytes; k++) b[k] = (uint8_t)(word >> (24 - 8 * k));
        }
        __syncthreads();                                         /* stage complete; s_tail is rewritten next tile */
        if (tid == THREADS - 1) s_tail[WAVES] = out_tail;        /* carry into the next tile */
        if (staged) {
            uint8_t *const g16 = reinterpret_cast<uint8_t *>(stage_addr);   /* (a flat store; as a global one - the pointer
                                                                               derived from `out` - it is exactly as fast) */
            for (uint32_t u = tid; 4 * u < i_hi; u += THREADS) {
                const uint32_t i0 = 4 * u;
                if (i0 >= i_lo && i0 + 4 <= i_hi) {
                    store_pack16(reinterpret_cast<uint4 *>(g16 + 4 * i0), *reinterpret_cast<const uint4 *>(s_stage + i0));
                } else {
                    for (uint32_t i = (i0 > i_lo ? i0 : i_lo); i < i0 + 4 && i < i_hi; i++)
                        *reinterpret_cast<uint32_t *>(g16 + 4 * i) = s_stage[i];
                }
            }
        }
        bitpos += tile_bits;
        (void)rec_hi;
    }
}

/* ======================================================================================
 * Blocks whose longest code has at most 15 bits (GROUP = 2: two codes a push) or 10 bits (GROUP = 3), i.e. text,
 * Zipf bytes, uniform bytes: round 5's placement loop.  pack_kernel spends its time issuing vector instructions
 * (15.4 a symbol in round 4, 86 % of the kernel's cycles), so the loop is built around what a push costs:
 *
 *   - the table entry is EIGHT bytes (one ds_read_b64: 64 banks): x = the code, right-aligned and clean, y = the
 *     length three times - bits 0..4 as it is, bits 16..20 as (32 - len) & 31, bits 27..31 as it is.  The sum
 *     of the y of a group's entries (one v_add_u32 / v_add3_u32, nothing to mask: the fields cannot carry into
 *     each other) is then, as it stands: the group's bit count (low half: the lane's 32 symbols summed,
 *     v_add3_u32 over the groups), the shift that makes room in the accumulator (v_lshl_or_b32 looks at bits
 *     0..4), the shift that recovers the bits pushed out of it (byte 2 through SDWA: -count mod 32), and -
 *     added to a running total kept in the same three fields - the carry out of bit 31 says "a 32-bit word is
 *     complete" (v_add_co_u32: sum and test in one instruction) while bits 0..4 of the total are the amount
 *     v_alignbit_b32 cuts the finished word out by.  A push of two codes is 4 instructions and 4 more when
 *     some lane finishes a word; round 4's was 8 + 8.
 *   - every finished word goes to the stage, the lane's first included (round 4 kept it in a register and
 *     tested "is this my first" at every push): its leading bits are zero - the accumulator starts empty - and
 *     the left neighbour's unfinished tail is OR-ed into them behind the barrier (ds_or_b32).
 * A tile whose words do not fit the stage (far above 12 bits a symbol) is placed symbol by symbol, straight to HBM.
 * ==================================================================================== */
struct PackEnt {
    uint32_t code;      /* right-aligned, at most 15 bits */
    uint32_t lens;      /* len | ((32 - len) & 31) << 16 | len << 27; 0 for an absent byte value */
};
__device__ __forceinline__ uint2 pack_ent_of(hufcode_t e)
{
    const uint32_t len = (uint32_t)(e & 0xffu);
    return make_uint2((uint32_t)(e >> 8), len | (((32u - len) & 31u) << 16) | (len << 27));
}
/* bits the push of `lens` shifts out of `acc`: acc >> (-count mod 32), the amount read from byte 2 of `lens` */
__device__ __forceinline__ uint32_t pack_pushed_out(uint32_t lens, uint32_t acc)
{
    uint32_t o;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD"
        : "=v"(o) : "v"(lens), "v"(acc));
    return o;
}

/* The lane's 32 bytes (w: eight dwords) looked up and joined group by group: pc = the group's codes as one bit
 * string, ph = the sum of their length fields; returns the lane's bit count.  PARTIAL: bytes from nsym on are absent.
 * (A batch of look-ups, then their joins: all 32 entries in flight at once would be 64 registers.) */
template <int GROUP, bool PARTIAL>
__device__ __forceinline__ uint32_t pack_join(const uint2 *s_ent, const uint32_t (&w)[8], uint32_t nsym,
                                              uint32_t (&pc)[(PACK_SPT + GROUP - 1) / GROUP], uint32_t (&ph)[(PACK_SPT + GROUP - 1) / GROUP])
{
    constexpr int NG = (PACK_SPT + GROUP - 1) / GROUP;
    constexpr int BATCH = GROUP == 2 ? 4 : 2;                    /* groups a batch */
    uint32_t bits = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g % BATCH == 0) __builtin_amdgcn_sched_barrier(0);
        const int k = g * GROUP;
        uint2 e[3];
#pragma unroll
        for (int j = 0; j < GROUP; j++) {
            const int kk = k + j;
            if (kk < PACK_SPT) {
                e[j] = s_ent[(w[kk >> 2] >> (8 * (kk & 3))) & 0xffu];
                if (PARTIAL && kk >= (int)nsym) e[j] = make_uint2(0u, 0u);
            }
        }
        if (GROUP == 3 && k + 2 < PACK_SPT) {
            pc[g] = (((e[0].x << (e[1].y & 31u)) | e[1].x) << (e[2].y & 31u)) | e[2].x;
            ph[g] = e[0].y + e[1].y + e[2].y;
        } else {
            pc[g] = (e[0].x << (e[1].y & 31u)) | e[1].x;
            ph[g] = e[0].y + e[1].y;
        }
        bits += ph[g];
    }
    __builtin_amdgcn_sched_barrier(0);
    return bits & 0xffffu;
}

template <int THREADS, int GROUP>
__device__ __forceinline__ void pack_block_multi(const uint8_t *__restrict__ src, uint64_t len64,
                                                 const hufcode_t *__restrict__ codes64,
                                                 const int16_t *__restrict__ tb, uint32_t tree_len,
                                                 uint8_t *__restrict__ out, uint64_t dst0, uint64_t dst1,
                                                 uint2 *s_ent, uint32_t *s_part, uint32_t *s_tail, uint32_t *s_stage,
                                                 uint64_t *__restrict__ sub_tiles, uint16_t *__restrict__ sub_groups)
{
    static_assert(GROUP == 2 || GROUP == 3, "two or three codes a push");
    static_assert(HUF_SUB_TILE == 64 * PACK_SPT, "a sub-index tile is one wave of a pack tile");
    constexpr uint32_t TILE = THREADS * PACK_SPT;
    constexpr int WAVES = THREADS / 64;
    const uint32_t len = (uint32_t)len64;                        /* (blocks of this kernel are shorter than 2 MiB: positions inside one are 32-bit, and a
                                                                    lane's place is ONE register around the tile loop, not an address pair) */
    constexpr int NG = (PACK_SPT + GROUP - 1) / GROUP;           /* pushes a lane: 16 pairs, or 10 triples and a pair */
    typedef __attribute__((address_space(3))) uint32_t *lds_word;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    uint8_t *g_a0 = out + (dst0 & ~3ull);
    const uint32_t rec_lo = (uint32_t)(dst0 & 3ull);            /* record bytes relative to A0 */
    const uint64_t rec_hi = rec_lo + (dst1 - dst0);
    uint32_t *g_w0 = reinterpret_cast<uint32_t *>(g_a0);

    for (int i = tid; i < HUF_NSYM; i += THREADS) s_ent[i] = pack_ent_of(codes64[i]);

    uint32_t hdr_end;
    if (!pack_header<THREADS>(len64, tb, tree_len, g_a0, g_w0, rec_lo, rec_hi, s_tail, &hdr_end)) return;

    uint64_t bitpos = (uint64_t)hdr_end * 8ull;                  /* relative to A0 bit 0 */

    /* A tile's input is requested ONE TILE AHEAD: right behind the look-ups of the tile before it, when the eight registers
     * that held that tile's bytes are free again - the loads then travel while the tile before is summed, placed and
     * flushed.  (Round 4 measured this at -1.5 %: the kernel was bound by instruction issue then; with half of round 4's
     * instructions a workgroup's life is mostly the latency of its loads - time = 0.26 + 1.18 / workgroups per CU ms.)
     * `ahead`: the wave's lanes of the next tile are all whole and aligned (wave-uniform). */
    uint4 n0 = make_uint4(0u, 0u, 0u, 0u), n1 = n0;
    bool ahead = false;
#define PACK_REQUEST(T0_)                                                                                      \
    {                                                                                                         \
        const uint32_t m0_ = (T0_) + (uint32_t)tid * PACK_SPT;                                                \
        const uint8_t *q_ = src + m0_;                                                                        \
        ahead = (T0_) < len && __builtin_amdgcn_ballot_w64(!(m0_ + PACK_SPT <= len && (((uintptr_t)q_) & 15u) == 0)) == 0ull; \
        if (ahead) {                                                                                          \
            n0 = load_stream16(reinterpret_cast<const uint4 *>(q_));                                          \
            n1 = load_stream16(reinterpret_cast<const uint4 *>(q_) + 1);                                      \
        }                                                                                                     \
    }
    PACK_REQUEST(0u)
    for (uint32_t t0 = 0; t0 < len; t0 += TILE) {
        /* ---- look up: the codes of a group are joined as they arrive ---- */
        const uint32_t my0 = t0 + (uint32_t)tid * PACK_SPT;
        const uint32_t nsym = my0 < len ? dmin<uint32_t>(PACK_SPT, len - my0) : 0u;
        const uint8_t *p = src + my0;
        uint32_t pc[NG], ph[NG];                                 /* a group's joined codes / summed length fields */
        uint32_t mybits;
        /* (one branch for the wave: the tile of whole, aligned lanes - every tile of a 64 KiB block - has no
         *  per-symbol "is it there" in it) */
        if (ahead) {
            const uint32_t w[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
            mybits = pack_join<GROUP, false>(s_ent, w, nsym, pc, ph);
        } else {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                w[i] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * i + j < (int)nsym) w[i] |= (uint32_t)p[4 * i + j] << (8 * j);
            }
            mybits = pack_join<GROUP, true>(s_ent, w, nsym, pc, ph);
        }
        PACK_REQUEST(t0 + TILE)
        uint32_t tile_bits;
        const uint32_t ex = block_excl_scan_u32<THREADS>(mybits, s_part, tile_bits);
        if (sub_groups) {                                        /* the sub-index: 2 bytes per 32 symbols */
            if (nsym) sub_groups[my0 / PACK_SPT] = (uint16_t)mybits;
            if (lane == 0 && nsym) sub_tiles[my0 / HUF_SUB_TILE] = bitpos + ex - (uint64_t)hdr_end * 8ull;   /* (my0 = t0 + wave * 2 048) */
        }

        /* ---- shift the codes out (stage, flush: as pack_block).  A lane's place relative to the tile's first word in
         *      32-bit arithmetic: the tile's first bit is the same for every lane ---- */
        const uint64_t w_lo = bitpos >> 5, w_hi = (bitpos + tile_bits) >> 5;   /* tile's finished words [w_lo, w_hi) */
        const uintptr_t stage_addr = (uintptr_t)(g_w0 + w_lo) & ~(uintptr_t)15;
        const uint32_t i_lo = (uint32_t)(((uintptr_t)(g_w0 + w_lo) - stage_addr) >> 2);
        const uint32_t i_hi = i_lo + (uint32_t)(w_hi - w_lo);
        const bool staged = i_hi + 2 <= PACK_STAGE_WORDS;        /* workgroup-uniform */
        const uint32_t rel = (uint32_t)(bitpos & 31u) + ex;      /* my first bit, from the first bit of word w_lo */
        const uint32_t n_in = rel & 31u;                         /* leading bits come from the left */
        uint32_t *const g_first = g_w0 + w_lo + (rel >> 5);
        uint32_t *const s_first = s_stage + (i_lo + (rel >> 5));
        uint32_t nwords, nacc, tail_val, first = 0;
        if (staged) {
            uint32_t acc = 0;
            uint32_t tot = n_in | (n_in << 27);                  /* bits so far: bits 0..4 and 27..31 count them mod 32 */
            const uint32_t gw0 = (uint32_t)(uintptr_t)(lds_word)s_first - 4u;
            uint32_t gw = gw0;                                   /* LDS byte address of the last finished word (moved, then stored
                                                                    through: one v_add in place, no copy of the address) */
#pragma unroll
            for (int g = 0; g < NG; g++) {
                uint32_t tot2;
                const bool word_done = __builtin_add_overflow(tot, ph[g], &tot2);
                const uint32_t acc2 = (acc << (ph[g] & 31u)) | pc[g];
                if (word_done) {
                    const uint32_t word = __builtin_amdgcn_alignbit(pack_pushed_out(ph[g], acc), acc2, tot2);
                    gw += 4;
                    asm volatile("" : "+v"(gw));
                    *(lds_word)(uintptr_t)gw = __builtin_bswap32(word);
                }
                acc = acc2;
                tot = tot2;
            }
            nwords = (gw - gw0) >> 2;
            nacc = tot & 31u;
            tail_val = acc & ((1u << nacc) - 1u);
        } else {
            PackAcc<uint32_t> a;
            a.acc = 0;
            a.nacc = n_in;
            a.have_first = false;
            a.first = 0;
            a.gw = g_first;
#pragma unroll 1
            for (uint32_t k = 0; k < nsym; k++) {
                const uint2 e = s_ent[src[my0 + k]];
                a.push32(e.x, e.y & 31u);
            }
            nwords = (uint32_t)(a.gw - g_first);
            nacc = a.nacc;
            tail_val = a.tail();
            first = a.first;
        }

        /* ---- tails hop one lane to the right ---- */
        uint32_t in_tail = wave_up1_u32(tail_val);
        if (lane == 63) s_tail[wave] = tail_val;
        __syncthreads();
        if (lane == 0) in_tail = (wave == 0) ? s_tail[WAVES] : s_tail[wave - 1];
        const bool is_last = (nsym > 0) && (my0 + nsym == len);  /* holds the block's last symbol */
        if (nwords) {
            const uint32_t lead = n_in ? (in_tail << (32 - n_in)) : 0u;
            if (staged) {
                if (n_in) __hip_atomic_fetch_or(s_first, __builtin_bswap32(lead), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                *g_first = __builtin_bswap32(first | lead);
            }
        }
        uint32_t out_tail = tail_val;
        if (!nwords && nsym > 0) {
            /* only the block's last lane can be shorter than a word: its tail continues the
             * neighbour's */
            out_tail = (n_in ? (in_tail << (nacc - n_in)) : 0u) | tail_val;
        }
        if (is_last && nacc) {
            /* zero-padded final byte(s) (encoder.c:123-128); bytes past the record belong to
             * the next block */
            const uint32_t word = out_tail << (32 - nacc);
            const uint32_t nbytes = (nacc + 7) >> 3;
            uint8_t *b = reinterpret_cast<uint8_t *>(g_first + nwords);
            for (uint32_t k = 0; k < nbytes; k++) b[k] = (uint8_t)(word >> (24 - 8 * k));
        }
        __syncthreads();                                         /* stage complete; s_tail is rewritten next tile */
        if (tid == THREADS - 1) s_tail[WAVES] = out_tail;        /* carry into the next tile */
        if (staged) {
            uint8_t *const g16 = reinterpret_cast<uint8_t *>(stage_addr);
            for (uint32_t u = tid; 4 * u < i_hi; u += THREADS) {
                const uint32_t i0 = 4 * u;
                if (i0 >= i_lo && i0 + 4 <= i_hi) {
                    store_pack16(reinterpret_cast<uint4 *>(g16 + 4 * i0), *reinterpret_cast<const uint4 *>(s_stage + i0));
                } else {
                    for (uint32_t i = (i0 > i_lo ? i0 : i_lo); i < i0 + 4 && i < i_hi; i++)
                        *reinterpret_cast<uint32_t *>(g16 + 4 * i) = s_stage[i];
                }
            }
        }
        bitpos += tile_bits;
        (void)rec_hi;
    }
}

#undef PACK_REQUEST

/* SHORT = true: the host guarantees that no code of this launch is longer than 24 bits (any
 * Huffman merge order on n <= 121392 symbols gives depth <= 23, plus the wrap-root bit; the
 * deepest tree needs Fibonacci weights), so only the 32-bit code path is compiled - fewer
 * registers, more waves. */
#ifndef PACK_WAVES_PER_SIMD
#define PACK_WAVES_PER_SIMD 6          /* Round 2's build with 7 (72 of 72 VGPRs) kept code[5] of the one-code-per-push form in v71, the
                                          LAST register of the allocation, and on gfx950 a 64-bit shift whose shift amount is the last
                                          allocated VGPR shifts by something else (there: by the lane number) in every wave that is not
                                          the first on its SIMD - wrong payload bits in blocks 256 and up.  Root cause, reproducer and the
                                          build-time check: DESIGN.md 3.3, tools/calib/last_vgpr_probe.hip, libhuffman_amd/isa_check.py
                                          (the build FAILS if any kernel shows the pattern).  Since round 3 the short-code accumulator shifts
                                          32 bits at a time (PackAcc<uint32_t>: 69 VGPRs, seven waves would fit and pass the check) - and
                                          seven waves are not faster: pack waits for the memory system, not for a wave slot. */
#endif
template <int THREADS, bool SHORT>
__global__ __launch_bounds__(THREADS, SHORT ? PACK_WAVES_PER_SIMD : 4) void pack_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                       uint64_t blocksize,
                                                       const hufcode_t *__restrict__ codetab,
                                                       const int16_t *__restrict__ treebuf,
                                                       const HufBlockMeta *__restrict__ meta,
                                                       uint64_t *__restrict__ offsets, TwoLevel sizes,
                                                       uint8_t *__restrict__ out, HufSubIndex sub)
{
    __shared__ hufcode_t s_code[HUF_NSYM];   /* eight-byte entries on the paths of two and three codes a push (PackEnt), u32[256] for codes up to 24 bits */
    __shared__ uint32_t s_part[THREADS / 64];
    __shared__ uint32_t s_tail[THREADS / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[PACK_STAGE_WORDS];

#ifdef PACK_VGPR_SLACK      /* test builds only: "v72" gives the 72-VGPR (7 waves per SIMD) build its register of slack */
    asm volatile("; one VGPR more than the kernel uses" ::: PACK_VGPR_SLACK);
#endif
    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    const HufBlockMeta m = meta[blk];
    const hufcode_t *codes = codetab + blk * HUF_NSYM;
    const int16_t *tb = treebuf + blk * HUF_TREE_STRIDE;
    uint64_t o0, o1;
    if (sizes.local) {                   /* sizes were summed by hist_tree_kernel: publish the index entry */
        o0 = sizes.gprefix[blk / SCAN_GROUP] + sizes.local[blk];
        o1 = o0 + encoded_block_bytes(m);
        if (threadIdx.x == 0) offsets[blk] = o0;
    } else {
        o0 = offsets[blk];
        o1 = offsets[blk + 1];
    }
    uint64_t *sub_tiles = sub.tile_bits ? sub.tile_bits + blk * sub.tpb : nullptr;
    uint16_t *sub_groups = sub.tile_bits ? sub.group_bits + blk * sub.gpb : nullptr;
    if (sub.tile_bits && m.tree_len != 5)
        for (int i = (int)threadIdx.x; i < HUF_NSYM; i += THREADS) sub.lens[blk * HUF_NSYM + i] = (uint8_t)(codes[i] & 0xffu);
#if !defined(PACK_ACC64)   /* (-DPACK_ACC64 = round 2's kernel, tests/test_isa_check.py) */
    if (m.max_len <= 10)                 /* three codes per push */
        pack_block_multi<THREADS, 3>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                     reinterpret_cast<uint2 *>(s_code), s_part, s_tail, s_stage, sub_tiles, sub_groups);
    else if (m.max_len <= 15)            /* two codes per push (at most 30 bits) */
        pack_block_multi<THREADS, 2>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                     reinterpret_cast<uint2 *>(s_code), s_part, s_tail, s_stage, sub_tiles, sub_groups);
#else
    if (m.max_len <= 10)
        pack_block<THREADS, uint32_t, 3>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                         reinterpret_cast<uint32_t *>(s_code), s_part, s_tail, s_stage, sub_tiles, sub_groups);
    else if (m.max_len <= 15)
        pack_block<THREADS, uint32_t, 2>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                            reinterpret_cast<uint32_t *>(s_code), s_part, s_tail, s_stage, sub_tiles, sub_groups);
#endif
    else if (SHORT || m.max_len <= 24)
        pack_block<THREADS, uint32_t>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                      reinterpret_cast<uint32_t *>(s_code), s_part, s_tail, s_stage, sub_tiles, sub_groups);
    else if constexpr (!SHORT)
        pack_block<THREADS, hufcode_t>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                       s_code, s_part, s_tail, s_stage, sub_tiles, sub_groups);
}

}  // namespace hufgpu
