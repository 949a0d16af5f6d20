/* pack_chunk.hpp - pack_chunk_kernel: header and bit-pack of blocks that are cut into chunks
   (src/encoder.c:85-131, 322-339 for blocks of tens of MiB up to the whole input, blocksize = 0).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../hufgpu_common.h"
#include "offsets.hpp"
#include "pack.hpp"

namespace hufgpu {

/* ======================================================================================
 * A block is the unit of parallelism of pack_kernel (pack.hpp): one workgroup per block.  With
 * blocksize = 0 (src/encoder.c:163-165: one block of the whole input) or blocks of many MiB that
 * leaves the chip to a handful of workgroups.  Here a block is cut into chunks of PACK_CHUNK_SYMS
 * symbols, one workgroup each: chunk_hist_kernel (hist_chunk.hpp) counts per chunk, the tree is
 * built from the block's totals, chunk_bits_kernel turns the chunk histograms and the code
 * lengths into the payload bit every chunk starts at, and this kernel packs the chunks where
 * they belong.  Chunks meet inside bytes: every byte has exactly one writer (see pack_segment).
 * ==================================================================================== */
/* ---- the stage: an LDS image of the output, accumulated with ds_or --------------------------------
 * Stream bit b (counted from a 16-byte aligned address, MSB first inside each byte) is bit
 * 31 - (b & 31) of stage word b >> 5 ("big-endian words": a finished word is byte-swapped when it
 * is stored).  A lane shifts its codes through a 64-bit accumulator that starts with as many zero
 * bits as lie in front of its first bit inside that bit's word, and ORs every 32 bits that are
 * complete into the zeroed stage (ds_or_b32), the unfinished rest at the end: words shared by
 * neighbouring lanes simply receive both parts.  No first-word special case, no tail hand-over
 * between lanes or waves, no seam logic between tiles (the word a tile leaves unfinished stays in
 * the stage).  Two forms were measured against this one: exact stores with tail hand-over (round 1:
 * 18 wave instructions per symbol, 0.56 ms per GiB) and one pair of ds_or per code pair with no
 * accumulator at all (13 per symbol, but 2 LDS atomics per pair: LDS bound, 0.60 ms). */
#define PACK_STAGE_WORDS2 4224                /* 16.5 KiB: a 256 x 32-symbol tile of 16-bit codes + slack */
#define PACK_STAGE_CAP_BITS ((PACK_STAGE_WORDS2 - 12) * 32)

struct StageAcc {
    uint64_t acc;        /* the low nacc bits are not in the stage yet */
    uint32_t nacc;       /* < 32 between pushes */
    uint32_t *wp;        /* stage word the next 32 complete bits go to */

    __device__ __forceinline__ void start(uint32_t *stage, uint32_t q)
    {
        acc = 0;
        nacc = q & 31u;
        wp = stage + (q >> 5);
    }
    __device__ __forceinline__ void push(uint32_t code, uint32_t len)       /* len <= 32 */
    {
        acc = (acc << len) | code;
        nacc += len;
        if (nacc >= 32u) {
            nacc -= 32u;
            atomicOr(wp, (uint32_t)(acc >> nacc));
            wp++;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc) atomicOr(wp, (uint32_t)acc << (32u - nacc));
    }
};

struct PackChunk {
    const uint64_t *chunk_bits;   /* NULL: a block is one chunk.  Else [chunk]: first payload bit of every chunk */
    uint64_t chunk_syms;          /* symbols per chunk (a multiple of the tile) */
    uint32_t cpb;                 /* chunks per block */
};

/* One chunk of a block: header (chunk 0), then the chunk's symbols from stream bit P on.
 *   src / len      the chunk's symbols
 *   rec0           stream byte where the block's record starts (chunk 0 writes the header there)
 *   P              stream bit of the chunk's first payload bit (chunk 0: right behind the header)
 *   last           the chunk holds the block's last symbol (the final byte is zero padded)
 * Bytes are owned exclusively: a chunk writes every byte from the one that holds bit P (its first
 * P % 8 bits are the previous chunk's last code bits: recomputed here from the symbols in front of
 * src) up to, and for a chunk that is not the block's last excluding, the byte that holds the
 * next chunk's first bit.  MODE 0: codes <= 16 bits, two symbols per placement; 1: <= 24 bits;
 * 2: any length (64-bit table entries). */
template <int THREADS, int MODE>
__device__ __forceinline__ void pack_segment(const uint8_t *__restrict__ src, uint64_t len, uint64_t block_len,
                                             const hufcode_t *__restrict__ codes64,
                                             const int16_t *__restrict__ tb, uint32_t tree_len, bool first,
                                             bool last, uint8_t *__restrict__ out, uint64_t rec0, uint64_t P,
                                             uint32_t *s_code32, uint32_t *s_part, uint32_t *s_stage,
                                             uint64_t *__restrict__ sub_tiles, uint16_t *__restrict__ sub_groups,
                                             uint64_t pay_rel0)
{
    constexpr int TILE = THREADS * PACK_SPT;
    constexpr int PARTS = MODE == 0 ? 1 : (MODE == 1 ? 2 : 4);
    static_assert((THREADS / PARTS) * PACK_SPT * (MODE == 0 ? 16 : (MODE == 1 ? 24 : HUF_CODE_MAXBITS)) + 64 <= PACK_STAGE_CAP_BITS,
                  "a part's bits fit the stage");
    typedef typename std::conditional<MODE == 2, hufcode_t, uint32_t>::type EntT;
    EntT *s_code = reinterpret_cast<EntT *>(s_code32);
    const int tid = (int)threadIdx.x;

    if (tree_len != 5) {                  /* (one-symbol blocks need neither table nor stage) */
        for (int i = tid; i < HUF_NSYM; i += THREADS) s_code[i] = (EntT)codes64[i];
        for (int i = tid; i < PACK_STAGE_WORDS2 / 4; i += THREADS) reinterpret_cast<uint4 *>(s_stage)[i] = make_uint4(0u, 0u, 0u, 0u);
    }

    /* ---- header (chunk 0): whole aligned words are stored here, the unfinished last word goes into
     *      the stage in front of the payload ---- */
    const uint32_t hdr_bytes = HUF_HEADER_FIXED + 2u * tree_len;
    uint8_t *g_a0 = out + (rec0 & ~3ull);
    const uint32_t rec_lo = (uint32_t)(rec0 & 3ull);             /* record bytes relative to A0 */
    const uint32_t hdr_end = rec_lo + hdr_bytes;                 /* relative to A0 */
    if (first) {
        uint32_t *g_w0 = reinterpret_cast<uint32_t *>(g_a0);
        for (uint32_t w = tid; w < (hdr_end >> 2); w += THREADS) {
            uint32_t v = 0;                                      /* little-endian memory word */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t bp = 4 * w + k;
                if (bp >= rec_lo) v |= header_byte(bp - rec_lo, block_len, tree_len, tb) << (8 * k);
            }
            if (4 * w >= rec_lo) g_w0[w] = v;
            else {
                for (uint32_t k = rec_lo - 4 * w; k < 4; k++) g_a0[4 * w + k] = (uint8_t)(v >> (8 * k));
            }
        }
    }
    /* byte addresses: Pb = the byte that holds bit P; base = stage word 0 (16-byte aligned) */
    uint8_t *const Pb = out + (P >> 3);
    uintptr_t base = (uintptr_t)Pb & ~(uintptr_t)15;
    uint32_t bitpos = (uint32_t)((uintptr_t)Pb - base) * 8u + (uint32_t)(P & 7u);   /* next bit to write, from base */
    uintptr_t own_lo;                                            /* first byte this chunk still has to store */

    if (tree_len == 5) {
        /* One distinct byte in the block: its code is the single bit 0 (tree.c:410-413 with one
         * leaf), so the payload is zero bytes - nothing of the input needs reading again (the
         * histogram already saw it).  P is a byte boundary here (chunks are multiples of 8 symbols). */
        if (first) {
            for (uint32_t bp = (hdr_end & ~3u) + tid; bp < hdr_end; bp += THREADS)   /* header bytes of the seam word */
                g_a0[bp] = (uint8_t)header_byte(bp - rec_lo, block_len, tree_len, tb);
        }
        uint8_t *z0 = Pb, *z1 = Pb + (last ? (len + 7) / 8 : len / 8);
        uint8_t *b0 = (uint8_t *)dmin<uintptr_t>(((uintptr_t)z0 + 15) & ~(uintptr_t)15, (uintptr_t)z1);
        uint8_t *b1 = (uint8_t *)dmax<uintptr_t>((uintptr_t)b0, (uintptr_t)z1 & ~(uintptr_t)15);
        for (uint8_t *q = z0 + tid; q < b0; q += THREADS) *q = 0;
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        for (uint4 *q = reinterpret_cast<uint4 *>(b0) + tid; q < reinterpret_cast<uint4 *>(b1); q += THREADS) store_pack16(q, zero4);
        for (uint8_t *q = b1 + tid; q < z1; q += THREADS) *q = 0;
        return;
    }
    __syncthreads();                                             /* table and zeroed stage are there */
    if (first) {
        /* the header's last, unfinished word: its bytes go in front of the payload */
        own_lo = (uintptr_t)Pb & ~(uintptr_t)3;
        if (tid == 0) {
            for (uintptr_t a = own_lo; a < (uintptr_t)Pb; a++) {
                const uint32_t rel = (uint32_t)(a - base);
                s_stage[rel >> 2] |= header_byte((uint32_t)(a - (uintptr_t)g_a0) - rec_lo, block_len, tree_len, tb) << (24 - 8 * (rel & 3u));
            }
        }
    } else {
        /* the P % 8 bits in front of P inside the byte this chunk owns: the end of the previous
         * chunk's last codes, recomputed from the symbols in front of src */
        own_lo = (uintptr_t)Pb;
        const uint32_t lead = (uint32_t)(P & 7u);
        if (tid == 0 && lead) {
            uint64_t bits = 0;
            uint32_t have = 0;
            for (int k = 1; have < lead; k++) {
                const uint64_t e = (uint64_t)s_code[src[-k]];
                bits |= (e >> 8) << have;
                have += (uint32_t)(e & 0xffu);
            }
            const uint32_t v = (uint32_t)bits & ((1u << lead) - 1u);
            const uint32_t b0 = bitpos - lead;                   /* a multiple of 8: the bits stay in one byte */
            s_stage[b0 >> 5] |= v << (32u - (b0 & 31u) - lead);
        }
    }
    __syncthreads();
    uint64_t done_bits = 0;                                      /* payload bits of the chunk's earlier tiles */

    for (uint64_t t0 = 0; t0 < len; t0 += TILE) {
        /* ---- load + look up ---- */
        const uint64_t my0 = t0 + (uint64_t)tid * PACK_SPT;
        uint32_t nsym = 0, mybits = 0;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};               /* the lane's 32 input bytes */
        uint32_t e[MODE == 2 ? 1 : PACK_SPT];
        if (my0 < len) {
            nsym = (uint32_t)dmin<uint64_t>(PACK_SPT, len - my0);
            const uint8_t *p = src + my0;
            if (nsym == PACK_SPT && (((uintptr_t)p) & 15u) == 0) {
                const uint4 v0 = load_stream16(reinterpret_cast<const uint4 *>(p));
                const uint4 v1 = load_stream16(reinterpret_cast<const uint4 *>(p) + 1);
                w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
            } else {
#pragma unroll
                for (int k = 0; k < PACK_SPT; k++)               /* (static indices: w[] stays in registers) */
                    if (k < (int)nsym) w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
            }
        }
        if (MODE != 2) {
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) {
                e[k] = (k < (int)nsym) ? (uint32_t)s_code[(w[k >> 2] >> (8 * (k & 3))) & 0xffu] : 0u;
                mybits += e[k] & 0xffu;
            }
        } else {
            for (uint32_t k = 0; k < nsym; k++) mybits += (uint32_t)(s_code[(w[k >> 2] >> (8 * (k & 3))) & 0xffu] & 0xffu);
        }
        uint32_t tile_bits;
        const uint32_t ex = block_excl_scan_u32<THREADS>(mybits, s_part, tile_bits);
        if (sub_groups) {                                        /* the sub-index: 2 bytes per 32 symbols */
            if (nsym) sub_groups[my0 / PACK_SPT] = (uint16_t)mybits;
            if ((tid & 63) == 0 && nsym) sub_tiles[my0 / HUF_SUB_TILE] = pay_rel0 + done_bits + ex;   /* (my0 = t0 + wave * 2 048) */
        }
        done_bits += tile_bits;
        const bool final_tile = t0 + TILE >= len;
        uint32_t tile_org = bitpos;                              /* stage bit of the tile's first bit (mod 2^32 once it has been flushed away) */

        /* ---- place, flush, carry: the lanes of one part at a time (one part unless codes are long) ---- */
#pragma unroll 1
        for (int part = 0; part < PARTS; part++) {
            const int l0 = part * (THREADS / PARTS), l1 = l0 + THREADS / PARTS;
            if (tid >= l0 && tid < l1 && nsym) {
                StageAcc a;
                a.start(s_stage, tile_org + ex);
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < PACK_SPT; k += 2) {      /* two codes of <= 16 bits per push */
                        const uint32_t x = e[k], y = e[k + 1];
                        const uint32_t ly = y & 0xffu;
                        a.push(((x >> 8) << ly) | (y >> 8), (x & 0xffu) + ly);
                    }
                } else if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < PACK_SPT; k++) a.push(e[k] >> 8, e[k] & 0xffu);
                } else {
                    for (uint32_t k = 0; k < nsym; k++) {
                        const hufcode_t x = (hufcode_t)s_code[(w[k >> 2] >> (8 * (k & 3))) & 0xffu];
                        uint32_t l = (uint32_t)(x & 0xffu);
                        const uint64_t c = x >> 8;
                        if (l > 32) {                            /* long code: high part first */
                            a.push((uint32_t)(c >> 32), l - 32);
                            l = 32;
                        }
                        a.push((uint32_t)c, l);
                    }
                }
                a.finish();
            }
            if (tid == l1 - 1) s_part[THREADS / 64] = ex + mybits;               /* the tile's bits up to this part's last lane */
            __syncthreads();
            const uint32_t end_bits = tile_org + uni32(s_part[THREADS / 64]);    /* stage bit behind this part */
            const bool final_part = final_tile && part == PARTS - 1;
            /* bytes [own_lo, own_hi) go out now: whole words, or on the chunk's very last flush the
             * bytes up to the one the next chunk owns (the block's last chunk: its zero-padded end) */
            const uintptr_t own_hi = base + (final_part ? (last ? (end_bits + 7u) >> 3 : end_bits >> 3) : (end_bits >> 5) * 4u);
            const uint32_t nfull = end_bits >> 5;                /* complete words of the stage */
            for (uint32_t g = (uint32_t)tid; 4u * g < nfull + 1u; g += THREADS) {
                const uintptr_t ga = base + 16u * g;
                uint4 v = *reinterpret_cast<const uint4 *>(s_stage + 4 * g);
                if (ga >= own_lo && ga + 16 <= own_hi) {
                    v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
                    store_pack16(reinterpret_cast<uint4 *>(ga), v);
                } else if (ga + 16 > own_lo && ga < own_hi) {
                    const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uintptr_t wa = ga + 4u * k;
                        if (wa >= own_lo && wa + 4 <= own_hi) *reinterpret_cast<uint32_t *>(wa) = __builtin_bswap32(ww[k]);
                        else if (wa + 4 > own_lo && wa < own_hi) {           /* a word shared with a neighbour: the chunk's ends only */
                            for (uint32_t j = 0; j < 4; j++)
                                if (wa + j >= own_lo && wa + j < own_hi) *reinterpret_cast<uint8_t *>(wa + j) = (uint8_t)(ww[k] >> (24 - 8 * j));
                        }
                    }
                }
                /* complete words are done with; the unfinished one is carried below */
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (4 * g + k < nfull) s_stage[4 * g + k] = 0;
            }
            __syncthreads();
            if (!final_part) {
                /* the unfinished word moves to its place relative to the new origin, the 16-byte
                 * group that holds it */
                if (tid == 0 && (nfull & ~3u)) {
                    const uint32_t v = s_stage[nfull];
                    s_stage[nfull] = 0;
                    s_stage[nfull & 3u] = v;
                }
                const uint32_t moved = (nfull & ~3u) * 32u;
                base += (uintptr_t)(moved >> 3);
                tile_org -= moved;
                bitpos = end_bits - moved;
                own_lo = own_hi;
                __syncthreads();
            }
        }
    }
}

/* SHORT = true: the host guarantees that no code of this launch is longer than 24 bits (any
 * Huffman merge order on n <= 121392 symbols gives depth <= 23, plus the wrap-root bit; the
 * deepest tree needs Fibonacci weights), so the 64-bit-entry path is not compiled. */
#ifndef PACK_CHUNK_WAVES_PER_SIMD
#define PACK_CHUNK_WAVES_PER_SIMD 4
#endif
template <int THREADS, bool SHORT>
__global__ __launch_bounds__(THREADS, PACK_CHUNK_WAVES_PER_SIMD) void pack_chunk_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                       uint64_t blocksize,
                                                       const hufcode_t *__restrict__ codetab,
                                                       const int16_t *__restrict__ treebuf,
                                                       const HufBlockMeta *__restrict__ meta,
                                                       uint64_t *__restrict__ offsets, TwoLevel sizes,
                                                       uint8_t *__restrict__ out, HufSubIndex sub, PackChunk ck)
{
    __shared__ hufcode_t s_code[SHORT ? HUF_NSYM / 2 : HUF_NSYM];   /* u32[256] on the short-code paths */
    __shared__ uint32_t s_part[THREADS / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[PACK_STAGE_WORDS2];

    const uint64_t blk = blockIdx.x / ck.cpb;
    const uint32_t c = (uint32_t)(blockIdx.x % ck.cpb);
    const uint64_t base = blk * blocksize;
    const uint64_t blen = dmin<uint64_t>(blocksize, n - base);
    const uint64_t sym0 = (uint64_t)c * ck.chunk_syms;
    if (sym0 >= blen) return;                                    /* the stream's last block may have fewer chunks */
    const uint64_t len = dmin<uint64_t>(ck.chunk_syms, blen - sym0);
    const HufBlockMeta m = meta[blk];
    const hufcode_t *codes = codetab + blk * HUF_NSYM;
    const int16_t *tb = treebuf + blk * HUF_TREE_STRIDE;
    uint64_t o0;
    if (sizes.local) {                   /* sizes were summed by hist_tree_kernel: publish the index entry */
        o0 = sizes.gprefix[blk / SCAN_GROUP] + sizes.local[blk];
        if (threadIdx.x == 0 && c == 0) offsets[blk] = o0;
    } else {
        o0 = offsets[blk];
    }
    const uint64_t pay_rel0 = ck.chunk_bits ? ck.chunk_bits[blk * ck.cpb + c] : 0ull;   /* chunk 0: 0 */
    const uint64_t P = (o0 + HUF_HEADER_FIXED + 2ull * m.tree_len) * 8ull + pay_rel0;
    const bool first = c == 0, last = sym0 + len == blen;
    uint64_t *sub_tiles = sub.tile_bits ? sub.tile_bits + blk * sub.tpb + sym0 / HUF_SUB_TILE : nullptr;
    uint16_t *sub_groups = sub.tile_bits ? sub.group_bits + blk * sub.gpb + sym0 / HUF_SUB_GROUP : nullptr;
    if (sub.tile_bits && first && m.tree_len != 5)
        for (int i = (int)threadIdx.x; i < HUF_NSYM; i += THREADS) sub.lens[blk * HUF_NSYM + i] = (uint8_t)(codes[i] & 0xffu);
    uint32_t *code32 = reinterpret_cast<uint32_t *>(s_code);
    const uint8_t *src = in + base + sym0;
    if (m.max_len <= 16)
        pack_segment<THREADS, 0>(src, len, blen, codes, tb, m.tree_len, first, last, out, o0, P, code32, s_part, s_stage, sub_tiles, sub_groups, pay_rel0);
    else if (SHORT || m.max_len <= 24)
        pack_segment<THREADS, 1>(src, len, blen, codes, tb, m.tree_len, first, last, out, o0, P, code32, s_part, s_stage, sub_tiles, sub_groups, pay_rel0);
    else if constexpr (!SHORT)
        pack_segment<THREADS, 2>(src, len, blen, codes, tb, m.tree_len, first, last, out, o0, P, code32, s_part, s_stage, sub_tiles, sub_groups, pay_rel0);
}

}  // namespace hufgpu
