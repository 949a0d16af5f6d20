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
 * lane is TOLD where its 32 symbols start, decodes them once through a 2^12-entry table (a second
 * level for codes of up to 18 bits) and stores them as one 32-byte sector.  What it is told is
 * verified, not trusted:
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
#define DSUB_SLACK_WORDS 16                 /* staged behind the last needed word: what a lane that runs wild (32 look-ups of at most 18 bits) or a long code's walk may look at */
#define DSUB_MAX_GROUP_BITS (DSUB_SPL * HUF_CODE_MAXBITS)
#define DSUB_CHUNK_SYMS 65536u               /* symbols one workgroup decodes: four tiles of 512 x 32 */

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

/* bit-serial walk from the root on the reversed stage (tables of dec_build_tables: sh.lr, sh.ent);
 * result as dec_rare_packed */
template <int THREADS>
__device__ __forceinline__ uint64_t dsub_rare_walk(const DecShared<THREADS> &sh, const uint32_t *top, uint32_t pos, uint32_t lim)
{
    uint32_t node = 0;
    uint32_t p = pos;
    for (;;) {
        if (p >= lim) return (uint64_t)CW_EXH << 40;
        const uint32_t bit = (rev_word(top, p >> 5) >> (31u - (p & 31u))) & 1u;
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

/* (Latency is what this costs - a workgroup builds its tables before it can do anything else: the tree entries
 * go from global memory straight into the registers that classify them, the checks do not vote one by one
 * - what follows a failed check works on clamped values and is thrown away by the one vote at the end -, the
 * scans are DPP scans with one barrier each, and blocks without codes beyond 12 bits skip the second level.) */
/* what a thread of the table build needs from global memory: the three aligned dwords that hold tree entries
 * 2t .. 2t+3, and (threads 0..63) four of the claimed lengths - requested early, used by dsub_fast_tables */
struct DsubTreeWords {
    uint32_t d0, d1, d2, lens4, mis;
};
template <int THREADS>
__device__ __forceinline__ DsubTreeWords dsub_tree_request(const uint8_t *tree, int tree_len, const uint8_t *__restrict__ lens_g)
{
    DsubTreeWords w;
    const int tid = (int)threadIdx.x;
    const uintptr_t a = (uintptr_t)uni64((uint64_t)(uintptr_t)tree);
    w.mis = (uint32_t)(a & 3u);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(a - w.mis);
    const uint32_t nbytes = w.mis + 2u * (uint32_t)(tree_len > 0 ? tree_len : 0);      /* bytes from q[0] to the tree's end */
    const uint32_t t4 = 4u * (uint32_t)tid;
    w.d0 = (t4 < nbytes) ? q[tid] : 0u;
    w.d1 = (t4 + 4u < nbytes) ? q[tid + 1] : 0u;
    w.d2 = (t4 + 8u < nbytes) ? q[tid + 2] : 0u;
    w.lens4 = (tid < 64) ? reinterpret_cast<const uint32_t *>(lens_g)[tid] : 0u;
    return w;
}

template <int THREADS>
__device__ bool dsub_fast_tables(DecShared<THREADS> &sh, int tree_len, const DsubTreeWords &tw)
{
    typedef DsubFastLds<THREADS> F;
    constexpr int ENT = DecShared<THREADS>::ENT;
    constexpr int WAVES = THREADS / 64;
    static_assert(THREADS * 2 >= ENT - 2 && THREADS >= 256 && WAVES <= 8, "two entries per thread");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t K = (uint32_t)(tree_len - 1) >> 2;
    if (tid == 0) { sh.fastk = 0; sh.l2n = 0; }
    if (tree_len < 9 || tree_len > HUF_TREE_MAX || ((tree_len - 1) & 3) != 0) return false;     /* uniform */
    uint8_t *s_lens = reinterpret_cast<uint8_t *>(sh.pay);                      /* [256] the claimed lengths by byte value */
    uint16_t *s_pos = reinterpret_cast<uint16_t *>(sh.pay + 64);                /* [256] entry index of the k-th leaf */
    uint32_t *s_cpart = sh.wtile;                                               /* [3][WAVES] partial sums of the code scan */
    static_assert(sizeof(sh.wtile) >= 3 * WAVES * sizeof(uint32_t), "partials of the code scan");
    bool ok = true;
    /* ---- shape: leaves, nodes, markers.  Thread t looks at entries 2t .. 2t+3 (three aligned dwords, shifted
     *      by the tree's byte misalignment; entries at or past tree_len read -1) ---- */
    uint32_t k0;                                                                /* leaves in front of entry 2t */
    bool l0, l1;
    int e0, e1;
    {
        const uint32_t mis = tw.mis, d0 = tw.d0, d1 = tw.d1, d2 = tw.d2;
        if (tid < 64) reinterpret_cast<uint32_t *>(s_lens)[tid] = tw.lens4;
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
        __syncthreads();                                                        /* (also: s_lens is written) */
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            const uint32_t x = sh.part[i];
            if (i < wave) base += x;
            tot += x;
        }
        if ((tot & 0xffffu) != K || (tot >> 16) != 2u * K) ok = false;
        k0 = (base + inc - mine) & 0xffffu;
        uint32_t k = k0;
        if (l0 && k < 256u) { s_pos[k] = (uint16_t)i0; F::sym(sh)[k] = (uint8_t)e0; k++; }
        if (l1 && k < 256u) { s_pos[k] = (uint16_t)(i0 + 1); F::sym(sh)[k] = (uint8_t)e1; }
    }
    __syncthreads();
    /* ---- claimed lengths -> codes; they must fill the left half of the code space exactly.  A code's share of
     *      the 32-bit code space is 2^(32 - d) <= 2^30: scanned as two 16-bit halves (DPP, no 64-bit shuffles) ---- */
    uint32_t d = 2;
    bool anylong;
    {
        uint32_t whi = 0, wlo = 0;
        if ((uint32_t)tid < K) {
            d = s_lens[F::sym(sh)[tid]];
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
        if ((uint32_t)tid < K) {
            F::code(sh)[tid] = (uint32_t)(base + (((uint64_t)(ihi - whi)) << 16) + (ilo - wlo));
            F::len(sh)[tid] = (uint8_t)d;
        }
    }
    __syncthreads();
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
                e[j] = DSE_BAD;                  /* the first bit leaves the tree (the root has no right child) */
            } else {
                while (k + 1u < K && code[k + 1u] <= v) k++;
                const uint32_t dk = F::len(sh)[k];
                e[j] = (dk <= (uint32_t)DEC_LUT_BITS) ? (((uint32_t)F::sym(sh)[k] << 8) | dk) : (uint32_t)DSE_LONG;
            }
        }
        *reinterpret_cast<uint4 *>(sh.lut + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
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

/* The tables of dec_build_tables (any grammar-valid tree; taken when dsub_fast_tables declines) in the sub-index
 * path's entry format: eight entries per thread.  `long` codes are then walked from the root. */
template <int THREADS>
__device__ __forceinline__ void dsub_convert_tables(DecShared<THREADS> &sh)
{
    static_assert((1 << DEC_LUT_BITS) == THREADS * 8, "eight entries per thread");
    uint32_t *t = reinterpret_cast<uint32_t *>(sh.lut) + 4 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t two = t[i];
        uint32_t r = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t e = (two >> (16 * h)) & 0xffffu;
            const uint32_t n = (e >= DEC_E_LONG) ? DSE_LONG : (e >= DEC_E_BAD) ? DSE_BAD : (((e & 0xffu) << 8) | (e >> 8));
            r |= n << (16 * h);
        }
        t[i] = r;
    }
    if (threadIdx.x == 0) { sh.fastk = 0; sh.l2n = 0; }
    __syncthreads();
}

/* a second-level entry of the first table resolved with the 32 bits at the codeword's start */
template <int THREADS>
__device__ __forceinline__ uint32_t dsub_l2(const DecShared<THREADS> &sh, uint32_t e, uint32_t bits32)
{
    const uint32_t nb = ((e >> 5) & 6u) + 2u;
    return reinterpret_cast<const uint16_t *>(sh.ent)[((e >> 8) << 2) + ((bits32 << DEC_LUT_BITS) >> (32u - nb))];
}

/* a codeword longer than the table's 12 bits with dsub_fast_tables' tables: the leaf whose code
 * interval holds the 32 bits at the position; result as dec_rare_packed */
template <int THREADS>
__device__ __forceinline__ uint64_t dec_rare_fast(const DecShared<THREADS> &sh, const uint32_t *top, uint32_t pos, uint32_t lim)
{
    typedef DsubFastLds<THREADS> F;
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
template <int THREADS>
__device__ __forceinline__ uint32_t dsub_next(const DecShared<THREADS> &sh, RevReader &rd, uint32_t lim, bool &ok)
{
    uint32_t e = sh.lut[rd.index()];
    if (__builtin_expect(__ballot(DSE_LEN(e) == 0u) != 0ull, 0)) {
        if (DSE_IS_L2(e)) {
            e = dsub_l2<THREADS>(sh, e, rd.hi);
        } else if (e == DSE_LONG) {
            const uint64_t r = sh.fastk ? dec_rare_fast<THREADS>(sh, rd.top, rd.pos(), lim) : dsub_rare_walk<THREADS>(sh, rd.top, rd.pos(), lim);
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
__device__ __noinline__ bool dsub_redo_group(const DecShared<THREADS> &sh, const uint32_t *top, uint32_t s, uint32_t nsym,
                                             uint32_t lim, uint32_t gb, uint8_t *dst)
{
    bool ok = true;
    RevReader rd;
    rd.top = top;
    rd.load(s);
    for (uint32_t k = 0; k < nsym; k++) {
        dst[k] = (uint8_t)(dsub_next<THREADS>(sh, rd, lim, ok) >> 8);
        if (rd.avail <= 32) rd.refill();
    }
    return ok && rd.pos() - s == gb;
}

/* A wave tile whose bits do not fit the wave's slice in one piece (codes far longer than the 9-bit average), or
 * whose words cannot be loaded 16 bytes at a time (the stream ends right behind them): the lanes in several
 * runs, staged word by word, decoded step by step.  Rare, and out of line for the hot path's registers.
 * first = the tile's first payload bit, ex / incl = the lane's exclusive / inclusive bit counts inside the tile,
 * active = the lane's group is to be decoded. */
template <int THREADS>
__device__ __noinline__ bool dsub_tile_slow(const DecShared<THREADS> &sh, uint32_t *top, const uint8_t *pay, uint64_t pay_bytes,
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
 * Tables are in sh, dsub_prefetch has been called.  Returns true (workgroup-uniform) when everything
 * was verified; *end_bit = the payload bit behind the chunk's last symbol.  T0 = the chunk's first
 * payload bit, as told; readable = bytes that may be loaded from `pay` on (to the end of the stream); safe = an
 * address with 19 readable bytes behind it (the block's header and tree).
 *
 * After one scan of the group counts every WAVE is on its own: a wave tile = 64 groups = 2 048
 * symbols; wave w takes tiles w, w + 8, ...; it stages the tile's payload words in its private LDS
 * slice (LDS operations of one wave are in order: no barrier), decodes, stores.  The words of the wave's
 * NEXT tile are requested before it decodes this one and wait in twelve registers: the HBM latency
 * (a quarter of a workgroup's life when every tile waited for its own loads) passes under the decoding. */
template <int THREADS>
__device__ bool decode_payload_sub(DecShared<THREADS> &sh, const uint8_t *tree, int tree_len, const uint8_t *lens_g,
                                   const uint8_t *pay, uint64_t pay_bytes, uint64_t readable, const uint8_t *safe,
                                   uint64_t sym0, uint64_t sym1, uint64_t T0, uint8_t *gout, uint64_t *end_bit)
{
    /* the table build's words from global memory are on their way while the group counts are scanned */
    const DsubTreeWords tw = dsub_tree_request<THREADS>(tree, tree_len, lens_g);
    typedef DsubLds<THREADS> L;
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CAP_BITS = L::CAP_BITS;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = (int)uni32((uint32_t)tid >> 6);       /* (the compiler does not know that a wave's threads share tid >> 6) */
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uint16_t *s_gb = reinterpret_cast<const uint16_t *>(L::gb(sh));
    uint32_t *top = L::slice(sh, wave) + (L::SLICE_WORDS - 1u);      /* staged word g of the wave's tile at top[-g] */
    const uint32_t ngrp = (uint32_t)((sym1 - sym0 + DSUB_SPL - 1) / DSUB_SPL);
    bool ok = true;
    unsigned long long pt = DPROF_T();

    /* the group counts requested by dsub_prefetch have landed (every wave waits for its own
     * requests, the barrier makes them everybody's) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sym0 == 0 && T0 != 0) ok = false;          /* (a) */

    /* bits of every wave tile, then their inclusive sums (lane q of every wave holds tile q's) */
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
    const uint32_t tincl = wave_incl_scan_u32(((uint32_t)lane < L::WTILES) ? sh.wtile[lane] : 0u);   /* lane q: bits of tiles 0 .. q */
    const uint32_t chunk_bits = wave_lane_u32(tincl, 63);
    if (T0 + chunk_bits > pay_bits) {                                /* (b): bits past the payload (the same in every thread) */
        *end_bit = 0;
        return false;
    }
    DPROF_ADD(8, pt);

    typedef const __attribute__((address_space(3))) uint32_t *lds_words;
    typedef const __attribute__((address_space(3))) uint16_t *lds_halves;
    typedef const __attribute__((address_space(1))) uint8_t *global_bytes;
    typedef uint32_t dwords4 __attribute__((ext_vector_type(4)));
    typedef dwords4 dwords4_a4 __attribute__((aligned(4)));                       /* 16 bytes at a 4-byte aligned address */
    typedef const __attribute__((address_space(1))) dwords4_a4 *global_q4;
    typedef __attribute__((address_space(1))) uint8_t *global_out;
    typedef __attribute__((address_space(1))) dwords4 *global_out4;
    /* The position register: R = 8 * (LDS byte address of top[-1]) + 31 - (position - 1), position = bits from the
     * stage's word 0.  With the words in reversed order, (R >> 3) & ~3 IS the LDS address of the SECOND word of the
     * pair that holds bit position - 1, and the low five bits of R are the amount v_alignbit_b32 shifts that pair
     * by to leave the 32 bits at the position (0..31: the position is never the pair's first bit) - no negation,
     * and a codeword is R -= len. */
    const uint32_t r_origin = 8u * (uint32_t)(uintptr_t)(lds_words)(top - 1) + 32u;       /* R of position 0 */
    const uint32_t lut_addr = (uint32_t)(uintptr_t)(lds_halves)sh.lut;
    const uintptr_t pay_a = (uintptr_t)uni64((uint64_t)(uintptr_t)pay);
    const uintptr_t cout_a = (uintptr_t)uni64((uint64_t)(uintptr_t)(gout + sym0));           /* the chunk's first output byte */
    const global_out cout = (global_out)cout_a;
    const uint32_t nchunk = (uint32_t)(sym1 - sym0);
    const uint64_t end_a = uni64((uint64_t)pay_a + readable);                                 /* the stream's end */
    const uintptr_t safe_a = ((uintptr_t)uni64((uint64_t)(uintptr_t)safe) + 3u) & ~(uintptr_t)3;

    /* A tile's staging, all wave-uniform: word 0 of the stage is the aligned 32-bit word of MEMORY that holds the
     * tile's first bit (the stage words are then byte-swapped dwords, whatever the payload's alignment);
     * `lead` = bits of that word in front of the tile, `nwords` = words to stage; `quick` = its bits fit the slice
     * and 16-byte loads of all of them stay inside the stream. */
    uint32_t q = (uint32_t)wave;
    uint4 V[3];
    uint32_t lead = 0, nwords = 0;
    bool quick = false;
/* (The loads are issued - and waited for further down - for every tile, also when there is nothing to load: the words
 * of the block's first bytes then, not stored.  Loads under a condition leave the compiler with "maybe pending"
 * registers at the top of the loop, and its wait for those is a wait for the previous tile's STORES as well.) */
#define DSUB_GEOMETRY(Q_, VALID_)                                                                              \
    {                                                                                                         \
        const uint32_t qq_ = (VALID_) ? (Q_) : 0u;                                                            \
        const uint32_t t0_ = qq_ ? wave_lane_u32(tincl, qq_ - 1u) : 0u;                                        \
        const uint32_t tb_ = wave_lane_u32(tincl, qq_) - t0_;                                                  \
        const uint64_t first_ = T0 + t0_;                                                                     \
        const uintptr_t a_ = pay_a + (uintptr_t)(first_ >> 3);                                                \
        lead = (uint32_t)(first_ & 7u) + 8u * (uint32_t)(a_ & 3u);                                            \
        const uint32_t need_ = lead + tb_;                                                                    \
        nwords = ((need_ + 31u) >> 5) + DSUB_SLACK_WORDS + 2u;                                                \
        /* (the aligned word may begin up to 3 bytes in front of the payload: header and tree lie there) */      \
        const int64_t left_ = (int64_t)(end_a - (uint64_t)(a_ & ~(uintptr_t)3));    /* readable bytes from the aligned word on */ \
        quick = (VALID_) && need_ <= CAP_BITS && left_ >= (int64_t)(4u * nwords + 16u);                       \
        global_bytes base_ = (global_bytes)(uintptr_t)uni64((uint64_t)(quick ? (a_ & ~(uintptr_t)3) : safe_a)); \
        const uint32_t last16_ = uni32(quick ? 4u * ((nwords - 1u) & ~3u) : 0u);                              \
        if (!quick) nwords = 0;                                                                               \
        uint32_t ln_ = (uint32_t)lane;                                                                        \
        asm volatile("" : "+v"(ln_));            /* (offsets computed here, not kept in registers around the tile loop) */ \
        _Pragma("unroll")                                                                                     \
        for (int k = 0; k < 3; k++) {                                                                         \
            /* every lane loads (lanes past the needed words load the last ones again and do not store them) */ \
            const uint32_t off_ = dmin<uint32_t>(16u * ln_ + 1024u * (uint32_t)k, last16_);                   \
            const dwords4 v_ = *(global_q4)(base_ + off_);                                                    \
            V[k] = make_uint4(v_.x, v_.y, v_.z, v_.w);                                                        \
        }                                                                                                     \
    }
/* the loaded words into the slice: four byte-swapped dwords per lane and step, one 16-byte LDS store
 * (words i4 .. i4 + 3 at top[-i4 - 3 .. -i4]) */
#define DSUB_TO_SLICE()                                                                                        \
    {                                                                                                         \
        uint32_t ln_ = (uint32_t)lane;                                                                        \
        asm volatile("" : "+v"(ln_));                                                                         \
        _Pragma("unroll")                                                                                     \
        for (int k = 0; k < 3; k++) {                                                                         \
            const uint32_t i4 = 4u * (ln_ + 64u * (uint32_t)k);                                               \
            if (i4 < nwords)                                                                                  \
                *reinterpret_cast<uint4 *>(top - (i4 + 3u)) =                                                 \
                    make_uint4(__builtin_bswap32(V[k].w), __builtin_bswap32(V[k].z), __builtin_bswap32(V[k].y), __builtin_bswap32(V[k].x)); \
        }                                                                                                     \
    }
    DSUB_GEOMETRY(q, q * 64u < ngrp)
    /* ... and the wave's first tile is on its way while the tables are built.  Tables the quick way from the
     * sub-index's code lengths, checked against the tree; any other tree: walked (dec_build_tables) */
    {
        unsigned long long kt = DPROF_T();
#ifdef DSUB_TABLES_TWICE        /* (what one table build costs where it stands: the kernel with two of them) */
        dsub_fast_tables<THREADS>(sh, tree_len, tw);
#endif
        if (!dsub_fast_tables<THREADS>(sh, tree_len, tw)) {
            int leaf = -1;
            const int rc = dec_build_tables<THREADS, false>(sh, tree, tree_len, &leaf);
            if (rc != HUFE_OK || leaf >= 0) {                       /* (an unusual one-leaf tree: the exact decoder's) */
                *end_bit = 0;
                return false;
            }
            dsub_convert_tables<THREADS>(sh);
        }
        DPROF_ADD(3, kt);
    }
    const bool use_l2 = uni32(sh.l2n) != 0u;                          /* the block has codes in a second-level table */
    DSUB_TO_SLICE()
    /* what the loop leaves for later (bit i = the wave's i-th tile): tiles that are not `quick`, and - per lane -
     * groups to be decoded again step by step.  Those go through dsub_tile_slow BEHIND the loop: a call inside
     * it would have the loop's registers saved and restored around a path that is next to never taken. */
    uint32_t slow_tiles = 0, redo_tiles = 0;

#pragma unroll 1
    for (uint32_t ti = 0; q * 64u < ngrp; q += WAVES, ti++) {
        pt = DPROF_T();
        const uint32_t g = q * 64u + (uint32_t)lane;
        const uint32_t my0 = g * DSUB_SPL;                              /* relative to the chunk: 32-bit arithmetic, one offset register */
        uint32_t nsym = 0, gb = 0;
        if (g < ngrp) {
            nsym = dmin<uint32_t>(DSUB_SPL, nchunk - my0);
            gb = dmin<uint32_t>((uint32_t)s_gb[g], DSUB_MAX_GROUP_BITS);
        }
        const uint32_t s = lead + wave_incl_scan_u32(gb) - gb;          /* the stage bit of my first symbol */
        const bool cur_quick = quick;
        /* the next tile's words are requested now and arrive while this one is decoded */
        DSUB_GEOMETRY(q + WAVES, (q + WAVES) * 64u < ngrp)
        DPROF_ADD(9, pt); pt = DPROF_T();
        if (!cur_quick) {
            slow_tiles |= 1u << ti;
        } else if (nsym) {
            bool redo = nsym != DSUB_SPL;                       /* the block's last, short group */
            bool group_ok = false;
            if (nsym == DSUB_SPL) {
                /* The common case has no branch.  Every two symbols the 32 bits at the position are read
                 * again from the stage (one ds_read2_b32, one v_alignbit_b32) - no bit buffer to refill -
                 * and looked up twice; the entries' low five bits shift the window and their sum moves the
                 * position as they stand.  An entry that is not a leaf has length 0: the lane stands still
                 * from then on (a `long` code, a walk that leaves the tree: the low byte is 0) and its last
                 * look-up tells; such a lane decodes its group again behind the loop, step by step.
                 * The lane's 32 bytes leave as two 16-byte stores back to back, DEFAULT cache policy: one whole
                 * 32-byte sector (a store per 16 symbols left half-written sectors in L2 for a round, and one in
                 * nine of them was evicted like that and written twice; streaming stores reach HBM as partial
                 * writes altogether).
                 * (Measured and dropped: a 64-bit buffer with refills, 14 instead of 8 instructions per
                 * symbol; a lane decoding the two halves of its group side by side.) */
                const bool aligned = (cout_a & 15u) == 0;             /* (the offset is a multiple of 32) */
                const uint32_t r0 = r_origin - s;
                uint32_t R = r0;
                uint32_t special = 0, e_last = 0;
/* L2 = the block has second-level entries (sh.l2n): a lookup that meets one - decided for the whole
 * wave by a ballot - goes on to the second table with the bits behind the 12-bit prefix; one that cannot
 * (too few bits left in the window) stays what it is and is remembered in `special`.  Blocks
 * without such codes (zipf255, uniform bytes) run the loop without the ballots. */
#define DSUB_WINDOW(PAIR, L2)                                                                                 \
                {                                                                                             \
                    lds_words wp_ = (lds_words)(uintptr_t)((R >> 3) & ~3u);                                    \
                    const uint32_t d1_ = __builtin_amdgcn_alignbit(wp_[1], wp_[0], R);                         \
                    uint32_t e1_ = *(lds_halves)(uintptr_t)(lut_addr + ((d1_ >> 19) & 0x1ffeu));              \
                    if (L2 && __ballot(DSE_IS_L2(e1_))) {                                                     \
                        if (DSE_IS_L2(e1_)) e1_ = dsub_l2<THREADS>(sh, e1_, d1_);                             \
                    }                                                                                         \
                    const uint32_t d2_ = d1_ << (e1_ & 31u);                                                  \
                    uint32_t e2_ = *(lds_halves)(uintptr_t)(lut_addr + ((d2_ >> 19) & 0x1ffeu));              \
                    if (L2 && __ballot(DSE_IS_L2(e2_))) {                                                     \
                        /* (the window has 32 - len bits left) */                                             \
                        if (DSE_IS_L2(e2_) && DSE_LEN(e1_) + DEC_LUT_BITS + DSUB_L2_BITS <= 32u)              \
                            e2_ = dsub_l2<THREADS>(sh, e2_, d2_);                                             \
                    }                                                                                         \
                    if (L2) {                                                                                 \
                        special |= e1_ | e2_;                                                                 \
                        asm volatile("" : "+v"(special));     /* (now: not 32 entries kept for one big OR at the end) */ \
                    }                                                                                         \
                    R -= (e1_ + e2_) & 0xffu;                                                                 \
                    PAIR = __builtin_amdgcn_perm(e2_, e1_, 0x0c0c0501u);                                      \
                    e_last = e2_;                                                                             \
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
                    /* (the address from the 32-bit offset right here: scalar base + offset register, not a   \
                     *  64-bit address kept in two registers around the loop) */                             \
                    uint32_t o_ = my0;                                                                        \
                    asm volatile("" : "+v"(o_));                                                              \
                    if (aligned) {                                                                            \
                        dwords4 a4_, b4_;                                                                     \
                        a4_.x = w[0]; a4_.y = w[1]; a4_.z = w[2]; a4_.w = w[3];                               \
                        b4_.x = w[4]; b4_.y = w[5]; b4_.z = w[6]; b4_.w = w[7];                               \
                        *(global_out4)(cout + o_) = a4_;                                                      \
                        *(global_out4)(cout + o_ + 16) = b4_;                                                 \
                    } else {                                                                                  \
                        _Pragma("unroll")                                                                     \
                        for (int k = 0; k < 32; k++) (cout + o_)[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));  \
                    }                                                                                         \
                }
                if (use_l2) { DSUB_ROUNDS(true) } else { DSUB_ROUNDS(false) }
#undef DSUB_ROUNDS
#undef DSUB_WINDOW
                redo = DSE_LEN(e_last) == 0u || (special & DSE_L2) != 0u;
                group_ok = r0 - R == gb;                        /* (b): exactly the bits of the group */
            }
            if (redo) redo_tiles |= 1u << ti;
            else if (!group_ok) ok = false;
        }
        DPROF_ADD(10, pt);
        /* the slice is free: the next tile's words (the wait for them counts the stores behind them out) */
        DSUB_TO_SLICE()
    }
#undef DSUB_TO_SLICE
#undef DSUB_GEOMETRY
    if (__builtin_expect(slow_tiles != 0u || __ballot(redo_tiles != 0u) != 0ull, 0)) {
#pragma unroll 1
        for (uint32_t ti = 0; ti < L::WTILES / WAVES; ti++) {
            const bool whole = ((slow_tiles >> ti) & 1u) != 0u;
            const bool mine = whole || ((redo_tiles >> ti) & 1u) != 0u;
            if (!__ballot(mine)) continue;
            q = (uint32_t)wave + ti * WAVES;
            const uint32_t g = q * 64u + (uint32_t)lane;
            const uint32_t my0 = g * DSUB_SPL;
            uint32_t nsym = 0, gb = 0;
            if (g < ngrp) {
                nsym = dmin<uint32_t>(DSUB_SPL, nchunk - my0);
                gb = dmin<uint32_t>((uint32_t)s_gb[g], DSUB_MAX_GROUP_BITS);
            }
            const uint32_t incl = wave_incl_scan_u32(gb);
            const uint64_t tfirst = T0 + (q ? wave_lane_u32(tincl, q - 1u) : 0u);
            if (!dsub_tile_slow<THREADS>(sh, top, pay, pay_bytes, tfirst, incl - gb, incl, nsym, mine, gout + sym0 + my0)) ok = false;
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
#ifdef DSUB_LDS_PAD             /* (occupancy experiments: fewer workgroups per CU) */
    __shared__ uint32_t lds_pad[DSUB_LDS_PAD / 4];
    if (stream_len == 0x123456789abcull) lds_pad[threadIdx.x] = 1;
#endif
    static_assert(DSUB_CHUNK_SYMS % (THREADS * DSUB_SPL) == 0 && (THREADS * DSUB_SPL) % HUF_SUB_TILE == 0, "chunks are whole tiles");
    const int tid = (int)threadIdx.x;
    const uint64_t blk = blockIdx.x / cpb;
    const uint32_t c = (uint32_t)(blockIdx.x % cpb);
    unsigned long long kt = DPROF_T();
    /* Everything about the block is the same in all lanes: kept in SGPRs (held in VGPRs these values pushed the
     * tile loop's state out to scratch), and everything is REQUESTED before the first of it is looked at: six
     * scalar loads, one wait (one after the other they were a sixth of a workgroup's life). */
    HufDecodeMeta m = dmeta[blk];
    const uint64_t out_group = lens.gprefix[blk / SCAN_GROUP], out_local = lens.local[blk];
    const uint64_t off0 = offsets[blk], off1 = offsets[blk + 1];
    const uint64_t t0_told = sub.tile_bits[blk * sub.tpb + (uint64_t)c * (DSUB_CHUNK_SYMS / HUF_SUB_TILE)];   /* (c < cpb: inside the block's row) */
    pin_uniform(m.block_len); pin_uniform(out_group); pin_uniform(out_local); pin_uniform(off0); pin_uniform(off1); pin_uniform(t0_told);
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
        dsub_prefetch<THREADS>(sh, sub.group_bits + blk * sub.gpb + sym0 / DSUB_SPL,
                               (uint32_t)((sym1 - sym0 + DSUB_SPL - 1) / DSUB_SPL));
        const uint64_t T0 = uni64(t0_told);                           /* first payload bit of the chunk, as told */
        DPROF_ADD(2, kt);
        uint64_t end_bit = 0;
        good = decode_payload_sub<THREADS>(sh, tree, m.tree_len, sub.lens + blk * HUF_NSYM, pay, pay_bytes,
                                           stream_len - (uint64_t)(pay - stream), stream + o0, sym0, sym1, T0, out + obase, &end_bit);
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
