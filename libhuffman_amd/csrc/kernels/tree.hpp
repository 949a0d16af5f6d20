/* tree.hpp - tree_kernel / tree_fast_wave: tree build, codes, serialized tree (src/tree.c:292-427, 12-47, 233-289).
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "util.hpp"

namespace hufgpu {

/* ======================================================================================
 * tree_kernel - replaces huf_tree_from_histogram (src/tree.c:292-427), the code walk
 * (src/tree.c:12-47 + src/encoder.c:40-81) and huf_tree_serialize (src/tree.c:233-289).
 *
 * One wavefront per block.  The 512 rate slots live in registers, 8 per lane (slot = lane +
 * 64*j).  A slot's sort key is (rate << 9) | (511 - slot): the plain minimum of the keys is
 * the reference's selection order "rate ascending, index descending" (tree.c:329-352), and
 * keys are unique.  Each round reduces the two smallest keys across the wave, makes the
 * smaller one the left child and the other the right child of the new node (tree.c:390-408),
 * and stops on the round that finds a single survivor, which becomes the left-only wrap root
 * (tree.c:410-413).  K = uint32_t serves blocks shorter than 2^22 bytes, uint64_t the rest.
 *
 * Then, level by level from the root: code bits, depth and the preorder position of every
 * node (position of a right child = parent + 1 + entries of the left subtree, a subtree with
 * L leaves holding 4L-1 entries), which gives codes and the serialized tree without recursion.
 * ==================================================================================== */
template <typename K, typename H = uint32_t>
__global__ __launch_bounds__(64) void tree_kernel(const H *__restrict__ hist, uint64_t n,
                                                  uint64_t blocksize, hufcode_t *__restrict__ codetab,
                                                  int16_t *__restrict__ treebuf,
                                                  HufBlockMeta *__restrict__ meta)
{
    __shared__ int16_t s_left[HUF_NSLOT];
    __shared__ int16_t s_right[HUF_NSLOT];
    __shared__ uint16_t s_leaves[HUF_NSLOT];  /* leaves below each slot */
    __shared__ uint16_t s_depth[HUF_NSLOT];   /* 0xffff = not reached */
    __shared__ uint16_t s_pos[HUF_NSLOT];     /* preorder position */
    __shared__ uint64_t s_code[HUF_NSLOT];
    __shared__ int16_t s_tree[HUF_TREE_STRIDE];

    const K KMAX = ~(K)0;
    const int lane = lane_id();
    const uint64_t blk = blockIdx.x;
    const H *h = hist + blk * HUF_NSYM;

    K key[8];
    H rate[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = lane + 64 * j;
        rate[j] = h[slot];
        key[j] = rate[j] ? (((K)rate[j] << 9) | (K)(511 - slot)) : KMAX;
    }
#pragma unroll
    for (int j = 4; j < 8; j++) key[j] = KMAX;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int slot = lane + 64 * j;
        s_left[slot] = -1;
        s_right[slot] = -1;
        s_leaves[slot] = (j < 4 && rate[j & 3]) ? 1 : 0;
        s_depth[slot] = 0xffffu;
    }
    __syncthreads();

    int node = HUF_NSYM;
    int root = -1;
    for (;;) {
        K a = KMAX, b = KMAX;              /* two smallest keys of this lane */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const K k = key[j];
            const K t = dmax(a, k);
            a = dmin(a, k);
            b = dmin(b, t);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {  /* butterfly: disjoint lane groups, unique keys */
            const K oa = shfl_xor_key(a, o);
            const K ob = shfl_xor_key(b, o);
            const K t = dmax(a, oa);
            a = dmin(a, oa);
            b = dmin(dmin(b, ob), t);
        }
        if (a == KMAX) {                   /* tree.c:355-358 (only for an empty histogram) */
            root = node - 1;
            break;
        }
        const int i1 = 511 - (int)(a & (K)511);
        if (b == KMAX) {                   /* tree.c:410-413: single survivor -> left-only root */
            if (lane == 0) {
                s_left[node] = (int16_t)i1;
                s_right[node] = -1;
                s_leaves[node] = s_leaves[i1];
            }
            root = node;
            node++;
            break;
        }
        const int i2 = 511 - (int)(b & (K)511);
        const K sum = (a >> 9) + (b >> 9);  /* tree.c:407 */
        const K nk = (sum << 9) | (K)(511 - node);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int slot = lane + 64 * j;
            if (slot == i1 || slot == i2) key[j] = KMAX;   /* tree.c:396,403 */
            if (slot == node) key[j] = nk;
        }
        if (lane == 0) {
            s_left[node] = (int16_t)i1;
            s_right[node] = (int16_t)i2;
            s_leaves[node] = (uint16_t)(s_leaves[i1] + s_leaves[i2]);
        }
        node++;
    }
    __syncthreads();

    const int nodes = node;
    const int nleaves = (root >= 0) ? (int)s_leaves[root] : 0;
    const int tree_len = (root >= 0) ? 4 * nleaves + 1 : 1;

    for (int i = lane; i < HUF_TREE_STRIDE; i += 64) s_tree[i] = -1;
    if (lane == 0 && root >= 0) {
        s_depth[root] = 0;
        s_code[root] = 0;
        s_pos[root] = 0;
    }
    __syncthreads();

    for (int d = 0; d < HUF_NSLOT; d++) {
        bool any = false;
#pragma unroll
        for (int j = 4; j < 8; j++) {
            const int slot = lane + 64 * j;
            if (slot < nodes && s_depth[slot] == (uint16_t)d) {
                any = true;
                const int l = s_left[slot], r = s_right[slot];
                const uint64_t c = s_code[slot];
                const int p = s_pos[slot];
                s_tree[p] = (int16_t)slot;
                s_depth[l] = (uint16_t)(d + 1);
                s_code[l] = c << 1;
                s_pos[l] = (uint16_t)(p + 1);
                if (l < HUF_NSYM) s_tree[p + 1] = (int16_t)l;
                if (r >= 0) {
                    const int pr = p + 1 + 4 * (int)s_leaves[l] - 1;
                    s_depth[r] = (uint16_t)(d + 1);
                    s_code[r] = (c << 1) | 1u;
                    s_pos[r] = (uint16_t)pr;
                    if (r < HUF_NSYM) s_tree[pr] = (int16_t)r;
                }
            }
        }
        __syncthreads();
        if (!__any(any)) break;
    }

    uint64_t bits = 0;
    uint32_t maxlen = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = lane + 64 * j;
        hufcode_t e = 0;
        if (rate[j]) {
            const uint32_t len = s_depth[slot];
            e = (s_code[slot] << 8) | (hufcode_t)len;
            bits += (uint64_t)rate[j] * len;
            maxlen = dmax(maxlen, len);
        }
        codetab[blk * HUF_NSYM + slot] = e;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        bits += shfl_xor_u64(bits, o);
        maxlen = dmax(maxlen, wave_xor_any(maxlen, o));
    }
    int16_t *tb = treebuf + blk * HUF_TREE_STRIDE;
    for (int i = lane; i < tree_len; i += 64) tb[i] = s_tree[i];
    if (lane == 0) {
        HufBlockMeta m;
        m.tree_len = (uint32_t)tree_len;
        m.max_len = maxlen;
        m.payload_bits = bits;
        meta[blk] = m;
    }
    (void)n;
    (void)blocksize;
}

/* ======================================================================================
 * tree_fast_wave - same algorithm and outputs as tree_kernel<uint32_t>, tuned for the wave (it runs
 * as the tail of hist_tree_kernel).
 *   - At most 256 items are alive at any time (k leaves, one fewer after every merge), so the
 *     live keys fit a pool of 4 registers per lane; the node created by a merge takes over the
 *     pool position of the smaller of the two items it replaces.  A key still carries the
 *     item's logical index (rate<<9 | 511-index), so the selection order is unchanged.
 *   - The wave minimum is a DPP reduction (quad_perm, row_half_mirror, row_mirror, row_bcast15,
 *     row_bcast31) ending in lane 63 and read back as a scalar: no LDS round trips in the loop.
 *   - Children are written to LDS fire-and-forget; a node's leaf count is the sum of its children's, read when
 *     the node is made (children are always made in an earlier round).
 * ==================================================================================== */
/* Wave minimum: six v_min_u32 with a DPP source operand (the compiler turns update_dpp + min into
 * mov, mov_dpp, min - three instructions per step; the merge loop runs two of these reductions
 * per round and is VALU bound once enough tree waves are resident).  s_nop 1 = the two wait
 * states a DPP read needs after the VALU write of its source.  Rows not named by row_mask keep
 * their value, the result is complete in lane 63. */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* Bitonic sort of the wave's 64 R keys k[0..R-1] (R = 1, 2, 4), ascending by sorted position lane * R + r
 * (blocked: for R >= 2 the keys at positions 2p and 2p + 1 end up in one lane).  Partners less than R
 * positions apart are registers of the same lane, all others the same register of lane ^ (distance / R):
 * DPP / ds_swizzle exchanges (util.hpp), no LDS round trip but for distance 32.  A tree round sorts only as
 * many registers as its live keys need: 36 compare-exchange stages on four registers for 256 keys, 28 on two
 * for 128, 21 on one for 64 - the later rounds of a block have a few dozen keys left. */
template <int R>
__device__ __forceinline__ void wave_sort_r(uint32_t (&k)[4])
{
    const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
    for (uint32_t kk = 2; kk <= 64u * R; kk <<= 1) {
#pragma unroll
        for (uint32_t jj = kk >> 1; jj >= 1; jj >>= 1) {
            if (jj >= (uint32_t)R) {
                const bool up = ((lane * (uint32_t)R) & kk) == 0u;        /* this block sorts ascending */
                const bool lower = (lane & (jj / R)) == 0u;               /* I hold the pair's lower position */
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t d = jj / R;
                    const uint32_t o = d == 1 ? wave_xor_u32<1>(k[r]) : d == 2 ? wave_xor_u32<2>(k[r]) : d == 4 ? wave_xor_u32<4>(k[r])
                                     : d == 8 ? wave_xor_u32<8>(k[r]) : d == 16 ? wave_xor_u32<16>(k[r]) : wave_xor_u32<32>(k[r]);
                    k[r] = (lower == up) ? dmin(k[r], o) : dmax(k[r], o);
                }
            } else {
#pragma unroll
                for (uint32_t r = 0; r < (uint32_t)R; r++) {
                    if (r & jj) continue;
                    const bool up = ((lane * (uint32_t)R + r) & kk) == 0u;
                    const uint32_t lo = dmin(k[r], k[r | jj]), hi = dmax(k[r], k[r | jj]);
                    k[r] = up ? lo : hi;
                    k[r | jj] = up ? hi : lo;
                }
            }
        }
    }
}
__device__ __forceinline__ void wave_sort256(uint32_t (&k)[4]) { wave_sort_r<4>(k); }

/* Round 6: the last phase of that sort alone - a bitonic MERGE: keys that go down and then up along the sorted positions (followed by
 * KMAX) come out ascending in log2(64 R) stages instead of the sort's 21 / 28 / 36.  What a tree round leaves behind is such a sequence
 * when it is laid out for it (tree_fast_wave): the keys it did not pair are still sorted, and the nodes it made come out in key order. */
template <int R>
__device__ __forceinline__ void wave_merge_r(uint32_t (&k)[4])
{
    const uint32_t lane = (uint32_t)lane_id();
    constexpr int STAGES = R == 4 ? 8 : (R == 2 ? 7 : 6);              /* log2(64 R) */
#pragma unroll
    for (int st = STAGES - 1; st >= 0; st--) {
        const uint32_t jj = 1u << st;
        if (jj >= (uint32_t)R) {
            const bool lower = (lane & (jj / R)) == 0u;               /* I hold the pair's lower position */
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t d = jj / R;
                const uint32_t o = d == 1 ? wave_xor_u32<1>(k[r]) : d == 2 ? wave_xor_u32<2>(k[r]) : d == 4 ? wave_xor_u32<4>(k[r])
                                 : d == 8 ? wave_xor_u32<8>(k[r]) : d == 16 ? wave_xor_u32<16>(k[r]) : wave_xor_u32<32>(k[r]);
                k[r] = lower ? dmin(k[r], o) : dmax(k[r], o);
            }
        } else {
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)R; r++) {
                if (r & jj) continue;
                const uint32_t lo = dmin(k[r], k[r | jj]), hi = dmax(k[r], k[r | jj]);
                k[r] = lo;
                k[r | jj] = hi;
            }
        }
    }
}

struct TreeLds {                  /* 5 KiB: what bounds the tree waves a CU holds (they are latency bound): 32 per CU */
    uint64_t state[HUF_NSLOT];    /* per node: its path to an ancestor, see TREE_STATE below.  The entries of nodes that do
                                     not exist yet also serve as scratch for keys on their way into fewer registers: with
                                     `node` the next index, at most 512 - node keys are alive and 2 (512 - node) words free */
    uint16_t lcnt[HUF_NSLOT];     /* leaves below each slot */
};

/* A node's state while codes, depths and preorder positions are worked out: the path from the node up to
 * an ancestor `anc` - how many edges (depth), the turns taken on them (code, first turn in the highest bit
 * used), and how many entries the serialized tree has between the ancestor's and the node's (pos).  When a
 * node is made its two children get (anc = the node, depth 1, code 0 / 1, pos 1 / 4 * leaves of the left
 * child - a subtree with L leaves holds 4L - 1 entries and the right child follows the left subtree).
 * Paths are then doubled - state(n) := state(n) + state(anc(n)) - until every anc is the root: at most five
 * rounds for depth <= 32, whatever the order the lanes' updates land in (a node's state is ONE 64-bit LDS
 * word, read and written whole, and every state it can have IS a valid path).  The level-by-level sweep this
 * replaces cost one pass over all slots per tree level: ~2 400 instructions where this has ~600.
 * low word: anc (10 bits, TREE_ANC_ROOT = reached) | depth << 10 (6 bits) | pos << 16 (11 bits); high: code. */
#define TREE_ANC_ROOT 0x3ffu
__device__ __forceinline__ uint64_t tree_state(uint32_t anc, uint32_t depth, uint32_t pos, uint32_t code)
{
    return ((uint64_t)code << 32) | (uint64_t)(anc | (depth << 10) | (pos << 16));
}

/* Single-wave synchronisation: LDS operations of one wave are in order, so only the compiler and
 * the LDS counter have to be fenced.  (The fused kernel calls this after its other waves have
 * retired, so a workgroup barrier must not be used here.) */
#define TREE_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

#ifndef TREE_ROUND_MIN
#define TREE_ROUND_MIN 16u        /* items below a + b that make a sorted round cheaper than their single merges */
#endif

/* Executed by ONE wavefront; rate[j] = count of byte (lane + 64 j) in the block.  Returns the
 * encoded size of the block in bytes (every lane). */
__device__ __forceinline__ uint64_t tree_fast_wave(const uint32_t (&rate)[4], TreeLds &L, uint64_t blk,
                                               hufcode_t *__restrict__ codetab, int16_t *__restrict__ treebuf,
                                               HufBlockMeta *__restrict__ meta)
{
    uint64_t *s_state = L.state;
    uint16_t *s_lcnt = L.lcnt;
    const uint32_t KMAX = 0xffffffffu;
    const int lane = lane_id();

    /* one distinct byte: the tree is [256, s, -1, -1, -1] and the code of s is the single bit 0
     * (tree.c:410-413 on the first round) - no need for the general machinery */
    {
        const unsigned long long nz0 = __ballot(rate[0] != 0), nz1 = __ballot(rate[1] != 0);
        const unsigned long long nz2 = __ballot(rate[2] != 0), nz3 = __ballot(rate[3] != 0);
        if (__popcll(nz0) + __popcll(nz1) + __popcll(nz2) + __popcll(nz3) == 1) {
            const int j1 = nz0 ? 0 : (nz1 ? 1 : (nz2 ? 2 : 3));
            const unsigned long long m1 = nz0 | nz1 | nz2 | nz3;
            const int sym = __builtin_ctzll(m1) + 64 * j1;
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) cnt += rate[j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) cnt += wave_xor_any(cnt, o);
            /* pack_kernel never looks codes up for a 5-entry tree (all-zero payload), so the 2 KiB
             * code table of this block is not written */
            int16_t *tb1 = treebuf + blk * HUF_TREE_STRIDE;
            if (lane < 5) tb1[lane] = (lane == 0) ? (int16_t)256 : (lane == 1 ? (int16_t)sym : (int16_t)-1);
            HufBlockMeta mm;
            mm.tree_len = 5;
            mm.max_len = 1;
            mm.payload_bits = cnt;
            if (lane == 0) meta[blk] = mm;
            return encoded_block_bytes(mm);
        }
    }

    uint32_t k[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = lane + 64 * j;
        k[j] = rate[j] ? ((rate[j] << 9) | (uint32_t)(511 - slot)) : KMAX;
        s_lcnt[slot] = rate[j] ? 1 : 0;
    }

    /* The live keys sit in the first R registers of every lane (R = 4, 2, 1 for up to 256, 128, 64 keys; the
     * other registers hold KMAX), because a round's sort costs what its register count costs.  Keys are moved
     * together through 1 KiB of LDS: a key's compact
     * index is its rank among the live ones. */
    uint32_t *s_scratch = reinterpret_cast<uint32_t *>(s_state + HUF_NSYM);   /* (moves up with `node`, see TreeLds) */
    uint32_t live, R = 4;
    {
        const unsigned long long nz0 = __ballot(rate[0] != 0), nz1 = __ballot(rate[1] != 0);
        const unsigned long long nz2 = __ballot(rate[2] != 0), nz3 = __ballot(rate[3] != 0);
        const uint32_t c0 = (uint32_t)__popcll(nz0), c1 = (uint32_t)__popcll(nz1), c2 = (uint32_t)__popcll(nz2), c3 = (uint32_t)__popcll(nz3);
        live = c0 + c1 + c2 + c3;
        if (live <= 128u) {
            const unsigned long long below = (1ull << lane) - 1ull;
            if (rate[0]) s_scratch[(uint32_t)__popcll(nz0 & below)] = k[0];
            if (rate[1]) s_scratch[c0 + (uint32_t)__popcll(nz1 & below)] = k[1];
            if (rate[2]) s_scratch[c0 + c1 + (uint32_t)__popcll(nz2 & below)] = k[2];
            if (rate[3]) s_scratch[c0 + c1 + c2 + (uint32_t)__popcll(nz3 & below)] = k[3];
            TREE_WAVE_SYNC();
            R = live <= 64u ? 1u : 2u;
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t c = (uint32_t)lane * R + r;
                k[r] = (r < R && c < live) ? s_scratch[c] : KMAX;
            }
            TREE_WAVE_SYNC();
        }
    }

    int node = HUF_NSYM;
    int root = -1;
    bool sorted = false;                                           /* (uniform) the keys stand in key order along the sorted positions lane * R + r */
    for (;;) {
        uint32_t a, b;
        uint32_t t[4];
        if (sorted) {
            /* (round 6) the two smallest are the first two: no reduction over the wave */
            a = wave_lane_u32(k[0], 0);
            b = R == 1u ? wave_lane_u32(k[0], 1) : wave_lane_u32(k[1], 0);
#pragma unroll
            for (int j = 0; j < 4; j++) t[j] = (k[j] == a) ? KMAX : k[j];
        } else {
            a = wave_min_u32(dmin(dmin(k[0], k[1]), dmin(k[2], k[3])));
            if (a != KMAX) {
#pragma unroll
                for (int j = 0; j < 4; j++) t[j] = (k[j] == a) ? KMAX : k[j];
                b = wave_min_u32(dmin(dmin(t[0], t[1]), dmin(t[2], t[3])));
            } else {
                b = KMAX;
#pragma unroll
                for (int j = 0; j < 4; j++) t[j] = k[j];
            }
        }
        if (a == KMAX) { root = node - 1; break; }                 /* tree.c:355-358 */
        const int i1 = 511 - (int)(a & 511u);
        if (b == KMAX) {                                           /* tree.c:410-413: left-only wrap root */
            if (lane == 0) {
                s_state[i1] = tree_state((uint32_t)node, 1u, 1u, 0u);
                s_lcnt[node] = s_lcnt[i1];
            }
            root = node;
            node++;
            break;
        }
        /* Every item whose rate is below a + b precedes every node still to be made (a node's rate
         * is at least that), so all of them pair up in key order no matter what is merged first:
         * when there are enough, sort the wave's keys once and merge all those pairs in one step
         * (pair p = sorted positions 2p, 2p + 1 -> node + p: the sequential order of tree.c:355-407).
         * Typical blocks take 5-7 such rounds instead of 255 merges with two wave minima each; a block
         * that never offers enough items at once (Fibonacci-like counts) falls through to the single
         * merge below.  "Enough" is what makes the sort cheaper than the single merges it replaces. */
        {
            const uint32_t thr = (a >> 9) + (b >> 9);
            uint32_t sel = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) sel += (uint32_t)__popcll(__ballot((k[j] >> 9) < thr));
            const uint32_t round_min = R == 4u ? TREE_ROUND_MIN : (R == 2u ? TREE_ROUND_MIN / 2u : TREE_ROUND_MIN / 4u);
            if (sel >= round_min) {
                const uint32_t pairs = sel >> 1;
                if (R == 4u) {
                    if (!sorted) wave_sort_r<4>(k);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t p = 2u * (uint32_t)lane + (uint32_t)h;
                        if (p < pairs) {
                            const uint32_t x = k[2 * h], y = k[2 * h + 1];
                            const int n = node + (int)p;
                            const int xi = 511 - (int)(x & 511u), yi = 511 - (int)(y & 511u);
                            const uint32_t lx = s_lcnt[xi];                          /* children were made in earlier rounds */
                            s_state[xi] = tree_state((uint32_t)n, 1u, 1u, 0u);
                            s_state[yi] = tree_state((uint32_t)n, 1u, 4u * lx, 1u);
                            s_lcnt[n] = (uint16_t)(lx + s_lcnt[yi]);
                            k[2 * h] = (((x >> 9) + (y >> 9)) << 9) | (uint32_t)(511 - n);
                            k[2 * h + 1] = KMAX;
                        }
                    }
                } else if (R == 2u) {
                    if (!sorted) wave_sort_r<2>(k);
                    const uint32_t p = (uint32_t)lane;
                    if (p < pairs) {
                        const uint32_t x = k[0], y = k[1];
                        const int n = node + (int)p;
                        const int xi = 511 - (int)(x & 511u), yi = 511 - (int)(y & 511u);
                        const uint32_t lx = s_lcnt[xi];
                        s_state[xi] = tree_state((uint32_t)n, 1u, 1u, 0u);
                        s_state[yi] = tree_state((uint32_t)n, 1u, 4u * lx, 1u);
                        s_lcnt[n] = (uint16_t)(lx + s_lcnt[yi]);
                        k[0] = (((x >> 9) + (y >> 9)) << 9) | (uint32_t)(511 - n);
                        k[1] = KMAX;
                    }
                } else {
                    if (!sorted) wave_sort_r<1>(k);
                    const uint32_t p = (uint32_t)lane >> 1;
                    const uint32_t o = wave_xor_u32<1>(k[0]);              /* the pair's other key */
                    if (p < pairs) {
                        if ((lane & 1) == 0) {
                            const uint32_t x = k[0], y = o;
                            const int n = node + (int)p;
                            const int xi = 511 - (int)(x & 511u), yi = 511 - (int)(y & 511u);
                            const uint32_t lx = s_lcnt[xi];
                            s_state[xi] = tree_state((uint32_t)n, 1u, 1u, 0u);
                            s_state[yi] = tree_state((uint32_t)n, 1u, 4u * lx, 1u);
                            s_lcnt[n] = (uint16_t)(lx + s_lcnt[yi]);
                            k[0] = (((x >> 9) + (y >> 9)) << 9) | (uint32_t)(511 - n);
                        } else {
                            k[0] = KMAX;
                        }
                    }
                }
                node += (int)pairs;
                const uint32_t was = live;                           /* keys at positions < was are live or just paired */
                live -= pairs;
                const uint32_t nr = live <= 64u ? 1u : (live <= 128u ? 2u : 4u);
                s_scratch = reinterpret_cast<uint32_t *>(s_state + node);
                {
                    /* Round 6: the next round does not sort again.  The keys that were NOT paired still stand in key order and the new
                     * nodes only have to be put among them: unpaired keys downwards from position 0, nodes upwards behind them - ONE
                     * bitonic sequence through LDS, in as many registers as the keys need - merged in log2(64 nr) stages (8 / 7 / 6
                     * where the sort takes 36 / 28 / 21).  The nodes' sums rise with p, but EQUAL sums come in falling key order (the
                     * later node has the larger index): while more than 64 keys are left - the rounds of the dear sorts, 250 and 650
                     * instructions, in which a Zipf-like block pairs a few dozen items of nearly equal rates - nodes that are not in
                     * order are sorted on their own first (64 keys in one register: 21 stages); the rounds on one register have no
                     * ties to speak of and take their chance.  The order is CHECKED after the merge (one compare with the next position): keys that are not in order
                     * are sorted by the next round as every round did before. */
                    const uint32_t unp = live - pairs;               /* keys that were not paired */
#pragma unroll
                    for (uint32_t r = 0; r < 4; r++) {
                        const uint32_t q = (uint32_t)lane * R + r;
                        if (r < R && q < was && k[r] != KMAX) s_scratch[q < 2u * pairs ? unp + (q >> 1) : was - 1u - q] = k[r];
                    }
                    TREE_WAVE_SYNC();
                    if (nr >= 2u) {
                        /* the nodes on their own: in order already (no two equal sums - one compare with the next node says), or
                         * sorted in one register (up to 64 of them) or two */
                        const uint32_t p0 = 2u * (uint32_t)lane, p1 = p0 + 1u;
                        uint32_t nd[4] = {KMAX, KMAX, KMAX, KMAX};
                        if (pairs <= 64u) {
                            nd[0] = (uint32_t)lane < pairs ? s_scratch[unp + (uint32_t)lane] : KMAX;
                            const uint32_t nf = (uint32_t)__builtin_amdgcn_update_dpp((int)KMAX, (int)nd[0], 0x130, 0xf, 0xf, false);   /* wave_shl:1, lane 63 <- KMAX */
                            if (__ballot(nd[0] > nf) != 0ull) {
                                wave_sort_r<1>(nd);
                                TREE_WAVE_SYNC();
                                if ((uint32_t)lane < pairs) s_scratch[unp + (uint32_t)lane] = nd[0];
                                TREE_WAVE_SYNC();
                            }
                        } else {
                            nd[0] = p0 < pairs ? s_scratch[unp + p0] : KMAX;
                            nd[1] = p1 < pairs ? s_scratch[unp + p1] : KMAX;
                            wave_sort_r<2>(nd);
                            TREE_WAVE_SYNC();
                            if (p0 < pairs) s_scratch[unp + p0] = nd[0];
                            if (p1 < pairs) s_scratch[unp + p1] = nd[1];
                            TREE_WAVE_SYNC();
                        }
                    }
                    R = nr;
#pragma unroll
                    for (uint32_t r = 0; r < 4; r++) {
                        const uint32_t c = (uint32_t)lane * R + r;
                        k[r] = (r < R && c < live) ? s_scratch[c] : KMAX;
                    }
                    TREE_WAVE_SYNC();
                    bool in_order;
                    if (R == 4u) {
                        wave_merge_r<4>(k);
                        const uint32_t nf = (uint32_t)__builtin_amdgcn_update_dpp((int)KMAX, (int)k[0], 0x130, 0xf, 0xf, false);   /* wave_shl:1, lane 63 <- KMAX */
                        in_order = k[0] <= k[1] && k[1] <= k[2] && k[2] <= k[3] && k[3] <= nf;
                    } else if (R == 2u) {
                        wave_merge_r<2>(k);
                        const uint32_t nf = (uint32_t)__builtin_amdgcn_update_dpp((int)KMAX, (int)k[0], 0x130, 0xf, 0xf, false);
                        in_order = k[0] <= k[1] && k[1] <= nf;
                    } else {
                        wave_merge_r<1>(k);
                        const uint32_t nf = (uint32_t)__builtin_amdgcn_update_dpp((int)KMAX, (int)k[0], 0x130, 0xf, 0xf, false);
                        in_order = k[0] <= nf;
                    }
                    sorted = __ballot(!in_order) == 0ull;
                }
                continue;
            }
        }
        const int i2 = 511 - (int)(b & 511u);
        const uint32_t nk = (((a >> 9) + (b >> 9)) << 9) | (uint32_t)(511 - node);   /* tree.c:407 */
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = (k[j] == a) ? nk : ((t[j] == b) ? KMAX : t[j]);
        sorted = false;                                            /* (the new key stands where the smaller of the two stood) */
        if (lane == 0) {
            const uint32_t lx = s_lcnt[i1];                        /* tree.c:390-404 */
            s_state[i1] = tree_state((uint32_t)node, 1u, 1u, 0u);
            s_state[i2] = tree_state((uint32_t)node, 1u, 4u * lx, 1u);
            s_lcnt[node] = (uint16_t)(lx + s_lcnt[i2]);
        }
        node++;
        live--;
    }
    if (lane == 0 && root >= 0) s_state[root] = tree_state(TREE_ANC_ROOT, 0u, 0u, 0u);
    TREE_WAVE_SYNC();
    const int nodes = node;
    const int nleaves = (root >= 0) ? (int)s_lcnt[root] : 0;
    const int tree_len = (root >= 0) ? 4 * nleaves + 1 : 1;

    /* paths doubled until every node has reached the root (TREE_STATE above); a lane keeps the states of its
     * eight slots (leaves lane + 64 j, nodes 256 + lane + 64 j) in registers and publishes every update */
    uint32_t w0[8], w1[8];
    bool have[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int slot = lane + 64 * j;
        have[j] = (j < 4) ? (rate[j & 3] != 0) : (slot < nodes);
        const uint64_t st = have[j] ? s_state[slot] : tree_state(TREE_ANC_ROOT, 0u, 0u, 0u);
        w0[j] = (uint32_t)st;
        w1[j] = (uint32_t)(st >> 32);
    }
#pragma unroll 1
    for (int round = 0; round < 6; round++) {
        bool pending = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t anc = w0[j] & 0x3ffu;
            if (anc != TREE_ANC_ROOT) {
                const uint64_t up = s_state[anc];
                const uint32_t u0 = (uint32_t)up, u1 = (uint32_t)(up >> 32);
                w1[j] |= u1 << ((w0[j] >> 10) & 63u);               /* my turns follow the ancestor's (own depth <= 31 here) */
                w0[j] = ((w0[j] & ~0x3ffu) + (u0 & ~0x3ffu)) | (u0 & 0x3ffu);
                s_state[lane + 64 * j] = ((uint64_t)w1[j] << 32) | w0[j];
                pending = pending || (u0 & 0x3ffu) != TREE_ANC_ROOT;
            }
        }
        if (!__any(pending)) break;
    }

    /* The serialized tree goes straight to HBM: a node at preorder position p writes its index there, a leaf
     * also the two -1 of its absent children behind it, and the wrap root (the only node without a right
     * child) the -1 where that child would start, the last entry - together exactly the 4k+1 entries, each
     * written once. */
    int16_t *tb = treebuf + blk * HUF_TREE_STRIDE;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (have[j]) {
            const uint32_t p = w0[j] >> 16;
            tb[p] = (int16_t)(lane + 64 * j);
            if (j < 4) { tb[p + 1] = -1; tb[p + 2] = -1; }
        }
    }
    if (lane == 0 && root >= 0) tb[tree_len - 1] = -1;

    uint64_t bits = 0;
    uint32_t maxlen = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int slot = lane + 64 * j;
        hufcode_t e = 0;
        if (rate[j]) {
            const uint32_t len = (w0[j] >> 10) & 63u;
            /* the path's turns, root first: `len` bits, the first turn in bit len - 1 */
            e = ((hufcode_t)w1[j] << 8) | (hufcode_t)len;
            bits += (uint64_t)rate[j] * len;
            maxlen = dmax(maxlen, len);
        }
        codetab[blk * HUF_NSYM + slot] = e;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        bits += shfl_xor_u64(bits, o);
        maxlen = dmax(maxlen, wave_xor_any(maxlen, o));
    }
    HufBlockMeta mm;
    mm.tree_len = (uint32_t)tree_len;
    mm.max_len = maxlen;
    mm.payload_bits = bits;
    if (lane == 0) meta[blk] = mm;
    return encoded_block_bytes(mm);
}

}  // namespace hufgpu
