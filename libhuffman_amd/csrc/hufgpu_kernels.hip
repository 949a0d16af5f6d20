/*
 * hufgpu_kernels.hip - CDNA4 (gfx950) kernels of the Huffman block codec.
 *
 * One libhuffman block (config->blocksize input bytes, src/encoder.c:288-293) is the unit of
 * parallelism everywhere: blocks never exchange data, so every kernel maps blocks to
 * workgroups and the grid is simply the block count.
 *
 *   encode:  hist_tree_kernel -> pack_kernel      (blocks >= 4 MiB: hist256 -> tree -> scan_sizes -> pack)
 *   decode:  decode_prepare_kernel -> decode_kernel
 *            (block index known), or decode_chain_kernel (raw stream, blocks in order)
 *
 * Wave size is 64 throughout (hard-coded, gfx950 only).  All arithmetic is integer.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hufgpu_common.h"

namespace hufgpu {

/* ======================================================================================
 * small helpers
 * ==================================================================================== */
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

/* Values every lane of the wave holds identically: tell the compiler, so they live in SGPRs. */
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v);
}

template <typename T>
__device__ __forceinline__ T dmin(T a, T b) { return a < b ? a : b; }
template <typename T>
__device__ __forceinline__ T dmax(T a, T b) { return a > b ? a : b; }

/* block bytes = 10 + 2*tree_len + ceil(payload_bits/8)   (src/encoder.c:325-348,123-128) */
__device__ __forceinline__ uint64_t encoded_block_bytes(const HufBlockMeta &m)
{
    return (uint64_t)HUF_HEADER_FIXED + 2ull * m.tree_len + ((m.payload_bits + 7) >> 3);
}

/* streaming accesses: the input of a pass is read once and its output written once */
__device__ __forceinline__ uint4 load_stream16(const uint4 *p)
{
#ifndef HUF_NO_NT_LOAD     /* measured: histogram of 1 GiB 0.202 -> 0.169 ms, pack 0.043 -> 0.030 ms */
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void store_stream16(uint4 *p, uint4 v)
{
#ifndef HUF_NO_NT_STORE    /* measured: one-symbol decode (a fill) 0.260 -> 0.204 ms per GiB */
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p));
#else
    *p = v;
#endif
}
/* The compressed stream is written with the default policy: it is what a decode that follows
 * reads, and a stream that fits the 256 MiB Infinity Cache is then served from there (measured on
 * config 2: decode 0.250 -> 0.21 ms when the 128 MiB stream is still cached). */
__device__ __forceinline__ void store_pack16(uint4 *p, uint4 v)
{
#ifdef HUF_PACK_NT_STORE
    store_stream16(p, v);
#else
    *p = v;
#endif
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_xor((int)lo, mask);
    hi = (uint32_t)__shfl_xor((int)hi, mask);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shfl_xor_key(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask); }
__device__ __forceinline__ uint64_t shfl_xor_key(uint64_t v, int mask) { return shfl_xor_u64(v, mask); }

/* Exclusive prefix sum over the workgroup (THREADS a multiple of 64). s_part needs THREADS/64
 * words. Returns this thread's exclusive prefix; `total` is the workgroup sum. */
template <int THREADS, typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *s_part, T &total)
{
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t;
        if constexpr (sizeof(T) == 8) {
            uint32_t lo = (uint32_t)inc, hi = (uint32_t)((uint64_t)inc >> 32);
            lo = (uint32_t)__shfl_up((int)lo, o);
            hi = (uint32_t)__shfl_up((int)hi, o);
            t = (T)(((uint64_t)hi << 32) | lo);
        } else {
            t = (T)__shfl_up((int)inc, o);
        }
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) {
        T x = s_part[i];
        if (i < wave) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

/* ======================================================================================
 * hist256 - replaces huf_histogram_populate (src/histogram.c:73-103, iota = 1).
 *
 * One workgroup per block; every wavefront owns a private 256-bin histogram in LDS so that
 * LDS atomics of different waves never collide; the wave copies are summed at the end.
 * Each lane reads 16 contiguous bytes per step (a wave reads 1 KiB, fully coalesced).
 * Runs of one byte value are folded before touching LDS: a 16-byte chunk of one value costs
 * one atomic, and a whole wave-step of one value costs one atomic for the wave - that is the
 * common case on BASELINE config 2 (all 0x41), where per-byte atomics would serialise 64-way.
 * ==================================================================================== */
#ifndef HIST_COPIES
#define HIST_COPIES 4
#endif

/* `one` is what a single occurrence adds: 1, or 1 << 16 when two 16-bit counters share a word */
__device__ __forceinline__ void hist_add_bytes(uint32_t *h, uint32_t w, uint32_t one)
{
    atomicAdd(&h[w & 0xffu], one);
    atomicAdd(&h[(w >> 8) & 0xffu], one);
    atomicAdd(&h[(w >> 16) & 0xffu], one);
    atomicAdd(&h[w >> 24], one);
}

__device__ __forceinline__ void hist_add_chunk(uint32_t *h, uint4 v, uint32_t one = 1u)
{
    const uint32_t b = v.x & 0xffu;
    const uint32_t rep = b * 0x01010101u;
    const bool uni = (v.x == rep) & (v.y == rep) & (v.z == rep) & (v.w == rep);
    const unsigned long long act = __ballot(1);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    const unsigned long long same = __ballot(uni && b == b0);
    if (same == act) {                       /* the whole wave step holds one byte value */
        if ((unsigned)lane_id() == (unsigned)__builtin_ctzll(act))
            atomicAdd(&h[b0], one * 16u * (uint32_t)__popcll(act));
        return;
    }
    if (uni) {
        atomicAdd(&h[b], one * 16u);
        return;
    }
    hist_add_bytes(h, v.x, one);
    hist_add_bytes(h, v.y, one);
    hist_add_bytes(h, v.z, one);
    hist_add_bytes(h, v.w, one);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void hist256_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                          uint64_t blocksize, uint32_t *__restrict__ hist)
{
    /* HIST_COPIES private histograms per wavefront, selected by lane: hot symbols of skewed data
     * then collide HIST_COPIES times less inside one ds_add (SQ_LDS_BANK_CONFLICT was 82 % of the
     * LDS cycles with one copy on Zipf data) */
    constexpr int WAVES = THREADS / 64;
    constexpr int COPIES = WAVES * HIST_COPIES;
    __shared__ uint32_t s_hist[COPIES * HUF_NSYM];

    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    const int tid = (int)threadIdx.x;

    for (int i = tid; i < COPIES * HUF_NSYM; i += THREADS) s_hist[i] = 0;
    __syncthreads();

    uint32_t *mine = s_hist + ((tid >> 6) * HIST_COPIES + (tid & (HIST_COPIES - 1))) * HUF_NSYM;
    const uint8_t *p = in + base;
    const uint64_t head = dmin<uint64_t>(len, (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    if ((uint64_t)tid < head) atomicAdd(&mine[p[tid]], 1u);

    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    uint64_t i = (uint64_t)tid;
    for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {           /* four loads in flight per lane */
        const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                    v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
        hist_add_chunk(mine, v0);
        hist_add_chunk(mine, v1);
        hist_add_chunk(mine, v2);
        hist_add_chunk(mine, v3);
    }
    for (; i < nvec; i += THREADS) hist_add_chunk(mine, load_stream16(q + i));

    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < len) atomicAdd(&mine[p[tail0 + tid]], 1u);   /* < 16 bytes */
    __syncthreads();

    for (int b = tid; b < HUF_NSYM; b += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < COPIES; w++) sum += s_hist[w * HUF_NSYM + b];
        hist[blk * HUF_NSYM + b] = sum;
    }
}

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
template <typename K>
__global__ __launch_bounds__(64) void tree_kernel(const uint32_t *__restrict__ hist, uint64_t n,
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
    const uint32_t *h = hist + blk * HUF_NSYM;

    K key[8];
    uint32_t rate[4];
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
        maxlen = dmax(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
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
 *   - Children are written to LDS fire-and-forget; leaf counts are derived after the loop.
 * ==================================================================================== */
/* Wave minimum: six v_min_u32 with a DPP source operand (the compiler turns update_dpp + min into
 * mov, mov_dpp, min - three instructions per step; the merge loop runs two of these reductions
 * per round and is VALU bound once enough tree waves are resident).  s_nop 1 = the two wait
 * states a DPP read needs after the VALU write of its source.  Rows not named by row_mask keep
 * their value, the result is complete in lane 63. */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#ifdef TREE_DPP_BUILTIN
    auto step = [](uint32_t x, int sel) {
        uint32_t o;
        switch (sel) {
        case 0: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false); break;
        case 1: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false); break;
        case 2: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false); break;
        case 3: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false); break;
        case 4: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xa, 0xf, false); break;
        default: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xc, 0xf, false); break;
        }
        return dmin(x, o);
    };
    for (int k = 0; k < 6; k++) v = step(v, k);
#else
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
#endif
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct TreeLds {                  /* 7 KiB: what bounds the tree waves a CU holds (they are latency bound) */
    uint32_t code[HUF_NSLOT];     /* blocks shorter than 2^22 bytes: depth <= 32 */
    int16_t left[HUF_NSLOT];
    int16_t right[HUF_NSLOT];
    uint16_t lcnt[HUF_NSLOT];     /* leaves below each slot, 0xffff = not known yet */
    uint16_t depth[HUF_NSLOT];    /* 0xffff = not reached */
    uint16_t pos[HUF_NSLOT];      /* preorder position */
};

/* Single-wave synchronisation: LDS operations of one wave are in order, so only the compiler and
 * the LDS counter have to be fenced.  (The fused kernel calls this after its other waves have
 * retired, so a workgroup barrier must not be used here.) */
#define TREE_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

/* Executed by ONE wavefront; rate[j] = count of byte (lane + 64 j) in the block.  Returns the
 * encoded size of the block in bytes (every lane). */
__device__ __forceinline__ uint64_t tree_fast_wave(const uint32_t (&rate)[4], TreeLds &L, uint64_t blk,
                                               hufcode_t *__restrict__ codetab, int16_t *__restrict__ treebuf,
                                               HufBlockMeta *__restrict__ meta)
{
    int16_t *s_left = L.left, *s_right = L.right;
    uint16_t *s_lcnt = L.lcnt, *s_depth = L.depth, *s_pos = L.pos;
    uint32_t *s_code = L.code;
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
            for (int o = 1; o < 64; o <<= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
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
        s_lcnt[256 + slot] = 0xffffu;
        s_depth[slot] = 0xffffu;
        s_depth[256 + slot] = 0xffffu;
        s_left[256 + slot] = -1;
        s_right[256 + slot] = -1;
    }

    int node = HUF_NSYM;
    int root = -1;
    for (;;) {
        const uint32_t a = wave_min_u32(dmin(dmin(k[0], k[1]), dmin(k[2], k[3])));
        if (a == KMAX) { root = node - 1; break; }                 /* tree.c:355-358 */
        uint32_t t[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = (k[j] == a) ? KMAX : k[j];
        const uint32_t b = wave_min_u32(dmin(dmin(t[0], t[1]), dmin(t[2], t[3])));
        const int i1 = 511 - (int)(a & 511u);
        if (b == KMAX) {                                           /* tree.c:410-413: left-only wrap root */
            if (lane == 0) s_left[node] = (int16_t)i1;
            root = node;
            node++;
            break;
        }
        const int i2 = 511 - (int)(b & 511u);
        const uint32_t nk = (((a >> 9) + (b >> 9)) << 9) | (uint32_t)(511 - node);   /* tree.c:407 */
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = (k[j] == a) ? nk : ((t[j] == b) ? KMAX : t[j]);
        if (lane == 0) {
            s_left[node] = (int16_t)i1;                            /* tree.c:390-404 */
            s_right[node] = (int16_t)i2;
        }
        node++;
    }
    TREE_WAVE_SYNC();
    const int nodes = node;

    /* leaves below every internal node: children always have smaller indices, so a few rounds of
     * "both children known -> sum" settle it (one tree level per round) */
    for (int round = 0; round < HUF_NSLOT; round++) {
        bool pending = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int slot = 256 + lane + 64 * j;
            if (slot < nodes && s_lcnt[slot] == 0xffffu) {
                const int l = s_left[slot], r = s_right[slot];
                const uint32_t cl = s_lcnt[l];
                const uint32_t cr = (r >= 0) ? (uint32_t)s_lcnt[r] : 0u;
                if (cl != 0xffffu && cr != 0xffffu) s_lcnt[slot] = (uint16_t)(cl + cr);
                else pending = true;
            }
        }
        TREE_WAVE_SYNC();
        if (!__any(pending)) break;
    }
    const int nleaves = (root >= 0) ? (int)s_lcnt[root] : 0;
    const int tree_len = (root >= 0) ? 4 * nleaves + 1 : 1;

    if (lane == 0 && root >= 0) {
        s_depth[root] = 0;
        s_code[root] = 0;
        s_pos[root] = 0;
    }
    TREE_WAVE_SYNC();

    /* level sweep: codes, depths, preorder positions (see tree_kernel).  The serialized tree goes
     * straight to HBM: a node at position p writes its index there, a leaf also the two -1 of its
     * absent children behind it, and a node without a right child (the wrap root) the -1 where
     * that child would start - together exactly the 4k+1 entries, each written once. */
    int16_t *tb = treebuf + blk * HUF_TREE_STRIDE;
    for (int d = 0; d < HUF_NSLOT; d++) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int slot = 256 + lane + 64 * j;
            if (slot < nodes && s_depth[slot] == (uint16_t)d) {
                any = true;
                const int l = s_left[slot], r = s_right[slot];
                const uint32_t c = s_code[slot];
                const int p = s_pos[slot];
                tb[p] = (int16_t)slot;
                s_depth[l] = (uint16_t)(d + 1);
                s_code[l] = c << 1;
                s_pos[l] = (uint16_t)(p + 1);
                if (l < HUF_NSYM) { tb[p + 1] = (int16_t)l; tb[p + 2] = -1; tb[p + 3] = -1; }
                const int pr = p + 4 * (int)s_lcnt[l];
                if (r >= 0) {
                    s_depth[r] = (uint16_t)(d + 1);
                    s_code[r] = (c << 1) | 1u;
                    s_pos[r] = (uint16_t)pr;
                    if (r < HUF_NSYM) { tb[pr] = (int16_t)r; tb[pr + 1] = -1; tb[pr + 2] = -1; }
                } else {
                    tb[pr] = -1;
                }
            }
        }
        TREE_WAVE_SYNC();
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
            e = ((hufcode_t)s_code[slot] << 8) | (hufcode_t)len;
            bits += (uint64_t)rate[j] * len;
            maxlen = dmax(maxlen, len);
        }
        codetab[blk * HUF_NSYM + slot] = e;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        bits += shfl_xor_u64(bits, o);
        maxlen = dmax(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
    }
    HufBlockMeta mm;
    mm.tree_len = (uint32_t)tree_len;
    mm.max_len = maxlen;
    mm.payload_bits = bits;
    if (lane == 0) meta[blk] = mm;
    return encoded_block_bytes(mm);
}


/* ======================================================================================
 * scan_sizes_kernel - byte offset of every block header in the output stream.
 * block bytes = 10 + 2*tree_len + ceil(payload_bits/8)   (src/encoder.c:325-348,123-128)
 * Single workgroup; offsets[nblocks] = stream length.
 * ==================================================================================== */
/* Exclusive prefix sum of f(i), i < n, by ONE workgroup (n is the block count: 16 384 per GiB).
 * A chunk is THREADS * 16 elements.  Wave w owns a contiguous run of 1 024 of them, swept in
 * SCAN_PASSES passes in which a lane owns SCAN_LANE consecutive elements.  Every f() of a chunk is
 * evaluated before the first use, so a chunk costs ONE memory round trip (two when f chases a
 * pointer), then SCAN_PASSES independent wave scans, one barrier for the wave totals, and 32-byte
 * stores - the whole of 16 384 elements in a few microseconds; it sits between two kernels that
 * cannot overlap with it. */
#define SCAN_LANE 4
#define SCAN_PASSES 4

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = (uint64_t)__shfl_up((unsigned long long)v, d);
        if (lane_id() >= d) v += t;
    }
    return v;
}

template <int THREADS, typename F>
__device__ __forceinline__ uint64_t chunked_excl_scan(uint64_t n, uint64_t *__restrict__ out, F f)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int PASS_ELEMS = 64 * SCAN_LANE;
    constexpr int WAVE_ELEMS = PASS_ELEMS * SCAN_PASSES;
    constexpr int CH = WAVES * WAVE_ELEMS;
    __shared__ uint64_t s_wave[2][WAVES];          /* double buffered: one barrier per chunk */
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    const bool vec = (((uintptr_t)out) & 15u) == 0;
    uint64_t carry = 0;
    int buf = 0;
    for (uint64_t base = 0; base < n; base += CH, buf ^= 1) {
        const uint64_t first = base + (uint64_t)(w * WAVE_ELEMS + lane * SCAN_LANE);
        uint64_t v[SCAN_PASSES][SCAN_LANE];
#pragma unroll
        for (int p = 0; p < SCAN_PASSES; p++)
#pragma unroll
            for (int k = 0; k < SCAN_LANE; k++) {
                const uint64_t i = first + (uint64_t)(p * PASS_ELEMS + k);
                v[p][k] = (i < n) ? f(i) : 0ull;
            }
        uint64_t incl[SCAN_PASSES], own[SCAN_PASSES];
#pragma unroll
        for (int p = 0; p < SCAN_PASSES; p++) {
            own[p] = 0;
#pragma unroll
            for (int k = 0; k < SCAN_LANE; k++) own[p] += v[p][k];
            incl[p] = wave_incl_scan_u64(own[p]);
        }
        uint64_t before[SCAN_PASSES], wsum = 0;
#pragma unroll
        for (int p = 0; p < SCAN_PASSES; p++) {
            before[p] = wsum;
            wsum += (uint64_t)__shfl((unsigned long long)incl[p], 63);
        }
        if (lane == 0) s_wave[buf][w] = wsum;
        __syncthreads();
        uint64_t wpre = 0, total = 0;
#pragma unroll
        for (int x = 0; x < WAVES; x++) {
            const uint64_t t = s_wave[buf][x];
            if (x < w) wpre += t;
            total += t;
        }
#pragma unroll
        for (int p = 0; p < SCAN_PASSES; p++) {
            const uint64_t i0 = first + (uint64_t)(p * PASS_ELEMS);
            uint64_t run = carry + wpre + before[p] + incl[p] - own[p];
            uint64_t r[SCAN_LANE];
#pragma unroll
            for (int k = 0; k < SCAN_LANE; k++) {
                r[k] = run;
                run += v[p][k];
            }
            if (vec && i0 + SCAN_LANE <= n) {
#pragma unroll
                for (int k = 0; k < SCAN_LANE; k += 2)
                    *reinterpret_cast<uint4 *>(out + i0 + k) =
                        make_uint4((uint32_t)r[k], (uint32_t)(r[k] >> 32), (uint32_t)r[k + 1], (uint32_t)(r[k + 1] >> 32));
            } else {
#pragma unroll
                for (int k = 0; k < SCAN_LANE; k++)
                    if (i0 + k < n) out[i0 + k] = r[k];
            }
        }
        carry += total;
    }
    return carry;
}

/* --------------------------------------------------------------------------------------
 * Two-level prefix sums without a launch of their own.  A one-workgroup scan between two big
 * kernels costs ~20 us of an otherwise ~450 us step (config 2), nearly all of it launch + drain.
 * Instead the kernel that produces the per-block values also sums them: blocks form groups of
 * SCAN_GROUP; whoever finishes LAST in a group (a ticket from an atomic counter - nobody waits)
 * scans the group (local[b] = sum of the group's earlier blocks, gsum[g] = group total), and
 * whoever finishes the last group scans the group totals (gprefix[g]).  The consumer kernel adds
 * gprefix[b / SCAN_GROUP] + local[b].  Counters are left at zero for the next launch.
 *
 * Ordering inside the producing kernel.  What one wave hands to another (vals, gsum, gmin) is
 * written and read with device-scope atomic stores / loads, which are performed at the coherence
 * point past the per-XCD L2s, and the writer waits for them (s_waitcnt vmcnt(0), handover_fence)
 * before it takes its ticket.  A device-scope __threadfence() would be correct too but on gfx950
 * it writes back and invalidates the whole L2 of the XCD: one per block made the fused
 * histogram kernel 6x slower (0.17 -> 1.02 ms per GiB).
 * ------------------------------------------------------------------------------------ */
#define SCAN_GROUP 256
#define SCAN_TICKET_STRIDE 64       /* one ticket counter per 256 bytes: neighbours in one line serialise in one L2 channel */

struct TwoLevel {
    uint64_t *vals;       /* [nblocks] the values, as handed over by their producers       */
    uint64_t *local;      /* [nblocks] exclusive sum inside the block's group              */
    uint64_t *gsum;       /* [ngroups] group totals                                        */
    uint64_t *gprefix;    /* [ngroups] exclusive sum of the group totals                   */
    uint32_t *gcount;     /* [ngroups * SCAN_TICKET_STRIDE] tickets, zero between launches */
    uint32_t *done;       /* [1] groups finished, zero between launches                    */
    uint64_t *total;      /* where the grand total goes (index[nblocks] / result word)     */
    uint64_t *total2;     /* optional second copy of the grand total                        */
    uint64_t *gmin;       /* optional [ngroups]: a minimum to combine along (first failing block) */
    uint64_t *min_out;    /* where that minimum goes                                       */
};

__device__ __forceinline__ void handover_store(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t handover_load(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* Every handover store of this wave has been performed (acknowledged at device scope) before
 * anything that follows is issued - in particular the ticket.  A workgroup-scope fence is NOT
 * enough: without threadgroup-split mode the compiler lowers it to s_waitcnt lgkmcnt(0) only,
 * and the ticket (another address, another L2 channel) can then overtake the value it
 * announces - tools/soak.py caught exactly that as one wrong block index in ~6 000 runs with
 * thousands of 64-byte blocks. */
__device__ __forceinline__ void handover_fence()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* Scan of the group totals by the wave that completed the last group.  The caller has stored
 * gsum[g] (and gmin[g]) of its group with handover_store(). */
__device__ __forceinline__ void two_level_finish(const TwoLevel &t, uint64_t ngroups)
{
    const int lane = lane_id();
    uint32_t k = 0;
    handover_fence();
    if (lane == 0) k = atomicAdd(t.done, 1u);
    k = uni32(k);
    if ((uint64_t)k != ngroups - 1) return;
    if (lane == 0) *t.done = 0;
    uint64_t carry = 0, low = ~0ull;
    for (uint64_t base = 0; base < ngroups; base += 64) {
        const uint64_t i = base + (uint64_t)lane;
        const uint64_t x = (i < ngroups) ? handover_load(t.gsum + i) : 0ull;
        const uint64_t incl = wave_incl_scan_u64(x);
        if (i < ngroups) t.gprefix[i] = carry + incl - x;
        carry += (uint64_t)__shfl((unsigned long long)incl, 63);
        if (t.gmin && i < ngroups) low = dmin(low, handover_load(t.gmin + i));
    }
    if (t.gmin) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) low = dmin(low, shfl_xor_u64(low, o));
    }
    if (lane == 0) {
        *t.total = carry;
        if (t.total2) *t.total2 = carry;
        if (t.gmin) *t.min_out = low;
    }
}

/* Called by ONE full wavefront with the value of its block. */
__device__ __forceinline__ void two_level_arrive(const TwoLevel &t, uint64_t b, uint64_t nblocks, uint64_t value)
{
    static_assert(SCAN_GROUP == 256, "a lane scans four blocks of its group");
    const int lane = lane_id();
    const uint64_t g = b / SCAN_GROUP;
    const uint64_t g0 = g * SCAN_GROUP;
    const uint32_t members = (uint32_t)dmin<uint64_t>(SCAN_GROUP, nblocks - g0);
    uint32_t k = 0;
    if (lane == 0) handover_store(t.vals + b, value);
    handover_fence();
    if (lane == 0) k = atomicAdd(&t.gcount[g * SCAN_TICKET_STRIDE], 1u);
    k = uni32(k);
    if (k != members - 1) return;
    if (lane == 0) t.gcount[g * SCAN_TICKET_STRIDE] = 0;
    uint64_t v[4], own = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = (uint32_t)(lane * 4 + j);
        v[j] = (i < members) ? handover_load(t.vals + g0 + i) : 0ull;
        own += v[j];
    }
    const uint64_t incl = wave_incl_scan_u64(own);
    uint64_t run = incl - own;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = (uint32_t)(lane * 4 + j);
        if (i < members) t.local[g0 + i] = run;
        run += v[j];
    }
    if (lane == 63) handover_store(t.gsum + g, incl);
    two_level_finish(t, (nblocks + SCAN_GROUP - 1) / SCAN_GROUP);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void scan_sizes_kernel(const HufBlockMeta *__restrict__ meta,
                                                             uint64_t nblocks, uint64_t *__restrict__ offsets)
{
    const uint64_t total = chunked_excl_scan<THREADS>(nblocks, offsets, [meta](uint64_t i) {
        return encoded_block_bytes(meta[i]);
    });
    if (threadIdx.x == 0) offsets[nblocks] = total;
}

#ifndef HT_COPIES
#define HT_COPIES 2
#endif

/* hist256 + tree in one launch: the block's byte counts never leave the CU.  All waves count;
 * then waves 1.. retire and wave 0 builds the tree in the LDS the histogram copies occupied.
 * The tree rounds are latency bound and the counting is memory bound, so on a CU the tree of
 * one block runs under the counting of the next ones.  The wave that finishes a group of blocks
 * last also prefix-sums the group's encoded sizes (no scan launch between this kernel and pack). */
#ifndef HTP_ARRAYS
#define HTP_ARRAYS 2        /* packed mode: 256-word arrays per wave, each holding two 16-bit copies (1: 0.79, 2: 0.71, 4: 0.86 ms on Zipf) */
#endif
#define HT_PACKED_MAX_BLOCK 131072u     /* a wave counts a quarter of the block: < 65 536 per 16-bit counter */

/* PACKED: the block is at most HT_PACKED_MAX_BLOCK bytes, so the private histograms use 16-bit
 * counters, two per word (lane parity picks the half): four copies per wave in the LDS of two
 * (hot symbols of skewed data collide half as often).  The totals are accumulated in place in the
 * last array, which lies behind the 7 KiB TreeLds: 8 KiB per workgroup = 20 resident groups per
 * CU instead of 13, and the latency-bound tree waves are what the kernel waits for on
 * multi-symbol data (uniform bytes 0.68 -> 0.60 ms, Zipf 0.77 -> 0.71 ms per GiB). */
template <int THREADS, bool PACKED>
__global__ __launch_bounds__(THREADS) void hist_tree_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                            uint64_t blocksize, hufcode_t *__restrict__ codetab,
                                                            int16_t *__restrict__ treebuf,
                                                            HufBlockMeta *__restrict__ meta, TwoLevel sizes)
{
    constexpr int WAVES = THREADS / 64;
    constexpr int COPIES = WAVES * (PACKED ? HTP_ARRAYS : HT_COPIES);   /* 256-word arrays */
    constexpr size_t HBYTES = (size_t)COPIES * HUF_NSYM * sizeof(uint32_t);
    constexpr size_t TBYTES = HUF_NSYM * sizeof(uint32_t);
    /* totals: the last histogram array when that lies behind the tree's area (summed in place:
     * a thread reads and writes only its own bin there), else right behind the tree's area */
    constexpr size_t TOT_OFF = (HBYTES >= sizeof(TreeLds) + TBYTES) ? HBYTES - TBYTES : sizeof(TreeLds);
    constexpr size_t UBYTES = (HBYTES > TOT_OFF + TBYTES) ? HBYTES : TOT_OFF + TBYTES;
    static_assert(TOT_OFF >= sizeof(TreeLds) && TOT_OFF % 16 == 0, "totals must survive the tree's initialisation");
    __shared__ __attribute__((aligned(16))) uint8_t s_union[UBYTES];
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_union);
    uint32_t *s_tot = reinterpret_cast<uint32_t *>(s_union + TOT_OFF);

    const uint64_t blk = blockIdx.x;
    const uint64_t base = blk * blocksize;
    const uint64_t len = dmin<uint64_t>(blocksize, n - base);
    const int tid = (int)threadIdx.x;

    for (int i = tid; i < COPIES * HUF_NSYM; i += THREADS) s_hist[i] = 0;
    __syncthreads();
    uint32_t *mine;
    uint32_t one = 1u;
    if (PACKED) {
        mine = s_hist + ((tid >> 6) * HTP_ARRAYS + ((tid >> 1) & (HTP_ARRAYS - 1))) * HUF_NSYM;
        one = (tid & 1) ? 0x10000u : 1u;
    } else {
        mine = s_hist + ((tid >> 6) * HT_COPIES + (tid & (HT_COPIES - 1))) * HUF_NSYM;
    }
    const uint8_t *p = in + base;
    const uint64_t head = dmin<uint64_t>(len, (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    if ((uint64_t)tid < head) atomicAdd(&mine[p[tid]], one);
    const uint4 *q = reinterpret_cast<const uint4 *>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    uint64_t i = (uint64_t)tid;
    for (; i + 3 * THREADS < nvec; i += 4 * THREADS) {           /* four loads in flight per lane */
        const uint4 v0 = load_stream16(q + i), v1 = load_stream16(q + i + THREADS),
                    v2 = load_stream16(q + i + 2 * THREADS), v3 = load_stream16(q + i + 3 * THREADS);
        hist_add_chunk(mine, v0, one);
        hist_add_chunk(mine, v1, one);
        hist_add_chunk(mine, v2, one);
        hist_add_chunk(mine, v3, one);
    }
    for (; i < nvec; i += THREADS) hist_add_chunk(mine, load_stream16(q + i), one);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)tid < len) atomicAdd(&mine[p[tail0 + tid]], one);   /* < 16 bytes */
    __syncthreads();
    for (int b = tid; b < HUF_NSYM; b += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < COPIES; w++) {
            const uint32_t x = s_hist[w * HUF_NSYM + b];
            sum += PACKED ? ((x & 0xffffu) + (x >> 16)) : x;
        }
        s_tot[b] = sum;
    }
    __syncthreads();                     /* copies are dead from here on; s_tot is complete */
    if (tid >= 64) return;               /* ended waves do not take part in anything below */
    uint32_t rate[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rate[j] = s_tot[tid + 64 * j];
    const uint64_t bytes = tree_fast_wave(rate, *reinterpret_cast<TreeLds *>(s_union), blk, codetab, treebuf, meta);
    /* stream offsets (the reference's running file position): summed here, see two_level_arrive */
    two_level_arrive(sizes, blk, gridDim.x, bytes);
}


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

/* byte j of the block header (encoder.c:325-339, little-endian fields) */
__device__ __forceinline__ uint32_t header_byte(uint32_t j, uint64_t block_len, uint32_t tree_len,
                                                const int16_t *__restrict__ tb)
{
    if (j < 8) return (uint32_t)(block_len >> (8 * j)) & 0xffu;
    if (j < 10) return (tree_len >> (8 * (j - 8))) & 0xffu;
    const uint16_t e = (uint16_t)tb[(j - 10) >> 1];
    return (e >> (8 * (j & 1))) & 0xffu;
}

template <int THREADS, typename CodeT>
__device__ __forceinline__ void pack_block(const uint8_t *__restrict__ src, uint64_t len,
                                           const hufcode_t *__restrict__ codes64,
                                           const int16_t *__restrict__ tb, uint32_t tree_len,
                                           uint8_t *__restrict__ out, uint64_t dst0, uint64_t dst1,
                                           CodeT *s_code, uint32_t *s_part, uint32_t *s_tail, uint32_t *s_stage)
{
    constexpr int TILE = THREADS * PACK_SPT;
    constexpr int WAVES = THREADS / 64;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    uint8_t *g_a0 = out + (dst0 & ~3ull);
    const uint32_t rec_lo = (uint32_t)(dst0 & 3ull);            /* record bytes relative to A0 */
    const uint64_t rec_hi = rec_lo + (dst1 - dst0);
    uint32_t *g_w0 = reinterpret_cast<uint32_t *>(g_a0);

    for (int i = tid; i < HUF_NSYM; i += THREADS) s_code[i] = (CodeT)codes64[i];

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
        return;
    }
    if (tid == 0) {
        uint32_t t = 0;                                          /* big-endian partial word */
        for (uint32_t bp = hdr_end & ~3u; bp < hdr_end; bp++)
            t = (t << 8) | header_byte(bp - rec_lo, len, tree_len, tb);
        s_tail[WAVES] = t;                                       /* carry: value of the (hdr_end&3)*8 leading bits */
    }
    __syncthreads();

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
        const uint32_t ex = block_excl_scan<THREADS, uint32_t>(mybits, s_part, tile_bits);

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
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) a.push(code[k]);  /* absent symbols have len 0 */
        } else {
            a.gw = g_first;
#pragma unroll
            for (int k = 0; k < PACK_SPT; k++) a.push(code[k]);
        }
        const uint32_t nwords = (uint32_t)(a.gw - (staged ? s_first : g_first));   /* finished words of this lane */
        const uint32_t tail_val = (uint32_t)(a.acc & ((1ull << a.nacc) - 1ull));

        /* ---- tails hop one lane to the right ---- */
        uint32_t in_tail = (uint32_t)__shfl_up((int)tail_val, 1);
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

/* SHORT = true: the host guarantees that no code of this launch is longer than 24 bits (any
 * Huffman merge order on n <= 121392 symbols gives depth <= 23, plus the wrap-root bit; the
 * deepest tree needs Fibonacci weights), so only the 32-bit code path is compiled - fewer
 * registers, more waves. */
template <int THREADS, bool SHORT>
__global__ __launch_bounds__(THREADS) void pack_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                       uint64_t blocksize,
                                                       const hufcode_t *__restrict__ codetab,
                                                       const int16_t *__restrict__ treebuf,
                                                       const HufBlockMeta *__restrict__ meta,
                                                       uint64_t *__restrict__ offsets, TwoLevel sizes,
                                                       uint8_t *__restrict__ out)
{
    __shared__ hufcode_t s_code[SHORT ? HUF_NSYM / 2 : HUF_NSYM];   /* u32[256] on the short-code path */
    __shared__ uint32_t s_part[THREADS / 64];
    __shared__ uint32_t s_tail[THREADS / 64 + 1];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[PACK_STAGE_WORDS];

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
    if (SHORT || m.max_len <= 24)
        pack_block<THREADS, uint32_t>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                      reinterpret_cast<uint32_t *>(s_code), s_part, s_tail, s_stage);
    else if constexpr (!SHORT)
        pack_block<THREADS, hufcode_t>(in + base, len, codes, tb, m.tree_len, out, o0, o1,
                                       s_code, s_part, s_tail, s_stage);
}

/* ======================================================================================
 * header parse of src/decoder.c:218-252 for every indexed block + output offsets
 * ==================================================================================== */
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p)
{
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) v |= (uint64_t)p[k] << (8 * k);
    return v;
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
                                                                    int32_t *__restrict__ status, TwoLevel lens)
{
    __shared__ uint64_t s_part[SCAN_GROUP / 64];
    __shared__ unsigned long long s_bad;
    if (threadIdx.x == 0) s_bad = ~0ull;
    const uint64_t b = (uint64_t)blockIdx.x * SCAN_GROUP + threadIdx.x;
    HufDecodeMeta m;
    m.block_len = 0;
    m.tree_len = 0;
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
                if (bl > 0xffffffffull) m.status = HUFE_ARGUMENT;               /* beyond kernel limits */
                else {
                    m.block_len = bl;
                    m.tree_len = tl;
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
    int16_t ent[ENT];
    uint16_t left[ENT];
    uint16_t right[ENT];
    uint16_t lut[1 << DEC_LUT_BITS];
    uint32_t pay[DEC_SUB_WORDS * COLS];  /* segment word i at pay[(i % W) * COLS + i / W]: lane-consecutive = bank-consecutive
                                            (a padded linear layout has a cheaper address but costs 2 KiB = one workgroup per CU) */
    uint16_t mark[DEC_SUB_WORDS][THREADS];  /* (codewords before << 5 | offset) of lane l's first visit to each word */
    uint32_t wend[THREADS / 64];         /* end position of the last lane of each wave (neighbours use shuffles) */
    uint32_t part[THREADS / 64];
    int efflen;
    uint32_t badsym;                     /* segment symbol index of the first walk that left the tree */
    uint32_t firstone;                   /* single-leaf trees: first set payload bit */
    uint32_t qend;                       /* segment bit right after the block's last symbol */
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

#ifdef DEC_RARE_NOINLINE
#define DEC_RARE_ATTR __noinline__
#else
#define DEC_RARE_ATTR __forceinline__
#endif
/* Bit-serial walk for `long` entries (and the verdict of a `bad` one), on the staged words.
 * CW_OK: sym, npos = position after the codeword.  CW_BAD: the walk left the tree, npos =
 * position after the failing bit.  CW_EXH: the walk needs bits past the readable payload.
 * Result packed in registers (no stack): bits 0-31 npos, 32-39 sym, 40-41 status. */
template <int THREADS>
__device__ DEC_RARE_ATTR uint64_t dec_rare_packed(const DecShared<THREADS> &sh, uint32_t e, uint32_t pos, uint32_t pay_rel)
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
        const uint32_t nx = bit ? sh.right[node] : sh.left[node];
        if (nx == DEC_NULL) return ((uint64_t)CW_BAD << 40) | p;
        node = nx;
        if (sh.left[node] == DEC_NULL && sh.right[node] == DEC_NULL) break;
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
#ifndef DEC_NO_WORDS
        if (!MERGE && start - sub_lo < 32u) dec_scan_words<THREADS>(sh, tr, start, sub_lo, pay_rel);
        else
#endif
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

/* Trees whose root has one leaf child on the left: every symbol is a single 0 bit and a 1 bit
 * leaves the tree (src/decoder.c:69-71).  Scan the needed payload bits for a set bit, then the
 * output is a fill.  sh.firstone must be DEC_NO_BAD on entry. */
template <int THREADS, bool STORE>
__device__ int decode_single_leaf(DecShared<THREADS> &sh, uint32_t symv, const uint8_t *pay, uint64_t block_len,
                                  uint64_t pay_bytes, uint8_t *gout, uint64_t *end_bits, uint64_t *produced_out)
{
    const int tid = (int)threadIdx.x;
    const uint64_t pay_bits = pay_bytes * 8ull;
    const uint64_t have = dmin<uint64_t>(block_len, pay_bits);       /* bits we may look at */
    const uint64_t nwords = (have + 31) >> 5;
    uint32_t first = DEC_NO_BAD;
    for (uint64_t w = (uint64_t)tid; w < nwords; w += THREADS) {
        uint32_t v = load_be32(pay, w * 4, pay_bytes);
        const uint64_t left_bits = have - (w << 5);
        if (left_bits < 32) v &= ~(0xffffffffu >> (uint32_t)left_bits);
        if (v) { first = (uint32_t)dmin<uint64_t>(first, (w << 5) + (uint32_t)__clz(v)); break; }
    }
    if (first != DEC_NO_BAD) atomicMin(&sh.firstone, first);
    __syncthreads();
    const uint32_t fo = sh.firstone;
    const uint64_t good = (fo != DEC_NO_BAD) ? (uint64_t)fo : have;
    if (STORE) {                        /* fill gout[0, good) */
        const uint32_t rep = symv * 0x01010101u;
        const uint64_t head = dmin<uint64_t>(good, (16u - (uint32_t)((uintptr_t)gout & 15u)) & 15u);
        if ((uint64_t)tid < head) gout[tid] = (uint8_t)symv;
        uint4 *q = reinterpret_cast<uint4 *>(gout + head);
        const uint64_t nvec = (good - head) >> 4;
        const uint4 v4 = make_uint4(rep, rep, rep, rep);
        for (uint64_t i = (uint64_t)tid; i < nvec; i += THREADS) store_stream16(q + i, v4);
        const uint64_t tail0 = head + (nvec << 4);
        if (tail0 + (uint64_t)tid < good) gout[tail0 + tid] = (uint8_t)symv;
    }
    *produced_out = good;
    if (fo != DEC_NO_BAD) return HUFE_CORRUPTED;                     /* decoder.c:69-71 */
    if (have < block_len) return HUFE_RW;                            /* decoder.c:53-56 */
    *end_bits = block_len;
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
    constexpr int ENT = DecShared<THREADS>::ENT;
    constexpr int COLS = DecShared<THREADS>::COLS;
    const int tid = (int)threadIdx.x;
    *produced_out = 0;

    /* ---- 0. the tree every one-symbol block carries, [root, leaf, -1, -1, -1],
     *         is recognised straight from its five entries; other shapes of single-leaf trees are
     *         caught after the general tree build below ---- */
    unsigned long long pt = DPROF_T();
    if (tree_len == 5) {
        int16_t e5[5];
#pragma unroll
        for (int i = 0; i < 5; i++) e5[i] = (int16_t)((uint16_t)tree[2 * i] | ((uint16_t)tree[2 * i + 1] << 8));
        if (e5[0] != -1 && e5[1] != -1 && e5[2] == -1 && e5[3] == -1 && e5[4] == -1) {
            __syncthreads();
            if (tid == 0) sh.firstone = DEC_NO_BAD;
            __syncthreads();
            return decode_single_leaf<THREADS, STORE>(sh, (uint32_t)(uint8_t)e5[1], tree + 10, block_len, pay_bytes, gout,
                                               end_bits, produced_out);
        }
    }

    /* ---- 1. tree ---- */
    __syncthreads();           /* previous user of sh is done */
    uint16_t *s_open = reinterpret_cast<uint16_t *>(&sh.pay[0]);   /* S(i); payload not staged yet */
    static_assert(sizeof(sh.pay) >= ENT * sizeof(uint16_t), "S(i) scratch must fit");
    for (int i = tid; i < ENT; i += THREADS) {
        int16_t v = -1;
        if (i < tree_len) v = (int16_t)((uint16_t)tree[2 * i] | ((uint16_t)tree[2 * i + 1] << 8));
        sh.ent[i] = v;
        sh.left[i] = DEC_NULL;
        sh.right[i] = DEC_NULL;
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
        const uint32_t ex = block_excl_scan<THREADS, uint32_t>((uint32_t)sum, sh.part, tot);
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
    /* Child links from subtree sizes: size(marker) = 1, size(node) = 1 + size(left) + size(right),
     * left child of node j is entry j+1, right child is entry j+1+size(j+1); entries at or past
     * `eff` do not exist (size 0, NULL).  Sizes become known bottom-up, one tree level per round
     * (s_open is reused as the size array, 0 = not known yet). */
    __syncthreads();
    uint16_t *s_size = s_open;
    for (int i = tid; i < ENT; i += THREADS) s_size[i] = (i < eff && sh.ent[i] == -1) ? 1 : 0;
    __syncthreads();
    for (int round = 0; round < ENT; round++) {
        int progress = 0;
        for (int j = tid; j < eff; j += THREADS) {
            if (s_size[j] != 0) continue;                 /* marker or already done */
            const int l = j + 1;
            uint32_t sl = 0, sr = 0;
            bool ready = true;
            if (l < eff) {
                sl = s_size[l];
                if (sl == 0) ready = false;
                else {
                    const int r = l + (int)sl;
                    if (r < eff) {
                        sr = s_size[r];
                        if (sr == 0) ready = false;
                        else if (sh.ent[r] != -1) sh.right[j] = (uint16_t)r;
                    }
                }
                if (ready && sh.ent[l] != -1) sh.left[j] = (uint16_t)l;
            }
            if (ready) {
                s_size[j] = (uint16_t)(1 + sl + sr);
                progress = 1;
            }
        }
        if (!__syncthreads_or(progress)) break;
    }
    __syncthreads();
    /* tree_len == 0 or a tree that starts with -1 is a NULL root: the reference crashes,
     * the decision is BTREE_CORRUPTED */
    if (!(eff > 0 && sh.ent[0] != -1)) return HUFE_CORRUPTED;

    const uint8_t *pay = tree + 2 * tree_len;
    const uint64_t pay_bits = pay_bytes * 8ull;
    DPROF_ADD(0, pt); pt = DPROF_T();

    /* ---- 2. single-leaf tree: every symbol is one 0 bit ---- */
    {
        const uint32_t l0 = sh.left[0];
        if (l0 != DEC_NULL && sh.right[0] == DEC_NULL && sh.left[l0] == DEC_NULL && sh.right[l0] == DEC_NULL)
            return decode_single_leaf<THREADS, STORE>(sh, (uint32_t)(uint8_t)sh.ent[l0], pay, block_len, pay_bytes, gout,
                                               end_bits, produced_out);
    }

    /* ---- 3. lookup table ----
     * Two hops of DEC_LUT_BITS/2 bits: first the node (or verdict) reached after the high half
     * of the index, kept in the low 64 table slots for a moment, then every entry continues
     * from there - half the dependent LDS steps of walking all 12 bits per entry. */
    constexpr int HALF = DEC_LUT_BITS / 2;
    uint32_t hop1 = 0;
    if (tid < (1 << HALF)) {
        uint32_t node = 0, e = 0;
        bool done = false;
#pragma unroll 1
        for (int b = 0; b < HALF; b++) {
            const uint32_t bit = ((uint32_t)tid >> (HALF - 1 - b)) & 1u;
            const uint32_t nx = bit ? sh.right[node] : sh.left[node];
            if (nx == DEC_NULL) { e = (2u << 14) | ((uint32_t)(b + 1) << 8); done = true; break; }
            node = nx;
            if (sh.left[node] == DEC_NULL && sh.right[node] == DEC_NULL) {
                e = ((uint32_t)(b + 1) << 8) | ((uint32_t)(uint8_t)sh.ent[node]);
                done = true;
                break;
            }
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
#pragma unroll
        for (int k = 0; k < PERL; k++) {
            const int idx = tid + k * THREADS;
            uint32_t e = s_hop[idx >> HALF];
            if ((e >> 14) == 1u) {                         /* still inside the tree after the first hop */
                uint32_t node = e & 0x7ffu;
                bool done = false;
#pragma unroll 1
                for (int b = HALF; b < DEC_LUT_BITS; b++) {
                    const uint32_t bit = ((uint32_t)idx >> (DEC_LUT_BITS - 1 - b)) & 1u;
                    const uint32_t nx = bit ? sh.right[node] : sh.left[node];
                    if (nx == DEC_NULL) { e = (2u << 14) | ((uint32_t)(b + 1) << 8); done = true; break; }
                    node = nx;
                    if (sh.left[node] == DEC_NULL && sh.right[node] == DEC_NULL) {
                        e = ((uint32_t)(b + 1) << 8) | ((uint32_t)(uint8_t)sh.ent[node]);
                        done = true;
                        break;
                    }
                }
                if (!done) e = (1u << 14) | node;
            }
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
#ifndef DEC_NO_FOLD
        /* A speculative lane that meets a run of failing bits decodes the codeword behind the run
         * in its next iteration; when run + codeword fit the window, one entry does both
         * (DEC_E_NOCW clear), which helps data with short codes (uniform bytes 2.61 -> 2.48 ms). */
        __syncthreads();
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
#endif
    }
    __syncthreads();
    DPROF_ADD(1, pt);
#if defined(DEC_ABLATE) && DEC_ABLATE == 2
    *end_bits = 0; return HUFE_OK;
#endif
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
#if defined(DEC_DUP) && DEC_DUP == 1     /* cost of a phase = time with the phase done twice - time */
        for (int i = tid; i < DEC_SUB_WORDS * COLS; i += THREADS)
            sh.pay[pay_slot<COLS>((uint32_t)i)] = load_be32(pay, byte0 + 4ull * i, pay_bytes);
        __syncthreads();
#endif
        const uint32_t pay_rel = (uint32_t)dmin<uint64_t>(pay_bits - seg0, 0xfffffff0ull);
        const uint32_t first_start = (uint32_t)(true_start - seg0);
        DPROF_ADD(2, pt); pt = DPROF_T();

        LaneTrack tr;
        dec_scan<THREADS, false>(sh, tr, tid == 0 ? first_start : sub_lo, sub_lo, pay_rel);
#if defined(DEC_DUP) && DEC_DUP == 2
        __syncthreads();
        dec_scan<THREADS, false>(sh, tr, tid == 0 ? first_start : sub_lo, sub_lo, pay_rel);
#endif
        if ((tid & 63) == 63) sh.wend[tid >> 6] = tr.end;
        __syncthreads();
        DPROF_ADD(3, pt); pt = DPROF_T();
        for (;;) {
            /* left neighbour's end: a shuffle inside the wave, LDS across the wave seams */
            uint32_t ns = (uint32_t)__shfl_up((int)tr.end, 1);
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
        const uint32_t ex = block_excl_scan<THREADS, uint32_t>(tr.cnt, sh.part, seg_total);
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
#if defined(DEC_DUP) && DEC_DUP == 3
                (void)dec_write<THREADS, true>(sh, tr.start, pay_rel, quota, gout + produced + ex);
#endif
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
            err = decode_block<THREADS>(sh, stream + o0 + HUF_HEADER_FIXED, m.tree_len, m.block_len,
                                        pay_bytes, out + obase, &end_bits, &produced);
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
 * result[0] = error, [1] = bytes written, [2] = reader bytes consumed, [3] = blocks done. */
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
    int err = HUFE_OK;
    while (length > rd) {                                             /* decoder.c:218 */
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
        if (want > 0xffffffffull) { err = HUFE_ARGUMENT; break; }
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
        if (block_offsets && nblk < max_index) block_offsets[nblk] = rd;
    }
}

/* ======================================================================================
 * Raw-stream block discovery (SURVEY §7.3-A option 2, §8f-2).
 *
 * The wire format stores no payload length, so the header of block i+1 is only found by
 * decoding block i.  To break that chain without changing any result:
 *   1. discover_kernel tests EVERY byte offset for a syntactically valid header
 *      (block_len in range, tree_len in [1, max], a preorder tree that consumes exactly tree_len
 *      entries, enough bytes left) - every real header passes, almost nothing else does;
 *   2. probe_kernel decodes every candidate in count-only mode (decode_block<.., false>), which
 *      yields the offset right behind its payload;
 *   3. link_kernel / walk_kernel follow the chain offset 0 -> end(0) -> ... through the sorted
 *      candidates; a false candidate can never be entered, because a real block's end is the
 *      next real header;
 *   4. the validated prefix is decoded by the indexed kernels; whatever the walk could not
 *      validate (an erroring block, a header the strict test rejects, trailing garbage) is left
 *      to decode_chain_kernel, the exact sequential restatement - so errors, partial output and
 *      consumed-byte counts are those of src/decoder.c in every case.
 * ==================================================================================== */
#define DISC_THREADS 256
#define DISC_PER 16
#define DISC_ITERS 4
#define DISC_CHUNK (DISC_THREADS * DISC_PER * DISC_ITERS)
#define LINK_BAD      0xfffffffdu
#define LINK_TERMINAL 0xfffffffeu
#define LINK_NOTFOUND 0xffffffffu

__device__ __forceinline__ bool tree_grammar_complete(const uint8_t *t, int tl)
{
    int open = 1;                               /* child slots still to be filled */
    for (int i = 0; i < tl; i++) {
        if (open == 0) return false;            /* entries behind a complete tree */
        const int16_t v = (int16_t)((uint16_t)t[2 * i] | ((uint16_t)t[2 * i + 1] << 8));
        open += (v != -1) ? 1 : -1;
    }
    return open == 0;
}

/* stream must be 16-byte aligned.  WRITE = false: per-workgroup candidate counts;
 * WRITE = true: candidates written in ascending order at wg_base[workgroup]. */
/* The same test by a whole wavefront (all 64 lanes call it with the same arguments): 64 entries
 * per step, open-slot counts by a wave prefix sum.  A lane walking the up to 1 025 entries alone
 * is ~1 000 pairs of dependent byte loads (~0.5 ms), and every real header costs one such walk. */
__device__ __noinline__ bool tree_grammar_complete_wave(const uint8_t *t, int tl)
{
    const int lane = lane_id();
    int open = 1;                                                /* child slots still to be filled */
    bool bad = false;
    for (int base = 0; base < tl; base += 64) {                  /* uniform */
        const int i = base + lane;
        int d = 0;
        if (i < tl) d = (((uint32_t)t[2 * i] | ((uint32_t)t[2 * i + 1] << 8)) != 0xffffu) ? 1 : -1;
        int inc = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (i < tl && open + inc - d <= 0) bad = true;           /* an entry behind a complete tree */
        open += __shfl(inc, 63);
    }
    return __ballot(bad) == 0ull && open == 0;
}

template <bool WRITE>
__global__ __launch_bounds__(DISC_THREADS) void discover_kernel(const uint8_t *__restrict__ stream, uint64_t avail,
                                                                uint64_t scan_len, int max_tree_len,
                                                                uint32_t *__restrict__ wg_counts,
                                                                const uint64_t *__restrict__ wg_base,
                                                                uint64_t *__restrict__ cand,
                                                                uint64_t *__restrict__ masks)
{
    /* A thread owns DISC_ITERS consecutive 16-byte pieces (thread order = stream order, one scan
     * per workgroup of 16 KiB: with 4 KiB workgroups the kernel was bound by their dispatch). */
    __shared__ uint32_t s_part[DISC_THREADS / 64];
    const uint64_t t0 = (uint64_t)blockIdx.x * DISC_CHUNK + (uint64_t)threadIdx.x * (DISC_PER * DISC_ITERS);
    const uint64_t slot = (uint64_t)blockIdx.x * DISC_THREADS + threadIdx.x;
    uint64_t mask = 0;
    /* the counting pass leaves its 64 verdicts per thread for the writing pass, which then reads
     * 1/8 of the stream's size instead of testing the whole stream again */
    if (WRITE) mask = masks[slot];
    else if (t0 < scan_len) {
        /* all five loads of the thread are issued before the first use (one memory round trip) */
        uint4 v[DISC_ITERS + 1];
        v[0] = *reinterpret_cast<const uint4 *>(stream + t0);                  /* 16-byte unit that holds a valid byte */
#pragma unroll
        for (int it = 1; it <= DISC_ITERS; it++) {
            v[it] = make_uint4(0u, 0u, 0u, 0u);
            if (t0 + (uint64_t)(it * DISC_PER) < avail) v[it] = *reinterpret_cast<const uint4 *>(stream + t0 + it * DISC_PER);
        }
#pragma unroll
        for (int it = 0; it < DISC_ITERS; it++) {
            const uint64_t p0 = t0 + (uint64_t)(it * DISC_PER);
            const uint4 a = v[it], b = v[it + 1];              /* zeros behind the data: no survivors there */
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            /* almost no offset survives "the upper half of block_len is zero": that test is done
             * for all 16 offsets without a branch, everything else only for the survivors */
            uint32_t maybe = 0;
#pragma unroll
            for (int k = 0; k < DISC_PER; k++) {
                const uint32_t hi = __funnelshift_r(w[(k + 4) >> 2], w[((k + 4) >> 2) + 1], 8 * ((k + 4) & 3));
                maybe |= (hi == 0u ? 1u : 0u) << k;                              /* block_len < 2^32 */
            }
            if (maybe) {                                    /* runs of zero bytes pass the first test everywhere: */
                uint32_t nz = 0;                            /* block_len != 0, again for all offsets at once */
#pragma unroll
                for (int k = 0; k < DISC_PER; k++) {
                    const uint32_t lo = __funnelshift_r(w[k >> 2], w[(k >> 2) + 1], 8 * (k & 3));
                    nz |= (lo != 0u ? 1u : 0u) << k;
                }
                maybe &= nz;
            }
            if (p0 >= scan_len) maybe = 0;
            /* survivors (rare): the lane checks the header fields of its next one, then the wave
             * checks the tree grammar of every lane's survivor together, one after the other */
            while (__ballot(maybe != 0u) != 0ull) {
                int k = 0, tl = 0;
                uint64_t p = 0;
                bool pre = false;
                if (maybe) {
                    k = __builtin_ctz(maybe);
                    maybe &= maybe - 1;
                    p = p0 + (uint64_t)k;
                    if (p < scan_len && avail - p >= HUF_HEADER_FIXED) {
                        const uint8_t *h = stream + p;
                        const uint32_t lo = (uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16) | ((uint32_t)h[3] << 24);
                        tl = (int)(int16_t)((uint16_t)h[8] | ((uint16_t)h[9] << 8));
                        if (lo != 0 && tl >= 1 && tl <= max_tree_len) {
                            const uint64_t hdr_end = p + HUF_HEADER_FIXED + 2ull * (uint64_t)tl;
                            pre = hdr_end <= avail && (uint64_t)lo <= (avail - hdr_end) * 8ull;   /* every symbol costs a bit */
                        }
                    }
                }
                unsigned long long pend = __ballot(pre);
                while (pend) {
                    const int src = __builtin_ctzll(pend);
                    pend &= pend - 1;
                    const uint64_t sp = uni64((uint64_t)__shfl((unsigned long long)p, src));
                    const int stl = (int)uni32((uint32_t)__shfl(tl, src));
                    const bool ok = tree_grammar_complete_wave(stream + sp + HUF_HEADER_FIXED, stl);
                    if (ok && lane_id() == src) mask |= 1ull << (it * DISC_PER + k);
                }
            }
        }
    }
    uint32_t total;
    const uint32_t ex = block_excl_scan<DISC_THREADS, uint32_t>((uint32_t)__popcll(mask), s_part, total);
    if (!WRITE) {
        masks[slot] = mask;
        if (threadIdx.x == 0) wg_counts[blockIdx.x] = total;
    } else {
        uint64_t at = wg_base[blockIdx.x] + ex;
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            cand[at++] = t0 + (uint64_t)k;
        }
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void scan_counts_kernel(const uint32_t *__restrict__ counts, uint64_t n,
                                                              uint64_t *__restrict__ base)
{
    const uint64_t total = chunked_excl_scan<THREADS>(n, base, [counts](uint64_t i) { return (uint64_t)counts[i]; });
    if (threadIdx.x == 0) base[n] = total;
}

/* Where the output of candidate i would start if every candidate were a block of the stream, in
 * order: the exclusive prefix sum of the block_len fields (ONE workgroup; spec_off[ncand] = sum). */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void cand_lens_kernel(const uint8_t *__restrict__ stream,
                                                            const uint64_t *__restrict__ cand, uint64_t ncand,
                                                            uint64_t *__restrict__ spec_off)
{
    const uint64_t total = chunked_excl_scan<THREADS>(ncand, spec_off, [=](uint64_t i) {
        return load_u64_unaligned(stream + cand[i]);
    });
    if (threadIdx.x == 0) spec_off[ncand] = total;
}

/* Decode of one candidate: where does its payload end, and does it decode at all?  Count-only,
 * unless all candidates together fit the output (spec_off[ncand] <= out_cap): then the symbols
 * are written where they belong if every candidate is a real block - the usual case, in which the
 * chain walk afterwards confirms exactly that and nothing has to be decoded twice. */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void probe_kernel(const uint8_t *__restrict__ stream, uint64_t avail,
                                                        const uint64_t *__restrict__ cand,
                                                        uint64_t *__restrict__ cand_end,
                                                        int32_t *__restrict__ cand_status,
                                                        const uint64_t *__restrict__ spec_off, uint8_t *__restrict__ out,
                                                        uint64_t out_cap)
{
    __shared__ DecShared<THREADS> sh;
    const uint64_t c = cand[blockIdx.x];
    const uint64_t block_len = load_u64_unaligned(stream + c);
    const int tl = (int)(int16_t)((uint16_t)stream[c + 8] | ((uint16_t)stream[c + 9] << 8));
    const uint64_t pay0 = c + HUF_HEADER_FIXED + 2ull * (uint64_t)tl;
    uint64_t end_bits = 0, produced = 0;
    int err;
    if (spec_off[gridDim.x] <= out_cap)
        err = decode_block<THREADS, true>(sh, stream + c + HUF_HEADER_FIXED, tl, block_len, avail - pay0,
                                          out + spec_off[blockIdx.x], &end_bits, &produced);
    else
        err = decode_block<THREADS, false>(sh, stream + c + HUF_HEADER_FIXED, tl, block_len, avail - pay0,
                                           nullptr, &end_bits, &produced);
    if (threadIdx.x == 0) {
        cand_status[blockIdx.x] = err;
        cand_end[blockIdx.x] = pay0 + ((end_bits + 7) >> 3);
    }
}

__global__ void link_kernel(const uint64_t *__restrict__ cand, const uint64_t *__restrict__ cand_end,
                            const int32_t *__restrict__ cand_status, uint64_t ncand, uint64_t length,
                            uint32_t *__restrict__ nxt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand) return;
    uint32_t r;
    if (cand_status[i] != HUFE_OK) r = LINK_BAD;
    else {
        const uint64_t e = cand_end[i];
        if (e >= length) r = LINK_TERMINAL;                      /* src/decoder.c:218 loop condition */
        else {
            uint64_t lo = i + 1, hi = ncand;                     /* first candidate with offset >= e */
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (cand[mid] < e) lo = mid + 1; else hi = mid;
            }
            r = (lo < ncand && cand[lo] == e) ? (uint32_t)lo : LINK_NOTFOUND;
        }
    }
    nxt[i] = r;
}

/* result: [0] validated blocks m, [1] offset where the sequential decoder must take over
 * (meaningful when [2] == 0), [2] 1 = the chain reached `length`, [3] bytes consumed then,
 * [4] see below.
 * block_offsets[0..m] receives the validated block index.
 * ONE wavefront follows the chain through an LDS copy of nxt[] (the chain only moves forward, so
 * the copy is refilled chunk by chunk).  In a stream without false candidates every link is
 * "the next candidate", so the wave tests 64 links per step and takes the whole run of such
 * links at once (16 384 blocks: 256 steps instead of 16 384 dependent LDS reads, 2.9 -> 0.1 ms
 * per GiB); any other link is followed one step at a time.  Every loop-control value is the
 * same in all lanes (ballots), so no flag is ever polled in memory. */
#define WALK_CHUNK 8192
__global__ __launch_bounds__(64) void walk_kernel(const uint64_t *__restrict__ cand,
                                                  const uint64_t *__restrict__ cand_end,
                                                  const uint32_t *__restrict__ nxt, uint64_t ncand,
                                                  uint64_t *__restrict__ block_offsets,
                                                  uint64_t *__restrict__ result,
                                                  const uint64_t *__restrict__ spec_off, uint64_t out_cap)
{
    __shared__ uint32_t s_nxt[WALK_CHUNK];
    const int lane = (int)threadIdx.x;
    uint64_t cur = 0, m = 0, resume = 0, consumed = 0;
    int complete = 0;
    bool contiguous = true;       /* validated block j is candidate j, for every j so far */
    bool stop = (ncand == 0) || (cand[0] != 0);       /* the stream must start with a header */
    while (!stop) {
        const uint64_t base = cur - (cur % WALK_CHUNK);
        const uint64_t top = dmin<uint64_t>(base + WALK_CHUNK, ncand);      /* candidates [base, top) are in LDS */
        for (uint64_t i = (uint64_t)lane; base + i < top; i += 64) s_nxt[i] = nxt[base + i];
        __syncthreads();
        uint64_t c = cur;
        while (c < top) {
            /* the run of plain links that starts at c */
            const uint64_t idx = c + (uint64_t)lane;
            const bool plain = idx < top && s_nxt[idx - base] == (uint32_t)(idx + 1);
            const unsigned long long mask = __ballot(plain);
            const uint32_t run = (~mask == 0ull) ? 64u : (uint32_t)__builtin_ctzll(~mask);
            if ((uint32_t)lane < run) block_offsets[m + (uint64_t)lane] = cand[idx];
            if (run && c != m) contiguous = false;
            m += run;
            c += run;
            if (run == 64u || c >= top) continue;
            /* one link of another kind */
            const uint32_t nx = s_nxt[c - base];
            if (nx == LINK_BAD) { resume = cand[c]; stop = true; break; }
            if (lane == 0) block_offsets[m] = cand[c];
            if (c != m) contiguous = false;
            m++;
            if (nx == LINK_TERMINAL) { complete = 1; consumed = cand_end[c]; stop = true; break; }
            if (nx == LINK_NOTFOUND) { resume = cand_end[c]; stop = true; break; }
            c = nx;                                    /* nx > c: the chain only moves forward */
        }
        cur = c;
        __syncthreads();                               /* before s_nxt is reused */
    }
    if (lane == 0) {
        result[0] = m;
        result[1] = resume;
        result[2] = (uint64_t)complete;
        result[3] = consumed;
        /* [4]: bytes the probe already put in place for the validated blocks (~0 = it did not) */
        result[4] = (contiguous && spec_off[ncand] <= out_cap) ? spec_off[m] : ~0ull;
        block_offsets[m] = complete ? consumed : resume;   /* end of the validated prefix */
    }
}

/* ======================================================================================
 * Synthetic inputs of SURVEY §8d (libhuffman_amd/datagen.py is the numpy twin).
 * ==================================================================================== */
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void fill_kernel(uint8_t *__restrict__ out, uint64_t n, int kind, uint64_t seed, uint64_t first,
                            const uint64_t *__restrict__ zipf_cum)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = first + i;                 /* global byte index */
        uint8_t v;
        if (kind == 0) v = 0x41;
        else if (kind == 1) v = (uint8_t)(splitmix64_at(seed, (g >> 3) + 1) >> (8 * (g & 7)));
        else if (kind == 2) v = (uint8_t)(splitmix64_at(seed, g + 1) % 255ull);
        else {
            const uint64_t u = splitmix64_at(seed, g + 1) % zipf_cum[254];
            int lo = 0, hi = 255;                     /* number of cum[r] <= u */
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (zipf_cum[mid] <= u) lo = mid + 1; else hi = mid;
            }
            v = (uint8_t)lo;
        }
        out[i] = v;
    }
}

}  // namespace hufgpu
