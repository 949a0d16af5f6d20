/* discover.hpp - raw-stream block discovery: discover / probe / link / walk kernels.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "decode.hpp"
#include "decode_fast.hpp"
#include "decode_regs.hpp"

namespace hufgpu {

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
#define DISC_SLOTS 4u                       /* candidates a discovery workgroup hands over directly (its 16 KiB hold one in four, at 64 KiB blocks) */
#define DISC_SCAN_GROUP 1024u              /* workgroup counts one workgroup of scan_counts_kernel sums */
/* the words the discovery's kernels share (ctx->d_walk): 0..4 walk_kernel's result, 5 the candidates found (may be more
 * than the arrays hold), 6 the exact probes' list length, 7 scan_counts_kernel's ticket, 8 links that are not "the next
 * candidate", 9 the candidates the arrays hold = min(found, capacity) */
enum { DISC_FOUND = 5, DISC_REDO = 6, DISC_TICKET = 7, DISC_ODD_LINKS = 8, DISC_NCAND = 9, DISC_WORDS = 16 };
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
/* The same test by a wavefront (its active lanes, a prefix of the wave, call it with the same arguments): 64 entries
 * per step, open-slot counts by a wave prefix sum.  A lane walking the up to 1 025 entries alone
 * is ~1 000 pairs of dependent byte loads (~0.5 ms), and every real header costs one such walk. */
__device__ __noinline__ bool tree_grammar_complete_wave(const uint8_t *t, int tl)
{
    const int lane = lane_id();
    /* The calling lanes are a prefix of the wave (the stream's last wave has lanes behind scan_len
     * switched off): entries are dealt over the nact active lanes and the step's total comes from
     * the last of them - a shuffle from an inactive lane is undefined. */
    const int nact = __popcll(__ballot(1));
    int open = 1;                                                /* child slots still to be filled */
    bool bad = false;
    if (nact == 64 && tl <= HUF_TREE_MAX) {
        /* Round 4: a whole wave (all but the stream's last).  The loop below fetches 64 entries, sums, fetches the next 64: seventeen
         * memory round trips for the 1 021 entries of a 255-symbol tree, 24 us in which the wave's workgroup keeps its place on
         * the CU - and every real header costs one such check: 16 384 of them were three quarters of the kernel's 260 us per GiB.
         * Here every lane asks for all of its entries first (one round trip), the sums are DPP scans. */
        typedef uint16_t __attribute__((aligned(1))) unaligned_u16;
        constexpr int STEPS = (HUF_TREE_MAX + 63) / 64;
        uint32_t e[STEPS];
#pragma unroll
        for (int j = 0; j < STEPS; j++) {
            const int i = j * 64 + lane;
            e[j] = (i < tl) ? (uint32_t)*reinterpret_cast<const unaligned_u16 *>(t + 2 * i) : 0u;
        }
#pragma unroll
        for (int j = 0; j < STEPS; j++) {
            if (j * 64 < tl) {                                   /* uniform */
                const int i = j * 64 + lane;
                const int d = (i < tl) ? ((e[j] != 0xffffu) ? 1 : -1) : 0;
                const int inc = (int)wave_incl_scan_u32((uint32_t)d);
                if (i < tl && open + inc - d <= 0) bad = true;       /* an entry behind a complete tree */
                open += (int)wave_lane_u32((uint32_t)inc, 63);
            }
        }
        return __ballot(bad) == 0ull && open == 0;
    }
    for (int base = 0; base < tl; base += nact) {                /* uniform */
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
        open += __shfl(inc, nact - 1);
    }
    return __ballot(bad) == 0ull && open == 0;
}

/* Round 6: ONE pass over the stream.  A workgroup that finds at most DISC_SLOTS candidates - all of them, in a stream of
 * blocks of a few KiB and more - leaves them (offset and block_len) in its own slots, in stream order; place_cands_kernel,
 * one thread a workgroup, moves them to their places once the counts are summed.  Until then the 64 verdicts of every
 * thread went to memory (an eighth of the stream's size written, and read again by a second launch of this kernel that
 * placed the candidates: 45 us per GiB beside the writes); only a workgroup with more candidates still does that, and
 * place_cands_kernel reads its verdicts. */
struct DiscSlot { uint64_t offset, block_len; };
__global__ __launch_bounds__(DISC_THREADS) void discover_kernel(const uint8_t *__restrict__ stream, uint64_t avail,
                                                                uint64_t scan_len, int max_tree_len,
                                                                uint32_t *__restrict__ wg_counts,
                                                                DiscSlot *__restrict__ slots,
                                                                uint64_t *__restrict__ masks)
{
    /* A workgroup scans 16 KiB (with 4 KiB workgroups the kernel was bound by their dispatch) in DISC_ITERS rows of
     * DISC_THREADS 16-byte pieces; a thread owns piece `tid` of every row.  (Round 4.  Until then a thread owned four
     * CONSECUTIVE pieces: every load of a wave touched 64 places 64 bytes apart.  Now a row is one contiguous
     * 4 KiB and the piece behind a thread's own - the window of an offset runs on into it - is the same load 16 bytes on.) */
    __shared__ uint32_t s_part[DISC_THREADS / 64], s_part2[DISC_THREADS / 64];
    const uint64_t wg0 = (uint64_t)blockIdx.x * DISC_CHUNK;
    const uint64_t t0 = wg0 + (uint64_t)threadIdx.x * DISC_PER;                 /* the thread's piece of row 0; row `it`: + it * DISC_THREADS * DISC_PER */
    constexpr uint64_t ROW = (uint64_t)DISC_THREADS * DISC_PER;
    const uint64_t slot = (uint64_t)blockIdx.x * DISC_THREADS + threadIdx.x;
    uint64_t mask = 0;                                                          /* bit 16 it + k: offset t0 + it * ROW + k */
    /* the counting pass leaves its 64 verdicts per thread for the writing pass, which then reads
     * 1/8 of the stream's size instead of testing the whole stream again */
    /* all loads of the thread are issued before the first use (one memory round trip).  Round 6: the piece BEHIND a thread's own - the
     * window of an offset runs on into it - is its right neighbour's own piece: taken from that lane's registers (wave_shl:1), only
     * the wave's last lane loads it (until then every piece was loaded twice; the second load hit the L1: 0.215 -> 0.207 ms per GiB).  Every lane loads its piece,
     * also one behind scan_len: its left neighbour looks into it. */
    uint4 va[DISC_ITERS], vb[DISC_ITERS];
#pragma unroll
    for (int it = 0; it < DISC_ITERS; it++) {
        const uint64_t q = t0 + (uint64_t)it * ROW;
        va[it] = make_uint4(0u, 0u, 0u, 0u);
        vb[it] = make_uint4(0u, 0u, 0u, 0u);
        if (q < avail) va[it] = *reinterpret_cast<const uint4 *>(stream + q);                  /* (a 16-byte unit that holds a valid byte) */
        if (lane_id() == 63 && q + DISC_PER < avail) vb[it] = *reinterpret_cast<const uint4 *>(stream + q + DISC_PER);
    }
#pragma unroll
    for (int it = 0; it < DISC_ITERS; it++) {
        uint4 nb;
        nb.x = (uint32_t)__builtin_amdgcn_update_dpp((int)vb[it].x, (int)va[it].x, 0x130, 0xf, 0xf, false);   /* wave_shl:1: lane i <- lane i + 1; lane 63 keeps what it loaded */
        nb.y = (uint32_t)__builtin_amdgcn_update_dpp((int)vb[it].y, (int)va[it].y, 0x130, 0xf, 0xf, false);
        nb.z = (uint32_t)__builtin_amdgcn_update_dpp((int)vb[it].z, (int)va[it].z, 0x130, 0xf, 0xf, false);
        nb.w = (uint32_t)__builtin_amdgcn_update_dpp((int)vb[it].w, (int)va[it].w, 0x130, 0xf, 0xf, false);
        vb[it] = nb;
    }
    if (t0 < scan_len) {
#pragma unroll
        for (int it = 0; it < DISC_ITERS; it++) {
            const uint64_t p0 = t0 + (uint64_t)it * ROW;
            const uint4 a = va[it], b = vb[it];                /* zeros behind the data: no survivors there */
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            /* almost no offset survives "the upper half of block_len is zero": that test is done
             * for all 16 offsets without a branch, everything else only for the survivors */
            uint32_t maybe = 0;
            /* (four zero bytes in a row hold an aligned zero halfword, wherever they begin: bytes 4..22 of the
             * window, words 1..5 - in a compressed stream one halfword in 65 536 is zero, and the sixteen tests
             * below were most of the kernel's instructions) */
            uint32_t zh = 0;
#pragma unroll
            for (int d = 1; d <= 5; d++) zh |= (w[d] - 0x00010001u) & ~w[d] & 0x80008000u;
            if (zh) {
#pragma unroll
                for (int k = 0; k < DISC_PER; k++) {
                    const uint32_t hi = __funnelshift_r(w[(k + 4) >> 2], w[((k + 4) >> 2) + 1], 8 * ((k + 4) & 3));
                    maybe |= (hi == 0u ? 1u : 0u) << k;                          /* block_len < 2^32 */
                }
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
    /* candidates in stream order: row by row, thread by thread inside a row - four counts (<= 16 a thread, <= 4 096 a row) in
     * two sums of 16-bit halves */
    static_assert(DISC_ITERS == 4 && DISC_PER == 16 && DISC_THREADS * DISC_PER < 65536, "four rows of 16-bit counts");
    uint32_t tot_lo, tot_hi;
    const uint32_t c_lo = (uint32_t)__popc((uint32_t)mask & 0xffffu) | ((uint32_t)__popc((uint32_t)(mask >> 16) & 0xffffu) << 16);
    const uint32_t c_hi = (uint32_t)__popc((uint32_t)(mask >> 32) & 0xffffu) | ((uint32_t)__popc((uint32_t)(mask >> 48)) << 16);
    /* (both sums behind ONE barrier - the partial words are written once a launch -, the waves' words summed by the lanes as in
     *  decode_regs.hpp's dreg_excl_scan: 0.211 -> 0.207 ms.  Measured beside it and not kept: workgroups of 512 / 1 024 threads
     *  0.27 / 0.45 ms, a launched workgroup scanning 2 / 4 / 8 pieces one after the other 0.227 / 0.232 / 0.237 - a wave lives one memory
     *  round trip, and many short waves hide it better than few long ones) */
    uint32_t ex_lo, ex_hi;
    {
        constexpr int WAVES = DISC_THREADS / 64;
        static_assert(WAVES == 4, "four waves' words summed by the first four lanes of a row");
        const uint32_t wv = uni32(threadIdx.x >> 6);
        const uint32_t i_lo = wave_incl_scan_u32(c_lo), i_hi = wave_incl_scan_u32(c_hi);
        if (lane_id() == 63) { s_part[wv] = i_lo; s_part2[wv] = i_hi; }
        __syncthreads();
        uint32_t x = s_part[lane_id() & (WAVES - 1)], y = s_part2[lane_id() & (WAVES - 1)];
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
        y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x111, 0xf, 0xf, true);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
        y += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x112, 0xf, 0xf, true);
        tot_lo = (uint32_t)__builtin_amdgcn_readlane((int)x, WAVES - 1);
        tot_hi = (uint32_t)__builtin_amdgcn_readlane((int)y, WAVES - 1);
        const uint32_t b_lo = wv != 0u ? (uint32_t)__builtin_amdgcn_readlane((int)x, (int)wv - 1) : 0u;
        const uint32_t b_hi = wv != 0u ? (uint32_t)__builtin_amdgcn_readlane((int)y, (int)wv - 1) : 0u;
        ex_lo = b_lo + i_lo - c_lo;
        ex_hi = b_hi + i_hi - c_hi;
    }
    const uint32_t tot[4] = {tot_lo & 0xffffu, tot_lo >> 16, tot_hi & 0xffffu, tot_hi >> 16};
    const uint32_t total = tot[0] + tot[1] + tot[2] + tot[3];
    if (total <= DISC_SLOTS) {                                     /* (uniform) */
        const uint32_t first[4] = {ex_lo & 0xffffu, tot[0] + (ex_lo >> 16), tot[0] + tot[1] + (ex_hi & 0xffffu), tot[0] + tot[1] + tot[2] + (ex_hi >> 16)};
#pragma unroll
        for (int it = 0; it < DISC_ITERS; it++) {
            uint32_t m = (uint32_t)(mask >> (16 * it)) & 0xffffu;
            uint32_t at = first[it];
            while (m) {
                const int k = __builtin_ctz(m);
                m &= m - 1;
                const uint64_t p = t0 + (uint64_t)it * ROW + (uint64_t)k;
                DiscSlot sl;
                sl.offset = p;
                sl.block_len = load_u64_unaligned(stream + p);      /* (cand_lens_kernel sums these: not 8 bytes from 16 384 places of the stream again) */
                slots[(uint64_t)blockIdx.x * DISC_SLOTS + at] = sl;
                at++;
            }
        }
    } else {
        masks[slot] = mask;
    }
    if (threadIdx.x == 0) wg_counts[blockIdx.x] = total | (total > DISC_SLOTS ? 0x80000000u : 0u);
}

/* a thread a discovery workgroup: its candidates to their places in stream order (a candidate beyond the arrays' capacity - the
 * launch was sized before the count was known - is not written, and the walk's result says there were more) */
__global__ __launch_bounds__(256) void place_cands_kernel(const uint8_t *__restrict__ stream, const uint32_t *__restrict__ wg_counts, uint64_t nwg,
                                                          const uint64_t *__restrict__ local, const uint64_t *__restrict__ group_base,
                                                          const DiscSlot *__restrict__ slots, const uint64_t *__restrict__ masks,
                                                          uint64_t *__restrict__ cand, uint64_t *__restrict__ cand_len, uint64_t cand_cap)
{
    const uint64_t wg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wg >= nwg) return;
    const uint32_t c = wg_counts[wg];
    uint64_t at = group_base[wg / DISC_SCAN_GROUP] + local[wg];
    if ((c >> 31) == 0u) {
        for (uint32_t j = 0; j < c; j++, at++) {
            if (at >= cand_cap) return;
            const DiscSlot sl = slots[wg * DISC_SLOTS + j];
            cand[at] = sl.offset;
            cand_len[at] = sl.block_len;
        }
        return;
    }
    /* (rare) the workgroup's verdicts, row by row, thread by thread inside a row */
    constexpr uint64_t ROW = (uint64_t)DISC_THREADS * DISC_PER;
    for (int it = 0; it < DISC_ITERS; it++)
        for (uint32_t t = 0; t < DISC_THREADS; t++) {
            uint32_t m = (uint32_t)(masks[wg * DISC_THREADS + t] >> (16 * it)) & 0xffffu;
            while (m) {
                const int k = __builtin_ctz(m);
                m &= m - 1;
                if (at >= cand_cap) return;
                const uint64_t p = wg * DISC_CHUNK + (uint64_t)t * DISC_PER + (uint64_t)it * ROW + (uint64_t)k;
                cand[at] = p;
                cand_len[at] = load_u64_unaligned(stream + p);
                at++;
            }
        }
}

/* The candidates in front of every discovery workgroup, in two levels (round 6; until then ONE workgroup summed the 65 536
 * counts of a GiB: 30 us): workgroup g sums the counts of discovery workgroups [g 1024, (g + 1) 1024) - local[i] = the
 * candidates of its group in front of i -, and the workgroup that finishes last (a ticket) sums the groups' totals:
 * group_base[g], and the candidates found / held in ctrl[]. */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void scan_counts_kernel(const uint32_t *__restrict__ counts, uint64_t n,
                                                              uint64_t *__restrict__ local, uint64_t *__restrict__ group_base,
                                                              uint64_t *__restrict__ group_total, uint64_t *__restrict__ ctrl, uint64_t cand_cap)
{
    static_assert(THREADS == (int)DISC_SCAN_GROUP, "one count a thread");
    __shared__ uint32_t s_part[THREADS / 64];
    __shared__ uint64_t s_part64[THREADS / 64];
    __shared__ uint32_t s_last;
    const uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    const uint32_t c = i < n ? (counts[i] & 0x7fffffffu) : 0u;      /* (bit 31: the workgroup's candidates are in its verdicts, not in its slots) */
    uint32_t total;
    const uint32_t ex = block_excl_scan_u32<THREADS>(c, s_part, total);       /* (a workgroup's candidates: < 2^22) */
    if (i < n) local[i] = ex;
    if (threadIdx.x == 0) {
        group_total[blockIdx.x] = total;
        __threadfence();
        s_last = atomicAdd((unsigned long long *)&ctrl[DISC_TICKET], 1ull) == (unsigned long long)gridDim.x - 1ull ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    /* the groups' totals: a chunk of THREADS at a time (16 GiB of stream are one chunk) */
    uint64_t carry = 0;
    for (uint64_t base = 0; base < gridDim.x; base += THREADS) {
        const uint64_t g = base + threadIdx.x;
        const uint64_t v = g < gridDim.x ? __builtin_nontemporal_load(&group_total[g]) : 0ull;
        uint64_t tot;
        const uint64_t e = block_excl_scan<THREADS, uint64_t>(v, s_part64, tot);
        if (g < gridDim.x) group_base[g] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        ctrl[DISC_FOUND] = carry;
        ctrl[DISC_NCAND] = carry < cand_cap ? carry : cand_cap;
    }
}

/* Where the output of candidate i would start if every candidate were a block of the stream, in
 * order: the exclusive prefix sum of the block_len fields (ONE workgroup; spec_off[ncand] = sum). */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void cand_lens_kernel(const uint64_t *__restrict__ cand_len, const uint64_t *__restrict__ ctrl,
                                                            uint64_t *__restrict__ spec_off)
{
    const uint64_t ncand = uni64(ctrl[DISC_NCAND]);
    const uint64_t total = chunked_excl_scan<THREADS>(ncand, spec_off, [=](uint64_t i) {
        return cand_len[i];
    });
    if (threadIdx.x == 0) spec_off[ncand] = total;
}

/* Decode of one candidate: where does its payload end, and does it decode at all?  Count-only,
 * unless all candidates together fit the output (spec_off[ncand] <= out_cap): then the symbols
 * are written where they belong if every candidate is a real block - the usual case, in which the
 * chain walk afterwards confirms exactly that and nothing has to be decoded twice.
 * Round 3: the candidate goes through the lean decoder first (decode_fast.hpp: verified; it also says where the
 * block's last symbol ends), and only what that cannot vouch for - a damaged block, an unusual tree, a false
 * candidate - through the exact one, which has the last word on status and end: 3.6 -> 1.7 ms per GiB. */
template <int THREADS>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void probe_kernel(const uint8_t *__restrict__ stream, uint64_t avail,
                                                        const uint64_t *__restrict__ cand,
                                                        uint64_t *__restrict__ cand_end,
                                                        int32_t *__restrict__ cand_status,
                                                        const uint64_t *__restrict__ spec_off, uint8_t *__restrict__ out,
                                                        uint64_t out_cap, uint32_t *__restrict__ redo, uint64_t *__restrict__ ctrl)
{
    /* (round 6: the launch is as wide as the arrays - it went out before anybody knew how many candidates there are) */
    const uint64_t ncand = uni64(ctrl[DISC_NCAND]);
    if (blockIdx.x >= ncand) return;
    unsigned long long *const redo_count = (unsigned long long *)&ctrl[DISC_REDO];
    /* Round 5: the lean decoder ONLY; a candidate it cannot vouch for goes on the list of probe_exact_kernel.  (With the exact
     * decoder in the same kernel the probe spilled 5-13 registers and could not take decode_fast's column stage: the two
     * stage forms and the exact decoder inlined side by side.) */
    __shared__ DecShared<THREADS> sh;
    const uint64_t c = uni64(cand[blockIdx.x]);
    const uint64_t block_len = uni64(load_u64_unaligned(stream + c));
    const int tl = (int)(int16_t)uni32((uint32_t)stream[c + 8] | ((uint32_t)stream[c + 9] << 8));
    const uint64_t pay0 = c + HUF_HEADER_FIXED + 2ull * (uint64_t)tl;
    uint64_t end_bits = 0;
    const bool store = uni64(spec_off[ncand]) <= out_cap;
    if (store && block_len != 0 && tl >= 9 && pay0 <= avail) {
        int leaf = -1;
        /* (round 4) the tables from the tree's shape where decode_fast_kernel takes them from it, and the next candidate's
         * offset as a first guess at where this one's payload ends: probe 1.62 -> 1.53 ms per GiB of zipf255, the indexed decoder's time (profiles/r04/raw_stream_kernels.txt) */
        const uint64_t nextc = (blockIdx.x + 1u < ncand) ? uni64(cand[blockIdx.x + 1u]) : 0ull;
        const uint64_t hint = nextc > pay0 ? nextc - pay0 : 0ull;
#ifndef DFAST_NO_REGS
        const int shaped = !(block_len >= DREG_MIN_BLOCK && tl <= HUF_TREE_MAX) ? 0 :
            decode_payload_regs<THREADS>(sh, stream + pay0, avail - pay0, avail - pay0, block_len, out + uni64(spec_off[blockIdx.x]), &end_bits, hint,
                                         [&]() { return dfast_tables_from_tree<THREADS, true, true>(sh, stream + c + HUF_HEADER_FIXED, tl) ? (uni32(sh.l2n) != 0u ? 2 : 1) : 0; });
        if (shaped != 0 ? shaped == 1 /* DREG_OK */
                        : (dec_build_tables<THREADS, true>(sh, stream + c + HUF_HEADER_FIXED, tl, &leaf) == HUFE_OK && leaf < 0 &&
                           decode_payload_dfast<THREADS>(sh, stream + pay0, avail - pay0, avail - pay0, block_len, out + uni64(spec_off[blockIdx.x]), &end_bits, hint))) {
#else
        const bool shaped = block_len >= 32768u && tl <= HUF_TREE_MAX && dfast_tables_from_tree<THREADS, true>(sh, stream + c + HUF_HEADER_FIXED, tl);
        if ((shaped || (dec_build_tables<THREADS, true>(sh, stream + c + HUF_HEADER_FIXED, tl, &leaf) == HUFE_OK && leaf < 0)) &&
            decode_payload_dfast<THREADS>(sh, stream + pay0, avail - pay0, avail - pay0, block_len, out + uni64(spec_off[blockIdx.x]), &end_bits, hint)) {
#endif
            if (threadIdx.x == 0) {
                cand_status[blockIdx.x] = HUFE_OK;
                cand_end[blockIdx.x] = pay0 + ((end_bits + 7) >> 3);
            }
            return;
        }
    }
    if (threadIdx.x == 0) redo[atomicAdd(redo_count, 1ull)] = blockIdx.x;
}

/* The candidates the lean decoder could not vouch for - a damaged block, an unusual tree, a false candidate, and every candidate
 * when the output does not take them all - through the exact decoder, which has the last word on status and end. */
template <int THREADS, bool STORE>
__global__ __launch_bounds__(THREADS, DEC_WAVES_PER_SIMD) void probe_exact_kernel(const uint8_t *__restrict__ stream, uint64_t avail,
                                                        const uint64_t *__restrict__ cand,
                                                        uint64_t *__restrict__ cand_end,
                                                        int32_t *__restrict__ cand_status,
                                                        const uint64_t *__restrict__ spec_off, uint8_t *__restrict__ out,
                                                        uint64_t out_cap, const uint32_t *__restrict__ redo, const uint64_t *__restrict__ ctrl)
{
    __shared__ DecShared<THREADS> sh;
    const uint64_t ncand = uni64(ctrl[DISC_NCAND]);
    const uint64_t n = uni64(ctrl[DISC_REDO]);
    if ((uni64(spec_off[ncand]) <= out_cap) != STORE) return;         /* (the other form of this kernel has the list) */
    for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t k = uni32(redo[i]);
        const uint64_t c = uni64(cand[k]);
        const uint64_t block_len = uni64(load_u64_unaligned(stream + c));
        const int tl = (int)(int16_t)uni32((uint32_t)stream[c + 8] | ((uint32_t)stream[c + 9] << 8));
        const uint64_t pay0 = c + HUF_HEADER_FIXED + 2ull * (uint64_t)tl;
        uint64_t end_bits = 0, produced = 0;
        int err;
        __syncthreads();
        err = decode_block<THREADS, STORE>(sh, stream + c + HUF_HEADER_FIXED, tl, block_len, avail - pay0,
                                           STORE ? out + spec_off[k] : nullptr, &end_bits, &produced);
        if (threadIdx.x == 0) {
            cand_status[k] = err;
            cand_end[k] = pay0 + ((end_bits + 7) >> 3);
        }
    }
}

__global__ void link_kernel(const uint64_t *__restrict__ cand, const uint64_t *__restrict__ cand_end,
                            const int32_t *__restrict__ cand_status, uint64_t *__restrict__ ctrl, uint64_t length,
                            uint32_t *__restrict__ nxt)
{
    const uint64_t ncand = ctrl[DISC_NCAND];
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
    /* (round 6) is the chain the plain one - every candidate a block, each ending where the next begins, the last at the
     * stream's end?  Then walk_kernel has nothing to follow.  Any other link is counted. */
    const bool plain = (i + 1 < ncand) ? r == (uint32_t)(i + 1) : r == LINK_TERMINAL;
    if (!plain || (i == 0 && cand[0] != 0)) atomicAdd((unsigned long long *)&ctrl[DISC_ODD_LINKS], 1ull);
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
#define WALK_THREADS 1024
/* (Round 3: the candidates of a chunk come into LDS with their links, loaded by sixteen waves - the one wave that
 * walks read cand[] from memory inside its loop, one round trip per step of 64: 115 us per GiB of 64 KiB blocks.) */
__global__ __launch_bounds__(WALK_THREADS) void walk_kernel(const uint64_t *__restrict__ cand,
                                                  const uint64_t *__restrict__ cand_end,
                                                  const uint32_t *__restrict__ nxt,
                                                  uint64_t *__restrict__ block_offsets,
                                                  uint64_t *__restrict__ result,
                                                  const uint64_t *__restrict__ spec_off, uint64_t out_cap)
{
    const uint64_t ncand = uni64(result[DISC_NCAND]);
    if (ncand != 0 && uni64(result[DISC_ODD_LINKS]) == 0ull && uni64(result[DISC_FOUND]) == ncand) {
        /* (round 6) the plain chain - what a stream is unless a payload holds the bytes of a header or a block is damaged:
         * validated block j IS candidate j, and the sixteen waves copy the offsets (the wave that walks took 256 steps of
         * 64 links for a GiB of 64 KiB blocks: 59 us) */
        for (uint64_t i = threadIdx.x; i < ncand; i += WALK_THREADS) block_offsets[i] = cand[i];
        if (threadIdx.x == 0) {
            const uint64_t consumed = cand_end[ncand - 1];
            result[0] = ncand;
            result[1] = 0;
            result[2] = 1;
            result[3] = consumed;
            result[4] = spec_off[ncand] <= out_cap ? spec_off[ncand] : ~0ull;
            block_offsets[ncand] = consumed;
        }
        return;
    }
    __shared__ uint32_t s_nxt[WALK_CHUNK];
    __shared__ uint64_t s_cand[WALK_CHUNK];
    __shared__ uint64_t s_cur;
    __shared__ int s_stop;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const bool walker = tid < 64;                      /* the first wave */
    uint64_t cur = 0, m = 0, resume = 0, consumed = 0;
    int complete = 0;
    bool contiguous = true;       /* validated block j is candidate j, for every j so far */
    bool stop = (ncand == 0) || (cand[0] != 0);       /* the stream must start with a header */
    while (!stop) {                                    /* (cur and stop: the same in every thread) */
        const uint64_t base = cur - (cur % WALK_CHUNK);
        const uint64_t top = dmin<uint64_t>(base + WALK_CHUNK, ncand);      /* candidates [base, top) are in LDS */
#pragma unroll
        for (int k = 0; k < WALK_CHUNK / WALK_THREADS; k++) {
            const uint64_t i = (uint64_t)(tid + k * WALK_THREADS);
            if (base + i < top) {
                s_nxt[i] = nxt[base + i];
                s_cand[i] = cand[base + i];
            }
        }
        __syncthreads();
        if (walker) {
            uint64_t c = cur;
            while (c < top) {
                /* the run of plain links that starts at c */
                const uint64_t idx = c + (uint64_t)lane;
                const bool plain = idx < top && s_nxt[idx - base] == (uint32_t)(idx + 1);
                const unsigned long long mask = __ballot(plain);
                const uint32_t run = (~mask == 0ull) ? 64u : (uint32_t)__builtin_ctzll(~mask);
                if ((uint32_t)lane < run) block_offsets[m + (uint64_t)lane] = s_cand[idx - base];
                if (run && c != m) contiguous = false;
                m += run;
                c += run;
                if (run == 64u || c >= top) continue;
                /* one link of another kind.  (Round 4: a link that jumps over a candidate - a block whose payload holds the
                 * bytes of a header - was never followed: hipcc 7.2 compiled the chain of early exits this was into code that
                 * kept `c` for every link below 2^31 and counted m up until block_offsets[] ended;
                 * test_raw_stream_with_a_false_header_inside_a_payload.  One exit now, the link read as a wave-uniform value.) */
                const uint32_t nx = uni32(s_nxt[c - base]);
                const uint64_t here = s_cand[c - base];
                if (nx == LINK_BAD) {
                    resume = here;
                    stop = true;
                } else {
                    if (lane == 0) block_offsets[m] = here;
                    if (c != m) contiguous = false;
                    m++;
                    if (nx == LINK_TERMINAL) {
                        complete = 1;
                        consumed = cand_end[c];
                        stop = true;
                    } else if (nx == LINK_NOTFOUND || (uint64_t)nx <= c) {   /* (the chain only moves forward) */
                        resume = cand_end[c];
                        stop = true;
                    } else {
                        c = (uint64_t)nx;
                    }
                }
                if (stop) break;
            }
            if (lane == 0) {
                s_cur = c;
                s_stop = stop ? 1 : 0;
            }
        }
        __syncthreads();                               /* the walker is done with this chunk; where it stands */
        cur = s_cur;
        stop = s_stop != 0;
    }
    if (!walker) return;
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

}  // namespace hufgpu
