/* offsets.hpp - block offsets in the output stream: scan_sizes_kernel and the in-kernel two-level prefix sums.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"
#include "util.hpp"

namespace hufgpu {

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
 * announces - tests/stress/soak.py caught exactly that as one wrong block index in ~6 000 runs with
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

}  // namespace hufgpu
