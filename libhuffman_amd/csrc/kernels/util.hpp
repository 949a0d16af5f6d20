/* util.hpp - lane helpers, streaming loads/stores, block-wide scans.
   Part of hufgpu_kernels.hip (one translation unit, gfx950 only). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hufgpu_common.h"

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
/* "this wave-uniform value is needed HERE": keeps the compiler from sinking the load of it to its first use
 * (scalar loads requested together wait once; one at a time they wait once each) */
__device__ __forceinline__ void pin_uniform(uint64_t v)
{
    asm volatile("" :: "s"((uint32_t)v), "s"((uint32_t)(v >> 32)));
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
    /* (non-temporal - measured: histogram of 1 GiB 0.202 -> 0.169 ms, pack 0.043 -> 0.030 ms) */
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream16(uint4 *p, uint4 v)
{
    /* (non-temporal - measured: one-symbol decode (a fill) 0.260 -> 0.204 ms per GiB) */
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p));
}
/* The compressed stream is written with the default policy: it is what a decode that follows
 * reads, and a stream that fits the 256 MiB Infinity Cache is then served from there (measured on
 * config 2: decode 0.250 -> 0.21 ms when the 128 MiB stream is still cached). */
__device__ __forceinline__ void store_pack16(uint4 *p, uint4 v)
{
    *p = v;
}

__device__ __forceinline__ uint32_t wave_xor_any(uint32_t v, int o);
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)      /* mask: a power of two, constant after unrolling */
{
    return ((uint64_t)wave_xor_any((uint32_t)(v >> 32), mask) << 32) | wave_xor_any((uint32_t)v, mask);
}
__device__ __forceinline__ uint32_t shfl_xor_key(uint32_t v, int mask) { return (uint32_t)__shfl_xor((int)v, mask); }
__device__ __forceinline__ uint64_t shfl_xor_key(uint64_t v, int mask) { return shfl_xor_u64(v, mask); }

/* Inclusive prefix sum over the wave by DPP: shifts inside the rows of 16 lanes, then lane 15 of a
 * row into the next row, then lane 31 into the upper half (six VALU instructions, no LDS, and no
 * per-step lane address to keep - the ds_bpermute form of __shfl_up had its six address registers
 * hoisted out of loops, spilled, and reloaded from scratch in every iteration). */
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   /* row_bcast:15 -> rows 1, 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   /* row_bcast:31 -> rows 2, 3 */
    return v;
}
/* For every lane the maximum over the lanes IN FRONT of it (0 for lane 0; values are unsigned, 0 is neutral): the DPP
 * steps of wave_incl_scan_u32 with v_max, then one wave_shr. */
__device__ __forceinline__ uint32_t wave_excl_max_u32(uint32_t v)
{
#define HUF_MAX_DPP_(ctrl, rmask, bc) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, bc); v = v > o_ ? v : o_; }
    HUF_MAX_DPP_(0x111, 0xf, true)      /* row_shr:1 */
    HUF_MAX_DPP_(0x112, 0xf, true)      /* row_shr:2 */
    HUF_MAX_DPP_(0x114, 0xf, true)      /* row_shr:4 */
    HUF_MAX_DPP_(0x118, 0xf, true)      /* row_shr:8 */
    HUF_MAX_DPP_(0x142, 0xa, false)     /* row_bcast:15 -> rows 1, 3 */
    HUF_MAX_DPP_(0x143, 0xc, false)     /* row_bcast:31 -> rows 2, 3 */
#undef HUF_MAX_DPP_
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);         /* wave_shr:1, lane 0 <- 0 */
}
/* the two 16-bit halves' maxima in one instruction */
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(uint32_t, r);
}

/* Value of lane (lane ^ D), D a power of two: DPP inside a row of 16 lanes (quad_perm for 1 and 2,
 * two row rotations and a select for 4, one rotation for 8), ds_swizzle for 16 (no address register),
 * ds_bpermute only for 32.  __shfl_xor is a ds_bpermute with a computed address for every distance:
 * an LDS round trip per exchange - 84 of them per 256-key bitonic sort (tree.hpp). */
template <int D>
__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v)
{
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);        /* quad_perm:[1,0,3,2] */
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   /* quad_perm:[2,3,0,1] */
    else if constexpr (D == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);             /* row_ror:4  */
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12C, 0xf, 0xf, false);             /* row_ror:12 */
        return (lane_id() & 4) ? a : b;
    } else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); /* row_ror:8 */
    else if constexpr (D == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                   /* bit mode: xor 16, and 31 */
    else return (uint32_t)__shfl_xor((int)v, D);
}

/* the same for a distance that is a constant after unrolling (for (o = 1; o < 64; o <<= 1) ...) */
__device__ __forceinline__ uint32_t wave_xor_any(uint32_t v, int o)
{
    return o == 1 ? wave_xor_u32<1>(v) : o == 2 ? wave_xor_u32<2>(v) : o == 4 ? wave_xor_u32<4>(v)
         : o == 8 ? wave_xor_u32<8>(v) : o == 16 ? wave_xor_u32<16>(v) : wave_xor_u32<32>(v);
}
/* value of the lane below (lane 0 keeps its own): wave_shr:1 */
__device__ __forceinline__ uint32_t wave_up1_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}

/* value of a lane given by a wave-uniform index (v_readlane: no LDS, no address register) */
__device__ __forceinline__ uint32_t wave_lane_u32(uint32_t v, uint32_t uniform_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)uniform_lane);
}

/* Exclusive prefix sum of 32-bit values over the workgroup: DPP inside the waves, one LDS round
 * across them.  s_part needs THREADS/64 words. */
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *s_part, uint32_t &total)
{
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) {
        const uint32_t x = s_part[i];
        if (i < wave) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

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

}  // namespace hufgpu
