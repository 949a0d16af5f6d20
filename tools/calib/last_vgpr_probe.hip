/* last_vgpr_probe.hip - does a 64-bit VALU shift read its 32-bit shift amount correctly when that amount
 * sits in the LAST VGPR the wave has been allocated?
 *
 * Found while root-causing pack_kernel's wrong payload bits in the 7-waves-per-SIMD build:
 * there the register allocator had put code[5] into v71 of a 72-VGPR kernel and
 * `v_lshlrev_b64 v[14:15], v71, v[14:15]` shifted by (lane id) instead of by v71[5:0] in some waves.
 *
 * Each test kernel pins the shift amount into v<LAST> by inline asm; the kernel's VGPR allocation is
 * LAST+1 (PAD = 0) or more (PAD > 0: v<LAST+PAD> is clobbered too, so v<LAST> is not the last one).
 * Every wave records its HW_ID so that failing waves can be mapped to SIMD slots.
 *
 *   hipcc -O2 --offload-arch=gfx950 tools/calib/last_vgpr_probe.hip -o tools/calib/last_vgpr_probe && tools/calib/last_vgpr_probe
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>

#define STR2(x) #x
#define STR(x) STR2(x)

enum { OP_LSHL64 = 0, OP_LSHR64 = 1, OP_ASHR64 = 2, OP_LSHL32 = 3, OP_MAD64 = 4, OP_VAL64 = 5, OP_LSHLADD = 6 };

__device__ __forceinline__ uint64_t want_of(int op, uint64_t x, uint32_t sh)
{
    if (op == OP_LSHL64) return x << sh;
    if (op == OP_LSHR64) return x >> sh;
    if (op == OP_ASHR64) return (uint64_t)((int64_t)x >> sh);
    if (op == OP_LSHL32) return (uint32_t)((uint32_t)x << sh);
    if (op == OP_VAL64) return ((uint64_t)sh << 32 | sh) << (sh & 15);     /* the VALUE is the pair v[LAST-1:LAST] */
    if (op == OP_LSHLADD) return (x << (sh & 3)) + x;
    return (uint64_t)sh * sh + x;
}

/* LASTREG / PADREG are register names ("v71"); PADREG == LASTREG means no padding */
#define DEFINE_PROBE(NAME, LASTREG, PADREG, OP, ASMTEXT)                                                        \
    __global__ __launch_bounds__(256) void NAME(uint64_t *out, uint32_t *hwid, int iters)                       \
    {                                                                                                           \
        const uint32_t tid = threadIdx.x;                                                                       \
        const uint64_t gid = (uint64_t)blockIdx.x * 256 + tid;                                                  \
        uint64_t x = 0x0123456789abcdefull ^ (gid * 0x9E3779B97F4A7C15ull);                                     \
        const uint32_t sh = (uint32_t)((gid * 7 + 3) % 29) + 1;                                                 \
        uint64_t bad = 0, r = 0;                                                                                \
        for (int it = 0; it < iters; it++) {                                                                    \
            uint32_t r32 = 0;                                                                                   \
            asm volatile("v_mov_b32 " LASTREG ", %2\n s_nop 4\n " ASMTEXT "\n s_nop 1"                          \
                         : "=&v"(r), "+v"(r32) : "v"(sh), "v"(x), "v"((uint32_t)x) : LASTREG, PADREG, "vcc");                     \
            if (OP == OP_LSHL32) r = r32;                                                                       \
            const uint64_t want = want_of(OP, x, sh);                                                           \
            if (r != want && bad == 0) bad = r | 1ull << 63;                                                    \
            x = x * 6364136223846793005ull + 1442695040888963407ull;                                            \
        }                                                                                                       \
        out[2 * gid] = bad;                                                                                     \
        out[2 * gid + 1] = x;                                                                                   \
        if ((tid & 63) == 0) {                                                                                  \
            uint32_t id;                                                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));                                    \
            hwid[2 * (gid >> 6)] = id;                                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_GPR_ALLOC)" : "=s"(id));                                \
            hwid[2 * (gid >> 6) + 1] = id;                                                                      \
        }                                                                                                       \
    }

typedef void (*probe_fn)(uint64_t *, uint32_t *, int);

static int run(const char *name, probe_fn fn, int blocks, int iters)
{
    uint64_t *d_out; uint32_t *d_hw;
    const size_t n = (size_t)blocks * 256;
    (void)hipMalloc(&d_out, n * 16); (void)hipMalloc(&d_hw, n / 64 * 8);
    (void)hipMemset(d_out, 0xff, n * 16);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void *)fn);
    fn<<<blocks, 256>>>(d_out, d_hw, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return -1; }
    std::vector<uint64_t> out(2 * n); std::vector<uint32_t> hw(n / 64 * 2);
    (void)hipMemcpy(out.data(), d_out, n * 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hw.data(), d_hw, n / 64 * 8, hipMemcpyDeviceToHost);
    size_t badlanes = 0, badwaves = 0;
    std::map<uint32_t, size_t> by_slot, all_slot;      /* VGPR base of the wave (HW_REG_GPR_ALLOC bits 5:0, in allocation granules) */
    uint32_t vsize = 0;
    int shown = 0;
    for (size_t w = 0; w < n / 64; w++) {
        bool wb = false;
        all_slot[hw[2 * w + 1] & 0x3f]++; vsize = (hw[2 * w + 1] >> 8) & 0x3f;
        for (int l = 0; l < 64; l++) {
            const size_t g = w * 64 + l;
            if (out[2 * g]) {
                badlanes++; wb = true;
                if (shown < 3) { printf("    blk %zu tid %zu: first wrong result %016llx hwid %08x gpr_alloc %08x\n", g / 256, g % 256, (unsigned long long)out[2 * g], hw[2 * w], hw[2 * w + 1]); shown++; }
            }
        }
        if (wb) { badwaves++; by_slot[hw[2 * w + 1] & 0x3f]++; }
    }
    printf("%-28s vgprs %3d: bad lanes %8zu, bad waves %6zu of %zu; vgpr_size field %u; failing VGPR bases (base:bad/all):", name, fa.numRegs, badlanes, badwaves, n / 64, vsize);
    for (auto &kv : by_slot) printf(" %u:%zu/%zu", kv.first, kv.second, all_slot[kv.first]);
    printf("   [all bases:");
    for (auto &kv : all_slot) printf(" %u", kv.first);
    printf("]\n");
    (void)hipFree(d_out); (void)hipFree(d_hw);
    return badlanes != 0;
}

#define LSHL64(R) "v_lshlrev_b64 %0, " R ", %3"
#define LSHR64(R) "v_lshrrev_b64 %0, " R ", %3"
#define ASHR64(R) "v_ashrrev_i64 %0, " R ", %3"
#define LSHL32(R) "v_lshlrev_b32 %1, " R ", %4"
#define MAD64(R)  "v_mad_u64_u32 %0, vcc, " R ", %2, %3"
#define MAD64B(R) "v_mad_u64_u32 %0, vcc, %2, " R ", %3"
#define VAL64(RLO, R) "v_mov_b32 " RLO ", %2\n v_and_b32 %1, 15, %2\n s_nop 4\n v_lshlrev_b64 %0, %1, v[70:71]"
#define LSHLADD(R) "v_and_b32 " R ", 3, " R "\n s_nop 4\n v_lshl_add_u64 %0, %3, " R ", %3"
DEFINE_PROBE(p_l71, "v71", "v71", OP_LSHL64, LSHL64("v71"))
DEFINE_PROBE(p_l71_p72, "v71", "v72", OP_LSHL64, LSHL64("v71"))
DEFINE_PROBE(p_l71_p79, "v71", "v79", OP_LSHL64, LSHL64("v71"))
DEFINE_PROBE(p_l70, "v70", "v70", OP_LSHL64, LSHL64("v70"))
DEFINE_PROBE(p_l63, "v63", "v63", OP_LSHL64, LSHL64("v63"))
DEFINE_PROBE(p_l63_p64, "v63", "v64", OP_LSHL64, LSHL64("v63"))
DEFINE_PROBE(p_l79, "v79", "v79", OP_LSHL64, LSHL64("v79"))
DEFINE_PROBE(p_l55, "v55", "v55", OP_LSHL64, LSHL64("v55"))
DEFINE_PROBE(p_l47, "v47", "v47", OP_LSHL64, LSHL64("v47"))
DEFINE_PROBE(p_l39, "v39", "v39", OP_LSHL64, LSHL64("v39"))
DEFINE_PROBE(p_l31, "v31", "v31", OP_LSHL64, LSHL64("v31"))
DEFINE_PROBE(p_l23, "v23", "v23", OP_LSHL64, LSHL64("v23"))
DEFINE_PROBE(p_l95, "v95", "v95", OP_LSHL64, LSHL64("v95"))
DEFINE_PROBE(p_l127, "v127", "v127", OP_LSHL64, LSHL64("v127"))
DEFINE_PROBE(p_r71, "v71", "v71", OP_LSHR64, LSHR64("v71"))
DEFINE_PROBE(p_a71, "v71", "v71", OP_ASHR64, ASHR64("v71"))
DEFINE_PROBE(p_s71, "v71", "v71", OP_LSHL32, LSHL32("v71"))
DEFINE_PROBE(p_m71, "v71", "v71", OP_MAD64, MAD64("v71"))
DEFINE_PROBE(p_r71_p72, "v71", "v72", OP_LSHR64, LSHR64("v71"))
DEFINE_PROBE(p_m71b, "v71", "v71", OP_MAD64, MAD64B("v71"))
DEFINE_PROBE(p_val7071, "v71", "v70", OP_VAL64, VAL64("v70", "v71"))
DEFINE_PROBE(p_la71, "v71", "v71", OP_LSHLADD, LSHLADD("v71"))
DEFINE_PROBE(p_l69, "v69", "v71", OP_LSHL64, LSHL64("v69"))

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 8192, iters = argc > 2 ? atoi(argv[2]) : 2000;
    int rc = 0;
#define RUN(F) rc |= run(#F, F, blocks, iters)
    RUN(p_l71); RUN(p_l71_p72); RUN(p_l71_p79); RUN(p_l70); RUN(p_l63); RUN(p_l63_p64); RUN(p_l79);
    RUN(p_l55); RUN(p_l47); RUN(p_l39); RUN(p_l31); RUN(p_l23); RUN(p_l95); RUN(p_l127);
    RUN(p_r71); RUN(p_a71); RUN(p_s71); RUN(p_m71); RUN(p_r71_p72); RUN(p_m71b); RUN(p_val7071); RUN(p_la71); RUN(p_l69);
    printf("last_vgpr_probe: %s\n", rc ? "SOME CONFIGURATION FAILED" : "all configurations right");
    return 0;
}
