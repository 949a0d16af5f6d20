// Calibration of per-instruction issue cost on gfx950 (MI355X): ITER iterations of 16 copies of one
// instruction on every wave of a grid that fills the chip with 8 waves per SIMD (or 1 wave per SIMD).
// "dep" = each instruction depends on the previous one, "ind" = 4 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 20000
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(name, body)                                                        \
__global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed)         \
{                                                                                 \
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = b ^ 0x55u, s = a & 31u;   \
    uint32_t a1 = a + 1, a2 = a + 2, a3 = a + 3;                                  \
    uint64_t q = ((uint64_t)a << 32) | b;                                         \
    for (int i = 0; i < ITER; i++) { body }                                       \
    out[blockIdx.x * 256 + threadIdx.x] = a + a1 + a2 + a3 + b + c + s + (uint32_t)q + (uint32_t)(q >> 32); \
}
#define DEP(name, ins)  KERNEL(name, REP16(asm volatile(ins : "+v"(a) : "v"(b), "v"(s) : "vcc", "s20", "s21", "scc");))
#define IND(name, ins)  KERNEL(name, REP4(asm volatile(ins : "+v"(a) : "v"(b), "v"(s) : "vcc", "s20", "s21", "scc"); \
                                         asm volatile(ins : "+v"(a1) : "v"(b), "v"(s) : "vcc", "s20", "s21", "scc"); \
                                         asm volatile(ins : "+v"(a2) : "v"(b), "v"(s) : "vcc", "s20", "s21", "scc"); \
                                         asm volatile(ins : "+v"(a3) : "v"(b), "v"(s) : "vcc", "s20", "s21", "scc");))
#define BOTH(n, ins) DEP(n##_dep, ins) IND(n##_ind, ins)

BOTH(add,      "v_add_u32 %0, %0, %1")
BOTH(or_,      "v_or_b32 %0, %0, %1")
BOTH(sub,      "v_sub_u32 %0, %0, %1")
BOTH(lshr,     "v_lshrrev_b32 %0, 3, %0")
BOTH(and_,     "v_and_b32 %0, 0x1ffe, %0")
BOTH(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
BOTH(bfe,      "v_bfe_u32 %0, %0, 8, 6")
BOTH(cmp,      "v_cmp_lt_u32 vcc, %0, %1")
BOTH(cmp_s,    "v_cmp_lt_u32 s[20:21], %0, %1")
BOTH(cmpcnd,   "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
BOTH(cmpcnd64, "v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]")
BOTH(cnd,      "v_cndmask_b32 %0, %0, %1, vcc")
BOTH(cnds,     "v_cndmask_b32_e64 %0, %0, %1, s[22:23]")      /* a mask no VALU instruction wrote lately */
BOTH(cndadd,   "v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %0, %1")
BOTH(bfi,      "v_bfi_b32 %0, %2, %0, %1")
BOTH(adds,     "v_add_u32 %0, s22, %0")
BOTH(min,      "v_min_u32 %0, %0, %1")
BOTH(lshladd,  "v_lshl_add_u32 %0, %0, 2, %1")
BOTH(add3,     "v_add3_u32 %0, %0, %1, %2")
BOTH(mad24,    "v_mad_u32_u24 %0, %0, %1, %2")
BOTH(perm,     "v_perm_b32 %0, %0, %1, %2")
BOTH(sdwa,     "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
BOTH(dppmov,   "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
BOTH(dppmin,   "s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
BOTH(readlane, "v_readlane_b32 s20, %0, 3")
BOTH(sadd,     "s_add_u32 s20, s20, 1")
BOTH(vs_mix,   "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1")
BOTH(cbranch,  "s_cmp_eq_u32 s20, 77\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:")
BOTH(saveexec, "v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]")

static double clk_hz, simds; static uint32_t *d; static hipEvent_t e0, e1;
template <typename K> static void run(const char *name, K k, int grid, int per)
{
    k<<<grid, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid, 256>>>(d, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)grid * 4 * ITER * 16 * per;
    printf("%-14s %3d waves/SIMD %8.3f ms  %6.2f cycles per wave-instruction\n", name, grid / 256, ms, ms * 1e-3 * clk_hz * simds / inst);
}
#define RUN(n, per) run(#n "_dep", n##_dep, cus * 8, per); run(#n "_ind", n##_ind, cus * 8, per); run(#n "_dep", n##_dep, cus, per); run(#n "_ind", n##_ind, cus, per);

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; simds = cus * 4.0; clk_hz = p.clockRate * 1e3;
    (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%d CUs, %.0f MHz; cycles = time x clock x SIMDs / wave-instructions\n", cus, clk_hz / 1e6);
    RUN(add, 1) RUN(or_, 1) RUN(sub, 1) RUN(lshr, 1) RUN(and_, 1) RUN(alignbit, 1) RUN(bfe, 1) RUN(cmp, 1) RUN(cmp_s, 1)
    RUN(cmpcnd, 2) RUN(cmpcnd64, 2) RUN(cnd, 1) RUN(cnds, 1) RUN(cndadd, 2) RUN(bfi, 1) RUN(adds, 1) RUN(min, 1) RUN(lshladd, 1) RUN(add3, 1) RUN(mad24, 1) RUN(perm, 1) RUN(sdwa, 1)
    RUN(dppmov, 1) RUN(dppmin, 1) RUN(readlane, 1) RUN(sadd, 1) RUN(vs_mix, 2) RUN(cbranch, 3) RUN(saveexec, 4)
    return 0;
}
