/* Issue rate of a few VALU instructions the decode loop could use, against v_add_u32: a dependent chain per wave,
 * eight waves a SIMD (the chain's latency is hidden, the SIMD's issue rate shows).
 * build: hipcc -O2 --offload-arch=gfx950 tools/calib/valu_rate.hip -o /tmp/valu_rate */
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void chain(int *out, int a, int iters)
{
    int r = (int)threadIdx.x, b = a + (int)threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_and_b32 %0, 0xffe, %0" : "+v"(r) : );) }
        if (OP == 4) { REP16(asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "s"(a));) }
        if (OP == 5) { REP16(asm volatile("v_or_b32 %0, %1, %0" : "+v"(r) : "v"(b));) }
        if (OP == 6) { REP16(asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_lshrrev_b32 %0, 19, %0" : "+v"(r) : );) }
        if (OP == 9) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r) : );) }
        if (OP == 10) { REP16(asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r) : "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));) }
        if (OP == 12) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r) : "v"(b));) }
        if (OP == 13) { REP16(asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(r) : "v"(b));) }
        if (OP == 14) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(b));) }
        if (OP == 15) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "s"(0xff00), "v"(b));) }
        if (OP == 16) { REP16(asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));) }
        if (OP == 17) { REP16(asm volatile("v_bfe_u32 %0, %0, 20, 11" : "+v"(r) : );) }
        if (OP == 18) { REP16(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(a), "v"(b));) }
        if (OP == 19) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "s"(0x05040100));) }
        if (OP == 20) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));) }
        if (OP == 21) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));) }
        if (OP == 22) { REP16(asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(r) : "v"(b));) }
        if (OP == 23) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(r) : "v"(b));) }
        if (OP == 24) { REP16(asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(b));) }
        if (OP == 25) { REP16(asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));) }
        if (OP == 26) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");) }
        if (OP == 27) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 28) { REP16(asm volatile("v_max_u16 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 29) { REP16(asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 30) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 31) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 32) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));) }
        if (OP == 33) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 34) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(r) : "v"(b) : "vcc");) }
        if (OP == 35) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r) : "v"(b) : "vcc");) }
        if (OP == 36) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(a));) }
        if (OP == 37) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b));) }
        if (OP == 38) { REP16(asm volatile("v_readlane_b32 s20, %0, 5" : "+v"(r) :  : "s20");) }
        if (OP == 39) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(r) : );) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> static int run(const char *name, int *d)
{
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    chain<OP><<<blocks, 256>>>(d, 3, 16);
    CK(hipEventRecord(e0));
    chain<OP><<<blocks, 256>>>(d, 3, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    /* wave instructions a SIMD: blocks * 4 waves / 1024 SIMDs * iters * 16 */
    const double per_simd = (double)blocks * 4 / 1024 * iters * 16;
    printf("%-18s %.3f ms, %.2f ns a wave instruction and SIMD\n", name, ms, ms * 1e6 / per_simd);
    return 0;
}
int main(void)
{
    int *d; CK(hipMalloc((void **)&d, 256 * 8 * 256 * 4));
    run<0>("(warm-up)", d);
    run<0>("v_add_u32", d);
    run<1>("v_sub_u32", d);
    run<2>("v_and_b32", d);
    run<3>("v_and_b32 lit", d);
    run<4>("v_and_b32 sgpr", d);
    run<5>("v_or_b32", d);
    run<6>("v_xor_b32", d);
    run<7>("v_mov_b32", d);
    run<8>("v_lshrrev_b32 imm", d);
    run<9>("v_lshlrev_b32 imm", d);
    run<10>("v_lshlrev_b32 v", d);
    run<11>("v_add3_u32", d);
    run<12>("v_lshl_add_u32", d);
    run<13>("v_add_lshl_u32", d);
    run<14>("v_lshl_or_b32", d);
    run<15>("v_and_or_b32", d);
    run<16>("v_or3_b32", d);
    run<17>("v_bfe_u32", d);
    run<18>("v_bfi_b32", d);
    run<19>("v_perm_b32", d);
    run<20>("v_alignbit_b32", d);
    run<21>("v_alignbyte_b32", d);
    run<22>("v_sub_u32_sdwa", d);
    run<23>("v_add_u32_sdwa", d);
    run<24>("v_add_u32_dpp", d);
    run<25>("v_dot4c_i32_i8", d);
    run<26>("v_cndmask_b32", d);
    run<27>("v_min_u32", d);
    run<28>("v_max_u16", d);
    run<29>("v_pk_max_u16", d);
    run<30>("v_pk_add_u16", d);
    run<31>("v_mul_u32_u24", d);
    run<32>("v_mad_u32_u24", d);
    run<33>("v_mul_lo_u32", d);
    run<34>("v_cmp_lt_u32", d);
    run<35>("v_add_co_u32", d);
    run<36>("v_fma_f32", d);
    run<37>("v_add_f32", d);
    run<38>("v_readlane_b32", d);
    run<39>("v_mbcnt_lo", d);
    return 0;
}
