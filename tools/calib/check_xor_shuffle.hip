// Checks the DPP / ds_swizzle forms of "value of lane ^ D" against __shfl_xor on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/calib/check_xor_shuffle.hip -o /tmp/check_xor && /tmp/check_xor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../libhuffman_amd/csrc/kernels/util.hpp"
using namespace hufgpu;
__global__ void k(uint32_t *out)
{
    const uint32_t v = threadIdx.x * 2654435761u + 12345u;
    out[0 * 64 + threadIdx.x] = wave_xor_u32<1>(v) ^ (uint32_t)__shfl_xor((int)v, 1);
    out[1 * 64 + threadIdx.x] = wave_xor_u32<2>(v) ^ (uint32_t)__shfl_xor((int)v, 2);
    out[2 * 64 + threadIdx.x] = wave_xor_u32<4>(v) ^ (uint32_t)__shfl_xor((int)v, 4);
    out[3 * 64 + threadIdx.x] = wave_xor_u32<8>(v) ^ (uint32_t)__shfl_xor((int)v, 8);
    out[4 * 64 + threadIdx.x] = wave_xor_u32<16>(v) ^ (uint32_t)__shfl_xor((int)v, 16);
    out[5 * 64 + threadIdx.x] = wave_xor_u32<32>(v) ^ (uint32_t)__shfl_xor((int)v, 32);
    const uint32_t up = wave_up1_u32(v), ref = (uint32_t)__shfl_up((int)v, 1);     /* (both by all lanes: DPP reads no disabled lane) */
    out[6 * 64 + threadIdx.x] = up ^ (threadIdx.x ? ref : v);
    out[7 * 64 + threadIdx.x] = wave_incl_scan_u32(threadIdx.x * 3u + 1u) ^ (uint32_t)(3u * (threadIdx.x * (threadIdx.x + 1) / 2) + threadIdx.x + 1u);
}
int main()
{
    uint32_t *d, h[8 * 64];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 8 * 64; i++) if (h[i]) { bad++; if (bad < 8) printf("mismatch test %d lane %d\n", i / 64, i % 64); }
    printf(bad ? "FAILED %d\n" : "xor shuffles ok\n", bad);
    return bad != 0;
}
