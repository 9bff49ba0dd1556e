// What do v_sub_u32_dpp / v_subrev_u32_dpp compute on gfx950?  (probe for k_stereobm_c4.hip's fused DPP window sums)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
__global__ void k(u32* out)
{
    const u32 lane = threadIdx.x;
    u32 x = 1000 + lane, acc = 5000000 + 100 * lane, r0, r1, r2, r3;
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r0) : "v"(x), "v"(acc));
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r1) : "v"(x), "v"(acc));
    asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r2) : "v"(x), "v"(acc));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r3) : "v"(x), "v"(acc));
    out[lane * 4 + 0] = r0; out[lane * 4 + 1] = r1; out[lane * 4 + 2] = r2; out[lane * 4 + 3] = r3;
}
int main()
{
    u32* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    u32 h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 5, 14, 15, 16, 17}) {
        long x = 1000 + l, acc = 5000000 + 100 * l, xm1 = (l % 16) ? 1000 + l - 1 : 0, xp2 = (l % 16) < 14 ? 1000 + l + 2 : 0;
        printf("lane %2d: subrev_shr1 %ld (acc-x[l-1]=%ld, x[l-1]-acc=%ld, acc[l-1]-x=%ld)  sub_shr1 %ld (x[l-1]-acc=%ld)  add_shr1 %ld (=%ld)  subrev_shl2 %ld (acc-x[l+2]=%ld)\n", l,
               (long)(int)h[l * 4], acc - xm1, xm1 - acc, ((l % 16) ? acc - 100 : 0) - x, (long)(int)h[l * 4 + 1], xm1 - acc, (long)(int)h[l * 4 + 2], acc + xm1, (long)(int)h[l * 4 + 3], acc - xp2);
    }
    return 0;
}
