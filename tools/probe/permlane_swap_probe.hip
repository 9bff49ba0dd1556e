// permlane_swap_probe.hip -- what v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950, as used by quarters_reduce() in
// csrc/k_stereobm_c4_impl.h.  build+run: hipcc --offload-arch=gfx950 -O2 tools/probe/permlane_swap_probe.hip -o /tmp/plsw && /tmp/plsw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o)
{
    const unsigned x = threadIdx.x;              // lane id as the payload
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = h[0]; o[192 + threadIdx.x] = h[1];
    unsigned s = x;                              // butterfly sum over lanes t, t+16, t+32, t+48
    auto a = __builtin_amdgcn_permlane16_swap(s, s, false, false); s = a[0] + a[1];
    auto b = __builtin_amdgcn_permlane32_swap(s, s, false, false); s = b[0] + b[1];
    o[256 + threadIdx.x] = s;
}
int main()
{
    unsigned *d, hbuf[320];
    hipMalloc(&d, sizeof hbuf);
    k<<<1, 64>>>(d);
    hipMemcpy(hbuf, d, sizeof hbuf, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap result[0] (vdst')", "permlane16_swap result[1] (src')", "permlane32_swap result[0]", "permlane32_swap result[1]"};
    for (int v = 0; v < 4; v++) {
        printf("%s: source lane per 16-lane row start:", names[v]);
        for (int row = 0; row < 4; row++) printf(" row%d<-lane%u", row, hbuf[v * 64 + row * 16]);
        printf("\n");
    }
    int bad = 0;
    for (int l = 0; l < 64; l++) bad += hbuf[256 + l] != (unsigned)(4 * (l & 15) + 96);
    printf("butterfly sum over the four quarters: %s\n", bad ? "WRONG" : "correct in all 64 lanes");
    return bad != 0;
}
