// k_stereobm_c4.hip -- dispatch of the column-sum SAD matcher (kernel: k_stereobm_c4_impl.h) by block radius.
//
// The kernel is instantiated for every block radius 2..15 (block 5..31: all the u16 window sums allow at cap 31) and every
// disparity count that is a multiple of 16 up to 256 -- the whole range of the reference's cfg/GPU.cfg including its constructor
// default createStereoBM(48, 19) (reference src/GPUStereoProcessor.cpp:18-20).  The instantiations are spread over seven translation
// units (k_stereobm_c4_r{a..g}.hip) so that they compile in parallel.
#include "gip_internal.h"
#include "bm_common.h"

int gip_bm_c4_launch_ra(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_rb(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_rc(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_rd(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_re(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_rf(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
int gip_bm_c4_launch_rg(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);

bool gip_bm_c4_can(const BmArgs& a)
{
#ifdef NO_C4
    return false;
#endif
    // aligned dword staging needs 4-byte aligned rows
    if ((a.pitch & 3) || ((uintptr_t)a.L & 3) || ((uintptr_t)a.R & 3) || (a.img_bs & 3)) return false;
    if (a.cend <= a.cbeg) return false;
    if (a.ndisp < 16 || a.ndisp > 256 || a.ndisp % 16) return false;
    if (a.uniq >= 200) return false;   // the kernel's multiply-shift form of  ms * uniq / 100  is exact below 200 (OpenCV's range is 5..15)
    return a.wsz2 >= 2 && a.wsz2 <= 15;
}

int gip_bm_c4_launch(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled)
{
    *handled = false;
    if (!gip_bm_c4_can(a)) return GIP_OK;
    if (a.wsz2 <= 4) return gip_bm_c4_launch_ra(a, batch, st, ctx, handled);
    if (a.wsz2 <= 6) return gip_bm_c4_launch_rb(a, batch, st, ctx, handled);
    if (a.wsz2 <= 8) return gip_bm_c4_launch_rc(a, batch, st, ctx, handled);
    if (a.wsz2 <= 10) return gip_bm_c4_launch_rd(a, batch, st, ctx, handled);
    if (a.wsz2 <= 12) return gip_bm_c4_launch_re(a, batch, st, ctx, handled);
    if (a.wsz2 <= 14) return gip_bm_c4_launch_rf(a, batch, st, ctx, handled);
    return gip_bm_c4_launch_rg(a, batch, st, ctx, handled);
}
