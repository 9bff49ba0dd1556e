// k_stereobm_wide.hip -- fallback block matcher for windows whose SAD does not fit 16 bits
// (2 * prefilterCap * block^2 >= 65535, i.e. block > 31 at cap 31 or block > 21 at cap 63; cv::StereoBM allows block up to 255).
//
// Same semantics as the other matchers (cv::StereoBM::compute on prefiltered images, reference
// src/GPUStereoProcessor.cpp:312-321), every output column with OpenCV's clamped-column rule, 32-bit window sums.
// Correctness first: one masked v_mqsad_pk_u16_u8 per (window column, 4 disparities) builds the ROW sums as packed u16
// (a row of <= 255 pixels of <= 126 each stays below 65536), which are widened into 32-bit vertical running sums; the
// entering and the leaving row are re-read from global memory each step (two LDS row buffers, no ring -- a ring of
// 2r+1 = 255 rows would not fit in LDS).  Cost is O(block) per (pixel, disparity): ~0.2-1 ms per 1242x375 frame.
#include "gip_internal.h"
#include "bm_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void __launch_bounds__(256) k_bm_wide(BmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    const int r = a.wsz2, blockw = 2 * r + 1;
    const int S = a.ndisp / 16, TC = a.TC;
    const int tc = threadIdx.x % TC, s = threadIdx.x / TC;
    const int c0 = blockIdx.x * TC;
    const int ncols = min(TC, a.width1 - c0);
    const int lwp = (TC + 2 * r + 4 + 15) & ~15;
    const int rwp = (TC + 2 * r + a.ndisp + 16 + 15) & ~15;
    const int cspb = 4 * a.ndisp + 16;
    uint8_t* lrows = smem_raw;                           // 2 x lwp  (0: entering row, 1: leaving row)
    uint8_t* rrows = lrows + 2 * lwp;                    // 2 x rwp
    uint8_t* cs = rrows + 2 * rwp;                       // TC x cspb, u32 costs of the current row
    u64* wta = (u64*)(cs + (size_t)TC * cspb);           // S x TC
    u32* m2 = (u32*)(wta + (size_t)S * TC);              // S x TC
    const int nthreads = blockDim.x;

    const uint8_t* Lb = a.L + blockIdx.z * a.img_bs;
    const uint8_t* Rb = a.R + blockIdx.z * a.img_bs;
    int16_t* disp = (int16_t*)((uint8_t*)a.disp + blockIdx.z * a.disp_bs);
    int32_t* cost = a.cost ? (int32_t*)((uint8_t*)a.cost + blockIdx.z * a.cost_bs) : nullptr;
    const int ya = a.row0 + blockIdx.y * a.rows_per_chunk;
    const int yb = min(ya + a.rows_per_chunk, a.row1);
    if (ya >= yb) return;

    const int ccmin = c0 - r;
    const int rbase = a.rofs + clampi(ccmin, -a.rofs, a.w - a.ndisp - a.rofs);
    auto load_row = [&](int yy, int slot) {
        const uint8_t* lrow = Lb + (size_t)yy * a.pitch;
        const uint8_t* rrow = Rb + (size_t)yy * a.pitch;
        uint8_t* ld = lrows + slot * lwp;
        uint8_t* rd = rrows + slot * rwp;
        for (int i = threadIdx.x; i < lwp; i += nthreads) ld[i] = (uint8_t)(lrow[a.lofs + clampi(ccmin + i, -a.lofs, a.w - 1 - a.lofs)] + 1);
        for (int i = threadIdx.x; i < rwp; i += nthreads) rd[i] = (uint8_t)(rrow[clampi(rbase + i, 0, a.w - 1)] + 1);
    };

    const int ft1 = a.ftzero + 1;
    u32 V[16];
#pragma unroll
    for (int i = 0; i < 16; i++) V[i] = 0;
    int tsum = 0;

    // row sums of |L - R| over the window columns for this thread's 16 disparities (packed u16), and of |L - ftzero|
    auto row_sums = [&](int slot, u64 (&T)[4], int& t) {
        const uint8_t* lrow = lrows + slot * lwp;
        const uint8_t* rrow = rrows + slot * rwp;
#pragma unroll
        for (int g = 0; g < 4; g++) T[g] = 0;
        t = 0;
        for (int j = 0; j < blockw; j++) {
            const int cc = c0 + tc - r + j;
            const u32 lb = lrow[tc + j];
            const int rindex = a.rofs + clampi(cc, -a.rofs, a.w - a.ndisp - a.rofs) - rbase + s * 16;
            const u32* rw32 = (const u32*)rrow + (rindex >> 2);
            const int rsh = rindex & 3;
            u32 w0 = rw32[0], w1 = rw32[1], w2 = rw32[2], w3 = rw32[3], w4 = rw32[4];
            T[0] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)__builtin_amdgcn_alignbyte(w1, w0, rsh), lb, T[0]);
            T[1] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)__builtin_amdgcn_alignbyte(w2, w1, rsh), lb, T[1]);
            T[2] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)__builtin_amdgcn_alignbyte(w3, w2, rsh), lb, T[2]);
            T[3] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)__builtin_amdgcn_alignbyte(w4, w3, rsh), lb, T[3]);
            int df = (int)lb - ft1;
            t += df < 0 ? -df : df;
        }
    };
    auto apply = [&](const u64 (&T)[4], int t, bool add) {
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 v = (u32)(T[g] >> (16 * k)) & 0xFFFFu;
                V[4 * g + k] = add ? V[4 * g + k] + v : V[4 * g + k] - v;
            }
        tsum += add ? t : -t;
    };

    // warm-up: rows ya-r .. ya+r-1
    for (int yy = ya - r; yy < ya + r; yy++) {
        __syncthreads();
        load_row(yy, 0);
        __syncthreads();
        u64 T[4];
        int t;
        row_sums(0, T, t);
        apply(T, t, true);
    }

    const int c = c0 + tc;
    const bool writer = (s == 0) && (tc < ncols);
    uint8_t* csrow = cs + (size_t)tc * cspb;
    for (int y = ya; y < yb; y++) {
        __syncthreads();
        load_row(y + r, 0);
        if (y > ya) load_row(y - r - 1, 1);
        __syncthreads();
        {
            u64 T[4];
            int t;
            row_sums(0, T, t);
            apply(T, t, true);
            if (y > ya) {
                row_sums(1, T, t);
                apply(T, t, false);
            }
        }
        // winner-take-all over 64-bit keys (cost << 32 | d); slices merged through LDS
        u64 best = ~0ull;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            u64 k = ((u64)V[i] << 32) | (u32)(s * 16 + i);
            best = k < best ? k : best;
        }
        wta[s * TC + tc] = best;
#pragma unroll
        for (int i = 0; i < 16; i++) ((u32*)csrow)[s * 16 + i] = V[i];
        __syncthreads();
        u64 key = ~0ull;
        for (int i = 0; i < S; i++) { u64 k = wta[i * TC + tc]; key = k < key ? k : key; }
        const int minsad = (int)(key >> 32), mind = (int)(key & 0xFFFFFFFFu);
        const u32* csw = (const u32*)csrow;
        const int ip = (mind + 1 == a.ndisp) ? a.ndisp - 2 : mind + 1;
        const int in = (mind == 0) ? 1 : mind - 1;
        const int p = (int)csw[ip], n = (int)csw[in];
        bool ok = tsum >= a.tex_thr;
        if (a.uniq > 0) {
            const long long thresh = (long long)minsad + ((long long)minsad * a.uniq / 100);
            __syncthreads();
            if (s == 0) {
                u32* cw = (u32*)csrow;
                if (mind > 0) cw[mind - 1] = 0xFFFFFFFFu;
                cw[mind] = 0xFFFFFFFFu;
                if (mind + 1 < a.ndisp) cw[mind + 1] = 0xFFFFFFFFu;
            }
            __syncthreads();
            u32 m = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; i++) { u32 v = csw[s * 16 + i]; m = v < m ? v : m; }
            m2[s * TC + tc] = m;
            __syncthreads();
            u32 mm = 0xFFFFFFFFu;
            for (int i = 0; i < S; i++) { u32 v = m2[i * TC + tc]; mm = v < mm ? v : mm; }
            ok = ok && ((long long)mm > thresh);
        }
        int d_out = a.filtered;
        if (ok) {
            const int den = p + n - 2 * minsad + (p > n ? p - n : n - p);
            const int v = (a.ndisp - mind - 1 + a.mindisp) * 256 + (den != 0 ? (int)(((long long)(p - n) * 256) / den) : 0) + 15;
            d_out = v >> 4;
        }
        if (writer) {
            disp[(size_t)y * a.dpitch + a.lofs + c] = (int16_t)d_out;
            if (cost) cost[(size_t)y * a.cpitch + a.lofs + c] = minsad;
        }
    }
}

}  // namespace

int gip_bm_wide_launch(const BmArgs& a0, int nrows, int batch, hipStream_t st)
{
    BmArgs a = a0;
    const int S = a.ndisp / 16, r = a.wsz2;
    if (S < 1 || S > 256) return GIP_ERR_UNSUPPORTED;
    int TC = 256 / S;
    if (TC < 1) return GIP_ERR_UNSUPPORTED;
    if (TC > 64) TC = 64;
    a.TC = TC;
    const int ntiles = gip_div_up(a.width1, TC);
    // 2r warm-up rows per chunk cost as much as 2r output rows here, so chunks stay long
    int nch = gip_div_up(4096, ntiles * batch);
    int maxch = nrows / (4 * r + 4);
    if (maxch < 1) maxch = 1;
    if (nch > maxch) nch = maxch;
    if (nch < 1) nch = 1;
    a.rows_per_chunk = gip_div_up(nrows, nch);
    nch = gip_div_up(nrows, a.rows_per_chunk);
    const size_t lwp = (TC + 2 * r + 4 + 15) & ~15, rwp = (TC + 2 * r + a.ndisp + 16 + 15) & ~15;
    const size_t sh = 2 * (lwp + rwp) + (size_t)TC * (4 * a.ndisp + 16) + (size_t)S * TC * 12 + 16;
    if (sh > 160 * 1024) return GIP_ERR_UNSUPPORTED;
    if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_bm_wide, dim3(ntiles, nch, batch), dim3(TC * S), sh, st, a);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}
