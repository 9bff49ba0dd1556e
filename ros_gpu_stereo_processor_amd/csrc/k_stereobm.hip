// k_stereobm.hip -- block-matching disparity (SAD + winner-take-all + texture/uniqueness + sub-pixel) for gfx950.
//
// Replaces: GpuStereoProcessor::computeDisparity (reference src/GPUStereoProcessor.cpp:264-321).  The semantics are
// those of the reference's CPU path, cv::StereoBM::compute (findStereoCorrespondenceBM): SAD over the block on the
// PREFILTERED images, first minimum in reversed-d order (ties -> larger disparity), texture threshold, uniqueness
// ratio, 3-point sub-pixel with C truncating division, x16 fixed point -- all integer, reproduced bit-exactly.
//
// Design (wave64, no MFMA -- integer abs-diff work):
//   * one thread = one output column x one slice of DPT disparities; it marches DOWN the rows of its row chunk keeping
//     the vertical running window sums of all its disparities in VGPRs as packed u16 (4 per 64-bit register pair);
//   * prefiltered rows are staged once per workgroup in an LDS ring of 2r+1 rows (+1 bias per byte so that a zero byte
//     can serve as the "masked" reference byte of v_mqsad_pk_u16_u8);
//   * the horizontal window sum of a row is formed directly by v_mqsad_pk_u16_u8: 4 reference bytes of the left row
//     against a sliding 8-byte window of the right row = 4 consecutive disparities per instruction, accumulating into
//     the packed u16 sums;  row entering the window is added, row leaving it is recomputed and subtracted;
//   * winner-take-all: per-thread min over 32-bit keys (cost<<16 | d), slices merged through LDS; the cost slice of the
//     row is parked in LDS so the sub-pixel neighbours and the uniqueness scan (poison the winner's neighbourhood,
//     packed-min the rest) need no dynamic register indexing.
//   k_bm_fast handles the interior columns (no clamping anywhere in the window); k_bm_generic handles arbitrary
//   columns with OpenCV's clamped-column rule, one window column at a time (used for the 2r border columns that only
//   matter to the L/R check, and as the fallback).
// u16 sums require 2*cap*block^2 < 65536 (cap 31 -> block <= 31; cap 63 -> block <= 21).
#include "gip_internal.h"
#include "bm_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;


__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ u32 umin(u32 a, u32 b) { return a < b ? a : b; }

// packed-u16 min of two dwords (v_pk_min_u16)
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    us2 r = __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(u32, r);
}

// LDS layout shared by both kernels (dynamic shared memory, every carve offset a multiple of 16 bytes)
struct Smem {
    uint8_t* lring;   // NRING x lwp
    uint8_t* rring;   // NRING x rwp
    uint8_t* cs;      // TCOLS x cspb  (u16 cost slice of the current row)
    u32* wta;         // S x TCOLS     (per-slice best key)
    u32* m2;          // S x TCOLS     (per-slice min outside the winner's neighbourhood)
    int lwp, rwp, cspb;
};

__device__ __forceinline__ Smem carve(uint8_t* base, int nring, int lw, int rw, int tcols, int ndisp, int S)
{
    Smem s;
    s.lwp = (lw + 4 + 15) & ~15;
    s.rwp = (rw + 8 + 15) & ~15;
    s.cspb = 2 * ndisp + 16;
    s.lring = base;
    s.rring = s.lring + (size_t)nring * s.lwp;
    s.cs = s.rring + (size_t)nring * s.rwp;
    s.wta = (u32*)(s.cs + (size_t)tcols * s.cspb);
    s.m2 = s.wta + (size_t)S * tcols;
    return s;
}

inline size_t smem_bytes(int nring, int lw, int rw, int tcols, int ndisp, int S)
{
    size_t lwp = (lw + 4 + 15) & ~15, rwp = (rw + 8 + 15) & ~15;
    return (size_t)nring * (lwp + rwp) + (size_t)tcols * (2 * ndisp + 16) + 2 * (size_t)S * tcols * 4 + 16;
}

// ------------------------------------------------------------------------------------------------------------------
// Post-SAD phases for one output row.  V: this thread's NG packed cost registers (4 x u16 each) for internal
// disparities [s*4*NG, (s+1)*4*NG) of tile column tc.  Every thread of the column returns the same result.
// Returns the x16 disparity (or FILTERED) and the winning SAD.
template <int NG>
__device__ __forceinline__ int wta_row(const u64 (&V)[NG], const Smem& sm, int tc, int s, int S, int tcols, const BmArgs& a, int tsum,
                                       int* out_cost)
{
    // phase A: local best key, park the cost slice in LDS
    u32 best = 0xFFFFFFFFu;
    const int dbase = s * 4 * NG;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        u32 lo = (u32)V[g], hi = (u32)(V[g] >> 32);
        u32 d0 = dbase + 4 * g;
        best = umin(best, (lo << 16) | d0);
        best = umin(best, (lo & 0xFFFF0000u) | (d0 + 1));
        best = umin(best, (hi << 16) | (d0 + 2));
        best = umin(best, (hi & 0xFFFF0000u) | (d0 + 3));
    }
    sm.wta[s * tcols + tc] = best;
    uint8_t* csrow = sm.cs + (size_t)tc * sm.cspb;
    {
        u64* dst = (u64*)(csrow + dbase * 2);
#pragma unroll
        for (int g = 0; g < NG; g++) dst[g] = V[g];
    }
    __syncthreads();

    // phase B: merge slices
    u32 key = 0xFFFFFFFFu;
    for (int i = 0; i < S; i++) key = umin(key, sm.wta[i * tcols + tc]);
    const int minsad = (int)(key >> 16), mind = (int)(key & 0xFFFFu);
    const unsigned short* csw = (const unsigned short*)csrow;
    const int ip = (mind + 1 == a.ndisp) ? a.ndisp - 2 : mind + 1;  // sad[ndisp] = sad[ndisp-2]
    const int in = (mind == 0) ? 1 : mind - 1;                      // sad[-1]    = sad[1]
    const int p = csw[ip], n = csw[in];

    bool ok = tsum >= a.tex_thr;
    if (a.uniq > 0) {
        const int thresh = minsad + (minsad * a.uniq / 100);
        __syncthreads();  // everyone has read p, n
        if (s == 0) {
            unsigned short* cw = (unsigned short*)csrow;
            if (mind > 0) cw[mind - 1] = 0xFFFFu;
            cw[mind] = 0xFFFFu;
            if (mind + 1 < a.ndisp) cw[mind + 1] = 0xFFFFu;
        }
        __syncthreads();
        const u32* mine = (const u32*)(csrow + dbase * 2);
        u32 m = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 2 * NG; i++) m = pk_min_u16(m, mine[i]);
        sm.m2[s * tcols + tc] = umin(m & 0xFFFFu, m >> 16);
        __syncthreads();
        u32 m2 = 0xFFFFu;
        for (int i = 0; i < S; i++) m2 = umin(m2, sm.m2[i * tcols + tc]);
        ok = ok && ((int)m2 > thresh);
    }
    *out_cost = minsad;
    if (!ok) return a.filtered;
    const int den = p + n - 2 * minsad + (p > n ? p - n : n - p);
    const int v = (a.ndisp - mind - 1 + a.mindisp) * 256 + (den != 0 ? (p - n) * 256 / den : 0) + 15;
    return v >> 4;
}

// ------------------------------------------------------------------------------------------------------------------
// Fast interior kernel: block = 64 columns x S slices (thread = s*64 + tc), DPT disparities per thread,
// NQ = ceil(block/4) reference quads per row.
template <int DPT, int NQ>
__global__ void __launch_bounds__(DPT == 16 ? 1024 : 256) k_bm_fast(BmArgs a)
{
    constexpr int NG = DPT / 4, NRD = NG + NQ;  // R dwords after alignment
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    const int r = a.wsz2, blockw = 2 * r + 1, nring = blockw;
    const int S = a.ndisp / DPT;
    const int tc = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int c0 = a.cbeg + blockIdx.x * 64;
    const int lw = 64 + 2 * r, rw = 64 + 2 * r + a.ndisp - 1 + 8;
    const Smem sm = carve(smem_raw, nring, lw, rw, 64, a.ndisp, S);
    const int nthreads = blockDim.x;

    const uint8_t* Lb = a.L + blockIdx.z * a.img_bs;
    const uint8_t* Rb = a.R + blockIdx.z * a.img_bs;
    int16_t* disp = (int16_t*)((uint8_t*)a.disp + blockIdx.z * a.disp_bs);
    int32_t* cost = a.cost ? (int32_t*)((uint8_t*)a.cost + blockIdx.z * a.cost_bs) : nullptr;

    const int ya = a.row0 + blockIdx.y * a.rows_per_chunk;
    const int yb = min(ya + a.rows_per_chunk, a.row1);
    if (ya >= yb) return;

    const int lcol0 = a.lofs + c0 - r;  // absolute column of lring byte 0
    const int rcol0 = a.rofs + c0 - r;  // absolute column of rring byte 0
    auto load_row = [&](int yy) {
        const int slot = yy % nring;
        const uint8_t* lrow = Lb + (size_t)yy * a.pitch;
        const uint8_t* rrow = Rb + (size_t)yy * a.pitch;
        uint8_t* ld = sm.lring + (size_t)slot * sm.lwp;
        uint8_t* rd = sm.rring + (size_t)slot * sm.rwp;
        for (int i = threadIdx.x; i < sm.lwp; i += nthreads) ld[i] = (uint8_t)(lrow[clampi(lcol0 + i, 0, a.w - 1)] + 1);
        for (int i = threadIdx.x; i < sm.rwp; i += nthreads) rd[i] = (uint8_t)(rrow[clampi(rcol0 + i, 0, a.w - 1)] + 1);
    };

    const u32 lastmask = (blockw & 3) ? ((1u << (8 * (blockw & 3))) - 1u) : 0xFFFFFFFFu;
    const u32 ft4 = (u32)(a.ftzero + 1) * 0x01010101u;
    const int lsh = tc & 3, lidx = tc >> 2;
    const int j0 = tc + s * DPT, rsh = j0 & 3, ridx = j0 >> 2;

    u64 V[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) V[g] = 0;
    int tsum = 0;

    // sign = +1: add the row's horizontal SADs into V; sign = -1: subtract them
    auto accumulate = [&](int yy, bool add) {
        const int slot = yy % nring;
        const u32* lw32 = (const u32*)(sm.lring + (size_t)slot * sm.lwp) + lidx;
        const u32* rw32 = (const u32*)(sm.rring + (size_t)slot * sm.rwp) + ridx;
        u32 Lq[NQ], Rd[NRD];
        {
            u32 prev = lw32[0];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                u32 nxt = lw32[q + 1];
                Lq[q] = __builtin_amdgcn_alignbyte(nxt, prev, lsh);
                prev = nxt;
            }
            Lq[NQ - 1] &= lastmask;
        }
        {
            u32 prev = rw32[0];
#pragma unroll
            for (int i = 0; i < NRD; i++) {
                u32 nxt = rw32[i + 1];
                Rd[i] = __builtin_amdgcn_alignbyte(nxt, prev, rsh);
                prev = nxt;
            }
        }
        // texture: sum |L - ftzero| over the window row (masked bytes replaced by ftzero contribute 0)
        u32 t = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            u32 lq = (q == NQ - 1) ? (Lq[q] | (ft4 & ~lastmask)) : Lq[q];
            t = __builtin_amdgcn_sad_u8(lq, ft4, t);
        }
        if (add) {
            tsum += (int)t;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                u64 acc = V[g];
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    acc = __builtin_amdgcn_mqsad_pk_u16_u8(((u64)Rd[g + q + 1] << 32) | Rd[g + q], Lq[q], acc);
                V[g] = acc;
            }
        } else {
            tsum -= (int)t;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                u64 acc = 0;
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    acc = __builtin_amdgcn_mqsad_pk_u16_u8(((u64)Rd[g + q + 1] << 32) | Rd[g + q], Lq[q], acc);
                // lane-wise V >= acc, so plain 32-bit subtraction never borrows across the packed halves
                u32 lo = (u32)V[g] - (u32)acc, hi = (u32)(V[g] >> 32) - (u32)(acc >> 32);
                V[g] = ((u64)hi << 32) | lo;
            }
        }
    };

    // warm-up: rows ya-r .. ya+r-1
    for (int yy = ya - r; yy < ya + r; yy++) load_row(yy);
    __syncthreads();
    for (int yy = ya - r; yy < ya + r; yy++) accumulate(yy, true);

    const int c = c0 + tc;
    const bool writer = (s == 0) && (c < a.cend);
    for (int y = ya; y < yb; y++) {
        __syncthreads();  // slot of row y-r-1 is free (last read by the previous iteration's subtract)
        load_row(y + r);
        __syncthreads();
        accumulate(y + r, true);
        int mincost;
        int d = wta_row<NG>(V, sm, tc, s, S, 64, a, tsum, &mincost);
        if (writer) {
            disp[(size_t)y * a.dpitch + a.lofs + c] = (int16_t)d;
            if (cost) cost[(size_t)y * a.cpitch + a.lofs + c] = mincost;
        }
        accumulate(y - r, false);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Generic kernel: any output column, OpenCV's clamped-column rule
//     left  column of window column cc:  lofs + clamp(cc, -lofs, w-1-lofs)
//     right column of (cc, d)         :  rofs + clamp(cc, -rofs, w-ndisp-rofs) + d
// block = TC columns x S slices of 16 disparities (thread = s*TC + tc); one masked mqsad per (window column, 4 d).
__global__ void __launch_bounds__(1024) k_bm_generic(BmArgs a)
{
    constexpr int NG = 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    const int r = a.wsz2, blockw = 2 * r + 1, nring = blockw;
    const int S = a.ndisp / 16, TC = a.TC;
    const int tc = threadIdx.x % TC, s = threadIdx.x / TC;
    const int ntl = (a.nleft + TC - 1) / TC;
    // tile -> first column and number of valid columns
    int c0, ncols;
    if ((int)blockIdx.x < ntl) {
        c0 = blockIdx.x * TC;
        ncols = min(TC, a.nleft - c0);
    } else {
        int t = blockIdx.x - ntl;
        c0 = a.width1 - a.nright + t * TC;
        ncols = min(TC, a.width1 - c0);
    }
    const int lw = TC + 2 * r;
    const int ccmin = c0 - r, ccmax = c0 + TC - 1 + r;
    const int rbase = a.rofs + clampi(ccmin, -a.rofs, a.w - a.ndisp - a.rofs);
    const int rtop = a.rofs + clampi(ccmax, -a.rofs, a.w - a.ndisp - a.rofs);
    const int rw = rtop - rbase + a.ndisp + 8;
    const int rw_max = TC + 2 * r + a.ndisp + 8;  // carve size must not depend on the tile
    const Smem sm = carve(smem_raw, nring, lw, rw_max, TC, a.ndisp, S);
    const int nthreads = blockDim.x;

    const uint8_t* Lb = a.L + blockIdx.z * a.img_bs;
    const uint8_t* Rb = a.R + blockIdx.z * a.img_bs;
    int16_t* disp = (int16_t*)((uint8_t*)a.disp + blockIdx.z * a.disp_bs);
    int32_t* cost = a.cost ? (int32_t*)((uint8_t*)a.cost + blockIdx.z * a.cost_bs) : nullptr;

    const int ya = a.row0 + blockIdx.y * a.rows_per_chunk;
    const int yb = min(ya + a.rows_per_chunk, a.row1);
    if (ya >= yb) return;

    auto load_row = [&](int yy) {
        const int slot = yy % nring;
        const uint8_t* lrow = Lb + (size_t)yy * a.pitch;
        const uint8_t* rrow = Rb + (size_t)yy * a.pitch;
        uint8_t* ld = sm.lring + (size_t)slot * sm.lwp;
        uint8_t* rd = sm.rring + (size_t)slot * sm.rwp;
        for (int i = threadIdx.x; i < sm.lwp; i += nthreads)
            ld[i] = (uint8_t)(lrow[a.lofs + clampi(ccmin + i, -a.lofs, a.w - 1 - a.lofs)] + 1);
        for (int i = threadIdx.x; i < sm.rwp; i += nthreads) rd[i] = (uint8_t)(rrow[clampi(rbase + i, 0, a.w - 1)] + 1);
        (void)rw;
    };

    const int ft1 = a.ftzero + 1;
    u64 V[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) V[g] = 0;
    int tsum = 0;

    auto accumulate = [&](int yy, bool add) {
        const int slot = yy % nring;
        const uint8_t* lrow = sm.lring + (size_t)slot * sm.lwp;
        const uint8_t* rrow = sm.rring + (size_t)slot * sm.rwp;
        u64 T[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) T[g] = 0;
        int t = 0;
        for (int j = 0; j < blockw; j++) {
            const int cc = c0 + tc - r + j;
            const u32 lb = lrow[tc + j];
            const int rindex = a.rofs + clampi(cc, -a.rofs, a.w - a.ndisp - a.rofs) - rbase + s * 16;
            const u32* rw32 = (const u32*)rrow + (rindex >> 2);
            const int rsh = rindex & 3;
            u32 w0 = rw32[0], w1 = rw32[1], w2 = rw32[2], w3 = rw32[3], w4 = rw32[4];
            u32 d0 = __builtin_amdgcn_alignbyte(w1, w0, rsh), d1 = __builtin_amdgcn_alignbyte(w2, w1, rsh);
            u32 d2 = __builtin_amdgcn_alignbyte(w3, w2, rsh), d3 = __builtin_amdgcn_alignbyte(w4, w3, rsh);
            // reference = the single left byte in byte 0 (bytes 1..3 are zero -> masked), so result k = |lb - window byte k|
            T[0] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)d0, lb, T[0]);
            T[1] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)d1, lb, T[1]);
            T[2] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)d2, lb, T[2]);
            T[3] = __builtin_amdgcn_mqsad_pk_u16_u8((u64)d3, lb, T[3]);
            int df = (int)lb - ft1;
            t += df < 0 ? -df : df;
        }
        if (add) {
            tsum += t;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                u32 lo = (u32)V[g] + (u32)T[g], hi = (u32)(V[g] >> 32) + (u32)(T[g] >> 32);
                V[g] = ((u64)hi << 32) | lo;
            }
        } else {
            tsum -= t;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                u32 lo = (u32)V[g] - (u32)T[g], hi = (u32)(V[g] >> 32) - (u32)(T[g] >> 32);
                V[g] = ((u64)hi << 32) | lo;
            }
        }
    };

    for (int yy = ya - r; yy < ya + r; yy++) load_row(yy);
    __syncthreads();
    for (int yy = ya - r; yy < ya + r; yy++) accumulate(yy, true);

    const int c = c0 + tc;
    const bool writer = (s == 0) && (tc < ncols);
    for (int y = ya; y < yb; y++) {
        __syncthreads();
        load_row(y + r);
        __syncthreads();
        accumulate(y + r, true);
        int mincost;
        int d = wta_row<NG>(V, sm, tc, s, S, TC, a, tsum, &mincost);
        if (writer) {
            disp[(size_t)y * a.dpitch + a.lofs + c] = (int16_t)d;
            if (cost) cost[(size_t)y * a.cpitch + a.lofs + c] = mincost;
        }
        accumulate(y - r, false);
    }
}

// fill the whole map with FILTERED (columns/rows the matcher never writes)
__global__ void k_fill_i16(int16_t* disp, size_t dpitch, int w, int h, int value, size_t bs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    ((int16_t*)((uint8_t*)disp + blockIdx.z * bs))[(size_t)y * dpitch + x] = (int16_t)value;
}

template <int DPT, int NQ>
int launch_fast(const BmArgs& a, int ntiles, int nchunks, int batch, hipStream_t st, const BmLaunchCtx* ctx)
{
    const int S = a.ndisp / DPT, r = a.wsz2;
    size_t sh = smem_bytes(2 * r + 1, 64 + 2 * r, 64 + 2 * r + a.ndisp - 1 + 8, 64, a.ndisp, S);
    if (sh > 160 * 1024) return GIP_ERR_UNSUPPORTED;
    if (sh > 64 * 1024)
        GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_fast<DPT, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    if (ctx && ctx->ev0) GIP_HIP_CHECK(hipEventRecord(ctx->ev0, st));
    hipLaunchKernelGGL((k_bm_fast<DPT, NQ>), dim3(ntiles, nchunks, batch), dim3(64 * S), sh, st, a);
    GIP_LAUNCH_CHECK();
    if (ctx && ctx->ev1) GIP_HIP_CHECK(hipEventRecord(ctx->ev1, st));
    return GIP_OK;
}

template <int DPT>
int dispatch_nq(int nq, const BmArgs& a, int ntiles, int nchunks, int batch, hipStream_t st, const BmLaunchCtx* ctx)
{
    switch (nq) {
    case 2: return launch_fast<DPT, 2>(a, ntiles, nchunks, batch, st, ctx);
    case 3: return launch_fast<DPT, 3>(a, ntiles, nchunks, batch, st, ctx);
    case 4: return launch_fast<DPT, 4>(a, ntiles, nchunks, batch, st, ctx);
    case 5: return launch_fast<DPT, 5>(a, ntiles, nchunks, batch, st, ctx);
    case 6: return launch_fast<DPT, 6>(a, ntiles, nchunks, batch, st, ctx);
    case 7: return launch_fast<DPT, 7>(a, ntiles, nchunks, batch, st, ctx);
    case 8: return launch_fast<DPT, 8>(a, ntiles, nchunks, batch, st, ctx);
    default: return GIP_ERR_UNSUPPORTED;
    }
}

}  // namespace

int gip_validate_bm_params(const gip_bm_params* p, int w, int h)
{
    // StereoBMImpl::compute's CV_Error checks
    if (!p) return GIP_ERR_INVALID_ARG;
    if (p->prefilter_type != GIP_PREFILTER_NORMALIZED_RESPONSE && p->prefilter_type != GIP_PREFILTER_XSOBEL) return GIP_ERR_INVALID_ARG;
    if (p->prefilter_size < 5 || p->prefilter_size > 255 || p->prefilter_size % 2 == 0) return GIP_ERR_INVALID_ARG;
    if (p->prefilter_cap < 1 || p->prefilter_cap > 63) return GIP_ERR_INVALID_ARG;
    if (p->block_size < 5 || p->block_size > 255 || p->block_size % 2 == 0 || p->block_size >= (w < h ? w : h)) return GIP_ERR_INVALID_ARG;
    if (p->num_disparities <= 0 || p->num_disparities % 16 != 0) return GIP_ERR_INVALID_ARG;
    if (p->texture_threshold < 0 || p->uniqueness_ratio < 0) return GIP_ERR_INVALID_ARG;
    return GIP_OK;
}

extern "C" int gipk_bm_match(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp,
                             size_t disp_pitch, int32_t* cost, size_t cost_pitch, int batch, size_t img_bstride, size_t disp_bstride,
                             size_t cost_bstride, gip_stream stream)
{
    return gip_bm_match_impl(left, right, pitch, w, h, p, disp, disp_pitch, cost, cost_pitch, batch, img_bstride, disp_bstride, cost_bstride, stream, false, nullptr, nullptr);
}

// skip_fill: the caller runs gipk_bm_finalize afterwards, which blanks every pixel the matcher does not write
int gip_bm_match_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                      int32_t* cost, size_t cost_pitch, int batch, size_t img_bstride, size_t disp_bstride, size_t cost_bstride, gip_stream stream,
                      bool skip_fill, const BmLaunchCtx* ctx, int* cost_u16)
{
    // cost_u16: in = the caller accepts a uint16 min-SAD plane (cost_pitch / cost_bstride then describe 2-byte elements), out = whether
    // the matcher wrote one (only the column-sum matcher does; the other kernels keep int32)
    const bool want16 = cost_u16 && *cost_u16;
    if (cost_u16) *cost_u16 = 0;
    if (!left || !right || !disp || w <= 0 || h <= 0 || batch <= 0) return GIP_ERR_INVALID_ARG;
    if (disp_pitch % 2 || (cost && cost_pitch % 4)) return GIP_ERR_INVALID_ARG;   // (a uint16 plane is addressed with the same 4-byte aligned byte pitch)
    int rc = gip_validate_bm_params(p, w, h);
    if (rc != GIP_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const GipBmGeom g = gip_bm_geom(w, h, p);
    // u16 window sums need 2*cap*block^2 < 65535 (which also keeps the poison value 0xFFFF unreachable); wider windows take
    // the 32-bit fallback matcher
    const bool wide = (long)2 * p->prefilter_cap * p->block_size * p->block_size >= 65535;
    if (p->num_disparities > 4096) return GIP_ERR_UNSUPPORTED;

    if (!skip_fill) {
        hipLaunchKernelGGL(k_fill_i16, dim3(gip_div_up(w, 256), h, batch), dim3(256), 0, st, disp, disp_pitch / 2, w, h, g.filtered, disp_bstride);
        GIP_LAUNCH_CHECK();
    }
    if (g.degenerate) return GIP_OK;

    BmArgs a;
    a.L = left; a.R = right; a.pitch = pitch; a.w = w; a.h = h;
    a.ndisp = g.ndisp; a.mindisp = g.mindisp; a.wsz2 = g.wsz2; a.lofs = g.lofs; a.rofs = g.rofs; a.width1 = g.width1;
    a.ftzero = p->prefilter_cap; a.tex_thr = p->texture_threshold; a.uniq = p->uniqueness_ratio;
    a.row0 = g.roi_y; a.row1 = g.roi_y + g.roi_h;
    a.disp = disp; a.dpitch = disp_pitch / 2; a.cost = cost; a.cpitch = cost ? cost_pitch / 4 : 0; a.cost16 = 0;
    a.ntiles = a.nchunks = a.nimg = 0;
    a.img_bs = img_bstride; a.disp_bs = disp_bstride; a.cost_bs = cost_bstride; a.filtered = g.filtered;
    a.cbeg = a.cend = a.nleft = a.nright = 0; a.TC = 64;
    if (wide) {
        if (ctx && ctx->ev0) GIP_HIP_CHECK(hipEventRecord(ctx->ev0, st));
        rc = gip_bm_wide_launch(a, g.roi_h, batch, st);
        if (rc == GIP_OK && ctx && ctx->ev1) GIP_HIP_CHECK(hipEventRecord(ctx->ev1, st));
        return rc;
    }

    const int r = g.wsz2, nrows = g.roi_h;
    // interior columns: the whole window is un-clamped for c in [r, width1 - r)
    int cbeg = r, cend = g.width1 - r;
    bool fast_ok = cend > cbeg && (g.ndisp % 16 == 0);
    int dpt = (g.ndisp % 64 == 0) ? 64 : (g.ndisp % 32 == 0) ? 32 : 16;
    if (64 * (g.ndisp / dpt) > (dpt == 16 ? 1024 : 256)) fast_ok = false;  // block would exceed the kernel's launch bound
    const int nq = (2 * r + 1 + 3) / 4;
    if (nq > 8) fast_ok = false;

    // row chunking: enough workgroups to fill 256 CUs, but keep the 2r-row warm-up overhead small
    auto pick_chunks = [&](int tiles) {
        int want = gip_div_up(2048, tiles * batch > 0 ? tiles * batch : 1);
        int maxchunks = nrows / (8 * r + 8);  // warm-up <= ~25 %
        if (maxchunks < 1) maxchunks = 1;
        int nch = want < 1 ? 1 : (want > maxchunks ? maxchunks : want);
        return nch;
    };

    // border columns matter only to validateDisparity (they are blanked afterwards)
    const bool need_border = p->disp12_max_diff >= 0;
    auto launch_border = [&](hipStream_t bst, bool border_only) -> int {
        const int S = g.ndisp / 16;
        if (S > 256) return GIP_ERR_UNSUPPORTED;
        int TC = 256 / S; if (TC < 1) TC = 1; if (TC > 64) TC = 64;
        int widest = a.nleft > a.nright ? a.nleft : a.nright;
        while (TC > 1 && TC / 2 >= widest) TC /= 2;
        a.TC = TC;
        int ntiles = gip_div_up(a.nleft, TC) + gip_div_up(a.nright, TC);
        int nch = pick_chunks(ntiles);
        if (border_only) {
            // border-only launch: tiny blocks whose per-row global loads are latency bound -- use short row chunks so that
            // many small workgroups overlap (the extra 2r warm-up rows are cheap next to the idle time they remove)
            int short_rows = 2 * r + 8;
            nch = gip_div_up(nrows, short_rows);
        }
        a.rows_per_chunk = gip_div_up(nrows, nch);
        nch = gip_div_up(nrows, a.rows_per_chunk);
        size_t sh = smem_bytes(2 * r + 1, TC + 2 * r, TC + 2 * r + g.ndisp + 8, TC, g.ndisp, S);
        if (sh > 160 * 1024) return GIP_ERR_UNSUPPORTED;
        if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_bm_generic, dim3(ntiles, nch, batch), dim3(TC * S), sh, bst, a);
        GIP_LAUNCH_CHECK();
        return GIP_OK;
    };

    // preferred: column-sum matcher (k_stereobm_c4.hip).  It implements the clamped-column rule itself, so it covers every
    // output column [0, width1) and no border launch is needed
    a.cbeg = 0; a.cend = g.width1;
    if (gip_bm_c4_can(a)) {
        bool c4_done = false;
        if (want16 && cost) {   // same buffer, 2-byte elements: pitch in elements doubles for the same byte pitch
            a.cost16 = 1;
            a.cpitch = cost_pitch / 2;
            *cost_u16 = 1;
        }
        rc = gip_bm_c4_launch(a, batch, st, ctx, &c4_done);
        if (rc != GIP_OK) return rc;
        return c4_done ? GIP_OK : GIP_ERR_UNSUPPORTED;
    }
    a.cbeg = cbeg; a.cend = cend;
    if (fast_ok) {
        int ntiles = gip_div_up(cend - cbeg, 64);
        int nch = pick_chunks(ntiles);
        a.rows_per_chunk = gip_div_up(nrows, nch);
        nch = gip_div_up(nrows, a.rows_per_chunk);
        if (dpt == 64) rc = dispatch_nq<64>(nq, a, ntiles, nch, batch, st, ctx);
        else if (dpt == 32) rc = dispatch_nq<32>(nq, a, ntiles, nch, batch, st, ctx);
        else rc = dispatch_nq<16>(nq, a, ntiles, nch, batch, st, ctx);
        if (rc != GIP_OK) return rc;
        a.nleft = need_border ? (r < g.width1 ? r : g.width1) : 0;
        a.nright = need_border ? r : 0;
    } else {
        a.nleft = g.width1; a.nright = 0;
    }
    if (a.nleft + a.nright > 0) {
        rc = launch_border(st, fast_ok);
        if (rc != GIP_OK) return rc;
    }
    return GIP_OK;
}
