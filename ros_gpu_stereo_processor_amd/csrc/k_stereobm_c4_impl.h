// k_stereobm_c4_impl.h -- the main SAD block matcher for gfx950 ("column-sum" form), every output column.
// (kernel template + launcher; instantiated for ranges of the block radius by k_stereobm_c4_r*.hip, dispatched by k_stereobm_c4.hip)
//
// Same semantics as k_stereobm.hip (cv::StereoBM, reference src/GPUStereoProcessor.cpp:312-321); different work mapping,
// built around what the packed-byte SAD unit and the DPP cross-lane network of a CDNA4 wavefront do well:
//
//   * ONE WAVEFRONT = one tile of 64 image columns x a slice of the disparities, marching down its row chunk.
//     lane = s*16 + t :  t in [0,16) owns the 4 adjacent columns 4t..4t+3,  s in [0,4) owns a quarter of the wave's disparity
//     range (NG groups of 4 disparities).
//   * INTERIOR tiles (no clamped column) keep, per (lane, group), the PREFIX sums over the lane's four columns of the vertical
//     running column sums:  P1 = V0, P2 = V0+V1, P3 = V0+V1+V2, T = V0+V1+V2+V3  (V = sum over the 2r+1 window rows of |L - R|),
//     each as 4 packed u16 in a 64-bit register pair.  A row entering the window is added by four v_mqsad_pk_u16_u8 whose
//     reference dword is the lane's left-image dword with 1, 2, 3 and 4 LIVE bytes (a zero byte is masked; pixels are biased by +1
//     so a real pixel is never zero): column q sits at byte q and reads byte i+q of the lane's aligned 8-byte right-image window
//     for disparity i, so one instruction forms a whole prefix sum -- no byte shuffling, no prefix adds.  The row leaving the
//     window is recomputed the same way and subtracted with two 32-bit subtracts (no borrow can cross the packed lanes).
//   * The HORIZONTAL window sum of column q of lane t is then
//         H_q = sum of T over the whole lanes inside the window  -  P_ja(lane of the first column)  +  P_jb+1(lane of the last)
//     i.e. for block 9:  H0 = A + P1(t+1),  H1 = A - P1(t-1) + P2(t+1),  H2 = A - P2(t-1) + P3(t+1),  H3 = A - P3(t-1) + T(t+1),
//     A = T + T(t-1):  8 instructions per dword, every one a v_add_u32_dpp / v_sub_u32_dpp that fetches its neighbour-lane
//     operand through the DPP row shift (hsum_pf below; the column-wise form took 21 per dword; a subtraction is written
//     t = x(d) - acc; H = y(e) - t  because the DPP subtract always has the SHIFTED operand as minuend).  Outputs are valid for
//     the columns whose window lies inside the tile, 4t+q in [r, 63-r]; tiles advance by 64-2r columns.
//   * BORDER tiles (the clamped-column rule of OpenCV applies somewhere in the tile) keep the four column sums V0..V3 themselves
//     (one live reference byte per v_mqsad, byte position = clamped column) and form the window sums column-wise.
//   * Winner-take-all: per 8-disparity group (4 when NG is odd) a packed min and one 32-bit key (cost << 8 | group); the quarters are
//     merged on the VALU with v_permlane16_swap / v_permlane32_swap (quarters_reduce).  The row's cost slice is parked in LDS (one row
//     per live lane and column, the halo lanes into bank-free dump slots); lane (s,t) then finalises pixel 4t+s: exact winner inside the
//     group, sub-pixel neighbours, texture test and the uniqueness test (count of groups <= threshold vs the groups that touch the
//     winner, then an exact check of those costs).
//   * Prefiltered rows are staged by the wave itself with aligned dword loads issued one row ahead (register prefetch), so global latency
//     overlaps the SAD work -- into an LDS ring of 2r+1 rows, or (refetch form, chosen at launch for one-wave large-block shapes that have the
//     device to themselves) into two slots, the leaving row being loaded a second time.
//   * The launcher (launch_c4_rf) picks the row chunking from the kernel's LDS / VGPR footprint and, when the caller declares other batches
//     in flight, leaves room for their kernels.
#pragma once
#include "gip_internal.h"
#include "bm_common.h"

#include <hip/hip_ext.h>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

// C4_ABLATE: bit mask for timing-only diagnostic builds (tools/ab_c4.sh): 1 no leaving-row subtract, 2 no window sums (costs = the
// prefix sums themselves), 4 no cost-slice parking, 8 no quarter merges (ds_bpermute), 16 no finalisation.  Any bit set gives
// WRONG disparities; the product build has none.
#ifndef C4_ABLATE
#define C4_ABLATE 0
#endif
// measured design switches (tools/ab_c4.sh, tools/bm_ab.py).  The variants that lost their A/B and are gone from the source:
// subtracting the leaving row at the top of the next iteration from registers read before finalisation (+4 % time), 8-byte parks (bank
// conflicts: 68 M vs 14 M conflict cycles), 4-disparity winner-take-all groups for even NG (+14 M instructions), ds_bpermute quarter merges
// (two LDS round trips per merge).
#ifndef C4_ROTFIN   /* 1: multi-wave workgroups: ONE wave (rotating per row) finalises all 64 pixels of the row with every lane active */
#define C4_ROTFIN 1
#endif
#ifndef C4_XCD_MAP  /* 1: contiguous ranges of tiles per XCD (L2 locality) instead of the natural blockIdx order */
#define C4_XCD_MAP 1
#endif
#ifndef C4_MINW     /* second argument of __launch_bounds__: waves per SIMD the register allocator must leave room for */
#define C4_MINW 1
#endif
#ifndef C4_NOSYNC   /* 1: one-wave workgroups order their LDS traffic by program order only (no s_waitcnt drain) */
#define C4_NOSYNC 1
#endif

constexpr int GIP_MAX_DEVICES = 64;

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ u32 umin(u32 a, u32 b) { return a < b ? a : b; }

__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(u32, __builtin_elementwise_min(x, y));
}

// value of lane (t + D) within the 16-lane DPP row; lanes without a source get 0
template <int D>
__device__ __forceinline__ u32 lane_at(u32 v)
{
    if constexpr (D == 0) return v;
    else if constexpr (D < 0) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + (-D), 0xF, 0xF, true);  // row_shr:-D
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + D, 0xF, 0xF, true);                           // row_shl:D
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// packed pair (two dwords = 4 x u16) helpers
struct P2 { u32 lo, hi; };
__device__ __forceinline__ P2 operator+(P2 a, P2 b) { return {a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ P2 operator-(P2 a, P2 b) { return {a.lo - b.lo, a.hi - b.hi}; }

// Reduction of a value over the four lanes t, t+16, t+32, t+48 (the disparity quarters of one column group), result in all four, for a
// commutative, associative op.  gfx950 has two VALU lane-swap instructions for exactly this (measured: tools/probe/permlane_swap_probe.hip):
// v_permlane16_swap a, b exchanges the ODD 16-lane rows of a with the EVEN rows of b, v_permlane32_swap the upper half of a with the lower
// half of b.  With a = b = x:  a' = {x.r0, x.r0, x.r2, x.r2}, b' = {x.r1, x.r1, x.r3, x.r3}, so op(a', b') is the xor-16 butterfly step in every
// lane, and the same with halves for xor-32 -- no LDS round trip (ds_bpermute: the wave parks ~100+ cycles per dependent step).
// NSL = 4: over all four rows; NSL = 2 (two sub-images per wave, each in one 32-lane half): over the two rows of each half only.
template <int NSL = 4, typename Op>
__device__ __forceinline__ u32 quarters_reduce(u32 x, Op op)
{
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    x = op((u32)r[0], (u32)r[1]);
    if constexpr (NSL == 4) {
        auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        x = op((u32)h[0], (u32)h[1]);
    }
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Column-wise window sum (border tiles): output column q of this lane from the in-lane prefix sums P[0..4] (P[0] = 0):
//   window = columns [4t+q-R, 4t+q+R];  lane t+d covers columns 4(t+d)..4(t+d)+3  ->  sub-columns [lo,hi]
template <int R, int Q, class T, class SHIFT>
__device__ __forceinline__ T hsum(const T (&P)[5], SHIFT&& shift)
{
    constexpr int K = (R + 3) / 4;
    constexpr int lo0 = (Q - R) > 0 ? (Q - R) : 0;
    constexpr int hi0 = (Q + R) < 3 ? (Q + R) : 3;
    T acc{};
    if constexpr (lo0 <= hi0) {
        if constexpr (lo0 == 0) acc = P[hi0 + 1];
        else acc = P[hi0 + 1] - P[lo0];
    }
    static_for<0, 2 * K + 1>([&](auto iv) {
        constexpr int d = decltype(iv)::value - K;
        constexpr int lo = (Q - R - 4 * d) > 0 ? (Q - R - 4 * d) : 0;
        constexpr int hi = (Q + R - 4 * d) < 3 ? (Q + R - 4 * d) : 3;
        if constexpr (d != 0 && lo <= hi) {
            if constexpr (lo == 0) acc = acc + shift(std::integral_constant<int, d>{}, P[hi + 1]);
            else acc = acc + shift(std::integral_constant<int, d>{}, P[hi + 1] - P[lo]);
        }
    });
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Prefix-form window sums (interior tiles).  In: the lane's T (sum of its four columns) and prefix sums P1..P3, one dword each
// (two packed u16 or one u32).  Out: the window sums H0..H3 of its four columns.  With R = 4m + rho:
//     H_q = S + [q < rho ? T(-m-1) - P_{q-rho+4}(-m-1) : -P_{q-rho}(-m)] + [q + rho > 3 ? T(m) + P_{q+rho-3}(m+1) : P_{q+rho+1}(m)]
//     S   = T(-m) + ... + T(m-1)                                  (X(d) = X of lane t+d, P_0 = 0, P_4 = T)
// Every line of the asm below is ONE VALU instruction: the neighbour-lane operand comes through the DPP row shift of
// v_add_u32_dpp / v_sub_u32_dpp (bound_ctrl: a lane without a source reads 0; those lanes hold no valid output).
// Measured on gfx950 (tools/probe/dpp_sub_probe.hip): v_sub_u32_dpp d, a, b computes a[shifted] - b and v_subrev_u32_dpp d, a, b
// computes b[shifted] - a -- the "rev" form moves the DPP shift to the other operand, it does not make the UNSHIFTED operand the
// minuend -- so "acc - x(d)" does not exist as one instruction (the compiler's own DPP fusion of such a subtract gives wrong
// results for the same reason).  Every subtraction is therefore written  t = x(d) - acc;  H = y(e) - t  (= acc - x(d) + y(e)).
// All sums are exact modulo 2^32 on the packed dword, so an intermediate that exceeds 16 bits in a half (or is "negative") is
// harmless: the final value of each half is < 2^16 by the u16 condition 2*cap*block^2 < 65535.
// The compiler does not know these are DPP instructions, so it inserts no wait states for the "VALU writes a VGPR -> a DPP
// instruction reads it" hazard (2 wait states).  Only T and P1..P3 are ever read through DPP, never a value produced inside a
// statement, and every statement starts with s_nop 1: whatever wrote them -- including a copy the register allocator may have put
// right in front of the statement -- is then at least 2 wait states old.
#define GIP_DPPM " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#ifndef GIP_DPP_HEAD
#define GIP_DPP_HEAD "s_nop 1\n\t"
#endif
#define GIP_ADD_SHR(d, x, acc, n) "v_add_u32_dpp " d ", " x ", " acc " row_shr:" n GIP_DPPM   /* d = x(t-n) + acc */
#define GIP_ADD_SHL(d, x, acc, n) "v_add_u32_dpp " d ", " x ", " acc " row_shl:" n GIP_DPPM   /* d = x(t+n) + acc */
#define GIP_SUB_SHR(d, x, b, n) "v_sub_u32_dpp " d ", " x ", " b " row_shr:" n GIP_DPPM       /* d = x(t-n) - b   */
#define GIP_SUB_SHL(d, x, b, n) "v_sub_u32_dpp " d ", " x ", " b " row_shl:" n GIP_DPPM       /* d = x(t+n) - b   */

template <int R>
__device__ __forceinline__ void hsum_pf(u32 T, u32 P1, u32 P2, u32 P3, u32& H0, u32& H1, u32& H2, u32& H3)
{
    constexpr int m = R / 4, rho = R % 4;
    static_assert(R >= 2 && R <= 15, "block radius");
    // operands: %0-%3 H0-H3, %4-%6 temporaries, then the inputs
    u32 x, y, z;
    if constexpr (m == 0 && rho == 2) {
        // R = 2: H0 = T(-1) - P2(-1) + P3,  H1 = T(-1) - P3(-1) + T,  H2 = T + P1(1),  H3 = T - P1 + P2(1)       [%7 T, %8-%10 P1-P3]
        asm(GIP_DPP_HEAD
            GIP_SUB_SHR("%4", "%9", "%10", "1")       // x  = P2(-1) - P3
            GIP_SUB_SHR("%5", "%10", "%7", "1")       // y  = P3(-1) - T
            "v_sub_u32 %6, %7, %8\n\t"                // z  = T - P1
            GIP_ADD_SHL("%2", "%8", "%7", "1")        // H2 = P1(1) + T
            GIP_SUB_SHR("%0", "%7", "%4", "1")        // H0 = T(-1) - x
            GIP_SUB_SHR("%1", "%7", "%5", "1")        // H1 = T(-1) - y
            GIP_ADD_SHL("%3", "%9", "%6", "1")        // H3 = P2(1) + z
            : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
            : "v"(T), "v"(P1), "v"(P2), "v"(P3));
    } else if constexpr (m == 0) {
        // R = 3: H0 = T(-1) - P1(-1) + T,  H1 = X - P2(-1) + P1(1),  H2 = X - P3(-1) + P2(1),  H3 = T + P3(1),  X = T + T(-1)
        asm(GIP_DPP_HEAD
            GIP_ADD_SHR("%4", "%7", "%7", "1")        // X  = T(-1) + T
            GIP_SUB_SHR("%5", "%8", "%7", "1")        // y  = P1(-1) - T
            GIP_ADD_SHL("%3", "%10", "%7", "1")       // H3 = P3(1) + T
            GIP_SUB_SHR("%1", "%9", "%4", "1")        // H1 = P2(-1) - X
            GIP_SUB_SHR("%2", "%10", "%4", "1")       // H2 = P3(-1) - X
            GIP_SUB_SHR("%0", "%7", "%5", "1")        // H0 = T(-1) - y
            GIP_SUB_SHL("%1", "%8", "%1", "1")        // H1 = P1(1) - H1
            GIP_SUB_SHL("%2", "%9", "%2", "1")        // H2 = P2(1) - H2
            : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
            : "v"(T), "v"(P1), "v"(P2), "v"(P3));
    } else {
        // S: the whole lanes -m .. m-1
        u32 S;
        if constexpr (m == 1) asm(GIP_DPP_HEAD GIP_ADD_SHR("%0", "%1", "%1", "1") : "=&v"(S) : "v"(T));
        else if constexpr (m == 2)
            asm(GIP_DPP_HEAD GIP_ADD_SHR("%0", "%1", "%1", "1") GIP_ADD_SHR("%0", "%1", "%0", "2") GIP_ADD_SHL("%0", "%1", "%0", "1") : "=&v"(S) : "v"(T));
        else
            asm(GIP_DPP_HEAD GIP_ADD_SHR("%0", "%1", "%1", "1") GIP_ADD_SHR("%0", "%1", "%0", "2") GIP_ADD_SHR("%0", "%1", "%0", "3")
                GIP_ADD_SHL("%0", "%1", "%0", "1") GIP_ADD_SHL("%0", "%1", "%0", "2") : "=&v"(S) : "v"(T));
        // inputs: %7 S, %8 T, %9-%11 P1-P3, %12 = m, %13 = m + 1
        if constexpr (rho == 0)
            // H0 = S + P1(m),  H1 = S - P1(-m) + P2(m),  H2 = S - P2(-m) + P3(m),  H3 = S - P3(-m) + T(m)
            asm(GIP_DPP_HEAD
                GIP_SUB_SHR("%1", "%9", "%7", "%12")      // H1 = P1(-m) - S
                GIP_SUB_SHR("%2", "%10", "%7", "%12")     // H2 = P2(-m) - S
                GIP_SUB_SHR("%3", "%11", "%7", "%12")     // H3 = P3(-m) - S
                GIP_ADD_SHL("%0", "%9", "%7", "%12")      // H0 = P1(m) + S
                GIP_SUB_SHL("%1", "%10", "%1", "%12")     // H1 = P2(m) - H1
                GIP_SUB_SHL("%2", "%11", "%2", "%12")     // H2 = P3(m) - H2
                GIP_SUB_SHL("%3", "%8", "%3", "%12")      // H3 = T(m) - H3
                : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
                : "v"(S), "v"(T), "v"(P1), "v"(P2), "v"(P3), "n"(m), "n"(m + 1));
        else if constexpr (rho == 1)
            // H0 = S + T(-m-1) - P3(-m-1) + P2(m),  H1 = S + P3(m),  H2 = S - P1(-m) + T(m),  H3 = S + T(m) - P2(-m) + P1(m+1)
            asm(GIP_DPP_HEAD
                GIP_ADD_SHR("%4", "%8", "%7", "%13")      // x  = T(-m-1) + S
                GIP_ADD_SHL("%5", "%8", "%7", "%12")      // y  = T(m) + S
                GIP_SUB_SHR("%2", "%9", "%7", "%12")      // H2 = P1(-m) - S
                GIP_ADD_SHL("%1", "%11", "%7", "%12")     // H1 = P3(m) + S
                GIP_SUB_SHR("%0", "%11", "%4", "%13")     // H0 = P3(-m-1) - x
                GIP_SUB_SHR("%3", "%10", "%5", "%12")     // H3 = P2(-m) - y
                GIP_SUB_SHL("%2", "%8", "%2", "%12")      // H2 = T(m) - H2
                GIP_SUB_SHL("%0", "%10", "%0", "%12")     // H0 = P2(m) - H0
                GIP_SUB_SHL("%3", "%9", "%3", "%13")      // H3 = P1(m+1) - H3
                : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
                : "v"(S), "v"(T), "v"(P1), "v"(P2), "v"(P3), "n"(m), "n"(m + 1));
        else if constexpr (rho == 2)
            // H0 = W - P2(-m-1) + P3(m),  H1 = W - P3(-m-1) + T(m),  H2 = U + P1(m+1),  H3 = U - P1(-m) + P2(m+1),  W = S + T(-m-1), U = S + T(m)
            asm(GIP_DPP_HEAD
                GIP_ADD_SHR("%4", "%8", "%7", "%13")      // W  = T(-m-1) + S
                GIP_ADD_SHL("%5", "%8", "%7", "%12")      // U  = T(m) + S
                GIP_SUB_SHR("%0", "%10", "%4", "%13")     // H0 = P2(-m-1) - W
                GIP_SUB_SHR("%1", "%11", "%4", "%13")     // H1 = P3(-m-1) - W
                GIP_SUB_SHR("%3", "%9", "%5", "%12")      // H3 = P1(-m) - U
                GIP_ADD_SHL("%2", "%9", "%5", "%13")      // H2 = P1(m+1) + U
                GIP_SUB_SHL("%0", "%11", "%0", "%12")     // H0 = P3(m) - H0
                GIP_SUB_SHL("%1", "%8", "%1", "%12")      // H1 = T(m) - H1
                GIP_SUB_SHL("%3", "%10", "%3", "%13")     // H3 = P2(m+1) - H3
                : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
                : "v"(S), "v"(T), "v"(P1), "v"(P2), "v"(P3), "n"(m), "n"(m + 1));
        else
            // H0 = W - P1(-m-1) + T(m),  H1 = X - P2(-m-1) + P1(m+1),  H2 = X - P3(-m-1) + P2(m+1),  H3 = U + P3(m+1),
            // W = S + T(-m-1),  U = S + T(m),  X = W + T(m)
            asm(GIP_DPP_HEAD
                GIP_ADD_SHR("%4", "%8", "%7", "%13")      // W  = T(-m-1) + S
                GIP_ADD_SHL("%5", "%8", "%7", "%12")      // U  = T(m) + S
                GIP_ADD_SHL("%6", "%8", "%4", "%12")      // X  = T(m) + W
                GIP_SUB_SHR("%0", "%9", "%4", "%13")      // H0 = P1(-m-1) - W
                GIP_ADD_SHL("%3", "%11", "%5", "%13")     // H3 = P3(m+1) + U
                GIP_SUB_SHR("%1", "%10", "%6", "%13")     // H1 = P2(-m-1) - X
                GIP_SUB_SHR("%2", "%11", "%6", "%13")     // H2 = P3(-m-1) - X
                GIP_SUB_SHL("%0", "%8", "%0", "%12")      // H0 = T(m) - H0
                GIP_SUB_SHL("%1", "%9", "%1", "%13")      // H1 = P1(m+1) - H1
                GIP_SUB_SHL("%2", "%10", "%2", "%13")     // H2 = P2(m+1) - H2
                : "=&v"(H0), "=&v"(H1), "=&v"(H2), "=&v"(H3), "=&v"(x), "=&v"(y), "=&v"(z)
                : "v"(S), "v"(T), "v"(P1), "v"(P2), "v"(P3), "n"(m), "n"(m + 1));
    }
}

// LDS budget of one workgroup and the resulting workgroups per CU (by LDS and wave slots only; registers are looked at at launch).
// Rows of the cost slice: one per (column q, live lane t).  The R / 4 lanes at either end of the 16-lane row hold no output column (their
// columns' windows leave the tile); they still execute the parks, so each gets ONE dump row for all four of its columns.
constexpr int c4_slice_rows(int R) { return 4 * (16 - 2 * (R / 4)); }
// ... and the dump area of the halo lanes: one row's length plus room to shift each halo lane's stores onto the LDS banks its eight-lane store
// group leaves free (a 16-byte store is issued eight lanes at a time over 32 banks; the seven or fewer live lanes of a group use 4 banks each)
constexpr int c4_dump_bytes(int cspb) { return cspb + 128; }
// dword offset, inside the dump area, at which halo lane t parks column q (bank-conflict free against the live lanes of its group)
constexpr int c4_dump_off(int q, int t, int R, int cspb)
{
    const int TLO = R / 4, NLIVE = 16 - 2 * TLO, W = (cspb / 4) % 32, bankD = (4 * NLIVE * (cspb / 4)) % 32;
    const int grp = t >> 3;
    bool used[8] = {false, false, false, false, false, false, false, false};
    for (int tt = 8 * grp; tt < 8 * grp + 8; tt++)
        if (tt >= TLO && tt <= 15 - TLO) used[(((q * NLIVE + tt - TLO) * W) % 32) / 4] = true;
    int k = t < TLO ? t : t - (16 - TLO);     // index among the halo lanes of this half
    for (int slot = 0; slot < 8; slot++)
        if (!used[slot]) {
            if (k == 0) return ((slot * 4 - bankD) % 32 + 32) % 32;
            k--;
        }
    return 0;
}
constexpr size_t c4_lds_bytes(int NG, int R, int NW, int nring, int NSUB = 1)
{
    const int NS = (4 / NSUB) * NW, RWD = 17 + NS * NG, RWDP = (RWD + 3) & ~3, SLOTD = 16 + RWDP;
    // ring + cost slice (one of each per sub-image) + (multi-wave workgroups only) the exchange area for winner keys and uniqueness counts
    const int cspb = 2 * 4 * NG * NS + 16;
    return (size_t)NSUB * ((size_t)nring * SLOTD * 4 + (size_t)c4_slice_rows(R) * cspb + (R / 4 > 0 ? c4_dump_bytes(cspb) : 0)) + (NW > 1 ? (size_t)2 * NW * 64 * 4 : 0);
}
constexpr int c4_per_cu(size_t lds, int NW)
{
    const size_t granule = 1280;          // LDS is handed out in 1/128ths of the 160 KiB
    const int p = (int)((160 * 1024) / ((lds + granule - 1) / granule * granule));
    return p > 32 / NW ? 32 / NW : p;
}
// REFETCH: the row ring (2r+1 rows, there so that the row LEAVING the window can be recomputed without touching memory again) grows
// with the block size and, at large blocks, costs a whole workgroup per CU.  When it does, the kernel keeps only two staging slots
// and fetches the leaving row a second time through L2 (the same workgroup loaded it 2r+1 iterations earlier, its neighbours too):
// one more dword load per thread and row buys up to a third more resident waves (nd 128 / block 21: 6 -> 8 workgroups per CU, -13.5 %;
// nd 64 / block 15: -12 %).  One-wave workgroups only: with several waves the extra staging needs two barriers per warm-up row and the
// measured result is a loss (nd 256 / block 21, 3 -> 4 workgroups per CU: +5 %; nd 192 / block 15: +3 %).
// C4_REFETCH: 0 never, 1 when it raises the workgroups per CU of a one-wave shape, 2 always.
#ifndef C4_REFETCH
#define C4_REFETCH 1
#endif
constexpr bool c4_refetch(int NG, int R, int NW, int NSUB = 1)
{
    // what the registers allow anyway (measured VGPR counts: NG <= 4 about 100-128 = 4 waves per SIMD, NG 5..7 up to 168 = 3, NG >= 8 more = 2)
    const int cap = NG <= 4 ? 16 : NG <= 7 ? 12 : 8;
    const int with = c4_per_cu(c4_lds_bytes(NG, R, NW, 2, NSUB), NW), without = c4_per_cu(c4_lds_bytes(NG, R, NW, 2 * R + 1, NSUB), NW);
    // NG > 12 (nd 208 / 240) holds all 256 VGPRs: register-bound whatever the LDS says (measured: +6 % with the refetch)
    return C4_REFETCH == 2 || (C4_REFETCH == 1 && NW == 1 && NG <= 12 && (with < cap ? with : cap) > (without < cap ? without : cap));
}

// NW = wavefronts per workgroup.  NW = 1 is the layout described above.  With NW > 1 the waves of a workgroup work on the SAME
// 64 columns and split the disparity range further (slice s = 4 * wave + lane / 16 of NS = 4 * NW slices): they share the row
// ring and the cost slice, so the LDS per wave -- what limits the number of resident waves -- drops by NW, and so do the
// vertical-sum registers per lane.  The price: real barriers, the winner keys / uniqueness counts of the waves are merged
// through LDS, the texture sums are computed by every wave, and a wave finalises only 64 / NW pixels per row.
//
// PF: prefix form (interior tile) or column form (border tile); uniform per workgroup.
//
// NSUB = 2 (small disparity counts, one-wave workgroups): the wave works on the same tile and row chunk of TWO images of the batch, one per
// 32-lane half; each half splits the disparity range over its two 16-lane rows only (NSL = 2 slices of NG groups, NG = ndisp / 8).  At
// ndisp <= 64 the work a lane does per row is dominated by what does not shrink with the disparity count -- staging, texture sums, the
// quarter merges, the uniqueness thresholds, one finalisation per lane: 11.4 lane-instructions per (pixel, disparity) at (48, 19) against
// 5.1 at (128, 9).  With two slices per image every one of those is paid for twice the disparities, the merges lose a step, and NG is even
// for every multiple of 16, so the 8-disparity winner-take-all groups apply to ndisp 16 and 48 too.  A lane finalises two pixels.  `img` is
// then the index of the image PAIR; the second half of the last pair of an odd batch works on a copy of the first and stores nothing.
template <int NG, int R, int NW, bool PF, bool RF, int NSUB = 1>
__device__ __forceinline__ void bm_c4_tile(const BmArgs& a, uint8_t* smem_raw, int c0, int ya, int yb, int img)
{
    static_assert(NSUB == 1 || (NSUB == 2 && NW == 1), "two sub-images per wave: one-wave workgroups only");
    constexpr int NRING = RF ? 2 : 2 * R + 1;   // RF: slot 0 = the entering row, slot 1 = the leaving row
    constexpr int NSL = 4 / NSUB;               // disparity slices per wave (and sub-image)
    constexpr int NS = NSL * NW;                // disparity slices
    constexpr int NT = 64 * NW;                 // threads
    constexpr int NTS = NT / NSUB;              // threads that stage one sub-image's rows
    constexpr int NDISP = 4 * NG * NS;
    constexpr int RWD = 17 + NS * NG;           // right-row dwords needed per ring slot
    constexpr int RWDP = (RWD + 3) & ~3;        // padded to 16 bytes
    constexpr int SLOTD = 16 + RWDP;            // dwords per ring slot
    constexpr int CSPB = 2 * NDISP + 16;        // bytes per column of the cost slice
    constexpr int NLD = (RWD + NTS - 1) / NTS;  // right-row dwords each thread stages per row
    constexpr int QPW = 4 / NW;                 // pixel columns q finalised per wave
    // Winner-take-all granularity: the key of a column is (min cost << 8 | group), the group's costs are re-read from the LDS cost slice
    // at finalisation.  With 8-disparity groups ("octets" = two v_mqsad groups, which are parked together anyway) the per-group work --
    // horizontal min, key, running min, uniqueness compare-and-count -- halves; finalisation reads three octets instead of three quads.
    // NG must be even so that no octet is split between two lanes (the count would see it twice).
    constexpr bool OCT = NG % 2 == 0;
    constexpr int NGK = OCT ? NG / 2 : NG;      // key groups per lane
    constexpr int SLICE_BYTES = c4_slice_rows(R) * CSPB + (R / 4 > 0 ? c4_dump_bytes(CSPB) : 0);

    const int tid = threadIdx.x, wave = tid >> 6;
    const int lane = tid & 63, t = lane & 15;
    const int sub = NSUB == 2 ? (lane >> 5) : 0;                 // sub-image of this lane
    const int sl = NSUB == 2 ? ((lane >> 4) & 1) : (lane >> 4);  // disparity slice inside the wave (and sub-image)
    const int s = NSL * wave + sl;
    const int ti = NSUB == 2 ? (tid & 31) : tid;                 // index among the threads that stage this sub-image

    u32* ring = (u32*)smem_raw + sub * SLOTD;                     // slot k of this sub-image: ring + k * NSUB * SLOTD
    uint8_t* cs = smem_raw + (size_t)NRING * NSUB * SLOTD * 4 + (size_t)sub * SLICE_BYTES;
    u32* xbest = (u32*)(cs + (size_t)c4_slice_rows(R) * CSPB + (R / 4 > 0 ? c4_dump_bytes(CSPB) : 0));   // [NW][16][4] winner keys of each wave (NW > 1 only)
    u32* xcnt = xbest + NW * 64;                    // [NW][16][4] uniqueness counts of each wave

    const bool sub_live = img * NSUB + sub < a.batch;             // (odd batch: the last pair has one image)
    const int im = sub_live ? img * NSUB + sub : a.batch - 1;
    const uint8_t* Lb = a.L + (size_t)im * a.img_bs;
    const uint8_t* Rb = a.R + (size_t)im * a.img_bs;
    int16_t* disp = (int16_t*)((uint8_t*)a.disp + (size_t)im * a.disp_bs);
    int32_t* cost = a.cost ? (int32_t*)((uint8_t*)a.cost + (size_t)im * a.cost_bs) : nullptr;

    // ---- row staging: aligned dword loads + funnel shift, clamped to the row's dword range (pitch % 4 == 0)
    const int lcol0 = a.lofs + c0;   // absolute column of L ring byte 0
    const int rcol0 = a.rofs + c0;   // absolute column of R ring byte 0
    const int ndw = (int)(a.pitch >> 2);
    const int lsh = lcol0 & 3, rsh = rcol0 & 3;
    const int ldw = (lcol0 >> 2) + ti;   // (arithmetic shift: floor for negative columns)
    const int rdw = (rcol0 >> 2) + ti;
    auto fetch = [&](const uint8_t* base, int yy, int dw, int sh) -> u32 {
        const u32* row = (const u32*)(base + (size_t)yy * a.pitch);
        u32 lo = row[clampi(dw, 0, ndw - 1)], hi = row[clampi(dw + 1, 0, ndw - 1)];
        u32 v = __builtin_amdgcn_alignbyte(hi, lo, sh);
        return (v & 0x7F7F7F7Fu) + 0x01010101u;   // +1 bias (valid pixels are <= 2*cap <= 126)
    };
    // left image, byte-granular edge replication (clamped-column rule: left column = clamp(lofs + cc, 0, w - 1)); only the
    // lanes whose quad straddles an image edge take the slow path
    auto fetch_l = [&](int yy) -> u32 {
        const int col = lcol0 + 4 * ti;
        if (col >= 0 && col + 3 <= a.w - 1) return fetch(Lb, yy, ldw, lsh);
        const uint8_t* row = Lb + (size_t)yy * a.pitch;
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) v |= (u32)row[clampi(col + j, 0, a.w - 1)] << (8 * j);
        return (v & 0x7F7F7F7Fu) + 0x01010101u;
    };
    // right image, clamped-column rule: window column cc reads right column rofs + clamp(cc, -rofs, w - ndisp - rofs) + d.
    // Every column of a lane's quad that is clamped lands on the same clamp column, and a partly clamped quad contains its
    // clamp column, so one window lane per lane (tw) and a byte position per column (psh = 8 * position) express the rule.
    // Interior tiles (PF): tw = t, psh[q] = 8 q.
    int tw = t;
    u32 psh[4] = {0u, 8u, 16u, 24u};
    if constexpr (!PF) {
        const int rlo = -a.rofs, rhi = a.w - a.ndisp - a.rofs;
        const int off0 = clampi(c0 + 4 * t, rlo, rhi) - c0;
        tw = off0 >> 2;
#pragma unroll
        for (int q = 0; q < 4; q++) psh[q] = 8u * (u32)((clampi(c0 + 4 * t + q, rlo, rhi) - c0) & 3);
    }
    // The prefetch registers hold the RAW dword pairs: aligning, masking and biasing them is left to commit_to, one row later.  (Done at load
    // time -- as until late in round 3 -- the very next instruction needs the data, the compiler puts s_waitcnt vmcnt(0) right behind the loads
    // and the "prefetch" waits out the whole memory round trip every row; next to an HBM-bound kernel of another batch that is many microseconds:
    // a 1 GiB fill looping on a second stream stretched the matcher by 44 %, tools/interfere_probe.py.)
    struct PfRegs { u32 l_lo, l_hi; u32 r_lo[NLD], r_hi[NLD]; };
    PfRegs pfE = {}, pfS = {};        // the row entering the window next; (RF) the row leaving it next
    const int lcol_q = lcol0 + 4 * ti;
    const bool l_edge = !(lcol_q >= 0 && lcol_q + 3 <= a.w - 1);      // this lane's left quad straddles an image edge: fetch_l's byte path, final at once
    auto prefetch_to = [&](PfRegs& pf, int yy) {
        if (ti < 16) {
            if (l_edge) pf.l_lo = fetch_l(yy);
            else {
                const u32* row = (const u32*)(Lb + (size_t)yy * a.pitch);
                pf.l_lo = row[clampi(ldw, 0, ndw - 1)];
                pf.l_hi = row[clampi(ldw + 1, 0, ndw - 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; i++)
            if (ti + NTS * i < RWD) {
                const u32* row = (const u32*)(Rb + (size_t)yy * a.pitch);
                pf.r_lo[i] = row[clampi(rdw + NTS * i, 0, ndw - 1)];
                pf.r_hi[i] = row[clampi(rdw + NTS * i + 1, 0, ndw - 1)];
            }
    };
    auto finish = [&](u32 lo, u32 hi, int sh) -> u32 { return (__builtin_amdgcn_alignbyte(hi, lo, sh) & 0x7F7F7F7Fu) + 0x01010101u; };
    auto commit_to = [&](const PfRegs& pf, int sidx) {
        u32* slot = ring + sidx * NSUB * SLOTD;
        if (ti < 16) slot[ti] = l_edge ? pf.l_lo : finish(pf.l_lo, pf.l_hi, lsh);
#pragma unroll
        for (int i = 0; i < NLD; i++)
            if (ti + NTS * i < RWD) slot[16 + ti + NTS * i] = finish(pf.r_lo[i], pf.r_hi[i], rsh);
    };
    auto prefetch = [&](int yy) { prefetch_to(pfE, yy); };
    auto commit = [&](int yy) { commit_to(pfE, RF ? 0 : yy % NRING); };

    // ---- state
    // PF:  V[j][g] = prefix sum over columns 0..j of the vertical column sums (V[3] = T), TV[j] likewise for |L - ftzero|
    // !PF: V[q][g] = vertical column sum of column q
    // 4 x u16 per entry, kept as 64-bit values so that v_mqsad accumulates in place in an aligned register pair
    u64 V[4][NG];
    u32 TV[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        TV[q] = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) V[q][g] = 0;
    }
    const u32 ft1 = (u32)(a.ftzero + 1);
    const u32 ft4 = ft1 * 0x01010101u;
    constexpr u32 PMASK[4] = {0x000000FFu, 0x0000FFFFu, 0x00FFFFFFu, 0xFFFFFFFFu};

    // A staged row as this lane needs it: its left dword and the NG + 1 dwords of its right-image window.  Reading a row from the
    // LDS ring and using it are separate steps so that the reads can be issued well ahead of the v_mqsad that consume them: with
    // two waves per SIMD an exposed LDS round trip is idle VALU time (SQ_WAIT_ANY, profiles/).
    struct RowRegs { u32 lw; u32 rd[NG + 1]; };
    auto read_slot = [&](int sidx) -> RowRegs {
        const u32* slot = ring + sidx * NSUB * SLOTD;
        RowRegs r;
        r.lw = slot[t];
        const u32* rw = slot + 16 + tw + NG * s;
#pragma unroll
        for (int i = 0; i <= NG; i++) r.rd[i] = rw[i];
        return r;
    };
    auto read_row = [&](int yy) -> RowRegs { return read_slot(yy % NRING); };   // (!RF)
    auto add_regs = [&](const RowRegs& r) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 ref;
            if constexpr (PF) {
                ref = r.lw & PMASK[q];
                TV[q] = __builtin_amdgcn_sad_u8(ref, ft4 & PMASK[q], TV[q]);
            } else {
                const u32 lb = (r.lw >> (8 * q)) & 0xFFu;
                ref = lb << psh[q];
                TV[q] = __builtin_amdgcn_sad_u8(lb, ft1, TV[q]);
            }
#pragma unroll
            for (int g = 0; g < NG; g++) V[q][g] = __builtin_amdgcn_mqsad_pk_u16_u8(((u64)r.rd[g + 1] << 32) | r.rd[g], ref, V[q][g]);
        }
    };
    auto sub_regs = [&](const RowRegs& r) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 ref;
            if constexpr (PF) {
                ref = r.lw & PMASK[q];
                TV[q] -= __builtin_amdgcn_sad_u8(ref, ft4 & PMASK[q], 0u);
            } else {
                const u32 lb = (r.lw >> (8 * q)) & 0xFFu;
                ref = lb << psh[q];
                TV[q] -= __builtin_amdgcn_sad_u8(lb, ft1, 0u);
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const u64 tq = __builtin_amdgcn_mqsad_pk_u16_u8(((u64)r.rd[g + 1] << 32) | r.rd[g], ref, 0ull);
                // lane-wise V >= t, so two independent 32-bit subtracts are exact; a 64-bit subtract would chain every pair
                // through VCC (v_sub_co / v_subb) and serialise the whole row
                u32 lo, hi;
                asm("v_sub_u32 %0, %1, %2" : "=v"(lo) : "v"((u32)V[q][g]), "v"((u32)tq));
                asm("v_sub_u32 %0, %1, %2" : "=v"(hi) : "v"((u32)(V[q][g] >> 32)), "v"((u32)(tq >> 32)));
                V[q][g] = ((u64)hi << 32) | lo;
            }
        }
    };
    // Workgroup-wide ordering of the LDS traffic.  One wave per workgroup: the LDS executes a wave's instructions in issue order, so
    // all that is needed is that the compiler keeps the program order of the accesses (no s_waitcnt drain, no s_barrier).
    auto wg_sync = [&]() {
        if constexpr (NW > 1 || !C4_NOSYNC) __syncthreads();
        else asm volatile("" ::: "memory");
    };

    // uniqueness threshold  thr = ms + ms * uniq / 100  (integer division) without a 32-bit multiply and divide: with K = ceil(uniq * 2^23 / 100)
    // (ms * K) >> 23 == ms * uniq / 100 for every ms < 2^16 and uniq < 200 (the error term ms * delta / 2^23 is below 2^-7 and a fraction of
    // x / 100 is at most 0.99; checked exhaustively in tests/test_capi_symbols.py).  Both factors are masked to 24 bits, so the compiler emits
    // v_mul_u32_u24 + v_mul_hi_u32_u24 + v_alignbit_b32 instead of v_mul_lo_u32, v_mul_hi_u32 (magic division), shift and add.
    const u32 UK = (u32)((((u64)(u32)a.uniq << 23) + 99u) / 100u) & 0xFFFFFFu;     // gip_bm_c4_can: uniq < 200
    auto uniq_thr = [&](u32 ms) -> u32 { return ms + (u32)(((u64)(ms & 0xFFFFFFu) * (u64)UK) >> 23); };

    // ---- warm-up: rows ya-R .. ya+R-1 enter the window
    if constexpr (RF) {
        prefetch(ya - R);
        for (int yy = ya - R; yy < ya + R; yy++) {
            wg_sync();            // the previous row's reads of slot 0 are done
            commit(yy);
            wg_sync();
            const RowRegs W = read_slot(0);
            prefetch(yy + 1);     // (yy + 1 <= ya + R < h)
            add_regs(W);
        }
        prefetch_to(pfS, ya - R);
    } else {
        for (int yy = ya - R; yy < ya + R; yy++) {
            prefetch(yy);
            commit(yy);
        }
        wg_sync();
        for (int yy = ya - R; yy < ya + R; yy++) add_regs(read_row(yy));
        prefetch(ya + R);
    }

    // the pixel this lane finalises: column 4t + qf.  One wave: qf = lane / 16.  NW waves: every wave ends up with the merged keys,
    // counts and texture sums of all four columns of its lanes, so ONE wave can finalise the whole row (qf = lane / 16, every lane
    // active); the waves take turns row by row (ROT).  The other waves skip the block and wait at the next row's first barrier -- no
    // extra barrier, the same critical path, but three quarters of the finalisation's issue slots are left to other workgroups
    // (without ROT wave w takes q in [w*QPW, (w+1)*QPW) and its lanes with lane / 16 >= QPW idle through the whole block).
    constexpr bool ROT = (C4_ROTFIN != 0) && NW > 1;
    // Two sub-images per wave (NSUB = 2): the two lanes that share a column quad finalise two pixels each, columns 4t + 2 sl + {0, 1}.
    constexpr int NFIN = NSUB;                  // pixels a lane finalises per row
    const bool fin = NSUB == 2 ? true : (ROT ? true : (sl < QPW));
    const int qf0 = NSUB == 2 ? 2 * sl : (ROT ? sl : ((wave * QPW + sl) & 3));
    // cost slice rows ordered [q][live lane] (row = NLIVE q + t - TLO): the stores of consecutive lanes are CSPB = 4 (mod 32) dwords apart
    // (bank-conflict free).  The TLO lanes at either end of the row, which hold no output column, park all four of their columns in one
    // dump row each (never read): 6 TLO rows less LDS per workgroup -- at the metric's shape what lets a 10 KB workgroup of another
    // kernel sit next to eight matcher workgroups on a CU instead of displacing one.
    constexpr int TLO = R / 4, NLIVE = 16 - 2 * TLO;
    const bool live_lane = t >= TLO && t <= 15 - TLO;
    uint8_t* const dump = cs + (size_t)4 * NLIVE * CSPB;
    uint8_t* csw[4];
    static_for<0, 4>([&](auto qv) {
        constexpr int q = decltype(qv)::value;
        int doff = 0;         // this halo lane's bank-free slot for column q (compile-time table, 2 TLO entries)
        static_for<0, 2 * TLO>([&](auto hv) {
            constexpr int h = decltype(hv)::value, th = h < TLO ? h : 16 - 2 * TLO + h;
            if (t == th) doff = c4_dump_off(q, th, R, CSPB);
        });
        csw[q] = (live_lane ? cs + (size_t)(q * NLIVE + t - TLO) * CSPB : dump + 4 * doff) + (size_t)(4 * NG * s) * 2;
    });

    // Big one-wave kernels (NG >= 7: two waves per SIMD whatever they do, registers to spare) STORE the results of a row one iteration later,
    // right before the next row's loads are issued: the s_waitcnt vmcnt(0) in front of commit() then finds loads and stores that are a whole row
    // old instead of stores issued a few hundred cycles earlier (gfx9 counts loads and stores in the one vmcnt): +0.8 % pairs/s at the metric's
    // shape.  Everywhere else the five registers per pixel cost more than they buy -- the pair kernels (two pixels per lane) fall from three to
    // two waves per SIMD with them, -7 % alone at ndisp 48 / block 19.
    constexpr bool DEFER = NW == 1 && NSUB == 1 && NG >= 7;
    bool pend_w[NFIN];
    int pend_d[NFIN], pend_ms[NFIN];
#pragma unroll
    for (int jf = 0; jf < NFIN; jf++) { pend_w[jf] = false; pend_d[jf] = 0; pend_ms[jf] = 0; }
    auto store_px = [&](int yr, int jf, int d_out, int minsad) {
        const int c_out = c0 + 4 * t + qf0 + jf;
        disp[(size_t)yr * a.dpitch + a.lofs + c_out] = (int16_t)d_out;
        if (cost) {
            if (a.cost16) ((uint16_t*)cost)[(size_t)yr * a.cpitch + a.lofs + c_out] = (uint16_t)minsad;
            else cost[(size_t)yr * a.cpitch + a.lofs + c_out] = minsad;
        }
    };
    auto flush_row = [&](int yr) {
        if constexpr (DEFER) {
#pragma unroll
            for (int jf = 0; jf < NFIN; jf++) {
                if (pend_w[jf]) store_px(yr, jf, pend_d[jf], pend_ms[jf]);
                pend_w[jf] = false;
            }
        }
    };
    // One iteration (row y): commit row y+R to the ring -> issue its LDS reads (A), (DEFER) the stores of row y-1 and the global loads of row
    // y+R+1 -> add row y+R -> window sums / WTA / parks -> merge, uniqueness, finalise, store -> subtract row y-R (the row that leaves the window).
    for (int y = ya; y < yb; y++) {
        wg_sync();                // every wave is done with the slot row y+R replaces
        commit(y + R);
        if constexpr (RF) commit_to(pfS, 1);        // row y-R
        wg_sync();
        const RowRegs A = RF ? read_slot(0) : read_row(y + R);
        flush_row(y - 1);
        if (y + R + 1 < a.h) prefetch(y + R + 1);
        if constexpr (RF) prefetch_to(pfS, y - R + 1);
        add_regs(A);

        // ---- horizontal window sums, group keys, cost slice
        u32 best[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        u32 gmin[4][NG];          // min cost of each group (for the uniqueness count)
        u32 tsum_q[4];
        // group key of one (column, group); the group's four costs go to the cost slice in 16-byte stores (two groups of one
        // column per ds_write_b128: with CSPB = 4 (mod 32) dwords between lanes that store is bank-conflict free, an 8-byte one is not)
        auto wta = [&](auto qv, int g, P2 H) {
            constexpr int q = decltype(qv)::value;
            u32 m = pk_min_u16(H.lo, H.hi);
            u32 gm = umin(m & 0xFFFFu, m >> 16);
            gmin[q][g] = gm;
            best[q] = umin(best[q], (gm << 8) | (u32)(NG * s + g));
        };
        auto wta8 = [&](auto qv, int o, P2 Ha, P2 Hb) {
            constexpr int q = decltype(qv)::value;
            u32 m = pk_min_u16(pk_min_u16(Ha.lo, Ha.hi), pk_min_u16(Hb.lo, Hb.hi));
            u32 gm = umin(m & 0xFFFFu, m >> 16);
            gmin[q][o] = gm;
            best[q] = umin(best[q], (gm << 8) | (u32)(NGK * s + o));
        };
        auto park = [&](auto qv, int g, P2 H0, P2 H1, bool two) {
            constexpr int q = decltype(qv)::value;
            if constexpr (C4_ABLATE & 4) return;
            if (two) *(uint4*)(csw[q] + g * 8) = make_uint4(H0.lo, H0.hi, H1.lo, H1.hi);
            else *(uint2*)(csw[q] + g * 8) = make_uint2(H0.lo, H0.hi);
        };
        auto window_sums = [&](int g, P2 (&H)[4]) {
            if constexpr (C4_ABLATE & 2) {
                static_for<0, 4>([&](auto qv) { constexpr int q = decltype(qv)::value; H[q] = P2{(u32)V[q][g], (u32)(V[q][g] >> 32)}; });
            } else if constexpr (PF) {
                hsum_pf<R>((u32)V[3][g], (u32)V[0][g], (u32)V[1][g], (u32)V[2][g], H[0].lo, H[1].lo, H[2].lo, H[3].lo);
                hsum_pf<R>((u32)(V[3][g] >> 32), (u32)(V[0][g] >> 32), (u32)(V[1][g] >> 32), (u32)(V[2][g] >> 32), H[0].hi, H[1].hi, H[2].hi, H[3].hi);
            } else {
                P2 P[5];
                P[0] = {0u, 0u};
                P[1] = {(u32)V[0][g], (u32)(V[0][g] >> 32)};
                P[2] = P[1] + P2{(u32)V[1][g], (u32)(V[1][g] >> 32)};
                P[3] = P[2] + P2{(u32)V[2][g], (u32)(V[2][g] >> 32)};
                P[4] = P[3] + P2{(u32)V[3][g], (u32)(V[3][g] >> 32)};
                auto shift = [](auto dv, P2 v) -> P2 {
                    constexpr int d = decltype(dv)::value;
                    return P2{lane_at<d>(v.lo), lane_at<d>(v.hi)};
                };
                static_for<0, 4>([&](auto qv) { H[decltype(qv)::value] = hsum<R, decltype(qv)::value>(P, shift); });
            }
        };
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            const bool two = g + 1 < NG;
            P2 Ha[4], Hb[4] = {};
            window_sums(g, Ha);
            if (two) window_sums(g + 1, Hb);
            static_for<0, 4>([&](auto qv) {
                constexpr int q = decltype(qv)::value;
                if constexpr (OCT) wta8(qv, g / 2, Ha[q], Hb[q]);
                else {
                    wta(qv, g, Ha[q]);
                    if (two) wta(qv, g + 1, Hb[q]);
                }
                park(qv, g, Ha[q], Hb[q], two);
            });
        }
        if constexpr (PF) {
            hsum_pf<R>(TV[3], TV[0], TV[1], TV[2], tsum_q[0], tsum_q[1], tsum_q[2], tsum_q[3]);
        } else {
            // texture sums (32-bit, same window)
            u32 P[5];
            P[0] = 0; P[1] = TV[0]; P[2] = P[1] + TV[1]; P[3] = P[2] + TV[2]; P[4] = P[3] + TV[3];
            auto shift = [](auto dv, u32 v) -> u32 { return lane_at<decltype(dv)::value>(v); };
            static_for<0, 4>([&](auto qv) { constexpr int q = decltype(qv)::value; tsum_q[q] = hsum<R, q>(P, shift); });
        }
        // merge the four disparity quarters (lanes t, t+16, t+32, t+48)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if constexpr (C4_ABLATE & 8) break;
            best[q] = quarters_reduce<NSL>(best[q], [](u32 x, u32 y) { return umin(x, y); });
        }
        if constexpr (NW > 1) {
            if (sl == 0) *(uint4*)(xbest + (wave * 16 + t) * 4) = make_uint4(best[0], best[1], best[2], best[3]);
        }
        wg_sync();                // cost slice (and the waves' keys) complete
        if constexpr (NW > 1) {
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) {
                const uint4 o = *(const uint4*)(xbest + (w2 * 16 + t) * 4);
                best[0] = umin(best[0], o.x); best[1] = umin(best[1], o.y); best[2] = umin(best[2], o.z); best[3] = umin(best[3], o.w);
            }
        }

        // uniqueness: groups whose minimum is <= the per-column threshold, summed over the quarters.  The four per-column counts (at most
        // ndisp / 4 = 64 each) travel PACKED in one dword, one byte per column: two lane shuffles instead of eight -- as four separate
        // values the compiler serialised them into eight dependent LDS round trips per row.
        u32 cpk = 0;
        if (a.uniq > 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 ms = best[q] >> 8;
                const u32 thr = uniq_thr(ms);
                u32 c = 0;
#pragma unroll
                for (int g = 0; g < NGK; g++) c += (gmin[q][g] <= thr) ? 1u : 0u;
                cpk |= c << (8 * q);
            }
            if constexpr (!(C4_ABLATE & 8)) {
                cpk = quarters_reduce<NSL>(cpk, [](u32 x, u32 y) { return x + y; });
            }
            if constexpr (NW > 1) {
                if (sl == 0) xcnt[wave * 16 + t] = cpk;
                wg_sync();                // a.uniq is uniform: every wave takes this branch
                cpk = 0;
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) cpk += xcnt[w2 * 16 + t];
            }
        }
        // ---- finalise the lane's pixel(s) (column 4t + qf)
        if (!ROT || wave == ((y - ya) & (NW - 1))) {
#pragma unroll
        for (int jf = 0; jf < NFIN; jf++) {
        const int qf = qf0 + jf;
        const int my_col = 4 * t + qf;
        const int c_out = c0 + my_col;
        // a column is valid when its whole window lies inside the tile's 64 columns (column granularity, not lane granularity: for a radius that
        // is not a multiple of 4 the outermost live lanes still produce 4 - R % 4 ... valid columns each)
        const bool writer = fin && sub_live && (my_col >= R) && (my_col <= 63 - R) && (c_out >= a.cbeg) && (c_out < a.cend);
        const uint8_t* csrow_r = live_lane ? cs + (size_t)(qf * NLIVE + t - TLO) * CSPB : dump;
        const u32 key = qf == 0 ? best[0] : qf == 1 ? best[1] : qf == 2 ? best[2] : best[3];
        const int tsum = (int)(qf == 0 ? tsum_q[0] : qf == 1 ? tsum_q[1] : qf == 2 ? tsum_q[2] : tsum_q[3]);
        const int ngle = (int)((cpk >> (8 * qf)) & 0xFFu);
        const int minsad = (int)(key >> 8), gstar = (int)(key & 0xFFu);
        int d_out = a.filtered;
        if constexpr (C4_ABLATE & 16) d_out = minsad + gstar + tsum + ngle;
        else {
            // The winner's group and its two neighbours in ONE LDS round trip: 12 costs as six packed dwords D0..D5, entry
            // 4 + k is the winner (k = first minimum inside its group), entries 3 + k / 5 + k its sub-pixel neighbours.
            // Groups beyond either end of the range read as 0xFFFF (never <= a threshold; a real cost is < 0xFFFF).
            const u32 ms = (u32)minsad;
            u32 D0, D1, D2, D3, D4, D5;
            int g4;                                   // 4-disparity group that holds the winner
            // octet mode: the costs of the two octets' halves that lie outside (D0..D5), and the minima of the three octets read
            u32 other = ~0u, om_m = ~0u, om_0 = ~0u, om_p = ~0u;
            if constexpr (OCT) {
                constexpr int NOT = NS * NGK;
                const bool has_m = gstar > 0, has_p = gstar + 1 < NOT;
                uint4 A = *(const uint4*)(csrow_r + (has_m ? gstar - 1 : gstar) * 16);
                const uint4 B = *(const uint4*)(csrow_r + gstar * 16);
                uint4 Cn = *(const uint4*)(csrow_r + (has_p ? gstar + 1 : gstar) * 16);
                // keep all three reads in ONE round trip: without this the compiler sinks the halves only the uniqueness test uses into its branch
                asm volatile("" : "+v"(A.x), "+v"(A.y), "+v"(Cn.z), "+v"(Cn.w));
                if (!has_m) A = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (!has_p) Cn = make_uint4(~0u, ~0u, ~0u, ~0u);
                const u32 mlo = pk_min_u16(B.x, B.y);
                const bool low = umin(mlo & 0xFFFFu, mlo >> 16) == ms;     // the first minimum lies in the octet's lower quad
                g4 = 2 * gstar + (low ? 0 : 1);
                D0 = low ? A.z : B.x; D1 = low ? A.w : B.y;
                D2 = low ? B.x : B.z; D3 = low ? B.y : B.w;
                D4 = low ? B.z : Cn.x; D5 = low ? B.w : Cn.y;
                // the 12 costs of the three octets that are not in D0..D5 lie entirely outside [mind - 1, mind + 1]
                other = pk_min_u16(pk_min_u16(pk_min_u16(A.x, A.y), pk_min_u16(Cn.z, Cn.w)), pk_min_u16(low ? Cn.x : A.z, low ? Cn.y : A.w));
                om_m = pk_min_u16(pk_min_u16(A.x, A.y), pk_min_u16(A.z, A.w));
                om_0 = pk_min_u16(mlo, pk_min_u16(B.z, B.w));
                om_p = pk_min_u16(pk_min_u16(Cn.x, Cn.y), pk_min_u16(Cn.z, Cn.w));
            } else {
                constexpr int NGT = NS * NG;
                const bool has_m = gstar > 0, has_p = gstar + 1 < NGT;
                const uint2 gm_ = *(const uint2*)(csrow_r + (has_m ? gstar - 1 : gstar) * 8);
                const uint2 g0_ = *(const uint2*)(csrow_r + gstar * 8);
                const uint2 gp_ = *(const uint2*)(csrow_r + (has_p ? gstar + 1 : gstar) * 8);
                D0 = has_m ? gm_.x : ~0u; D1 = has_m ? gm_.y : ~0u; D2 = g0_.x; D3 = g0_.y; D4 = has_p ? gp_.x : ~0u; D5 = has_p ? gp_.y : ~0u;
                g4 = gstar;
            }
            const u32 h1 = D1 >> 16, h2 = D2 & 0xFFFFu, h3 = D2 >> 16, h4 = D3 & 0xFFFFu, h5 = D3 >> 16, h6 = D4 & 0xFFFFu;
            const int k = (h2 == ms) ? 0 : (h3 == ms) ? 1 : (h4 == ms) ? 2 : 3;   // first minimum
            const int mind = 4 * g4 + k;
            bool ok = tsum >= a.tex_thr;
            if (a.uniq > 0) {
                // exact part of the uniqueness test on the 12 costs, packed: "some u16 of x is <= thr"  <=>  pk_min(x, T1) != T1
                // with T1 = thr + 1 in both halves
                const u32 thr = uniq_thr(ms);
                const u32 t1 = umin(thr + 1u, 0xFFFFu), T1 = t1 | (t1 << 16);
                // key groups among the three read whose minimum is <= thr (what the count over ALL key groups must not exceed)
                int touch_le;
                if constexpr (OCT) touch_le = (pk_min_u16(om_m, T1) != T1) + (pk_min_u16(om_0, T1) != T1) + (pk_min_u16(om_p, T1) != T1);
                else touch_le = (pk_min_u16(pk_min_u16(D0, D1), T1) != T1) + (pk_min_u16(pk_min_u16(D2, D3), T1) != T1) + (pk_min_u16(pk_min_u16(D4, D5), T1) != T1);
                // blank entries 3+k .. 5+k (the winner and its neighbours): a 48-bit window sliding over (D1, D2, D3, D4)
                const u64 wlo = 0xFFFFFFFFFFFF0000ull << (16 * k), whi = 0x0000FFFFFFFFFFFFull >> (16 * (3 - k));
                const u32 E1 = D1 | (u32)wlo, E2 = D2 | (u32)(wlo >> 32), E3 = D3 | (u32)whi, E4 = D4 | (u32)(whi >> 32);
                const u32 rest = pk_min_u16(pk_min_u16(pk_min_u16(pk_min_u16(D0, E1), pk_min_u16(E2, E3)), pk_min_u16(E4, D5)), other);
                const bool viol = pk_min_u16(rest, T1) != T1;
                // a group that does not touch the winner lies entirely outside [mind-1, mind+1]
                ok = ok && !viol && (ngle <= touch_le);
            }
            {
                u32 n = k == 0 ? h1 : k == 1 ? h2 : k == 2 ? h3 : h4;
                u32 p = k == 0 ? h3 : k == 1 ? h4 : k == 2 ? h5 : h6;
                const u32 n0 = n;
                if (mind == 0) n = p;                    // OpenCV: the missing neighbour is replaced by the other one
                if (mind + 1 == NDISP) p = n0;
                // (p - n) * 256 / den, C truncation; den = 2 * (max(p, n) - minsad) >= 2 |p - n|, so |quotient| <= 128 and
                // |p - n| * 256 < 2^24: one float reciprocal plus a one-step correction is exact
                const u32 ad = p > n ? p - n : n - p;
                const u32 den = p + n - 2u * ms + ad;
                const u32 num = ad << 8;
                // v_rcp_f32 (1 ulp) instead of the correctly rounded reciprocal (ten instructions): the quotient is at most 128, so q0 is within
                // one of the exact value and the correction step below repairs either direction
                u32 q0 = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
                const int r = (int)(num - q0 * den);
                q0 += (r >= (int)den) ? 1u : 0u;
                q0 -= (r < 0) ? 1u : 0u;
                const int quot = den != 0 ? (p >= n ? (int)q0 : -(int)q0) : 0;
                const int v = (NDISP - mind - 1 + a.mindisp) * 256 + quot + 15;
                if (ok) d_out = v >> 4;
            }
        }
        if constexpr (DEFER) {
            pend_w[jf] = writer;
            pend_d[jf] = d_out;
            pend_ms[jf] = minsad;
        } else {
            if (writer) store_px(y, jf, d_out, minsad);
        }
        }   // the lane's pixels
        }   // finalising wave
        if (!(C4_ABLATE & 1)) sub_regs(RF ? read_slot(1) : read_row(y - R));
    }
    flush_row(yb - 1);
}

template <int NG, int R, int NW, bool RF, int NSUB = 1>
__global__ void __launch_bounds__(64 * NW, C4_MINW) k_bm_c4(BmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    // XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so with the natural order
    // neighbouring tiles -- which share 2/3 of their right-image window and all of their rows -- would land on 8 different L2s.
    // Instead XCD x (= workgroup id mod 8) walks the CONTIGUOUS range [x * M, (x + 1) * M) of the (image, chunk, tile) space, tile
    // fastest: the overlap of adjacent tiles and the warm-up rows of adjacent chunks are then L2 hits (FETCH_SIZE per launch, KITTI
    // workload: profiles/traffic.json).  Placement is a speed matter only; nothing depends on it.
    int tile, chunk, img;
#if C4_XCD_MAP
    {
        const int nwg = a.ntiles * a.nchunks * a.nimg, M = (nwg + 7) >> 3;
        const int v = (int)(blockIdx.x & 7u) * M + (int)(blockIdx.x >> 3);
        if (v >= nwg) return;
        tile = v % a.ntiles;
        const int rest = v / a.ntiles;
        chunk = rest % a.nchunks;
        img = rest / a.nchunks;
    }
#else
    tile = blockIdx.x; chunk = blockIdx.y; img = blockIdx.z;
#endif
    const int c0 = a.cbeg - R + tile * (64 - 2 * R);   // relative column of lane t=0, q=0: tiles overlap by exactly 2R columns
    const int ya = a.row0 + chunk * a.rows_per_chunk;
    const int yb = min(ya + a.rows_per_chunk, a.row1);
    if (ya >= yb) return;
    // interior tile: no column of the tile is touched by the right-image clamp (rofs + clamp(cc, -rofs, w - ndisp - rofs) + d)
    const bool interior = c0 >= -a.rofs && c0 + 63 <= a.w - a.ndisp - a.rofs;
#ifdef C4_NO_PF
    bm_c4_tile<NG, R, NW, false, RF, NSUB>(a, smem_raw, c0, ya, yb, img);
#else
    if (interior) bm_c4_tile<NG, R, NW, true, RF, NSUB>(a, smem_raw, c0, ya, yb, img);
    else bm_c4_tile<NG, R, NW, false, RF, NSUB>(a, smem_raw, c0, ya, yb, img);
#endif
}

template <int NG, int R, int NW, bool RF, int NSUB = 1>
int launch_c4_rf(const BmArgs& a, int batch_images, hipStream_t st, const BmLaunchCtx* ctx)
{
    const int batch = (batch_images + NSUB - 1) / NSUB;      // work items along the image axis: images, or pairs of images
    // measurement knobs (tools/chunk_sweep.sh; compiled in only with -DC4_MEASURE_KNOBS, never in the product build -- a stray environment
    // variable must not change occupancy or chunking in production): GIP_C4_LDS_PAD=n allocates n more bytes of LDS per workgroup,
    // GIP_C4_CHUNKS=n forces the chunk count; both read once
#ifdef C4_MEASURE_KNOBS
    static const int env_pad = [] { const char* e = getenv("GIP_C4_LDS_PAD"); return e ? atoi(e) : 0; }();
    static const int env_chunks = [] { const char* e = getenv("GIP_C4_CHUNKS"); return e ? atoi(e) : 0; }();
#else
    constexpr int env_pad = 0, env_chunks = 0;
#endif
    size_t sh = c4_lds_bytes(NG, R, NW, RF ? 2 : 2 * R + 1, NSUB) + (size_t)(env_pad > 0 ? env_pad : 0);
    const int ncols = a.cend - a.cbeg;
    const int ntiles = (ncols + (64 - 2 * R) - 1) / (64 - 2 * R);
    const int nrows = a.row1 - a.row0;
    // Row chunks.  A CU holds min(32 / NW, floor(160 KiB / LDS per workgroup), what the VGPRs allow) workgroups.  Pick the chunk count
    // that minimises  (rounds + 0.5) x (rows per chunk + warm-up cost): whole "rounds" of resident workgroups matter because a partially
    // filled last round idles most of the chip, every chunk re-reads 2R warm-up rows (each worth about 0.4 of a full row: SAD only,
    // no winner-take-all), and the workgroups do not all take the same time (border tiles run the slower column form), which costs
    // about half a workgroup's duration at the end whatever the fill.  The constants are fitted to forced-chunk sweeps of seven shapes
    // (tools/chunk_sweep.sh, profiles/r02_chunk_sweep.txt): the pick is within 2 % of the best forced count on each.
    // per DEVICE (a process may own several: MultiDeviceStereo, gip_rowsplit_*), filled once each; racing first calls write the same value
    static std::atomic<int> n_cu_of[GIP_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GIP_MAX_DEVICES) dev = 0;
    int n_cu = n_cu_of[dev].load(std::memory_order_relaxed);
    if (n_cu == 0) {
        int v = 0;
        n_cu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        n_cu_of[dev].store(n_cu, std::memory_order_relaxed);
    }
    int per_cu = c4_per_cu(sh, NW);
    // ... and no more waves per SIMD than the kernel's register count leaves room for (512 VGPRs per lane and SIMD, allocated in blocks of 8)
    // (the register count is a property of the code object, the same on every gfx950 device: one value per instantiation)
    static std::atomic<int> vgpr_alloc_c{0};
    int vgpr_alloc = vgpr_alloc_c.load(std::memory_order_relaxed);
    if (vgpr_alloc == 0) {
        hipFuncAttributes fa;
        vgpr_alloc = 128;
        if (hipFuncGetAttributes(&fa, (const void*)k_bm_c4<NG, R, NW, RF, NSUB>) == hipSuccess && fa.numRegs > 0) vgpr_alloc = (fa.numRegs + 7) & ~7;
        vgpr_alloc_c.store(vgpr_alloc, std::memory_order_relaxed);
    }
    const int vgpr_waves = 512 / vgpr_alloc < 1 ? 1 : (512 / vgpr_alloc > 8 ? 8 : 512 / vgpr_alloc);
    if (per_cu > 4 * vgpr_waves / NW) per_cu = 4 * vgpr_waves / NW;
    if (per_cu < 1) per_cu = 1;
    // Co-scheduled (other batches' kernels run next to this launch): a matcher that fills every SIMD's register file (4 waves x 128 VGPRs at
    // nd 256 / block 21) leaves no room for a single wave of the small kernels that feed it and take its results -- they then run only in the
    // cracks between matcher workgroups and the STEP gets longer although the matcher gets shorter (4 instead of 3 workgroups per CU: matcher
    // -10 %, pipeline -3.4 %; capped: pipeline +1.3 %).  So: never more than 90 % of the CU's registers, by padding the LDS request just far
    // enough that one workgroup less fits (the rest of the LDS stays free for the other kernels).  Shapes below that line are left alone:
    // a stricter rule (always leave 128 registers per SIMD) cost nd 192 / block 15 11 % and nd 48 / block 19 7 % (tools/bench_ab.sh).
    if (ctx && ctx->co_scheduled) {
        int cap = per_cu;
        while (cap > 1 && cap * NW * vgpr_alloc * 10 > 4 * 512 * 9) cap--;
        if (cap < per_cu) {
            per_cu = cap;
            // the SMALLEST request of which cap + 1 no longer fit
            const size_t granule = 1280, want = ((size_t)160 * 1024 / (cap + 1)) / granule * granule + granule;
            if (want > sh) sh = want;
        }
    }
#ifdef C4_FIXED_CHUNKS
    int nch = C4_FIXED_CHUNKS;
#else
    const double capacity = (double)n_cu * per_cu;
    int nch = 1;
    double best_cost = 1e300;
    const int max_nch = nrows / (2 * R + 2) > 0 ? nrows / (2 * R + 2) : 1;
    for (int c = 1; c <= max_nch && c <= 64; c++) {
        const int rpc = (nrows + c - 1) / c;
        const int real_c = (nrows + rpc - 1) / rpc;
        const double wgs = (double)ntiles * real_c * batch;
        const double rounds = wgs / capacity < 1.0 ? 1.0 : ceil(wgs / capacity - 1e-9);
        const double cost = (rounds + 0.5) * (rpc + 0.4 * 2 * R);
        if (cost < best_cost - 1e-9) { best_cost = cost; nch = real_c; }
    }
    if (env_chunks > 0) nch = env_chunks < max_nch ? env_chunks : max_nch;
#endif
    BmArgs b = a;
    b.rows_per_chunk = (nrows + nch - 1) / nch;
    nch = (nrows + b.rows_per_chunk - 1) / b.rows_per_chunk;
    b.ntiles = ntiles; b.nchunks = nch; b.nimg = batch; b.batch = batch_images;
#if C4_XCD_MAP
    const dim3 grid((unsigned)(8 * (((size_t)ntiles * nch * batch + 7) / 8)));
#else
    const dim3 grid(ntiles, nch, batch);
#endif
    if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_c4<NG, R, NW, RF, NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    // the matcher may run on a stream of its own (BmLaunchCtx::matcher_stream), fenced by events on both sides
    hipStream_t ks = st;
    const bool forked = ctx && ctx->matcher_stream && ctx->matcher_stream != st;
    if (forked) {
        ks = ctx->matcher_stream;
        GIP_HIP_CHECK(hipEventRecord(ctx->fork, st));
        GIP_HIP_CHECK(hipStreamWaitEvent(ks, ctx->fork, 0));
    }
    // Events attached to the dispatch itself (hipExtLaunchKernelGGL): the timing pair brackets the kernel's own start and end timestamps (what
    // rocprofv3 reports; recorded around the launch they include 25-30 us of the command processor walking from packet to packet), and the stop
    // event doubles as the join event of a forked launch -- one packet less on the matcher stream between two matcher launches.
    hipEvent_t ev_start = ctx ? ctx->ev0 : nullptr;
    hipEvent_t ev_stop = ctx ? ctx->ev1 : nullptr;
    if (forked && !ev_stop) ev_stop = ctx->join;
    if (ev_start || ev_stop) hipExtLaunchKernelGGL((k_bm_c4<NG, R, NW, RF, NSUB>), grid, dim3(64 * NW), sh, ks, ev_start, ev_stop, 0, b);
    else hipLaunchKernelGGL((k_bm_c4<NG, R, NW, RF, NSUB>), grid, dim3(64 * NW), sh, ks, b);
    GIP_LAUNCH_CHECK();
    if (forked) {
        if (ctx->defer_join) ctx->pending_join = ev_stop;
        else GIP_HIP_CHECK(hipStreamWaitEvent(st, ev_stop, 0));
    }
    return GIP_OK;
}

// The refetch form is the faster kernel on its own (profiles/r02_matcher_refetch.txt) but the slower one inside the throughput pipeline,
// where the matcher of one batch runs next to the small latency-bound kernels of two others: the extra matcher workgroups per CU crowd
// those out and the STEP gets 1.5-3 % longer although the matcher gets 2-11 % shorter (same-box A/B, tools/bench_ab.sh).  So: refetch
// when the matcher is launched on the caller's own stream (single calls, gip_process_pair, one batch in flight), the ring form when the
// caller hands over a separate matcher stream (BmLaunchCtx::matcher_stream -- the batch pipeline with several batches in flight).
template <int NG, int R, int NW, int NSUB = 1>
int launch_c4(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx)
{
    if constexpr (c4_refetch(NG, R, NW, NSUB)) {
        const bool pipelined = ctx && (ctx->co_scheduled || (ctx->matcher_stream && ctx->matcher_stream != st));
        // (re-measured for the two-images-per-wave form at (48, 19), whose ring costs it a third of its workgroups per CU: refetch inside the
        // pipeline = matcher 0.886 -> 0.795 ms, step 0.919 -> 0.985 ms, 69.6 -> 64.9 k pairs/s: the rule holds)
        if (!pipelined) return launch_c4_rf<NG, R, NW, true, NSUB>(a, batch, st, ctx);
    }
    return launch_c4_rf<NG, R, NW, false, NSUB>(a, batch, st, ctx);
}

// C4_PAIR: 1 = disparity counts up to 64 run two images of the batch per wave (NSUB = 2) whenever the batch has at least two
#ifndef C4_PAIR
#define C4_PAIR 1
#endif

// (groups per slice, waves) for a disparity count: ndisp = 16 * NG * NW.  The wave counts are measured choices (tools/bm_ab.py,
// DESIGN.md 3.1): 2 or 4 waves are 12 % slower at ndisp 128 (any batch size), 2 waves are 38 % faster at ndisp 192 (4 waves: 28 %),
// 4 waves 25 % faster at ndisp 256 (2 waves: 12 %); the other multiples of 16 take one wave when ndisp / 16 is odd or <= 8.
template <int R>
int dispatch_ndisp(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled)
{
    *handled = true;
#if C4_PAIR
    // Two images per wave (NSUB = 2) for ndisp <= 48 when the batch is large enough to fill the chip with half as many workgroups
    // (profiles/r03_pair_kernel.txt: (48, 19) 254.5 M -> 227.5 M wave-instructions, -3 % alone, +2.6 % pairs/s pipelined; ndisp 64 needs NG = 8,
    // 189 VGPRs = two waves per SIMD, and LOSES 7 %; batches of 3 or 5 images lose 20-40 %: too few workgroups).
    if (batch >= 8) {
        switch (a.ndisp) {
        case 16: return launch_c4<2, R, 1, 2>(a, batch, st, ctx);
        case 32: return launch_c4<4, R, 1, 2>(a, batch, st, ctx);
        case 48: return launch_c4<6, R, 1, 2>(a, batch, st, ctx);
        default: break;
        }
    }
#endif
    switch (a.ndisp) {
    case 16: return launch_c4<1, R, 1>(a, batch, st, ctx);
    case 32: return launch_c4<2, R, 1>(a, batch, st, ctx);
    case 48: return launch_c4<3, R, 1>(a, batch, st, ctx);
    case 64: return launch_c4<4, R, 1>(a, batch, st, ctx);
    case 80: return launch_c4<5, R, 1>(a, batch, st, ctx);
    case 96: return launch_c4<6, R, 1>(a, batch, st, ctx);
    case 112: return launch_c4<7, R, 1>(a, batch, st, ctx);
#ifdef C4_CFG_128
    case 128: return launch_c4<C4_CFG_128>(a, batch, st, ctx);
#else
    case 128: return launch_c4<8, R, 1>(a, batch, st, ctx);
#endif
    case 144: return launch_c4<9, R, 1>(a, batch, st, ctx);
    case 160: return launch_c4<5, R, 2>(a, batch, st, ctx);
    case 176: return launch_c4<11, R, 1>(a, batch, st, ctx);
#ifdef C4_CFG_192
    case 192: return launch_c4<C4_CFG_192>(a, batch, st, ctx);
#else
    case 192: return launch_c4<6, R, 2>(a, batch, st, ctx);
#endif
    case 208: return launch_c4<13, R, 1>(a, batch, st, ctx);
    case 224: return launch_c4<7, R, 2>(a, batch, st, ctx);
    case 240: return launch_c4<15, R, 1>(a, batch, st, ctx);
#ifdef C4_CFG_256
    case 256: return launch_c4<C4_CFG_256>(a, batch, st, ctx);
#else
    case 256: return launch_c4<4, R, 4>(a, batch, st, ctx);
#endif
    default: *handled = false; return GIP_OK;
    }
}

}  // namespace

#define GIP_C4_RANGE_TU(NAME, ...)                                                                                             \
    int NAME(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled)                                \
    {                                                                                                                          \
        *handled = false;                                                                                                      \
        switch (a.wsz2) { __VA_ARGS__ default: return GIP_OK; }                                                                \
    }
#define GIP_C4_CASE(R) case R: return dispatch_ndisp<R>(a, batch, st, ctx, handled);
