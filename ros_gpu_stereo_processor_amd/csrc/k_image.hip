// k_image.hip -- colour conversion, fixed-point remap and StereoBM prefilters for gfx950.
//
// Reference behaviour being replaced (file:line relative to the reference tree):
//   convertColor        src/GPUStereoProcessor.cpp:119-172  (cv::cuda::cvtColor / GpuMat::convertTo)
//   rectifyImage        src/GPUStereoProcessor.cpp:236-262  (remap; the CPU fixed-point form is reproduced bit-exactly)
//   StereoBM prefilter  src/GPUStereoProcessor.cpp:312-321  (cv::StereoBM::compute -> prefilterXSobel / prefilterNorm)
// All kernels are HBM-bound element-wise / gather passes: one thread per output pixel (or 4 for mono), rows
// coalesced across the 64 lanes of a wavefront, batch in blockIdx.z.
#include "gip_internal.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------- colour
__global__ void k_bgr_to_gray(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                              int src_is_rgb, size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = src + blockIdx.z * sbs + (size_t)y * sp + 3 * x;
    int c0 = s[0], c1 = s[1], c2 = s[2];
    int b = src_is_rgb ? c2 : c0, r = src_is_rgb ? c0 : c2;
    dst[blockIdx.z * dbs + (size_t)y * dp + x] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + 8192) >> 14);
}

__global__ void k_gray_to_bgr(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                              size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint8_t v = src[blockIdx.z * sbs + (size_t)y * sp + x];
    uint8_t* d = dst + blockIdx.z * dbs + (size_t)y * dp + 3 * x;
    d[0] = v; d[1] = v; d[2] = v;
}

__global__ void k_swap_rb(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                          size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = src + blockIdx.z * sbs + (size_t)y * sp + 3 * x;
    uint8_t* d = dst + blockIdx.z * dbs + (size_t)y * dp + 3 * x;
    uint8_t a = s[0], b = s[1], c = s[2];
    d[0] = c; d[1] = b; d[2] = a;
}

// colour of a Bayer site as BGR channel index (0 B, 1 G, 2 R); pattern numbering = GIP_BAYER_*
__device__ __forceinline__ int bayer_site(int pattern, int y, int x)
{
    // 2x2 tiles {rggb, bggr, gbrg, grbg} packed 2 bits per site, row-major
    const unsigned tiles[4] = {2u | (1u << 2) | (1u << 4) | (0u << 6), 0u | (1u << 2) | (1u << 4) | (2u << 6),
                               1u | (0u << 2) | (2u << 4) | (1u << 6), 1u | (2u << 2) | (0u << 4) | (1u << 6)};
    return (tiles[pattern] >> (2 * ((y & 1) * 2 + (x & 1)))) & 3;
}

// Border rule of the CPU demosaic: column 0 / w-1 copy columns 1 / w-2, then row 0 / h-1 copy rows 1 / h-2.
// Equivalent gather form: output (y,x) = interior value at (clamp(y,1,h-2), clamp(x,1,w-2)).
// BY_ROWS pixels of one column per thread (independent pixels: their loads are in flight together; see k_remap_r4).
constexpr int BY_ROWS = 4;
template <bool GRAY>
__device__ __forceinline__ void bayer_pixel(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h, int pattern,
                                            size_t sbs, size_t dbs, int x, int y)
{
    uint8_t* d = dst + blockIdx.z * dbs + (size_t)y * dp + (GRAY ? x : 3 * x);
    if (w <= 2 || h <= 2) {
        if (GRAY) d[0] = 0; else { d[0] = 0; d[1] = 0; d[2] = 0; }
        return;
    }
    int yy = clampi(y, 1, h - 2), xx = clampi(x, 1, w - 2);
    const uint8_t* s = src + blockIdx.z * sbs + (size_t)yy * sp + xx;
    int up = s[-(ptrdiff_t)sp], dn = s[sp], lf = s[-1], rt = s[1], c = s[0];
    int site = bayer_site(pattern, yy, xx);
    if (GRAY) {
        const int coef[3] = {1868, 9617, 4899};
        unsigned v;
        if (site != 1) {
            unsigned diag = s[-(ptrdiff_t)sp - 1] + s[-(ptrdiff_t)sp + 1] + s[sp - 1] + s[sp + 1];
            unsigned cross = up + dn + lf + rt;
            v = (diag * coef[2 - site] + cross * 9617u + (unsigned)c * 4u * coef[site] + (1u << 15)) >> 16;
        } else {
            int hsite = bayer_site(pattern, yy, xx + 1);
            v = ((unsigned)(up + dn) * coef[2 - hsite] + (unsigned)(lf + rt) * coef[hsite] + (unsigned)c * 2u * 9617u + (1u << 14)) >> 15;
        }
        d[0] = (uint8_t)v;
    } else {
        int o0, o1, o2;
        if (site != 1) {
            int diag = (s[-(ptrdiff_t)sp - 1] + s[-(ptrdiff_t)sp + 1] + s[sp - 1] + s[sp + 1] + 2) >> 2;
            int cross = (up + dn + lf + rt + 2) >> 2;
            o1 = cross;
            if (site == 0) { o0 = c; o2 = diag; } else { o2 = c; o0 = diag; }
        } else {
            int vert = (up + dn + 1) >> 1, horz = (lf + rt + 1) >> 1;
            int hsite = bayer_site(pattern, yy, xx + 1);
            o1 = c;
            if (hsite == 0) { o0 = horz; o2 = vert; } else { o2 = horz; o0 = vert; }
        }
        d[0] = (uint8_t)o0; d[1] = (uint8_t)o1; d[2] = (uint8_t)o2;
    }
}

template <bool GRAY>
__global__ void k_bayer(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h, int pattern,
                        size_t sbs, size_t dbs)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = (blockIdx.y * blockDim.y + threadIdx.y) * BY_ROWS;
    if (x >= w) return;
#pragma unroll
    for (int j = 0; j < BY_ROWS; j++)
        if (y0 + j < h) bayer_pixel<GRAY>(src, sp, dst, dp, w, h, pattern, sbs, dbs, x, y0 + j);
}

__global__ void k_mono8_to_mono16(const uint8_t* __restrict__ src, size_t sp, uint16_t* __restrict__ dst, size_t dp, int w, int h,
                                  size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float scale = (float)(65535. / 255.);
    float v = __builtin_rintf(src[blockIdx.z * sbs + (size_t)y * sp + x] * scale);
    v = v < 0.f ? 0.f : (v > 65535.f ? 65535.f : v);
    ((uint16_t*)((uint8_t*)dst + blockIdx.z * dbs + (size_t)y * dp))[x] = (uint16_t)v;
}

__global__ void k_mono16_to_mono8(const uint16_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                                  size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float scale = (float)(255. / 65535.);
    float v = __builtin_rintf(((const uint16_t*)((const uint8_t*)src + blockIdx.z * sbs + (size_t)y * sp))[x] * scale);
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    dst[blockIdx.z * dbs + (size_t)y * dp + x] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------- remap
// cv::remap(CV_16SC2 + CV_16UC1, INTER_LINEAR, BORDER_CONSTANT 0): weights (32-ay)(32-ax)*32..., (sum + 2^14) >> 15.
// Gather kernel: 6 B/px of map traffic + 4 taps (L2/L1 hits for mild distortion) + CN B/px store.
// 4 pixels per thread (same column, 4 consecutive rows).  A gather kernel like this one is latency bound, not
// bandwidth bound: its time is (waves / resident wave slots) x (map load -> tap load round trip).  Four pixels per thread
// issue 8 map loads and then 16 taps back to back, so a quarter of the waves pay the same two round trips.  The lanes of a
// wave stay on ADJACENT columns: every load instruction still touches one or two cache lines (4 adjacent columns per lane
// were measured 30 % slower than the scalar kernel: 4x the lines per instruction).
#ifndef RM_ROWS_N
#define RM_ROWS_N 4
#endif
constexpr int RM_ROWS = RM_ROWS_N;
template <int CN, int IM>
__global__ void k_remap_r4(const uint8_t* __restrict__ src, size_t sp, int sw, int sh, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                           const int16_t* __restrict__ map1, const uint16_t* __restrict__ map2, size_t sbs, size_t dbs, int nearest)
{
    // IM images of the batch per thread: the table entry, the tap offsets, the border predicates and the blend weights of a
    // pixel position are the same for every image of a batch, only the taps differ -- decode once, gather IM times.
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = (blockIdx.y * blockDim.y + threadIdx.y) * RM_ROWS;
    if (x >= w || y0 >= h) return;
    int sx[RM_ROWS], sy[RM_ROWS], fr[RM_ROWS];
#pragma unroll
    for (int j = 0; j < RM_ROWS; j++) {
        const int yy = min(y0 + j, h - 1);
        const size_t o = (size_t)yy * w + x;
        const short2 t = ((const short2*)map1)[o];
        sx[j] = t.x;
        sy[j] = t.y;
        fr[j] = map2[o] & 1023;
    }
    const uint8_t* s0 = src + (size_t)blockIdx.z * IM * sbs;
    uint8_t* d0 = dst + (size_t)blockIdx.z * IM * dbs + (size_t)y0 * dp + (size_t)x * CN;
    if (nearest) {
#pragma unroll
        for (int j = 0; j < RM_ROWS; j++) {
            const int nx = sx[j] + ((fr[j] & 31) >= 16), ny = sy[j] + ((fr[j] >> 5) >= 16);
            const bool in = (unsigned)nx < (unsigned)sw && (unsigned)ny < (unsigned)sh;
#pragma unroll
            for (int im = 0; im < IM; im++)
#pragma unroll
                for (int k = 0; k < CN; k++) {
                    const uint8_t v = in ? s0[im * sbs + (size_t)ny * sp + nx * CN + k] : (uint8_t)0;
                    if (y0 + j < h) d0[im * dbs + (size_t)j * dp + k] = v;
                }
        }
        return;
    }
    // (a branch-free variant -- taps loaded from clamped coordinates and zeroed by select -- executes 23 % MORE vector
    // instructions than these predicated loads: the compiler turns the predicates into exec-mask branches on the scalar unit)
    int p[IM][RM_ROWS][CN][4];
#pragma unroll
    for (int j = 0; j < RM_ROWS; j++) {
        const bool bx0 = (unsigned)sx[j] < (unsigned)sw, bx1 = (unsigned)(sx[j] + 1) < (unsigned)sw;
        const bool by0 = (unsigned)sy[j] < (unsigned)sh, by1 = (unsigned)(sy[j] + 1) < (unsigned)sh;
        const ptrdiff_t off = (ptrdiff_t)sy[j] * (ptrdiff_t)sp + (ptrdiff_t)sx[j] * CN;
#pragma unroll
        for (int im = 0; im < IM; im++) {
            const uint8_t* q = s0 + im * sbs + off;
#pragma unroll
            for (int k = 0; k < CN; k++) {
                p[im][j][k][0] = (bx0 && by0) ? q[k] : 0;
                p[im][j][k][1] = (bx1 && by0) ? q[CN + k] : 0;
                p[im][j][k][2] = (bx0 && by1) ? q[sp + k] : 0;
                p[im][j][k][3] = (bx1 && by1) ? q[sp + CN + k] : 0;
            }
        }
    }
    // cv::remap's fixed-point blend  (sum p*w + 2^14) >> 15  with w00 = (32-ay)(32-ax)*32 ... factors exactly into two 1-D blends:
    // sum p*w = 32 * T,  T = (32-ay) * (p00 (32-ax) + p01 ax) + ay * (p10 (32-ax) + p11 ax),  and (32 T + 2^14) >> 15 = (T + 512) >> 10
    // (T <= 255 * 1024, so the result never exceeds 255).
#pragma unroll
    for (int j = 0; j < RM_ROWS; j++) {
        const int ax = fr[j] & 31, ay = fr[j] >> 5, bx = 32 - ax, by = 32 - ay;
#pragma unroll
        for (int im = 0; im < IM; im++)
#pragma unroll
            for (int k = 0; k < CN; k++) {
                const int top = p[im][j][k][0] * bx + p[im][j][k][1] * ax, bot = p[im][j][k][2] * bx + p[im][j][k][3] * ax;
                const int v = (top * by + bot * ay + 512) >> 10;
                if (y0 + j < h) d0[im * dbs + (size_t)j * dp + k] = (uint8_t)v;
            }
    }
}

// ---------------------------------------------------------------------------------------------- prefilters
// XSOBEL: 3x3 Sobel-x with reflect-101 rows, columns 0 / w-1 and (odd h) the last row forced to cap.
__global__ void k_prefilter_xsobel(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h, int cap,
                                   size_t sbs, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint8_t* d = dst + blockIdx.z * dbs + (size_t)y * dp + x;
    if (x == 0 || x == w - 1 || ((h & 1) && y == h - 1)) { *d = (uint8_t)cap; return; }
    int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0);
    int yp = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const uint8_t* s = src + blockIdx.z * sbs;
    const uint8_t *r0 = s + (size_t)ym * sp + x, *r1 = s + (size_t)y * sp + x, *r2 = s + (size_t)yp * sp + x;
    int v = (r0[1] - r0[-1]) + 2 * (r1[1] - r1[-1]) + (r2[1] - r2[-1]);
    *d = (uint8_t)(clampi(v, -cap, cap) + cap);
}

// XSOBEL, 4 columns x XS_ROWS rows per thread: per source row three aligned dwords (previous / own / next quad) give the four
// horizontal differences; the rows roll through the 1-2-1 vertical sum, so XS_ROWS + 2 rows are read for XS_ROWS rows written
// (one dword store each).  Like every streaming pass here it is bound by (waves / wave slots) x load latency, hence the
// 16 pixels per thread with all loads of a row issued together.
// Requires 4-byte aligned rows (base, pitch, batch stride); the launcher falls back to the scalar kernel otherwise.
#ifndef XS_ROWS_N
#define XS_ROWS_N 4
#endif
constexpr int XS_ROWS = XS_ROWS_N;
__global__ void k_prefilter_xsobel_v4(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h, int cap,
                                      size_t sbs, size_t dbs)
{
    const int xq = blockIdx.x * blockDim.x + threadIdx.x, yb = (blockIdx.y * blockDim.y + threadIdx.y) * XS_ROWS;
    const int x0 = xq * 4;
    if (x0 >= w || yb >= h) return;
    const int nq = (int)(sp >> 2);                      // dwords per source row
    const uint8_t* s = src + blockIdx.z * sbs;
    // horizontal differences (right neighbour - left neighbour) of the quad in source row yy
    int hd[XS_ROWS + 2][4];
#pragma unroll
    for (int k = 0; k < XS_ROWS + 2; k++) {
        int yy = yb - 1 + k;
        yy = yy < 0 ? (h > 1 ? 1 : 0) : (yy > h - 1 ? (h > 1 ? 2 * (h - 1) - yy : 0) : yy);   // reflect-101
        yy = yy < 0 ? 0 : yy;                                                                 // rows past the image (not written)
        const uint32_t* row = (const uint32_t*)(s + (size_t)yy * sp);
        const uint32_t a = xq > 0 ? row[xq - 1] : 0u, b = row[xq], c = (xq + 1 < nq) ? row[xq + 1] : 0u;
        const int p_m1 = (int)(a >> 24), p0 = (int)(b & 0xFF), p1 = (int)((b >> 8) & 0xFF), p2 = (int)((b >> 16) & 0xFF), p3 = (int)(b >> 24),
                  p4 = (int)(c & 0xFF);
        hd[k][0] = p1 - p_m1;
        hd[k][1] = p2 - p0;
        hd[k][2] = p3 - p1;
        hd[k][3] = p4 - p2;
    }
#pragma unroll
    for (int r = 0; r < XS_ROWS; r++) {
        const int y = yb + r;
        if (y >= h) break;
        uint32_t out = 0;
        if ((h & 1) && y == h - 1) {
            out = (uint32_t)cap * 0x01010101u;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                const int acc = hd[r][j] + 2 * hd[r + 1][j] + hd[r + 2][j];
                const int v = (x == 0 || x >= w - 1) ? cap : clampi(acc, -cap, cap) + cap;
                out |= (uint32_t)v << (8 * j);
            }
        }
        uint8_t* d = dst + blockIdx.z * dbs + (size_t)y * dp + x0;
        if (x0 + 3 < w) *(uint32_t*)d = out;
        else
            for (int j = 0; x0 + j < w; j++) d[j] = (uint8_t)(out >> (8 * j));
    }
}

// XSOBEL in packed 16-bit arithmetic: the same 4 columns x XS_ROWS rows per thread and the same three aligned dwords per source row as
// k_prefilter_xsobel_v4, but the four horizontal differences of a row are formed as two packed i16 pairs -- three byte permutes
// (v_perm_b32) put (p1,p2), (p-1,p0), (p3,p4) into 16-bit lanes, two v_pk_sub_i16 subtract -- and the 1-2-1 vertical sum, the clamp
// to [-cap, cap], the +cap bias and the repacking to bytes stay packed (v_pk_add_i16, v_pk_max_i16 / v_pk_min_i16, one v_perm_b32):
// about 6 lane-instructions per pixel instead of 22.  This kernel runs next to the VALU-bound matcher of another batch, where the
// step is bound by the sum of everybody's vector instructions.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__global__ void k_prefilter_xsobel_pk(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h, int cap,
                                      size_t sbs, size_t dbs)
{
    const int xq = blockIdx.x * blockDim.x + threadIdx.x, yb = (blockIdx.y * blockDim.y + threadIdx.y) * XS_ROWS;
    const int x0 = xq * 4;
    if (x0 >= w || yb >= h) return;
    const int nq = (int)(sp >> 2);                      // dwords per source row
    // 32-bit dword indices from one base pointer per image (scalar base + 32-bit vector offset addressing; an image is < 2^31 bytes)
    const uint32_t* s32 = (const uint32_t*)(src + blockIdx.z * sbs);
    // horizontal differences of source row yy as two packed pairs: lo = (hd0, hd1), hi = (hd2, hd3)
    s16x2 hlo[XS_ROWS + 2], hhi[XS_ROWS + 2];
#pragma unroll
    for (int k = 0; k < XS_ROWS + 2; k++) {
        int yy = yb - 1 + k;
        yy = yy < 0 ? (h > 1 ? 1 : 0) : (yy > h - 1 ? (h > 1 ? 2 * (h - 1) - yy : 0) : yy);   // reflect-101
        yy = yy < 0 ? 0 : yy;                                                                 // rows past the image (not written)
        const int i0 = yy * nq + xq;
        const uint32_t a = xq > 0 ? s32[i0 - 1] : 0u, b = s32[i0], c = (xq + 1 < nq) ? s32[i0 + 1] : 0u;
        // v_perm_b32(hi, lo, sel): selector byte 0-3 = byte of lo, 4-7 = byte of hi, 0x0c = constant 0
        const s16x2 p12 = as_s16x2(__builtin_amdgcn_perm(0u, b, 0x0c020c01u));     // (p1, p2)
        const s16x2 pm0 = as_s16x2(__builtin_amdgcn_perm(b, a, 0x0c040c03u));      // (p-1, p0): a byte 3, b byte 0
        const s16x2 p34 = as_s16x2(__builtin_amdgcn_perm(c, b, 0x0c040c03u));      // (p3, p4): b byte 3, c byte 0
        hlo[k] = p12 - pm0;
        hhi[k] = p34 - p12;
    }
    // s_k = hd[k] + hd[k+1];  hd[r] + 2 hd[r+1] + hd[r+2] = s_r + s_{r+1}
    s16x2 slo[XS_ROWS + 1], shi[XS_ROWS + 1];
#pragma unroll
    for (int k = 0; k < XS_ROWS + 1; k++) {
        slo[k] = hlo[k] + hlo[k + 1];
        shi[k] = hhi[k] + hhi[k + 1];
    }
    const s16x2 capv = {(short)cap, (short)cap}, ncapv = {(short)-cap, (short)-cap};
#pragma unroll
    for (int r = 0; r < XS_ROWS; r++) {
        const int y = yb + r;
        if (y >= h) break;
        uint32_t out;
        if ((h & 1) && y == h - 1) {
            out = (uint32_t)cap * 0x01010101u;
        } else {
            s16x2 lo = slo[r] + slo[r + 1], hi = shi[r] + shi[r + 1];
            lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, ncapv), capv) + capv;
            hi = __builtin_elementwise_min(__builtin_elementwise_max(hi, ncapv), capv) + capv;
            out = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);   // low bytes of the four lanes
            if (x0 == 0) out = (out & 0xFFFFFF00u) | (uint32_t)cap;                    // column 0 = cap
            if (x0 + 3 >= w - 1) {                                                     // column w-1 (and beyond) = cap
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (x0 + j >= w - 1) out = (out & ~(0xFFu << (8 * j))) | ((uint32_t)cap << (8 * j));
            }
        }
        uint8_t* d = dst + blockIdx.z * dbs + (unsigned)(y * (int)dp + x0);
        if (x0 + 3 < w) *(uint32_t*)d = out;
        else
            for (int j = 0; x0 + j < w; j++) d[j] = (uint8_t)(out >> (8 * j));
    }
}

// NORMALIZED_RESPONSE: ((4c + l + r + u + d) * scale_g - boxsum * scale_s) >> 10, replicate borders.
// NORMALIZED_RESPONSE with separable box sums: a workgroup owns 256 columns (+ the 2r halo columns) and walks down a chunk
// of rows; every thread keeps the vertical sum of the 2r+1 window rows for its one or two columns in a register (one pixel
// enters, one leaves per row), parks it in LDS (double buffered: one barrier per row) and the output pixel adds up the 2r+1
// column sums around it.  O(winsize) LDS reads per pixel instead of the O(winsize^2) global reads of k_prefilter_norm
// (0.75 ms -> see DESIGN.md for 64 images at winsize 9).
constexpr int PN_TPB = 256;
constexpr int PN_ROWS = 32;
__global__ void __launch_bounds__(PN_TPB) k_prefilter_norm_sep(const uint8_t* __restrict__ src, size_t sp, uint8_t* __restrict__ dst, size_t dp, int w, int h,
                                                                int winsize, int cap, size_t sbs, size_t dbs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    const int r = winsize / 2, ncol = PN_TPB + 2 * r;          // ncol <= 510: at most two columns per thread
    int* colsum = (int*)smem_raw;                               // [2][ncol]
    const int t = threadIdx.x, x0 = blockIdx.x * PN_TPB;
    const int ya = blockIdx.y * PN_ROWS, yb = min(ya + PN_ROWS, h);
    const uint8_t* s = src + blockIdx.z * sbs;
    int scale_g = winsize * winsize / 8, scale_s = (1024 + scale_g) / (scale_g * 2);
    scale_g *= scale_s;
    // my columns (clamped to the image: replicate border)
    const int c0 = clampi(x0 - r + t, 0, w - 1), c1 = clampi(x0 - r + t + PN_TPB, 0, w - 1);
    const bool has1 = t + PN_TPB < ncol;
    int v0 = 0, v1 = 0;
    for (int dy = -r; dy <= r; dy++) {
        const uint8_t* row = s + (size_t)clampi(ya + dy, 0, h - 1) * sp;
        v0 += row[c0];
        if (has1) v1 += row[c1];
    }
    const int x = x0 + t;
    for (int y = ya; y < yb; y++) {
        int* cs = colsum + ((y - ya) & 1) * ncol;
        cs[t] = v0;
        if (has1) cs[t + PN_TPB] = v1;
        __syncthreads();
        if (x < w) {
            int sum = 0;
            for (int k = 0; k <= 2 * r; k++) sum += cs[t + k];
            const uint8_t* cur = s + (size_t)y * sp;
            const int c = cur[x], l = cur[clampi(x - 1, 0, w - 1)], rr = cur[clampi(x + 1, 0, w - 1)];
            const int u = s[(size_t)clampi(y - 1, 0, h - 1) * sp + x], dn = s[(size_t)clampi(y + 1, 0, h - 1) * sp + x];
            const int val = ((c * 4 + l + rr + u + dn) * scale_g - sum * scale_s) >> 10;
            dst[blockIdx.z * dbs + (size_t)y * dp + x] = (uint8_t)(clampi(val, -cap, cap) + cap);
        }
        // slide the window one row down
        const uint8_t* rin = s + (size_t)clampi(y + r + 1, 0, h - 1) * sp;
        const uint8_t* rout = s + (size_t)clampi(y - r, 0, h - 1) * sp;
        v0 += (int)rin[c0] - (int)rout[c0];
        if (has1) v1 += (int)rin[c1] - (int)rout[c1];
    }
}

// ---------------------------------------------------------------------------------------------- fused rectify + XSOBEL
// One pass raw -> prefiltered plane (and, when asked, the rectified plane as a by-product): the front end of "the remap+SAD kernel"
// of the hot path (reference src/GPUStereoProcessor.cpp:236-250 rectifyImage, then :312-321 StereoBM's prefilter).  The rectified
// image never makes the HBM round trip between k_remap_r4 and k_prefilter_xsobel_v4.
//   * a workgroup = 4 waves stacked vertically over a strip of 62 output columns (lanes 1..62; lanes 0 and 63 rectify the strip's
//     two halo columns); wave ty rectifies RX_ROWS consecutive rows, all of them in ONE gather step exactly like k_remap_r4 (table
//     entries, then all taps in flight together; one table decode serves IM images of the batch) -- a gather kernel is bound by
//     (waves / wave slots) x round trips, so the thread-level parallelism of the plain remap is kept (a first version that marched
//     down the rows with carried state was 3.6x slower than the two kernels it replaced: 9 dependent gather steps per wave);
//   * the horizontal difference r[x+1] - r[x-1] of a rectified row comes from the neighbour lanes (two lane shuffles); the vertical
//     1-2-1 of the Sobel kernel needs the rows above and below: inside a thread's RX_ROWS rows they are its own registers, at the
//     seam between two waves the first / last row of differences goes through LDS (one barrier).  The block's first and last rows
//     are halo: 4 * RX_ROWS rectified rows give 4 * RX_ROWS - 2 output rows.
// Bit-exact with k_remap_r4 followed by k_prefilter_xsobel (reflect-101 rows, columns 0 / w-1 and an odd last row = cap).
#ifndef RX_ROWS_N
#define RX_ROWS_N 5
#endif
constexpr int RX_ROWS = RX_ROWS_N;
constexpr int RX_OUT = 4 * RX_ROWS - 2;     // output rows per workgroup
constexpr int RX_COLS = 62;
template <int IM>
__global__ void __launch_bounds__(256) k_rectify_xsobel(const uint8_t* __restrict__ src, size_t sp, int sw, int sh, uint8_t* __restrict__ pf, size_t pfp,
                                                        uint8_t* __restrict__ rect, size_t rp, int w, int h, const int16_t* __restrict__ map1,
                                                        const uint16_t* __restrict__ map2, size_t sbs, size_t pfbs, size_t rbs, int cap)
{
    __shared__ short seam[2][4][IM][64];                    // [0]: first row of differences of wave ty, [1]: last row
    const int lane = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * RX_COLS - 1 + lane;
    const int xc = clampi(x, 0, w - 1);
    const bool out_lane = lane >= 1 && lane <= RX_COLS && x < w;
    const int ya = blockIdx.y * RX_OUT, yb = min(ya + RX_OUT, h);   // output rows of the block
    const int r0 = ya - 1 + ty * RX_ROWS;                            // first virtual row of this wave
    const uint8_t* s0 = src + (size_t)blockIdx.z * IM * sbs;
    uint8_t* pf0 = pf + (size_t)blockIdx.z * IM * pfbs;
    uint8_t* rc0 = rect ? rect + (size_t)blockIdx.z * IM * rbs : nullptr;
    int sx[RX_ROWS], sy[RX_ROWS], fr[RX_ROWS];
#pragma unroll
    for (int j = 0; j < RX_ROWS; j++) {
        int r = r0 + j;                                     // virtual row: reflect-101 at the image edge
        r = r < 0 ? (h > 1 ? 1 : 0) : (r > h - 1 ? (h > 1 ? 2 * (h - 1) - r : 0) : r);
        r = clampi(r, 0, h - 1);                            // rows far past the image: no output needs them
        const size_t o = (size_t)r * w + xc;
        const short2 t = ((const short2*)map1)[o];
        sx[j] = t.x;
        sy[j] = t.y;
        fr[j] = map2[o] & 1023;
    }
    int p[IM][RX_ROWS][4];
#pragma unroll
    for (int j = 0; j < RX_ROWS; j++) {
        const bool bx0 = (unsigned)sx[j] < (unsigned)sw, bx1 = (unsigned)(sx[j] + 1) < (unsigned)sw;
        const bool by0 = (unsigned)sy[j] < (unsigned)sh, by1 = (unsigned)(sy[j] + 1) < (unsigned)sh;
        const ptrdiff_t off = (ptrdiff_t)sy[j] * (ptrdiff_t)sp + (ptrdiff_t)sx[j];
#pragma unroll
        for (int im = 0; im < IM; im++) {
            const uint8_t* q = s0 + im * sbs + off;
            p[im][j][0] = (bx0 && by0) ? q[0] : 0;
            p[im][j][1] = (bx1 && by0) ? q[1] : 0;
            p[im][j][2] = (bx0 && by1) ? q[sp] : 0;
            p[im][j][3] = (bx1 && by1) ? q[sp + 1] : 0;
        }
    }
    int hd[IM][RX_ROWS];
#pragma unroll
    for (int j = 0; j < RX_ROWS; j++) {
        const int r = r0 + j;
        const int ax = fr[j] & 31, ay = fr[j] >> 5, bx = 32 - ax, by = 32 - ay;
#pragma unroll
        for (int im = 0; im < IM; im++) {
            const int top = p[im][j][0] * bx + p[im][j][1] * ax, bot = p[im][j][2] * bx + p[im][j][3] * ax;
            const int v = (top * by + bot * ay + 512) >> 10;                // cv::remap fixed point, see k_remap_r4
            if (rc0 && out_lane && r >= ya && r < yb) rc0[im * rbs + (size_t)r * rp + x] = (uint8_t)v;
            hd[im][j] = __shfl_down(v, 1) - __shfl_up(v, 1);                // r[x+1] - r[x-1]
        }
    }
#pragma unroll
    for (int im = 0; im < IM; im++) {
        seam[0][ty][im][lane] = (short)hd[im][0];
        seam[1][ty][im][lane] = (short)hd[im][RX_ROWS - 1];
    }
    __syncthreads();
    const bool odd_last = (h & 1) != 0;
#pragma unroll
    for (int j = 0; j < RX_ROWS; j++) {
        const int y = r0 + j;
        if (y < ya || y >= yb || !out_lane) continue;       // the block's halo rows, rows past the image, halo lanes
#pragma unroll
        for (int im = 0; im < IM; im++) {
            const int up = j > 0 ? hd[im][j - 1] : (int)seam[1][ty > 0 ? ty - 1 : 0][im][lane];
            const int dn = j < RX_ROWS - 1 ? hd[im][j + 1] : (int)seam[0][ty < 3 ? ty + 1 : 3][im][lane];
            const int acc = up + 2 * hd[im][j] + dn;
            const int o = (x == 0 || x == w - 1 || (odd_last && y == h - 1)) ? cap : clampi(acc, -cap, cap) + cap;
            pf0[im * pfbs + (size_t)y * pfp + x] = (uint8_t)o;
        }
    }
}

inline dim3 grid2d(int w, int h, int batch) { return dim3(gip_div_up(w, TPB), h, batch); }

}  // namespace

#define GIPK_ARGCHECK(cond) do { if (!(cond)) return GIP_ERR_INVALID_ARG; } while (0)

extern "C" {

int gipk_cvt_bgr_to_gray(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int src_is_rgb, int batch,
                         size_t sbs, size_t dbs, gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0);
    hipLaunchKernelGGL(k_bgr_to_gray, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, src_is_rgb, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_gray_to_bgr(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int batch, size_t sbs, size_t dbs,
                         gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0);
    hipLaunchKernelGGL(k_gray_to_bgr, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_swap_rb(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int batch, size_t sbs, size_t dbs,
                     gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0);
    hipLaunchKernelGGL(k_swap_rb, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_bayer_to_bgr(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int pattern, int batch, size_t sbs,
                          size_t dbs, gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0 && pattern >= 0 && pattern < 4);
    hipLaunchKernelGGL(k_bayer<false>, dim3(gip_div_up(w, 64), gip_div_up(gip_div_up(h, BY_ROWS), 4), batch), dim3(64, 4), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, pattern, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_bayer_to_gray(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int pattern, int batch, size_t sbs,
                           size_t dbs, gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0 && pattern >= 0 && pattern < 4);
    hipLaunchKernelGGL(k_bayer<true>, dim3(gip_div_up(w, 64), gip_div_up(gip_div_up(h, BY_ROWS), 4), batch), dim3(64, 4), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, pattern, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_mono8_to_mono16(const uint8_t* src, size_t src_pitch, uint16_t* dst, size_t dst_pitch, int w, int h, int batch, size_t sbs, size_t dbs,
                             gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0);
    hipLaunchKernelGGL(k_mono8_to_mono16, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_cvt_mono16_to_mono8(const uint16_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int batch, size_t sbs, size_t dbs,
                             gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0);
    hipLaunchKernelGGL(k_mono16_to_mono8, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, sbs, dbs);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_remap_u8(const uint8_t* src, size_t src_pitch, int sw, int sh, uint8_t* dst, size_t dst_pitch, int w, int h, int cn, const int16_t* map1,
                  const uint16_t* map2, int batch, size_t sbs, size_t dbs, gip_stream stream)
{
    return gipk_remap_u8_interp(src, src_pitch, sw, sh, dst, dst_pitch, w, h, cn, map1, map2, GIP_INTER_LINEAR, batch, sbs, dbs, stream);
}

int gipk_remap_u8_interp(const uint8_t* src, size_t src_pitch, int sw, int sh, uint8_t* dst, size_t dst_pitch, int w, int h, int cn, const int16_t* map1,
                         const uint16_t* map2, int interpolation, int batch, size_t sbs, size_t dbs, gip_stream stream)
{
    if (interpolation != GIP_INTER_LINEAR && interpolation != GIP_INTER_NEAREST) return GIP_ERR_UNSUPPORTED;
    const int nearest = interpolation == GIP_INTER_NEAREST;
    GIPK_ARGCHECK(src && dst && map1 && map2 && w > 0 && h > 0 && sw > 0 && sh > 0 && batch > 0 && (cn == 1 || cn == 3));
    GIPK_ARGCHECK(((uintptr_t)map1 & 3) == 0 && ((uintptr_t)map2 & 1) == 0);
    const bool quad = cn == 1 && batch % 4 == 0;   // four images of the batch per thread (mono)
    const dim3 g4(gip_div_up(w, 64), gip_div_up(gip_div_up(h, RM_ROWS), 4), quad ? batch / 4 : batch), b4(64, 4);
    if (quad)
        hipLaunchKernelGGL((k_remap_r4<1, 4>), g4, b4, 0, (hipStream_t)stream, src, src_pitch, sw, sh, dst, dst_pitch, w, h, map1, map2, sbs, dbs, nearest);
    else if (cn == 1)
        hipLaunchKernelGGL((k_remap_r4<1, 1>), g4, b4, 0, (hipStream_t)stream, src, src_pitch, sw, sh, dst, dst_pitch, w, h, map1, map2, sbs, dbs, nearest);
    else
        hipLaunchKernelGGL((k_remap_r4<3, 1>), g4, b4, 0, (hipStream_t)stream, src, src_pitch, sw, sh, dst, dst_pitch, w, h, map1, map2, sbs, dbs, nearest);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_prefilter(const uint8_t* src, size_t src_pitch, uint8_t* dst, size_t dst_pitch, int w, int h, int prefilter_type, int prefilter_size,
                   int prefilter_cap, int batch, size_t sbs, size_t dbs, gip_stream stream)
{
    GIPK_ARGCHECK(src && dst && w > 0 && h > 0 && batch > 0 && prefilter_cap >= 1 && prefilter_cap <= 63);
    if (prefilter_type == GIP_PREFILTER_XSOBEL) {
        const bool aligned = ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0) && src_pitch % 4 == 0 && dst_pitch % 4 == 0 && sbs % 4 == 0 && dbs % 4 == 0 &&
                             src_pitch >= (size_t)((w + 3) & ~3);
#ifndef XS_SCALAR
        if (aligned)
            hipLaunchKernelGGL(k_prefilter_xsobel_pk, dim3(gip_div_up(gip_div_up(w, 4), 64), gip_div_up(gip_div_up(h, XS_ROWS), 4), batch), dim3(64, 4), 0, (hipStream_t)stream, src, src_pitch, dst,
                               dst_pitch, w, h, prefilter_cap, sbs, dbs);
#else
        if (aligned)
            hipLaunchKernelGGL(k_prefilter_xsobel_v4, dim3(gip_div_up(gip_div_up(w, 4), 64), gip_div_up(gip_div_up(h, XS_ROWS), 4), batch), dim3(64, 4), 0, (hipStream_t)stream, src, src_pitch, dst,
                               dst_pitch, w, h, prefilter_cap, sbs, dbs);
#endif
        else
            hipLaunchKernelGGL(k_prefilter_xsobel, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, prefilter_cap, sbs, dbs);
    } else if (prefilter_type == GIP_PREFILTER_NORMALIZED_RESPONSE) {
        GIPK_ARGCHECK(prefilter_size >= 5 && prefilter_size <= 255 && (prefilter_size & 1));
        hipLaunchKernelGGL(k_prefilter_norm_sep, dim3(gip_div_up(w, PN_TPB), gip_div_up(h, PN_ROWS), batch), dim3(PN_TPB), (size_t)2 * (PN_TPB + 2 * (prefilter_size / 2)) * sizeof(int),
                           (hipStream_t)stream, src, src_pitch, dst, dst_pitch, w, h, prefilter_size, prefilter_cap, sbs, dbs);
    } else {
        return GIP_ERR_INVALID_ARG;
    }
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_rectify_xsobel(const uint8_t* src, size_t src_pitch, int sw, int sh, uint8_t* prefiltered, size_t pf_pitch, uint8_t* rectified, size_t rect_pitch,
                        int w, int h, const int16_t* map1, const uint16_t* map2, int prefilter_cap, int batch, size_t sbs, size_t pf_bs, size_t rect_bs,
                        gip_stream stream)
{
    GIPK_ARGCHECK(src && prefiltered && map1 && map2 && w > 0 && h > 0 && sw > 0 && sh > 0 && batch > 0 && prefilter_cap >= 1 && prefilter_cap <= 63);
    GIPK_ARGCHECK(((uintptr_t)map1 & 3) == 0 && ((uintptr_t)map2 & 1) == 0);
#ifndef RX_IM
#define RX_IM 2    /* measured (tools/fe_ab.py): 2 images x 5 rows per thread 113 us, 4 x 5: 149 us, 1 x 5: 125 us per 64 images of 1242x375 */
#endif
    const bool multi = batch % RX_IM == 0 && RX_IM > 1;
    const dim3 grid(gip_div_up(w, RX_COLS), gip_div_up(h, RX_OUT), multi ? batch / RX_IM : batch), block(64, 4);
    if (multi)
        hipLaunchKernelGGL((k_rectify_xsobel<RX_IM>), grid, block, 0, (hipStream_t)stream, src, src_pitch, sw, sh, prefiltered, pf_pitch, rectified, rect_pitch, w, h, map1, map2,
                           sbs, pf_bs, rect_bs, prefilter_cap);
    else
        hipLaunchKernelGGL((k_rectify_xsobel<1>), grid, block, 0, (hipStream_t)stream, src, src_pitch, sw, sh, prefiltered, pf_pitch, rectified, rect_pitch, w, h, map1, map2,
                           sbs, pf_bs, rect_bs, prefilter_cap);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

}  // extern "C"
