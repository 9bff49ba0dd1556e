/*
 * gip_oracle.c -- CPU restatement of the reference's stereo hot path (plain C99 + OpenMP).
 *
 * TEST INFRASTRUCTURE ONLY -- see gip_oracle.h for the rules and the parity status.
 *
 * Every function cites the reference call site it stands in for (paths relative to
 * /root/reference) and, where the arithmetic lives in un-vendored OpenCV 3.x, the OpenCV
 * module whose published algorithm is restated.  No reference or OpenCV source is copied.
 */
#include "gip_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int mini(int a, int b) { return a < b ? a : b; }
static inline int maxi(int a, int b) { return a > b ? a : b; }

int gipo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================
 * Colour conversion.  Reference: GpuStereoProcessor::convertColor
 * (src/GPUStereoProcessor.cpp:119-172) -> cv_bridge::getConversionCode -> cv::cuda::cvtColor.
 * Restated: OpenCV imgproc color.cpp (RGB2Gray fixed point) and demosaicing.cpp (bilinear).
 * ====================================================================================== */

/* gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14 ; pinned bit-exact by aloe-L-Mono.png */
void gipo_bgr_to_gray(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int src_is_rgb)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * src_step;
        uint8_t *d       = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++) {
            int c0 = s[3 * x], c1 = s[3 * x + 1], c2 = s[3 * x + 2];
            int b = src_is_rgb ? c2 : c0, r = src_is_rgb ? c0 : c2;
            d[x] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + 8192) >> 14);
        }
    }
}

void gipo_gray_to_bgr(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * src_step;
        uint8_t *d       = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
    }
}

void gipo_swap_rb(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * src_step;
        uint8_t *d       = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++) {
            uint8_t a = s[3 * x], b = s[3 * x + 1], c = s[3 * x + 2];
            d[3 * x] = c, d[3 * x + 1] = b, d[3 * x + 2] = a;
        }
    }
}

/* colour of a Bayer site: 0 = B, 1 = G, 2 = R (BGR channel index) */
static inline int bayer_site(int pattern, int y, int x)
{
    /* 2x2 tile written row-major, as BGR channel indices */
    static const int tiles[4][4] = {
        {2, 1, 1, 0}, /* rggb */
        {0, 1, 1, 2}, /* bggr */
        {1, 0, 2, 1}, /* gbrg */
        {1, 2, 0, 1}, /* grbg */
    };
    return tiles[pattern][(y & 1) * 2 + (x & 1)];
}

/* Bilinear demosaic, interior pixels from their 3x3 neighbourhood:
 *   non-green site: own colour = centre, opposite colour = (4 diagonals + 2) >> 2, green = (4 cross + 2) >> 2
 *   green site    : (2 vertical + 1) >> 1 -> colour of the vertical neighbours, likewise horizontal
 * then first/last column copied from the adjacent interior column, then first/last row copied from the
 * adjacent row (OpenCV demosaicing.cpp, Bayer2RGB_). */
void gipo_bayer_to_bgr(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int pattern)
{
    if (h <= 2 || w <= 2) {
        for (int y = 0; y < h; y++) memset(dst + (size_t)y * dst_step, 0, (size_t)w * 3);
        return;
    }
    for (int y = 1; y < h - 1; y++) {
        const uint8_t *s = src + (size_t)y * src_step;
        uint8_t *d       = dst + (size_t)y * dst_step;
        for (int x = 1; x < w - 1; x++) {
            int site = bayer_site(pattern, y, x);
            int c    = s[x];
            int out[3];
            if (site != 1) {
                int diag  = (s[x - 1 - src_step] + s[x + 1 - src_step] + s[x - 1 + src_step] + s[x + 1 + src_step] + 2) >> 2;
                int cross = (s[x - src_step] + s[x + src_step] + s[x - 1] + s[x + 1] + 2) >> 2;
                out[site]     = c;
                out[2 - site] = diag;
                out[1]        = cross;
            } else {
                int vert  = (s[x - src_step] + s[x + src_step] + 1) >> 1;
                int horz  = (s[x - 1] + s[x + 1] + 1) >> 1;
                int hsite = bayer_site(pattern, y, x + 1); /* colour of the horizontal neighbours */
                out[1]         = c;
                out[hsite]     = horz;
                out[2 - hsite] = vert;
            }
            d[3 * x] = (uint8_t)out[0], d[3 * x + 1] = (uint8_t)out[1], d[3 * x + 2] = (uint8_t)out[2];
        }
        for (int k = 0; k < 3; k++) {
            d[k]               = d[3 + k];
            d[3 * (w - 1) + k] = d[3 * (w - 2) + k];
        }
    }
    memcpy(dst, dst + dst_step, (size_t)w * 3);
    memcpy(dst + (size_t)(h - 1) * dst_step, dst + (size_t)(h - 2) * dst_step, (size_t)w * 3);
}

/* Single-rounding Bayer -> gray (OpenCV demosaicing.cpp Bayer2Gray_, scalar form):
 *   non-green: (diag4*coef(opposite) + cross4*9617 + centre*4*coef(own) + 2^15) >> 16
 *   green    : (vert2*coef(vertical colour) + horz2*coef(horizontal colour) + centre*2*9617 + 2^14) >> 15 */
void gipo_bayer_to_gray(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int pattern)
{
    static const int coef[3] = {1868, 9617, 4899}; /* B, G, R */
    if (h <= 2 || w <= 2) {
        for (int y = 0; y < h; y++) memset(dst + (size_t)y * dst_step, 0, (size_t)w);
        return;
    }
    for (int y = 1; y < h - 1; y++) {
        const uint8_t *s = src + (size_t)y * src_step;
        uint8_t *d       = dst + (size_t)y * dst_step;
        for (int x = 1; x < w - 1; x++) {
            int site = bayer_site(pattern, y, x);
            unsigned c = s[x], v;
            if (site != 1) {
                unsigned diag  = s[x - 1 - src_step] + s[x + 1 - src_step] + s[x - 1 + src_step] + s[x + 1 + src_step];
                unsigned cross = s[x - src_step] + s[x + src_step] + s[x - 1] + s[x + 1];
                v = (diag * coef[2 - site] + cross * 9617u + c * 4u * coef[site] + (1u << 15)) >> 16;
            } else {
                unsigned vert = s[x - src_step] + s[x + src_step];
                unsigned horz = s[x - 1] + s[x + 1];
                int hsite     = bayer_site(pattern, y, x + 1);
                v = (vert * coef[2 - hsite] + horz * coef[hsite] + c * 2u * 9617u + (1u << 14)) >> 15;
            }
            d[x] = (uint8_t)v;
        }
        d[0]     = d[1];
        d[w - 1] = d[w - 2];
    }
    memcpy(dst, dst + dst_step, (size_t)w);
    memcpy(dst + (size_t)(h - 1) * dst_step, dst + (size_t)(h - 2) * dst_step, (size_t)w);
}

/* GpuMat::convertTo(.., 65535./255.) / (.., 255./65535.)  (GPUStereoProcessor.cpp:153-158);
 * float multiply, round-half-even, saturate. */
void gipo_mono8_to_mono16(const uint8_t *src, int src_step, uint16_t *dst, int dst_step_elems, int w, int h)
{
    const float scale = (float)(65535. / 255.);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            long v = lrintf(src[(size_t)y * src_step + x] * scale);
            dst[(size_t)y * dst_step_elems + x] = (uint16_t)(v < 0 ? 0 : v > 65535 ? 65535 : v);
        }
}

void gipo_mono16_to_mono8(const uint16_t *src, int src_step_elems, uint8_t *dst, int dst_step, int w, int h)
{
    const float scale = (float)(255. / 65535.);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            long v = lrintf(src[(size_t)y * src_step_elems + x] * scale);
            dst[(size_t)y * dst_step + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* ======================================================================================
 * Rectification.  Reference: GpuStereoProcessor::rectifyImageLeft/Right
 * (src/GPUStereoProcessor.cpp:252-262) -> image_geometry::PinholeCameraModel::rectifyImage ->
 * cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2) + cv::remap(INTER_LINEAR, BORDER_CONSTANT 0).
 * Pinned by left/right-0022_rect.png (test/UTest.cpp:247-260).
 * ====================================================================================== */

/* 3x3 inverse by LU with partial pivoting (cv::Matx::inv(DECOMP_LU)), double */
static int inv3x3_lu(const double a[9], double out[9])
{
    double m[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            m[i][j]     = a[i * 3 + j];
            m[i][j + 3] = (i == j) ? 1.0 : 0.0;
        }
    for (int i = 0; i < 3; i++) {
        int k = i;
        for (int j = i + 1; j < 3; j++)
            if (fabs(m[j][i]) > fabs(m[k][i])) k = j;
        if (fabs(m[k][i]) < 1e-300) return -1;
        if (k != i)
            for (int j = 0; j < 6; j++) {
                double t = m[i][j];
                m[i][j]  = m[k][j];
                m[k][j]  = t;
            }
        double d = -1.0 / m[i][i];
        for (int j = i + 1; j < 3; j++) {
            double alpha = m[j][i] * d;
            for (int c = i + 1; c < 6; c++) m[j][c] += alpha * m[i][c];
        }
    }
    for (int i = 2; i >= 0; i--)
        for (int c = 3; c < 6; c++) {
            double s = m[i][c];
            for (int k = i + 1; k < 3; k++) s -= m[i][k] * m[k][c];
            m[i][c] = s / m[i][i];
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = m[i][j + 3];
    return 0;
}

void gipo_init_undistort_rectify_map(const double K[9], const double *D, int nD, const double R[9], const double P[12],
                                     int w, int h, int16_t *map1, uint16_t *map2)
{
    double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0;
    double k3 = nD > 4 ? D[4] : 0, k4 = nD > 5 ? D[5] : 0, k5 = nD > 6 ? D[6] : 0, k6 = nD > 7 ? D[7] : 0;
    double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    double PR[9], ir[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[i * 4 + k] * R[k * 3 + j];
            PR[i * 3 + j] = s;
        }
    if (inv3x3_lu(PR, ir) != 0) memset(ir, 0, sizeof ir);

    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double iw = 1. / _w, x = _x * iw, y = _y * iw;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
            double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            double u = fx * xd + u0, v = fy * yd + v0;
            double su = u * 32.0, sv = v * 32.0;
            /* saturate_cast<int>(double) = round half to even, saturating */
            int iu = su >= 2147483647.0 ? INT_MAX : su <= -2147483648.0 ? INT_MIN : (int)lrint(su);
            int iv = sv >= 2147483647.0 ? INT_MAX : sv <= -2147483648.0 ? INT_MIN : (int)lrint(sv);
            size_t o        = (size_t)i * w + j;
            map1[2 * o]     = (int16_t)(iu >> 5);
            map1[2 * o + 1] = (int16_t)(iv >> 5);
            map2[o]         = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

/* cv::remap, CV_16SC2 + CV_16UC1 maps, INTER_LINEAR, BORDER_CONSTANT(0), 8-bit, cn channels.
 * weights (32-ay)(32-ax)*32 ... sum 32768; dst = (sum + 16384) >> 15. */
void gipo_remap_bilinear_u8(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int dst_step, int w, int h,
                            int cn, const int16_t *map1, const uint16_t *map2)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        uint8_t *d = dst + (size_t)y * dst_step;
        for (int x = 0; x < w; x++) {
            size_t o = (size_t)y * w + x;
            int sx = map1[2 * o], sy = map1[2 * o + 1];
            int a  = map2[o] & 1023;
            int ax = a & 31, ay = a >> 5;
            int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32;
            int w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
            int in_x0 = (unsigned)sx < (unsigned)sw, in_x1 = (unsigned)(sx + 1) < (unsigned)sw;
            int in_y0 = (unsigned)sy < (unsigned)sh, in_y1 = (unsigned)(sy + 1) < (unsigned)sh;
            for (int k = 0; k < cn; k++) {
                int p00 = (in_x0 && in_y0) ? src[(size_t)sy * src_step + sx * cn + k] : 0;
                int p01 = (in_x1 && in_y0) ? src[(size_t)sy * src_step + (sx + 1) * cn + k] : 0;
                int p10 = (in_x0 && in_y1) ? src[(size_t)(sy + 1) * src_step + sx * cn + k] : 0;
                int p11 = (in_x1 && in_y1) ? src[(size_t)(sy + 1) * src_step + (sx + 1) * cn + k] : 0;
                int v   = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
                d[x * cn + k] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
    }
}

/* cv::remap INTER_NEAREST with the same fixed-point maps: the 5-bit fractions are rounded half-up (NNDeltaTab_i of imgwarp.cpp) */
void gipo_remap_nearest_u8(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int dst_step, int w, int h, int cn,
                           const int16_t *map1, const uint16_t *map2)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t o = (size_t)y * w + x;
            int a = map2[o] & 1023;
            int sx = map1[2 * o] + ((a & 31) >= 16), sy = map1[2 * o + 1] + ((a >> 5) >= 16);
            int in = (unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh;
            for (int k = 0; k < cn; k++) dst[(size_t)y * dst_step + x * cn + k] = in ? src[(size_t)sy * src_step + sx * cn + k] : 0;
        }
}

/* ======================================================================================
 * StereoBM.  Reference: GpuStereoProcessor::computeDisparity(cv::Mat&, cv::Mat&, cv::Mat&)
 * (src/GPUStereoProcessor.cpp:312-321) -> cv::StereoBM::compute (OpenCV 3.x calib3d/stereobm.cpp)
 * PARITY UNPINNED (no reference test asserts a disparity value).
 * ====================================================================================== */

/* prefilterXSobel: 3x3 Sobel-x, rows processed in pairs => reflect-101 rows, columns 0 and w-1 and
 * (odd h) the last row forced to cap. */
void gipo_prefilter_xsobel(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int cap)
{
    int y;
    for (y = 0; y < h - 1; y += 2) {
        const uint8_t *r1 = src + (size_t)y * src_step;
        const uint8_t *r0 = y > 0 ? r1 - src_step : (h > 1 ? r1 + src_step : r1);
        const uint8_t *r2 = y < h - 1 ? r1 + src_step : (h > 1 ? r1 - src_step : r1);
        const uint8_t *r3 = y < h - 2 ? r1 + 2 * (size_t)src_step : r1;
        uint8_t *d0 = dst + (size_t)y * dst_step, *d1 = d0 + dst_step;
        d0[0] = d0[w - 1] = d1[0] = d1[w - 1] = (uint8_t)cap;
        for (int x = 1; x < w - 1; x++) {
            int e0 = r0[x + 1] - r0[x - 1], e1 = r1[x + 1] - r1[x - 1];
            int e2 = r2[x + 1] - r2[x - 1], e3 = r3[x + 1] - r3[x - 1];
            d0[x] = (uint8_t)(clampi(e0 + 2 * e1 + e2, -cap, cap) + cap);
            d1[x] = (uint8_t)(clampi(e1 + 2 * e2 + e3, -cap, cap) + cap);
        }
    }
    for (; y < h; y++) memset(dst + (size_t)y * dst_step, cap, (size_t)w);
}

/* prefilterNorm: centre-weighted 5-point response minus the winsize^2 box mean, replicate borders */
void gipo_prefilter_norm(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int winsize, int cap)
{
    int wsz2 = winsize / 2;
    int scale_g = winsize * winsize / 8, scale_s = (1024 + scale_g) / (scale_g * 2);
    scale_g *= scale_s;
    int *vbuf = (int *)calloc((size_t)w + 2 * (wsz2 + 1) + 2, sizeof(int));
    int *vsum = vbuf + wsz2 + 1;

    for (int x = 0; x < w; x++) vsum[x] = (uint16_t)(src[x] * (wsz2 + 2));
    for (int y = 1; y < wsz2; y++)
        for (int x = 0; x < w; x++) vsum[x] = (uint16_t)(vsum[x] + src[(size_t)src_step * mini(y, h - 1) + x]);

    for (int y = 0; y < h; y++) {
        const uint8_t *top    = src + (size_t)src_step * maxi(y - wsz2 - 1, 0);
        const uint8_t *bottom = src + (size_t)src_step * mini(y + wsz2, h - 1);
        const uint8_t *prev   = src + (size_t)src_step * maxi(y - 1, 0);
        const uint8_t *curr   = src + (size_t)src_step * y;
        const uint8_t *next   = src + (size_t)src_step * mini(y + 1, h - 1);
        uint8_t *d            = dst + (size_t)y * dst_step;

        for (int x = 0; x < w; x++) vsum[x] = (uint16_t)(vsum[x] + bottom[x] - top[x]);
        for (int x = 0; x <= wsz2; x++) {
            vsum[-x - 1] = vsum[0];
            vsum[w + x]  = vsum[w - 1];
        }
        int sum = vsum[0] * (wsz2 + 1);
        for (int x = 1; x <= wsz2; x++) sum += vsum[x];

        int val = ((curr[0] * 5 + curr[1] + prev[0] + next[0]) * scale_g - sum * scale_s) >> 10;
        d[0]    = (uint8_t)(clampi(val, -cap, cap) + cap);
        int x;
        for (x = 1; x < w - 1; x++) {
            sum += vsum[x + wsz2] - vsum[x - wsz2 - 1];
            val  = ((curr[x] * 4 + curr[x - 1] + curr[x + 1] + prev[x] + next[x]) * scale_g - sum * scale_s) >> 10;
            d[x] = (uint8_t)(clampi(val, -cap, cap) + cap);
        }
        sum += vsum[x + wsz2] - vsum[x - wsz2 - 1];
        val  = ((curr[x] * 5 + curr[x - 1] + prev[x] + next[x]) * scale_g - sum * scale_s) >> 10;
        d[x] = (uint8_t)(clampi(val, -cap, cap) + cap);
    }
    free(vbuf);
}

/* getValidDisparityROI (calib3d/stereosgbm.cpp) for full-image roi1/roi2, intersected with the image.
 * roi = {x, y, width, height}; all zero when empty. */
void gipo_valid_disparity_roi(int w, int h, int min_disparity, int num_disparities, int block_size, int roi[4])
{
    int sw2 = block_size / 2, maxd = min_disparity + num_disparities - 1;
    int xmin = maxi(0, maxd) + sw2, xmax = mini(w, w - min_disparity) - sw2;
    int ymin = sw2, ymax = h - sw2;
    roi[0] = roi[1] = roi[2] = roi[3] = 0;
    if (xmax - xmin <= 0 || ymax - ymin <= 0) return;
    int x0 = maxi(xmin, 0), x1 = mini(xmax, w), y0 = maxi(ymin, 0), y1 = mini(ymax, h);
    if (x1 <= x0 || y1 <= y0) return;
    roi[0] = x0, roi[1] = y0, roi[2] = x1 - x0, roi[3] = y1 - y0;
}

/*
 * findStereoCorrespondenceBM on one stripe of rows [row0,row1) of the valid ROI (all window rows are real
 * image rows because row0 >= wsz2 and row1 <= h - wsz2).
 *
 * Relative column c (output column = lofs + c) maps to image columns
 *     lc(c) = lofs + clamp(c, -lofs, w-1-lofs)            (left)
 *     rc(c) = rofs + clamp(c, -rofs, w-ndisp-rofs) + d    (right, internal index d in [0,ndisp))
 * per-column cost C(c,y,d) = |L[y][lc(c)] - R[y][rc(c)+d]|, SAD = box sum of C over the window; the internal
 * index is reversed: disparity D = ndisp-1-d+mindisp.  First minimum in the d scan wins (=> ties go to the
 * largest disparity).
 */
static void bm_stripe(const uint8_t *left, const uint8_t *right, int step, int w, int h, const gipo_bm_params *p,
                      int row0, int row1, int16_t *disp, int dstep, int32_t *cost, int cstep)
{
    (void)h;
    const int wsz = p->block_size, wsz2 = wsz / 2, ndisp = p->num_disparities, mindisp = p->min_disparity;
    const int lofs = maxi(ndisp - 1 + mindisp, 0), rofs = -mini(ndisp - 1 + mindisp, 0);
    /* OpenCV: width1 = w - rofs - ndisp + 1.  For minDisparity > 0 that runs mindisp columns past the row end (the
     * library writes them into the next row's blanked border); the oracle stops at the last image column -- every
     * in-bounds output is unchanged. */
    const int width1 = mini(w - rofs - ndisp + 1, w - lofs);
    const int ftzero = p->prefilter_cap, tex_thr = p->texture_threshold, uniq = p->uniqueness_ratio;
    const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);
    const int ext0 = row0 - wsz2, next = (row1 - row0) + 2 * wsz2; /* extended rows */

    int *hsad  = (int *)malloc((size_t)next * ndisp * sizeof(int)); /* horizontal window sums per ext row */
    int *htext = (int *)malloc((size_t)next * sizeof(int));
    int *sadb  = (int *)malloc((size_t)(ndisp + 2) * sizeof(int));
    int *sad   = sadb + 1;
    memset(hsad, 0, (size_t)next * ndisp * sizeof(int));
    memset(htext, 0, (size_t)next * sizeof(int));

#define LC(c) (lofs + clampi((c), -lofs, w - 1 - lofs))
#define RC(c) (rofs + clampi((c), -rofs, w - ndisp - rofs))

    /* window for x = 0 minus its right-most column; the x loop adds column x+wsz2 first */
    for (int c = -wsz2; c < wsz2; c++) {
        int lc = LC(c), rc = RC(c);
        for (int ye = 0; ye < next; ye++) {
            const uint8_t *lrow = left + (size_t)(ext0 + ye) * step, *rrow = right + (size_t)(ext0 + ye) * step + rc;
            int lval = lrow[lc];
            int *hs  = hsad + (size_t)ye * ndisp;
            for (int d = 0; d < ndisp; d++) hs[d] += abs(lval - rrow[d]);
            htext[ye] += abs(lval - ftzero);
        }
    }

    for (int x = 0; x < width1; x++) {
        /* slide the horizontal window: add column x+wsz2, drop column x-wsz2-1 (not for x = 0) */
        {
            int ca = x + wsz2, cs = x - wsz2 - 1;
            int lca = LC(ca), rca = RC(ca), lcs = LC(cs), rcs = RC(cs);
            for (int ye = 0; ye < next; ye++) {
                const uint8_t *lrow = left + (size_t)(ext0 + ye) * step, *rrow = right + (size_t)(ext0 + ye) * step;
                int la = lrow[lca];
                int *hs = hsad + (size_t)ye * ndisp;
                const uint8_t *ra = rrow + rca;
                if (x > 0) {
                    int ls = lrow[lcs];
                    const uint8_t *rs = rrow + rcs;
                    for (int d = 0; d < ndisp; d++) hs[d] += abs(la - ra[d]) - abs(ls - rs[d]);
                    htext[ye] += abs(la - ftzero) - abs(ls - ftzero);
                } else {
                    for (int d = 0; d < ndisp; d++) hs[d] += abs(la - ra[d]);
                    htext[ye] += abs(la - ftzero);
                }
            }
        }
        /* vertical window for the first row of the stripe */
        for (int d = 0; d < ndisp; d++) sad[d] = 0;
        int tsum = 0;
        for (int ye = 0; ye < 2 * wsz2; ye++) {
            const int *hs = hsad + (size_t)ye * ndisp;
            for (int d = 0; d < ndisp; d++) sad[d] += hs[d];
            tsum += htext[ye];
        }
        int16_t *dcol = disp + (size_t)row0 * dstep + lofs + x;
        int32_t *ccol = cost ? cost + (size_t)row0 * cstep + lofs + x : NULL;
        for (int y = 0; y < row1 - row0; y++) {
            const int *hadd = hsad + (size_t)(y + 2 * wsz2) * ndisp;
            int minsad = INT_MAX, mind = -1;
            if (y > 0) {
                const int *hsub = hsad + (size_t)(y - 1) * ndisp;
                for (int d = 0; d < ndisp; d++) sad[d] += hadd[d] - hsub[d];
                tsum += htext[y + 2 * wsz2] - htext[y - 1];
            } else {
                for (int d = 0; d < ndisp; d++) sad[d] += hadd[d];
                tsum += htext[y + 2 * wsz2];
            }
            for (int d = 0; d < ndisp; d++)
                if (sad[d] < minsad) minsad = sad[d], mind = d;

            if (tsum < tex_thr) {
                dcol[(size_t)y * dstep] = FILTERED;
                continue;
            }
            if (uniq > 0) {
                int thresh = minsad + (minsad * uniq / 100), d;
                for (d = 0; d < ndisp; d++)
                    if ((d < mind - 1 || d > mind + 1) && sad[d] <= thresh) break;
                if (d < ndisp) {
                    dcol[(size_t)y * dstep] = FILTERED;
                    continue;
                }
            }
            sad[-1]    = sad[1];
            sad[ndisp] = sad[ndisp - 2];
            int pp = sad[mind + 1], nn = sad[mind - 1];
            int den = pp + nn - 2 * sad[mind] + abs(pp - nn);
            int v   = (ndisp - mind - 1 + mindisp) * 256 + (den != 0 ? (pp - nn) * 256 / den : 0) + 15;
            dcol[(size_t)y * dstep] = (int16_t)(v >> 4);
            if (ccol) ccol[(size_t)y * cstep] = sad[mind];
        }
    }
#undef LC
#undef RC
    free(hsad);
    free(htext);
    free(sadb);
}

int gipo_bm_match(const uint8_t *left, const uint8_t *right, int step, int w, int h, const gipo_bm_params *p,
                  int16_t *disp, int dstep, int32_t *cost, int cstep, int nthreads)
{
    const int ndisp = p->num_disparities, mindisp = p->min_disparity, wsz2 = p->block_size / 2;
    const int lofs = maxi(ndisp - 1 + mindisp, 0), rofs = -mini(ndisp - 1 + mindisp, 0);
    const int width1 = mini(w - rofs - ndisp + 1, w - lofs);
    const int16_t FILTERED = (int16_t)((mindisp - 1) * 16);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) disp[(size_t)y * dstep + x] = FILTERED;
    if (cost)
        for (int y = 0; y < h; y++) memset(cost + (size_t)y * cstep, 0, (size_t)w * sizeof(int32_t));
    if (lofs >= w || rofs >= w || width1 < 1) return 0;
    int roi[4];
    gipo_valid_disparity_roi(w, h, mindisp, ndisp, p->block_size, roi);
    if (roi[2] <= 0 || roi[3] <= 0) return 0; /* OpenCV leaves the map unwritten here; the oracle defines FILTERED */
    int r0 = roi[1], r1 = roi[1] + roi[3];
    (void)wsz2;
#ifdef _OPENMP
    int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    int nt = 1;
    (void)nthreads;
#endif
    int nstripes = mini(nt, maxi(1, (r1 - r0) / 8));
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int s = 0; s < nstripes; s++) {
        int a = r0 + (int)((long)(r1 - r0) * s / nstripes), b = r0 + (int)((long)(r1 - r0) * (s + 1) / nstripes);
        if (b > a) bm_stripe(left, right, step, w, h, p, a, b, disp, dstep, cost, cstep);
    }
    return 0;
}

/* cv::validateDisparity (calib3d/stereosgbm.cpp) on rows [row0,row1) */
void gipo_validate_disparity(int16_t *disp, int dstep, const int32_t *cost, int cstep, int w, int h, int row0, int row1,
                             int min_disparity, int num_disparities, int disp12_max_diff)
{
    (void)h;
    const int minD = min_disparity, maxD = min_disparity + num_disparities;
    const int minX1 = maxi(maxD, 0), maxX1 = w + mini(minD, 0);
    const int INVALID = (minD - 1) * 16;
    const int maxdiff = disp12_max_diff * 16;
    int *disp2     = (int *)malloc((size_t)w * 2 * sizeof(int));
    int *disp2cost = disp2 + w;
    for (int y = row0; y < row1; y++) {
        int16_t *dptr       = disp + (size_t)y * dstep;
        const int32_t *cptr = cost + (size_t)y * cstep;
        for (int x = 0; x < w; x++) disp2[x] = INVALID, disp2cost[x] = INT_MAX;
        for (int x = minX1; x < maxX1; x++) {
            int d = dptr[x], c = cptr[x];
            if (d == INVALID) continue;
            int x2 = x - ((d + 8) >> 4);
            if (x2 < 0 || x2 >= w) continue; /* cannot happen for BM output; guards the oracle only */
            if (disp2cost[x2] > c) disp2cost[x2] = c, disp2[x2] = d;
        }
        for (int x = minX1; x < maxX1; x++) {
            int d = dptr[x];
            if (d == INVALID) continue;
            int d0 = d >> 4, d1 = (d + 15) >> 4;
            int x0 = x - d0, x1 = x - d1;
            if ((0 <= x0 && x0 < w && disp2[x0] > INVALID && abs(disp2[x0] - d) > maxdiff) &&
                (0 <= x1 && x1 < w && disp2[x1] > INVALID && abs(disp2[x1] - d) > maxdiff))
                dptr[x] = (int16_t)INVALID;
        }
    }
    free(disp2);
}

int gipo_stereobm_compute(const uint8_t *left, const uint8_t *right, int step, int w, int h, const gipo_bm_params *p,
                          int16_t *disp, int dstep, int nthreads)
{
    /* StereoBMImpl::compute parameter validation (OpenCV throws; the oracle returns < 0) */
    if (p->prefilter_type != GIPO_PREFILTER_NORMALIZED_RESPONSE && p->prefilter_type != GIPO_PREFILTER_XSOBEL) return -1;
    if (p->prefilter_size < 5 || p->prefilter_size > 255 || p->prefilter_size % 2 == 0) return -2;
    if (p->prefilter_cap < 1 || p->prefilter_cap > 63) return -3;
    if (p->block_size < 5 || p->block_size > 255 || p->block_size % 2 == 0 || p->block_size >= mini(w, h)) return -4;
    if (p->num_disparities <= 0 || p->num_disparities % 16 != 0) return -5;
    if (p->texture_threshold < 0) return -6;
    if (p->uniqueness_ratio < 0) return -7;

    uint8_t *lf = (uint8_t *)malloc((size_t)w * h), *rf = (uint8_t *)malloc((size_t)w * h);
    int32_t *cost = p->disp12_max_diff >= 0 ? (int32_t *)malloc((size_t)w * h * sizeof(int32_t)) : NULL;
    const uint8_t *srcs[2] = {left, right};
    uint8_t *dsts[2]       = {lf, rf};
#pragma omp parallel for num_threads(2)
    for (int i = 0; i < 2; i++) {
        if (p->prefilter_type == GIPO_PREFILTER_XSOBEL)
            gipo_prefilter_xsobel(srcs[i], step, dsts[i], w, w, h, p->prefilter_cap);
        else
            gipo_prefilter_norm(srcs[i], step, dsts[i], w, w, h, p->prefilter_size, p->prefilter_cap);
    }
    gipo_bm_match(lf, rf, w, w, h, p, disp, dstep, cost, w, nthreads);

    const int16_t FILTERED = (int16_t)((p->min_disparity - 1) * 16);
    int roi[4];
    gipo_valid_disparity_roi(w, h, p->min_disparity, p->num_disparities, p->block_size, roi);
    if (roi[2] > 0 && roi[3] > 0) {
        if (p->disp12_max_diff >= 0)
            gipo_validate_disparity(disp, dstep, cost, w, w, h, roi[1], roi[1] + roi[3], p->min_disparity,
                                    p->num_disparities, p->disp12_max_diff);
        for (int y = roi[1]; y < roi[1] + roi[3]; y++) {
            int16_t *d = disp + (size_t)y * dstep;
            for (int x = 0; x < roi[0]; x++) d[x] = FILTERED;
            for (int x = roi[0] + roi[2]; x < w; x++) d[x] = FILTERED;
        }
    }
    if (p->speckle_range >= 0 && p->speckle_window_size > 0)
        gipo_filter_speckles(disp, dstep, w, h, FILTERED, p->speckle_window_size, p->speckle_range);
    free(lf);
    free(rf);
    free(cost);
    return 0;
}

void gipo_disparity_to_f32(const int16_t *disp, int dstep, float *dst, int dst_step_elems, int w, int h, double cx_l,
                           double cx_r)
{
    /* Mat::convertTo(CV_32F, alpha, beta) for 16S -> 32F computes in float: src*(float)alpha + (float)beta */
    const float alpha = (float)(1.0 / 16), beta = (float)(-(cx_l - cx_r));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * dst_step_elems + x] = disp[(size_t)y * dstep + x] * alpha + beta;
}

/* ======================================================================================
 * cv::filterSpeckles (calib3d/stereosgbm.cpp), reference call GPUStereoProcessor.cpp:367-385.
 * 4-connected regions of pixels != new_val whose neighbour difference is <= max_diff; regions with
 * <= max_speckle_size pixels are set to new_val.  Result is traversal-order independent.
 * ====================================================================================== */
void gipo_filter_speckles(int16_t *img, int step, int w, int h, int new_val, int max_speckle_size, int max_diff)
{
    size_t n      = (size_t)w * h;
    int *labels   = (int *)calloc(n, sizeof(int));
    int *stack    = (int *)malloc(n * sizeof(int));
    uint8_t *small = (uint8_t *)calloc(n + 1, 1);
    int cur = 0;
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            int16_t *ds = img + (size_t)i * step;
            if (ds[j] == new_val) continue;
            int lab = labels[(size_t)i * w + j];
            if (lab) {
                if (small[lab]) ds[j] = (int16_t)new_val;
                continue;
            }
            cur++;
            int sp = 0, count = 0;
            labels[(size_t)i * w + j] = cur;
            stack[sp++] = i * w + j;
            while (sp > 0) {
                int pidx = stack[--sp];
                int py = pidx / w, px = pidx - py * w;
                count++;
                int dp = img[(size_t)py * step + px];
                if (py < h - 1 && !labels[pidx + w]) {
                    int q = img[(size_t)(py + 1) * step + px];
                    if (q != new_val && abs(dp - q) <= max_diff) labels[pidx + w] = cur, stack[sp++] = pidx + w;
                }
                if (py > 0 && !labels[pidx - w]) {
                    int q = img[(size_t)(py - 1) * step + px];
                    if (q != new_val && abs(dp - q) <= max_diff) labels[pidx - w] = cur, stack[sp++] = pidx - w;
                }
                if (px < w - 1 && !labels[pidx + 1]) {
                    int q = img[(size_t)py * step + px + 1];
                    if (q != new_val && abs(dp - q) <= max_diff) labels[pidx + 1] = cur, stack[sp++] = pidx + 1;
                }
                if (px > 0 && !labels[pidx - 1]) {
                    int q = img[(size_t)py * step + px - 1];
                    if (q != new_val && abs(dp - q) <= max_diff) labels[pidx - 1] = cur, stack[sp++] = pidx - 1;
                }
            }
            if (count <= max_speckle_size) {
                small[cur] = 1;
                ds[j]      = (int16_t)new_val;
            }
        }
    }
    free(labels);
    free(stack);
    free(small);
}

/* ======================================================================================
 * Reprojection + cloud.  Reference: projectDisparityTo3DPoints (GPUStereoProcessor.cpp:332-346) ->
 * image_geometry::StereoCameraModel (updateQ, MISSING_Z = 10000) -> cv::reprojectImageTo3D;
 * GPUSenderPc2::fillInData (GpuSenderPc2.cpp:15-89).
 * ====================================================================================== */
void gipo_stereo_q(const double Pl[12], const double Pr[12], double Q[16])
{
    double fx = Pl[0], fy = Pl[5], cx = Pl[2], cy = Pl[6], cx_r = Pr[2];
    double baseline = -Pr[3] / Pr[0];
    double Tx = -baseline;
    memset(Q, 0, 16 * sizeof(double));
    Q[0 * 4 + 0] = fy * Tx;
    Q[0 * 4 + 3] = -fy * cx * Tx;
    Q[1 * 4 + 1] = fx * Tx;
    Q[1 * 4 + 3] = -fx * cy * Tx;
    Q[2 * 4 + 3] = fx * fy * Tx;
    Q[3 * 4 + 2] = -fy;
    Q[3 * 4 + 3] = fy * (cx - cx_r);
}

void gipo_reproject_to_3d(const float *disp, int dstep, float *xyz, int xstep, int w, int h, const double Q[16],
                          int handle_missing)
{
    const double bigZ = 10000.;
    double min_disp   = 3.402823466e+38; /* FLT_MAX */
    if (handle_missing) {
        min_disp = INFINITY;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                double v = disp[(size_t)y * dstep + x];
                if (v < min_disp) min_disp = v;
            }
    }
    for (int y = 0; y < h; y++) {
        double qx = Q[1] * y + Q[3], qy = Q[5] * y + Q[7], qz = Q[9] * y + Q[11], qw = Q[13] * y + Q[15];
        float *o = xyz + (size_t)y * xstep;
        for (int x = 0; x < w; x++, qx += Q[0], qy += Q[4], qz += Q[8], qw += Q[12]) {
            double d  = disp[(size_t)y * dstep + x];
            double iW = 1. / (qw + Q[14] * d);
            double X = (qx + Q[2] * d) * iW, Y = (qy + Q[6] * d) * iW, Z = (qz + Q[10] * d) * iW;
            if (fabs(d - min_disp) <= 1.1920928955078125e-07) Z = bigZ;
            o[3 * x] = (float)X, o[3 * x + 1] = (float)Y, o[3 * x + 2] = (float)Z;
        }
    }
}

void gipo_pack_pointcloud2(const float *xyz, int xstep, const uint8_t *bgr, int bgr_step, uint8_t *out, int w, int h)
{
    const float nanv = NAN;
    memset(out, 0, (size_t)w * h * 32);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *p = xyz + (size_t)y * xstep + 3 * x;
            uint8_t *rec   = out + ((size_t)y * w + x) * 32;
            float v[3];
            if (p[2] != 10000.f && !isinf(p[2]))
                v[0] = p[0], v[1] = p[1], v[2] = p[2];
            else
                v[0] = v[1] = v[2] = nanv;
            memcpy(rec, v, 12);
            const uint8_t *c = bgr + (size_t)y * bgr_step + 3 * x;
            rec[16] = c[0], rec[17] = c[1], rec[18] = c[2]; /* b, g, r ; rec[19] stays 0 */
        }
}

/* cv::cuda::drawColorDisp (cudastereo util.cu): HSV ramp, S = V = 1.  Input here is the build's int16 x16 map:
 * integer disparity = d16 >> 4, invalid (< 0) -> 0 (the reference GPU matcher's invalid value). */
void gipo_draw_color_disp(const int16_t *disp16, int dstep, uint8_t *bgra, int bgra_step, int w, int h, int ndisp)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int d16 = disp16[(size_t)y * dstep + x];
            int d   = d16 < 0 ? 0 : d16 >> 4;
            unsigned H  = (unsigned)(((ndisp - d) * 240) / ndisp);
            unsigned hi = (H / 60) % 6;
            float f = H / 60.f - (float)(H / 60);
            float p = 0.f, q = 1.f - f, t = 1.f - (1.f - f), V = 1.f;
            float b = 0, g = 0, r = 0;
            switch (hi) {
            case 0: b = p, g = t, r = V; break;
            case 1: b = p, g = V, r = q; break;
            case 2: b = t, g = V, r = p; break;
            case 3: b = V, g = q, r = p; break;
            case 4: b = V, g = p, r = t; break;
            default: b = q, g = p, r = V; break;
            }
            uint8_t *o = bgra + (size_t)y * bgra_step + 4 * x;
            o[0] = (uint8_t)(unsigned)(fmaxf(0.f, fminf(b, 1.f)) * 255.f);
            o[1] = (uint8_t)(unsigned)(fmaxf(0.f, fminf(g, 1.f)) * 255.f);
            o[2] = (uint8_t)(unsigned)(fmaxf(0.f, fminf(r, 1.f)) * 255.f);
            o[3] = 255;
        }
}
