/*
 * gip_oracle.h -- CPU restatement (plain C99) of the reference's stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path (libgpuimageproc_hip.so)
 * never links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - rectification  : PINNED by the reference's own goldens (left/right-0022_rect.png,
 *                      asserted bit-exact in test/UTest.cpp:247-260)
 *   - colour -> mono : PINNED by aloe-L.png -> aloe-L-Mono.png and the 1x1 cases
 *                      (test/UTest.cpp:190-245)
 *   - StereoBM / validateDisparity / filterSpeckles / reprojectImageTo3D / Bayer / PC2:
 *                      PARITY UNPINNED -- the arithmetic lives in OpenCV 3.x (un-vendored,
 *                      version unpinned, CMakeLists.txt:21-24) and a forked image_geometry,
 *                      neither present in the container, and no reference test asserts a
 *                      disparity value (test/UTest.cpp:290-398).  These functions restate the
 *                      published OpenCV 3.4 algorithms as called from
 *                      src/GPUStereoProcessor.cpp:312-321 (CPU StereoBM), :367-385
 *                      (filterSpeckles), :332-346 (reproject) and src/GpuSenderPc2.cpp:15-89.
 */
#ifndef GIP_ORACLE_H
#define GIP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- colour conversion (reference: GPUStereoProcessor.cpp:119-172 -> cv::cvtColor) ---- */
/* Bayer patterns are named by the ROS encoding (cv_bridge maps bayer_rggb8 -> BayerBG etc.) */
enum { GIPO_BAYER_RGGB = 0, GIPO_BAYER_BGGR = 1, GIPO_BAYER_GBRG = 2, GIPO_BAYER_GRBG = 3 };

void gipo_bgr_to_gray(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int src_is_rgb);
void gipo_gray_to_bgr(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h);
void gipo_swap_rb(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h);
void gipo_bayer_to_bgr(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int pattern);
void gipo_bayer_to_gray(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int pattern);
void gipo_mono8_to_mono16(const uint8_t *src, int src_step, uint16_t *dst, int dst_step_elems, int w, int h);
void gipo_mono16_to_mono8(const uint16_t *src, int src_step_elems, uint8_t *dst, int dst_step, int w, int h);

/* ---- rectification (reference: GPUStereoProcessor.cpp:252-262 -> image_geometry rectifyImage) ---- */
/* map1: w*h*2 int16 (x,y integer parts); map2: w*h uint16 ((fy<<5)|fx, 5 fractional bits each) */
void gipo_init_undistort_rectify_map(const double K[9], const double *D, int nD, const double R[9], const double P[12],
                                     int w, int h, int16_t *map1, uint16_t *map2);
void gipo_remap_bilinear_u8(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int dst_step, int w, int h,
                            int cn, const int16_t *map1, const uint16_t *map2);

void gipo_remap_nearest_u8(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int dst_step, int w, int h, int cn,
                           const int16_t *map1, const uint16_t *map2);

/* ---- StereoBM (reference: GPUStereoProcessor.cpp:312-321 -> cv::StereoBM::compute) ---- */
enum { GIPO_PREFILTER_NORMALIZED_RESPONSE = 0, GIPO_PREFILTER_XSOBEL = 1 };

typedef struct gipo_bm_params {
    int prefilter_type;    /* 0 NORMALIZED_RESPONSE, 1 XSOBEL */
    int prefilter_size;    /* odd, 5..255 (NORMALIZED_RESPONSE only) */
    int prefilter_cap;     /* 1..63 */
    int block_size;        /* odd, 5..255 */
    int min_disparity;
    int num_disparities;   /* > 0, multiple of 16 */
    int texture_threshold; /* >= 0 */
    int uniqueness_ratio;  /* >= 0 */
    int disp12_max_diff;   /* < 0 disables the L/R check */
    int speckle_window_size;
    int speckle_range;
} gipo_bm_params;

void gipo_prefilter_xsobel(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int cap);
void gipo_prefilter_norm(const uint8_t *src, int src_step, uint8_t *dst, int dst_step, int w, int h, int winsize, int cap);

/* Block matching on PREFILTERED images.  disp: int16 x16 fixed point; cost (may be NULL): int32 min SAD.
 * Produces the pre-validate, pre-ROI-blank map exactly as findStereoCorrespondenceBM leaves it for the rows
 * of the valid ROI; rows outside are FILTERED.  nthreads <= 0 -> all cores. */
int gipo_bm_match(const uint8_t *left, const uint8_t *right, int step, int w, int h, const gipo_bm_params *p,
                  int16_t *disp, int disp_step_elems, int32_t *cost, int cost_step_elems, int nthreads);
void gipo_validate_disparity(int16_t *disp, int disp_step_elems, const int32_t *cost, int cost_step_elems, int w, int h,
                             int row0, int row1, int min_disparity, int num_disparities, int disp12_max_diff);
void gipo_valid_disparity_roi(int w, int h, int min_disparity, int num_disparities, int block_size, int roi[4]);
/* Full StereoBM::compute: prefilter + match + validate + ROI blanking + (optional) speckle.  Returns 0 on success,
 * negative on parameter-validation failure (OpenCV would throw). */
int gipo_stereobm_compute(const uint8_t *left, const uint8_t *right, int step, int w, int h, const gipo_bm_params *p,
                          int16_t *disp, int disp_step_elems, int nthreads);
/* disparity16 -> float: d/16 - (cx_l - cx_r)   (GPUStereoProcessor.cpp:320) */
void gipo_disparity_to_f32(const int16_t *disp, int disp_step_elems, float *dst, int dst_step_elems, int w, int h,
                           double cx_l, double cx_r);

/* ---- speckle filter (reference: GPUStereoProcessor.cpp:367-385 -> cv::filterSpeckles) ---- */
void gipo_filter_speckles(int16_t *img, int step_elems, int w, int h, int new_val, int max_speckle_size, int max_diff);

/* ---- reprojection and cloud (GPUStereoProcessor.cpp:332-346, GpuSenderPc2.cpp:15-89) ---- */
/* Q as built by image_geometry::StereoCameraModel::updateQ */
void gipo_stereo_q(const double P_left[12], const double P_right[12], double Q[16]);
/* cv::reprojectImageTo3D on a float disparity image, handleMissingValues semantic included */
void gipo_reproject_to_3d(const float *disp, int disp_step_elems, float *xyz, int xyz_step_elems, int w, int h,
                          const double Q[16], int handle_missing);
/* PointCloud2 body: 32-byte records (x,y,z f32 @0/4/8, rgb @16 as b,g,r,0), NaN for Z==10000 or inf */
void gipo_pack_pointcloud2(const float *xyz, int xyz_step_elems, const uint8_t *bgr, int bgr_step, uint8_t *out, int w,
                           int h);
/* cv::cuda::drawColorDisp on integer disparity d (0..ndisp) -> BGRA */
void gipo_draw_color_disp(const int16_t *disp16, int step_elems, uint8_t *bgra, int bgra_step, int w, int h, int ndisp);

int gipo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
