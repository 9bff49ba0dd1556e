// gip_internal.h -- shared helpers for the HIP implementation (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "gpuimageproc_hip.h"

#define GIP_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) { gip_note_hip_error(_e, #expr, __FILE__, __LINE__); return GIP_ERR_HIP; } \
    } while (0)

// Records the last HIP failure (thread-local text returned through gip_last_error).
void gip_note_hip_error(hipError_t e, const char* what, const char* file, int line);
const char* gip_last_hip_error_text();

// Kernel launches cannot fail asynchronously in a way we can see here, but a bad configuration does.
#define GIP_LAUNCH_CHECK() GIP_HIP_CHECK(hipGetLastError())

static inline int gip_div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t gip_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Derived StereoBM geometry shared by host launchers and kernels (cv::StereoBM, stereobm.cpp).
struct GipBmGeom {
    int ndisp, mindisp, wsz, wsz2;
    int lofs, rofs, width1;   // first output column, right-image offset, number of output columns
    int roi_x, roi_y, roi_w, roi_h;  // getValidDisparityROI & image
    int filtered;             // (mindisp - 1) * 16
    bool degenerate;          // whole map FILTERED
};

static inline GipBmGeom gip_bm_geom(int w, int h, const gip_bm_params* p)
{
    GipBmGeom g;
    g.ndisp = p->num_disparities; g.mindisp = p->min_disparity; g.wsz = p->block_size; g.wsz2 = g.wsz / 2;
    int t = g.ndisp - 1 + g.mindisp;
    g.lofs = t > 0 ? t : 0;
    g.rofs = t < 0 ? -t : 0;
    g.width1 = w - g.rofs - g.ndisp + 1;
    // OpenCV's formula runs mindisp columns past the row end when minDisparity > 0; stop at the last image column
    if (g.width1 > w - g.lofs) g.width1 = w - g.lofs;
    g.filtered = (g.mindisp - 1) * 16;
    int maxd = g.mindisp + g.ndisp - 1;
    int xmin = (maxd > 0 ? maxd : 0) + g.wsz2;
    int xmax = (w < w - g.mindisp ? w : w - g.mindisp) - g.wsz2;
    int ymin = g.wsz2, ymax = h - g.wsz2;
    g.roi_x = g.roi_y = g.roi_w = g.roi_h = 0;
    if (xmax - xmin > 0 && ymax - ymin > 0) {
        int x0 = xmin > 0 ? xmin : 0, x1 = xmax < w ? xmax : w, y0 = ymin > 0 ? ymin : 0, y1 = ymax < h ? ymax : h;
        if (x1 > x0 && y1 > y0) { g.roi_x = x0; g.roi_y = y0; g.roi_w = x1 - x0; g.roi_h = y1 - y0; }
    }
    g.degenerate = (g.lofs >= w || g.rofs >= w || g.width1 < 1 || g.roi_w <= 0 || g.roi_h <= 0);
    return g;
}

int gip_validate_bm_params(const gip_bm_params* p, int w, int h);
struct BmLaunchCtx;   // bm_common.h
int gip_bm_match_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                      int32_t* cost, size_t cost_pitch, int batch, size_t img_bstride, size_t disp_bstride, size_t cost_bstride, gip_stream stream,
                      bool skip_fill, const BmLaunchCtx* ctx, int* cost_u16);
// gipk_bm_finalize on a uint16 min-SAD plane (cost_pitch / cost_bstride in bytes)
int gip_bm_finalize_impl(int16_t* disp, size_t disp_pitch, const void* cost, size_t cost_pitch, int w, int h, const gip_bm_params* p, int batch, size_t disp_bstride,
                         size_t cost_bstride, gip_stream stream, bool cost_u16);
// gipk_stereobm with a launch context for the matcher (events around it, optional stream of its own).  left == right == nullptr
// means the caller has already written the PREFILTERED planes into the scratch (gip_stereobm_planes tells where): the fused
// rectify + XSOBEL front end of gip_batch_run does that.
int gip_stereobm_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                      void* scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride, gip_stream stream, const BmLaunchCtx* ctx);
int gip_stereobm_phase_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                            void* scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride, gip_stream stream, const BmLaunchCtx* ctx,
                            int phase, int* cost_u16_io);
// prefiltered planes inside a gipk_stereobm scratch buffer: pitch, per-image plane size, and the two base pointers
void gip_stereobm_planes(void* scratch, int w, int h, int batch, uint8_t** left_f, uint8_t** right_f, size_t* pitch, size_t* plane);

// Fused tail of the chain (float disparity + PointCloud2 records) with an optional table of everything that depends on the disparity value alone
// (k_post.hip: gip_w_table_build; (ndisp + 1) * 16 + 1 entries of 16 bytes from FILTERED upwards).  Bit-identical with and without the table.
size_t gip_w_table_bytes(int ndisp);
int gip_w_table_build(void* table, const double Q[16], double cx_l, double cx_r, int min_disparity, int ndisp, gip_stream stream);
int gip_disparity_outputs_impl(const int16_t* disp, size_t disp_pitch, const uint8_t* color, size_t color_pitch, int color_cn, float* f32_out, uint8_t* pc_out, int w, int h,
                               int y_origin, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch, size_t disp_bstride, size_t color_bstride,
                               gip_stream stream, const void* w_table, int table_ndisp, bool own_disparity = false);
// own_disparity: every value of `disp` was produced by this library's matcher (+ speckle filter) with the parameters the table was built for,
// i.e. lies inside the table -- the table-only kernel may run (k_disp_outputs_tab)
