/*
 * gpuimageproc_hip.h -- C-ABI of libgpuimageproc_hip.so, the MI355X (gfx950) implementation of the
 * `gpuimageproc` stereo hot path (debayer -> rectify -> StereoBM -> speckle -> reproject/PointCloud2).
 *
 * Drop-in boundary: the reference sits behind the C++ class gpuimageproc::GpuStereoProcessor
 * (reference include/gpuimageproc/GPUStereoProcessor.h:63-104); every method of that class has one
 * extern "C" entry point here with the same argument meaning, plain pointers and sizes only.  The
 * reference throws / asserts; this ABI returns an int status (0 = GIP_OK) and never throws.
 * A C++ mirror of the class on top of this ABI lives in include/gpuimageproc/GPUStereoProcessor.hpp;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Three layers:
 *   gipk_*  stateless kernel launchers on DEVICE pointers (batched, stream-ordered)   -- used by tests/bench
 *   gip_*   handle API mirroring GpuStereoProcessor (slots, two streams, senders)     -- the drop-in
 *   gip_batch_* device-resident multi-frame pipeline used by the throughput benchmark
 */
#ifndef GPUIMAGEPROC_HIP_H
#define GPUIMAGEPROC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
enum {
    GIP_OK                  = 0,
    GIP_ERR_INVALID_ARG     = 1, /* reference: assert()/cv::Exception */
    GIP_ERR_NOT_INITIALISED = 2, /* reference: assert(model_.initialized()), GPUStereoProcessor.cpp:238,:266,:307 */
    GIP_ERR_HIP             = 3, /* a hip* call failed; see gip_last_error */
    GIP_ERR_UNSUPPORTED     = 4,
    GIP_ERR_NO_DEVICE       = 5,
    GIP_ERR_EMPTY_SLOT      = 6,
    GIP_ERR_IO              = 7
};

/* ------------------------------------------------------------------ buffer slots
 * The integer values of the reference's GpuMatSource enum, verbatim (GPUStereoProcessor.h:21-57). */
enum {
    GIP_SIDE_L = 1 << 0,
    GIP_SIDE_R = 1 << 1,
    GIP_SIDE_MASK = GIP_SIDE_L | GIP_SIDE_R,
    GIP_SRC_RAW = 1 << 2,
    GIP_SRC_MONO = 1 << 3,
    GIP_SRC_COLOR = 1 << 4,
    GIP_SRC_RECT_MONO = 1 << 5,
    GIP_SRC_RECT_COLOR = 1 << 6,
    GIP_SRC_DISPARITY = 1 << 7,     /* int16 x16 fixed point (CPU StereoBM semantics) */
    GIP_SRC_DISPARITY_32F = 1 << 8, /* float = d16/16 - (cx_l - cx_r)  (GPUStereoProcessor.cpp:320) */
    GIP_SRC_DISPARITY_IMG = 1 << 9, /* BGRA8 */
    GIP_SRC_POINTS2 = 1 << 10       /* 32FC3 xyz */
};

/* element types (OpenCV numbering so tests port 1:1): depth | ((cn-1) << 3) */
enum { GIP_8U = 0, GIP_16U = 2, GIP_16S = 3, GIP_32F = 5 };
#define GIP_MAKETYPE(depth, cn) ((depth) | (((cn) - 1) << 3))
#define GIP_8UC1 GIP_MAKETYPE(GIP_8U, 1)
#define GIP_8UC3 GIP_MAKETYPE(GIP_8U, 3)
#define GIP_8UC4 GIP_MAKETYPE(GIP_8U, 4)
#define GIP_16UC1 GIP_MAKETYPE(GIP_16U, 1)
#define GIP_16SC1 GIP_MAKETYPE(GIP_16S, 1)
#define GIP_32FC1 GIP_MAKETYPE(GIP_32F, 1)
#define GIP_32FC3 GIP_MAKETYPE(GIP_32F, 3)

enum { GIP_PREFILTER_NORMALIZED_RESPONSE = 0, GIP_PREFILTER_XSOBEL = 1 }; /* cv::StereoBM::PREFILTER_* */
enum { GIP_INTER_NEAREST = 0, GIP_INTER_LINEAR = 1 };                     /* cv::InterpolationFlags */
enum { GIP_BAYER_RGGB = 0, GIP_BAYER_BGGR = 1, GIP_BAYER_GBRG = 2, GIP_BAYER_GRBG = 3 };

/* ConnectedTopics bit order (reference include/gpuimageproc/ConnectedTopics.h:11-21) */
enum {
    GIP_TOPIC_DEBAYER_MONO_LEFT = 1 << 0,
    GIP_TOPIC_DEBAYER_MONO_RIGHT = 1 << 1,
    GIP_TOPIC_DEBAYER_COLOR_LEFT = 1 << 2,
    GIP_TOPIC_DEBAYER_COLOR_RIGHT = 1 << 3,
    GIP_TOPIC_RECTIFY_MONO_LEFT = 1 << 4,
    GIP_TOPIC_RECTIFY_MONO_RIGHT = 1 << 5,
    GIP_TOPIC_RECTIFY_COLOR_LEFT = 1 << 6,
    GIP_TOPIC_RECTIFY_COLOR_RIGHT = 1 << 7,
    GIP_TOPIC_DISPARITY = 1 << 8,
    GIP_TOPIC_DISPARITY_VIS = 1 << 9,
    GIP_TOPIC_POINTCLOUD = 1 << 10
};

/* ------------------------------------------------------------------ parameter block
 * Union of the reference's setters (GPUStereoProcessor.cpp:389-419, :202-208), cfg/GPU.cfg and the
 * cv::StereoBM parameters its constructor mirrors into the CPU matcher (GPUStereoProcessor.cpp:18-38). */
typedef struct gip_bm_params {
    int32_t prefilter_type;      /* GIP_PREFILTER_* */
    int32_t prefilter_size;      /* odd 5..255 (NORMALIZED_RESPONSE) */
    int32_t prefilter_cap;       /* 1..63 */
    int32_t block_size;          /* odd 5..255 ("correlation_window_size") */
    int32_t min_disparity;       /* "disparity_min" */
    int32_t num_disparities;     /* multiple of 16 ("disparity_range") */
    int32_t texture_threshold;   /* >= 0 */
    int32_t uniqueness_ratio;    /* >= 0, 0 disables */
    int32_t disp12_max_diff;     /* < 0 disables the L/R consistency check */
    int32_t speckle_window_size; /* StereoBM-internal speckle pass (0 = off) */
    int32_t speckle_range;
} gip_bm_params;

/* sensor_msgs/CameraInfo fields the model needs (GPUStereoProcessor.cpp:41-53) */
typedef struct gip_camera_info {
    int32_t width, height;   /* FULL resolution the camera was calibrated at (sensor_msgs/CameraInfo width / height) */
    double K[9];
    double D[8];
    int32_t nD;
    double R[9];
    double P[12];
    /* sensor_msgs/CameraInfo binning and region of interest, applied as image_geometry::PinholeCameraModel::fromCameraInfo /
     * initRectificationMaps do: the images the processor sees are the ROI of the full frame, decimated by the binning factors;
     * K and P are shifted by the ROI offset and scaled by 1 / binning, the rectification tables are the ROI window of the tables of
     * the binned full frame with the ROI origin subtracted.  All zero (the default) = no binning, full-frame ROI. */
    uint32_t binning_x, binning_y;
    uint32_t roi_x_offset, roi_y_offset, roi_height, roi_width;
} gip_camera_info;

/* ------------------------------------------------------------------ message PODs (the senders' payloads) */
typedef struct gip_header {
    uint32_t seq;
    uint32_t stamp_sec, stamp_nsec;
    char frame_id[64];
} gip_header;

/* sensor_msgs/Image as filled by GPUSenderImage::fillInData (GpuSenderImage.cpp:15-26) */
typedef struct gip_image_msg {
    gip_header header;
    uint32_t height, width;
    char encoding[16];
    uint8_t is_bigendian;
    uint32_t step;
    uint8_t *data; /* height*step bytes, owned by the sender until gip_clean_senders */
    size_t data_size;
} gip_image_msg;

/* stereo_msgs/DisparityImage as filled by GPUSenderDisparity::fillInData (GpuSenderDisparity.cpp:18-48) */
typedef struct gip_disparity_msg {
    gip_header header;
    gip_image_msg image; /* 32FC1 */
    float f, T;
    uint32_t valid_x_offset, valid_y_offset, valid_width, valid_height;
    float min_disparity, max_disparity, delta_d;
} gip_disparity_msg;

/* sensor_msgs/PointCloud2 as filled by GPUSenderPc2::fillInData (GpuSenderPc2.cpp:15-72) */
typedef struct gip_pointfield {
    char name[8];
    uint32_t offset;
    uint8_t datatype; /* 7 = FLOAT32 */
    uint32_t count;
} gip_pointfield;
typedef struct gip_pointcloud2_msg {
    gip_header header;
    uint32_t height, width;
    gip_pointfield fields[4];
    uint32_t nfields;
    uint8_t is_bigendian;
    uint32_t point_step, row_step;
    uint8_t *data;
    size_t data_size;
    uint8_t is_dense;
} gip_pointcloud2_msg;

typedef void (*gip_image_cb)(const gip_image_msg *, void *user);
typedef void (*gip_disparity_cb)(const gip_disparity_msg *, void *user);
typedef void (*gip_pointcloud2_cb)(const gip_pointcloud2_msg *, void *user);

typedef struct gip_processor *gip_handle;
typedef void *gip_stream; /* hipStream_t */

/* ================================================================== handle API (mirrors GpuStereoProcessor) */
/* ctor / dtor (GPUStereoProcessor.cpp:12-39).  device < 0 -> current device. */
int gip_create(int device, gip_handle *out);
int gip_destroy(gip_handle h);
const char *gip_last_error(gip_handle h);
/* the HIP device the handle was created on */
int gip_device(gip_handle h);
const char *gip_status_string(int status);
int gip_device_count(void);
/* n (1..16) new non-blocking streams of `device` that make progress SIDE BY SIDE.  The reference creates its two side streams and lets the
 * driver overlap them (GPUStereoProcessor.cpp:190-200); on MI355X the runtime maps the streams of a process onto four hardware queues by a
 * policy of its own, and two streams that land on one queue run strictly one after the other -- the throughput pipeline (three batch
 * streams + the shared matcher stream, gip_batch_desc::matcher_stream) loses 12-25 % when that happens.  This call measures instead of
 * guessing: it keeps a candidate stream only if a one-workgroup kernel on it completes while a chip-filling kernel runs on each stream kept
 * so far (a few milliseconds per candidate, at set-up time; the device should be otherwise idle).  *n_independent (may be NULL) = how many of
 * the returned streams are mutually concurrent; it is < n only when n exceeds the hardware queues the runtime uses (GPU_MAX_HW_QUEUES) --
 * the rest of the array is then filled with ordinary streams.  Destroy with gip_concurrent_streams_destroy (or hipStreamDestroy). */
int gip_concurrent_streams_create(int device, int n, gip_stream *streams, int *n_independent);
int gip_concurrent_streams_destroy(int device, int n, gip_stream *streams);

/* initStereoModel(CameraInfo, CameraInfo) / (left_cal_file, right_cal_file)  (GPUStereoProcessor.cpp:41-61) */
int gip_init_stereo_model(gip_handle h, const gip_camera_info *left, const gip_camera_info *right);
int gip_init_stereo_model_files(gip_handle h, const char *left_cal_file, const char *right_cal_file);
int gip_is_stereo_model_initialised(gip_handle h); /* :63 */
int gip_read_calibration(const char *file, gip_camera_info *out, char *camera_name, size_t name_len);
/* model accessors used by senders/tests */
/* host-only (no device needed): the fixed-point rectification tables of one camera, as initStereoModel builds them */
int gip_build_rectify_maps(const gip_camera_info *info, int16_t *map1, uint16_t *map2);
int gip_model_size(gip_handle h, int *width, int *height);
int gip_model_get(gip_handle h, double *fx, double *fy, double *cx_l, double *cy, double *cx_r, double *baseline, double Q[16]);
/* remap tables as built on the host (double), for broadcasting to other ranks: map1 = w*h*2 int16, map2 = w*h uint16 */
int gip_get_rectify_maps(gip_handle h, int side, int16_t *map1, uint16_t *map2);

/* uploadMat / downloadMat (GPUStereoProcessor.cpp:89-117).  encoding may be NULL/"" */
int gip_upload_mat(gip_handle h, int slot, const void *data, int rows, int cols, size_t step, int type, const char *encoding);
int gip_download_mat(gip_handle h, int slot, void *dst, size_t dst_step);
int gip_mat_info(gip_handle h, int slot, int *rows, int *cols, int *type);
/* device pointer + pitch of a slot (for zero-copy interop); NULL when the slot is empty */
int gip_mat_device_ptr(gip_handle h, int slot, void **dptr, size_t *pitch);

/* convertRawToColor / convertRawToMono / convertColor (GPUStereoProcessor.cpp:65-88, :119-172) */
int gip_convert_raw_to_color(gip_handle h, int side);
int gip_convert_raw_to_mono(gip_handle h, int side);
int gip_convert_color(gip_handle h, int src_slot, int dst_slot, const char *src_encoding, const char *dst_encoding);

/* rectifyImage (GPU, :236-250).  The remap is the CPU oracle's fixed-point form (rectifyImageLeft/Right :252-262) */
int gip_rectify_image(gip_handle h, int src_slot, int dst_slot, int interpolation);

/* computeDisparity(GpuMatSource x3) (:264-303): writes dst (int16 x16) and the matching DISPARITY_32F slot */
int gip_compute_disparity(gip_handle h, int left_slot, int right_slot, int disparity_slot);
/* filterSpeckles(GpuMatSource) (:356-365): on device, asynchronous; no-op when max_speckle_size <= 0 */
int gip_filter_speckles(gip_handle h, int disparity_slot);
/* computeDisparityImage (:323-330) */
int gip_compute_disparity_image(gip_handle h, int disparity_slot, int image_slot);
/* projectDisparityTo3DPoints (:332-346) */
int gip_project_disparity_to_3d_points(gip_handle h, int disparity_slot, int points_slot);

/* enqueueSendImage / enqueueSendDisparity / enqueueSendPoints (:210-234) + GPUSenderIfc::enqueueSend
 * (GpuSenderIfc.cpp:13-26): async D2H on the slot's stream, then the callback runs on a host thread
 * (hipLaunchHostFunc).  cb may be NULL (the reference tests pass a NULL publisher).  sender_id (optional out)
 * identifies the sender for gip_sender_was_sent / gip_sender_*_msg until gip_clean_senders. */
int gip_enqueue_send_image(gip_handle h, int slot, const gip_header *header, const char *encoding, gip_image_cb cb, void *user, int *sender_id);
int gip_enqueue_send_disparity(gip_handle h, int slot, const gip_header *header, gip_disparity_cb cb, void *user, int *sender_id);
int gip_enqueue_send_points(gip_handle h, int points_slot, int color_slot, const gip_header *header, gip_pointcloud2_cb cb, void *user, int *sender_id);
/* valid_window, f, T, min / max disparity, delta_d of a stereo_msgs/DisparityImage for a width x height map of this processor -- what
 * GPUSenderDisparity::fillInData derives (GpuSenderDisparity.cpp:29-43); header and image fields are left untouched.  For callers that
 * receive the float map through gip_streamer_* instead of a sender. */
int gip_disparity_info(gip_handle h, int width, int height, gip_disparity_msg *out);
int gip_sender_was_sent(gip_handle h, int sender_id); /* GPUSenderIfc::wasDataSent */
const gip_image_msg *gip_sender_image_msg(gip_handle h, int sender_id);
const gip_disparity_msg *gip_sender_disparity_msg(gip_handle h, int sender_id);
const gip_pointcloud2_msg *gip_sender_pointcloud2_msg(gip_handle h, int sender_id);
int gip_clean_senders(gip_handle h); /* :387 */

/* waitForStream / waitForAllStreams (:348-354) */
int gip_wait_for_stream(gip_handle h, int slot_or_side);
int gip_wait_for_all_streams(gip_handle h);

/* setters (:389-419, :202-208); the extra four expose cv::StereoBM state the reference sets in its ctor */
int gip_set_pre_filter_type(gip_handle h, int filter_type);
int gip_set_refine_disparity(gip_handle h, int ref_disp); /* accepted, no effect on CPU-StereoBM semantics (cuda-only knob) */
int gip_set_block_size(gip_handle h, int block_size);
int gip_set_num_disparities(gip_handle h, int num_disp);
int gip_set_min_disparity(gip_handle h, int min_disp);
int gip_set_texture_threshold(gip_handle h, int threshold);
int gip_set_max_speckle_size(gip_handle h, int max_speckle_size);
int gip_get_max_speckle_size(gip_handle h);
int gip_set_max_speckle_diff(gip_handle h, double max_speckle_diff);
double gip_get_max_speckle_diff(gip_handle h);
int gip_set_pre_filter_size(gip_handle h, int v);
int gip_set_pre_filter_cap(gip_handle h, int v);
int gip_set_uniqueness_ratio(gip_handle h, int v);
int gip_set_disp12_max_diff(gip_handle h, int v);
int gip_get_params(gip_handle h, gip_bm_params *out);
int gip_set_params(gip_handle h, const gip_bm_params *p);

/* The per-frame schedule of StereoProcessor::imageCb (reference src/StereoProcessor.cpp:157-298) for the
 * connected-topic mask: upload both raws, run only the stages the mask needs, leave results in the slots.
 * Synchronous like the reference (ends with waitForAllStreams). */
int gip_process_pair(gip_handle h, const void *left_raw, const void *right_raw, int rows, int cols, size_t step, int type,
                     const char *encoding, uint32_t connected_topics);

/* ================================================================== ROS1 wire format of the published messages
 * Body of sensor_msgs/Image, stereo_msgs/DisparityImage, sensor_msgs/PointCloud2 exactly as roscpp serialises what the reference's
 * senders publish (src/GpuSenderImage.cpp, src/GpuSenderDisparity.cpp, src/GpuSenderPc2.cpp): little-endian, fields in declaration
 * order, uint32 length prefixes on strings and arrays.  Host-only, no device needed.  `out` may be NULL (size query).  Returns the
 * number of bytes of the message; a NEGATIVE value -n means `cap` was too small for the n bytes needed; 0 = invalid message. */
int64_t gip_ros1_serialize_image(const gip_image_msg *msg, uint8_t *out, size_t cap);
int64_t gip_ros1_serialize_disparity(const gip_disparity_msg *msg, uint8_t *out, size_t cap);
int64_t gip_ros1_serialize_pointcloud2(const gip_pointcloud2_msg *msg, uint8_t *out, size_t cap);

/* ================================================================== streaming pipeline (gip_streamer_*)
 * The host link of the nodelet: the reference uploads both raws per frame (src/StereoProcessor.cpp:179-180), runs the chain,
 * hands every output to a stream host callback (src/GpuSenderIfc.cpp:13-26) and then BLOCKS on waitForAllStreams()
 * (src/StereoProcessor.cpp:284), i.e. one frame in flight.  The streamer keeps `depth` batches in flight instead: a ring of
 * pinned input / output buffers, hipMemcpyAsync H2D on a copy-in stream, the chain (gip_batch_run) on `compute_lanes` compute streams whose
 * SAD matchers share one stream, D2H on a copy-out stream, delivery through hipLaunchHostFunc on the copy-out stream.  Upload of frame i+1,
 * compute of frames i and i-1 and download of frame i-2 overlap; gip_streamer_submit blocks only when every ring slot is still in flight
 * (back-pressure).
 * The caller submits pair by pair and never chooses a batch size (batching 0, the default): a pair joins the batch being filled, and that
 * batch is launched as soon as a compute lane is free and no earlier batch is still on its way back to the host (or it is full) -- batches of
 * one pair and minimal latency for a camera, full batches for a caller that saturates the host link (the copy-out is the slowest stage then, and
 * cutting batches short would only add launches).
 * The processor handle must have its model and parameters set; it must not be used for anything else, nor have its parameters changed,
 * while a streamer is open (the other compute lanes run on clones of it made at open). */
typedef struct gip_streamer_s *gip_streamer;
typedef struct gip_streamer_config {
    int32_t width, height;
    int32_t encoding;        /* as gip_batch_desc::encoding */
    uint32_t outputs;        /* GIP_TOPIC_DISPARITY (float disparity, 4 B/px) and/or GIP_TOPIC_POINTCLOUD (32 B/px) delivered */
    int32_t want_disparity16;/* also deliver the int16 x16 map (2 B/px) */
    int32_t depth;           /* ring slots = batches in flight (>= 2; 0 -> compute_lanes + 2) */
    int32_t frames_per_batch;/* CAPACITY of a ring slot = the most pairs one launch set processes (0 -> automatic: as many as keep a slot's
                              * largest buffer near 256 MB, at most 64).  With batching 1 it is the fixed batch size (0 -> 1) */
    int32_t inputs_pinned;   /* 1: the raw images passed to gip_streamer_submit live in pinned (hipHostMalloc / hipHostRegister) memory with
                              * tight rows and stay valid until their pair has been delivered: the H2D copy then reads them in place and the
                              * submitting thread copies nothing */
    int32_t batching;        /* 0 (default): adaptive -- a partly filled batch is launched whenever fewer than compute_lanes batches are
                              * computing and the copy-out stream is idle; 1: fixed -- only full batches are launched (a partial one by
                              * gip_streamer_flush) */
    int32_t compute_lanes;   /* batches computing concurrently, each on its own stream and processor handle (0 -> 2; 1..4) */
} gip_streamer_config;
typedef struct gip_streamer_result {
    uint64_t seq;                 /* submission index of this pair (0, 1, ...) */
    gip_header header;            /* as submitted */
    int32_t width, height;
    const int16_t *disparity16;   /* pinned host memory, valid until the callback returns; NULL when not requested */
    const float *disparity_f32;
    const uint8_t *pointcloud;    /* height * width * 32 bytes, PointCloud2 body */
} gip_streamer_result;
/* runs on a HIP runtime thread, once per pair, in submission order */
typedef void (*gip_streamer_cb)(const gip_streamer_result *result, void *user);

int gip_streamer_open(gip_handle h, const gip_streamer_config *cfg, gip_streamer_cb cb, void *user, gip_streamer *out);
/* copies both raw images (rows `step` bytes apart; 0 = tight) into the ring and returns.  After a failed launch (a HIP error, or a
 * parameter block gip_batch_run rejects) the streamer is in a STICKY error state: everything is drained, the pairs that were queued are
 * lost, and submit / flush return that error from then on (close and reopen). */
int gip_streamer_submit(gip_streamer s, const void *left_raw, const void *right_raw, size_t step, const gip_header *header);
/* launch a partially filled batch and wait until every submitted pair has been delivered */
int gip_streamer_flush(gip_streamer s);
/* pairs delivered so far (callback returned) */
uint64_t gip_streamer_delivered(gip_streamer s);
/* launch sets issued so far (pairs submitted / launches = the average batch the adaptive policy formed) */
uint64_t gip_streamer_launches(gip_streamer s);
int gip_streamer_close(gip_streamer s);
const char *gip_streamer_last_error(gip_streamer s);

/* ================================================================== stateless device kernels (gipk_*)
 * All pointers are DEVICE pointers; pitches in BYTES unless suffixed _elems; `batch` images are laid out
 * `*_bstride` BYTES apart; stream is a hipStream_t (NULL = default stream).  Launch errors are returned. */
int gipk_cvt_bgr_to_gray(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int src_is_rgb,
                         int batch, size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_gray_to_bgr(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int batch,
                         size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_swap_rb(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int batch,
                     size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_bayer_to_bgr(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int pattern,
                          int batch, size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_bayer_to_gray(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int pattern,
                           int batch, size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_mono8_to_mono16(const uint8_t *src, size_t src_pitch, uint16_t *dst, size_t dst_pitch, int w, int h, int batch,
                             size_t src_bstride, size_t dst_bstride, gip_stream stream);
int gipk_cvt_mono16_to_mono8(const uint16_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int batch,
                             size_t src_bstride, size_t dst_bstride, gip_stream stream);

/* fixed-point bilinear remap (maps shared by all images of the batch); cn = 1 or 3 */
int gipk_remap_u8(const uint8_t *src, size_t src_pitch, int sw, int sh, uint8_t *dst, size_t dst_pitch, int w, int h, int cn,
                  const int16_t *map1, const uint16_t *map2, int batch, size_t src_bstride, size_t dst_bstride,
                  gip_stream stream);

/* same with cv::InterpolationFlags: GIP_INTER_LINEAR or GIP_INTER_NEAREST (fractions rounded half-up, as cv::remap does) */
int gipk_remap_u8_interp(const uint8_t *src, size_t src_pitch, int sw, int sh, uint8_t *dst, size_t dst_pitch, int w, int h, int cn,
                         const int16_t *map1, const uint16_t *map2, int interpolation, int batch, size_t src_bstride,
                         size_t dst_bstride, gip_stream stream);

int gipk_prefilter(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, int w, int h, int prefilter_type,
                   int prefilter_size, int prefilter_cap, int batch, size_t src_bstride, size_t dst_bstride,
                   gip_stream stream);

/* Fused front end of the matcher: rectify (INTER_LINEAR remap, as gipk_remap_u8) and StereoBM's XSOBEL prefilter in ONE pass, raw
 * mono image -> prefiltered plane; `rectified` (may be NULL) also receives the rectified image.  Replaces rectifyImage followed by
 * the prefilter inside cv::StereoBM::compute (reference src/GPUStereoProcessor.cpp:236-250, :312-321); bit-exact with
 * gipk_remap_u8 + gipk_prefilter(XSOBEL). */
int gipk_rectify_xsobel(const uint8_t *src, size_t src_pitch, int sw, int sh, uint8_t *prefiltered, size_t pf_pitch,
                        uint8_t *rectified, size_t rect_pitch, int w, int h, const int16_t *map1, const uint16_t *map2,
                        int prefilter_cap, int batch, size_t src_bstride, size_t pf_bstride, size_t rect_bstride,
                        gip_stream stream);

/* Block matching on PREFILTERED images: SAD + WTA + texture + uniqueness + sub-pixel -> int16 x16 and (optional) the
 * min-SAD plane, i.e. the state findStereoCorrespondenceBM leaves before validateDisparity / ROI blanking. */
int gipk_bm_match(const uint8_t *left, const uint8_t *right, size_t pitch, int w, int h, const gip_bm_params *p, int16_t *disp,
                  size_t disp_pitch, int32_t *cost, size_t cost_pitch, int batch, size_t img_bstride, size_t disp_bstride,
                  size_t cost_bstride, gip_stream stream);
/* validateDisparity (if disp12_max_diff >= 0) + blanking outside the valid ROI */
int gipk_bm_finalize(int16_t *disp, size_t disp_pitch, const int32_t *cost, size_t cost_pitch, int w, int h,
                     const gip_bm_params *p, int batch, size_t disp_bstride, size_t cost_bstride, gip_stream stream);
/* cv::StereoBM::compute: prefilter + match + finalize (+ StereoBM-internal speckle).  scratch: gipk_stereobm_scratch_bytes */
size_t gipk_stereobm_scratch_bytes(int w, int h, int batch, const gip_bm_params *p);
int gipk_stereobm(const uint8_t *left, const uint8_t *right, size_t pitch, int w, int h, const gip_bm_params *p, int16_t *disp,
                  size_t disp_pitch, void *scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride,
                  gip_stream stream);

int gipk_disparity_to_f32(const int16_t *disp, size_t disp_pitch, float *dst, size_t dst_pitch, int w, int h, double cx_l,
                          double cx_r, int batch, size_t disp_bstride, size_t dst_bstride, gip_stream stream);

/* speckle filter on int16 maps; scratch >= gipk_speckle_scratch_bytes(w,h,batch) */
size_t gipk_speckle_scratch_bytes(int w, int h, int batch);
int gipk_filter_speckles(int16_t *disp, size_t disp_pitch, int w, int h, int new_val, int max_speckle_size, int max_diff,
                         void *scratch, size_t scratch_bytes, int batch, size_t disp_bstride, gip_stream stream);
/* The two phases of the filter, for a frame that is split into row tiles over several GPUs (BASELINE configs[4]): label = find the
 * components of this tile and their sizes; row_labels = per pixel of one row the id of its component's root (-1: invalid pixel) and
 * the size stored there; set_sizes = overwrite the sizes of given roots (with the size of the whole component once the ranks have
 * merged the components that touch the tile borders); apply = erase components of at most max_speckle_size pixels.
 * gipk_filter_speckles = label + apply. */
int gipk_speckle_label(const int16_t *disp, size_t disp_pitch, int w, int h, int new_val, int max_diff, void *scratch, size_t scratch_bytes,
                       int batch, size_t disp_bstride, gip_stream stream);
int gipk_speckle_row_labels(void *scratch, size_t scratch_bytes, int w, int h, int batch, int row, int32_t *root_out, int32_t *size_out,
                            gip_stream stream);
int gipk_speckle_set_sizes(void *scratch, size_t scratch_bytes, int w, int h, int batch, const int32_t *roots, const int32_t *sizes, int n,
                           gip_stream stream);
int gipk_speckle_apply(int16_t *disp, size_t disp_pitch, int w, int h, int new_val, int max_speckle_size, void *scratch,
                       size_t scratch_bytes, int batch, size_t disp_bstride, gip_stream stream);

/* Device-side merge of the components that cross tile borders (row-tile split; no host round trip):
 *   border_record  after gipk_speckle_label: tile `tile`'s record of its first and last row -- per border pixel the disparity, a union-find
 *                  parent (the component's canonical border pixel of this tile, as an index into the buffer that holds the records of all
 *                  tiles in tile order) and the tile-local component size: gipk_speckle_border_record_bytes(w) = 20 bytes per column,
 *                  rounded up to 256 -- and border_roots[2 w], the tile-local root of every border pixel;
 *   border_merge   `records` = the records of ALL n_tiles tiles, contiguous in tile order, on this device (gathered by peer copies or an
 *                  all-gather; MODIFIED in place: they are the union-find forest); unites the components that touch across the borders
 *                  and overwrites the sizes of tile `tile`'s own border components with the size of the whole component.
 * gipk_speckle_apply then erases exactly what cv::filterSpeckles erases on the full frame. */
size_t gipk_speckle_border_record_bytes(int w);
int gipk_speckle_border_record(const int16_t *disp, size_t disp_pitch, int w, int h, int tile, void *scratch, size_t scratch_bytes, void *record,
                               int32_t *border_roots, gip_stream stream);
/* the records of all tiles (tile_records[t]: device pointers; peer memory is read in place when peer access is enabled) copied side by
 * side into `records` in ONE launch -- the alternative to n_tiles hipMemcpyPeerAsync calls */
int gipk_speckle_border_gather(const void *const *tile_records, int n_tiles, int w, void *records, gip_stream stream);
int gipk_speckle_border_merge(void *records, int n_tiles, int tile, int w, int h, int new_val, int max_diff, void *scratch,
                              size_t scratch_bytes, const int32_t *border_roots, gip_stream stream);

/* reprojectImageTo3D on the float disparity (missing = the image minimum, i.e. `missing_value`) */
int gipk_reproject_to_3d(const float *disp, size_t disp_pitch, float *xyz, size_t xyz_pitch, int w, int h, const double Q[16],
                         float missing_value, int handle_missing, int batch, size_t disp_bstride, size_t xyz_bstride,
                         gip_stream stream);
/* PointCloud2 body from xyz + BGR */
int gipk_pack_pointcloud2(const float *xyz, size_t xyz_pitch, const uint8_t *bgr, size_t bgr_pitch, uint8_t *out, int w, int h,
                          int batch, size_t xyz_bstride, size_t bgr_bstride, size_t out_bstride, gip_stream stream);
/* fused: int16 disparity (+ BGR) -> 32-byte PointCloud2 records, no xyz round trip */
int gipk_disparity_to_pointcloud2(const int16_t *disp, size_t disp_pitch, const uint8_t *bgr, size_t bgr_pitch, uint8_t *out,
                                  int w, int h, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch,
                                  size_t disp_bstride, size_t bgr_bstride, size_t out_bstride, gip_stream stream);
/* fused tail of the chain: int16 disparity -> float image (f32_out, contiguous, may be NULL) and/or PointCloud2 records
 * (pc_out, contiguous batch*h*w*32 bytes, may be NULL); color_cn = 1 (mono, replicated like GRAY2BGR) or 3 (BGR) */
int gipk_disparity_outputs(const int16_t *disp, size_t disp_pitch, const uint8_t *color, size_t color_pitch, int color_cn, float *f32_out,
                           uint8_t *pc_out, int w, int h, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch,
                           size_t disp_bstride, size_t color_bstride, gip_stream stream);
/* the same for rows [y_origin, y_origin + h) of a larger frame (a row tile): only the row term of the reprojection sees y_origin */
int gipk_disparity_outputs_rows(const int16_t *disp, size_t disp_pitch, const uint8_t *color, size_t color_pitch, int color_cn, float *f32_out,
                                uint8_t *pc_out, int w, int h, int y_origin, const double Q[16], double cx_l, double cx_r, int min_disparity,
                                int batch, size_t disp_bstride, size_t color_bstride, gip_stream stream);
int gipk_draw_color_disp(const int16_t *disp, size_t disp_pitch, uint8_t *bgra, size_t bgra_pitch, int w, int h, int ndisp,
                         int batch, size_t disp_bstride, size_t bgra_bstride, gip_stream stream);

/* ================================================================== device-resident batch pipeline (benchmark path)
 * One "step" = the whole chain over `batch` stereo pairs already resident in HBM:
 *   raw (mono8 / bayer / bgr8) -> [debayer] -> rectify L,R (+ colour L) -> StereoBM -> speckle -> PointCloud2 (+f32 disparity)
 * Remap tables come from the handle's model or are installed from device memory (received by RCCL broadcast). */
typedef struct gip_batch_desc {
    int32_t width, height, batch;
    int32_t encoding;       /* 0 mono8, 1 bgr8, 2 rgb8, 3.. = 3 + GIP_BAYER_* */
    uint32_t outputs;       /* GIP_TOPIC_* mask: DISPARITY and/or POINTCLOUD */
    const uint8_t *left_raw, *right_raw; /* device, batch images, pitch = width*channels, contiguous */
    int16_t *disparity;     /* device out: batch*h*w int16 (always written) */
    float *disparity_f32;   /* device out or NULL */
    uint8_t *pointcloud;    /* device out or NULL: batch*h*w*32 bytes */
    gip_stream matcher_stream; /* NULL, or a stream the dominant kernel (the SAD matcher) is launched on instead of `stream`,
                                * fenced by events on both sides.  Pipelines that keep several batches in flight pass ONE
                                * shared stream here: the matchers then run back to back instead of time-sharing, which is what
                                * keeps the batches from falling into lock step */
    int32_t fused_front_end;   /* 1: rectify + XSOBEL prefilter run as ONE pass raw -> prefiltered plane (gipk_rectify_xsobel) where the
                                * configuration allows (XSOBEL, distorted camera); 0 (default): remap, then prefilter.  Bit-identical
                                * results; the two-pass form is the faster one on MI355X, so it is the default */
    int32_t batches_in_flight; /* 0 / 1: this batch has the device to itself.  > 1: the caller keeps that many batches in flight on streams of
                                * its own (throughput pipelines): the matcher then leaves a quarter of every SIMD's registers to the small
                                * kernels of the other batches instead of taking every workgroup slot it can get -- faster alone, slower as
                                * a pipeline otherwise.  Setting matcher_stream implies > 1 */
    int32_t phase;             /* GIP_BATCH_WHOLE (0, default): the whole chain.  GIP_BATCH_FRONT: colour, rectify, prefilter and the matcher
                                * only -- with a matcher_stream the matcher is launched there and `stream` does NOT wait for it.
                                * GIP_BATCH_BACK (same handle, same descriptor and same `stream`, after its FRONT): wait for that matcher, then L/R check,
                                * speckle filter and the outputs.  A host that enqueues FRONT(k+1) before BACK(k) software-pipelines the
                                * batches on TWO streams in all: the matchers back to back on matcher_stream, everything else in order on
                                * `stream` next to them (handles and buffers: three sets in rotation).  With a matcher_stream the BACK phase may
                                * also run on a stream of its OWN (all FRONT phases in order on one stream, all BACK phases on another, the
                                * matchers on the third): the handle orders its own buffers by events.  See DESIGN.md section 4 */
} gip_batch_desc;
enum { GIP_BATCH_WHOLE = 0, GIP_BATCH_FRONT = 1, GIP_BATCH_BACK = 2 };
int gip_batch_set_maps_device(gip_handle h, int width, int height, const int16_t *map1_l, const uint16_t *map2_l,
                              const int16_t *map1_r, const uint16_t *map2_r);
int gip_batch_run(gip_handle h, const gip_batch_desc *desc, gip_stream stream);
/* time (ms) the dominant kernel (SAD block matcher) took in the last gip_batch_run, from hipEvents on its stream;
 * valid after the stream has been synchronised */
int gip_batch_last_kernel_ms(gip_handle h, float *bm_ms, float *total_ms);
/* every gip_batch_run since the last reset records its own event pairs; read returns the sums (ms) and the run count.
 * The stream(s) used must have been synchronised before reading. */
int gip_batch_timing_reset(gip_handle h);
int gip_batch_timing_read(gip_handle h, double *bm_ms_sum, double *total_ms_sum, int *runs);
/* the same plus the "remap + SAD" span of every run: from the start of the chain to the end of validateDisparity (colour
 * conversion, rectify, prefilter, matcher, L/R check) -- the stages whose compulsory HBM traffic is the 16 bytes per pixel of
 * SURVEY.md section 8(d) */
int gip_batch_timing_read_spans(gip_handle h, double *bm_ms_sum, double *total_ms_sum, double *remap_sad_ms_sum, int *runs);

/* ================================================================== row-tile split of ONE large frame (gip_rowsplit_*)
 * BASELINE configs[4]: a 4096x2160 frame cut into N row tiles, one per GPU.  The reference launches every stage once per frame on one
 * GPU and then waits (reference src/GPUStereoProcessor.cpp:283 block_matcher_gpu_->compute, src/StereoProcessor.cpp:284
 * waitForAllStreams); here ONE process owns N processor handles -- one per tile, each on its own device (tiles share devices when there
 * are fewer devices than tiles) -- and every buffer of the steady state is allocated at open:
 *   tile t owns output rows [y0, y1) and works on the extended rows [e0, e1) = its own rows + the halo (block radius + 1 for XSOBEL;
 *   + the prefilter radius for NORMALIZED_RESPONSE) of the RECTIFIED images;
 *   front   rectify the tile's rows of both eyes (the raw frame is resident on every tile's device) into ONE buffer whose rows interleave
 *           the two eyes, so that the halo rows of both eyes are one contiguous block;
 *   halo    one hipMemcpyPeerAsync per neighbour pulls that block (over xGMI between devices; a device-to-device copy when the tiles
 *           share a device) -- or, halo_mode 1, every tile rectifies its own halo rows and nothing is exchanged;
 *   match   prefilter + SAD matcher + L/R check on the extended tile (row-local: the tile's own rows equal the full frame's bit for bit);
 *   speckle every tile labels its own rows, publishes its border record (gipk_speckle_border_record), pulls the records of the other
 *           tiles and merges on the device (gipk_speckle_border_merge); then erases small components and writes float disparity and
 *           PointCloud2 records for its rows.
 * One host thread per tile enqueues that tile's work (host_threads); results stay on the tiles' devices (gip_rowsplit_outputs_device) or
 * are stitched into host memory (gip_rowsplit_download).  No CPU fallback, no host round trip in the steady state. */
typedef struct gip_rowsplit_s *gip_rowsplit;
typedef struct gip_rowsplit_config {
    int32_t width, height;
    int32_t encoding;        /* as gip_batch_desc::encoding */
    uint32_t outputs;        /* GIP_TOPIC_DISPARITY (float disparity) and/or GIP_TOPIC_POINTCLOUD; the int16 map is always produced */
    int32_t n_tiles;         /* 1..16 */
    int32_t devices[16];     /* HIP device of tile t; a negative entry = t mod gip_device_count() */
    int32_t halo_mode;       /* 0: neighbours exchange rectified halo rows (hipMemcpyPeerAsync); 1: every tile rectifies its halo rows itself */
    int32_t host_threads;    /* 0: one enqueueing host thread per tile when n_tiles > 1; 1: the calling thread enqueues everything */
    int32_t gather_mode;     /* how a tile collects the other tiles' border records: 0 = auto (one gather kernel when all tiles share its
                              * device, else one hipMemcpyPeerAsync per tile), 1 = always copies, 2 = always the gather kernel (reads peer
                              * memory in place; needs peer access between the devices) */
} gip_rowsplit_config;
/* per-tile GPU time of the last run (ms, hipEvents on the tile's stream): front = colour + rectify, halo = waiting for and pulling the
 * neighbours' rows, match = prefilter + matcher + L/R check + speckle labelling + border record, gather = waiting for and pulling the
 * other tiles' records, post = merge + erase + float disparity + cloud */
typedef struct gip_rowsplit_timing {
    float front_ms, halo_ms, match_ms, gather_ms, post_ms, total_ms;
} gip_rowsplit_timing;

/* the plan alone (host only, no device needed): own rows [y0, y1), extended rows [e0, e1) of tile `tile` and the halo row count for a
 * frame of `height` rows cut into n_tiles tiles with these matcher parameters; INVALID_ARG when the tiles would be thinner than halo + 2 */
int gip_rowsplit_plan_rows(int height, int n_tiles, const gip_bm_params *params, int tile, int *y0, int *y1, int *e0, int *e1, int *halo);
int gip_rowsplit_open(const gip_camera_info *left, const gip_camera_info *right, const gip_bm_params *params, int max_speckle_size,
                      double max_speckle_diff, const gip_rowsplit_config *cfg, gip_rowsplit *out);
int gip_rowsplit_close(gip_rowsplit rs);
const char *gip_rowsplit_last_error(gip_rowsplit rs);
/* the plan: own rows [y0, y1), extended rows [e0, e1), device of a tile */
int gip_rowsplit_plan(gip_rowsplit rs, int tile, int *y0, int *y1, int *e0, int *e1, int *device);
/* the raw frame as resident on tile `tile`'s device (full frame, tight rows); must stay valid until the run that uses it has finished */
int gip_rowsplit_set_inputs_device(gip_rowsplit rs, int tile, const uint8_t *left_raw, const uint8_t *right_raw);
/* host frame -> every tile's device (pinned staging inside; asynchronous on the tiles' streams); step 0 = tight rows */
int gip_rowsplit_upload(gip_rowsplit rs, const void *left_raw, const void *right_raw, size_t step);
/* enqueue one frame on every tile's stream (returns when everything is enqueued) / wait for it */
int gip_rowsplit_run(gip_rowsplit rs);
int gip_rowsplit_wait(gip_rowsplit rs);
/* one frame with the tiles run ONE AFTER THE OTHER, each with its device to itself (what a rank on its own GPU sees); synchronous.
 * Fills timing[n_tiles] (may be NULL); the frame's results are the same as gip_rowsplit_run's. */
int gip_rowsplit_run_serial(gip_rowsplit rs, gip_rowsplit_timing *timing);
int gip_rowsplit_last_timing(gip_rowsplit rs, int tile, gip_rowsplit_timing *out);
/* results of the tile's own rows on its device: int16 x16 map (pitch = width), float image and 32-byte records (NULL when not requested) */
int gip_rowsplit_outputs_device(gip_rowsplit rs, int tile, const int16_t **disparity, const float **disparity_f32, const uint8_t **pointcloud);
/* stitched frame into host memory (any of the three may be NULL); synchronous */
int gip_rowsplit_download(gip_rowsplit rs, int16_t *disparity, float *disparity_f32, uint8_t *pointcloud);

#ifdef __cplusplus
}
#endif
#endif /* GPUIMAGEPROC_HIP_H */
