// GPUStereoProcessor.hpp -- C++ mirror of the reference's gpuimageproc::GpuStereoProcessor
// (reference include/gpuimageproc/GPUStereoProcessor.h:63-104) on top of the C-ABI of libgpuimageproc_hip.so.
//
// Same method names and argument meaning; cv::Mat -> gpuimageproc::Mat, ros::Publisher* -> a std::function callback
// (nullptr allowed, as in the reference's tests), failed preconditions throw std::runtime_error where the reference
// asserts or lets cv::Exception escape.  Header-only; link with -lgpuimageproc_hip.
#pragma once
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gpuimageproc/Mat.hpp"
#include "gpuimageproc_hip.h"

namespace gpuimageproc
{

// GpuMatSource: the reference's enum values verbatim (GPUStereoProcessor.h:21-57)
enum GpuMatSource {
    GPU_MAT_SIDE_L = GIP_SIDE_L,
    GPU_MAT_SIDE_R = GIP_SIDE_R,
    GPU_MAT_SIDE_MASK = GIP_SIDE_MASK,
    GPU_MAT_SRC_RAW = GIP_SRC_RAW,
    GPU_MAT_SRC_MONO = GIP_SRC_MONO,
    GPU_MAT_SRC_COLOR = GIP_SRC_COLOR,
    GPU_MAT_SRC_RECT_MONO = GIP_SRC_RECT_MONO,
    GPU_MAT_SRC_RECT_COLOR = GIP_SRC_RECT_COLOR,
    GPU_MAT_SRC_DISPARITY = GIP_SRC_DISPARITY,
    GPU_MAT_SRC_DISPARITY_32F = GIP_SRC_DISPARITY_32F,
    GPU_MAT_SRC_DISPARITY_IMG = GIP_SRC_DISPARITY_IMG,
    GPU_MAT_SRC_POINTS2 = GIP_SRC_POINTS2,
    GPU_MAT_SRC_L_RAW = GPU_MAT_SRC_RAW | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_RAW = GPU_MAT_SRC_RAW | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_MONO = GPU_MAT_SRC_MONO | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_MONO = GPU_MAT_SRC_MONO | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_COLOR = GPU_MAT_SRC_COLOR | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_COLOR = GPU_MAT_SRC_COLOR | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_RECT_MONO = GPU_MAT_SRC_RECT_MONO | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_RECT_MONO = GPU_MAT_SRC_RECT_MONO | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_RECT_COLOR = GPU_MAT_SRC_RECT_COLOR | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_RECT_COLOR = GPU_MAT_SRC_RECT_COLOR | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_DISPARITY = GPU_MAT_SRC_DISPARITY | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_DISPARITY = GPU_MAT_SRC_DISPARITY | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_DISPARITY_32F = GPU_MAT_SRC_DISPARITY_32F | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_DISPARITY_32F = GPU_MAT_SRC_DISPARITY_32F | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_DISPARITY_IMG = GPU_MAT_SRC_DISPARITY_IMG | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_DISPARITY_IMG = GPU_MAT_SRC_DISPARITY_IMG | GPU_MAT_SIDE_R,
    GPU_MAT_SRC_L_POINTS2 = GPU_MAT_SRC_POINTS2 | GPU_MAT_SIDE_L,
    GPU_MAT_SRC_R_POINTS2 = GPU_MAT_SRC_POINTS2 | GPU_MAT_SIDE_R
};
inline GpuMatSource operator&(GpuMatSource a, GpuMatSource b) { return static_cast<GpuMatSource>(static_cast<int>(a) & static_cast<int>(b)); }
inline GpuMatSource operator|(GpuMatSource a, GpuMatSource b) { return static_cast<GpuMatSource>(static_cast<int>(a) | static_cast<int>(b)); }

class GpuStereoProcessor;

// GPUSenderIfc / GPUSenderImage / GPUSenderDisparity / GPUSenderPc2 (reference GpuSender*.h): a handle on one queued
// asynchronous send; the message stays valid until GpuStereoProcessor::cleanSenders().
class GPUSenderIfc
{
  public:
    GPUSenderIfc(gip_handle h, int id) : h_(h), id_(id) {}
    virtual ~GPUSenderIfc() {}
    bool wasDataSent() const { return gip_sender_was_sent(h_, id_) != 0; }

  protected:
    gip_handle h_;
    int id_;
};
class GPUSenderImage : public GPUSenderIfc
{
  public:
    using GPUSenderIfc::GPUSenderIfc;
    const gip_image_msg *getImageMessage() const { return gip_sender_image_msg(h_, id_); }
};
class GPUSenderDisparity : public GPUSenderIfc
{
  public:
    using GPUSenderIfc::GPUSenderIfc;
    const gip_disparity_msg *getDisparityMessage() const { return gip_sender_disparity_msg(h_, id_); }
};
class GPUSenderPc2 : public GPUSenderIfc
{
  public:
    using GPUSenderIfc::GPUSenderIfc;
    const gip_pointcloud2_msg *getPointMessage() const { return gip_sender_pointcloud2_msg(h_, id_); }
};
typedef std::shared_ptr<GPUSenderIfc> GPUSenderIfcPtr;
typedef std::shared_ptr<GPUSenderImage> GPUSenderImagePtr;
typedef std::shared_ptr<GPUSenderDisparity> GPUSenderDisparityPtr;
typedef std::shared_ptr<GPUSenderPc2> GPUSenderPc2Ptr;

typedef std::function<void(const gip_image_msg &)> ImagePublisher;
typedef std::function<void(const gip_disparity_msg &)> DisparityPublisher;
typedef std::function<void(const gip_pointcloud2_msg &)> PointCloud2Publisher;

class GpuStereoProcessor
{
  public:
    explicit GpuStereoProcessor(int device = -1)
    {
        int rc = gip_create(device, &h_);
        if (rc != GIP_OK) throw std::runtime_error(std::string("GpuStereoProcessor: ") + gip_status_string(rc));
    }
    ~GpuStereoProcessor()
    {
        if (h_) gip_destroy(h_);
    }
    GpuStereoProcessor(const GpuStereoProcessor &) = delete;
    GpuStereoProcessor &operator=(const GpuStereoProcessor &) = delete;
    gip_handle handle() const { return h_; }

    void initStereoModel(const gip_camera_info &l_info_msg, const gip_camera_info &r_info_msg) { ck(gip_init_stereo_model(h_, &l_info_msg, &r_info_msg)); }
    void initStereoModel(const std::string &left_cal_file, const std::string &right_cal_file) { ck(gip_init_stereo_model_files(h_, left_cal_file.c_str(), right_cal_file.c_str())); }
    bool isStereoModelInitialised() { return gip_is_stereo_model_initialised(h_) != 0; }
    void convertRawToColor(GpuMatSource side) { ck(gip_convert_raw_to_color(h_, side)); }
    void convertRawToMono(GpuMatSource side) { ck(gip_convert_raw_to_mono(h_, side)); }
    void uploadMat(GpuMatSource mat_source, const Mat &cv_mat, std::string encoding = "")
    {
        ck(gip_upload_mat(h_, mat_source, cv_mat.data, cv_mat.rows, cv_mat.cols, cv_mat.step, cv_mat.type(), encoding.c_str()));
    }
    void downloadMat(GpuMatSource mat_source, Mat &cv_mat)
    {
        int rows, cols, type;
        ck(gip_mat_info(h_, mat_source, &rows, &cols, &type));
        cv_mat.create(rows, cols, type);
        ck(gip_download_mat(h_, mat_source, cv_mat.data, cv_mat.step));
    }
    void convertColor(GpuMatSource mat_source, GpuMatSource mat_dst, const std::string &src_encoding, const std::string &dst_encoding)
    {
        ck(gip_convert_color(h_, mat_source, mat_dst, src_encoding.c_str(), dst_encoding.c_str()));
    }
    GPUSenderImagePtr enqueueSendImage(GpuMatSource source, const gip_header &imagePattern, std::string encoding, const ImagePublisher *pub)
    {
        int id = -1;
        ck(gip_enqueue_send_image(h_, source, &imagePattern, encoding.c_str(), pub ? &GpuStereoProcessor::image_tramp : nullptr, (void *)pub, &id));
        return std::make_shared<GPUSenderImage>(h_, id);
    }
    GPUSenderDisparityPtr enqueueSendDisparity(GpuMatSource source, const gip_header &imagePattern, const DisparityPublisher *pub)
    {
        int id = -1;
        ck(gip_enqueue_send_disparity(h_, source, &imagePattern, pub ? &GpuStereoProcessor::disparity_tramp : nullptr, (void *)pub, &id));
        return std::make_shared<GPUSenderDisparity>(h_, id);
    }
    // valid window, f, T, disparity range and delta_d of a DisparityImage of this processor (for maps that arrive through the streaming API)
    gip_disparity_msg disparityInfo(int width, int height)
    {
        gip_disparity_msg m{};
        ck(gip_disparity_info(h_, width, height, &m));
        return m;
    }
    GPUSenderPc2Ptr enqueueSendPoints(GpuMatSource points_source, GpuMatSource color_source, const gip_header &imagePattern, const PointCloud2Publisher *pub)
    {
        int id = -1;
        ck(gip_enqueue_send_points(h_, points_source, color_source, &imagePattern, pub ? &GpuStereoProcessor::pc2_tramp : nullptr, (void *)pub, &id));
        return std::make_shared<GPUSenderPc2>(h_, id);
    }
    void rectifyImage(GpuMatSource source, GpuMatSource dest, int interpolation = GIP_INTER_LINEAR) { ck(gip_rectify_image(h_, source, dest, interpolation)); }
    void computeDisparity(GpuMatSource left, GpuMatSource right, GpuMatSource disparity) { ck(gip_compute_disparity(h_, left, right, disparity)); }
    void computeDisparityImage(GpuMatSource disparity_src, GpuMatSource disp_image_dest) { ck(gip_compute_disparity_image(h_, disparity_src, disp_image_dest)); }
    void projectDisparityTo3DPoints(GpuMatSource disparity_src, GpuMatSource points_src) { ck(gip_project_disparity_to_3d_points(h_, disparity_src, points_src)); }
    void waitForStream(GpuMatSource stream_source) { ck(gip_wait_for_stream(h_, stream_source)); }
    void waitForAllStreams() { ck(gip_wait_for_all_streams(h_)); }
    void cleanSenders() { ck(gip_clean_senders(h_)); }
    void setPreFilterType(int filter_type) { ck(gip_set_pre_filter_type(h_, filter_type)); }
    void setRefineDisparity(bool ref_disp) { ck(gip_set_refine_disparity(h_, ref_disp)); }
    void setBlockSize(int block_size) { ck(gip_set_block_size(h_, block_size)); }
    void setNumDisparities(int numDisp) { ck(gip_set_num_disparities(h_, numDisp)); }
    void setMinDisparity(int minDisp) { ck(gip_set_min_disparity(h_, minDisp)); }
    void setTextureThreshold(int threshold) { ck(gip_set_texture_threshold(h_, threshold)); }
    void filterSpeckles(GpuMatSource disparity_src) { ck(gip_filter_speckles(h_, disparity_src)); }
    int getMaxSpeckleSize() const { return gip_get_max_speckle_size(h_); }
    void setMaxSpeckleSize(int maxSpeckleSize) { ck(gip_set_max_speckle_size(h_, maxSpeckleSize)); }
    double getMaxSpeckleDiff() const { return gip_get_max_speckle_diff(h_); }
    void setMaxSpeckleDiff(double maxSpeckleDiff) { ck(gip_set_max_speckle_diff(h_, maxSpeckleDiff)); }
    // cv::StereoBM state the reference fixes in its constructor (GPUStereoProcessor.cpp:18-38), exposed here
    void setPreFilterSize(int v) { ck(gip_set_pre_filter_size(h_, v)); }
    void setPreFilterCap(int v) { ck(gip_set_pre_filter_cap(h_, v)); }
    void setUniquenessRatio(int v) { ck(gip_set_uniqueness_ratio(h_, v)); }
    void setDisp12MaxDiff(int v) { ck(gip_set_disp12_max_diff(h_, v)); }
    // the whole StereoBM parameter block at once (validated like cv::StereoBM::compute does)
    void setParams(const gip_bm_params &p) { ck(gip_set_params(h_, &p)); }
    gip_bm_params getParams() const { gip_bm_params p{}; gip_get_params(h_, &p); return p; }
    // the imageCb schedule (StereoProcessor.cpp:157-298) in one call
    void processPair(const Mat &l_raw, const Mat &r_raw, const std::string &encoding, uint32_t connected_topics)
    {
        ck(gip_process_pair(h_, l_raw.data, r_raw.data, l_raw.rows, l_raw.cols, l_raw.step, l_raw.type(), encoding.c_str(), connected_topics));
    }

  private:
    void ck(int rc)
    {
        if (rc != GIP_OK) throw std::runtime_error(std::string(gip_status_string(rc)) + ": " + gip_last_error(h_));
    }
    static void image_tramp(const gip_image_msg *m, void *u) { (*(const ImagePublisher *)u)(*m); }
    static void disparity_tramp(const gip_disparity_msg *m, void *u) { (*(const DisparityPublisher *)u)(*m); }
    static void pc2_tramp(const gip_pointcloud2_msg *m, void *u) { (*(const PointCloud2Publisher *)u)(*m); }
    gip_handle h_ = nullptr;
};

} // namespace gpuimageproc
