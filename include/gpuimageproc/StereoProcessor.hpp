// StereoProcessor.hpp -- ROS-free counterpart of the reference's orchestration class gpuimageproc::StereoProcessor
// (reference src/StereoProcessor.cpp): the per-frame stage schedule of imageCb (:157-298) driven by the connected-topic
// mask, and the dynamic_reconfigure mapping of configCb (:307-322) for the parameters of cfg/GPU.cfg.
//
// ROS itself (roscpp, message_filters, image_transport, dynamic_reconfigure, pluginlib) is absent from the build image,
// so publishers are std::function callbacks and messages are the POD structs of gpuimageproc_hip.h.  A catkin shim that
// forwards sensor_msgs to imageCb() is all a nodelet needs.
#pragma once
#include <algorithm>
#include <memory>
#include <string>

#include "gpuimageproc/ConnectedTopics.hpp"
#include "gpuimageproc/GPUStereoProcessor.hpp"

namespace gpuimageproc
{

// cfg/GPU.cfg:12-35 -- names, defaults and ranges of the reconfigure parameters
struct GPUConfig {
    bool xsobel = false;                    // :12
    bool refine_disparity = false;          // :13
    int correlation_window_size = 15;       // :16  [5,255]
    int disparity_min = 0;                  // :17  [-128,128]  (declared but never read by the reference)
    int disparity_range = 128;              // :18  [16,128] in the reference; lifted to 256 here (BASELINE configs need it)
    bool bilateral_filter = false;          // :21  (dead code in the reference: StereoProcessor.cpp:324-335)
    int filter_ndisp = 64, filter_radius = 3, filter_iters = 1;                                      // :22-24 unused
    double filter_edge_threshold = 0.1, filter_max_disc_threshold = 0.2, filter_sigma_range = 10;    // :25-27 unused
    double texture_threshold = 10;          // :31  [0,10000]
    int max_speckle_size = 800;             // :34  [0,10000]
    double max_speckle_diff = 5;            // :35  [0,50]
};

class StereoProcessor
{
  public:
    explicit StereoProcessor(int device = -1) : stereoProcessor_(new GpuStereoProcessor(device)) {}

    GpuStereoProcessor &processor() { return *stereoProcessor_; }
    ConnectedTopics &connected() { return connected_; }

    // publishers (nullptr = nobody subscribed to that topic's callback)
    ImagePublisher pub_mono_left_, pub_mono_right_, pub_color_left_, pub_color_right_;
    ImagePublisher pub_mono_rect_left_, pub_mono_rect_right_, pub_color_rect_left_, pub_color_rect_right_;
    ImagePublisher pub_disparity_vis_;
    DisparityPublisher pub_disparity_;
    PointCloud2Publisher pub_pointcloud_;

    // configCb (StereoProcessor.cpp:307-322)
    void configCb(GPUConfig &config)
    {
        config.correlation_window_size |= 0x1;                       // must be odd
        config.disparity_range = (config.disparity_range / 16) * 16; // must be a multiple of 16
        config.correlation_window_size = std::min(255, std::max(5, config.correlation_window_size));
        config.disparity_range = std::max(16, config.disparity_range);
        stereoProcessor_->setPreFilterType(config.xsobel ? GIP_PREFILTER_XSOBEL : GIP_PREFILTER_NORMALIZED_RESPONSE);
        stereoProcessor_->setRefineDisparity(config.refine_disparity);
        stereoProcessor_->setBlockSize(config.correlation_window_size);
        stereoProcessor_->setNumDisparities(config.disparity_range);
        // The reference calls setMinDisparity(config.disparity_range) here (:317): a no-op for its CUDA matcher, but it would
        // put the CPU-semantics matcher at minDisparity = 128.  The declared-but-unused disparity_min parameter is what the
        // call evidently meant.
        stereoProcessor_->setMinDisparity(config.disparity_min);
        stereoProcessor_->setTextureThreshold((int)config.texture_threshold);
        stereoProcessor_->setMaxSpeckleDiff(config.max_speckle_diff);
        stereoProcessor_->setMaxSpeckleSize(config.max_speckle_size);
    }

    // imageAndInfoCb (:144-155): first frame initialises the model from the CameraInfo pair
    void imageAndInfoCb(const Mat &l_raw, const Mat &r_raw, const std::string &encoding, const gip_header &header, const gip_camera_info &l_info,
                        const gip_camera_info &r_info)
    {
        if (!stereoProcessor_->isStereoModelInitialised()) stereoProcessor_->initStereoModel(l_info, r_info);
        imageCb(l_raw, r_raw, encoding, header);
    }

    // imageCb (:157-298)
    void imageCb(const Mat &l_raw, const Mat &r_raw, const std::string &encoding, const gip_header &header)
    {
        GpuStereoProcessor &p = *stereoProcessor_;
        const ConnectedTopics &c = connected_;
        p.cleanSenders();                                                                                      // :166
        p.uploadMat(GPU_MAT_SRC_L_RAW, l_raw, encoding);                                                      // :179
        p.uploadMat(GPU_MAT_SRC_R_RAW, r_raw, encoding);                                                      // :180
        const bool disp_any = c.Disparity() || c.DisparityVis() || c.Pointcloud();
        if (c.DebayerMonoLeft() || c.RectifyMonoLeft() || disp_any) p.convertRawToMono(GPU_MAT_SIDE_L);       // :183-186
        if (c.DebayerMonoLeft()) p.enqueueSendImage(GPU_MAT_SRC_L_MONO, header, "mono8", pub(pub_mono_left_));
        if (c.DebayerMonoRight() || c.RectifyMonoRight() || disp_any) p.convertRawToMono(GPU_MAT_SIDE_R);     // :192-195
        if (c.DebayerMonoRight()) p.enqueueSendImage(GPU_MAT_SRC_R_MONO, header, "mono8", pub(pub_mono_right_));
        if (c.DebayerColorLeft() || c.RectifyColorLeft() || c.Pointcloud()) p.convertRawToColor(GPU_MAT_SIDE_L);   // :201-204
        if (c.DebayerColorLeft()) p.enqueueSendImage(GPU_MAT_SRC_L_COLOR, header, "bgr8", pub(pub_color_left_));
        if (c.DebayerColorRight() || c.RectifyColorRight()) p.convertRawToColor(GPU_MAT_SIDE_R);              // :210-213
        if (c.DebayerColorRight()) p.enqueueSendImage(GPU_MAT_SRC_R_COLOR, header, "bgr8", pub(pub_color_right_));
        if (c.RectifyMonoLeft() || disp_any) p.rectifyImage(GPU_MAT_SRC_L_MONO, GPU_MAT_SRC_L_RECT_MONO, GIP_INTER_LINEAR);    // :220-223
        if (c.RectifyMonoRight() || disp_any) p.rectifyImage(GPU_MAT_SRC_R_MONO, GPU_MAT_SRC_R_RECT_MONO, GIP_INTER_LINEAR);   // :224-227
        if (c.RectifyMonoLeft()) p.enqueueSendImage(GPU_MAT_SRC_L_RECT_MONO, header, "mono8", pub(pub_mono_rect_left_));
        if (c.RectifyMonoRight()) p.enqueueSendImage(GPU_MAT_SRC_R_RECT_MONO, header, "mono8", pub(pub_mono_rect_right_));
        if (c.RectifyColorLeft() || c.Pointcloud()) p.rectifyImage(GPU_MAT_SRC_L_COLOR, GPU_MAT_SRC_L_RECT_COLOR, GIP_INTER_LINEAR);  // :239-242
        if (c.RectifyColorRight()) p.rectifyImage(GPU_MAT_SRC_R_COLOR, GPU_MAT_SRC_R_RECT_COLOR, GIP_INTER_LINEAR);                   // :243-246
        if (c.RectifyColorLeft()) p.enqueueSendImage(GPU_MAT_SRC_L_RECT_COLOR, header, "bgr8", pub(pub_color_rect_left_));
        if (c.RectifyColorRight()) p.enqueueSendImage(GPU_MAT_SRC_R_RECT_COLOR, header, "bgr8", pub(pub_color_rect_right_));
        if (disp_any) {
            p.computeDisparity(GPU_MAT_SRC_L_RECT_MONO, GPU_MAT_SRC_R_RECT_MONO, GPU_MAT_SRC_L_DISPARITY);    // :261
            p.filterSpeckles(GPU_MAT_SRC_L_DISPARITY);                                                        // :262
        }
        if (c.Disparity()) p.enqueueSendDisparity(GPU_MAT_SRC_L_DISPARITY, header, pub_disparity_ ? &pub_disparity_ : nullptr);   // :265-268
        if (c.DisparityVis()) {
            p.computeDisparityImage(GPU_MAT_SRC_L_DISPARITY, GPU_MAT_SRC_L_DISPARITY_IMG);                    // :273
            p.enqueueSendImage(GPU_MAT_SRC_L_DISPARITY_IMG, header, "bgra8", pub(pub_disparity_vis_));        // :274
        }
        if (c.Pointcloud()) {
            p.projectDisparityTo3DPoints(GPU_MAT_SRC_L_DISPARITY, GPU_MAT_SRC_L_POINTS2);                     // :280
            p.enqueueSendPoints(GPU_MAT_SRC_L_POINTS2, GPU_MAT_SRC_L_RECT_COLOR, header, pub_pointcloud_ ? &pub_pointcloud_ : nullptr);  // :281
        }
        p.waitForAllStreams();                                                                                // :284
        p.cleanSenders();                                                                                     // :285
    }

  private:
    static const ImagePublisher *pub(const ImagePublisher &p) { return p ? &p : nullptr; }
    std::unique_ptr<GpuStereoProcessor> stereoProcessor_;
    ConnectedTopics connected_;
};

} // namespace gpuimageproc
