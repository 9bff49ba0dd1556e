// ros/StereoProcessorNodelet.cpp -- ROS1 shim over the ROS-free classes of include/gpuimageproc/*.hpp: the nodelet `gpuimageproc/Stereoproc`
// (same plugin name, topics and parameters as the reference's src/StereoProcessorNodelet.cpp:6-18 + src/StereoProcessor.cpp:20-142,
// plugins/nodelet_plugins.xml:1-2), built only when catkin is found (CMakeLists.txt).  NOT compiled in the build image (no ROS there):
// everything numeric lives below the C-ABI and is tested without ROS; this file only moves sensor_msgs in and out.
//
//   inputs   left/image_raw, right/image_raw (image_transport "raw"), left/camera_info, right/camera_info
//            (or ~camera_info_file_left / ~camera_info_file_right: calibration files, no CameraInfo subscription)
//   outputs  {left,right}/image_mono, image_color, rect_mono, rect_color, disparity, disparity_vis, pointcloud -- computed only while
//            somebody is subscribed (ConnectedTopics), as in the reference
//   params   ~queue_size (5), ~approximate_sync (false), ~publisher_queue_size (1), ~device (-1 = current), ~streaming (false),
//            dynamic_reconfigure: ros/cfg/GPU.cfg
//
// ~streaming:=true replaces the reference's one-frame-in-flight cycle (upload, enqueue, waitForAllStreams: src/StereoProcessor.cpp:179-180,
// :284) for the disparity and pointcloud topics by the pipelined host link (gip_streamer_*): imageCb returns as soon as the pair is in the
// pinned ring and the messages are published from the streamer's callback, in order.
#include <algorithm>
#include <cstdio>
#include <memory>
#include <string>

#include <boost/bind.hpp>
#include <boost/make_shared.hpp>
#include <boost/thread/recursive_mutex.hpp>

#include <dynamic_reconfigure/server.h>
#include <image_transport/image_transport.h>
#include <image_transport/subscriber_filter.h>
#include <message_filters/subscriber.h>
#include <message_filters/sync_policies/approximate_time.h>
#include <message_filters/sync_policies/exact_time.h>
#include <message_filters/synchronizer.h>
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/image_encodings.h>
#include <stereo_msgs/DisparityImage.h>

#include <gpuimageproc_hip/GPUConfig.h>      // generated from ros/cfg/GPU.cfg

#include "gpuimageproc/StereoProcessor.hpp"
#include "gpuimageproc/StreamingStereo.hpp"

namespace gpuimageproc
{

namespace
{

gip_header to_gip(const std_msgs::Header &h)
{
    gip_header g{};
    g.seq = h.seq;
    g.stamp_sec = h.stamp.sec;
    g.stamp_nsec = h.stamp.nsec;
    std::snprintf(g.frame_id, sizeof g.frame_id, "%s", h.frame_id.c_str());
    return g;
}
std_msgs::Header to_ros(const gip_header &g)
{
    std_msgs::Header h;
    h.seq = g.seq;
    h.stamp.sec = g.stamp_sec;
    h.stamp.nsec = g.stamp_nsec;
    h.frame_id = g.frame_id;
    return h;
}
gip_camera_info to_gip(const sensor_msgs::CameraInfo &ci)
{
    gip_camera_info g{};
    g.width = (int32_t)ci.width;
    g.height = (int32_t)ci.height;
    for (int i = 0; i < 9; i++) { g.K[i] = ci.K[i]; g.R[i] = ci.R[i]; }
    for (int i = 0; i < 12; i++) g.P[i] = ci.P[i];
    g.nD = (int32_t)std::min<size_t>(ci.D.size(), 8);
    for (int i = 0; i < g.nD; i++) g.D[i] = ci.D[i];
    g.binning_x = ci.binning_x; g.binning_y = ci.binning_y;
    g.roi_x_offset = ci.roi.x_offset; g.roi_y_offset = ci.roi.y_offset; g.roi_height = ci.roi.height; g.roi_width = ci.roi.width;
    return g;
}
// a view of the message's pixels (no copy), like cv_bridge::toCvShare in the reference (src/StereoProcessor.cpp:176-177)
Mat view_of(const sensor_msgs::Image &img)
{
    const int cn = sensor_msgs::image_encodings::numChannels(img.encoding);
    const int depth = sensor_msgs::image_encodings::bitDepth(img.encoding) == 16 ? GIP_16U : GIP_8U;
    return Mat((int)img.height, (int)img.width, GIP_MAKETYPE(depth, cn), const_cast<uint8_t *>(img.data.data()), img.step);
}
sensor_msgs::ImagePtr to_ros(const gip_image_msg &m)
{
    sensor_msgs::ImagePtr out = boost::make_shared<sensor_msgs::Image>();
    out->header = to_ros(m.header);
    out->height = m.height; out->width = m.width; out->encoding = m.encoding; out->is_bigendian = m.is_bigendian; out->step = m.step;
    out->data.assign(m.data, m.data + m.data_size);
    return out;
}
stereo_msgs::DisparityImagePtr to_ros(const gip_disparity_msg &m)
{
    stereo_msgs::DisparityImagePtr out = boost::make_shared<stereo_msgs::DisparityImage>();
    out->header = to_ros(m.header);
    out->image = *to_ros(m.image);
    out->f = m.f; out->T = m.T;
    out->valid_window.x_offset = m.valid_x_offset; out->valid_window.y_offset = m.valid_y_offset;
    out->valid_window.width = m.valid_width; out->valid_window.height = m.valid_height;
    out->min_disparity = m.min_disparity; out->max_disparity = m.max_disparity; out->delta_d = m.delta_d;
    return out;
}
sensor_msgs::PointCloud2Ptr to_ros(const gip_pointcloud2_msg &m)
{
    sensor_msgs::PointCloud2Ptr out = boost::make_shared<sensor_msgs::PointCloud2>();
    out->header = to_ros(m.header);
    out->height = m.height; out->width = m.width; out->is_bigendian = m.is_bigendian; out->is_dense = m.is_dense;
    out->point_step = m.point_step; out->row_step = m.row_step;
    out->fields.resize(m.nfields);
    for (uint32_t i = 0; i < m.nfields; i++) {
        out->fields[i].name = m.fields[i].name; out->fields[i].offset = m.fields[i].offset;
        out->fields[i].datatype = m.fields[i].datatype; out->fields[i].count = m.fields[i].count;
    }
    out->data.assign(m.data, m.data + m.data_size);
    return out;
}

}  // namespace

class RosStereoProcessor
{
    typedef sensor_msgs::Image Image;
    typedef sensor_msgs::CameraInfo CameraInfo;
    typedef message_filters::sync_policies::ExactTime<Image, Image> ExactImages;
    typedef message_filters::sync_policies::ApproximateTime<Image, Image> ApproxImages;
    typedef message_filters::sync_policies::ExactTime<Image, CameraInfo, Image, CameraInfo> ExactWithInfo;
    typedef message_filters::sync_policies::ApproximateTime<Image, CameraInfo, Image, CameraInfo> ApproxWithInfo;

  public:
    RosStereoProcessor(ros::NodeHandle &nh, ros::NodeHandle &pnh) : nh_(nh), pnh_(pnh), it_(nh)
    {
        int device = -1, queue_size = 5, pub_queue = 1;
        bool approx = false;
        pnh_.param("device", device, -1);
        pnh_.param("queue_size", queue_size, 5);
        pnh_.param("approximate_sync", approx, false);
        pnh_.param("publisher_queue_size", pub_queue, 1);
        pnh_.param("streaming", streaming_, false);
        pnh_.param("camera_info_file_left", cal_left_, std::string());
        pnh_.param("camera_info_file_right", cal_right_, std::string());
        from_files_ = !cal_left_.empty() && !cal_right_.empty();
        core_.reset(new StereoProcessor(device));

        // one of four synchronisers: {exact, approximate} x {images only, images + CameraInfo}
        if (from_files_) {
            if (approx) { approx_images_.reset(new message_filters::Synchronizer<ApproxImages>(ApproxImages(queue_size), sub_l_image_, sub_r_image_));
                          approx_images_->registerCallback(boost::bind(&RosStereoProcessor::onImages, this, _1, _2)); }
            else { exact_images_.reset(new message_filters::Synchronizer<ExactImages>(ExactImages(queue_size), sub_l_image_, sub_r_image_));
                   exact_images_->registerCallback(boost::bind(&RosStereoProcessor::onImages, this, _1, _2)); }
        } else {
            if (approx) { approx_info_.reset(new message_filters::Synchronizer<ApproxWithInfo>(ApproxWithInfo(queue_size), sub_l_image_, sub_l_info_, sub_r_image_, sub_r_info_));
                          approx_info_->registerCallback(boost::bind(&RosStereoProcessor::onImagesAndInfo, this, _1, _2, _3, _4)); }
            else { exact_info_.reset(new message_filters::Synchronizer<ExactWithInfo>(ExactWithInfo(queue_size), sub_l_image_, sub_l_info_, sub_r_image_, sub_r_info_));
                   exact_info_->registerCallback(boost::bind(&RosStereoProcessor::onImagesAndInfo, this, _1, _2, _3, _4)); }
        }

        reconfigure_.reset(new dynamic_reconfigure::Server<gpuimageproc_hip::GPUConfig>(mutex_, pnh_));
        reconfigure_->setCallback(boost::bind(&RosStereoProcessor::onConfig, this, _1, _2));

        // lazy publishers: every (dis)connection re-evaluates which stages are needed
        ros::SubscriberStatusCallback img_cb = boost::bind(&RosStereoProcessor::onConnect, this);
        image_transport::SubscriberStatusCallback it_cb = boost::bind(&RosStereoProcessor::onConnect, this);
        boost::lock_guard<boost::recursive_mutex> lock(mutex_);
        pub_[0] = it_.advertise("left/image_mono", pub_queue, it_cb, it_cb);
        pub_[1] = it_.advertise("right/image_mono", pub_queue, it_cb, it_cb);
        pub_[2] = it_.advertise("left/image_color", pub_queue, it_cb, it_cb);
        pub_[3] = it_.advertise("right/image_color", pub_queue, it_cb, it_cb);
        pub_[4] = it_.advertise("left/rect_mono", pub_queue, it_cb, it_cb);
        pub_[5] = it_.advertise("right/rect_mono", pub_queue, it_cb, it_cb);
        pub_[6] = it_.advertise("left/rect_color", pub_queue, it_cb, it_cb);
        pub_[7] = it_.advertise("right/rect_color", pub_queue, it_cb, it_cb);
        pub_vis_ = it_.advertise("disparity_vis", pub_queue, it_cb, it_cb);
        pub_disparity_ = nh_.advertise<stereo_msgs::DisparityImage>("disparity", pub_queue, img_cb, img_cb);
        pub_cloud_ = nh_.advertise<sensor_msgs::PointCloud2>("pointcloud", pub_queue, img_cb, img_cb);

        // the core publishes through std::function callbacks: forward them to the ROS publishers
        ImagePublisher *core_pubs[8] = {&core_->pub_mono_left_, &core_->pub_mono_right_, &core_->pub_color_left_, &core_->pub_color_right_,
                                        &core_->pub_mono_rect_left_, &core_->pub_mono_rect_right_, &core_->pub_color_rect_left_, &core_->pub_color_rect_right_};
        for (int i = 0; i < 8; i++) *core_pubs[i] = [this, i](const gip_image_msg &m) { pub_[i].publish(to_ros(m)); };
        core_->pub_disparity_vis_ = [this](const gip_image_msg &m) { pub_vis_.publish(to_ros(m)); };
        core_->pub_disparity_ = [this](const gip_disparity_msg &m) { pub_disparity_.publish(to_ros(m)); };
        core_->pub_pointcloud_ = [this](const gip_pointcloud2_msg &m) { pub_cloud_.publish(to_ros(m)); };
    }

  private:
    void onConnect()
    {
        boost::lock_guard<boost::recursive_mutex> lock(mutex_);
        ConnectedTopics &c = core_->connected();
        c.DebayerMonoLeft(pub_[0].getNumSubscribers() > 0);   c.DebayerMonoRight(pub_[1].getNumSubscribers() > 0);
        c.DebayerColorLeft(pub_[2].getNumSubscribers() > 0);  c.DebayerColorRight(pub_[3].getNumSubscribers() > 0);
        c.RectifyMonoLeft(pub_[4].getNumSubscribers() > 0);   c.RectifyMonoRight(pub_[5].getNumSubscribers() > 0);
        c.RectifyColorLeft(pub_[6].getNumSubscribers() > 0);  c.RectifyColorRight(pub_[7].getNumSubscribers() > 0);
        c.Disparity(pub_disparity_.getNumSubscribers() > 0);
        c.DisparityVis(pub_vis_.getNumSubscribers() > 0);
        c.Pointcloud(pub_cloud_.getNumSubscribers() > 0);
        if (c.level() == 0) {                 // nobody listens: stop pulling images
            sub_l_image_.unsubscribe(); sub_r_image_.unsubscribe(); sub_l_info_.unsubscribe(); sub_r_info_.unsubscribe();
            subscribed_ = false;
        } else if (!subscribed_) {
            image_transport::TransportHints hints("raw", ros::TransportHints(), pnh_);
            sub_l_image_.subscribe(it_, "left/image_raw", 1, hints);
            sub_r_image_.subscribe(it_, "right/image_raw", 1, hints);
            if (!from_files_) { sub_l_info_.subscribe(nh_, "left/camera_info", 1); sub_r_info_.subscribe(nh_, "right/camera_info", 1); }
            subscribed_ = true;
        }
    }

    void onConfig(gpuimageproc_hip::GPUConfig &cfg, uint32_t)
    {
        boost::lock_guard<boost::recursive_mutex> lock(mutex_);
        GPUConfig c;
        c.xsobel = cfg.xsobel; c.refine_disparity = cfg.refine_disparity; c.correlation_window_size = cfg.correlation_window_size;
        c.disparity_min = cfg.disparity_min; c.disparity_range = cfg.disparity_range; c.texture_threshold = cfg.texture_threshold;
        c.max_speckle_size = cfg.max_speckle_size; c.max_speckle_diff = cfg.max_speckle_diff;
        stream_.reset();                      // a streamer captures the parameters at open: reopen with the new ones
        core_->configCb(c);
        cfg.correlation_window_size = c.correlation_window_size;      // report the values the fix-ups produced (odd window, range % 16)
        cfg.disparity_range = c.disparity_range;
    }

    void onImagesAndInfo(const sensor_msgs::ImageConstPtr &l, const sensor_msgs::CameraInfoConstPtr &li, const sensor_msgs::ImageConstPtr &r,
                         const sensor_msgs::CameraInfoConstPtr &ri)
    {
        boost::lock_guard<boost::recursive_mutex> lock(mutex_);
        if (!core_->processor().isStereoModelInitialised()) core_->processor().initStereoModel(to_gip(*li), to_gip(*ri));
        process(l, r);
    }
    void onImages(const sensor_msgs::ImageConstPtr &l, const sensor_msgs::ImageConstPtr &r)
    {
        boost::lock_guard<boost::recursive_mutex> lock(mutex_);
        if (!core_->processor().isStereoModelInitialised()) core_->processor().initStereoModel(cal_left_, cal_right_);
        process(l, r);
    }

    void process(const sensor_msgs::ImageConstPtr &l, const sensor_msgs::ImageConstPtr &r)
    {
        const Mat lm = view_of(*l), rm = view_of(*r);
        const gip_header hdr = to_gip(l->header);
        const ConnectedTopics &c = core_->connected();
        const bool only_streamable = c.level() >= 9 && !c.DisparityVis() && !(c.DebayerMonoLeft() || c.DebayerMonoRight() || c.DebayerColorLeft() || c.DebayerColorRight() ||
                                                                               c.RectifyMonoLeft() || c.RectifyMonoRight() || c.RectifyColorLeft() || c.RectifyColorRight());
        try {
            if (streaming_ && only_streamable) {
                const uint32_t outputs = (c.Disparity() ? GIP_TOPIC_DISPARITY : 0u) | (c.Pointcloud() ? GIP_TOPIC_POINTCLOUD : 0u);
                if (!stream_ || stream_outputs_ != outputs || stream_encoding_ != l->encoding) openStream((int)l->width, (int)l->height, l->encoding, outputs);
                stream_->imageCb(lm, rm, hdr);                // returns once the pair is in the pinned ring; publishStreamed() delivers
            } else {
                stream_.reset();
                core_->imageCb(lm, rm, l->encoding, hdr);     // the reference's synchronous schedule
            }
        } catch (const std::exception &e) {
            ROS_ERROR_THROTTLE(1.0, "gpuimageproc: %s", e.what());
        }
    }

    void openStream(int w, int h, const std::string &encoding, uint32_t outputs)
    {
        stream_.reset(new StreamingStereo(core_->processor(), w, h, encoding, outputs, boost::bind(&RosStereoProcessor::publishStreamed, this, _1)));
        stream_outputs_ = outputs;
        stream_encoding_ = encoding;
        stream_meta_ = core_->processor().disparityInfo(w, h);      // fixed while the streamer is open (onConfig closes it)
    }

    // HIP runtime thread, once per pair, in submission order
    void publishStreamed(const gip_streamer_result &res)
    {
        const std_msgs::Header hdr = to_ros(res.header);
        const size_t npx = (size_t)res.width * res.height;
        if (res.disparity_f32 && (stream_outputs_ & GIP_TOPIC_DISPARITY)) {
            stereo_msgs::DisparityImagePtr d = boost::make_shared<stereo_msgs::DisparityImage>();
            d->header = hdr; d->image.header = hdr;
            d->image.height = res.height; d->image.width = res.width; d->image.encoding = sensor_msgs::image_encodings::TYPE_32FC1; d->image.step = res.width * 4;
            d->image.data.assign((const uint8_t *)res.disparity_f32, (const uint8_t *)res.disparity_f32 + npx * 4);
            // valid window / f / T / disparity range: the numbers GPUSenderDisparity::fillInData computes (reference src/GpuSenderDisparity.cpp:29-43)
            const gip_disparity_msg &m = stream_meta_;
            d->f = m.f; d->T = m.T; d->min_disparity = m.min_disparity; d->max_disparity = m.max_disparity; d->delta_d = m.delta_d;
            d->valid_window.x_offset = m.valid_x_offset; d->valid_window.y_offset = m.valid_y_offset;
            d->valid_window.width = m.valid_width; d->valid_window.height = m.valid_height;
            pub_disparity_.publish(d);
        }
        if (res.pointcloud) {
            sensor_msgs::PointCloud2Ptr pc = boost::make_shared<sensor_msgs::PointCloud2>();
            pc->header = hdr;
            pc->height = res.height; pc->width = res.width; pc->is_dense = false; pc->is_bigendian = false; pc->point_step = 32; pc->row_step = 32 * res.width;
            const char *names[4] = {"x", "y", "z", "rgb"};
            const uint32_t offs[4] = {0, 4, 8, 16};
            pc->fields.resize(4);
            for (int i = 0; i < 4; i++) { pc->fields[i].name = names[i]; pc->fields[i].offset = offs[i]; pc->fields[i].datatype = sensor_msgs::PointField::FLOAT32; pc->fields[i].count = 1; }
            pc->data.assign(res.pointcloud, res.pointcloud + npx * 32);
            pub_cloud_.publish(pc);
        }
    }
    ros::NodeHandle nh_, pnh_;
    image_transport::ImageTransport it_;
    image_transport::SubscriberFilter sub_l_image_, sub_r_image_;
    message_filters::Subscriber<CameraInfo> sub_l_info_, sub_r_info_;
    boost::shared_ptr<message_filters::Synchronizer<ExactImages>> exact_images_;
    boost::shared_ptr<message_filters::Synchronizer<ApproxImages>> approx_images_;
    boost::shared_ptr<message_filters::Synchronizer<ExactWithInfo>> exact_info_;
    boost::shared_ptr<message_filters::Synchronizer<ApproxWithInfo>> approx_info_;
    boost::shared_ptr<dynamic_reconfigure::Server<gpuimageproc_hip::GPUConfig>> reconfigure_;
    boost::recursive_mutex mutex_;
    image_transport::Publisher pub_[8], pub_vis_;
    ros::Publisher pub_disparity_, pub_cloud_;
    std::unique_ptr<StereoProcessor> core_;
    std::unique_ptr<StreamingStereo> stream_;
    uint32_t stream_outputs_ = 0;
    gip_disparity_msg stream_meta_{};
    std::string stream_encoding_, cal_left_, cal_right_;
    bool from_files_ = false, streaming_ = false, subscribed_ = false;
};

class StereoProcNodelet : public nodelet::Nodelet
{
  public:
    void onInit() override { impl_.reset(new RosStereoProcessor(getNodeHandle(), getPrivateNodeHandle())); }

  private:
    std::unique_ptr<RosStereoProcessor> impl_;
};

}  // namespace gpuimageproc

PLUGINLIB_EXPORT_CLASS(gpuimageproc::StereoProcNodelet, nodelet::Nodelet)
