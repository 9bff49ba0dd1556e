// StreamingStereo.hpp -- C++ side of the pipelined host link (gip_streamer_* of gpuimageproc_hip.h) and of the multi-device deployment.
//
// The reference's StereoProcessor::imageCb uploads a pair, enqueues the stages, hands every output to a stream host callback and then
// blocks on waitForAllStreams() (reference src/StereoProcessor.cpp:179-180, :284; src/GpuSenderIfc.cpp:13-26): one frame in flight.
// StreamingStereo keeps the same callbacks but several frames in flight: imageCb() only copies the pair into a pinned ring and returns;
// results arrive on a HIP runtime thread in submission order.  MultiDeviceStereo owns one GpuStereoProcessor + StreamingStereo per
// visible device and deals frames round-robin -- frame i -> device i mod N; every handle builds its rectification tables from the same
// CameraInfo doubles, so they are bit-identical on every device and no collective is needed.
#pragma once
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gpuimageproc/GPUStereoProcessor.hpp"

namespace gpuimageproc
{

inline int streamer_encoding(const std::string &encoding)
{
    static const char *names[] = {"mono8", "bgr8", "rgb8", "bayer_rggb8", "bayer_bggr8", "bayer_gbrg8", "bayer_grbg8"};
    for (int i = 0; i < 7; i++)
        if (encoding == names[i]) return i;
    throw std::runtime_error("StreamingStereo: unsupported encoding " + encoding);
}

class StreamingStereo
{
  public:
    typedef std::function<void(const gip_streamer_result &)> ResultCb;

    // processor: model and parameters set; outputs: GIP_TOPIC_DISPARITY and/or GIP_TOPIC_POINTCLOUD
    StreamingStereo(GpuStereoProcessor &processor, int width, int height, const std::string &encoding, uint32_t outputs, ResultCb cb, int depth = 3,
                    int frames_per_batch = 1, bool want_disparity16 = false)
        : cb_(std::move(cb))
    {
        gip_streamer_config cfg{};
        cfg.width = width;
        cfg.height = height;
        cfg.encoding = streamer_encoding(encoding);
        cfg.outputs = outputs;
        cfg.want_disparity16 = want_disparity16 ? 1 : 0;
        cfg.depth = depth;
        cfg.frames_per_batch = frames_per_batch;
        int rc = gip_streamer_open(processor.handle(), &cfg, &StreamingStereo::trampoline, this, &s_);
        if (rc != GIP_OK) throw std::runtime_error(std::string("gip_streamer_open: ") + gip_status_string(rc));
    }
    ~StreamingStereo()
    {
        if (s_) gip_streamer_close(s_);
    }
    StreamingStereo(const StreamingStereo &) = delete;
    StreamingStereo &operator=(const StreamingStereo &) = delete;

    // the streaming imageCb: returns as soon as the pair is in the ring (blocks only when every ring slot is in flight)
    void imageCb(const Mat &l_raw, const Mat &r_raw, const gip_header &header) { ck(gip_streamer_submit(s_, l_raw.data, r_raw.data, l_raw.step, &header)); }
    void flush() { ck(gip_streamer_flush(s_)); }
    uint64_t delivered() const { return gip_streamer_delivered(s_); }

  private:
    static void trampoline(const gip_streamer_result *r, void *user)
    {
        StreamingStereo *self = static_cast<StreamingStereo *>(user);
        if (self->cb_) self->cb_(*r);
    }
    void ck(int rc)
    {
        if (rc != GIP_OK) throw std::runtime_error(std::string("gip_streamer: ") + gip_streamer_last_error(s_));
    }
    gip_streamer s_ = nullptr;
    ResultCb cb_;
};

// One processor + streamer per device; frame i -> device i mod N.  With fewer devices than `n_handles` the handles share devices
// (useful for tests on a one-GPU box; on an 8-GPU node pass n_handles = 0 = one per visible device).
class MultiDeviceStereo
{
  public:
    typedef std::function<void(int device_slot, const gip_streamer_result &)> ResultCb;

    MultiDeviceStereo(int n_handles, const gip_camera_info &left, const gip_camera_info &right, const gip_bm_params &params, int max_speckle_size,
                      double max_speckle_diff, int width, int height, const std::string &encoding, uint32_t outputs, ResultCb cb, int depth = 3,
                      int frames_per_batch = 1)
    {
        const int ndev = gip_device_count();
        if (ndev <= 0) throw std::runtime_error("MultiDeviceStereo: no HIP device");
        if (n_handles <= 0) n_handles = ndev;
        for (int i = 0; i < n_handles; i++) {
            procs_.emplace_back(new GpuStereoProcessor(i % ndev));
            GpuStereoProcessor &p = *procs_.back();
            p.initStereoModel(left, right);
            p.setParams(params);
            p.setMaxSpeckleSize(max_speckle_size);
            p.setMaxSpeckleDiff(max_speckle_diff);
            streams_.emplace_back(new StreamingStereo(p, width, height, encoding, outputs, [cb, i](const gip_streamer_result &r) { cb(i, r); }, depth, frames_per_batch));
        }
    }
    int size() const { return (int)streams_.size(); }
    // frame `frame_index` goes to handle frame_index mod N
    void imageCb(uint64_t frame_index, const Mat &l_raw, const Mat &r_raw, const gip_header &header) { streams_[frame_index % streams_.size()]->imageCb(l_raw, r_raw, header); }
    void flush()
    {
        for (auto &s : streams_) s->flush();
    }

  private:
    // streams before processors in destruction order: a streamer must be closed before its processor is destroyed
    std::vector<std::unique_ptr<GpuStereoProcessor>> procs_;
    std::vector<std::unique_ptr<StreamingStereo>> streams_;
};

}  // namespace gpuimageproc
