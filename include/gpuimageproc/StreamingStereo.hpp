// StreamingStereo.hpp -- C++ side of the pipelined host link (gip_streamer_* of gpuimageproc_hip.h) and of the multi-device deployments
// (MultiDeviceStereo: frames dealt over the devices; RowSplitStereo: one frame cut into row tiles over the devices).
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
    // depth / frames_per_batch 0 = automatic; adaptive: the streamer forms the batches from pair-by-pair submissions (gip_streamer_config::batching)
    StreamingStereo(GpuStereoProcessor &processor, int width, int height, const std::string &encoding, uint32_t outputs, ResultCb cb, int depth = 0,
                    int frames_per_batch = 0, bool want_disparity16 = false, bool adaptive = true)
        : cb_(std::move(cb)), width_(width), height_(height)
    {
        gip_streamer_config cfg{};
        cfg.width = width;
        cfg.height = height;
        cfg.encoding = streamer_encoding(encoding);
        cfg.outputs = outputs;
        cfg.want_disparity16 = want_disparity16 ? 1 : 0;
        cfg.depth = depth;
        cfg.frames_per_batch = frames_per_batch;
        cfg.batching = adaptive ? 0 : 1;
        channels_ = (cfg.encoding == 1 || cfg.encoding == 2) ? 3 : 1;
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
    void imageCb(const Mat &l_raw, const Mat &r_raw, const gip_header &header)
    {
        // gip_streamer_submit reads height x width x channels bytes from both pointers: a wrong-size Mat would be an out-of-bounds read
        if (l_raw.rows != height_ || l_raw.cols != width_ || r_raw.rows != height_ || r_raw.cols != width_ || l_raw.channels() != channels_ || r_raw.channels() != channels_ ||
            l_raw.elemSize() != (size_t)channels_ || r_raw.elemSize() != (size_t)channels_ || l_raw.step != r_raw.step || !l_raw.data || !r_raw.data)
            throw std::runtime_error("StreamingStereo::imageCb: image size / type differs from the streamer's configuration");
        ck(gip_streamer_submit(s_, l_raw.data, r_raw.data, l_raw.step, &header));
    }
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
    int width_ = 0, height_ = 0, channels_ = 1;
};

// One processor + streamer per device; frame i -> device i mod N.  With fewer devices than `n_handles` the handles share devices
// (useful for tests on a one-GPU box; on an 8-GPU node pass n_handles = 0 = one per visible device).
class MultiDeviceStereo
{
  public:
    typedef std::function<void(int device_slot, const gip_streamer_result &)> ResultCb;

    MultiDeviceStereo(int n_handles, const gip_camera_info &left, const gip_camera_info &right, const gip_bm_params &params, int max_speckle_size,
                      double max_speckle_diff, int width, int height, const std::string &encoding, uint32_t outputs, ResultCb cb, int depth = 0,
                      int frames_per_batch = 0)
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

// ONE large frame cut into row tiles over several devices (BASELINE configs[4]; gip_rowsplit_* of gpuimageproc_hip.h): one process owns
// one processor handle per tile (tile t on device t mod the visible devices), the halo rows of the rectified images travel by
// hipMemcpyPeerAsync, the speckle components that cross a tile border are merged on the device.  The reference runs every stage of a
// frame as one launch on one GPU and then waits (reference src/GPUStereoProcessor.cpp:283, src/StereoProcessor.cpp:284); this is the
// low-LATENCY deployment for frames that are large against one GPU's 0.7 ms (MultiDeviceStereo above is the throughput one).
class RowSplitStereo
{
  public:
    RowSplitStereo(int n_tiles, const gip_camera_info &left, const gip_camera_info &right, const gip_bm_params &params, int max_speckle_size,
                   double max_speckle_diff, int width, int height, const std::string &encoding, uint32_t outputs, int halo_mode = 0)
        : width_(width), height_(height), outputs_(outputs)
    {
        gip_rowsplit_config cfg{};
        cfg.width = width;
        cfg.height = height;
        cfg.encoding = streamer_encoding(encoding);
        cfg.outputs = outputs;
        cfg.n_tiles = n_tiles > 0 ? n_tiles : gip_device_count();
        for (int i = 0; i < 16; i++) cfg.devices[i] = -1;
        cfg.halo_mode = halo_mode;
        int rc = gip_rowsplit_open(&left, &right, &params, max_speckle_size, max_speckle_diff, &cfg, &rs_);
        if (rc != GIP_OK) throw std::runtime_error(std::string("gip_rowsplit_open: ") + gip_status_string(rc));
        n_tiles_ = cfg.n_tiles;
    }
    ~RowSplitStereo()
    {
        if (rs_) gip_rowsplit_close(rs_);
    }
    RowSplitStereo(const RowSplitStereo &) = delete;
    RowSplitStereo &operator=(const RowSplitStereo &) = delete;

    int tiles() const { return n_tiles_; }
    // the synchronous imageCb of a split frame: upload to every tile's device, run, stitch the results into host memory.
    // disparity16 / disparity32 / cloud may be null (cloud: height x width x 32 bytes, the PointCloud2 body).
    void imageCb(const Mat &l_raw, const Mat &r_raw, Mat *disparity16, Mat *disparity32, std::vector<uint8_t> *cloud)
    {
        if (l_raw.rows != height_ || l_raw.cols != width_ || r_raw.rows != height_ || r_raw.cols != width_ || l_raw.type() != r_raw.type() || l_raw.step != r_raw.step)
            throw std::runtime_error("RowSplitStereo::imageCb: image size / type differs from the configuration");
        ck(gip_rowsplit_upload(rs_, l_raw.data, r_raw.data, l_raw.step));
        ck(gip_rowsplit_run(rs_));
        if (disparity16) disparity16->create(height_, width_, GIP_16SC1);
        if (disparity32) disparity32->create(height_, width_, GIP_32FC1);
        if (cloud) cloud->resize((size_t)height_ * width_ * 32);
        ck(gip_rowsplit_download(rs_, disparity16 ? (int16_t *)disparity16->data : nullptr, disparity32 && (outputs_ & GIP_TOPIC_DISPARITY) ? (float *)disparity32->data : nullptr,
                                 cloud && (outputs_ & GIP_TOPIC_POINTCLOUD) ? cloud->data() : nullptr));
    }
    // asynchronous form: frames already on the tiles' devices (gip_rowsplit_set_inputs_device), run() enqueues, wait() drains
    gip_rowsplit handle() { return rs_; }
    void run() { ck(gip_rowsplit_run(rs_)); }
    void wait() { ck(gip_rowsplit_wait(rs_)); }
    // per-tile GPU time with every tile's device to itself (what a rank on its own GPU spends), one entry per tile
    std::vector<gip_rowsplit_timing> profileSerial()
    {
        std::vector<gip_rowsplit_timing> t(n_tiles_);
        ck(gip_rowsplit_run_serial(rs_, t.data()));
        return t;
    }

  private:
    void ck(int rc)
    {
        if (rc != GIP_OK) throw std::runtime_error(std::string("gip_rowsplit: ") + gip_rowsplit_last_error(rs_));
    }
    gip_rowsplit rs_ = nullptr;
    int n_tiles_ = 0, width_, height_;
    uint32_t outputs_;
};

}  // namespace gpuimageproc
