// gip_rosmsg.cpp -- ROS1 wire serialisation of the three messages the hot path publishes (host code, no ROS needed).
//
// The reference hands sensor_msgs/Image, stereo_msgs/DisparityImage and sensor_msgs/PointCloud2 objects to ros::Publisher
// (reference src/GpuSenderImage.cpp:15-33, src/GpuSenderDisparity.cpp:18-56, src/GpuSenderPc2.cpp:15-89); roscpp then writes them in
// the ROS1 serialisation format: fields in declaration order, little-endian, no padding, strings and variable-length arrays
// prefixed by a uint32 length, time = uint32 secs + uint32 nsecs, bool = uint8.  These functions produce exactly that body from the
// POD messages of gpuimageproc_hip.h (what a rosbag record or a TCPROS frame carries after its own length prefix), so a ROS-free
// consumer of the sender / streamer callbacks can write bag-compatible messages.
//   std_msgs/Header         : uint32 seq, time stamp, string frame_id
//   sensor_msgs/Image       : Header, uint32 height, uint32 width, string encoding, uint8 is_bigendian, uint32 step, uint8[] data
//   sensor_msgs/RegionOfInterest : uint32 x_offset, y_offset, height, width, bool do_rectify
//   stereo_msgs/DisparityImage   : Header, Image image, float32 f, float32 T, RegionOfInterest valid_window, float32 min_disparity,
//                                  float32 max_disparity, float32 delta_d
//   sensor_msgs/PointField  : string name, uint32 offset, uint8 datatype, uint32 count
//   sensor_msgs/PointCloud2 : Header, uint32 height, uint32 width, PointField[] fields, bool is_bigendian, uint32 point_step,
//                             uint32 row_step, uint8[] data, bool is_dense
#include <cstring>

#include "gpuimageproc_hip.h"

namespace {

struct Writer {
    uint8_t* p;
    size_t cap, n = 0;
    void bytes(const void* src, size_t k)
    {
        if (p && n + k <= cap) std::memcpy(p + n, src, k);
        n += k;
    }
    void u8(uint8_t v) { bytes(&v, 1); }
    void u32(uint32_t v) { bytes(&v, 4); }          // the target is little-endian, as the wire format
    void f32(float v) { bytes(&v, 4); }
    void str(const char* s, size_t maxlen)
    {
        const uint32_t k = (uint32_t)strnlen(s, maxlen);
        u32(k);
        bytes(s, k);
    }
    void header(const gip_header& h)
    {
        u32(h.seq);
        u32(h.stamp_sec);
        u32(h.stamp_nsec);
        str(h.frame_id, sizeof h.frame_id);
    }
    void image(const gip_image_msg& m)
    {
        header(m.header);
        u32(m.height);
        u32(m.width);
        str(m.encoding, sizeof m.encoding);
        u8(m.is_bigendian);
        u32(m.step);
        u32((uint32_t)m.data_size);
        bytes(m.data, m.data_size);
    }
    int64_t done() const { return (p && n > cap) ? -(int64_t)n : (int64_t)n; }
};

}  // namespace

extern "C" {

int64_t gip_ros1_serialize_image(const gip_image_msg* m, uint8_t* out, size_t cap)
{
    if (!m || (m->data_size && !m->data) || m->data_size > 0xFFFFFFFFull) return 0;
    Writer w{out, cap};
    w.image(*m);
    return w.done();
}

int64_t gip_ros1_serialize_disparity(const gip_disparity_msg* m, uint8_t* out, size_t cap)
{
    if (!m || (m->image.data_size && !m->image.data) || m->image.data_size > 0xFFFFFFFFull) return 0;
    Writer w{out, cap};
    w.header(m->header);
    w.image(m->image);
    w.f32(m->f);
    w.f32(m->T);
    w.u32(m->valid_x_offset);
    w.u32(m->valid_y_offset);
    w.u32(m->valid_height);
    w.u32(m->valid_width);
    w.u8(0);   // do_rectify: the reference never sets it
    w.f32(m->min_disparity);
    w.f32(m->max_disparity);
    w.f32(m->delta_d);
    return w.done();
}

int64_t gip_ros1_serialize_pointcloud2(const gip_pointcloud2_msg* m, uint8_t* out, size_t cap)
{
    if (!m || (m->data_size && !m->data) || m->nfields > 4 || m->data_size > 0xFFFFFFFFull) return 0;
    Writer w{out, cap};
    w.header(m->header);
    w.u32(m->height);
    w.u32(m->width);
    w.u32(m->nfields);
    for (uint32_t i = 0; i < m->nfields; i++) {
        w.str(m->fields[i].name, sizeof m->fields[i].name);
        w.u32(m->fields[i].offset);
        w.u8(m->fields[i].datatype);
        w.u32(m->fields[i].count);
    }
    w.u8(m->is_bigendian);
    w.u32(m->point_step);
    w.u32(m->row_step);
    w.u32((uint32_t)m->data_size);
    w.bytes(m->data, m->data_size);
    w.u8(m->is_dense);
    return w.done();
}

}  // extern "C"
