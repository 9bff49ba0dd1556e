"""CPU test: the ROS1 wire serialisation of the published messages (gip_ros1_serialize_*) against an independent struct-based writer that
follows the message definitions (std_msgs/Header, sensor_msgs/Image, stereo_msgs/DisparityImage, sensor_msgs/PointCloud2 + PointField,
sensor_msgs/RegionOfInterest).  Host-only entry points: no GPU needed."""
import ctypes as C
import struct

import numpy as np

from ros_gpu_stereo_processor_amd import capi


def _s(b):
    return struct.pack("<I", len(b)) + b


def _hdr(seq, sec, nsec, frame):
    return struct.pack("<III", seq, sec, nsec) + _s(frame)


def _image(h, height, width, enc, step, data):
    return h + struct.pack("<II", height, width) + _s(enc) + struct.pack("<BI", 0, step) + _s(data)


def _header(seq, sec, nsec, frame):
    return capi.Header(seq=seq, stamp_sec=sec, stamp_nsec=nsec, frame_id=frame)


def _call(fn, msg):
    n = fn(C.byref(msg), None, 0)
    assert n > 0
    buf = (C.c_uint8 * n)()
    assert fn(C.byref(msg), buf, n) == n
    assert fn(C.byref(msg), buf, n - 1) == -n            # too small: says how much is needed
    return bytes(buf)


def test_ros1_serialisation_matches_the_message_definitions():
    L = capi.lib()
    rng = np.random.default_rng(1)
    # sensor_msgs/Image
    h, w = 5, 7
    pix = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img = capi.ImageMsg()
    img.header = _header(9, 100, 2000, b"left_optical")
    img.height, img.width, img.encoding, img.is_bigendian, img.step = h, w, b"bgra8", 0, w * 4
    img.data = pix.ctypes.data_as(C.POINTER(C.c_uint8))
    img.data_size = pix.nbytes
    hb = _hdr(9, 100, 2000, b"left_optical")
    assert _call(L.gip_ros1_serialize_image, img) == _image(hb, h, w, b"bgra8", w * 4, pix.tobytes())
    # stereo_msgs/DisparityImage
    d = rng.random((h, w), dtype=np.float32)
    dm = capi.DisparityMsg()
    dm.header = _header(10, 101, 5, b"cam")
    dm.image.header = dm.header
    dm.image.height, dm.image.width, dm.image.encoding, dm.image.is_bigendian, dm.image.step = h, w, b"32FC1", 0, w * 4
    dm.image.data = d.ctypes.data_as(C.POINTER(C.c_uint8))
    dm.image.data_size = d.nbytes
    dm.f, dm.T = 458.5, 0.11
    dm.valid_x_offset, dm.valid_y_offset, dm.valid_width, dm.valid_height = 70, 7, 600, 400
    dm.min_disparity, dm.max_disparity, dm.delta_d = 1.0, 63.0, 0.0625
    hb = _hdr(10, 101, 5, b"cam")
    exp = hb + _image(hb, h, w, b"32FC1", w * 4, d.tobytes()) + struct.pack("<ff", 458.5, np.float32(0.11)) + struct.pack("<IIIIB", 70, 7, 400, 600, 0) + \
        struct.pack("<fff", 1.0, 63.0, 0.0625)
    assert _call(L.gip_ros1_serialize_disparity, dm) == exp
    # sensor_msgs/PointCloud2 (the reference's organised 32-byte layout: x y z @0/4/8, rgb @16, all FLOAT32)
    pts = rng.integers(0, 256, (h, w, 32), dtype=np.uint8)
    pc = capi.PointCloud2Msg()
    pc.header = _header(11, 102, 6, b"")
    pc.height, pc.width, pc.nfields, pc.is_bigendian, pc.point_step, pc.row_step, pc.is_dense = h, w, 4, 0, 32, 32 * w, 0
    for i, (name, off) in enumerate(((b"x", 0), (b"y", 4), (b"z", 8), (b"rgb", 16))):
        pc.fields[i].name, pc.fields[i].offset, pc.fields[i].datatype, pc.fields[i].count = name, off, 7, 1
    pc.data = pts.ctypes.data_as(C.POINTER(C.c_uint8))
    pc.data_size = pts.nbytes
    exp = _hdr(11, 102, 6, b"") + struct.pack("<II", h, w) + struct.pack("<I", 4)
    for name, off in ((b"x", 0), (b"y", 4), (b"z", 8), (b"rgb", 16)):
        exp += _s(name) + struct.pack("<IBI", off, 7, 1)
    exp += struct.pack("<BII", 0, 32, 32 * w) + _s(pts.tobytes()) + struct.pack("<B", 0)
    assert _call(L.gip_ros1_serialize_pointcloud2, pc) == exp
    # invalid message: data pointer missing
    bad = capi.ImageMsg()
    bad.data_size = 4
    assert L.gip_ros1_serialize_image(C.byref(bad), None, 0) == 0
