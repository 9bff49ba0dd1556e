"""Host-side mirror of the reference's C++ class gpuimageproc::GpuStereoProcessor
(reference include/gpuimageproc/GPUStereoProcessor.h:63-104) on top of the C-ABI.

Method names, argument meaning and error behaviour follow the reference so that the parity tests read like the
reference's own gtests (test/UTest.cpp); cv::Mat becomes a numpy array.  Every call goes through
libgpuimageproc_hip.so -- there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (L_RAW, R_RAW, L_MONO, R_MONO, L_COLOR, R_COLOR, L_RECT_MONO, R_RECT_MONO, L_RECT_COLOR, R_RECT_COLOR,  # noqa: F401
                   L_DISPARITY, R_DISPARITY, L_DISPARITY_32F, R_DISPARITY_32F, L_DISPARITY_IMG, R_DISPARITY_IMG, L_POINTS2,
                   R_POINTS2, SIDE_L, SIDE_R)

_NP_OF_TYPE = {capi.T_8UC1: (np.uint8, 1), capi.T_8UC3: (np.uint8, 3), capi.T_8UC4: (np.uint8, 4), capi.T_16UC1: (np.uint16, 1),
               capi.T_16SC1: (np.int16, 1), capi.T_32FC1: (np.float32, 1), capi.T_32FC3: (np.float32, 3)}


def _type_of(arr):
    cn = 1 if arr.ndim == 2 else arr.shape[2]
    depth = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 2, np.dtype(np.int16): 3, np.dtype(np.float32): 5}[arr.dtype]
    return depth | ((cn - 1) << 3)


def camera_info(width, height, K, D, R, P, binning=(0, 0), roi=(0, 0, 0, 0)):
    """sensor_msgs/CameraInfo as the C-ABI takes it; binning = (x, y), roi = (x_offset, y_offset, width, height), zeros = none."""
    ci = capi.CameraInfo()
    ci.width, ci.height = int(width), int(height)
    ci.binning_x, ci.binning_y = int(binning[0]), int(binning[1])
    ci.roi_x_offset, ci.roi_y_offset, ci.roi_width, ci.roi_height = (int(v) for v in roi)
    D = list(np.asarray(D, np.float64).reshape(-1))
    ci.nD = len(D)
    for i, v in enumerate(np.asarray(K, np.float64).reshape(9)):
        ci.K[i] = v
    for i, v in enumerate(D):
        ci.D[i] = v
    for i, v in enumerate(np.asarray(R, np.float64).reshape(9)):
        ci.R[i] = v
    for i, v in enumerate(np.asarray(P, np.float64).reshape(12)):
        ci.P[i] = v
    return ci


def read_calibration(path):
    ci = capi.CameraInfo()
    name = C.create_string_buffer(128)
    capi.check(capi.lib().gip_read_calibration(path.encode(), C.byref(ci), name, 128))
    return ci, name.value.decode()


class GpuStereoProcessor:
    def __init__(self, device=-1):
        self._lib = capi.lib()
        self._h = C.c_void_p()
        capi.check(self._lib.gip_create(device, C.byref(self._h)))
        self._keep = []  # keep ctypes callbacks alive while senders are pending

    def close(self):
        if self._h:
            self._lib.gip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, status):
        capi.check(status, self._h)

    @property
    def handle(self):
        return self._h

    # ---- model
    def initStereoModel(self, left, right):
        """(CameraInfo, CameraInfo) or (left_cal_file, right_cal_file)  -- GPUStereoProcessor.cpp:41-61"""
        if isinstance(left, str):
            self._ck(self._lib.gip_init_stereo_model_files(self._h, left.encode(), right.encode()))
        else:
            self._ck(self._lib.gip_init_stereo_model(self._h, C.byref(left), C.byref(right)))

    def isStereoModelInitialised(self):
        return bool(self._lib.gip_is_stereo_model_initialised(self._h))

    def model(self):
        vals = [C.c_double() for _ in range(6)]
        Q = (C.c_double * 16)()
        self._ck(self._lib.gip_model_get(self._h, *[C.byref(v) for v in vals], Q))
        fx, fy, cx_l, cy, cx_r, baseline = [v.value for v in vals]
        return dict(fx=fx, fy=fy, cx_l=cx_l, cy=cy, cx_r=cx_r, baseline=baseline, Q=np.array(Q[:]).reshape(4, 4))

    def rectifyMaps(self, side):
        w, h = C.c_int(), C.c_int()
        self._ck(self._lib.gip_model_size(self._h, C.byref(w), C.byref(h)))
        rows, cols = h.value, w.value
        m1 = np.empty((rows, cols, 2), np.int16)
        m2 = np.empty((rows, cols), np.uint16)
        self._ck(self._lib.gip_get_rectify_maps(self._h, side, m1.ctypes.data, m2.ctypes.data))
        return m1, m2

    # ---- buffers
    def uploadMat(self, mat_source, mat, encoding=""):
        a = np.asarray(mat)
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        rows, cols = a.shape[:2]
        self._ck(self._lib.gip_upload_mat(self._h, mat_source, a.ctypes.data, rows, cols, a.strides[0], _type_of(a), (encoding or "").encode()))

    def matInfo(self, mat_source):
        r, c, t = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._lib.gip_mat_info(self._h, mat_source, C.byref(r), C.byref(c), C.byref(t)))
        return r.value, c.value, t.value

    def downloadMat(self, mat_source):
        rows, cols, t = self.matInfo(mat_source)
        dt, cn = _NP_OF_TYPE[t]
        out = np.empty((rows, cols) if cn == 1 else (rows, cols, cn), dt)
        self._ck(self._lib.gip_download_mat(self._h, mat_source, out.ctypes.data, out.strides[0]))
        return out

    # ---- stages
    def convertRawToColor(self, side):
        self._ck(self._lib.gip_convert_raw_to_color(self._h, side))

    def convertRawToMono(self, side):
        self._ck(self._lib.gip_convert_raw_to_mono(self._h, side))

    def convertColor(self, mat_source, mat_dst, src_encoding, dst_encoding):
        self._ck(self._lib.gip_convert_color(self._h, mat_source, mat_dst, src_encoding.encode(), dst_encoding.encode()))

    def rectifyImage(self, source, dest, interpolation=capi.INTER_LINEAR):
        self._ck(self._lib.gip_rectify_image(self._h, source, dest, interpolation))

    def computeDisparity(self, left, right, disparity):
        self._ck(self._lib.gip_compute_disparity(self._h, left, right, disparity))

    def filterSpeckles(self, disparity_src):
        self._ck(self._lib.gip_filter_speckles(self._h, disparity_src))

    def computeDisparityImage(self, disparity_src, disp_image_dest):
        self._ck(self._lib.gip_compute_disparity_image(self._h, disparity_src, disp_image_dest))

    def projectDisparityTo3DPoints(self, disparity_src, points_src):
        self._ck(self._lib.gip_project_disparity_to_3d_points(self._h, disparity_src, points_src))

    # ---- senders (publisher == callback; None like the reference tests' NULL publisher)
    @staticmethod
    def _header(seq=0, sec=0, nsec=0, frame_id=""):
        h = capi.Header()
        h.seq, h.stamp_sec, h.stamp_nsec, h.frame_id = seq, sec, nsec, frame_id.encode()
        return h

    def enqueueSendImage(self, source, header=None, encoding="mono8", pub=None):
        cb = capi.IMAGE_CB(pub) if pub else C.cast(None, capi.IMAGE_CB)
        self._keep.append(cb)
        sid = C.c_int(-1)
        hdr = header or self._header()
        self._ck(self._lib.gip_enqueue_send_image(self._h, source, C.byref(hdr), encoding.encode(), cb, None, C.byref(sid)))
        return sid.value

    def enqueueSendDisparity(self, source, header=None, pub=None):
        cb = capi.DISPARITY_CB(pub) if pub else C.cast(None, capi.DISPARITY_CB)
        self._keep.append(cb)
        sid = C.c_int(-1)
        hdr = header or self._header()
        self._ck(self._lib.gip_enqueue_send_disparity(self._h, source, C.byref(hdr), cb, None, C.byref(sid)))
        return sid.value

    def enqueueSendPoints(self, points_source, color_source, header=None, pub=None):
        cb = capi.POINTCLOUD2_CB(pub) if pub else C.cast(None, capi.POINTCLOUD2_CB)
        self._keep.append(cb)
        sid = C.c_int(-1)
        hdr = header or self._header()
        self._ck(self._lib.gip_enqueue_send_points(self._h, points_source, color_source, C.byref(hdr), cb, None, C.byref(sid)))
        return sid.value

    def wasDataSent(self, sender_id):
        return bool(self._lib.gip_sender_was_sent(self._h, sender_id))

    def imageMessage(self, sender_id):
        m = self._lib.gip_sender_image_msg(self._h, sender_id).contents
        data = np.ctypeslib.as_array(m.data, shape=(m.data_size,)).copy()
        return dict(height=m.height, width=m.width, encoding=m.encoding.decode(), step=m.step, is_bigendian=m.is_bigendian, data=data,
                    frame_id=m.header.frame_id.decode(), seq=m.header.seq)

    def disparityMessage(self, sender_id):
        m = self._lib.gip_sender_disparity_msg(self._h, sender_id).contents
        img = m.image
        data = np.ctypeslib.as_array(img.data, shape=(img.data_size,)).copy()
        return dict(height=img.height, width=img.width, encoding=img.encoding.decode(), step=img.step,
                    image=data.view(np.float32).reshape(img.height, img.width), f=m.f, T=m.T,
                    valid_window=(m.valid_x_offset, m.valid_y_offset, m.valid_width, m.valid_height), min_disparity=m.min_disparity,
                    max_disparity=m.max_disparity, delta_d=m.delta_d)

    def pointCloudMessage(self, sender_id):
        m = self._lib.gip_sender_pointcloud2_msg(self._h, sender_id).contents
        data = np.ctypeslib.as_array(m.data, shape=(m.data_size,)).copy()
        fields = [(m.fields[i].name.decode(), m.fields[i].offset, m.fields[i].datatype, m.fields[i].count) for i in range(m.nfields)]
        return dict(height=m.height, width=m.width, fields=fields, point_step=m.point_step, row_step=m.row_step, is_dense=m.is_dense,
                    is_bigendian=m.is_bigendian, data=data.reshape(m.height, m.width, 32))

    def cleanSenders(self):
        self._ck(self._lib.gip_clean_senders(self._h))
        self._keep.clear()

    # ---- sync
    def waitForStream(self, stream_source):
        self._ck(self._lib.gip_wait_for_stream(self._h, stream_source))

    def waitForAllStreams(self):
        self._ck(self._lib.gip_wait_for_all_streams(self._h))

    # ---- setters (GPUStereoProcessor.cpp:389-419, :202-208)
    def setPreFilterType(self, v): self._ck(self._lib.gip_set_pre_filter_type(self._h, int(v)))
    def setRefineDisparity(self, v): self._ck(self._lib.gip_set_refine_disparity(self._h, int(v)))
    def setBlockSize(self, v): self._ck(self._lib.gip_set_block_size(self._h, int(v)))
    def setNumDisparities(self, v): self._ck(self._lib.gip_set_num_disparities(self._h, int(v)))
    def setMinDisparity(self, v): self._ck(self._lib.gip_set_min_disparity(self._h, int(v)))
    def setTextureThreshold(self, v): self._ck(self._lib.gip_set_texture_threshold(self._h, int(v)))
    def setMaxSpeckleSize(self, v): self._ck(self._lib.gip_set_max_speckle_size(self._h, int(v)))
    def getMaxSpeckleSize(self): return self._lib.gip_get_max_speckle_size(self._h)
    def setMaxSpeckleDiff(self, v): self._ck(self._lib.gip_set_max_speckle_diff(self._h, float(v)))
    def getMaxSpeckleDiff(self): return self._lib.gip_get_max_speckle_diff(self._h)
    def setPreFilterSize(self, v): self._ck(self._lib.gip_set_pre_filter_size(self._h, int(v)))
    def setPreFilterCap(self, v): self._ck(self._lib.gip_set_pre_filter_cap(self._h, int(v)))
    def setUniquenessRatio(self, v): self._ck(self._lib.gip_set_uniqueness_ratio(self._h, int(v)))
    def setDisp12MaxDiff(self, v): self._ck(self._lib.gip_set_disp12_max_diff(self._h, int(v)))

    def getParams(self):
        p = capi.BMParams()
        self._ck(self._lib.gip_get_params(self._h, C.byref(p)))
        return p

    def setParams(self, p):
        self._ck(self._lib.gip_set_params(self._h, C.byref(p)))

    # ---- the imageCb schedule (StereoProcessor.cpp:157-298)
    def processPair(self, left_raw, right_raw, encoding, connected_topics):
        l = np.ascontiguousarray(left_raw)
        r = np.ascontiguousarray(right_raw)
        rows, cols = l.shape[:2]
        self._ck(self._lib.gip_process_pair(self._h, l.ctypes.data, r.ctypes.data, rows, cols, l.strides[0], _type_of(l), encoding.encode(),
                                            int(connected_topics)))
