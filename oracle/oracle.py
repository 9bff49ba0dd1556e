"""ctypes front-end of the CPU oracle (oracle/gip_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
never by the product package (ros_gpu_stereo_processor_amd), which fails loudly without its HIP library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PREFILTER_NORMALIZED_RESPONSE = 0
PREFILTER_XSOBEL = 1
BAYER = {"bayer_rggb8": 0, "bayer_bggr8": 1, "bayer_gbrg8": 2, "bayer_grbg8": 3}


class BMParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "prefilter_type", "prefilter_size", "prefilter_cap", "block_size", "min_disparity", "num_disparities",
        "texture_threshold", "uniqueness_ratio", "disp12_max_diff", "speckle_window_size", "speckle_range")]


def bm_params(prefilter_type=PREFILTER_XSOBEL, prefilter_size=9, prefilter_cap=31, block_size=21, min_disparity=0,
              num_disparities=64, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=-1,
              speckle_window_size=0, speckle_range=0):
    """cv::StereoBM defaults (StereoBMParams) unless overridden."""
    return BMParams(prefilter_type, prefilter_size, prefilter_cap, block_size, min_disparity, num_disparities,
                    texture_threshold, uniqueness_ratio, disp12_max_diff, speckle_window_size, speckle_range)


def build(force=False):
    so = os.path.join(_HERE, "libgip_oracle.so")
    src = os.path.join(_HERE, "gip_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.gipo_stereobm_compute.restype = C.c_int
        _LIB.gipo_bm_match.restype = C.c_int
        _LIB.gipo_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def num_threads():
    return lib().gipo_num_threads()


# ---------------------------------------------------------------- colour
def bgr_to_gray(img, src_is_rgb=False):
    img = _u8(img)
    h, w, _ = img.shape
    out = np.empty((h, w), np.uint8)
    lib().gipo_bgr_to_gray(_p(img), C.c_int(w * 3), _p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(int(src_is_rgb)))
    return out


def gray_to_bgr(img):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().gipo_gray_to_bgr(_p(img), C.c_int(w), _p(out), C.c_int(w * 3), C.c_int(w), C.c_int(h))
    return out


def swap_rb(img):
    img = _u8(img)
    h, w, _ = img.shape
    out = np.empty_like(img)
    lib().gipo_swap_rb(_p(img), C.c_int(w * 3), _p(out), C.c_int(w * 3), C.c_int(w), C.c_int(h))
    return out


def bayer_to_bgr(img, encoding):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().gipo_bayer_to_bgr(_p(img), C.c_int(w), _p(out), C.c_int(w * 3), C.c_int(w), C.c_int(h), C.c_int(BAYER[encoding]))
    return out


def bayer_to_gray(img, encoding):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().gipo_bayer_to_gray(_p(img), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(BAYER[encoding]))
    return out


def mono8_to_mono16(img):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint16)
    lib().gipo_mono8_to_mono16(_p(img), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h))
    return out


def mono16_to_mono8(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().gipo_mono16_to_mono8(_p(img), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h))
    return out


def convert_color(img, src_encoding, dst_encoding):
    """cv_bridge::getConversionCode dispatch of GpuStereoProcessor::convertColor (GPUStereoProcessor.cpp:119-172)."""
    if src_encoding == dst_encoding:
        return np.array(img, copy=True)
    if src_encoding in BAYER:
        if dst_encoding == "mono8":
            return bayer_to_gray(img, src_encoding)
        if dst_encoding == "bgr8":
            return bayer_to_bgr(img, src_encoding)
    if src_encoding == "bgr8" and dst_encoding == "mono8":
        return bgr_to_gray(img, False)
    if src_encoding == "rgb8" and dst_encoding == "mono8":
        return bgr_to_gray(img, True)
    if src_encoding == "rgb8" and dst_encoding == "bgr8":
        return swap_rb(img)
    if src_encoding == "mono8" and dst_encoding == "bgr8":
        return gray_to_bgr(img)
    if src_encoding == "mono8" and dst_encoding == "mono16":
        return mono8_to_mono16(img)
    if src_encoding == "mono16" and dst_encoding == "mono8":
        return mono16_to_mono8(img)
    raise ValueError("unsupported conversion %s -> %s" % (src_encoding, dst_encoding))


# ---------------------------------------------------------------- rectification
def init_undistort_rectify_map(K, D, R, P, w, h):
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(-1)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(P, np.float64).reshape(12)
    map1 = np.empty((h, w, 2), np.int16)
    map2 = np.empty((h, w), np.uint16)
    lib().gipo_init_undistort_rectify_map(_p(K), _p(D), C.c_int(D.size), _p(R), _p(P), C.c_int(w), C.c_int(h), _p(map1), _p(map2))
    return map1, map2


def remap(src, map1, map2, nearest=False):
    src = _u8(src)
    cn = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    h, w = map2.shape
    out = np.empty((h, w) if cn == 1 else (h, w, cn), np.uint8)
    fn = lib().gipo_remap_nearest_u8 if nearest else lib().gipo_remap_bilinear_u8
    fn(_p(src), C.c_int(sw * cn), C.c_int(sw), C.c_int(sh), _p(out), C.c_int(w * cn), C.c_int(w),
                                 C.c_int(h), C.c_int(cn), _p(np.ascontiguousarray(map1)), _p(np.ascontiguousarray(map2)))
    return out


# ---------------------------------------------------------------- StereoBM
def prefilter_xsobel(img, cap=31):
    img = _u8(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().gipo_prefilter_xsobel(_p(img), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(cap))
    return out


def prefilter_norm(img, winsize=9, cap=31):
    img = _u8(img)
    h, w = img.shape
    out = np.empty_like(img)
    lib().gipo_prefilter_norm(_p(img), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(winsize), C.c_int(cap))
    return out


def bm_match(left_f, right_f, params, nthreads=0):
    """Matching on prefiltered images; returns (disp16 pre-validate/pre-blank, cost)."""
    left_f, right_f = _u8(left_f), _u8(right_f)
    h, w = left_f.shape
    disp = np.empty((h, w), np.int16)
    cost = np.empty((h, w), np.int32)
    lib().gipo_bm_match(_p(left_f), _p(right_f), C.c_int(w), C.c_int(w), C.c_int(h), C.byref(params), _p(disp), C.c_int(w),
                        _p(cost), C.c_int(w), C.c_int(nthreads))
    return disp, cost


def valid_disparity_roi(w, h, params):
    roi = (C.c_int * 4)()
    lib().gipo_valid_disparity_roi(C.c_int(w), C.c_int(h), C.c_int(params.min_disparity), C.c_int(params.num_disparities),
                                   C.c_int(params.block_size), roi)
    return tuple(roi)


def validate_disparity(disp, cost, params, row0, row1):
    disp = np.ascontiguousarray(disp, np.int16).copy()
    cost = np.ascontiguousarray(cost, np.int32)
    h, w = disp.shape
    lib().gipo_validate_disparity(_p(disp), C.c_int(w), _p(cost), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(row0),
                                  C.c_int(row1), C.c_int(params.min_disparity), C.c_int(params.num_disparities),
                                  C.c_int(params.disp12_max_diff))
    return disp


def stereobm(left, right, params, nthreads=0):
    """cv::StereoBM::compute -> int16 x16 disparity."""
    left, right = _u8(left), _u8(right)
    h, w = left.shape
    disp = np.empty((h, w), np.int16)
    rc = lib().gipo_stereobm_compute(_p(left), _p(right), C.c_int(w), C.c_int(w), C.c_int(h), C.byref(params), _p(disp),
                                     C.c_int(w), C.c_int(nthreads))
    if rc != 0:
        raise ValueError("StereoBM parameter validation failed (%d)" % rc)
    return disp


def disparity_to_f32(disp16, cx_l, cx_r):
    disp16 = np.ascontiguousarray(disp16, np.int16)
    h, w = disp16.shape
    out = np.empty((h, w), np.float32)
    lib().gipo_disparity_to_f32(_p(disp16), C.c_int(w), _p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_double(cx_l), C.c_double(cx_r))
    return out


def filter_speckles(disp16, new_val, max_speckle_size, max_diff):
    out = np.ascontiguousarray(disp16, np.int16).copy()
    h, w = out.shape
    lib().gipo_filter_speckles(_p(out), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(int(new_val)), C.c_int(int(max_speckle_size)),
                               C.c_int(int(max_diff)))
    return out


# ---------------------------------------------------------------- reprojection / cloud / vis
def stereo_q(P_left, P_right):
    Pl = np.ascontiguousarray(P_left, np.float64).reshape(12)
    Pr = np.ascontiguousarray(P_right, np.float64).reshape(12)
    Q = np.zeros(16, np.float64)
    lib().gipo_stereo_q(_p(Pl), _p(Pr), _p(Q))
    return Q.reshape(4, 4)


def reproject_to_3d(disp_f32, Q, handle_missing=True):
    d = np.ascontiguousarray(disp_f32, np.float32)
    h, w = d.shape
    out = np.empty((h, w, 3), np.float32)
    Q = np.ascontiguousarray(Q, np.float64).reshape(16)
    lib().gipo_reproject_to_3d(_p(d), C.c_int(w), _p(out), C.c_int(w * 3), C.c_int(w), C.c_int(h), _p(Q), C.c_int(int(handle_missing)))
    return out


def pack_pointcloud2(xyz, bgr):
    xyz = np.ascontiguousarray(xyz, np.float32)
    bgr = _u8(bgr)
    h, w, _ = xyz.shape
    out = np.empty((h, w, 32), np.uint8)
    lib().gipo_pack_pointcloud2(_p(xyz), C.c_int(w * 3), _p(bgr), C.c_int(w * 3), _p(out), C.c_int(w), C.c_int(h))
    return out


def draw_color_disp(disp16, ndisp):
    d = np.ascontiguousarray(disp16, np.int16)
    h, w = d.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().gipo_draw_color_disp(_p(d), C.c_int(w), _p(out), C.c_int(w * 4), C.c_int(w), C.c_int(h), C.c_int(ndisp))
    return out
