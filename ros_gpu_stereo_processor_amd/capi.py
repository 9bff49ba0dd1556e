"""ctypes binding of libgpuimageproc_hip.so (include/gpuimageproc_hip.h).

The library is the product; this module only loads it and declares prototypes.  There is NO CPU fallback: if the
library is missing or cannot be loaded, importing callers get a RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GIP_LIB_PATH: an alternative build of the same library (A/B experiments: tools/ab_*.sh); default = the in-tree build
LIB_PATH = os.environ.get("GIP_LIB_PATH") or os.path.join(_HERE, "lib", "libgpuimageproc_hip.so")

# status codes
OK, ERR_INVALID_ARG, ERR_NOT_INITIALISED, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_EMPTY_SLOT, ERR_IO = range(8)

# GpuMatSource values (reference include/gpuimageproc/GPUStereoProcessor.h:21-57)
SIDE_L, SIDE_R = 1, 2
SIDE_MASK = 3
SRC_RAW, SRC_MONO, SRC_COLOR, SRC_RECT_MONO, SRC_RECT_COLOR = 4, 8, 16, 32, 64
SRC_DISPARITY, SRC_DISPARITY_32F, SRC_DISPARITY_IMG, SRC_POINTS2 = 128, 256, 512, 1024
L_RAW, R_RAW = SRC_RAW | SIDE_L, SRC_RAW | SIDE_R
L_MONO, R_MONO = SRC_MONO | SIDE_L, SRC_MONO | SIDE_R
L_COLOR, R_COLOR = SRC_COLOR | SIDE_L, SRC_COLOR | SIDE_R
L_RECT_MONO, R_RECT_MONO = SRC_RECT_MONO | SIDE_L, SRC_RECT_MONO | SIDE_R
L_RECT_COLOR, R_RECT_COLOR = SRC_RECT_COLOR | SIDE_L, SRC_RECT_COLOR | SIDE_R
L_DISPARITY, R_DISPARITY = SRC_DISPARITY | SIDE_L, SRC_DISPARITY | SIDE_R
L_DISPARITY_32F, R_DISPARITY_32F = SRC_DISPARITY_32F | SIDE_L, SRC_DISPARITY_32F | SIDE_R
L_DISPARITY_IMG, R_DISPARITY_IMG = SRC_DISPARITY_IMG | SIDE_L, SRC_DISPARITY_IMG | SIDE_R
L_POINTS2, R_POINTS2 = SRC_POINTS2 | SIDE_L, SRC_POINTS2 | SIDE_R

# element types (OpenCV numbering)
T_8UC1, T_8UC3, T_8UC4, T_16UC1, T_16SC1, T_32FC1, T_32FC3 = 0, 16, 24, 2, 3, 5, 21

PREFILTER_NORMALIZED_RESPONSE, PREFILTER_XSOBEL = 0, 1
INTER_NEAREST, INTER_LINEAR = 0, 1
BAYER = {"bayer_rggb8": 0, "bayer_bggr8": 1, "bayer_gbrg8": 2, "bayer_grbg8": 3}

TOPIC_DEBAYER_MONO_LEFT, TOPIC_DEBAYER_MONO_RIGHT = 1 << 0, 1 << 1
TOPIC_DEBAYER_COLOR_LEFT, TOPIC_DEBAYER_COLOR_RIGHT = 1 << 2, 1 << 3
TOPIC_RECTIFY_MONO_LEFT, TOPIC_RECTIFY_MONO_RIGHT = 1 << 4, 1 << 5
TOPIC_RECTIFY_COLOR_LEFT, TOPIC_RECTIFY_COLOR_RIGHT = 1 << 6, 1 << 7
TOPIC_DISPARITY, TOPIC_DISPARITY_VIS, TOPIC_POINTCLOUD = 1 << 8, 1 << 9, 1 << 10
BATCH_WHOLE, BATCH_FRONT, BATCH_BACK = 0, 1, 2    # gip_batch_desc::phase


class BMParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "prefilter_type", "prefilter_size", "prefilter_cap", "block_size", "min_disparity", "num_disparities",
        "texture_threshold", "uniqueness_ratio", "disp12_max_diff", "speckle_window_size", "speckle_range")]


class CameraInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("K", C.c_double * 9), ("D", C.c_double * 8), ("nD", C.c_int32),
                ("R", C.c_double * 9), ("P", C.c_double * 12), ("binning_x", C.c_uint32), ("binning_y", C.c_uint32), ("roi_x_offset", C.c_uint32),
                ("roi_y_offset", C.c_uint32), ("roi_height", C.c_uint32), ("roi_width", C.c_uint32)]


class Header(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id", C.c_char * 64)]


class ImageMsg(C.Structure):
    _fields_ = [("header", Header), ("height", C.c_uint32), ("width", C.c_uint32), ("encoding", C.c_char * 16),
                ("is_bigendian", C.c_uint8), ("step", C.c_uint32), ("data", C.POINTER(C.c_uint8)), ("data_size", C.c_size_t)]


class DisparityMsg(C.Structure):
    _fields_ = [("header", Header), ("image", ImageMsg), ("f", C.c_float), ("T", C.c_float),
                ("valid_x_offset", C.c_uint32), ("valid_y_offset", C.c_uint32), ("valid_width", C.c_uint32), ("valid_height", C.c_uint32),
                ("min_disparity", C.c_float), ("max_disparity", C.c_float), ("delta_d", C.c_float)]


class PointField(C.Structure):
    _fields_ = [("name", C.c_char * 8), ("offset", C.c_uint32), ("datatype", C.c_uint8), ("count", C.c_uint32)]


class PointCloud2Msg(C.Structure):
    _fields_ = [("header", Header), ("height", C.c_uint32), ("width", C.c_uint32), ("fields", PointField * 4), ("nfields", C.c_uint32),
                ("is_bigendian", C.c_uint8), ("point_step", C.c_uint32), ("row_step", C.c_uint32), ("data", C.POINTER(C.c_uint8)),
                ("data_size", C.c_size_t), ("is_dense", C.c_uint8)]


class BatchDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("batch", C.c_int32), ("encoding", C.c_int32), ("outputs", C.c_uint32),
                ("left_raw", C.c_void_p), ("right_raw", C.c_void_p), ("disparity", C.c_void_p), ("disparity_f32", C.c_void_p),
                ("pointcloud", C.c_void_p), ("matcher_stream", C.c_void_p), ("fused_front_end", C.c_int32), ("batches_in_flight", C.c_int32), ("phase", C.c_int32)]


class StreamerConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("encoding", C.c_int32), ("outputs", C.c_uint32), ("want_disparity16", C.c_int32),
                ("depth", C.c_int32), ("frames_per_batch", C.c_int32), ("inputs_pinned", C.c_int32), ("batching", C.c_int32), ("compute_lanes", C.c_int32)]


class StreamerResult(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("header", Header), ("width", C.c_int32), ("height", C.c_int32), ("disparity16", C.c_void_p),
                ("disparity_f32", C.c_void_p), ("pointcloud", C.c_void_p)]


class RowSplitConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("encoding", C.c_int32), ("outputs", C.c_uint32), ("n_tiles", C.c_int32),
                ("devices", C.c_int32 * 16), ("halo_mode", C.c_int32), ("host_threads", C.c_int32), ("gather_mode", C.c_int32)]


class RowSplitTiming(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("front_ms", "halo_ms", "match_ms", "gather_ms", "post_ms", "total_ms")]


STREAMER_CB = C.CFUNCTYPE(None, C.POINTER(StreamerResult), C.c_void_p)
IMAGE_CB = C.CFUNCTYPE(None, C.POINTER(ImageMsg), C.c_void_p)
DISPARITY_CB = C.CFUNCTYPE(None, C.POINTER(DisparityMsg), C.c_void_p)
POINTCLOUD2_CB = C.CFUNCTYPE(None, C.POINTER(PointCloud2Msg), C.c_void_p)

_vp, _sz, _i, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_double

# name -> (restype, argtypes); every symbol include/gpuimageproc_hip.h declares
PROTOTYPES = {
    "gip_create": (_i, [_i, C.POINTER(_vp)]),
    "gip_destroy": (_i, [_vp]),
    "gip_last_error": (C.c_char_p, [_vp]),
    "gip_device": (_i, [_vp]),
    "gip_status_string": (C.c_char_p, [_i]),
    "gip_device_count": (_i, []),
    "gip_concurrent_streams_create": (_i, [_i, _i, C.POINTER(_vp), C.POINTER(_i)]),
    "gip_concurrent_streams_destroy": (_i, [_i, _i, C.POINTER(_vp)]),
    "gip_init_stereo_model": (_i, [_vp, C.POINTER(CameraInfo), C.POINTER(CameraInfo)]),
    "gip_init_stereo_model_files": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "gip_is_stereo_model_initialised": (_i, [_vp]),
    "gip_read_calibration": (_i, [C.c_char_p, C.POINTER(CameraInfo), C.c_char_p, _sz]),
    "gip_build_rectify_maps": (_i, [C.POINTER(CameraInfo), _vp, _vp]),
    "gip_model_size": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "gip_model_get": (_i, [_vp] + [C.POINTER(_d)] * 6 + [C.POINTER(_d)]),
    "gip_get_rectify_maps": (_i, [_vp, _i, _vp, _vp]),
    "gip_upload_mat": (_i, [_vp, _i, _vp, _i, _i, _sz, _i, C.c_char_p]),
    "gip_download_mat": (_i, [_vp, _i, _vp, _sz]),
    "gip_mat_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "gip_mat_device_ptr": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "gip_convert_raw_to_color": (_i, [_vp, _i]),
    "gip_convert_raw_to_mono": (_i, [_vp, _i]),
    "gip_convert_color": (_i, [_vp, _i, _i, C.c_char_p, C.c_char_p]),
    "gip_rectify_image": (_i, [_vp, _i, _i, _i]),
    "gip_compute_disparity": (_i, [_vp, _i, _i, _i]),
    "gip_filter_speckles": (_i, [_vp, _i]),
    "gip_compute_disparity_image": (_i, [_vp, _i, _i]),
    "gip_project_disparity_to_3d_points": (_i, [_vp, _i, _i]),
    "gip_enqueue_send_image": (_i, [_vp, _i, C.POINTER(Header), C.c_char_p, IMAGE_CB, _vp, C.POINTER(_i)]),
    "gip_enqueue_send_disparity": (_i, [_vp, _i, C.POINTER(Header), DISPARITY_CB, _vp, C.POINTER(_i)]),
    "gip_enqueue_send_points": (_i, [_vp, _i, _i, C.POINTER(Header), POINTCLOUD2_CB, _vp, C.POINTER(_i)]),
    "gip_disparity_info": (_i, [_vp, _i, _i, C.POINTER(DisparityMsg)]),
    "gip_sender_was_sent": (_i, [_vp, _i]),
    "gip_sender_image_msg": (C.POINTER(ImageMsg), [_vp, _i]),
    "gip_sender_disparity_msg": (C.POINTER(DisparityMsg), [_vp, _i]),
    "gip_sender_pointcloud2_msg": (C.POINTER(PointCloud2Msg), [_vp, _i]),
    "gip_clean_senders": (_i, [_vp]),
    "gip_wait_for_stream": (_i, [_vp, _i]),
    "gip_wait_for_all_streams": (_i, [_vp]),
    "gip_set_pre_filter_type": (_i, [_vp, _i]),
    "gip_set_refine_disparity": (_i, [_vp, _i]),
    "gip_set_block_size": (_i, [_vp, _i]),
    "gip_set_num_disparities": (_i, [_vp, _i]),
    "gip_set_min_disparity": (_i, [_vp, _i]),
    "gip_set_texture_threshold": (_i, [_vp, _i]),
    "gip_set_max_speckle_size": (_i, [_vp, _i]),
    "gip_get_max_speckle_size": (_i, [_vp]),
    "gip_set_max_speckle_diff": (_i, [_vp, _d]),
    "gip_get_max_speckle_diff": (_d, [_vp]),
    "gip_set_pre_filter_size": (_i, [_vp, _i]),
    "gip_set_pre_filter_cap": (_i, [_vp, _i]),
    "gip_set_uniqueness_ratio": (_i, [_vp, _i]),
    "gip_set_disp12_max_diff": (_i, [_vp, _i]),
    "gip_get_params": (_i, [_vp, C.POINTER(BMParams)]),
    "gip_set_params": (_i, [_vp, C.POINTER(BMParams)]),
    "gip_process_pair": (_i, [_vp, _vp, _vp, _i, _i, _sz, _i, C.c_char_p, C.c_uint32]),
    "gipk_cvt_bgr_to_gray": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_gray_to_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_swap_rb": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_bayer_to_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_bayer_to_gray": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_mono8_to_mono16": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_cvt_mono16_to_mono8": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_remap_u8": (_i, [_vp, _sz, _i, _i, _vp, _sz, _i, _i, _i, _vp, _vp, _i, _sz, _sz, _vp]),
    "gipk_remap_u8_interp": (_i, [_vp, _sz, _i, _i, _vp, _sz, _i, _i, _i, _vp, _vp, _i, _i, _sz, _sz, _vp]),
    "gipk_prefilter": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _i, _i, _i, _sz, _sz, _vp]),
    "gipk_rectify_xsobel": (_i, [_vp, _sz, _i, _i, _vp, _sz, _vp, _sz, _i, _i, _vp, _vp, _i, _i, _sz, _sz, _sz, _vp]),
    "gipk_bm_match": (_i, [_vp, _vp, _sz, _i, _i, C.POINTER(BMParams), _vp, _sz, _vp, _sz, _i, _sz, _sz, _sz, _vp]),
    "gipk_bm_finalize": (_i, [_vp, _sz, _vp, _sz, _i, _i, C.POINTER(BMParams), _i, _sz, _sz, _vp]),
    "gipk_stereobm_scratch_bytes": (_sz, [_i, _i, _i, C.POINTER(BMParams)]),
    "gipk_stereobm": (_i, [_vp, _vp, _sz, _i, _i, C.POINTER(BMParams), _vp, _sz, _vp, _sz, _i, _sz, _sz, _vp]),
    "gipk_disparity_to_f32": (_i, [_vp, _sz, _vp, _sz, _i, _i, _d, _d, _i, _sz, _sz, _vp]),
    "gipk_speckle_scratch_bytes": (_sz, [_i, _i, _i]),
    "gipk_filter_speckles": (_i, [_vp, _sz, _i, _i, _i, _i, _i, _vp, _sz, _i, _sz, _vp]),
    "gipk_speckle_label": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _sz, _i, _sz, _vp]),
    "gipk_speckle_row_labels": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gipk_speckle_set_sizes": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp, _i, _vp]),
    "gipk_speckle_apply": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _sz, _i, _sz, _vp]),
    "gipk_reproject_to_3d": (_i, [_vp, _sz, _vp, _sz, _i, _i, C.POINTER(_d), C.c_float, _i, _i, _sz, _sz, _vp]),
    "gipk_pack_pointcloud2": (_i, [_vp, _sz, _vp, _sz, _vp, _i, _i, _i, _sz, _sz, _sz, _vp]),
    "gipk_disparity_to_pointcloud2": (_i, [_vp, _sz, _vp, _sz, _vp, _i, _i, C.POINTER(_d), _d, _d, _i, _i, _sz, _sz, _sz, _vp]),
    "gipk_disparity_outputs": (_i, [_vp, _sz, _vp, _sz, _i, _vp, _vp, _i, _i, C.POINTER(_d), _d, _d, _i, _i, _sz, _sz, _vp]),
    "gipk_draw_color_disp": (_i, [_vp, _sz, _vp, _sz, _i, _i, _i, _i, _sz, _sz, _vp]),
    "gip_batch_set_maps_device": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "gip_batch_run": (_i, [_vp, C.POINTER(BatchDesc), _vp]),
    "gip_batch_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gip_batch_timing_reset": (_i, [_vp]),
    "gip_ros1_serialize_image": (C.c_int64, [C.POINTER(ImageMsg), _vp, _sz]),
    "gip_ros1_serialize_disparity": (C.c_int64, [C.POINTER(DisparityMsg), _vp, _sz]),
    "gip_ros1_serialize_pointcloud2": (C.c_int64, [C.POINTER(PointCloud2Msg), _vp, _sz]),
    "gip_streamer_open": (_i, [_vp, C.POINTER(StreamerConfig), STREAMER_CB, _vp, C.POINTER(_vp)]),
    "gip_streamer_submit": (_i, [_vp, _vp, _vp, _sz, C.POINTER(Header)]),
    "gip_streamer_flush": (_i, [_vp]),
    "gip_streamer_delivered": (C.c_uint64, [_vp]),
    "gip_streamer_launches": (C.c_uint64, [_vp]),
    "gip_streamer_close": (_i, [_vp]),
    "gip_streamer_last_error": (C.c_char_p, [_vp]),
    "gipk_disparity_outputs_rows": (_i, [_vp, _sz, _vp, _sz, _i, _vp, _vp, _i, _i, _i, C.POINTER(_d), _d, _d, _i, _i, _sz, _sz, _vp]),
    "gipk_speckle_border_record_bytes": (_sz, [_i]),
    "gipk_speckle_border_record": (_i, [_vp, _sz, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "gipk_speckle_border_gather": (_i, [_vp, _i, _i, _vp, _vp]),
    "gipk_speckle_border_merge": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "gip_rowsplit_plan_rows": (_i, [_i, _i, C.POINTER(BMParams), _i] + [C.POINTER(_i)] * 5),
    "gip_rowsplit_open": (_i, [C.POINTER(CameraInfo), C.POINTER(CameraInfo), C.POINTER(BMParams), _i, _d, C.POINTER(RowSplitConfig), C.POINTER(_vp)]),
    "gip_rowsplit_close": (_i, [_vp]),
    "gip_rowsplit_last_error": (C.c_char_p, [_vp]),
    "gip_rowsplit_plan": (_i, [_vp, _i] + [C.POINTER(_i)] * 5),
    "gip_rowsplit_set_inputs_device": (_i, [_vp, _i, _vp, _vp]),
    "gip_rowsplit_upload": (_i, [_vp, _vp, _vp, _sz]),
    "gip_rowsplit_run": (_i, [_vp]),
    "gip_rowsplit_wait": (_i, [_vp]),
    "gip_rowsplit_run_serial": (_i, [_vp, C.POINTER(RowSplitTiming)]),
    "gip_rowsplit_last_timing": (_i, [_vp, _i, C.POINTER(RowSplitTiming)]),
    "gip_rowsplit_outputs_device": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "gip_rowsplit_download": (_i, [_vp, _vp, _vp, _vp]),
    "gip_batch_timing_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i)]),
    "gip_batch_timing_read_spans": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d), C.POINTER(_i)]),
}

_LIB = None


def lib():
    """Load the HIP library (building nothing: use ros_gpu_stereo_processor_amd.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgpuimageproc_hip.so is missing (%s): run `python -m ros_gpu_stereo_processor_amd.build`; "
                               "there is no CPU fallback" % LIB_PATH)
        # torch first: its wheel bundles a HIP runtime (torch/lib/libamdhip64.so) with the same SONAME as /opt/rocm's.  Whichever
        # is loaded first serves the whole process; loading ours before torch left this library without a visible device on
        # the GPU box (gip_create -> GIP_ERR_NO_DEVICE) while torch itself still saw the GPU.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class GipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gpuimageproc_hip status %d (%s): %s" % (status, lib().gip_status_string(status).decode(), message))
        self.status = status


def check(status, handle=None):
    if status != OK:
        msg = lib().gip_last_error(handle).decode() if handle else ""
        raise GipError(status, msg)


def concurrent_streams(n, device=0):
    """n HIP stream handles (ints) of `device` that dispatch side by side (gip_concurrent_streams_create), and how many of them are
    mutually independent.  Wrap them with torch.cuda.ExternalStream to use them from torch; release with release_streams()."""
    arr = (C.c_void_p * n)()
    k = C.c_int(0)
    rc = lib().gip_concurrent_streams_create(device, n, arr, C.byref(k))
    if rc != 0:
        raise GipError(rc, "gip_concurrent_streams_create")
    return [int(arr[i]) for i in range(n)], k.value


def release_streams(handles, device=0):
    arr = (C.c_void_p * len(handles))(*handles)
    lib().gip_concurrent_streams_destroy(device, len(handles), arr)
