"""CPU tests: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/*.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpuimageproc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gipk?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_cb")))


@pytest.fixture(scope="module")
def built_lib():
    from ros_gpu_stereo_processor_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) > 60
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing


def test_python_prototypes_cover_the_header(built_lib):
    from ros_gpu_stereo_processor_amd import capi
    assert sorted(capi.PROTOTYPES) == _declared_symbols()
    capi.lib()  # resolves every prototype against the .so


def test_no_gpu_paths_fail_loudly(built_lib):
    """Without a device the handle cannot be created; nothing falls back to the CPU."""
    import torch
    from ros_gpu_stereo_processor_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert capi.lib().gip_create(-1, ctypes.byref(h)) == capi.ERR_NO_DEVICE
    assert capi.lib().gip_device_count() == 0
    assert capi.lib().gip_status_string(capi.ERR_UNSUPPORTED) == b"unsupported configuration"


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ros_gpu_stereo_processor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gip_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "gip_oracle" not in open(p).read()


def test_calibration_yaml_reader(built_lib, calib_files, calib):
    from ros_gpu_stereo_processor_amd import processor as P
    import numpy as np
    for path, side in zip(calib_files, ("left", "right")):
        ci, name = P.read_calibration(path)
        assert (ci.width, ci.height, ci.nD) == (752, 480, 5)
        assert name == "narrow_stereo/" + side
        assert np.array_equal(np.array(ci.K[:]), calib[side]["K"])
        assert np.array_equal(np.array(ci.D[:5]), calib[side]["D"])
        assert np.array_equal(np.array(ci.R[:]), calib[side]["R"])
        assert np.array_equal(np.array(ci.P[:]), calib[side]["P"])


def test_calibration_ini_reader(built_lib, calib, tmp_path):
    """camera_calibration_parsers' INI layout (readCalibration picks the parser by extension)"""
    from ros_gpu_stereo_processor_amd import processor as P
    import numpy as np
    c = calib["left"]
    fmt = lambda a, n: "\n".join(" ".join("%.6f" % v for v in a[i:i + n]) for i in range(0, len(a), n))  # noqa: E731
    text = ("# oST version 5.0 parameters\n\n\n[image]\n\nwidth\n752\n\nheight\n480\n\n[narrow_stereo/left]\n\ncamera matrix\n%s\n\ndistortion\n%s\n\n"
            "rectification\n%s\n\nprojection\n%s\n\n" % (fmt(c["K"], 3), fmt(c["D"], 5), fmt(c["R"], 3), fmt(c["P"], 4)))
    path = tmp_path / "left.ini"
    path.write_text(text)
    ci, name = P.read_calibration(str(path))
    assert (ci.width, ci.height, ci.nD, name) == (752, 480, 5, "narrow_stereo/left")
    assert np.array_equal(np.array(ci.K[:]), c["K"]) and np.array_equal(np.array(ci.D[:5]), c["D"])
    assert np.array_equal(np.array(ci.R[:]), c["R"]) and np.array_equal(np.array(ci.P[:]), c["P"])


def test_camera_info_binning_and_roi_tables(built_lib, calib):
    """sensor_msgs/CameraInfo binning and ROI as image_geometry applies them (fromCameraInfo / initRectificationMaps): the tables of a
    binned camera are those of the K, P scaled by 1/binning at the binned size; an ROI takes the window of the (binned) full-frame
    tables and subtracts its origin from the integer coordinates.  Host-only entry point, no device needed."""
    import numpy as np
    from oracle import oracle as O
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd.processor import camera_info
    c = calib["left"]
    W, H = c["width"], c["height"]
    L = capi.lib()

    def tables(ci, w, h):
        m1, m2 = np.zeros((h, w, 2), np.int16), np.zeros((h, w), np.uint16)
        capi.check(L.gip_build_rectify_maps(ctypes.byref(ci), m1.ctypes.data, m2.ctypes.data))
        return m1, m2
    K, P = np.asarray(c["K"], float).reshape(3, 3), np.asarray(c["P"], float).reshape(3, 4)
    # binning 2 x 2
    Kb, Pb = K.copy(), P.copy()
    Kb[0, 0] *= 0.5; Kb[0, 2] *= 0.5; Kb[1, 1] *= 0.5; Kb[1, 2] *= 0.5
    Pb[0, 0] *= 0.5; Pb[0, 2] *= 0.5; Pb[0, 3] *= 0.5; Pb[1, 1] *= 0.5; Pb[1, 2] *= 0.5; Pb[1, 3] *= 0.5
    e1, e2 = O.init_undistort_rectify_map(Kb.ravel(), c["D"], c["R"], Pb.ravel(), W // 2, H // 2)
    g1, g2 = tables(camera_info(W, H, c["K"], c["D"], c["R"], c["P"], binning=(2, 2)), W // 2, H // 2)
    assert np.array_equal(g1, e1) and np.array_equal(g2, e2)
    # ROI on the full-resolution camera
    f1, f2 = O.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], W, H)
    rx, ry, rw, rh = 40, 24, 320, 200
    g1, g2 = tables(camera_info(W, H, c["K"], c["D"], c["R"], c["P"], roi=(rx, ry, rw, rh)), rw, rh)
    assert np.array_equal(g1, f1[ry:ry + rh, rx:rx + rw] - np.array([rx, ry], np.int16)) and np.array_equal(g2, f2[ry:ry + rh, rx:rx + rw])
    # ROI + binning: ROI given in full-resolution pixels, window taken on the binned tables
    g1, g2 = tables(camera_info(W, H, c["K"], c["D"], c["R"], c["P"], binning=(2, 2), roi=(rx, ry, rw, rh)), rw // 2, rh // 2)
    assert np.array_equal(g1, e1[ry // 2:(ry + rh) // 2, rx // 2:(rx + rw) // 2] - np.array([rx // 2, ry // 2], np.int16))
    assert np.array_equal(g2, e2[ry // 2:(ry + rh) // 2, rx // 2:(rx + rw) // 2])
    # full-frame ROI given explicitly = no ROI; an ROI outside the frame is rejected
    g1, g2 = tables(camera_info(W, H, c["K"], c["D"], c["R"], c["P"], roi=(0, 0, W, H)), W, H)
    assert np.array_equal(g1, f1) and np.array_equal(g2, f2)
    bad = camera_info(W, H, c["K"], c["D"], c["R"], c["P"], roi=(W - 10, 0, 64, 64))
    assert L.gip_build_rectify_maps(ctypes.byref(bad), g1.ctypes.data, g2.ctypes.data) == capi.ERR_INVALID_ARG


def test_uniqueness_threshold_multiply_shift_identity():
    """k_stereobm_c4_impl.h computes  ms * uniq / 100  as  (ms * K) >> 23  with K = ceil(uniq * 2^23 / 100): exact for every minimum cost the u16
    path can produce and every uniqueness ratio the fast path accepts (gip_bm_c4_can: uniq < 200)."""
    import numpy as np
    ms = np.arange(65536, dtype=np.uint64)
    for u in range(200):
        K = (u * (1 << 23) + 99) // 100
        assert K < (1 << 24)
        assert np.array_equal((ms * u) // 100, (ms * K) >> 23), u


def test_rowsplit_plan_is_host_logic_and_matches_the_distributed_form(built_lib):
    """gip_rowsplit_plan_rows needs no device: the native row split (gip_rowsplit_*) and the torch.distributed form (rowsplit.py) must cut
    a frame into the same tiles with the same halos -- contiguous cover, even first extended row, XSOBEL / NORMALIZED_RESPONSE halo,
    rejection of tiles thinner than the rows they lend; and without a device the split itself cannot be opened (no CPU fallback)."""
    import ctypes as C
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    L = capi.lib()
    for height, n, block, ptype, psize in [(2160, 4, 15, 1, 9), (375, 3, 9, 1, 9), (301, 5, 11, 1, 9), (1080, 8, 21, 0, 9), (480, 2, 31, 0, 21), (97, 1, 5, 1, 5)]:
        p = capi.BMParams(prefilter_type=ptype, prefilter_size=psize, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=64, texture_threshold=10,
                          uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
        halo = RS.halo_rows(block, ptype, psize)
        ranges = RS.plan(height, n)
        prev_end = 0
        for t in range(n):
            v = [C.c_int() for _ in range(5)]
            assert L.gip_rowsplit_plan_rows(height, n, C.byref(p), t, *[C.byref(x) for x in v]) == capi.OK
            y0, y1, e0, e1, hl = [x.value for x in v]
            assert (y0, y1) == ranges[t] and y0 == prev_end and hl == halo
            assert (e0, e1) == RS.extended_range(height, y0, y1, halo) and e0 % 2 == 0 and e0 <= y0 and e1 >= y1
            prev_end = y1
        assert prev_end == height
    p = capi.BMParams(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=21, min_disparity=0, num_disparities=64, texture_threshold=10,
                      uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    v = [C.c_int() for _ in range(5)]
    assert L.gip_rowsplit_plan_rows(100, 8, C.byref(p), 0, *[C.byref(x) for x in v]) == capi.ERR_INVALID_ARG      # 12-row tiles, 11 + 2 needed
    assert L.gip_rowsplit_plan_rows(100, 17, C.byref(p), 0, *[C.byref(x) for x in v]) == capi.ERR_INVALID_ARG
    import torch
    if not torch.cuda.is_available():
        cfg = capi.RowSplitConfig()
        cfg.width, cfg.height, cfg.n_tiles = 640, 480, 2
        ci = capi.CameraInfo()
        h = C.c_void_p()
        assert L.gip_rowsplit_open(C.byref(ci), C.byref(ci), C.byref(p), 0, 0.0, C.byref(cfg), C.byref(h)) == capi.ERR_NO_DEVICE
