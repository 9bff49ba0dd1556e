"""GPU tests (-m gpu) of the handle API, written after the reference's own gtests (test/UTest.cpp): same slots, same call
sequences, same fixtures -- but with numeric assertions against the CPU oracle where the reference asserts nothing."""
import threading

import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def proc(gpu):
    from ros_gpu_stereo_processor_amd.processor import GpuStereoProcessor
    p = GpuStereoProcessor()
    yield p
    p.close()


def _init_kitchen(proc, calib_files):
    """CudaStereoBMTf::initStereoModelKitchen (UTest.cpp:68-76)"""
    from ros_gpu_stereo_processor_amd import capi
    proc.initStereoModel(*calib_files)
    proc.setPreFilterType(capi.PREFILTER_XSOBEL)
    proc.setMaxSpeckleDiff(200)
    proc.setMaxSpeckleSize(200)


def test_GpuTransfer(proc, aloe):
    """UTest.cpp:179-188"""
    from ros_gpu_stereo_processor_amd import processor as P
    img = aloe["left_rgb"]
    proc.uploadMat(P.L_RECT_MONO, img)
    proc.waitForStream(P.L_RECT_MONO)
    assert np.array_equal(proc.downloadMat(P.L_RECT_MONO), img)
    # strided source (a cv::Mat view)
    view = img[10:200, 20:300]
    proc.uploadMat(P.R_RAW, view)
    assert np.array_equal(proc.downloadMat(P.R_RAW), view)


@pytest.mark.parametrize("pixel,encoding,color,mono", [
    ([10, 20, 30], "bgr8", [10, 20, 30], 22),   # GpuColorConversionColorMono  UTest.cpp:190-207
    ([30, 20, 10], "rgb8", [10, 20, 30], 22),   # GpuColorConversionColorMono2 UTest.cpp:209-226
    ([22], "mono8", [22, 22, 22], 22),          # GpuColorConversionMonoColor  UTest.cpp:228-245
])
def test_GpuColorConversion(proc, pixel, encoding, color, mono):
    from ros_gpu_stereo_processor_amd import processor as P
    img = np.array([[pixel]], np.uint8) if len(pixel) == 3 else np.array([pixel], np.uint8).reshape(1, 1)
    proc.uploadMat(P.L_RAW, img, encoding)
    proc.convertRawToColor(P.SIDE_L)
    proc.convertRawToMono(P.SIDE_L)
    proc.waitForStream(P.SIDE_L)
    assert proc.downloadMat(P.L_COLOR).reshape(-1).tolist() == color
    assert proc.downloadMat(P.L_MONO).reshape(-1).tolist() == [mono]


def test_convert_bayer_and_errors(proc, oracle):
    from ros_gpu_stereo_processor_amd import processor as P, capi
    rng = np.random.default_rng(31)
    raw = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    proc.uploadMat(P.R_RAW, raw, "bayer_grbg8")
    proc.convertRawToColor(P.SIDE_R)
    proc.convertRawToMono(P.SIDE_R)
    assert np.array_equal(proc.downloadMat(P.R_COLOR), oracle.bayer_to_bgr(raw, "bayer_grbg8"))
    assert np.array_equal(proc.downloadMat(P.R_MONO), oracle.bayer_to_gray(raw, "bayer_grbg8"))
    with pytest.raises(capi.GipError):  # empty encoding: the reference asserts (GPUStereoProcessor.cpp:121)
        proc.convertColor(P.R_RAW, P.R_MONO, "", "mono8")
    with pytest.raises(capi.GipError):
        proc.downloadMat(P.L_POINTS2)  # never written
    with pytest.raises(capi.GipError) as e:
        proc.rectifyImage(P.R_MONO, P.R_RECT_MONO)  # model not initialised (assert in the reference, :238)
    assert e.value.status == capi.ERR_NOT_INITIALISED


def test_RectifyMonoGpu(proc, oracle, kitchen, calib, calib_files):
    """UTest.cpp:262-288 -- the reference tolerates a mean one-sided error of 0.1; here the GPU remap is the fixed-point
    form, bit-exact with the oracle and within the golden's rounding-boundary residue."""
    from ros_gpu_stereo_processor_amd import processor as P
    _init_kitchen(proc, calib_files)
    assert proc.isStereoModelInitialised()
    proc.uploadMat(P.L_RAW, kitchen["left_raw"])
    proc.uploadMat(P.R_RAW, kitchen["right_raw"])
    proc.rectifyImage(P.L_RAW, P.L_RECT_MONO)
    proc.rectifyImage(P.R_RAW, P.R_RECT_MONO)
    for slot, side in ((P.L_RECT_MONO, "left"), (P.R_RECT_MONO, "right")):
        got = proc.downloadMat(slot)
        c = calib[side]
        m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], c["width"], c["height"])
        assert np.array_equal(got, oracle.remap(kitchen[side + "_raw"], m1, m2))
        diff = np.abs(got.astype(int) - kitchen[side + "_rect"].astype(int))
        assert diff.max() <= 2 and (diff > 0).mean() <= 0.0002
        # the host-built tables themselves are identical to the oracle's
        gm1, gm2 = proc.rectifyMaps(P.SIDE_L if side == "left" else P.SIDE_R)
        assert np.array_equal(gm1, m1) and np.array_equal(gm2, m2)
    m = proc.model()
    assert abs(m["baseline"] - 44.133133 / 441.238411) < 1e-12 and m["fx"] == 441.238411
    assert np.array_equal(m["Q"], oracle.stereo_q(calib["left"]["P"], calib["right"]["P"]))


def test_DisparityGpu(proc, oracle, kitchen, calib_files, calib):
    """UTest.cpp:290-331: rectify -> disparity -> sender.  The reference constructor's effective CPU-matcher parameters
    (ndisp 48, block 19, XSOBEL, texture 3, uniqueness 0, disp12MaxDiff 0) are the handle's defaults."""
    from ros_gpu_stereo_processor_amd import processor as P
    _init_kitchen(proc, calib_files)
    proc.uploadMat(P.L_RAW, kitchen["left_raw"])
    proc.uploadMat(P.R_RAW, kitchen["right_raw"])
    proc.rectifyImage(P.L_RAW, P.L_RECT_MONO)
    proc.rectifyImage(P.R_RAW, P.R_RECT_MONO)
    proc.computeDisparity(P.L_RECT_MONO, P.R_RECT_MONO, P.L_DISPARITY)
    got = {}
    done = threading.Event()

    def pub(msg, user):
        m = msg.contents
        got["shape"] = (m.image.height, m.image.width, m.image.step)
        got["f"], got["T"] = m.f, m.T
        done.set()

    sid = proc.enqueueSendDisparity(P.L_DISPARITY, pub=pub)
    l_rect, r_rect = proc.downloadMat(P.L_RECT_MONO), proc.downloadMat(P.R_RECT_MONO)
    proc.waitForAllStreams()
    assert proc.wasDataSent(sid) and done.wait(5)
    p = proc.getParams()
    po = oracle.bm_params(prefilter_type=p.prefilter_type, prefilter_size=p.prefilter_size, prefilter_cap=p.prefilter_cap, block_size=p.block_size,
                          min_disparity=p.min_disparity, num_disparities=p.num_disparities, texture_threshold=p.texture_threshold,
                          uniqueness_ratio=p.uniqueness_ratio, disp12_max_diff=p.disp12_max_diff)
    assert (p.num_disparities, p.block_size, p.texture_threshold, p.uniqueness_ratio, p.disp12_max_diff) == (48, 19, 3, 0, 0)
    exp16 = oracle.stereobm(l_rect, r_rect, po)
    d16 = proc.downloadMat(P.L_DISPARITY)
    assert np.array_equal(d16, exp16)
    d32 = proc.downloadMat(P.L_DISPARITY_32F)
    assert np.array_equal(d32, oracle.disparity_to_f32(exp16, calib["left"]["P"][2], calib["right"]["P"][2]))
    msg = proc.disparityMessage(sid)
    assert np.array_equal(msg["image"], d32) and msg["encoding"] == "32FC1" and msg["step"] == 752 * 4
    border = 19 // 2
    assert msg["valid_window"] == (48 + 0 + border - 1, border, (752 - 1 - border) - (48 + border - 1), (480 - 1 - border) - border)
    assert msg["min_disparity"] == 1 and msg["max_disparity"] == 47
    assert abs(msg["f"] - 441.238411) < 1e-3 and abs(msg["T"] - 0.10002) < 1e-4 and msg["delta_d"] == 1 / 16
    assert got["shape"] == (480, 752, 752 * 4)
    proc.cleanSenders()
    # speckle filter as the node runs it right after (StereoProcessor.cpp:262); x16 map => maxDiff * 16, newVal FILTERED
    proc.filterSpeckles(P.L_DISPARITY)
    assert np.array_equal(proc.downloadMat(P.L_DISPARITY), oracle.filter_speckles(exp16, -16, 200, 200 * 16))
    assert np.array_equal(proc.downloadMat(P.L_DISPARITY_32F),
                          oracle.disparity_to_f32(oracle.filter_speckles(exp16, -16, 200, 200 * 16), calib["left"]["P"][2], calib["right"]["P"][2]))


def test_ExportDisparities_configureProcessor(proc, oracle, kitchen, calib_files):
    """UTest.cpp:333-363 with configureProcessor (:78-87): block 11, ndisp 128, prefilter type 0, texture 0 on a 200-px-wide
    downscaled pair.  The reference only dumps CSV files; here GPU == oracle."""
    from ros_gpu_stereo_processor_amd import processor as P
    _init_kitchen(proc, calib_files)
    proc.setBlockSize(11)
    proc.setMaxSpeckleSize(0)
    proc.setMinDisparity(0)
    proc.setNumDisparities(128)
    proc.setPreFilterType(0)
    proc.setRefineDisparity(0)
    proc.setTextureThreshold(0)
    step = 752 // 200
    l = np.ascontiguousarray(kitchen["left_rect"][::step, ::step][:, :200])
    r = np.ascontiguousarray(kitchen["right_rect"][::step, ::step][:, :200])
    proc.uploadMat(P.L_RECT_MONO, l)
    proc.uploadMat(P.R_RECT_MONO, r)
    proc.computeDisparity(P.L_RECT_MONO, P.R_RECT_MONO, P.L_DISPARITY)
    p = proc.getParams()
    po = oracle.bm_params(prefilter_type=0, prefilter_size=p.prefilter_size, prefilter_cap=31, block_size=11, min_disparity=0, num_disparities=128,
                          texture_threshold=0, uniqueness_ratio=0, disp12_max_diff=0)
    assert np.array_equal(proc.downloadMat(P.L_DISPARITY), oracle.stereobm(l, r, po))


def test_PointCloud(proc, oracle, kitchen, calib_files, calib):
    """UTest.cpp:365-398: the full chain to PointCloud2 (colour from mono8 via GRAY2BGR)."""
    from ros_gpu_stereo_processor_amd import processor as P
    _init_kitchen(proc, calib_files)
    proc.setNumDisparities(64)
    proc.setBlockSize(15)
    proc.uploadMat(P.L_RAW, kitchen["left_raw"], "mono8")
    proc.uploadMat(P.R_RAW, kitchen["right_raw"], "mono8")
    proc.convertRawToMono(P.SIDE_L)
    proc.convertRawToMono(P.SIDE_R)
    proc.convertRawToColor(P.SIDE_L)
    proc.rectifyImage(P.L_MONO, P.L_RECT_MONO)
    proc.rectifyImage(P.R_MONO, P.R_RECT_MONO)
    proc.rectifyImage(P.L_COLOR, P.L_RECT_COLOR)
    proc.computeDisparity(P.L_RECT_MONO, P.R_RECT_MONO, P.L_DISPARITY)
    proc.projectDisparityTo3DPoints(P.L_DISPARITY, P.L_POINTS2)
    sid = proc.enqueueSendPoints(P.L_POINTS2, P.L_RECT_COLOR)
    proc.waitForAllStreams()
    assert proc.wasDataSent(sid)
    msg = proc.pointCloudMessage(sid)
    assert (msg["height"], msg["width"], msg["point_step"], msg["row_step"], msg["is_dense"]) == (480, 752, 32, 32 * 752, 0)
    assert msg["fields"] == [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("rgb", 16, 7, 1)]
    l_rect = proc.downloadMat(P.L_RECT_MONO)
    assert np.array_equal(proc.downloadMat(P.L_RECT_COLOR), np.stack([l_rect] * 3, -1))
    d32 = proc.downloadMat(P.L_DISPARITY_32F)
    Q = oracle.stereo_q(calib["left"]["P"], calib["right"]["P"])
    xyz = oracle.reproject_to_3d(d32, Q, True)
    pts = proc.downloadMat(P.L_POINTS2)
    fin = np.isfinite(xyz)
    assert np.array_equal(np.isfinite(pts), fin) and np.allclose(pts[fin], xyz[fin], rtol=2.4e-7, atol=0)
    exp = oracle.pack_pointcloud2(pts, np.stack([l_rect] * 3, -1))
    assert np.array_equal(msg["data"], exp)
    nan_frac = np.isnan(msg["data"][..., :4].copy().view(np.float32)).mean()
    assert 0.2 < nan_frac < 0.99


def test_process_pair_schedule(proc, oracle, calib):
    """StereoProcessor::imageCb (StereoProcessor.cpp:157-298) for a Bayer pair with every topic connected."""
    from ros_gpu_stereo_processor_amd import processor as P, capi
    w, h = 752, 480
    cal = synth.scaled_calibration(w, h)
    proc.initStereoModel(P.camera_info(w, h, **cal["left"]), P.camera_info(w, h, **cal["right"]))
    proc.setNumDisparities(64)
    proc.setBlockSize(9)
    proc.setMaxSpeckleSize(100)
    proc.setMaxSpeckleDiff(2)
    lr, rr, _, _ = synth.raw_pair(w, h, 64, seed=77)
    lb = synth.bayer_mosaic(synth.colorize(lr, 1), "bayer_rggb8")
    rb = synth.bayer_mosaic(synth.colorize(rr, 2), "bayer_rggb8")
    proc.processPair(lb, rb, "bayer_rggb8", 0x7FF)
    mono_l, mono_r = oracle.bayer_to_gray(lb, "bayer_rggb8"), oracle.bayer_to_gray(rb, "bayer_rggb8")
    assert np.array_equal(proc.downloadMat(P.L_MONO), mono_l) and np.array_equal(proc.downloadMat(P.R_MONO), mono_r)
    col_l = oracle.bayer_to_bgr(lb, "bayer_rggb8")
    assert np.array_equal(proc.downloadMat(P.L_COLOR), col_l)
    ml = oracle.init_undistort_rectify_map(cal["left"]["K"], cal["left"]["D"], cal["left"]["R"], cal["left"]["P"], w, h)
    mr = oracle.init_undistort_rectify_map(cal["right"]["K"], cal["right"]["D"], cal["right"]["R"], cal["right"]["P"], w, h)
    rect_l, rect_r = oracle.remap(mono_l, *ml), oracle.remap(mono_r, *mr)
    assert np.array_equal(proc.downloadMat(P.L_RECT_MONO), rect_l) and np.array_equal(proc.downloadMat(P.R_RECT_MONO), rect_r)
    assert np.array_equal(proc.downloadMat(P.L_RECT_COLOR), oracle.remap(col_l, *ml))
    p = proc.getParams()
    po = oracle.bm_params(prefilter_type=p.prefilter_type, prefilter_size=p.prefilter_size, prefilter_cap=p.prefilter_cap, block_size=9, num_disparities=64,
                          texture_threshold=p.texture_threshold, uniqueness_ratio=p.uniqueness_ratio, disp12_max_diff=p.disp12_max_diff)
    d = oracle.filter_speckles(oracle.stereobm(rect_l, rect_r, po), -16, 100, 32)
    assert np.array_equal(proc.downloadMat(P.L_DISPARITY), d)
    assert (d != -16).mean() > 0.3  # the synthetic scene is matched over most of the frame
    assert np.array_equal(proc.downloadMat(P.L_DISPARITY_IMG), oracle.draw_color_disp(d, 64))
    # only what the mask asks for: with just the disparity topic connected no colour slot is produced
    from ros_gpu_stereo_processor_amd.processor import GpuStereoProcessor
    q = GpuStereoProcessor()
    q.initStereoModel(P.camera_info(w, h, **cal["left"]), P.camera_info(w, h, **cal["right"]))
    q.processPair(lr, rr, "mono8", capi.TOPIC_DISPARITY)
    with pytest.raises(capi.GipError):
        q.downloadMat(P.L_COLOR)
    assert q.downloadMat(P.L_DISPARITY).shape == (h, w)
    q.close()


def test_setters_validate_like_configcb(gpu):
    """Setters reject values that can never be valid and leave the parameter unchanged (the reference forwards anything to
    cv::StereoBM and fails inside compute(); its configCb fixes window and range up first, src/StereoProcessor.cpp:309-311)."""
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    p = P.GpuStereoProcessor()
    before = p.getParams()
    for setter, bad in ((p.setBlockSize, 8), (p.setBlockSize, 3), (p.setBlockSize, 257), (p.setNumDisparities, 40), (p.setNumDisparities, 0),
                        (p.setPreFilterType, 2), (p.setPreFilterCap, 64), (p.setPreFilterSize, 4), (p.setTextureThreshold, -1),
                        (p.setUniquenessRatio, -5), (p.setMaxSpeckleSize, -1), (p.setMaxSpeckleDiff, -0.5)):
        with pytest.raises(capi.GipError) as e:
            setter(bad)
        assert e.value.status == capi.ERR_INVALID_ARG
    after = p.getParams()
    assert (before.block_size, before.num_disparities, before.prefilter_cap) == (after.block_size, after.num_disparities, after.prefilter_cap) == (19, 48, 31)
    p.setBlockSize(21); p.setNumDisparities(64); p.setMinDisparity(-16); p.setMaxSpeckleSize(0)
    q = p.getParams()
    assert (q.block_size, q.num_disparities, q.min_disparity) == (21, 64, -16)
    bad = p.getParams()
    bad.block_size = 10
    with pytest.raises(capi.GipError):
        p.setParams(bad)
    assert p.getParams().block_size == 21
    p.close()
