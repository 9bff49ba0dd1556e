"""More GPU parity cases (-m gpu): nearest-neighbour rectification, processor-level INTER_NEAREST."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("side", ["left", "right"])
def test_remap_nearest(gpu, oracle, kitchen, calib, side):
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    c = calib[side]
    m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], c["width"], c["height"])
    raw = kitchen[side + "_raw"]
    got = K.remap(_t(raw, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu), capi.INTER_NEAREST).cpu().numpy()
    exp = oracle.remap(raw, m1, m2, nearest=True)
    assert np.array_equal(got, exp)
    # nearest differs from bilinear but stays close on a smooth image
    lin = oracle.remap(raw, m1, m2)
    assert (got != lin).mean() > 0.3 and np.abs(got.astype(int) - lin.astype(int)).mean() < 6
    bgr = np.stack([raw, 255 - raw, raw // 2], -1)
    got3 = K.remap(_t(bgr, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu), capi.INTER_NEAREST).cpu().numpy()
    assert np.array_equal(got3, oracle.remap(bgr, m1, m2, nearest=True))


def test_processor_rectify_nearest(gpu, oracle, kitchen, calib, calib_files):
    from ros_gpu_stereo_processor_amd import processor as P, capi
    p = P.GpuStereoProcessor()
    p.initStereoModel(*calib_files)
    p.uploadMat(P.L_RAW, kitchen["left_raw"])
    p.rectifyImage(P.L_RAW, P.L_RECT_MONO, capi.INTER_NEAREST)
    c = calib["left"]
    m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], c["width"], c["height"])
    assert np.array_equal(p.downloadMat(P.L_RECT_MONO), oracle.remap(kitchen["left_raw"], m1, m2, nearest=True))
    with pytest.raises(capi.GipError):
        p.rectifyImage(P.L_RAW, P.L_RECT_MONO, 2)  # INTER_CUBIC: not implemented
    p.close()


@pytest.mark.parametrize("w,h,nd,block,cap,mind,d12", [(320, 96, 32, 33, 31, 0, 1), (400, 120, 64, 51, 31, 0, -1), (300, 80, 16, 23, 63, 0, 1),
                                                        (520, 300, 48, 255, 31, 0, 0), (260, 90, 32, 41, 63, -8, 1)])
def test_wide_window_fallback(gpu, oracle, w, h, nd, block, cap, mind, d12):
    """windows whose SAD exceeds 16 bits (cv::StereoBM allows block up to 255, cap up to 63): 32-bit fallback matcher"""
    from ros_gpu_stereo_processor_amd import kernels as K, synth
    l, r, _ = synth.rectified_pair(w, h, max(16, min(nd, w // 4)), seed=w + block)
    kw = dict(block_size=block, num_disparities=nd, prefilter_cap=cap, min_disparity=mind, texture_threshold=10, uniqueness_ratio=10, disp12_max_diff=d12)
    exp = oracle.stereobm(l, r, oracle.bm_params(**kw))
    got = K.stereobm(_t(l, gpu), _t(r, gpu), K.bm_params(**kw)).cpu().numpy()
    assert np.array_equal(got, exp)
    if block < min(w, h) // 2:
        assert (exp != (mind - 1) * 16).any()
