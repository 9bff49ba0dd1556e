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
