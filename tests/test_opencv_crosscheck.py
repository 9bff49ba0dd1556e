"""Opportunistic cross-check of the CPU oracle against a real OpenCV: only where `import cv2` already works.
Nothing is installed for it; in this image (and on the GPU box) OpenCV is absent and every test here is skipped -- which is
why the StereoBM-family functions of the oracle stay "parity unpinned" (oracle/gip_oracle.h)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed (and must not be installed): parity of the StereoBM family stays unpinned")


def _bm(cv2, p):
    bm = cv2.StereoBM_create(numDisparities=p.num_disparities, blockSize=p.block_size)
    bm.setPreFilterType(p.prefilter_type)
    bm.setPreFilterSize(p.prefilter_size)
    bm.setPreFilterCap(p.prefilter_cap)
    bm.setMinDisparity(p.min_disparity)
    bm.setTextureThreshold(p.texture_threshold)
    bm.setUniquenessRatio(p.uniqueness_ratio)
    bm.setDisp12MaxDiff(p.disp12_max_diff)
    bm.setSpeckleWindowSize(0)
    bm.setSpeckleRange(0)
    return bm


@pytest.mark.parametrize("kw", [dict(block_size=15, num_disparities=64), dict(block_size=9, num_disparities=128, disp12_max_diff=1),
                                dict(block_size=21, num_disparities=48, prefilter_type=0, prefilter_size=9, uniqueness_ratio=0)])
def test_stereobm_matches_opencv(oracle, kitchen, kw):
    p = oracle.bm_params(**kw)
    exp = _bm(cv2, p).compute(kitchen["left_rect"], kitchen["right_rect"])
    got = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], p)
    assert np.array_equal(got, exp)


def test_filter_speckles_matches_opencv(oracle, kitchen):
    d = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], oracle.bm_params(block_size=15, num_disparities=64))
    exp = d.copy()
    cv2.filterSpeckles(exp, -16, 800, 80)
    assert np.array_equal(oracle.filter_speckles(d, -16, 800, 80), exp)


def test_reproject_matches_opencv(oracle, calib, kitchen):
    Q = oracle.stereo_q(calib["left"]["P"], calib["right"]["P"])
    d16 = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], oracle.bm_params(block_size=15, num_disparities=64))
    f = oracle.disparity_to_f32(d16, calib["left"]["P"][2], calib["right"]["P"][2])
    exp = cv2.reprojectImageTo3D(f, np.asarray(Q, np.float64).reshape(4, 4), handleMissingValues=True)
    got = oracle.reproject_to_3d(f, Q, True)
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], exp[fin], rtol=2.4e-7, atol=0)
