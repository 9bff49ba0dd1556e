"""CPU tests (-m "not gpu"): the oracle against every golden vector / known-answer case the reference holds for the path
 and against independent brute-force restatements on small inputs."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth


# ---------------------------------------------------------------- goldens held by the reference
@pytest.mark.parametrize("side", ["left", "right"])
def test_rectify_matches_reference_golden(oracle, kitchen, calib, side):
    """reference test RectifyMonoCpu (test/UTest.cpp:247-260): CPU rectification of left/right-0022.png vs *_rect.png.
    The restatement reproduces the goldens except for 32 / 35 of 360 960 pixels that sit on 1/32-px rounding boundaries of
    the map (|delta| <= 2) -- acceptance: >= 99.98 % identical, |delta| <= 2."""
    c = calib[side]
    m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], c["width"], c["height"])
    rect = oracle.remap(kitchen[side + "_raw"], m1, m2)
    diff = np.abs(rect.astype(int) - kitchen[side + "_rect"].astype(int))
    assert diff.max() <= 2
    assert (diff > 0).mean() <= 0.0002


def test_rgb_to_gray_matches_reference_golden(oracle, aloe):
    """aloe-L.png -> aloe-L-Mono.png pins (4899 R + 9617 G + 1868 B + 8192) >> 14 bit-exactly."""
    assert np.array_equal(oracle.bgr_to_gray(aloe["left_rgb"], src_is_rgb=True), aloe["left_mono"])


def test_colour_1x1_cases(oracle):
    """GpuColorConversionColorMono / ColorMono2 / MonoColor (test/UTest.cpp:190-245)"""
    bgr = np.array([[[10, 20, 30]]], np.uint8)
    rgb = np.array([[[30, 20, 10]]], np.uint8)
    mono = np.array([[22]], np.uint8)
    assert oracle.convert_color(bgr, "bgr8", "mono8")[0, 0] == 22
    assert oracle.convert_color(bgr, "bgr8", "bgr8").tolist() == bgr.tolist()
    assert oracle.convert_color(rgb, "rgb8", "mono8")[0, 0] == 22
    assert oracle.convert_color(rgb, "rgb8", "bgr8").tolist() == bgr.tolist()
    assert oracle.convert_color(mono, "mono8", "bgr8").tolist() == [[[22, 22, 22]]]
    assert oracle.convert_color(mono, "mono8", "mono8")[0, 0] == 22


# ---------------------------------------------------------------- independent restatements (numpy brute force)
def _brute_sad(L, R, p, x, y):
    """SAD vector of output column x (absolute), row y by the definition in SURVEY.md Appendix A.1 (clamped columns)."""
    h, w = L.shape
    nd, mind, r = p.num_disparities, p.min_disparity, p.block_size // 2
    lofs, rofs = max(nd - 1 + mind, 0), -min(nd - 1 + mind, 0)
    c = x - lofs
    out = np.zeros(nd, np.int64)
    for dx in range(-r, r + 1):
        cc = c + dx
        lc = lofs + min(max(cc, -lofs), w - 1 - lofs)
        rc = rofs + min(max(cc, -rofs), w - nd - rofs)
        lcol = L[y - r:y + r + 1, lc].astype(np.int64)
        for d in range(nd):
            out[d] += np.abs(lcol - R[y - r:y + r + 1, rc + d]).sum()
    return out


def _brute_pixel(L, R, p, x, y):
    nd, mind, r, cap = p.num_disparities, p.min_disparity, p.block_size // 2, p.prefilter_cap
    h, w = L.shape
    lofs = max(nd - 1 + mind, 0)
    FILTERED = (mind - 1) * 16
    sad = _brute_sad(L, R, p, x, y)
    cols = [lofs + min(max(x - lofs + dx, -lofs), w - 1 - lofs) for dx in range(-r, r + 1)]
    tsum = int(np.abs(L[y - r:y + r + 1][:, cols].astype(np.int64) - cap).sum())
    mind_i = int(np.argmin(sad))
    minsad = int(sad[mind_i])
    if tsum < p.texture_threshold:
        return FILTERED, minsad
    if p.uniqueness_ratio > 0:
        thresh = minsad + minsad * p.uniqueness_ratio // 100
        for d in range(nd):
            if (d < mind_i - 1 or d > mind_i + 1) and sad[d] <= thresh:
                return FILTERED, minsad
    pp = int(sad[mind_i + 1]) if mind_i + 1 < nd else int(sad[nd - 2])
    nn = int(sad[mind_i - 1]) if mind_i > 0 else int(sad[1])
    den = pp + nn - 2 * minsad + abs(pp - nn)
    frac = int((pp - nn) * 256 / den) if den != 0 else 0  # C division truncates toward zero
    if den != 0:
        q = abs((pp - nn) * 256) // den
        frac = q if (pp - nn) >= 0 else -q
    return ((nd - mind_i - 1 + mind) * 256 + frac + 15) >> 4, minsad


@pytest.mark.parametrize("mind,nd,block", [(0, 16, 5), (0, 32, 9), (-8, 16, 7), (4, 16, 5), (-20, 16, 5)])
def test_bm_match_against_bruteforce(oracle, mind, nd, block):
    rng = np.random.default_rng(5)
    h, w = 24, 72
    L = rng.integers(0, 63, (h, w), dtype=np.uint8)
    R = np.roll(L, -3, axis=1)
    R[:, ::5] = rng.integers(0, 63, (h, len(range(0, w, 5))), dtype=np.uint8)
    p = oracle.bm_params(block_size=block, num_disparities=nd, min_disparity=mind, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    disp, cost = oracle.bm_match(L, R, p)
    r = block // 2
    lofs, rofs = max(nd - 1 + mind, 0), -min(nd - 1 + mind, 0)
    width1 = min(w - rofs - nd + 1, w - lofs)
    FILTERED = (mind - 1) * 16
    roi = oracle.valid_disparity_roi(w, h, p)
    for y in range(h):
        for x in range(w):
            inside_rows = roi[1] <= y < roi[1] + roi[3]
            if not inside_rows or x < lofs or x >= lofs + width1:
                assert disp[y, x] == FILTERED
                continue
            d, c = _brute_pixel(L, R, p, x, y)
            assert disp[y, x] == d, (x, y)
            if d != FILTERED:
                assert cost[y, x] == c


def test_prefilter_xsobel_bruteforce(oracle):
    rng = np.random.default_rng(1)
    for h, w in [(9, 17), (10, 16), (2, 8), (1, 8)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = oracle.prefilter_xsobel(img, 31)
        exp = np.full((h, w), 31, np.int64)
        a = img.astype(np.int64)
        for y in range(h):
            if h % 2 == 1 and y == h - 1:
                continue
            ym = y - 1 if y > 0 else (1 if h > 1 else 0)
            yp = y + 1 if y < h - 1 else (h - 2 if h > 1 else 0)
            for x in range(1, w - 1):
                v = (a[ym, x + 1] - a[ym, x - 1]) + 2 * (a[y, x + 1] - a[y, x - 1]) + (a[yp, x + 1] - a[yp, x - 1])
                exp[y, x] = min(max(v, -31), 31) + 31
        assert np.array_equal(got, exp), (h, w)


def test_prefilter_norm_bruteforce(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    for ws in (5, 9):
        got = oracle.prefilter_norm(img, ws, 31)
        a = np.pad(img.astype(np.int64), ws // 2 + 1, mode="edge")
        o = ws // 2 + 1
        sg = ws * ws // 8
        ss = (1024 + sg) // (sg * 2)
        sg *= ss
        exp = np.zeros_like(img, dtype=np.int64)
        for y in range(img.shape[0]):
            for x in range(img.shape[1]):
                box = a[y + o - ws // 2:y + o + ws // 2 + 1, x + o - ws // 2:x + o + ws // 2 + 1].sum()
                c = a[y + o, x + o]
                v = ((4 * c + a[y + o, x + o - 1] + a[y + o, x + o + 1] + a[y + o - 1, x + o] + a[y + o + 1, x + o]) * sg - box * ss) >> 10
                exp[y, x] = min(max(v, -31), 31) + 31
        assert np.array_equal(got, exp)


def test_filter_speckles_against_scipy(oracle):
    """components of the fixed graph (4-neighbours, |diff| <= maxDiff) -- independent implementation with networkx-free BFS"""
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 6, (40, 50)) * 16).astype(np.int16)
    img[rng.random(img.shape) < 0.2] = -16
    new_val, max_size, max_diff = -16, 6, 16
    got = oracle.filter_speckles(img, new_val, max_size, max_diff)
    exp = img.copy()
    seen = np.zeros(img.shape, bool)
    for y0 in range(img.shape[0]):
        for x0 in range(img.shape[1]):
            if seen[y0, x0] or img[y0, x0] == new_val:
                continue
            comp, q = [], [(y0, x0)]
            seen[y0, x0] = True
            while q:
                y, x = q.pop()
                comp.append((y, x))
                for yy, xx in ((y + 1, x), (y - 1, x), (y, x + 1), (y, x - 1)):
                    if 0 <= yy < img.shape[0] and 0 <= xx < img.shape[1] and not seen[yy, xx] and img[yy, xx] != new_val \
                            and abs(int(img[yy, xx]) - int(img[y, x])) <= max_diff:
                        seen[yy, xx] = True
                        q.append((yy, xx))
            if len(comp) <= max_size:
                for y, x in comp:
                    exp[y, x] = new_val
    assert np.array_equal(got, exp)


def test_bayer_to_bgr_interior_and_border(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (12, 14), dtype=np.uint8)
    for enc, first in (("bayer_rggb8", 2), ("bayer_bggr8", 0), ("bayer_gbrg8", 1), ("bayer_grbg8", 1)):
        out = oracle.bayer_to_bgr(img, enc)
        # own-colour samples are copied through at interior sites
        tiles = {"bayer_rggb8": (2, 1, 1, 0), "bayer_bggr8": (0, 1, 1, 2), "bayer_gbrg8": (1, 0, 2, 1), "bayer_grbg8": (1, 2, 0, 1)}[enc]
        for y in range(1, 11):
            for x in range(1, 13):
                ch = tiles[(y & 1) * 2 + (x & 1)]
                assert out[y, x, ch] == img[y, x]
        # borders replicate the adjacent interior pixel
        assert np.array_equal(out[:, 0], out[:, 1]) and np.array_equal(out[:, -1], out[:, -2])
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[-1], out[-2])
    # a flat image stays flat in every channel and in gray
    flat = np.full((8, 8), 100, np.uint8)
    assert (oracle.bayer_to_bgr(flat, "bayer_rggb8") == 100).all()
    g = oracle.bayer_to_gray(flat, "bayer_rggb8")
    assert (g == ((100 * (4899 + 9617 + 1868) + 8192) >> 14)).all()


def test_stereobm_recovers_planted_disparities(oracle):
    """property test: piece-wise constant integer disparities are recovered exactly away from discontinuities"""
    l, r, d = synth.rectified_pair(320, 120, 64, seed=7, noise_sigma=0.0)
    p = oracle.bm_params(block_size=9, num_disparities=64, texture_threshold=0, uniqueness_ratio=0)
    disp = oracle.stereobm(l, r, p)
    roi = oracle.valid_disparity_roi(320, 120, p)
    sub = disp[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]
    truth = d[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]] * 16
    # pixels whose 9x9 neighbourhood has a single disparity
    from scipy.ndimage import maximum_filter, minimum_filter
    flat = (maximum_filter(d, 11) == minimum_filter(d, 11))[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]]
    assert flat.mean() > 0.3
    assert (np.abs(sub[flat].astype(int) - truth[flat]) <= 8).mean() > 0.98  # within half a pixel (occlusion fringes excepted)
    # invariants: valid values within [minD*16, (minD+ndisp-1)*16], everything outside the ROI FILTERED
    valid = disp != -16
    assert disp[valid].min() >= 0 and disp[valid].max() <= 63 * 16
    mask = np.ones_like(disp, bool)
    mask[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]] = False
    assert (disp[mask] == -16).all()


def test_stereobm_thread_count_independent(oracle, kitchen):
    p = oracle.bm_params(block_size=15, num_disparities=64, disp12_max_diff=1)
    a = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], p, nthreads=1)
    b = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], p, nthreads=4)
    assert np.array_equal(a, b)


def test_reproject_missing_rule_equals_filtered(oracle, calib):
    """handleMissingValues marks the image MINIMUM; for a StereoBM map that is exactly the FILTERED value."""
    Q = oracle.stereo_q(calib["left"]["P"], calib["right"]["P"])
    d16 = np.full((8, 10), -16, np.int16)
    d16[2:6, 3:8] = np.arange(20).reshape(4, 5) * 16 + 160
    d32 = oracle.disparity_to_f32(d16, calib["left"]["P"][2], calib["right"]["P"][2])
    xyz = oracle.reproject_to_3d(d32, Q, True)
    assert (xyz[..., 2][d16 == -16] == 10000).all()
    fx, base = calib["left"]["P"][0], -calib["right"]["P"][3] / calib["right"]["P"][0]
    z = xyz[..., 2][d16 != -16]
    assert np.allclose(z, fx * base / (d16[d16 != -16] / 16.0), rtol=1e-6)
    pc = oracle.pack_pointcloud2(xyz, np.zeros((8, 10, 3), np.uint8) + np.array([1, 2, 3], np.uint8))
    rec = pc.reshape(-1, 32)
    f = rec[:, :12].copy().view(np.float32).reshape(-1, 3)
    assert np.isnan(f[(d16 == -16).ravel()]).all() and not np.isnan(f[(d16 != -16).ravel()]).any()
    assert (rec[:, 16:19] == [1, 2, 3]).all() and (rec[:, 12:16] == 0).all() and (rec[:, 19:] == 0).all()
