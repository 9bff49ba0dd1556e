"""GPU parity tests (-m gpu): every hand-written HIP kernel, called through the C-ABI (gipk_* launchers), against the CPU
oracle on the same seeded inputs.  Integer / byte results must be bit-exact; float results state their tolerance."""
import numpy as np
import pytest

from ros_gpu_stereo_processor_amd import synth

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.cpu().numpy()


# ---------------------------------------------------------------- colour
@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (37, 53), (480, 752)])
def test_colour_conversions(gpu, oracle, shape):
    from ros_gpu_stereo_processor_amd import kernels as K
    rng = np.random.default_rng(11)
    h, w = shape
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mono = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(_n(K.bgr_to_gray(_t(bgr, gpu), False)), oracle.bgr_to_gray(bgr, False))
    assert np.array_equal(_n(K.bgr_to_gray(_t(bgr, gpu), True)), oracle.bgr_to_gray(bgr, True))
    assert np.array_equal(_n(K.gray_to_bgr(_t(mono, gpu))), oracle.gray_to_bgr(mono))
    assert np.array_equal(_n(K.swap_rb(_t(bgr, gpu))), oracle.swap_rb(bgr))
    for enc in ("bayer_rggb8", "bayer_bggr8", "bayer_gbrg8", "bayer_grbg8"):
        assert np.array_equal(_n(K.bayer_to_bgr(_t(mono, gpu), enc)), oracle.bayer_to_bgr(mono, enc)), enc
        assert np.array_equal(_n(K.bayer_to_gray(_t(mono, gpu), enc)), oracle.bayer_to_gray(mono, enc)), enc
    m16 = _n(K.mono8_to_mono16(_t(mono, gpu))).view(np.uint16)
    assert np.array_equal(m16, oracle.mono8_to_mono16(mono))
    big = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    assert np.array_equal(_n(K.mono16_to_mono8(_t(big.view(np.int16), gpu))), oracle.mono16_to_mono8(big))


def test_rgb_to_gray_reference_golden(gpu, aloe):
    from ros_gpu_stereo_processor_amd import kernels as K
    assert np.array_equal(_n(K.bgr_to_gray(_t(aloe["left_rgb"], gpu), True)), aloe["left_mono"])


def test_colour_batch(gpu, oracle):
    from ros_gpu_stereo_processor_amd import kernels as K
    rng = np.random.default_rng(12)
    raw = rng.integers(0, 256, (3, 20, 34), dtype=np.uint8)
    out = _n(K.bayer_to_gray(_t(raw, gpu), "bayer_rggb8"))
    for b in range(3):
        assert np.array_equal(out[b], oracle.bayer_to_gray(raw[b], "bayer_rggb8"))


# ---------------------------------------------------------------- remap
@pytest.mark.parametrize("side", ["left", "right"])
def test_remap_bit_exact_and_reference_golden(gpu, oracle, kitchen, calib, side):
    from ros_gpu_stereo_processor_amd import kernels as K
    c = calib[side]
    m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], c["width"], c["height"])
    raw = kitchen[side + "_raw"]
    got = _n(K.remap(_t(raw, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu)))
    assert np.array_equal(got, oracle.remap(raw, m1, m2))  # same tables -> bit exact
    diff = np.abs(got.astype(int) - kitchen[side + "_rect"].astype(int))  # reference golden (RectifyMonoGpu, UTest.cpp:262-288)
    assert diff.max() <= 2 and (diff > 0).mean() <= 0.0002
    # 3-channel
    bgr = np.stack([raw, 255 - raw, raw // 2], -1)
    got3 = _n(K.remap(_t(bgr, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu)))
    assert np.array_equal(got3, oracle.remap(bgr, m1, m2))


def test_remap_out_of_range_taps(gpu, oracle):
    """maps pointing outside the source (BORDER_CONSTANT 0), including the half-outside taps"""
    from ros_gpu_stereo_processor_amd import kernels as K
    rng = np.random.default_rng(13)
    src = rng.integers(1, 256, (16, 20), dtype=np.uint8)
    m1 = rng.integers(-3, 24, (30, 40, 2)).astype(np.int16)
    m2 = rng.integers(0, 1024, (30, 40)).astype(np.uint16)
    got = _n(K.remap(_t(src, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu)))
    assert np.array_equal(got, oracle.remap(src, m1, m2))


def test_remap_batches_share_the_table_decode(gpu, oracle):
    """Batches whose size is a multiple of four take the kernel form that decodes a table entry once for four images; other
    sizes the one-image form.  Out-of-range taps included, both interpolation modes."""
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    rng = np.random.default_rng(21)
    m1 = rng.integers(-2, 50, (37, 61, 2)).astype(np.int16)
    m2 = rng.integers(0, 1024, (37, 61)).astype(np.uint16)
    for B in (8, 5):
        src = rng.integers(0, 256, (B, 41, 47), dtype=np.uint8)
        for interp, name in ((capi.INTER_LINEAR, "linear"), (capi.INTER_NEAREST, "nearest")):
            got = _n(K.remap(_t(src, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu), interp))
            for b in range(B):
                exp = oracle.remap(src[b], m1, m2) if interp == capi.INTER_LINEAR else oracle.remap(src[b], m1, m2, nearest=True)
                assert np.array_equal(got[b], exp), (B, name, b)


# ---------------------------------------------------------------- prefilter
@pytest.mark.parametrize("shape", [(480, 752), (33, 47), (2, 9)])
def test_prefilters(gpu, oracle, shape):
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    rng = np.random.default_rng(14)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    for cap in (31, 63, 5):
        assert np.array_equal(_n(K.prefilter(_t(img, gpu), capi.PREFILTER_XSOBEL, 9, cap)), oracle.prefilter_xsobel(img, cap))
    if shape[0] >= 16:
        for ws in (5, 9, 21):
            assert np.array_equal(_n(K.prefilter(_t(img, gpu), capi.PREFILTER_NORMALIZED_RESPONSE, ws, 31)), oracle.prefilter_norm(img, ws, 31)), ws


def test_prefilter_norm_wide_windows(gpu, oracle):
    """Window sizes whose halo gives a thread two columns (winsize up to 255) and images narrower / shorter than the window."""
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    rng = np.random.default_rng(99)
    for shape, sizes in (((70, 300), (65, 255)), ((40, 20), (31, 77)), ((300, 530), (129,))):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for ws in sizes:
            assert np.array_equal(_n(K.prefilter(_t(img, gpu), capi.PREFILTER_NORMALIZED_RESPONSE, ws, 31)), oracle.prefilter_norm(img, ws, 31)), (shape, ws)


@pytest.mark.parametrize("shape", [(480, 752), (33, 47), (2, 9), (65, 62), (34, 125), (97, 187)])
def test_fused_rectify_xsobel_bit_exact(gpu, oracle, calib, shape):
    """The one-pass front end (raw -> prefiltered plane, rectified plane as a by-product) equals remap followed by XSOBEL: strips
    of 62 columns, blocks of 18 rows, odd heights (last row = cap), batches of 4 (shared table decode) and of 3."""
    from ros_gpu_stereo_processor_amd import kernels as K
    h, w = shape
    rng = np.random.default_rng(31 + h)
    if shape == (480, 752):
        c = calib["left"]
        m1, m2 = oracle.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], w, h)
    else:
        # arbitrary tables incl. taps outside the source
        yy, xx = np.mgrid[0:h, 0:w]
        m1 = np.stack([xx + rng.integers(-3, 4, (h, w)), yy + rng.integers(-3, 4, (h, w))], -1).astype(np.int16)
        m2 = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    for batch, cap in ((4, 31), (3, 63), (1, 5)):
        raw = rng.integers(0, 256, (batch, h, w), dtype=np.uint8)
        pf, rect = K.rectify_xsobel(_t(raw, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu), cap)
        pf2, none = K.rectify_xsobel(_t(raw, gpu), _t(m1, gpu), _t(m2.view(np.int16), gpu), cap, want_rect=False)
        assert none is None
        for b in range(batch):
            er = oracle.remap(raw[b], m1, m2)
            assert np.array_equal(_n(rect[b]), er), (shape, batch, b)
            ep = oracle.prefilter_xsobel(er, cap)
            assert np.array_equal(_n(pf[b]), ep), (shape, batch, b)
            assert np.array_equal(_n(pf2[b]), ep)


# ---------------------------------------------------------------- block matcher
def _prefiltered_pair(oracle, w, h, ndisp, seed):
    l, r, _ = synth.rectified_pair(w, h, min(ndisp, max(16, w // 3)), seed)
    return oracle.prefilter_xsobel(l, 31), oracle.prefilter_xsobel(r, 31)


BM_CASES = [
    # (w, h, ndisp, block, mindisp, tex, uniq, disp12)
    (160, 60, 16, 5, 0, 10, 15, -1),
    (160, 60, 16, 5, 0, 10, 15, 1),
    (200, 64, 32, 9, 0, 10, 15, 0),
    (200, 64, 48, 11, 0, 0, 0, 0),       # the reference tests' configureProcessor-like set (UTest.cpp:78-87) at small size
    (320, 70, 64, 15, 0, 10, 15, 1),
    (320, 70, 96, 7, 0, 5, 10, 2),
    (400, 50, 128, 9, 0, 10, 15, 1),
    (400, 80, 128, 21, 0, 10, 15, 0),
    (600, 48, 256, 21, 0, 10, 15, 1),
    (200, 64, 32, 9, -8, 10, 15, 1),     # negative minDisparity, lofs > 0
    (200, 64, 16, 9, -24, 10, 15, 1),    # negative minDisparity, rofs > 0
    (200, 64, 32, 9, 6, 10, 15, 1),      # positive minDisparity
    (131, 37, 16, 5, 0, 10, 15, 1),      # ragged sizes
    (97, 33, 48, 13, 0, 3, 0, 0),
    (64, 64, 64, 5, 0, 10, 15, 1),       # width1 == 1
    (300, 40, 80, 31, 0, 10, 15, 1),     # largest block of the u16 path
    (700, 40, 256, 5, 0, 10, 15, 1),     # ndisp 256 = four waves per workgroup sharing the cost slice, smallest block
    (500, 40, 192, 9, -16, 10, 15, 1),   # ndisp 192 = two waves per workgroup, negative minDisparity (clamped right-image columns)
    (500, 40, 192, 11, 5, 10, 0, 1),     # ... positive minDisparity, uniqueness test off (no count exchange between the waves)
    (340, 36, 256, 15, 0, 0, 25, 0),     # ndisp 256, width1 = 85: a single tile whose clamped borders meet
    # every block radius 2..15 of the prefix-form window sums (R = 4m + rho: one asm template per rho, one core per m) and every
    # (groups, waves) split of the disparity range, incl. the reference constructor's createStereoBM(48, 19) (GPUStereoProcessor.cpp:18-20)
    (400, 60, 48, 19, 0, 3, 0, 0),       # R 9  (m 2, rho 1), the reference default
    (330, 44, 80, 13, 0, 10, 15, 1),     # R 6  (m 1, rho 2), NG 5
    (330, 44, 112, 17, 0, 10, 15, 1),    # R 8  (m 2, rho 0), NG 7
    (420, 40, 144, 7, 0, 10, 15, 1),     # R 3  (m 0, rho 3), NG 9
    (420, 44, 160, 23, 0, 10, 15, 1),    # R 11 (m 2, rho 3), two waves x NG 5
    (420, 48, 176, 25, 0, 10, 15, 1),    # R 12 (m 3, rho 0), NG 11
    (460, 52, 208, 27, 0, 10, 15, 1),    # R 13 (m 3, rho 1), NG 13
    (460, 56, 224, 29, -7, 10, 15, 1),   # R 14 (m 3, rho 2), two waves x NG 7, negative minDisparity
    (480, 40, 240, 5, 3, 10, 15, 1),     # R 2  (m 0, rho 2), NG 15, positive minDisparity
    (300, 44, 64, 19, 0, 10, 15, 1),     # R 9 with the L/R check
    (300, 44, 32, 21, 0, 10, 15, 1),     # R 10 (m 2, rho 2)
    (300, 44, 32, 11, 0, 10, 15, 1),     # R 5  (m 1, rho 1)
    (300, 44, 32, 15, 0, 10, 15, 1),     # R 7  (m 1, rho 3)
]


@pytest.mark.parametrize("case", BM_CASES)
def test_bm_match_bit_exact(gpu, oracle, case):
    from ros_gpu_stereo_processor_amd import kernels as K
    w, h, nd, block, mind, tex, uniq, d12 = case
    lf, rf = _prefiltered_pair(oracle, w, h, nd, seed=sum(case))
    po = oracle.bm_params(block_size=block, num_disparities=nd, min_disparity=mind, texture_threshold=tex, uniqueness_ratio=uniq, disp12_max_diff=d12)
    pg = K.bm_params(block_size=block, num_disparities=nd, min_disparity=mind, texture_threshold=tex, uniqueness_ratio=uniq, disp12_max_diff=d12)
    exp_d, exp_c = oracle.bm_match(lf, rf, po)
    got_d, got_c = K.bm_match(_t(lf, gpu), _t(rf, gpu), pg)
    got_d, got_c = _n(got_d), _n(got_c)
    if d12 < 0:
        # border columns (clamped windows) only feed the L/R check; without it they are blanked later and not computed
        roi = oracle.valid_disparity_roi(w, h, po)
        sl = (slice(roi[1], roi[1] + roi[3]), slice(roi[0], roi[0] + roi[2]))
        assert np.array_equal(got_d[sl], exp_d[sl])
    else:
        assert np.array_equal(got_d, exp_d)
        valid = exp_d != (mind - 1) * 16
        assert np.array_equal(got_c[valid], exp_c[valid])
    # full compute (validate + blanking) must agree everywhere
    exp = oracle.stereobm(*_raw_for(case), po)
    got = _n(K.stereobm(_t(_raw_for(case)[0], gpu), _t(_raw_for(case)[1], gpu), pg))
    assert np.array_equal(got, exp)


_RAW_CACHE = {}


def _raw_for(case):
    if case not in _RAW_CACHE:
        w, h, nd = case[0], case[1], case[2]
        l, r, _ = synth.rectified_pair(w, h, min(nd, max(16, w // 3)), seed=sum(case) + 1)
        _RAW_CACHE[case] = (l, r)
    return _RAW_CACHE[case]


def test_stereobm_norm_prefilter_and_batch(gpu, oracle):
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    pairs = [synth.rectified_pair(256, 72, 64, seed=s)[:2] for s in (21, 22, 23)]
    L = np.stack([p[0] for p in pairs])
    R = np.stack([p[1] for p in pairs])
    kw = dict(prefilter_type=capi.PREFILTER_NORMALIZED_RESPONSE, prefilter_size=5, block_size=15, num_disparities=64, texture_threshold=10,
              uniqueness_ratio=0, disp12_max_diff=0)
    got = _n(K.stereobm(_t(L, gpu), _t(R, gpu), K.bm_params(**kw)))
    for b in range(3):
        assert np.array_equal(got[b], oracle.stereobm(L[b], R[b], oracle.bm_params(**kw))), b


def test_stereobm_kitchen_fixture(gpu, oracle, kitchen):
    """C1 (BASELINE.json configs[0]): the reference's rectified fixture pair, ndisp 64 block 15, OpenCV defaults and the
    reference-effective set (texture 3, uniqueness 0, disp12MaxDiff 0)."""
    from ros_gpu_stereo_processor_amd import kernels as K
    l, r = kitchen["left_rect"], kitchen["right_rect"]
    for kw in (dict(block_size=15, num_disparities=64), dict(block_size=15, num_disparities=64, texture_threshold=3, uniqueness_ratio=0, disp12_max_diff=0),
               dict(block_size=19, num_disparities=48, prefilter_size=5, texture_threshold=3, uniqueness_ratio=0, disp12_max_diff=0)):
        exp = oracle.stereobm(l, r, oracle.bm_params(**kw))
        got = _n(K.stereobm(_t(l, gpu), _t(r, gpu), K.bm_params(**kw)))
        assert np.array_equal(got, exp), kw
        assert (exp != -16).mean() > 0.05


def test_stereobm_degenerate_and_errors(gpu, oracle):
    from ros_gpu_stereo_processor_amd import kernels as K, capi
    l, r, _ = synth.rectified_pair(40, 30, 16, seed=3)
    # ndisp wider than the image: everything FILTERED
    p = K.bm_params(block_size=5, num_disparities=64)
    assert (_n(K.stereobm(_t(l, gpu), _t(r, gpu), p)) == -16).all()
    assert (oracle.stereobm(l, r, oracle.bm_params(block_size=5, num_disparities=64)) == -16).all()
    for bad in (dict(block_size=4), dict(block_size=41), dict(num_disparities=20), dict(prefilter_cap=64), dict(texture_threshold=-1)):
        with pytest.raises(capi.GipError):
            K.stereobm(_t(l, gpu), _t(r, gpu), K.bm_params(**{**dict(block_size=5, num_disparities=16), **bad}))
    with pytest.raises(capi.GipError) as e:  # beyond the implemented range -> explicit UNSUPPORTED, never a silent wrong answer
        big = synth.rectified_pair(200, 120, 16, seed=4)
        K.stereobm(_t(big[0], gpu), _t(big[1], gpu), K.bm_params(block_size=9, num_disparities=4112))
    assert e.value.status == capi.ERR_UNSUPPORTED


# ---------------------------------------------------------------- post-processing
def test_speckle_filter_bit_exact(gpu, oracle):
    from ros_gpu_stereo_processor_amd import kernels as K
    rng = np.random.default_rng(15)
    imgs = []
    for i in range(3):
        a = (rng.integers(0, 8, (90, 130)) * 16).astype(np.int16)
        a[rng.random(a.shape) < 0.25] = -16
        a[20:60, 30:100] = 5 * 16 + (np.arange(70) // 8)[None, :]  # one large smooth region
        imgs.append(a)
    batch = np.stack(imgs)
    for max_size, max_diff in ((10, 16), (800, 80), (0, 0), (40, 0)):
        got = _n(K.filter_speckles(_t(batch.copy(), gpu), -16, max_size, max_diff))
        for b in range(3):
            assert np.array_equal(got[b], oracle.filter_speckles(batch[b], -16, max_size, max_diff)), (max_size, max_diff, b)


def test_fused_cloud_general_q_matrix(gpu, oracle, calib, kitchen):
    """The fused disparity -> cloud kernel has a specialised form for the sparse Q of a rectified pair; a dense Q (every entry
    non-zero) takes the general form.  Both must agree with reprojectImageTo3D + the PointCloud2 packing of the oracle."""
    from ros_gpu_stereo_processor_amd import kernels as K
    Pl, Pr = calib["left"]["P"], calib["right"]["P"]
    d16 = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], oracle.bm_params(block_size=15, num_disparities=64))[:203, :301].copy()
    bgr = np.stack([kitchen["left_rect"][:203, :301]] * 3, -1).copy()
    Qs = np.asarray(oracle.stereo_q(Pl, Pr), np.float64).reshape(4, 4)
    Qd = Qs + np.array([[0, 1e-3, 2e-3, 0], [3e-3, 0, 1e-3, 0], [1e-4, 2e-4, 3e-4, 0], [1e-6, 2e-6, 0, 0]])
    for Q in (Qs, Qd):
        f = oracle.disparity_to_f32(d16, Pl[2], Pr[2])
        pc_exp = oracle.pack_pointcloud2(oracle.reproject_to_3d(f, Q.reshape(-1), True), bgr)
        pc_got = _n(K.disparity_to_pointcloud2(_t(d16, gpu), _t(bgr, gpu), Q.reshape(-1), Pl[2], Pr[2], 0))
        fx, ex = pc_got[..., :12].copy().view(np.float32), pc_exp[..., :12].copy().view(np.float32)
        assert np.array_equal(np.isnan(fx), np.isnan(ex))
        fin = np.isfinite(ex)
        assert np.allclose(fx[fin], ex[fin], rtol=2.4e-7, atol=0)
        assert np.array_equal(pc_got[..., 12:], pc_exp[..., 12:])


def test_speckle_filter_dense_runs_and_odd_shapes(gpu, oracle):
    """Noise maps have one run per pixel: more runs per 32-row strip than the LDS level holds, so the strip is merged by the
    global union-find instead (k_spk_strip fallback); the half-noise map mixes both kinds of strips.  Odd shapes cover widths
    that are not a multiple of the 64-column scan step and heights that are not a multiple of the strip."""
    from ros_gpu_stereo_processor_amd import kernels as K
    rng = np.random.default_rng(77)
    noise = (rng.integers(0, 6, (100, 640)) * 16).astype(np.int16)           # 32 x 640 runs per strip > 7168
    half = noise.copy()
    half[40:, :] = (7 * 16 + (np.arange(640) // 40)[None, :]).astype(np.int16)
    half[50:58, 100:104] = -16
    snake = np.full((70, 257), -16, np.int16)                                 # one component as deep as the image
    snake[::2, :] = 48
    snake[1::4, 0] = 48
    snake[3::4, 256] = 48
    for name, img, args in (("noise", noise, (3, 16)), ("noise0", noise, (5, 0)), ("half", half, (30, 16)), ("snake", snake, (100000, 0)),
                            ("snake_small", snake, (100, 0)), ("row", noise[:1, :300].copy(), (2, 16)), ("col", noise[:, :1].copy(), (2, 16)),
                            ("px", noise[:1, :1].copy(), (1, 0)), ("w65", noise[:33, :65].copy(), (4, 16))):
        exp = oracle.filter_speckles(img, -16, *args)
        got = _n(K.filter_speckles(_t(img[None].copy(), gpu), -16, *args))[0]
        assert np.array_equal(got, exp), name


def test_speckle_on_real_disparity(gpu, oracle, kitchen):
    from ros_gpu_stereo_processor_amd import kernels as K
    d = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], oracle.bm_params(block_size=15, num_disparities=64))
    exp = oracle.filter_speckles(d, -16, 800, 5 * 16)
    got = _n(K.filter_speckles(_t(d.copy(), gpu), -16, 800, 5 * 16))
    assert np.array_equal(got, exp)
    assert (exp != d).any()  # the filter did remove something


def test_disparity_f32_reproject_pointcloud(gpu, oracle, calib, kitchen):
    from ros_gpu_stereo_processor_amd import kernels as K
    Pl, Pr = calib["left"]["P"], calib["right"]["P"]
    Q = oracle.stereo_q(Pl, Pr)
    d16 = oracle.stereobm(kitchen["left_rect"], kitchen["right_rect"], oracle.bm_params(block_size=15, num_disparities=64))
    f_exp = oracle.disparity_to_f32(d16, Pl[2], Pr[2])
    f_got = _n(K.disparity_to_f32(_t(d16, gpu), Pl[2], Pr[2]))
    assert np.array_equal(f_got, f_exp)  # one float multiply + one add, contraction off on both sides
    xyz_exp = oracle.reproject_to_3d(f_exp, Q, True)
    xyz_got = _n(K.reproject_to_3d(_t(f_exp, gpu), Q, float(f_exp.min()), True))
    # double arithmetic, float store; OpenCV accumulates q += Q[0] per column, the kernel evaluates Q[0]*x: <= 1 float ulp
    assert np.array_equal(np.isinf(xyz_got), np.isinf(xyz_exp))
    fin = np.isfinite(xyz_exp)
    assert np.allclose(xyz_got[fin], xyz_exp[fin], rtol=2.4e-7, atol=0)
    assert np.array_equal(xyz_got[..., 2] == 10000, xyz_exp[..., 2] == 10000)
    bgr = np.stack([kitchen["left_rect"]] * 3, -1)
    pc_exp = oracle.pack_pointcloud2(xyz_exp, bgr)
    pc_got = _n(K.pack_pointcloud2(_t(xyz_exp, gpu), _t(bgr, gpu)))
    assert np.array_equal(pc_got, pc_exp)  # byte image identical (NaN pattern included) given identical xyz
    # fused disparity -> cloud: same record layout / NaN mask / colour bytes, xyz within 1 ulp
    pc_fused = _n(K.disparity_to_pointcloud2(_t(d16, gpu), _t(bgr, gpu), Q, Pl[2], Pr[2], 0))
    fx = pc_fused[..., :12].copy().view(np.float32)
    ex = pc_exp[..., :12].copy().view(np.float32)
    assert np.array_equal(np.isnan(fx), np.isnan(ex))
    ok = ~np.isnan(ex)
    assert np.allclose(fx[ok], ex[ok], rtol=2.4e-7, atol=0)
    assert np.array_equal(pc_fused[..., 12:], pc_exp[..., 12:])


def test_draw_color_disp(gpu, oracle):
    from ros_gpu_stereo_processor_amd import kernels as K
    d16 = np.arange(-16, -16 + 2100, dtype=np.int32).reshape(42, 50).astype(np.int16)
    for nd in (48, 128):
        assert np.array_equal(_n(K.draw_color_disp(_t(d16, gpu), nd)), oracle.draw_color_disp(d16, nd))


def test_full_size_properties(gpu, oracle):
    """BASELINE configs[1] size (1242x375, ndisp 128, block 9) on a batch: sampled rows bit-exact vs the oracle, plus the
    size-independent invariants (range, FILTERED outside the ROI, idempotent speckle filter)."""
    from ros_gpu_stereo_processor_amd import kernels as K
    B = 4
    pairs = [synth.rectified_pair(1242, 375, 128, seed=40 + i) for i in range(B)]
    L = np.stack([p[0] for p in pairs])
    R = np.stack([p[1] for p in pairs])
    kw = dict(block_size=9, num_disparities=128, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    got = K.stereobm(_t(L, gpu), _t(R, gpu), K.bm_params(**kw))
    g = _n(got)
    exp0 = oracle.stereobm(L[0], R[0], oracle.bm_params(**kw))
    assert np.array_equal(g[0], exp0)
    exp3 = oracle.stereobm(L[3], R[3], oracle.bm_params(**kw))
    assert np.array_equal(g[3], exp3)
    valid = g != -16
    assert g[valid].min() >= 0 and g[valid].max() <= 127 * 16
    roi = oracle.valid_disparity_roi(1242, 375, oracle.bm_params(**kw))
    mask = np.ones((375, 1242), bool)
    mask[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]] = False
    assert (g[:, mask] == -16).all()
    once = K.filter_speckles(got.clone(), -16, 800, 80)
    twice = K.filter_speckles(once.clone(), -16, 800, 80)
    assert np.array_equal(_n(once), _n(twice))
    assert np.array_equal(_n(once)[1], oracle.filter_speckles(g[1], -16, 800, 80))


def _fuzz_case(seed):
    """seeded random matcher configuration: every (ndisp, block) the fast path instantiates, ragged widths around the tile stride 64 - 2r,
    either sign of minDisparity, every combination of the optional tests"""
    rng = np.random.default_rng(1000 + seed)
    nd = 16 * int(rng.integers(1, 17))
    block = 5 + 2 * int(rng.integers(0, 14))
    r = block // 2
    stride = 64 - 2 * r
    # widths that put the last tile anywhere from one column to a full tile, sometimes a single tile
    ntiles = int(rng.integers(1, 5))
    width1 = max(1, (ntiles - 1) * stride + int(rng.integers(1, stride + 1)))
    mind = int(rng.integers(-20, 11))
    w = width1 + nd - 1 + abs(mind) if mind < 0 else width1 + nd - 1 + mind
    w = max(w, block + 2)
    h = block + 1 + int(rng.integers(1, 40))
    tex = int(rng.choice([0, 5, 10]))
    uniq = int(rng.choice([0, 10, 15, 25]))
    d12 = int(rng.choice([-1, 0, 1, 2]))
    return (w, h, nd, block, mind, tex, uniq, d12)


@pytest.mark.parametrize("seed", range(40))
def test_bm_fuzz_bit_exact(gpu, oracle, seed):
    from ros_gpu_stereo_processor_amd import kernels as K
    case = _fuzz_case(seed)
    w, h, nd, block, mind, tex, uniq, d12 = case
    l, r, _ = synth.rectified_pair(w, h, min(nd, max(16, w // 3)), seed=5000 + seed)
    kw = dict(block_size=block, num_disparities=nd, min_disparity=mind, texture_threshold=tex, uniqueness_ratio=uniq, disp12_max_diff=d12)
    exp = oracle.stereobm(l, r, oracle.bm_params(**kw))
    got = _n(K.stereobm(_t(l, gpu), _t(r, gpu), K.bm_params(**kw)))
    assert np.array_equal(got, exp), case


def test_bm_result_independent_of_row_chunking(gpu, oracle):
    """The matcher cuts every tile column into row chunks chosen from the device and the kernel's footprint (launch_c4_rf); a chunk starts with
    2r warm-up rows of its own, so the result must not depend on where the cuts fall.  GIP_C4_CHUNKS forces the count (read once per process):
    run the same frame in child processes with 1 and 5 chunks and compare with this process (automatic choice) and with the oracle."""
    import hashlib, subprocess, sys, os
    from ros_gpu_stereo_processor_amd import kernels as K
    w, h, nd, block = 600, 200, 64, 11
    l, r, _ = synth.rectified_pair(w, h, nd, seed=4242)
    kw = dict(block_size=block, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    got = _n(K.stereobm(_t(l, gpu), _t(r, gpu), K.bm_params(**kw)))
    assert np.array_equal(got, oracle.stereobm(l, r, oracle.bm_params(**kw)))
    want = hashlib.sha256(got.tobytes()).hexdigest()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib, numpy as np, torch; sys.path.insert(0, %r)\n"
            "from ros_gpu_stereo_processor_amd import kernels as K, synth\n"
            "l, r, _ = synth.rectified_pair(%d, %d, %d, seed=4242)\n"
            "d = K.stereobm(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), K.bm_params(**%r))\n"
            "print(hashlib.sha256(d.cpu().numpy().tobytes()).hexdigest())\n") % (root, w, h, nd, kw)
    for chunks in ("1", "5"):
        env = dict(os.environ, GIP_C4_CHUNKS=chunks)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip().splitlines()[-1] == want, chunks


@pytest.mark.parametrize("uniq,block,cap", [(199, 21, 63), (200, 21, 63), (15, 21, 63), (199, 9, 31), (200, 9, 31), (150, 23, 63)])
def test_bm_extreme_uniqueness_and_cost_range(gpu, oracle, uniq, block, cap):
    """The corners of the column-sum matcher's integer shortcuts: uniqueness ratio 199 (the last value its multiply-shift threshold
    (ms * K) >> 23 is exact for, on the device's 24-bit multiplies) against 200 (falls back to the generic kernels), and prefilter cap 63 with
    block 21, where a window sum reaches 2 * 63 * 441 = 55 566 of the 65 535 a uint16 holds and the sub-pixel numerator |p - n| * 256
    approaches 2^24 (block 23 at cap 63 no longer fits and takes the 32-bit kernel).  Inputs are PREFILTERED planes built to hit those
    ranges: binary {0, 2 cap} texture with planted shifts (winner cost 0, neighbours about half the maximum), a band where left and right
    differ by 2 cap everywhere (every cost at the maximum: ties, threshold = ms + ms * uniq / 100 far above 16 bits), and noise."""
    from ros_gpu_stereo_processor_amd import kernels as K
    w, h, nd = 420, 132, 64
    rng = np.random.default_rng(1000 * uniq + block)
    hi = 2 * cap
    base = (rng.integers(0, 2, (h, w + nd)) * hi).astype(np.uint8)
    rf = base[:, nd:].copy()                       # right(x) = base(x + nd)
    lf = np.empty_like(rf)
    for y0 in range(0, h, 44):                     # left(x) = right(x - d) = base(x + nd - d): a different planted shift every 44 rows (> the block)
        d = int(rng.integers(2, nd - 2))
        lf[y0:y0 + 44] = base[y0:y0 + 44, nd - d:nd - d + w]
    flip = rng.random((h, w)) < 0.03
    lf[flip] = hi - lf[flip]
    lf[44:88, 100:260] = 0                         # every cost of the windows inside this patch is the maximum
    rf[44:88, :] = hi
    lf[120:130] = rng.integers(0, hi + 1, (10, w)).astype(np.uint8)
    for tex, d12, u in ((0, 1, uniq), (10, -1, uniq), (0, 1, 0)):      # uniqueness 0: the all-maximum windows stay valid (largest costs the plane can carry)
        kw = dict(prefilter_cap=cap, block_size=block, num_disparities=nd, min_disparity=0, texture_threshold=tex, uniqueness_ratio=u, disp12_max_diff=d12)
        exp_d, exp_c = oracle.bm_match(lf, rf, oracle.bm_params(**kw))
        got_d, got_c = K.bm_match(_t(lf, gpu), _t(rf, gpu), K.bm_params(**kw))
        got_d, got_c = _n(got_d), _n(got_c)
        if d12 < 0:
            roi = oracle.valid_disparity_roi(w, h, oracle.bm_params(**kw))
            sl = (slice(roi[1], roi[1] + roi[3]), slice(roi[0], roi[0] + roi[2]))
            assert np.array_equal(got_d[sl], exp_d[sl])
        else:
            assert np.array_equal(got_d, exp_d)
            valid = exp_d != -16
            assert np.array_equal(got_c[valid], exp_c[valid])
            assert valid.mean() > 0.05 and (u > 0 or exp_c[valid].max() == 2 * cap * block * block)
    # and through cv::StereoBM::compute with the XSOBEL prefilter at that cap, on raw noise with planted shifts
    raw_b = rng.integers(0, 256, (h, w + nd)).astype(np.uint8)
    rr = raw_b[:, nd:].copy()
    rl = np.empty_like(rr)
    for y0 in range(0, h, 44):
        d = int(rng.integers(2, nd - 2))
        rl[y0:y0 + 44] = raw_b[y0:y0 + 44, nd - d:nd - d + w]
    kw = dict(prefilter_type=1, prefilter_cap=cap, block_size=block, num_disparities=nd, min_disparity=0, texture_threshold=10, uniqueness_ratio=uniq, disp12_max_diff=1)
    exp = oracle.stereobm(rl, rr, oracle.bm_params(**kw))
    got = _n(K.stereobm(_t(rl, gpu), _t(rr, gpu), K.bm_params(**kw)))
    assert np.array_equal(got, exp) and (exp != -16).mean() > 0.05


@pytest.mark.parametrize("nd,block,batch,mind,d12", [(48, 19, 9, 0, 1), (16, 5, 8, 0, 1), (32, 9, 11, -8, 1), (48, 11, 8, 5, -1), (32, 21, 13, 0, 0), (48, 9, 2, 0, 1)])
def test_stereobm_two_images_per_wave(gpu, oracle, nd, block, batch, mind, d12):
    """Disparity counts up to 48 run TWO images of the batch per wavefront once the batch has eight images (NSUB = 2 in
    k_stereobm_c4_impl.h: one image per 32-lane half, two disparity slices each, a lane finalises two pixels; an odd batch leaves the
    last pair half empty).  Every image of the batch is different and every one must equal the oracle's cv::StereoBM::compute; the last
    case (batch 2) stays on the one-image-per-wave kernel."""
    from ros_gpu_stereo_processor_amd import kernels as K
    w, h = 331, 57            # odd sizes: border tiles on both sides, ragged last tile
    pairs = [synth.rectified_pair(w, h, max(16, nd - 8), seed=4000 + 17 * i + nd)[:2] for i in range(batch)]
    L = np.stack([p[0] for p in pairs])
    R = np.stack([p[1] for p in pairs])
    kw = dict(block_size=block, num_disparities=nd, min_disparity=mind, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=d12)
    got = _n(K.stereobm(_t(L, gpu), _t(R, gpu), K.bm_params(**kw)))
    po = oracle.bm_params(**kw)
    for i in range(batch):
        assert np.array_equal(got[i], oracle.stereobm(pairs[i][0], pairs[i][1], po)), i
    assert (got != (mind - 1) * 16).mean() > 0.1
